// main_topk.cpp -- drop-in executable for the call site in test_spmv_topk.py:62-64,71-82, i.e. where
// `approximate-spmv-gpu-csr-topk` (src/gpu/host_spmv_topk_csr_gpu.cu:291-480) and
// `spmv_coo_hbm_topk_multicore_mega_main` (src/fpga/src/host_spmv_bscsr.cpp:510-707) are spawned today.
// Same flags (options.hpp), same flow (load -> gold -> setup -> loop{new x, gold, reset, run, read, check}),
// same CSV schema on stdout (GPU-host field order, host_spmv_topk_csr_gpu.cu:452,466-467) or -d verbose text.
//
// The engine is reached only through the C ABI (include/tkspmv.h), the way a maintainer of the reference would
// bind it; the CPU gold computed here is the per-iteration self-check every reference main performs, not a
// fallback for the hot path.
//
// Environment (additions, the flag surface is unchanged):
//   TKSPMV_INDEX_BASE = 0 | 1 | auto   index base of the MTX file (default auto; the reference compiles in 0)
//   TKSPMV_SEED       = n              seed for the query vectors (iteration i uses n+i); default random_device
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <tuple>
#include <unordered_set>
#include <vector>

#include "../../include/tkspmv.h"
#include <sys/stat.h>

#include "host_utils.hpp"

namespace chrono = std::chrono;
using clock_type = chrono::high_resolution_clock;
using tkspmv::CooMatrix;

// Engine wrapper with the reference's four verbs.
struct SpMV {
    tkspmv_t *engine = nullptr;
    int k;

    // packed != nullptr: a packed matrix from the cache (TKSPMV_CACHE_DIR); if this GPU cannot use it (packed for a
    // larger launch geometry) the engine is built from the COO instead. cache_path non-empty and no usable packed
    // matrix: the matrix is packed here and written there for the next run.
    SpMV(const CooMatrix &m, uint32_t rows, uint32_t cols, float *vec, int k_, int debug, const tkspmv_packed *packed = nullptr,
         const std::string &cache_path = std::string(), int precision = TKSPMV_F32, int fixed_width = 0) : k(k_) {
        tkspmv_desc d{};
        d.rows = rows;
        d.cols = cols;
        d.nnz = m.nnz();
        d.row = m.row.data();
        d.col = m.col.data();
        d.val = m.val.data();
        d.k = k_;
        d.precision = precision;
        d.fixed_width = fixed_width;
        d.device = -1;
        d.min_score = 0.0f;
        bool have = false;
        if (packed) {
            have = tkspmv_create_packed(&engine, packed, &d) == TKSPMV_OK;
            if (!have && debug) std::cout << "cached packed matrix not used: " << tkspmv_last_error() << std::endl;
        }
        if (!have && !cache_path.empty()) {
            uint32_t n_parts = 0;
            tkspmv_packed *fresh = nullptr;
            if (tkspmv_wave_partitions(&d, &n_parts) == TKSPMV_OK && tkspmv_pack(&d, n_parts, &fresh) == TKSPMV_OK) {
                have = tkspmv_create_packed(&engine, fresh, &d) == TKSPMV_OK;
                if (have && tkspmv_packed_save(fresh, cache_path.c_str()) != TKSPMV_OK && debug)
                    std::cout << "could not write " << cache_path << ": " << tkspmv_last_error() << std::endl;
                tkspmv_packed_free(fresh);
            }
        }
        if (!have && tkspmv_create(&engine, &d) != TKSPMV_OK) {
            std::cerr << "engine setup failed: " << tkspmv_last_error() << std::endl;
            exit(1);
        }
        if (debug) {
            tkspmv_info info;
            tkspmv_get_info(engine, &info);
            std::cout << "packed " << info.nnz << " nnz into " << info.n_packets << " packets of "
                      << info.packet_entries << " entries (" << info.packed_bytes / 1e6 << " MB), "
                      << info.n_wave_partitions << " wave partitions; grid=" << info.grid << "x" << info.block
                      << ", " << info.num_cus << " CUs" << std::endl;
        }
        reset(vec, 0);
    }
    ~SpMV() { tkspmv_destroy(engine); }

    // Runs one query; returns the device time of the kernels in ns.
    float operator()(int debug) {
        if (debug) std::cout << "Execute the kernel" << std::endl;
        double ns = 0;
        if (tkspmv_run(engine, &ns) != TKSPMV_OK) {
            std::cerr << "kernel launch failed: " << tkspmv_last_error() << std::endl;
            exit(1);
        }
        if (debug) std::cout << "Kernel terminated\nComputation took " << ns / 1e6 << " ms" << std::endl;
        return (float)ns;
    }

    void read_result(std::vector<float> &res, std::vector<uint32_t> &res_idx, int debug = 0) {
        (void)debug;
        int32_t n = 0;
        if (tkspmv_read(engine, res_idx.data(), res.data(), &n) != TKSPMV_OK) {
            std::cerr << "read_result failed: " << tkspmv_last_error() << std::endl;
            exit(1);
        }
    }

    long reset(float *vec, int debug) {
        double ns = 0;
        if (tkspmv_set_query(engine, vec, &ns) != TKSPMV_OK) {
            std::cerr << "reset failed: " << tkspmv_last_error() << std::endl;
            exit(1);
        }
        if (debug) std::cout << "Reset took " << ns / 1e6 << " ms" << std::endl;
        return (long)ns;
    }
};

static std::tuple<float, float> sw_test(const CooMatrix &m, std::vector<uint32_t> &res_idx_sw,
                                        std::vector<float> &res_sim_sw, const float *vec, int top_k) {
    // The GPU hosts leave the full-matrix leg commented out (host_spmv_topk_csr_gpu.cu:270-273): time 0.
    float sw_time_1 = 0.0f;
    auto t0 = clock_type::now();
    tkspmv::gold_topk(m.row.data(), m.col.data(), m.val.data(), m.nnz(), vec, top_k, res_idx_sw.data(),
                      res_sim_sw.data());
    tkspmv::sort_tuples((size_t)top_k, res_idx_sw.data(), res_sim_sw.data());
    float sw_time_2 = (float)chrono::duration_cast<chrono::microseconds>(clock_type::now() - t0).count() / 1000;
    return std::make_tuple(sw_time_1, sw_time_2);
}

int main(int argc, char *argv[]) {
    tkspmv::Options options(argc, argv);
    const int debug = options.debug;
    const bool reset = options.reset;
    const int top_k_value = options.top_k_value;
    if (top_k_value < 1 || top_k_value > TKSPMV_MAX_K) {
        std::cerr << "k must be in [1, " << TKSPMV_MAX_K << "]" << std::endl;
        return 1;
    }
    // -a: the CUDA comparator's half mode (values stored as fp16, host_spmv_topk_csr_gpu.cu:132-136) -> TKSPMV_F16
    int precision = options.use_half_precision_gpu ? TKSPMV_F16 : TKSPMV_F32;
    // TKSPMV_FIXED_WIDTH=W: the FPGA builds' fixed-point real_type (the reference fixes FIXED_WIDTH at compile time,
    // types.hpp:20; its driver lists 20/21/25/26/32-bit bitstreams, test_spmv_topk.py:42-47) -> TKSPMV_FIXED
    int fixed_width = 0;
    if (const char *fw = getenv("TKSPMV_FIXED_WIDTH")) {
        fixed_width = atoi(fw);
        if (fixed_width < 8 || fixed_width > 32) {
            std::cerr << "TKSPMV_FIXED_WIDTH must be in [8, 32]" << std::endl;
            return 1;
        }
        precision = TKSPMV_FIXED;
    }

    int index_base = -1;
    if (const char *ib = getenv("TKSPMV_INDEX_BASE")) {
        std::string s(ib);
        index_base = (s == "0") ? 0 : (s == "1") ? 1 : -1;
    }
    int seed = 0;
    if (const char *sd = getenv("TKSPMV_SEED")) seed = atoi(sd);

    auto start_1 = clock_type::now();
    CooMatrix coo;
    const std::string path = options.use_sample_matrix ? tkspmv::Options::default_matrix() : options.matrix_path;
    // TKSPMV_CACHE_DIR: keep the packed matrix next to... wherever that directory is, keyed by the file's name, size
    // and the -v flag. A hit skips the MatrixMarket parser and the packer (the reference redoes both on every run);
    // the COO the software check needs is decoded from the packed matrix.
    std::string cache_path;
    tkspmv_packed *cached = nullptr;
    if (const char *dir = getenv("TKSPMV_CACHE_DIR")) {
        struct stat sb;
        if (stat(path.c_str(), &sb) == 0) {
            const size_t slash = path.find_last_of('/');
            cache_path = std::string(dir) + "/" + (slash == std::string::npos ? path : path.substr(slash + 1)) + "." +
                         std::to_string((long long)sb.st_size) + (options.ignore_matrix_values ? ".v" : "") + (options.use_half_precision_gpu ? ".h" : "") +
                         (fixed_width ? ".w" + std::to_string(fixed_width) : "") + (index_base >= 0 ? ".b" + std::to_string(index_base) : "") + ".tkspmv";
            if (tkspmv_packed_load(cache_path.c_str(), &cached) != TKSPMV_OK) cached = nullptr;
        }
    }
    if (cached) {
        tkspmv_info pi;
        tkspmv_packed_info(cached, &pi);
        coo.rows = coo.num_rows_coo = pi.rows;
        coo.cols = pi.cols;
        coo.index_base = index_base;
        coo.row.resize(pi.nnz);
        coo.col.resize(pi.nnz);
        coo.val.resize(pi.nnz);
        uint64_t n = 0;
        tkspmv_packed_decode(cached, coo.row.data(), coo.col.data(), coo.val.data(), &n);
        if (n != pi.nnz) {  // cannot happen for a file that passed its checks; fall back to the text
            tkspmv_packed_free(cached);
            cached = nullptr;
        } else if (debug) {
            std::cout << "packed matrix read from " << cache_path << std::endl;
        }
    }
    if (!cached) {
        tkspmv::IoError io = tkspmv::read_mtx(path, index_base, !options.ignore_matrix_values, false, coo);
        if (io.code) {  // the reference prints and exit(1)s (utils.hpp:486-500)
            if (io.message.rfind("File ", 0) == 0)
                std::cerr << io.message << std::endl;
            else
                std::cout << io.message << std::endl;
            return 1;
        }
    }
    const uint32_t rows = std::max(coo.rows, coo.num_rows_coo);
    const uint32_t cols = coo.cols;
    const uint64_t nnz = coo.nnz();

    std::vector<float> vec(cols);
    tkspmv::sample_vector(vec.data(), (int)cols, true, true, false, seed);
    auto loading_time = chrono::duration_cast<chrono::milliseconds>(clock_type::now() - start_1).count();
    if (debug) {
        std::cout << "loaded matrix with " << rows << " rows, " << cols << " columns and " << nnz
                  << " non-zero elements (index base " << coo.index_base << ")" << std::endl;
        std::cout << "setup time=" << loading_time << " ms" << std::endl;
    }

    std::vector<float> res_sim_sw(top_k_value, 0);
    std::vector<uint32_t> res_idx_sw(top_k_value, 0);
    std::tuple<float, float> sw_time = sw_test(coo, res_idx_sw, res_sim_sw, vec.data(), top_k_value);
    float sw_time_1 = std::get<0>(sw_time), sw_time_2 = std::get<1>(sw_time);
    if (debug) {
        std::cout << "\nsw results =" << std::endl;
        for (int i = 0; i < top_k_value; i++)
            std::cout << i << ") document " << res_idx_sw[i] << " = " << res_sim_sw[i] << std::endl;
        std::cout << "sw time, full matrix=" << sw_time_1 << " ms; sw time, top-k=" << sw_time_2 << " ms" << std::endl;
    }

    auto start_4 = clock_type::now();
    SpMV spmv(coo, rows, cols, vec.data(), top_k_value, debug, cached, cached ? std::string() : cache_path, precision, fixed_width);
    if (cached) tkspmv_packed_free(cached);
    auto gpu_setup_time = chrono::duration_cast<chrono::milliseconds>(clock_type::now() - start_4).count();
    if (debug) std::cout << "gpu setup time=" << gpu_setup_time << " ms" << std::endl;

    const unsigned num_tests = options.num_tests;
    std::vector<float> exec_times, readback_times, precision_vec;

    for (unsigned i = 0; i < num_tests; i++) {
        if (debug) std::cout << "\nIteration " << i << ")" << std::endl;
        if (reset) {
            tkspmv::sample_vector(vec.data(), (int)cols, true, false, true, seed ? seed + (int)i + 1 : 0);
            sw_time = sw_test(coo, res_idx_sw, res_sim_sw, vec.data(), top_k_value);
            sw_time_1 = std::get<0>(sw_time);
            sw_time_2 = std::get<1>(sw_time);
        }
        spmv.reset(vec.data(), debug);

        std::vector<float> hw_res(top_k_value);
        std::vector<uint32_t> hw_res_idx(top_k_value);

        auto start_5 = clock_type::now();
        float spmv_only_time = spmv(debug) / 1e6f;
        float gpu_exec_time = (float)chrono::duration_cast<chrono::nanoseconds>(clock_type::now() - start_5).count() / 1e6f;
        exec_times.push_back(gpu_exec_time);

        auto start_6 = clock_type::now();
        spmv.read_result(hw_res, hw_res_idx, debug);
        float readback_time = (float)chrono::duration_cast<chrono::nanoseconds>(clock_type::now() - start_6).count() / 1e6f;
        readback_times.push_back(readback_time);

        const int res_size = (int)hw_res_idx.size();
        const int n_cmp = std::min(top_k_value, res_size);
        int error_idx = tkspmv::check_array_equality(hw_res_idx.data(), res_idx_sw.data(), n_cmp);
        int error = tkspmv::check_array_equality(hw_res.data(), res_sim_sw.data(), n_cmp, 10e-6f);
        std::unordered_set<uint32_t> s(res_idx_sw.begin(), res_idx_sw.end());
        int inter = (int)std::count_if(hw_res_idx.begin(), hw_res_idx.end(), [&](uint32_t v) { return s.count(v) != 0; });
        precision_vec.push_back((float)inter / (float)top_k_value);

        if (debug) {
            std::cout << "sw results =" << std::endl;
            for (int j = 0; j < top_k_value; j++)
                std::cout << j << ") document " << res_idx_sw[j] << " = " << res_sim_sw[j] << std::endl;
            std::cout << "hw results=" << std::endl;
            for (int j = 0; j < n_cmp; j++)
                std::cout << j << ") document " << hw_res_idx[j] << " = " << hw_res[j] << std::endl;
            std::cout << "num errors on indices=" << error_idx << std::endl;
            std::cout << "num errors on values=" << error << std::endl;
            std::cout << "precision=" << precision_vec.back() << std::endl;
            std::cout << "gpu exec time=" << gpu_exec_time << " ms" << std::endl;
        } else {
            if (i == 0)
                std::cout << "iteration,error_idx,error_val,sw_full_time_ms,sw_topk_time_ms,hw_setup_time_ms,"
                             "hw_spmv_only_time_ms,hw_exec_time_ms,readback_time_ms,k,sw_res_idx,sw_res_val,"
                             "hw_res_idx,hw_res_val"
                          << std::endl;
            std::string sw_i, sw_v, hw_i, hw_v;
            for (size_t j = 0; j < res_idx_sw.size(); j++) {
                const char *sep = (j + 1 < res_idx_sw.size()) ? ";" : "";
                sw_i += std::to_string(res_idx_sw[j]) + sep;
                sw_v += std::to_string(res_sim_sw[j]) + sep;
            }
            for (size_t j = 0; j < hw_res_idx.size(); j++) {
                const char *sep = (j + 1 < hw_res_idx.size()) ? ";" : "";
                hw_i += std::to_string(hw_res_idx[j]) + sep;
                hw_v += std::to_string(hw_res[j]) + sep;
            }
            std::cout << i << "," << error_idx << "," << error << "," << sw_time_1 << "," << sw_time_2 << ","
                      << gpu_setup_time << "," << spmv_only_time << "," << gpu_exec_time << "," << readback_time << ","
                      << top_k_value << "," << sw_i << "," << sw_v << "," << hw_i << "," << hw_v << std::endl;
        }
    }
    if (debug) {
        auto old_precision = std::cout.precision();
        std::cout.precision(4);
        std::cout << "----------------" << std::endl;
        std::cout << "Mean GPU execution time=" << tkspmv::mean(exec_times, 2) << "±" << tkspmv::st_dev(exec_times, 2)
                  << " ms" << std::endl;
        std::cout << "Mean read-back time=" << tkspmv::mean(readback_times, 2) << "±"
                  << tkspmv::st_dev(readback_times, 2) << " ms" << std::endl;
        std::cout << "Mean precision=" << tkspmv::mean(precision_vec, 2) << "±" << tkspmv::st_dev(precision_vec, 2)
                  << std::endl;
        std::cout << "----------------" << std::endl;
        std::cout.precision(old_precision);
    }
    return 0;
}
