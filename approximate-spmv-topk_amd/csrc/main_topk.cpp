// main_topk.cpp -- drop-in executable for the call site in test_spmv_topk.py:62-64,71-82, i.e. where
// `approximate-spmv-gpu-csr-topk` (src/gpu/host_spmv_topk_csr_gpu.cu:291-480) and
// `spmv_coo_hbm_topk_multicore_mega_main` (src/fpga/src/host_spmv_bscsr.cpp:510-707) are spawned today.
// Contract kept: the flags of options.hpp, the flow (load -> gold -> setup -> per test {new x, gold, reset, run, read,
// check}), the CSV schema on stdout in the GPU hosts' field order (host_spmv_topk_csr_gpu.cu:452,466-467) and the -d
// verbose text the plotting scripts do not parse. Organisation (matrix source, engine wrapper, checker, reporters) is
// this repository's own.
//
// The engine is reached only through the C ABI (include/tkspmv.h), the way a maintainer of the reference would bind it;
// the CPU gold computed here is the per-iteration self-check every reference main performs, not a fallback for the hot
// path.
//
//   -i N   engine variant (the reference's GPU_IMPL selector, options.hpp:35): 0 = fused streaming kernel (default),
//          1 = one row per lane over the wave-sliced ELL copy, 2 = full y = A.x + radix select (the reference GPU host's
//          structure: SpMV, then a selection over all rows). Same index lists from all of them.
//   -a     fp16 values (the comparator's half mode)
//
// Environment (additions, the flag surface is unchanged):
//   TKSPMV_INDEX_BASE = 0 | 1 | auto   index base of the MTX file. Default 0: what every reference main compiles in
//                                      (readMtx(..., zero_indexed_file = true)). Files written by create_matrices.py are
//                                      one-based: pass 1 (or auto: 0 if index 0 occurs anywhere, else 1) for those.
//   TKSPMV_SEED       = n              seed for the query vectors (iteration i uses n+i); default random_device
//   TKSPMV_FIXED_WIDTH= W              the FPGA builds' fixed-point real_type of W bits (8..32)
//   TKSPMV_CACHE_DIR  = dir            keep the packed matrix there between runs
//   TKSPMV_GENERATE   = rows,cols,nnz,dist,seed   no file at all: the matrix is generated in memory (dist: uniform | gamma; the
//                                      distributions of create_matrices.py) -- the reference's grid reaches 15M rows x 40 non-zeros,
//                                      7 GB of MatrixMarket text per matrix (test_spmv_topk.py:12-47); -m is ignored
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/tkspmv.h"
#include "host_utils.hpp"

namespace {

using Clock = std::chrono::high_resolution_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

[[noreturn]] void die(const std::string &what) {
    std::cerr << what << ": " << tkspmv_last_error() << std::endl;
    exit(1);
}

// ---- what to run --------------------------------------------------------------------------------------------------
struct RunConfig {
    tkspmv::Options opt;
    int precision = TKSPMV_F32;
    int fixed_width = 0;
    int impl = TKSPMV_IMPL_STREAM;
    int index_base = 0;
    int seed = 0;
    std::string cache_dir;

    RunConfig(int argc, char **argv) : opt(argc, argv) {
        if (opt.top_k_value < 1 || opt.top_k_value > TKSPMV_MAX_K) {
            std::cerr << "k must be in [1, " << TKSPMV_MAX_K << "]" << std::endl;
            exit(1);
        }
        if (opt.use_half_precision_gpu) precision = TKSPMV_F16;
        if (const char *fw = getenv("TKSPMV_FIXED_WIDTH")) {
            fixed_width = atoi(fw);
            if (fixed_width < 8 || fixed_width > 32) {
                std::cerr << "TKSPMV_FIXED_WIDTH must be in [8, 32]" << std::endl;
                exit(1);
            }
            precision = TKSPMV_FIXED;
        }
        if (opt.gpu_impl < 0 || opt.gpu_impl > 2) {
            std::cerr << "-i/--gpu_impl must be 0 (streaming kernel), 1 (row per lane) or 2 (scores + select)" << std::endl;
            exit(1);
        }
        impl = opt.gpu_impl;
        if (const char *ib = getenv("TKSPMV_INDEX_BASE")) {
            const std::string s(ib);
            index_base = s == "1" ? 1 : (s == "auto" || s == "-1" ? -1 : 0);
        }
        if (const char *sd = getenv("TKSPMV_SEED")) seed = atoi(sd);
        if (const char *cd = getenv("TKSPMV_CACHE_DIR")) cache_dir = cd;
    }
    std::string matrix_path() const { return opt.use_sample_matrix ? tkspmv::Options::default_matrix() : opt.matrix_path; }
};

// ---- matrix source: the MatrixMarket text, or a packed matrix cached by an earlier run -------------------------------
struct MatrixSource {
    tkspmv::CooMatrix coo;            // what the software check runs on (decoded from the cache on a hit)
    tkspmv_packed *cached = nullptr;  // non-null on a cache hit
    std::string cache_path;           // where a freshly packed matrix should be written (empty: no caching)
    double load_ms = 0;

    // Cache key: file name, size, modification time, a hash of the absolute path, and everything that changes the packed
    // bytes (-v, value type, index base). An edited or a different same-named file can never hit a stale entry.
    static std::string cache_key(const RunConfig &c, const std::string &path, const struct stat &sb) {
        char abs[PATH_MAX];
        const std::string full = realpath(path.c_str(), abs) ? std::string(abs) : path;
        uint64_t h = 1469598103934665603ull;  // FNV-1a
        for (unsigned char ch : full) h = (h ^ ch) * 1099511628211ull;
        const size_t slash = path.find_last_of('/');
        std::ostringstream k;
        k << (slash == std::string::npos ? path : path.substr(slash + 1)) << "." << (long long)sb.st_size << "."
          << (long long)sb.st_mtim.tv_sec << "_" << (long long)sb.st_mtim.tv_nsec << "." << std::hex << h << std::dec
          << (c.opt.ignore_matrix_values ? ".v" : "") << (c.precision == TKSPMV_F16 ? ".h" : "")
          << (c.fixed_width ? ".w" + std::to_string(c.fixed_width) : "") << ".b" << c.index_base << ".tkspmv";
        return k.str();
    }

    explicit MatrixSource(const RunConfig &c) {
        const auto t0 = Clock::now();
        if (const char *g = getenv("TKSPMV_GENERATE")) {  // rows,cols,nnz,dist,seed
            unsigned long long rows = 0, cols = 0, nnz = 0, seed = 1;
            char dist[32] = "gamma";
            if (sscanf(g, "%llu,%llu,%llu,%31[a-z],%llu", &rows, &cols, &nnz, dist, &seed) < 3 || rows == 0 || cols == 0 || nnz == 0 ||
                (std::string(dist) != "uniform" && std::string(dist) != "gamma")) {
                std::cerr << "TKSPMV_GENERATE must be rows,cols,nnz[,uniform|gamma[,seed]]" << std::endl;
                exit(1);
            }
            tkspmv::generate_matrix((uint32_t)rows, (uint32_t)cols, (uint32_t)nnz, std::string(dist) == "gamma" ? 1 : 0, seed, coo);
            if (c.opt.ignore_matrix_values) std::fill(coo.val.begin(), coo.val.end(), 1.0f);
            load_ms = ms_since(t0);
            return;
        }
        const std::string path = c.matrix_path();
        struct stat sb;
        if (!c.cache_dir.empty() && stat(path.c_str(), &sb) == 0) {
            cache_path = c.cache_dir + "/" + cache_key(c, path, sb);
            if (tkspmv_packed_load(cache_path.c_str(), &cached) != TKSPMV_OK) cached = nullptr;
        }
        if (cached && !decode_cached(c)) {
            tkspmv_packed_free(cached);
            cached = nullptr;
        }
        if (cached) {
            if (c.opt.debug) std::cout << "packed matrix read from " << cache_path << std::endl;
        } else {
            const tkspmv::IoError io = tkspmv::read_mtx(path, c.index_base, !c.opt.ignore_matrix_values, false, coo);
            if (io.code) {  // the reference prints and exit(1)s (utils.hpp:486-500)
                (io.message.rfind("File ", 0) == 0 ? std::cerr : std::cout) << io.message << std::endl;
                exit(1);
            }
            if (coo.index_base == 0 && coo.nnz() && *std::max_element(coo.col.begin(), coo.col.end()) >= coo.cols) {
                std::cerr << "column index " << coo.cols << " is out of range for " << coo.cols << " columns read zero-based (the "
                          << "reference's compiled-in behaviour): the file looks one-based -- set TKSPMV_INDEX_BASE=1 (or auto)"
                          << std::endl;
                exit(1);
            }
        }
        load_ms = ms_since(t0);
    }
    ~MatrixSource() {
        if (cached) tkspmv_packed_free(cached);
    }
    uint32_t rows() const { return std::max(coo.rows, coo.num_rows_coo); }

   private:
    bool decode_cached(const RunConfig &c) {
        tkspmv_info pi;
        tkspmv_packed_info(cached, &pi);
        coo.rows = coo.num_rows_coo = pi.rows;
        coo.cols = pi.cols;
        coo.index_base = c.index_base;
        coo.row.resize(pi.nnz);
        coo.col.resize(pi.nnz);
        coo.val.resize(pi.nnz);
        uint64_t n = 0;
        tkspmv_packed_decode(cached, coo.row.data(), coo.col.data(), coo.val.data(), &n);
        return n == pi.nnz;  // (cannot fail for a file that passed its checks)
    }
};

// ---- the engine behind the reference's four verbs (struct SpMV of every reference host) ----------------------------
class SpMV {
   public:
    SpMV(const RunConfig &c, MatrixSource &src, const float *vec) : k_(c.opt.top_k_value), debug_(c.opt.debug) {
        tkspmv_desc d{};
        d.rows = src.rows();
        d.cols = src.coo.cols;
        d.nnz = src.coo.nnz();
        d.row = src.coo.row.data();
        d.col = src.coo.col.data();
        d.val = src.coo.val.data();
        d.k = k_;
        d.precision = c.precision;
        d.fixed_width = c.fixed_width;
        d.impl = c.impl;
        d.device = -1;
        bool have = false;
        if (src.cached) {  // packed for a larger launch geometry than this GPU's => built from the COO instead
            have = tkspmv_create_packed(&engine_, src.cached, &d) == TKSPMV_OK;
            if (!have && debug_) std::cout << "cached packed matrix not used: " << tkspmv_last_error() << std::endl;
        } else if (!src.cache_path.empty()) {  // pack here, keep the packed matrix for the next run
            uint32_t n_parts = 0;
            tkspmv_packed *fresh = nullptr;
            if (tkspmv_wave_partitions(&d, &n_parts) == TKSPMV_OK && tkspmv_pack(&d, n_parts, &fresh) == TKSPMV_OK) {
                have = tkspmv_create_packed(&engine_, fresh, &d) == TKSPMV_OK;
                if (have && tkspmv_packed_save(fresh, src.cache_path.c_str()) != TKSPMV_OK && debug_)
                    std::cout << "could not write " << src.cache_path << ": " << tkspmv_last_error() << std::endl;
                tkspmv_packed_free(fresh);
            }
        }
        if (!have && tkspmv_create(&engine_, &d) != TKSPMV_OK) die("engine setup failed");
        if (debug_) {
            tkspmv_info info;
            tkspmv_get_info(engine_, &info);
            std::cout << "packed " << info.nnz << " nnz into " << info.n_packets << " packets of " << info.packet_entries
                      << " entries (" << info.packed_bytes / 1e6 << " MB), " << info.n_wave_partitions
                      << " wave partitions; grid=" << info.grid << "x" << info.block << ", " << info.num_cus
                      << " CUs; engine variant " << c.impl << std::endl;
        }
        reset(vec);
    }
    ~SpMV() { tkspmv_destroy(engine_); }
    SpMV(const SpMV &) = delete;
    SpMV &operator=(const SpMV &) = delete;

    long reset(const float *vec) {  // SpMV::reset(vec): install a new query vector; ns
        double ns = 0;
        if (tkspmv_set_query(engine_, vec, &ns) != TKSPMV_OK) die("reset failed");
        if (debug_) std::cout << "Reset took " << ns / 1e6 << " ms" << std::endl;
        return (long)ns;
    }
    double operator()() {  // SpMV::operator()(debug): one query; device time of the kernels in ns
        if (debug_) std::cout << "Execute the kernel" << std::endl;
        double ns = 0;
        if (tkspmv_run(engine_, &ns) != TKSPMV_OK) die("kernel launch failed");
        if (debug_) std::cout << "Kernel terminated\nComputation took " << ns / 1e6 << " ms" << std::endl;
        return ns;
    }
    void read_result(std::vector<float> &val, std::vector<uint32_t> &idx) {  // sorted (score desc, row desc)
        int32_t n = 0;
        idx.resize(k_);
        val.resize(k_);
        if (tkspmv_read(engine_, idx.data(), val.data(), &n) != TKSPMV_OK) die("read_result failed");
    }

   private:
    tkspmv_t *engine_ = nullptr;
    int k_;
    int debug_;
};

// ---- the per-iteration self-check of every reference main ---------------------------------------------------------------
struct TopK {
    std::vector<uint32_t> idx;
    std::vector<float> val;
    explicit TopK(int k) : idx(k, 0), val(k, 0.0f) {}
};
struct GoldTimes {
    float full_ms = 0.0f;  // the GPU hosts leave the full-matrix leg commented out (host_spmv_topk_csr_gpu.cu:270-273)
    float topk_ms = 0.0f;
};
GoldTimes software_gold(const tkspmv::CooMatrix &m, const float *vec, TopK &out) {
    GoldTimes t;
    const auto t0 = Clock::now();
    tkspmv::gold_topk(m.row.data(), m.col.data(), m.val.data(), m.nnz(), vec, (int)out.idx.size(), out.idx.data(), out.val.data());
    tkspmv::sort_tuples(out.idx.size(), out.idx.data(), out.val.data());
    t.topk_ms = (float)((long long)(ms_since(t0) * 1000.0)) / 1000;  // microsecond granularity, like the reference
    return t;
}
struct Verdict {
    int error_idx, error_val;
    float precision;
};
Verdict compare(const TopK &sw, const TopK &hw) {
    const int n = (int)std::min(sw.idx.size(), hw.idx.size());
    Verdict v;
    v.error_idx = tkspmv::check_array_equality(hw.idx.data(), sw.idx.data(), n);
    v.error_val = tkspmv::check_array_equality(hw.val.data(), sw.val.data(), n, 10e-6f);
    const std::unordered_set<uint32_t> gold(sw.idx.begin(), sw.idx.end());
    const long hits = std::count_if(hw.idx.begin(), hw.idx.end(), [&](uint32_t r) { return gold.count(r) != 0; });
    v.precision = (float)hits / (float)sw.idx.size();
    return v;
}

// ---- reporters: CSV (what test_spmv_topk.py tees and the plot scripts parse) or the -d text -------------------------------
struct IterationTimes {
    GoldTimes gold;
    double setup_ms, kernel_ms, exec_ms, readback_ms;
};
template <class T>
std::string joined(const std::vector<T> &v) {
    std::string s;
    for (size_t j = 0; j < v.size(); j++) s += std::to_string(v[j]) + (j + 1 < v.size() ? ";" : "");
    return s;
}
void print_list(const char *title, const TopK &t) {
    std::cout << title << std::endl;
    for (size_t j = 0; j < t.idx.size(); j++) std::cout << j << ") document " << t.idx[j] << " = " << t.val[j] << std::endl;
}
void report_csv(unsigned it, const Verdict &v, const IterationTimes &t, const TopK &sw, const TopK &hw) {
    if (it == 0)
        std::cout << "iteration,error_idx,error_val,sw_full_time_ms,sw_topk_time_ms,hw_setup_time_ms,hw_spmv_only_time_ms,"
                     "hw_exec_time_ms,readback_time_ms,k,sw_res_idx,sw_res_val,hw_res_idx,hw_res_val"
                  << std::endl;
    std::cout << it << "," << v.error_idx << "," << v.error_val << "," << t.gold.full_ms << "," << t.gold.topk_ms << ","
              << (long long)t.setup_ms << "," << (float)t.kernel_ms << "," << (float)t.exec_ms << "," << (float)t.readback_ms
              << "," << sw.idx.size() << "," << joined(sw.idx) << "," << joined(sw.val) << "," << joined(hw.idx) << ","
              << joined(hw.val) << std::endl;
}
void report_verbose(const Verdict &v, const IterationTimes &t, const TopK &sw, const TopK &hw) {
    print_list("sw results =", sw);
    print_list("hw results=", hw);
    std::cout << "num errors on indices=" << v.error_idx << std::endl;
    std::cout << "num errors on values=" << v.error_val << std::endl;
    std::cout << "precision=" << v.precision << std::endl;
    std::cout << "gpu exec time=" << (float)t.exec_ms << " ms" << std::endl;
}

}  // namespace

int main(int argc, char *argv[]) {
    if (argc == 2 && std::string(argv[1]) == "--options") {  // the library's documented options (include/tkspmv.h: tkspmv_set_option)
        for (int i = 0; i < tkspmv_option_count(); ++i) {
            const char *name, *kind, *values, *doc;
            tkspmv_option_info(i, &name, &kind, &values, &doc);
            std::cout << "TKSPMV_" << name << "  [" << kind << "]  " << values << "\n    " << doc << "\n";
        }
        std::cout << "this program only: TKSPMV_FIXED_WIDTH (8..32: fixed-point values of that width), TKSPMV_INDEX_BASE (0 | 1 | auto), "
                     "TKSPMV_SEED (query vector), TKSPMV_CACHE_DIR (packed-matrix cache), TKSPMV_GENERATE (rows,cols,nnz,dist,seed: matrix generated in memory)\n";
        return 0;
    }
    const RunConfig cfg(argc, argv);
    const int debug = cfg.opt.debug;
    const int k = cfg.opt.top_k_value;

    MatrixSource src(cfg);
    const uint32_t cols = src.coo.cols;
    std::vector<float> vec(cols);
    tkspmv::sample_vector(vec.data(), (int)cols, true, true, false, cfg.seed);
    if (debug) {
        std::cout << "loaded matrix with " << src.rows() << " rows, " << cols << " columns and " << src.coo.nnz()
                  << " non-zero elements (index base " << src.coo.index_base << ")" << std::endl;
        std::cout << "setup time=" << (long long)src.load_ms << " ms" << std::endl;
    }

    TopK sw(k), hw(k);
    GoldTimes gold = software_gold(src.coo, vec.data(), sw);
    if (debug) {
        std::cout << std::endl;
        print_list("sw results =", sw);
        std::cout << "sw time, full matrix=" << gold.full_ms << " ms; sw time, top-k=" << gold.topk_ms << " ms" << std::endl;
    }

    const auto t_setup = Clock::now();
    SpMV spmv(cfg, src, vec.data());
    const double setup_ms = ms_since(t_setup);
    if (debug) std::cout << "gpu setup time=" << (long long)setup_ms << " ms" << std::endl;

    std::vector<float> exec_times, readback_times, precisions;
    for (unsigned it = 0; it < cfg.opt.num_tests; it++) {
        if (debug) std::cout << "\nIteration " << it << ")" << std::endl;
        if (cfg.opt.reset) {  // a fresh L2-normalised random x per test (host_spmv_topk_csr_gpu.cu:399-402)
            tkspmv::sample_vector(vec.data(), (int)cols, true, false, true, cfg.seed ? cfg.seed + (int)it + 1 : 0);
            gold = software_gold(src.coo, vec.data(), sw);
        }
        spmv.reset(vec.data());

        IterationTimes t{gold, setup_ms, 0, 0, 0};
        const auto t_run = Clock::now();
        t.kernel_ms = spmv() / 1e6;
        t.exec_ms = ms_since(t_run);
        const auto t_read = Clock::now();
        spmv.read_result(hw.val, hw.idx);
        t.readback_ms = ms_since(t_read);

        const Verdict v = compare(sw, hw);
        exec_times.push_back((float)t.exec_ms);
        readback_times.push_back((float)t.readback_ms);
        precisions.push_back(v.precision);
        if (debug) report_verbose(v, t, sw, hw);
        else report_csv(it, v, t, sw, hw);
    }
    if (debug) {  // summary statistics skip the first two tests (host_spmv_bscsr.cpp:699)
        const auto old_precision = std::cout.precision(4);
        std::cout << "----------------" << std::endl;
        std::cout << "Mean GPU execution time=" << tkspmv::mean(exec_times, 2) << "±" << tkspmv::st_dev(exec_times, 2) << " ms" << std::endl;
        std::cout << "Mean read-back time=" << tkspmv::mean(readback_times, 2) << "±" << tkspmv::st_dev(readback_times, 2) << " ms" << std::endl;
        std::cout << "Mean precision=" << tkspmv::mean(precisions, 2) << "±" << tkspmv::st_dev(precisions, 2) << std::endl;
        std::cout << "----------------" << std::endl;
        std::cout.precision(old_precision);
    }
    return 0;
}
