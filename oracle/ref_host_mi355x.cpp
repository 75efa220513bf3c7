// ref_host_mi355x.cpp -- the host program a maintainer of the reference would add next to
// src/gpu/host_spmv_topk_csr_gpu.cu to run THIS engine from THEIR tree: the reference's own Options, readMtx, coo_t,
// create_sample_vector, spmv_coo_gold_top_k, sort_tuples and check_array_equality (all #included from /root/reference,
// nothing copied), with `struct SpMV` bound to the C ABI of include/tkspmv.h -- the snippet INTEGRATION.md section 2
// quotes. `make ref` compiles it into oracle/_ref/host_spmv_topk_mi355x (links libtkspmv.so), which proves that the
// binding compiles against the real headers; on a GPU box the binary runs the reference's flow end to end
// (tests/test_gpu_engine.py::test_reference_side_host_program). TEST INFRASTRUCTURE: nothing in the product uses it.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <random>
#include <tuple>
#include <unordered_set>
#include <vector>

typedef unsigned int int_type;  // the reference's GPU hosts: int_type = unsigned int, real_type = float
typedef unsigned int index_type;  // (host_spmv_topk_csr_gpu.cu:28-29)
typedef float num_type;
typedef float real_type;

#include "src/common/utils/utils.hpp"
#include "src/common/utils/options.hpp"
#include "src/common/utils/evaluation_utils.hpp"
#include "src/fpga/src/ip/coo_matrix.hpp"
#include "src/fpga/src/gold_algorithms/gold_algorithms.hpp"

#include "tkspmv.h"

// ---- the engine behind the reference's four verbs ----------------------------------------------
struct SpMV {
    tkspmv_t *engine = nullptr;
    int k;

    SpMV(int_type *x_, int_type *y_, real_type *val_, int_type num_rows_, int_type num_cols_, int_type num_nnz_,
         real_type *vec_, int k_, int debug = 0, int impl = 0, bool half = false) : k(k_) {
        tkspmv_desc d = {};
        d.rows = num_rows_;  d.cols = num_cols_;  d.nnz = num_nnz_;
        d.row = x_;  d.col = y_;  d.val = val_;          // row-sorted COO, read only during create
        d.k = k_;    d.precision = half ? TKSPMV_F16 : TKSPMV_F32;  d.device = -1;  d.min_score = 0.0f;
        d.impl = impl;                                    // -i: the engine variant
        if (tkspmv_create(&engine, &d) != TKSPMV_OK) { std::cerr << tkspmv_last_error() << std::endl; exit(1); }
        reset(vec_, debug);
    }
    float operator()(int debug) {                         // returns kernel time in ns, like the CUDA host
        double ns = 0;
        if (tkspmv_run(engine, &ns) != TKSPMV_OK) { std::cerr << tkspmv_last_error() << std::endl; exit(1); }
        return (float) ns;
    }
    void read_result(std::vector<real_type> &res_, std::vector<int_type> &res_idx_, int debug = 0) {
        int32_t n = 0;                                     // k entries, value-descending (sort_tuples order)
        tkspmv_read(engine, res_idx_.data(), res_.data(), &n);
    }
    long reset(real_type *vec_, int debug) {
        double ns = 0;
        tkspmv_set_query(engine, vec_, &ns);
        return (long) ns;
    }
    ~SpMV() { tkspmv_destroy(engine); }
};

// ---- a main() in the reference's flow, written against the reference's own helpers ---------------------------------------
int main(int argc, char *argv[]) {
    Options options = Options(argc, argv);
    const int debug = options.debug;
    const int top_k = options.top_k_value;
    std::string path = options.use_sample_matrix ? DEFAULT_MTX_FILE : options.matrix_path;

    int_type rows = 0, cols = 0, nnz = 0;
    std::vector<int_type> x, y;
    std::vector<real_type> val;
    readMtx<int_type, real_type>(path.c_str(), &x, &y, &val, &rows, &cols, &nnz, 0, !options.ignore_matrix_values, debug, true, false);
    coo_t<int_type, real_type> coo(x, y, val);
    rows = coo.num_rows;

    std::vector<real_type> vec(cols);
    create_sample_vector(vec.data(), cols, true, false, true, 7);
    std::vector<int_type> sw_idx(top_k), hw_idx(top_k);
    std::vector<real_type> sw_val(top_k), hw_val(top_k);

    SpMV spmv(coo.start.data(), coo.end.data(), coo.val.data(), rows, cols, nnz, vec.data(), top_k, debug, options.gpu_impl,
              options.use_half_precision_gpu);
    int failures = 0;
    for (unsigned i = 0; i < options.num_tests; i++) {
        create_sample_vector(vec.data(), cols, true, false, true, 8 + (int) i);
        spmv_coo_gold_top_k(coo, vec.data(), top_k, sw_idx.data(), sw_val.data());
        sort_tuples((size_t) top_k, sw_idx.data(), sw_val.data());
        spmv.reset(vec.data(), debug);
        const float ns = spmv(debug);
        spmv.read_result(hw_val, hw_idx, debug);
        const int error_idx = check_array_equality(hw_idx.data(), sw_idx.data(), top_k);
        const int error_val = check_array_equality(hw_val.data(), sw_val.data(), top_k, 10e-6);
        std::unordered_set<int_type> s(sw_idx.begin(), sw_idx.end());
        int hits = 0;
        for (int_type r : hw_idx) hits += (int) s.count(r);
        std::cout << i << "," << error_idx << "," << error_val << "," << ns / 1e6 << "," << (float) hits / top_k << std::endl;
        failures += (hits != top_k) || error_val != 0;
    }
    return failures ? 2 : 0;
}
