// ref_shim.cpp -- builds oracle/_ref/libref_gold.so FROM THE REFERENCE'S OWN HEADERS, where they lie under
// /root/reference (make ref; -I$(REFERENCE)). Nothing of the reference is copied into this repository: this file
// only #includes those headers and exposes plain-C entry points around the reference functions so that the
// tests can pin oracle/oracle.c and the host-side mirror against the real thing. TEST INFRASTRUCTURE ONLY.
//
// The translation unit is what the reference's CUDA hosts use with T = float
// (src/gpu/host_spmv_topk_csr_gpu.cu:15-19,28-29).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <iostream>
#include <random>
#include <tuple>
#include <vector>

typedef unsigned int int_type;
typedef unsigned int index_type;
typedef float num_type;

#include "src/common/utils/utils.hpp"
#include "src/common/utils/options.hpp"
#include "src/common/utils/evaluation_utils.hpp"
#include "src/fpga/src/ip/coo_matrix.hpp"
#include "src/fpga/src/gold_algorithms/gold_algorithms.hpp"

extern "C" {

// spmv_coo_gold_top_k + sort_tuples, exactly the sw_test sequence (host_spmv_topk_csr_gpu.cu:268-286).
void ref_gold_topk(const unsigned *row, const unsigned *col, const float *val, unsigned long long nnz, const float *vec,
                   int k, int sort, unsigned *res_idx, float *res_val) {
    std::vector<unsigned> x(row, row + nnz), y(col, col + nnz);
    std::vector<float> v(val, val + nnz);
    coo_t<unsigned, float> coo(x, y, v);
    std::vector<float> vec_copy(vec, vec + (*std::max_element(y.begin(), y.end()) + 1));
    spmv_coo_gold_top_k(coo, vec_copy.data(), k, res_idx, res_val);
    if (sort) sort_tuples((size_t)k, res_idx, res_val);
}

unsigned ref_coo_num_rows(const unsigned *row, unsigned long long nnz) {
    std::vector<unsigned> x(row, row + nnz), y(nnz, 0u);
    std::vector<float> v(nnz, 0.0f);
    coo_t<unsigned, float> coo(x, y, v);
    return coo.num_rows;
}

void ref_sort_tuples(unsigned long long n, unsigned *idx, float *val) { sort_tuples((size_t)n, idx, val); }

void ref_create_sample_vector(float *vec, int size, int random, int sum_to_one, int norm_one, int seed) {
    create_sample_vector(vec, size, random != 0, sum_to_one != 0, norm_one != 0, seed);
}

// readMtx as every reference host calls it: directed=0, debug=false, sort_tuples=false
// (host_spmv_bscsr.cpp:539, host_spmv_topk_csr_gpu.cu:326). Two-call protocol: first with NULL arrays for sizes.
static std::vector<unsigned> g_x, g_y;
static std::vector<float> g_val;
int ref_read_mtx(const char *path, int read_values, int zero_indexed, unsigned *rows, unsigned *cols,
                 unsigned *nnz_header, unsigned long long *n_read) {
    g_x.clear();
    g_y.clear();
    g_val.clear();
    unsigned r = 0, c = 0, n = 0;
    int rc = readMtx(path, &g_x, &g_y, &g_val, &r, &c, &n, 0, read_values != 0, false, zero_indexed != 0, false);
    *rows = r;
    *cols = c;
    *nnz_header = n;
    *n_read = g_x.size();
    return rc;
}
void ref_read_mtx_fetch(unsigned *row, unsigned *col, float *val) {
    std::memcpy(row, g_x.data(), g_x.size() * sizeof(unsigned));
    std::memcpy(col, g_y.data(), g_y.size() * sizeof(unsigned));
    std::memcpy(val, g_val.data(), g_val.size() * sizeof(float));
}

struct ref_options_c {
    char matrix_path[1024];
    int use_sample_matrix, reset, num_tests, debug, ignore_matrix_values, top_k_value;
    char xclbin_path[1024];
    int gpu_impl, use_half_precision_gpu, block_size_1d, block_size_2d, num_blocks;
};
void ref_options_parse(int argc, char **argv, ref_options_c *out) {
    optind = 1;
    Options o(argc, argv);
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->matrix_path, o.matrix_path.c_str(), sizeof(out->matrix_path) - 1);
    std::strncpy(out->xclbin_path, o.xclbin_path.c_str(), sizeof(out->xclbin_path) - 1);
    out->use_sample_matrix = o.use_sample_matrix;
    out->reset = o.reset;
    out->num_tests = (int)o.num_tests;
    out->debug = o.debug;
    out->ignore_matrix_values = o.ignore_matrix_values;
    out->top_k_value = o.top_k_value;
    out->gpu_impl = (int)o.gpu_impl;
    out->use_half_precision_gpu = o.use_half_precision_gpu;
    out->block_size_1d = o.block_size_1d;
    out->block_size_2d = o.block_size_2d;
    out->num_blocks = o.num_blocks;
}

int ref_check_array_equality_f(float *x, float *y, int n, float tol) { return check_array_equality(x, y, n, tol, false); }
float ref_mean(const float *x, int n, int skip) { return mean(std::vector<float>(x, x + n), skip); }
float ref_st_dev(const float *x, int n, int skip) { return st_dev(std::vector<float>(x, x + n), skip); }

// coo2csr + spmv_gold (gold_algorithms.hpp:5-18): full y = A.x in CSR order, fp32.
void ref_spmv_gold_csr(const unsigned *row, const unsigned *col, const float *val, unsigned long long nnz,
                       unsigned rows, unsigned cols, const float *vec, float *y) {
    std::vector<unsigned> x(row, row + nnz), yy(col, col + nnz);
    std::vector<float> v(val, val + nnz);
    std::vector<unsigned> ptr(rows + 1), idx(nnz);
    std::vector<float> cv(nnz);
    coo2csr(ptr.data(), idx.data(), cv.data(), x, yy, v, rows, cols, false);
    std::vector<float> vc(vec, vec + cols);
    spmv_gold(ptr.data(), idx.data(), cv.data(), rows, y, vc.data());
}

}  // extern "C"
