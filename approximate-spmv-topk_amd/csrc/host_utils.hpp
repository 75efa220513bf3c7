// host_utils.hpp -- host-side mirror of the reference's common layer (L0 in SURVEY.md):
//   Options            <- src/common/utils/options.hpp:37-133
//   read_mtx           <- src/common/utils/utils.hpp:372-404,474-520 + src/common/utils/mmio.hpp:124-230
//   sample_vector      <- src/common/utils/utils.hpp:234-267 (create_sample_vector)
//   sort_tuples & co.  <- src/common/utils/evaluation_utils.hpp:40-62,273-297; utils.hpp:204-217
//   generate_matrix    <- src/resources/python/create_matrices.py:58-128 (distributions only; own PRNG)
//   gold_topk          <- src/fpga/src/gold_algorithms/gold_algorithms.hpp:188-246 (the self-check every
//                         reference `main` runs per iteration; used by the drop-in executable only)
// Re-implemented from the behaviour described in SURVEY.md 2.1/2.2; nothing here touches the GPU.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace tkspmv {

// ---- CLI ------------------------------------------------------------------------------------------
enum GpuImpl { IMPL_CSR = 0, IMPL_CSR_LIGHTSPMV = 1, IMPL_COO = 2 };

struct Options {
    std::string matrix_path = "../../data/matrices_for_testing/matrices_small/matrix_1000_512_20_gamma.mtx";
    bool use_sample_matrix = false;
    bool reset = true;  // reference quirk: default true and -r also sets true => a fresh x every iteration
    unsigned num_tests = 3;
    int debug = 0;
    bool ignore_matrix_values = false;
    int top_k_value = 20;
    std::string xclbin_path = "../approximate_spmv.xclbin";  // accepted and ignored (no bitstream on a GPU)
    int gpu_impl = 0;
    bool use_half_precision_gpu = false;
    int block_size_1d = 32;
    int block_size_2d = 8;
    int num_blocks = 64;

    Options() = default;
    Options(int argc, char *argv[]);
    static const char *default_matrix();
};

// ---- MatrixMarket ------------------------------------------------------------------------------------
struct CooMatrix {
    uint32_t rows = 0, cols = 0;  // size line
    std::vector<uint32_t> row, col;
    std::vector<float> val;
    uint32_t num_rows_coo = 0;  // max(row)+1
    int index_base = 0;
    bool symmetric = false;
    uint64_t nnz() const { return row.size(); }
};

struct IoError {
    int code = 0;  // 0 ok; otherwise tkspmv_status
    std::string message;
};

// index_base: 0, 1 or -1 (auto). read_values=false => values forced to 1. sort => (row, col) ordering.
IoError read_mtx(const std::string &path, int index_base, bool read_values, bool sort, CooMatrix &out);
IoError write_mtx(const std::string &path, uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row,
                  const uint32_t *col, const float *val, int index_base, int precision);
void sort_coo(CooMatrix &m);  // stable (row, col) sort, customSort semantics

// ---- query vector --------------------------------------------------------------------------------------
void sample_vector(float *vec, int size, bool random, bool sum_to_one, bool norm_one, int seed);

// ---- synthetic matrices --------------------------------------------------------------------------------
enum Distribution { DIST_UNIFORM = 0, DIST_GAMMA = 1 };
void generate_matrix(uint32_t rows, uint32_t cols, uint32_t avg_nnz, int dist, uint64_t seed, CooMatrix &out);
// Rows [row_begin, row_end) of the same matrix, with LOCAL row ids (r - row_begin): what a rank of a row-sharded job
// builds (every row has its own PRNG streams, so a slice equals the corresponding rows of the whole matrix).
void generate_matrix_rows(uint32_t row_begin, uint32_t row_end, uint32_t cols, uint32_t avg_nnz, int dist, uint64_t seed,
                          CooMatrix &out);
// Row lengths only (cheap): what nnz-balanced shard bounds are computed from.
void generate_degrees(uint32_t row_begin, uint32_t row_end, uint32_t avg_nnz, int dist, uint64_t seed, uint32_t *deg);

// ---- evaluation -----------------------------------------------------------------------------------------
void sort_tuples(size_t n, uint32_t *idx, float *val);  // value desc, ties idx desc
int check_array_equality(const float *x, const float *y, int n, float tol);
int check_array_equality(const uint32_t *x, const uint32_t *y, int n);
float mean(const std::vector<float> &x, int skip = 0);
float st_dev(const std::vector<float> &x, int skip = 0);

// Sequential streaming top-k over a row-sorted COO: a row replaces the current worst entry when its
// score is >= the worst; list starts as k x (0, 0.0f). Output is NOT sorted (call sort_tuples).
void gold_topk(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec, int k,
               uint32_t *res_idx, float *res_val);

}  // namespace tkspmv
