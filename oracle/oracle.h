/*
 * oracle.h -- CPU ORACLE for the Top-K SpMV hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. The product
 * (approximate-spmv-topk_amd/) never links, imports or executes anything under oracle/.
 *
 * Parity status: PINNED. Every function that restates reference code is checked in tests/test_oracle_pin.py
 * against oracle/_ref/libref_gold.so (the reference's own headers compiled here by `make ref`) and against the
 * committed golden vectors in tests/golden/ that were generated from it (tests/golden/make_golden.py).
 * Exception: oracle_cpu_topn restates the third-party package sparse_dot_topn (un-pinned "pip install",
 * reference README.md:49-50; call site test_cpu.py:104), which is absent here: "parity unpinned" at that
 * boundary; it is pinned instead against scipy's csr @ x on the same inputs.
 */
#ifndef TKSPMV_ORACLE_H
#define TKSPMV_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* spmv_coo_gold_top_k (src/fpga/src/gold_algorithms/gold_algorithms.hpp:188-246), T = float as the GPU hosts
 * instantiate it (src/gpu/host_spmv_topk_csr_gpu.cu:28-29,278-281). Output unsorted, as the reference leaves it. */
void oracle_gold_topk(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec, int k,
                      uint32_t *res_idx, float *res_val);
/* sort_tuples (src/common/utils/evaluation_utils.hpp:40-62): value descending, ties by index descending. */
void oracle_sort_tuples(uint64_t n, uint32_t *idx, float *val);
/* sw_test's top-k leg (host_spmv_topk_csr_gpu.cu:268-286) = gold_topk followed by sort_tuples. */
void oracle_gold_topk_sorted(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz,
                             const float *vec, int k, uint32_t *res_idx, float *res_val);

/* Row scores. present[r] = 1 iff row r has at least one entry. y arrays have `rows` elements. */
void oracle_scores_f32_seq(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                           uint32_t rows, float *y, uint8_t *present); /* gold's sequential fp32 order */
/* Order of the row-per-lane kernels (approximate-spmv-topk_amd/csrc/wsell.hpp): the gold's sequential fp32 sum for rows of
 * at most `seg` entries; a longer row is cut into ceil(len / seg) nearly equal segments (their length rounded up to a
 * multiple of 4, the last one taking what is left), summed sequentially each, and the sums are added left to right. */
void oracle_scores_f32_segmented(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                                 uint32_t rows, uint32_t seg, float *y, uint8_t *present);
void oracle_scores_f64(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                       uint32_t rows, double *y, uint8_t *present);

/* Q1.7 fixed point (BASELINE configs[4], "FIXED_WIDTH-style"): restatement of ap_ufixed<8,1,AP_TRN_ZERO> arithmetic as
 * the FPGA kernel applies it (src/fpga/src/ip/fpga_types.hpp:16-23; products and sums in `real_type`,
 * spmv_bscsr_top_k_multicore.hpp:121-141): values and x are truncated to 1.7 bits, every product is truncated to 1.7
 * bits and wraps at 2.0, sums wrap at 2.0. Deviation, stated: float -> Q1.7 SATURATES at 255/128 here (the HLS type
 * would wrap); inputs are expected in [0, 2). No reference test pins this arithmetic (needs Xilinx headers):
 * parity unpinned; the GPU path is checked bit for bit against THIS integer model. y[r] = wrapped_sum / 128. */
void oracle_q17_scores(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                       uint32_t rows, float *y, uint8_t *present);
/* The same arithmetic for any FIXED_WIDTH W in [8, 32] (reference default 32, src/common/types.hpp:20; the builds its
 * driver lists are 20/21/25/26/32 bits, test_spmv_topk.py:42-47): real_type = ap_ufixed<W,1,AP_TRN_ZERO>
 * (fpga_types.hpp:20): values and x truncated to W-1 fraction bits (saturating, as above), every product truncated to
 * W-1 fraction bits and wrapped at 2.0 (the assignment to real_type, spmv_bscsr_top_k_multicore.hpp:121-125), sums wrap
 * at 2.0 (:133-141). y[r] = (float)wrapped_sum * 2^-(W-1): the u32 -> fp32 conversion rounds to nearest even (W > 24
 * does not fit the mantissa); the engine ranks on these fp32 scores, where the FPGA ranks on the fixed-point words
 * (they can differ only between scores equal to 24 bits). Parity unpinned like Q1.7. W = 8 reproduces
 * oracle_q17_scores. Returns 0, or -1 for a W out of range. */
int oracle_fixed_scores(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                        uint32_t rows, uint32_t W, float *y, uint8_t *present);
/* TKSPMV_Q1_7_WIDE (this repository's own variant, no reference counterpart): Q1.7 values; x scaled by 2^s, s the
 * largest integer in [0,15] with max(x) * 2^s <= 255/128, then truncated to Q1.7; products truncated to 7 fraction
 * bits WITHOUT wrap; sums exact. y[r] = sum / (128 * 2^s). Returns s. */
int oracle_q17_wide_scores(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                           uint32_t cols, uint32_t rows, float *y, uint8_t *present);

/* Exact top-k by (score desc, row desc) among present rows with score >= min_score; padded with (0, 0.0f). */
void oracle_select_topk(const float *y, const uint8_t *present, uint32_t rows, int k, float min_score,
                        uint32_t first_row, uint32_t *res_idx, float *res_val);

/* Order-matched model of the fused kernel's arithmetic on the wave-BSCSR layout (wbscsr.hpp): same products,
 * same in-lane sums, same clipped Kogge-Stone tree, same carries => bit-identical fp32 row scores.
 * C = entries per lane (4 or 8). Rows never finished by a ROW_END keep present = 0. */
void oracle_packed_scores(const uint8_t *packets, uint64_t packet_bytes, const uint32_t *pkt_row,
                          const uint32_t *part_first, const uint32_t *part_count, uint32_t n_parts, uint32_t C,
                          const float *x, uint32_t rows, float *y, uint8_t *present);

/* TKSPMV_F16 (the CUDA comparator's half mode, host_spmv_topk_csr_gpu.cu:132-136,152-160): values rounded to IEEE
 * binary16 (nearest even), everything else fp32. oracle_packed_scores reads a 2-byte value stream itself. */
uint16_t oracle_float_to_half(float f);
float oracle_half_to_float(uint16_t h);
void oracle_round_values_to_half(const float *in, float *out, uint64_t n);

/* create_sample_vector(vec, size, random=true, sum_to_one, norm_one, seed != 0) (src/common/utils/utils.hpp:234-267)
 * with std::mt19937 and std::uniform_real_distribution<double> restated (libstdc++ generate_canonical). */
void oracle_sample_vector(float *vec, int size, int sum_to_one, int norm_one, uint32_t seed);

/* CPU baseline: restatement of sparse_dot_topn's threaded kernel for an N x 1 right-hand side
 * (awesome_cossim_topn(csr, vec.T, K, 0.0, use_threads=True, n_jobs), test_cpu.py:104): fp64 CSR rows, contiguous
 * row blocks over n_threads, keep scores > lower_bound. Returns 0 on success.
 * csr_* from oracle_coo_to_csr_f64 (duplicates summed, like scipy's csr_matrix((val,(x,y)))). */
int oracle_coo_to_csr_f64(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, uint32_t rows,
                          uint64_t *ptr /* rows+1 */, uint32_t *idx /* nnz */, double *v /* nnz */, uint64_t *nnz_out);
int oracle_cpu_topn(const uint64_t *ptr, const uint32_t *idx, const double *v, uint32_t rows, const double *x,
                    double lower_bound, int n_threads, double *scores /* rows */, uint8_t *kept /* rows */);
/* Global top-k over the kept scores (the step a user of the CPU path does next); (score desc, row desc). */
void oracle_cpu_global_topk(const double *scores, const uint8_t *kept, uint32_t rows, int k, uint32_t *res_idx,
                            double *res_val);
/* fp32 variant of the threaded SpMV (values and x in float, float accumulation), for the fp32 baseline line. */
int oracle_cpu_spmv_f32(const uint64_t *ptr, const uint32_t *idx, const float *v, uint32_t rows, const float *x,
                        int n_threads, float *scores);

/* values rounded to Q1.7 bytes (nearest, saturating) and back: the value stream of TKSPMV_Q1_7_F32 */
void oracle_round_values_to_q17(const float *in, float *out, uint64_t n);

/* bench.py's cpu_baseline leg, timed natively: `warm` untimed + `reps` timed queries (query i = xs + (i % n_x) * cols),
 * per query the threaded SpMV (fp64: oracle_cpu_topn; use_f32: oracle_cpu_spmv_f32) and the global top-k; times in ms. */
int oracle_cpu_bench(const uint64_t *ptr, const uint32_t *idx, const double *v64, const float *v32, uint32_t rows,
                     const double *xs64, const float *xs32, int n_x, uint32_t cols, int k, int n_threads, int warm, int reps,
                     int use_f32, double *spmv_ms, double *total_ms);

#ifdef __cplusplus
}
#endif
/* oracle/hls_model.c: the reference's HLS dataflow restated (at most LIMITED_FINISHED_ROWS row segments per packet of B entries,
 * one K-list per packet slot and partition, last row of a partition never flushed, host merge): spmv_bscsr_top_k_multicore.hpp
 * :104-149,246-326,331-409 and host_spmv_bscsr.cpp:133-248,399-448. W = FIXED_WIDTH (0: fp32). Returns the number of merged
 * candidates (at most max_out written), -1 on bad parameters. row_slot [rows] (optional): slot each finished row was offered to,
 * 0xFF = never offered; row_local [rows] (optional): the partition-local id the kernel gave it. PARITY UNPINNED (see the file). */
int hls_model_topk(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, uint32_t rows, const float *vec,
                   uint32_t P, uint32_t B, uint32_t K, uint32_t limited, uint32_t W, uint32_t *out_idx, float *out_val,
                   uint32_t max_out, uint8_t *row_slot, uint32_t *row_local);

#endif
