/* oracle/hls_model.c -- TEST INFRASTRUCTURE (only tests/, bench.py's checks and tools/ may use it; the product never links it).
 *
 * Plain-C restatement of the reference's HLS Top-K SpMV DATAFLOW for the approximations the exact engine does not have
 * (SURVEY.md 2.4 / 8 f4): at most LIMITED_FINISHED_ROWS row segments per packet, one K-list per packet SLOT and partition, the
 * last row of every partition never flushed, and the host-side merge of the partitions' lists. Followed line by line:
 *
 *   packer        src/fpga/src/host_spmv_bscsr.cpp:133-248  (packet_coo, packet_coo_partition: xf bit, cumulative row-end offsets)
 *   aggregation   src/fpga/src/ip/spmv/spmv_bscsr_top_k_multicore.hpp:104-149 (inner_spmv_topk_product_stream)
 *   summary       .../spmv_bscsr_top_k_multicore.hpp:246-326  (spmv_coo_loop_3: row numbering, carry of the last row)
 *   top-k update  .../spmv_bscsr_top_k_multicore.hpp:331-409  (spmv_coo_loop_4: LIMITED_FINISHED_ROWS lists, argmin; LAST_ROW commented out)
 *   merge         src/fpga/src/host_spmv_bscsr.cpp:399-448    (read_result: first_row added, values <= 0 skipped, map insert, sort_tuples)
 *
 * Parameters the reference fixes at compile time are arguments here: B = BSCSR_PACKET_SIZE ((512 - 1) / (W + 10 + 4):
 * 15 at 20 bits, 13 at 25, 11 at 32; types.hpp:57-79), K (types.hpp:49), LIMITED_FINISHED_ROWS (types.hpp:77), W = FIXED_WIDTH.
 * Arithmetic: real_type = ap_ufixed<W,1,AP_TRN_ZERO> restated as in oracle_fixed_scores (products truncated to W - 1 fraction
 * bits and wrapped at 2.0, sums wrapping at 2.0), or fp32 when W = 0 (the reference's USE_FLOAT build).
 *
 * PARITY UNPINNED: the HLS kernel needs Xilinx headers (ap_fixed.h, hls_stream.h) and cannot be built here; the reference
 * holds no fixture of its output. Two places where the reference reads past the end of a partition's tuple vector
 * (host_spmv_bscsr.cpp:196,205-212: the padding entries of the last packet) are restated as "a padding entry belongs to no
 * row and opens an empty segment", which is what those reads yield when the memory behind the vector holds other rows' ids. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HLS_MAX_B 32
#define HLS_MAX_LIMITED 8

static uint64_t hls_to_fixed(float v, uint32_t W) {
    if (!(v > 0.0f)) return 0;
    const double d = (double)v * (double)(1ull << (W - 1));
    const double top = (double)(1ull << W);
    if (d >= top) return (1ull << W) - 1ull;
    return (uint64_t)d;
}

/* One partition (rows [row0, row1) of the matrix, entries [e0, e1) of the row-sorted COO) through the dataflow.
 * prev_last_row: last row id of the previous partition (0 for the first: host_spmv_bscsr.cpp:153-157).
 * out_idx / out_val: [limited][K] -- list s holds what the kernel's res_local[s] holds at the end (row ids LOCAL to the
 * partition as the kernel counts them: finished rows so far, not matrix row ids); scores in units of 1.0.
 * row_slot (optional, [rows of the matrix]): for every row of the partition that the kernel FINISHED, the slot it was offered
 * to (0..limited-1); 0xFF: never offered (dropped segment, or the partition's last row); row_local (optional): the local id the
 * kernel gave it. */
static void hls_partition(const uint32_t *row, const uint32_t *col, const float *val, uint64_t e0, uint64_t e1, uint32_t prev_last_row,
                          const float *vec, uint32_t B, uint32_t K, uint32_t limited, uint32_t W, uint32_t *out_idx, float *out_val,
                          uint8_t *row_slot, uint32_t *row_local) {
    const uint64_t n = e1 - e0;
    const uint64_t n_packets = (n + B - 1) / B;
    const uint64_t mask = W ? ((1ull << W) - 1ull) : 0ull;
    const float unit = W ? ldexpf(1.0f, -(int)(W - 1)) : 1.0f;
    /* top-k state (spmv_bscsr_top_k_multicore.hpp:466-480: zero-initialised lists, worst = element 0) */
    double res[HLS_MAX_LIMITED][64];
    uint32_t res_idx[HLS_MAX_LIMITED][64];
    uint32_t worst_idx[HLS_MAX_LIMITED];
    double worst_val[HLS_MAX_LIMITED];
    for (uint32_t s = 0; s < limited; s++) {
        for (uint32_t j = 0; j < K; j++) {
            res[s][j] = 0.0;
            res_idx[s][j] = 0;
        }
        worst_idx[s] = 0;
        worst_val[s] = 0.0;
    }
    uint32_t curr_row = prev_last_row;         /* packer: row of the last tuple looked at */
    uint32_t last_row_of_packet = 0;           /* loop 3: local row counter */
    double last_row_of_packet_output = 0.0;    /* loop 3: sum of the packet's last (unfinished) row */
    uint32_t last_segment_row = 0xFFFFFFFFu;   /* (model only: matrix row id behind last_row_of_packet_output) */
    for (uint64_t i = 0; i < n_packets; i++) {
        /* ---- packer (host_spmv_bscsr.cpp:189-246) ---- */
        const uint64_t base = e0 + i * B;
        const int xf = row[base] != curr_row;
        uint32_t x_local[HLS_MAX_B];
        memset(x_local, 0, sizeof(x_local));
        for (uint32_t j = 0; j < B; j++)
            if (base + j < e1) curr_row = row[base + j];
        uint32_t pos = 0, same = 1;
        for (uint32_t j = 1; j < B; j++) {
            if (base + j - 1 < e1) {
                if (base + j < e1 && row[base + j] == row[base + j - 1]) {
                    same++;
                } else {
                    x_local[pos++] = same;
                    same = 1;
                }
            } else {
                x_local[pos++] = 0;
            }
        }
        if (base + B - 1 < e1) x_local[pos] = same;
        for (uint32_t j = 1; j < B; j++) x_local[j] += x_local[j - 1];
        /* ---- aggregation (spmv_bscsr_top_k_multicore.hpp:104-149): the first `limited` segments only ---- */
        double agg[HLS_MAX_LIMITED + 1];
        uint32_t seg_row[HLS_MAX_LIMITED + 1]; /* (model only) matrix row of each aggregated segment */
        uint32_t num_rows_in_packet = 0;
        for (uint32_t s = 0; s < limited; s++) {
            const uint32_t start = s > 0 ? x_local[s - 1] : 0u, end = x_local[s];
            num_rows_in_packet += (start != end);
            seg_row[s] = start != end ? row[base + start] : 0xFFFFFFFFu;
            if (W) {
                uint64_t a = 0;
                for (uint32_t j = start; j < end && base + j < e1; j++) {
                    const uint64_t p = ((hls_to_fixed(val[base + j], W) * hls_to_fixed(vec[col[base + j]], W)) >> (W - 1)) & mask;
                    a = (a + p) & mask;
                }
                agg[s] = (double)a;
            } else {
                float a = 0.0f;
                for (uint32_t j = start; j < end && base + j < e1; j++) a += val[base + j] * vec[col[base + j]];
                agg[s] = (double)a;
            }
        }
        /* ---- summary (spmv_coo_loop_3, :246-326) ---- */
        const uint32_t starts_new = (i != 0) ? (uint32_t)xf : 0u;
        const uint32_t finished_rows_num = num_rows_in_packet + starts_new - 1u;
        const uint32_t start_row_of_packet = last_row_of_packet + starts_new;
        last_row_of_packet += finished_rows_num;
        double local[HLS_MAX_LIMITED + 1];
        uint32_t local_row[HLS_MAX_LIMITED + 1];
        int finished[HLS_MAX_LIMITED + 1];
        for (uint32_t j = 0; j <= limited; j++) {
            local[j] = 0.0;
            local_row[j] = 0xFFFFFFFFu;
            finished[j] = 0;
        }
        local[1] = agg[0];
        local_row[1] = seg_row[0];
        for (uint32_t j = 1; j < limited; j++) {
            local[1 + j] = agg[j];
            local_row[1 + j] = seg_row[j];
            finished[j] = x_local[j - 1] != (j > 1 ? x_local[j - 2] : 0u);
        }
        if (num_rows_in_packet <= limited) finished[num_rows_in_packet] = 0;
        if (!starts_new) {
            if (W) local[1] = (double)(((uint64_t)local[1] + (uint64_t)last_row_of_packet_output) & mask);
            else local[1] = (double)((float)local[1] + (float)last_row_of_packet_output);
            local[0] = 0.0;
            finished[0] = 0;
        } else {
            local[0] = last_row_of_packet_output;
            local_row[0] = last_segment_row;
            finished[0] = 1;
        }
        last_row_of_packet_output = local[num_rows_in_packet <= limited ? num_rows_in_packet : limited];
        last_segment_row = local_row[num_rows_in_packet <= limited ? num_rows_in_packet : limited];
        /* ---- top-k update (spmv_coo_loop_4, :331-409): slots 0 .. limited-1 ---- */
        for (uint32_t j = 0; j < limited; j++) {
            const double v = local[j];
            if (finished[j] && local_row[j] != 0xFFFFFFFFu) {
                if (row_slot) row_slot[local_row[j]] = (uint8_t)j;
                if (row_local) row_local[local_row[j]] = start_row_of_packet + j - 1u;
            }
            if (v >= worst_val[j] && finished[j]) {
                res_idx[j][worst_idx[j]] = start_row_of_packet + j - 1u;
                res[j][worst_idx[j]] = v;
            }
            uint32_t m = 0; /* argmin: the first minimum */
            for (uint32_t t = 1; t < K; t++)
                if (res[j][t] < res[j][m]) m = t;
            worst_idx[j] = m;
            worst_val[j] = res[j][m];
        }
    }
    for (uint32_t s = 0; s < limited; s++)
        for (uint32_t j = 0; j < K; j++) {
            out_idx[s * K + j] = res_idx[s][j];
            out_val[s * K + j] = (float)(W ? (double)(uint32_t)res[s][j] * (double)unit : res[s][j]);
        }
}

static int cmp_tuple_desc(const void *a, const void *b) {
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? 1 : (x > y ? -1 : 0);
}

/* The whole design: P partitions of ceil(rows / P) rows (host_spmv_bscsr.cpp:136-141), each through hls_partition, then
 * read_result's merge (:399-448): local ids + the partition's first row, values <= 0 skipped, one entry per row id, sorted by
 * (value desc, id desc) like sort_tuples; at most max_out results are written. Returns the number of merged candidates.
 * row_slot / row_local: as in hls_partition, for the whole matrix (may be NULL). */
int hls_model_topk(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, uint32_t rows, const float *vec,
                   uint32_t P, uint32_t B, uint32_t K, uint32_t limited, uint32_t W, uint32_t *out_idx, float *out_val,
                   uint32_t max_out, uint8_t *row_slot, uint32_t *row_local) {
    if (B < 2 || B > HLS_MAX_B || K < 1 || K > 64 || limited < 1 || limited > HLS_MAX_LIMITED || limited > B || P < 1 || (W != 0 && (W < 8 || W > 32)))
        return -1;
    if (row_slot) memset(row_slot, 0xFF, rows);
    const uint32_t per = (rows + P - 1) / P;
    uint64_t *cand = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)P * limited * K + 8);
    uint32_t *pidx = (uint32_t *)malloc(sizeof(uint32_t) * limited * K);
    float *pval = (float *)malloc(sizeof(float) * limited * K);
    size_t n_cand = 0;
    uint64_t e = 0;
    uint32_t prev_last_row = 0;
    for (uint32_t p = 0; p < P; p++) {
        const uint64_t e0 = e;
        while (e < nnz && row[e] / per == p) e++;
        if (e == e0) continue; /* (the reference would fault on an empty partition: coo_partition[0]) */
        const uint32_t first_row = row[e0];
        hls_partition(row, col, val, e0, e, prev_last_row, vec, B, K, limited, W, pidx, pval, row_slot, row_local);
        prev_last_row = row[e - 1];
        for (uint32_t t = 0; t < limited * K; t++) {
            if (!(pval[t] > 0.0f)) continue; /* "Skip empty results" */
            uint32_t bits;
            memcpy(&bits, &pval[t], 4);
            const uint64_t key = ((uint64_t)(bits | 0x80000000u) << 32) | (uint64_t)(pidx[t] + first_row); /* positive floats order like their bits */
            int dup = 0; /* result_map.insert: one entry per row id (the first one wins) */
            for (size_t c = 0; c < n_cand; c++)
                if ((uint32_t)cand[c] == (uint32_t)key) dup = 1;
            if (!dup) cand[n_cand++] = key;
        }
    }
    qsort(cand, n_cand, sizeof(uint64_t), cmp_tuple_desc);
    for (size_t c = 0; c < n_cand && c < max_out; c++) {
        out_idx[c] = (uint32_t)cand[c];
        const uint32_t bits = (uint32_t)(cand[c] >> 32) & 0x7FFFFFFFu;
        memcpy(&out_val[c], &bits, 4);
    }
    free(cand);
    free(pidx);
    free(pval);
    return (int)n_cand;
}
