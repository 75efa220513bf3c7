/*
 * oracle.c -- CPU ORACLE for the Top-K SpMV hot path (see oracle.h). TEST INFRASTRUCTURE ONLY: the product never
 * links this file. Each function cites the reference code it restates (paths relative to the reference root).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------
 * spmv_coo_gold_top_k -- src/fpga/src/gold_algorithms/gold_algorithms.hpp:188-246
 * Streams the row-sorted COO; when a row is complete and its score is >= the current worst entry of the k-slot
 * list it overwrites that entry, then the worst is re-located by a left-to-right scan with a strict '<'
 * (:219-230). The list starts as k x (0, 0.0) (:203-206). The last row is offered without a re-scan (:241-245).
 * ---------------------------------------------------------------------------------------------------------- */
void oracle_gold_topk(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec, int k,
                      uint32_t *res_idx, float *res_val) {
    for (int i = 0; i < k; i++) {
        res_idx[i] = 0;
        res_val[i] = 0.0f;
    }
    if (nnz == 0 || k <= 0) return;
    uint32_t curr_row = row[0];
    float curr_out = 0.0f;
    uint32_t worst_idx = 0;
    float worst_val = 0.0f;
    for (uint64_t i = 0; i < nnz; i++) {
        /* the reference first scatters vec into an nnz-long array (:191-194); same values, same order */
        float contribution = val[i] * vec[col[i]];
        if (row[i] == curr_row) {
            curr_out += contribution;
        } else {
            if (curr_out >= worst_val) {
                res_idx[worst_idx] = curr_row;
                res_val[worst_idx] = curr_out;
                uint32_t w_i = 0;
                float w_v = res_val[0];
                for (int j = 0; j < k; j++) {
                    if (res_val[j] < w_v) {
                        w_i = (uint32_t)j;
                        w_v = res_val[j];
                    }
                }
                worst_idx = w_i;
                worst_val = w_v;
            }
            curr_row = row[i];
            curr_out = contribution;
        }
    }
    if (curr_out >= worst_val) {
        res_idx[worst_idx] = curr_row;
        res_val[worst_idx] = curr_out;
    }
}

/* sort_tuples -- src/common/utils/evaluation_utils.hpp:40-62 */
typedef struct {
    uint32_t idx;
    float val;
} tuple_t;
static int tuple_cmp(const void *a, const void *b) {
    const tuple_t *l = (const tuple_t *)a, *r = (const tuple_t *)b;
    if (l->val != r->val) return (l->val > r->val) ? -1 : 1;
    if (l->idx != r->idx) return (l->idx > r->idx) ? -1 : 1;
    return 0;
}
void oracle_sort_tuples(uint64_t n, uint32_t *idx, float *val) {
    tuple_t *t = (tuple_t *)malloc((n ? n : 1) * sizeof(tuple_t));
    for (uint64_t i = 0; i < n; i++) {
        t[i].idx = idx[i];
        t[i].val = val[i];
    }
    qsort(t, n, sizeof(tuple_t), tuple_cmp);
    for (uint64_t i = 0; i < n; i++) {
        idx[i] = t[i].idx;
        val[i] = t[i].val;
    }
    free(t);
}

void oracle_gold_topk_sorted(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz,
                             const float *vec, int k, uint32_t *res_idx, float *res_val) {
    oracle_gold_topk(row, col, val, nnz, vec, k, res_idx, res_val);
    oracle_sort_tuples((uint64_t)(k > 0 ? k : 0), res_idx, res_val);
}

/* ------------------------------------------------------------------------------------------------------------
 * Row scores (the quantity the gold accumulates per row, :208-217)
 * ---------------------------------------------------------------------------------------------------------- */
void oracle_scores_f32_seq(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                           uint32_t rows, float *y, uint8_t *present) {
    memset(y, 0, (size_t)rows * sizeof(float));
    if (present) memset(present, 0, rows);
    uint64_t i = 0;
    while (i < nnz) {
        uint32_t r = row[i];
        float acc = val[i] * vec[col[i]];
        i++;
        while (i < nnz && row[i] == r) {
            acc += val[i] * vec[col[i]];
            i++;
        }
        if (r < rows) {
            y[r] = acc;
            if (present) present[r] = 1;
        }
    }
}

void oracle_scores_f64(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                       uint32_t rows, double *y, uint8_t *present) {
    memset(y, 0, (size_t)rows * sizeof(double));
    if (present) memset(present, 0, rows);
    for (uint64_t i = 0; i < nnz; i++) {
        uint32_t r = row[i];
        if (r >= rows) continue;
        y[r] += (double)val[i] * (double)vec[col[i]];
        if (present) present[r] = 1;
    }
}

/* The row-per-lane kernels' order (wsell.hpp): a row of at most `seg` entries is summed sequentially -- the gold's order;
 * a longer row is cut into ceil(len / seg) nearly equal segments (their length rounded up to a multiple of 4, the last one
 * taking what is left), each summed sequentially, and the segment sums are added left to right.
 * seg >= the longest row: oracle_scores_f32_seq. */
static uint32_t segment_length(uint32_t len, uint32_t seg) {
    if (len <= seg) return len;
    const uint32_t nseg = (len + seg - 1) / seg;
    return ((len + nseg - 1) / nseg + 3u) & ~3u;
}
void oracle_scores_f32_segmented(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                                 uint32_t rows, uint32_t seg, float *y, uint8_t *present) {
    memset(y, 0, (size_t)rows * sizeof(float));
    if (present) memset(present, 0, rows);
    uint64_t i = 0;
    while (i < nnz) {
        const uint32_t r = row[i];
        uint64_t end = i;
        while (end < nnz && row[end] == r) ++end;
        const uint32_t slen = segment_length((uint32_t)(end - i), seg);
        float total = 0.0f;
        uint32_t n_seg = 0;
        while (i < end) {
            float s = 0.0f;
            for (uint32_t e = 0; e < slen && i < end; ++e, ++i) {
                const float p = val[i] * vec[col[i]];
                s = s + p;
            }
            total = n_seg == 0 ? s : total + s;
            ++n_seg;
        }
        if (r < rows) {
            y[r] = total;
            if (present) present[r] = 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Q1.7 integer model (see oracle.h). Integer arithmetic is associative, so the summation order is irrelevant.
 * ---------------------------------------------------------------------------------------------------------- */
static inline uint32_t to_q1_7(float v) {
    if (!(v > 0.0f)) return 0;
    float s = v * 128.0f;
    if (s >= 255.0f) return 255;
    return (uint32_t)s; /* truncation toward zero (AP_TRN_ZERO) */
}
void oracle_q17_scores(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                       uint32_t rows, float *y, uint8_t *present) {
    memset(y, 0, (size_t)rows * sizeof(float));
    if (present) memset(present, 0, rows);
    uint64_t i = 0;
    while (i < nnz) {
        uint32_t r = row[i];
        uint32_t acc = 0;
        while (i < nnz && row[i] == r) {
            uint32_t p = ((to_q1_7(val[i]) * to_q1_7(vec[col[i]])) >> 7) & 255u; /* product in real_type */
            acc = (acc + p) & 255u;                                                /* sum in real_type */
            i++;
        }
        if (r < rows) {
            y[r] = (float)acc * (1.0f / 128.0f);
            if (present) present[r] = 1;
        }
    }
}

/* Generic width W (see oracle.h): plain W-bit integers (right-aligned), 64-bit products. */
static inline uint64_t to_fixed_w(float v, uint32_t W) {
    if (!(v > 0.0f)) return 0;
    const double d = (double)v * (double)(1ull << (W - 1)); /* exact */
    const double top = (double)(1ull << W);
    if (d >= top) return (1ull << W) - 1ull; /* saturation (stated deviation, as for Q1.7) */
    return (uint64_t)d;                      /* truncation toward zero (AP_TRN_ZERO) */
}
int oracle_fixed_scores(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                        uint32_t rows, uint32_t W, float *y, uint8_t *present) {
    if (W < 8 || W > 32) return -1;
    memset(y, 0, (size_t)rows * sizeof(float));
    if (present) memset(present, 0, rows);
    const uint64_t mask = (1ull << W) - 1ull;
    const float unit = ldexpf(1.0f, -(int)(W - 1));
    uint64_t i = 0;
    while (i < nnz) {
        uint32_t r = row[i];
        uint64_t acc = 0;
        while (i < nnz && row[i] == r) {
            const uint64_t p = ((to_fixed_w(val[i], W) * to_fixed_w(vec[col[i]], W)) >> (W - 1)) & mask; /* product in real_type */
            acc = (acc + p) & mask;                                                                       /* sum in real_type */
            i++;
        }
        if (r < rows) {
            y[r] = (float)(uint32_t)acc * unit; /* u32 -> fp32 rounds to nearest even; the scaling is exact */
            if (present) present[r] = 1;
        }
    }
    return 0;
}

int oracle_q17_wide_scores(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, const float *vec,
                           uint32_t cols, uint32_t rows, float *y, uint8_t *present) {
    memset(y, 0, (size_t)rows * sizeof(float));
    if (present) memset(present, 0, rows);
    float xmax = 0.0f;
    for (uint32_t i = 0; i < cols; i++) xmax = fmaxf(xmax, vec[i]);
    int sh = 0;
    if (xmax > 0.0f) {
        float ratio = 1.9921875f / xmax;
        uint32_t bits;
        memcpy(&bits, &ratio, 4);
        sh = (int)((bits >> 23) & 255u) - 127;
        sh = sh < 0 ? 0 : (sh > 15 ? 15 : sh);
    }
    const float x_scale = (float)(1u << sh);
    const float inv_unit = 1.0f / (128.0f * x_scale);
    uint64_t i = 0;
    while (i < nnz) {
        uint32_t r = row[i];
        uint32_t acc = 0;
        while (i < nnz && row[i] == r) {
            acc += (to_q1_7(val[i]) * to_q1_7(vec[col[i]] * x_scale)) >> 7;
            i++;
        }
        if (r < rows) {
            y[r] = (float)acc * inv_unit;
            if (present) present[r] = 1;
        }
    }
    return sh;
}

/* ------------------------------------------------------------------------------------------------------------
 * Exact top-k with the sort_tuples total order; min-heap of composite keys.
 * ---------------------------------------------------------------------------------------------------------- */
static inline uint32_t order_key(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float key_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void heap_sift_down(uint64_t *h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) return;
        uint64_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}
static int u64_desc(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) ? -1 : (x < y) ? 1 : 0;
}
void oracle_select_topk(const float *y, const uint8_t *present, uint32_t rows, int k, float min_score,
                        uint32_t first_row, uint32_t *res_idx, float *res_val) {
    uint64_t *h = (uint64_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(uint64_t));
    int n = 0;
    for (uint32_t r = 0; r < rows; r++) {
        if (present && !present[r]) continue;
        if (!(y[r] >= min_score)) continue;
        uint64_t key = ((uint64_t)order_key(y[r]) << 32) | r;
        if (n < k) {
            h[n++] = key;
            if (n == k)
                for (int i = n / 2 - 1; i >= 0; i--) heap_sift_down(h, n, i);
        } else if (key > h[0]) {
            h[0] = key;
            heap_sift_down(h, n, 0);
        }
    }
    qsort(h, (size_t)n, sizeof(uint64_t), u64_desc);
    for (int i = 0; i < k; i++) {
        if (i < n) {
            res_idx[i] = (uint32_t)(h[i] & 0xFFFFFFFFu) + first_row;
            res_val[i] = key_to_float((uint32_t)(h[i] >> 32));
        } else {
            res_idx[i] = 0;
            res_val[i] = 0.0f;
        }
    }
    free(h);
}

/* ------------------------------------------------------------------------------------------------------------
 * Order-matched model of the fused kernel on the wave-BSCSR layout. Mirrors, statement for statement, the
 * arithmetic of stream_kernel in approximate-spmv-topk_amd/csrc/engine.hip (products, in-lane segmented sums,
 * clipped Kogge-Stone scan over the 64 lanes, packet carry). Column word: bit0 ROW_END, bit1 SKIP, bits 15..2 col.
 * ---------------------------------------------------------------------------------------------------------- */
/* IEEE binary16 <-> binary32, round to nearest even, overflow to infinity: what __float2half does in the CUDA
 * comparator's half mode (host_spmv_topk_csr_gpu.cu:132-136,152-160). Pinned against numpy.float16 in the tests. */
uint16_t oracle_float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);
    uint32_t q, rem, half;
    if (x < 0x38800000u) {
        if (x < 0x33000000u) return sign;
        const uint32_t e = x >> 23, M = (x & 0x7FFFFFu) | 0x800000u, shift = 126u - e;
        q = M >> shift;
        rem = M & ((1u << shift) - 1u);
        half = 1u << (shift - 1u);
    } else {
        q = (((x >> 23) - 112u) << 10) | ((x & 0x7FFFFFu) >> 13);
        rem = x & 0x1FFFu;
        half = 0x1000u;
    }
    if (rem > half || (rem == half && (q & 1u))) ++q;
    return (uint16_t)(sign | q);
}
float oracle_half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
    uint32_t x;
    if (e == 31u) {
        x = sign | 0x7F800000u | (m << 13);
    } else if (e != 0u) {
        x = sign | ((e + 112u) << 23) | (m << 13);
    } else if (m == 0u) {
        x = sign;
    } else {
        uint32_t mm = m, ee = 113u;
        while (!(mm & 0x400u)) {
            mm <<= 1;
            --ee;
        }
        x = sign | (ee << 23) | ((mm & 0x3FFu) << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}
/* values rounded to fp16 and back, in place semantics of TKSPMV_F16's value stream */
void oracle_round_values_to_half(const float *in, float *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) out[i] = oracle_half_to_float(oracle_float_to_half(in[i]));
}

/* values rounded to Q1.7 (nearest, ties up, saturating at 255/128: ap_ufixed<8,1,AP_RND,AP_SAT>) and back: the value
 * stream of TKSPMV_Q1_7_F32. (byte / 128) * x and byte * (x / 128) are the same fp32 number wherever nothing underflows,
 * so the fp32 oracles run on these values model that precision. */
void oracle_round_values_to_q17(const float *in, float *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) {
        float v = in[i];
        uint32_t q = 0;
        if (v > 0.0f) {
            float s = v * 128.0f + 0.5f;
            q = s >= 255.0f ? 255u : (uint32_t)s;
        }
        out[i] = (float)q * 0.0078125f;
    }
}

void oracle_packed_scores(const uint8_t *packets, uint64_t packet_bytes, const uint32_t *pkt_row,
                          const uint32_t *part_first, const uint32_t *part_count, uint32_t n_parts, uint32_t C,
                          const float *x, uint32_t rows, float *y, uint8_t *present) {
    memset(y, 0, (size_t)rows * sizeof(float));
    if (present) memset(present, 0, rows);
    const uint32_t PE = 64 * C;
    for (uint32_t q = 0; q < n_parts; q++) {
        float carry = 0.0f;
        for (uint32_t i = 0; i < part_count[q]; i++) {
            const uint32_t pidx = part_first[q] + i;
            const uint8_t *pk = packets + (size_t)pidx * packet_bytes;
            /* value type from the packet size: 4 bytes = fp32, 2 bytes = fp16 (TKSPMV_F16; converted exactly to fp32
             * before the multiply, as the kernel does) */
            /* 5.5 bytes per entry: fp32 values with 12-BIT column words (the engine's F32C12 layout, csrc/wbscsr.hpp: a split
             * plane of one dword and one halfword per lane); same arithmetic as 16-bit words */
            const int c12 = packet_bytes * 2u == (uint64_t)PE * 11u;
            const uint32_t vb = c12 ? 4u : (uint32_t)(packet_bytes / PE) - 2u;
            const float *vals = (const float *)pk;
            const uint16_t *hvals = (const uint16_t *)pk;
            const uint16_t *cws = (const uint16_t *)(pk + (size_t)PE * vb);
            const uint8_t *c12p = pk + (size_t)PE * 4u;
            float s[64][8], rs[64][8], head[64], tail[64], vv[64], nv[64];
            uint32_t e[64][8], skip[64][8];
            int first[64], any_e[64], dist[64];
            for (uint32_t l = 0; l < 64; l++) {
                float p[8] = {0};
                for (uint32_t j = 0; j < C; j++) {
                    uint32_t at = (j >> 2) * 256 + l * 4 + (j & 3);
                    uint16_t w;
                    if (c12) {
                        /* split plane: per PAIR of lanes 12 bytes [A_even][B_even | B_odd << 16][A_odd]; A = col0 << 2 | col1 << 12 |
                         * col2 << 22 | SKIP0 | SKIP1 << 1, B = col3 << 2 | SKIP2 | SKIP3 << 1 | ROW_END0..3 << 12 (little-endian) */
                        const uint32_t t = at & 255u, ln = t >> 2, jj = t & 3u;
                        const uint8_t *pl = c12p + (size_t)(at >> 8) * 384u;
                        const uint8_t *ab = pl + (ln >> 1) * 12u + (ln & 1u) * 8u, *bb = pl + (ln >> 1) * 12u + 4u + (ln & 1u) * 2u;
                        const uint32_t A = (uint32_t)ab[0] | ((uint32_t)ab[1] << 8) | ((uint32_t)ab[2] << 16) | ((uint32_t)ab[3] << 24);
                        const uint32_t Bw = (uint32_t)bb[0] | ((uint32_t)bb[1] << 8);
                        const uint32_t col = jj == 0u ? (A >> 2) & 1023u : (jj == 1u ? (A >> 12) & 1023u : (jj == 2u ? (A >> 22) & 1023u : (Bw >> 2) & 1023u));
                        const uint32_t sk = jj == 0u ? (A & 1u) : (jj == 1u ? ((A >> 1) & 1u) : (jj == 2u ? (Bw & 1u) : ((Bw >> 1) & 1u)));
                        w = (uint16_t)((col << 2) | (sk << 1) | ((Bw >> (12u + jj)) & 1u));
                    } else {
                        w = cws[at];
                    }
                    float xv = x[w >> 2];
                    if (vb == 1u) /* Q1.7 bytes, fp32 arithmetic (TKSPMV_Q1_7_F32): byte * (x * 2^-7), as the kernel does */
                        p[j] = (float)pk[at] * (xv * 0.0078125f);
                    else
                        p[j] = (vb == 2u ? oracle_half_to_float(hvals[at]) : vals[at]) * xv;
                    e[l][j] = w & 1u;
                    skip[l][j] = w & 2u;
                }
                if (l == 0) p[0] = p[0] + carry;
                s[l][0] = p[0];
                for (uint32_t j = 1; j < C; j++) s[l][j] = (e[l][j - 1] ? 0.0f : s[l][j - 1]) + p[j];
                any_e[l] = 0;
                first[l] = (int)C - 1;
                for (int j = (int)C - 1; j >= 0; j--) {
                    any_e[l] |= (int)e[l][j];
                    if (e[l][j]) first[l] = j;
                }
                head[l] = s[l][C - 1];
                for (int j = (int)C - 2; j >= 0; j--)
                    if (e[l][j]) head[l] = s[l][j];
                tail[l] = e[l][C - 1] ? 0.0f : s[l][C - 1];
            }
            /* distance to the nearest lane at or below l holding a row end (lane 0 if none) */
            for (int l = 0; l < 64; l++) {
                int m = 0;
                for (int t = l; t >= 0; t--)
                    if (any_e[t]) {
                        m = t;
                        break;
                    }
                dist[l] = l - m;
                vv[l] = tail[l];
            }
            /* Kogge-Stone inside each row of 16 lanes (DPP row_shr:1,2,4,8; lanes without a source add 0) */
            for (int d = 1; d < 16; d <<= 1) {
                for (int l = 0; l < 64; l++) {
                    float up = ((l & 15) >= d) ? vv[l - d] : 0.0f;
                    nv[l] = (dist[l] >= d) ? (vv[l] + up) : vv[l];
                }
                memcpy(vv, nv, sizeof(vv));
            }
            /* row_bcast:15 (row_mask 0xA): lane 15 -> row 1, lane 47 -> row 3 */
            for (int l = 0; l < 64; l++) {
                int row = l >> 4;
                float up = (row == 1 || row == 3) ? vv[row * 16 - 1] : 0.0f;
                nv[l] = (dist[l] > (l & 15)) ? (vv[l] + up) : vv[l];
            }
            memcpy(vv, nv, sizeof(vv));
            /* row_bcast:31 (row_mask 0xC): lane 31 -> rows 2 and 3 */
            for (int l = 0; l < 64; l++) {
                float up = (l >= 32) ? vv[31] : 0.0f;
                nv[l] = (dist[l] > (l & 31)) ? (vv[l] + up) : vv[l];
            }
            memcpy(vv, nv, sizeof(vv));
            uint32_t r = pkt_row[pidx];
            for (int l = 0; l < 64; l++) {
                float cin = (l == 0) ? 0.0f : vv[l - 1];
                float S = cin + head[l];
                for (uint32_t j = 0; j < C; j++) {
                    rs[l][j] = ((int)j == first[l]) ? S : s[l][j];
                    if (e[l][j]) {
                        if (!skip[l][j] && r < rows) {
                            y[r] = rs[l][j];
                            if (present) present[r] = 1;
                        }
                        r++;
                    }
                }
            }
            carry = vv[63];
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * create_sample_vector -- src/common/utils/utils.hpp:234-267 with std::mt19937 / uniform_real_distribution<double>
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937_t;
static void mt_seed(mt19937_t *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(mt19937_t *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t yv = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7FFFFFFFu);
            uint32_t v = g->mt[(i + 397) % 624] ^ (yv >> 1);
            if (yv & 1u) v ^= 0x9908B0DFu;
            g->mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t yv = g->mt[g->idx++];
    yv ^= yv >> 11;
    yv ^= (yv << 7) & 0x9D2C5680u;
    yv ^= (yv << 15) & 0xEFC60000u;
    yv ^= yv >> 18;
    return yv;
}
/* libstdc++ generate_canonical<double, 53>: two 32-bit draws, low word first, divided by 2^64 */
static double mt_canonical(mt19937_t *g) {
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; k++) {
        sum += (double)mt_next(g) * tmp;
        tmp *= 4294967296.0;
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}
void oracle_sample_vector(float *vec, int size, int sum_to_one, int norm_one, uint32_t seed) {
    mt19937_t g;
    mt_seed(&g, seed);
    for (int i = 0; i < size; i++) vec[i] = (float)mt_canonical(&g);
    if (sum_to_one) {
        float sum = 0;
        for (int i = 0; i < size; i++) sum += vec[i];
        for (int i = 0; i < size; i++) vec[i] = vec[i] / sum;
    } else if (norm_one) {
        double sum = 0;
        for (int i = 0; i < size; i++) sum += vec[i] * vec[i];
        double root = sqrt(sum);
        for (int i = 0; i < size; i++) vec[i] = (float)(vec[i] / root);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * CPU baseline: sparse_dot_topn restated (third party, absent here; un-pinned upstream). For an N x 1 right-hand
 * side its kernel reduces, per row, to sum_j A[i,j] * x[j] in CSR order (fp64), kept when > lower_bound; the
 * threaded variant gives each of n_jobs threads one contiguous block of rows (test_cpu.py:104 uses n_jobs=40).
 * ---------------------------------------------------------------------------------------------------------- */
int oracle_coo_to_csr_f64(const uint32_t *row, const uint32_t *col, const float *val, uint64_t nnz, uint32_t rows,
                          uint64_t *ptr, uint32_t *idx, double *v, uint64_t *nnz_out) {
    /* csr_matrix((val,(x,y))): counting sort by row, then per row sort by column and sum duplicates */
    memset(ptr, 0, ((size_t)rows + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < nnz; i++) {
        if (row[i] >= rows) return 1;
        ptr[row[i] + 1]++;
    }
    for (uint32_t r = 0; r < rows; r++) ptr[r + 1] += ptr[r];
    uint64_t *fill = (uint64_t *)malloc(((size_t)rows + 1) * sizeof(uint64_t));
    uint32_t *tidx = (uint32_t *)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
    double *tv = (double *)malloc((nnz ? nnz : 1) * sizeof(double));
    if (!fill || !tidx || !tv) {
        free(fill);
        free(tidx);
        free(tv);
        return 2;
    }
    memcpy(fill, ptr, ((size_t)rows + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < nnz; i++) {
        uint64_t d = fill[row[i]]++;
        tidx[d] = col[i];
        tv[d] = (double)val[i];
    }
    uint64_t out = 0;
    for (uint32_t r = 0; r < rows; r++) {
        uint64_t b = ptr[r], e = ptr[r + 1];
        /* insertion sort by column (rows are short), stable */
        for (uint64_t i = b + 1; i < e; i++) {
            uint32_t ci = tidx[i];
            double vi = tv[i];
            uint64_t j = i;
            while (j > b && tidx[j - 1] > ci) {
                tidx[j] = tidx[j - 1];
                tv[j] = tv[j - 1];
                j--;
            }
            tidx[j] = ci;
            tv[j] = vi;
        }
        uint64_t start = out;
        for (uint64_t i = b; i < e; i++) {
            if (out > start && idx[out - 1] == tidx[i]) {
                v[out - 1] += tv[i];
            } else {
                idx[out] = tidx[i];
                v[out] = tv[i];
                out++;
            }
        }
        ptr[r] = start;
    }
    ptr[rows] = out;
    if (nnz_out) *nnz_out = out;
    free(fill);
    free(tidx);
    free(tv);
    return 0;
}

typedef struct {
    const uint64_t *ptr;
    const uint32_t *idx;
    const double *v;
    const double *x;
    const float *vf;
    const float *xf;
    uint32_t r0, r1;
    double lower_bound;
    double *scores;
    float *scores_f;
    uint8_t *kept;
} job_t;

static void *topn_job(void *arg) {
    job_t *j = (job_t *)arg;
    for (uint32_t r = j->r0; r < j->r1; r++) {
        double sum = 0.0;
        for (uint64_t p = j->ptr[r]; p < j->ptr[r + 1]; p++) sum += j->v[p] * j->x[j->idx[p]];
        int keep = sum > j->lower_bound;
        j->scores[r] = keep ? sum : 0.0;
        j->kept[r] = (uint8_t)keep;
    }
    return NULL;
}
static void *spmv_f32_job(void *arg) {
    job_t *j = (job_t *)arg;
    for (uint32_t r = j->r0; r < j->r1; r++) {
        float sum = 0.0f;
        for (uint64_t p = j->ptr[r]; p < j->ptr[r + 1]; p++) sum += j->vf[p] * j->xf[j->idx[p]];
        j->scores_f[r] = sum;
    }
    return NULL;
}
static int run_jobs(job_t *proto, uint32_t rows, int n_threads, void *(*fn)(void *)) {
    if (n_threads < 1) n_threads = 1;
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    job_t *jobs = (job_t *)malloc((size_t)n_threads * sizeof(job_t));
    if (!th || !jobs) {
        free(th);
        free(jobs);
        return 2;
    }
    uint32_t per = (rows + (uint32_t)n_threads - 1) / (uint32_t)n_threads;
    int started = 0;
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = *proto;
        uint64_t a = (uint64_t)per * (uint64_t)t, b = a + per;
        jobs[t].r0 = (uint32_t)(a < rows ? a : rows);
        jobs[t].r1 = (uint32_t)(b < rows ? b : rows);
        if (n_threads == 1) {
            fn(&jobs[t]);
        } else if (pthread_create(&th[t], NULL, fn, &jobs[t]) == 0) {
            started++;
        } else {
            fn(&jobs[t]);
        }
    }
    if (n_threads > 1)
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}
int oracle_cpu_topn(const uint64_t *ptr, const uint32_t *idx, const double *v, uint32_t rows, const double *x,
                    double lower_bound, int n_threads, double *scores, uint8_t *kept) {
    job_t proto;
    memset(&proto, 0, sizeof(proto));
    proto.ptr = ptr;
    proto.idx = idx;
    proto.v = v;
    proto.x = x;
    proto.lower_bound = lower_bound;
    proto.scores = scores;
    proto.kept = kept;
    return run_jobs(&proto, rows, n_threads, topn_job);
}
int oracle_cpu_spmv_f32(const uint64_t *ptr, const uint32_t *idx, const float *v, uint32_t rows, const float *x,
                        int n_threads, float *scores) {
    job_t proto;
    memset(&proto, 0, sizeof(proto));
    proto.ptr = ptr;
    proto.idx = idx;
    proto.vf = v;
    proto.xf = x;
    proto.scores_f = scores;
    return run_jobs(&proto, rows, n_threads, spmv_f32_job);
}

typedef struct {
    double v;
    uint32_t r;
} dpair_t;
static int dpair_less(const dpair_t *a, const dpair_t *b) { /* a ranks below b */
    if (a->v != b->v) return a->v < b->v;
    return a->r < b->r;
}
static void dheap_sift(dpair_t *h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && dpair_less(&h[l], &h[m])) m = l;
        if (r < n && dpair_less(&h[r], &h[m])) m = r;
        if (m == i) return;
        dpair_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}
static int dpair_desc(const void *a, const void *b) {
    const dpair_t *x = (const dpair_t *)a, *y = (const dpair_t *)b;
    if (dpair_less(y, x)) return -1;
    if (dpair_less(x, y)) return 1;
    return 0;
}
void oracle_cpu_global_topk(const double *scores, const uint8_t *kept, uint32_t rows, int k, uint32_t *res_idx,
                            double *res_val) {
    dpair_t *h = (dpair_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(dpair_t));
    int n = 0;
    for (uint32_t r = 0; r < rows; r++) {
        if (kept && !kept[r]) continue;
        dpair_t c = {scores[r], r};
        if (n < k) {
            h[n++] = c;
            if (n == k)
                for (int i = n / 2 - 1; i >= 0; i--) dheap_sift(h, n, i);
        } else if (dpair_less(&h[0], &c)) {
            h[0] = c;
            dheap_sift(h, n, 0);
        }
    }
    qsort(h, (size_t)n, sizeof(dpair_t), dpair_desc);
    for (int i = 0; i < k; i++) {
        res_idx[i] = i < n ? h[i].r : 0;
        res_val[i] = i < n ? h[i].v : 0.0;
    }
    free(h);
}

/* The CPU baseline timed natively (bench.py's cpu_baseline leg; SURVEY.md 8(d): fp64 and fp32 variants, SpMV-only and
 * SpMV + global top-k, median of >= 10 runs after warm-ups): `warm` untimed + `reps` timed queries, query i uses
 * xs + (i % n_x) * cols. All buffers are allocated once outside the timed region; like sparse_dot_topn_threaded
 * (test_cpu.py:104, n_jobs threads per call) the threads are created per query. use_f32: values, x and sums in float
 * (v32 / xs32), the global top-k then runs over the float scores widened to double (part of its time). */
#include <time.h>
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}
int oracle_cpu_bench(const uint64_t *ptr, const uint32_t *idx, const double *v64, const float *v32, uint32_t rows,
                     const double *xs64, const float *xs32, int n_x, uint32_t cols, int k, int n_threads, int warm, int reps,
                     int use_f32, double *spmv_ms /* reps */, double *total_ms /* reps */) {
    if (n_x < 1 || reps < 1 || k < 1) return 1;
    double *scores = (double *)malloc((size_t)(rows ? rows : 1) * sizeof(double));
    float *scores_f = (float *)malloc((size_t)(rows ? rows : 1) * sizeof(float));
    uint8_t *kept = (uint8_t *)malloc((size_t)(rows ? rows : 1));
    uint32_t *ri = (uint32_t *)malloc((size_t)k * sizeof(uint32_t));
    double *rv = (double *)malloc((size_t)k * sizeof(double));
    if (!scores || !scores_f || !kept || !ri || !rv) {
        free(scores); free(scores_f); free(kept); free(ri); free(rv);
        return 2;
    }
    int rc = 0;
    for (int i = 0; i < warm + reps && rc == 0; i++) {
        const double t0 = now_ms();
        if (use_f32) {
            rc = oracle_cpu_spmv_f32(ptr, idx, v32, rows, xs32 + (size_t)(i % n_x) * cols, n_threads, scores_f);
        } else {
            rc = oracle_cpu_topn(ptr, idx, v64, rows, xs64 + (size_t)(i % n_x) * cols, 0.0, n_threads, scores, kept);
        }
        const double t1 = now_ms();
        if (use_f32) {
            for (uint32_t r = 0; r < rows; r++) {
                scores[r] = (double)scores_f[r];
                kept[r] = scores_f[r] > 0.0f;
            }
        }
        oracle_cpu_global_topk(scores, kept, rows, k, ri, rv);
        const double t2 = now_ms();
        if (i >= warm) {
            spmv_ms[i - warm] = t1 - t0;
            total_ms[i - warm] = t2 - t0;
        }
    }
    free(scores); free(scores_f); free(kept); free(ri); free(rv);
    return rc;
}

