// engine.hip -- fused Top-K SpMV for MI355X (gfx950, wave64). Hand-written HIP; no MFMA (there is no dense
// contraction on this path), HBM-streaming bound.
//
// Replaces the reference's FPGA kernel spmv_bscsr_top_k_main (src/fpga/src/ip/spmv/
// spmv_bscsr_top_k_multicore.cpp:8-186, .hpp:104-504: scatter -> aggregation -> summary -> top-k update) and the
// GPU baseline's cusparseSpMV + thrust::sort_by_key + get_topk (src/gpu/host_spmv_topk_csr_gpu.cu:171-231) with
//
//   stream_kernel : one wave per row partition streams wave-BSCSR packets (wbscsr.hpp); x lives in LDS; per
//                   packet: gather x, multiply, in-lane segmented sums, cross-lane segmented scan, one compare
//                   of the lane's best finished row against the running threshold tau. Rows that pass are
//                   appended to a per-workgroup candidate list in LDS (rare). No N-vector is written.
//   tau           : every workgroup publishes the best score it has seen (one u32 per group, single writer).
//                   The published maxima are scores of distinct rows, so the k-th largest of them is a valid lower
//                   bound of the global k-th best score and rows below it can be dropped. Stale or missing values
//                   only make tau smaller: correctness never depends on inter-workgroup timing, only the
//                   candidate count does. All exchange traffic is issued by one "server" wave per workgroup.
//   select tail   : (last workgroup to finish, or select_kernel) exact top-k of the surviving candidates, ordered (score desc, row desc) = sort_tuples
//                   (src/common/utils/evaluation_utils.hpp:40-62); pads with (0, 0.0f) like the gold's
//                   zero-initialised list (gold_algorithms.hpp:203-206).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "engine.hpp"
#include "wsell.hpp"

namespace tkspmv {

// ------------------------------------------------------------------------------------------------------------
// Device helpers
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key(float f) {  // monotone float -> u32; 0 is "nothing"
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

struct StreamParams {
    const uint8_t *packets;
    const uint32_t *pkt_row;
    const uint32_t *part_first;
    const uint32_t *part_count;
    const float *x;
    uint32_t n_parts, cols, packet_bytes;
    uint32_t n_sets;        // 0 => threshold exchange disabled (fewer publishing groups than k), else 1
    uint32_t k;
    uint32_t n_groups_pub;  // groups [0, n_groups_pub) publish maxima (<= 1024)
    uint32_t gpw;           // groups per workgroup
    float min_score;
    uint32_t fixed_width, fixed_mask;  // TKSPMV_FIXED: bits per value and the mask of the top fixed_width bits
    uint32_t *gmax;  // [MAX_GM*64] order keys of the group maxima (zero beyond n_groups_pub)
    uint32_t *tau_g; // one word: order key of the broadcast threshold (monotone, atomic max)
    uint32_t tau_possible;  // 1: at least k publishing groups own rows, so a threshold can form (else nobody waits for one)
    uint32_t n_reducers;  // workgroups [0, n_reducers) reduce gmax -> tau_g; the others only read tau_g
    unsigned long long *wg_cand;  // [grid][WG_SLOTS] packed {score bits | row << 32}; unused slots: row SLOT_INVALID
    unsigned long long *ovf_cand;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    uint32_t fused;  // 1: the last workgroup to finish runs the selection (no second launch)
    // 1: deferred selection. Workgroup 0 of this launch selects the PREVIOUS query's top-k (its survivors sit in
    // the other exchange-state set, complete and visible since that launch ended) and exits; workgroups 1..grid-1
    // stream the current query and end with the flush. No ticket, no second launch, nothing on the critical path.
    uint32_t deferred;
    float *unit_inv_out;  // 1 / (score units per 1.0) of this query, for a selection that runs in a later launch
    float *scores;  // SCORES variant only
    uint32_t dbg_flags;       // ablation switches (TKSPMV_DBG_FLAGS): 1 no publish, 2 no offers, 4 no tau duty, 8 no flush
    const uint8_t *rep_packets[4];  // experiment (TKSPMV_DBG_REPEAT): stream copies the repeats rotate over
    uint32_t dbg_repeat;            // experiment: passes over the partition within ONE launch (0/1 = normal)
    unsigned long long *trace;   // optional (TKSPMV_TRACE=1): per-wave s_memrealtime stamps, [grid+1][9 waves][8]
    unsigned long long *stamps;  // optional (TKSPMV_STAMPS=1): s_memtime stamps of the selection tail, last workgroup
    unsigned long long *dbg;  // optional counters (TKSPMV_STATS=1): [0] slow-path executions, [1] appended rows
};

constexpr int MISC_CAND_CNT = 0, MISC_TAU = 1, MISC_DONE = 5, MISC_XMAX = 6, MISC_SLOW_CNT = 7,
              MISC_GRPMAX = 8 /* [8] */, MISC_PUBLISHED = 16 /* [8] */, MISC_WORDS = 32;  // <= 8 groups per workgroup
// Private candidate list of a streaming wave (entries in LDS): 256 while x is small, 128 when x itself takes 64 KiB
// (two workgroups must still fit the CU's 160 KiB).
template <int XCOLS>
struct ListGeom {
    static constexpr uint32_t WAVE_CAP = XCOLS <= 1024 ? 256u : 128u;
    static constexpr uint32_t CAND_CAP = 8u * WAVE_CAP;  // per workgroup: up to 8 streaming waves
};
constexpr uint32_t WG_SLOTS = 8;              // fixed result slots every workgroup writes (no count round trip)
constexpr uint32_t SLOT_INVALID = 0xFFFFFFFFu;  // row id of an unused slot

// One lane's share of a packet. VT = value type of the stream: 0 = C fp32 values; 1 = C Q1.7 values packed four to a
// dword; 2 = C fp16 values packed two to a dword.
// QM (kernel template parameter): 0 = fp32, 1 = Q1.7 strict (8-bit wrapping sums, the FPGA's real_type), 2 = Q1.7 values
// with x block-scaled by a power of two per query and exact wide accumulation, 3 = fp16 values, fp32 x, fp32 arithmetic
// (the CUDA comparator's half mode, -a: host_spmv_topk_csr_gpu.cu:132-136,152-160), 4 = fixed point of W bits (the
// FPGA's real_type for any FIXED_WIDTH): values and x as left-aligned Q1.31 words, integer products and sums.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int value_type_of(int QM) { return QM == 3 ? 2 : ((QM == 1 || QM == 2) ? 1 : 0); }  // QM 4: one u32 per value, loaded like fp32

// The packet stream is read once per query: nontemporal loads (a plain read kernel over the same bytes gains 12 %
// from them when the stream comes from HBM, tools/stream_probe.hip).
template <int C, int VT>
struct Pkt {
    float v[VT == 0 ? C : 1];
    uint32_t vq[VT == 1 ? C / 4 : (VT == 2 ? C / 2 : 1)];
    uint32_t cw[C / 2];
};

template <int C, int VT>
__device__ __forceinline__ void load_packet(const uint8_t *__restrict__ pk, uint32_t lane, Pkt<C, VT> &o) {
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        if (VT == 1) {
            o.vq[VT == 1 ? q : 0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(pk + q * 256 + lane * 4));
            const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + C * 64 + q * 512 + lane * 8));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        } else if (VT == 2) {
            const u32x2 hv = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + q * 512 + lane * 8));
            o.vq[VT == 2 ? 2 * q + 0 : 0] = hv.x;
            o.vq[VT == 2 ? 2 * q + 1 : 0] = hv.y;
            const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + C * 128 + q * 512 + lane * 8));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        } else {
            const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(pk + q * 1024 + lane * 16));
            o.v[VT == 0 ? 4 * q + 0 : 0] = f.x;
            o.v[VT == 0 ? 4 * q + 1 : 0] = f.y;
            o.v[VT == 0 ? 4 * q + 2 : 0] = f.z;
            o.v[VT == 0 ? 4 * q + 3 : 0] = f.w;
            const u32x2 c = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pk + C * 256 + q * 512 + lane * 8));
            o.cw[2 * q + 0] = c.x;
            o.cw[2 * q + 1] = c.y;
        }
    }
}

// Q1.7 helpers (restating ap_ufixed<8,1,AP_TRN_ZERO>, fpga_types.hpp:20: 1 integer + 7 fraction bits, truncation).
// Conversion from float saturates at the top of the range (the HLS type would wrap there; inputs are expected in
// [0, 2)). Products are truncated to Q1.7 and wrap to 8 bits; sums wrap to 8 bits (mod 2.0).
__device__ __forceinline__ uint32_t to_q1_7_dev(float v) {
    const float s = fminf(fmaxf(v * 128.0f, 0.0f), 255.0f);  // NaN -> 0
    return (uint32_t)s;                                       // truncation
}
// Generic width (wbscsr.hpp to_fixed): W bits, 1 integer bit, left-aligned in a u32; truncation, saturation at the top.
__device__ __forceinline__ uint32_t to_fixed_dev(float v, uint32_t W) {
    if (!(v > 0.0f)) return 0u;
    const float s = v * (float)(1u << (W - 1u));
    const float top = W == 32u ? 4294967296.0f : (float)(1u << W);
    const uint32_t q = s >= top ? (W == 32u ? 0xFFFFFFFFu : (1u << W) - 1u) : (uint32_t)s;
    return q << (32u - W);
}
__device__ __forceinline__ float q17_wrap(float units) {  // units = exact integer sum held in fp32
    return (float)(((uint32_t)units) & 255u);
}

// ------------------------------------------------------------------------------------------------------------
// Final exact selection over the surviving candidates. Runs either as the tail of the stream kernel (in the last
// workgroup to finish: no second launch) or as its own single-workgroup kernel.
// ------------------------------------------------------------------------------------------------------------
struct SelectParams {
    unsigned long long *wg_cand;  // [n_wg][WG_SLOTS] packed {score bits | row << 32}; row SLOT_INVALID = empty
    uint32_t n_wg;
    const unsigned long long *ovf_cand;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
    uint32_t k, first_row;
    float out_scale;  // 1 for fp32; 1/128 for Q1.7 (scores travel as integer units)
    const float *unit_inv_in;  // if set: out_scale is read from here (written by the stream kernel of that query)
    uint32_t *out_idx;
    float *out_val;
    uint32_t *gmax;
    uint32_t *tau_g;
    uint32_t *done_count;  // ticket counter of the fused tail
    uint32_t n_groups_pub;
    uint32_t use_gmax;  // n_sets != 0 and n_groups_pub >= k
    unsigned long long *scratch;  // [n_wg*WG_SLOTS + ovf_cap] composite keys (general path)
    // Multi-query kernel: candidates carry their POSITION in the wave-sliced ELL stream (slice * 64 + lane) instead of a
    // row id; the selection, off the streaming waves' path, looks the row ids up here. NULL: candidates carry row ids.
    const uint32_t *pos_to_row;
    unsigned long long *stats;    // [0] += candidates, [1] += queries, [2] = max candidates, [3] += general-path runs
};

constexpr int MAX_GM = 16;  // n_groups_pub <= 1024 => at most 16 published maxima per lane
constexpr uint32_t SEL_THREADS = 1024;
constexpr uint32_t SEL_CAP = 2048;
constexpr uint32_t SEL_PER_THREAD = 8;  // slot entries held in registers per thread

struct SelectShared {
    unsigned long long keys[SEL_CAP + 8];
    uint32_t cnt, total, thr, last;
};

__device__ __forceinline__ unsigned long long pack_cand(uint32_t score_bits, uint32_t row) {
    return (unsigned long long)score_bits | ((unsigned long long)row << 32);
}
__device__ __forceinline__ unsigned long long make_ckey(unsigned long long packed) {  // (order key << 32) | row
    return ((unsigned long long)order_key(__uint_as_float((uint32_t)packed)) << 32) | (packed >> 32);
}
// Agent-scope (sc1) accesses: the candidates were written by other workgroups of the same launch in fused mode.
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lower bound of the k-th largest of the 64*GM keys held GM per lane: bisection on the top PROBES bits of the
// order key (the remaining low bits are left zero, so the result never exceeds the true k-th largest). 17 bits =
// sign + exponent + 8 mantissa bits: within 0.4 % of the exact value.
template <int GM, int PROBES>
__device__ __forceinline__ uint32_t kth_largest_prefix(const uint32_t (&gk)[MAX_GM], uint32_t k) {
    uint32_t prefix = 0u;
    for (int bit = 31; bit > 31 - PROBES; --bit) {
        const uint32_t trial = prefix | (1u << bit);
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < GM; ++i) c += (uint32_t)__popcll(__ballot(gk[i] >= trial));
        if (c >= k) prefix = trial;
    }
    return prefix;
}

__device__ __forceinline__ void select_body(const SelectParams &P, const uint32_t tid, const uint32_t nthreads,
                                            SelectShared &S, const uint32_t dbg_flags = 0u,
                                            unsigned long long *stamps = nullptr, const float out_scale_override = 0.0f) {
    const float out_scale =
        out_scale_override != 0.0f ? out_scale_override : (P.unit_inv_in ? __hip_atomic_load(P.unit_inv_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P.out_scale);
    const uint32_t lane = tid & 63u;
    const uint32_t n_slots = P.n_wg * WG_SLOTS;  // host guarantees n_slots <= SEL_PER_THREAD * nthreads

    // One round trip: every thread loads its slots, the overflow count and (wave 0) the group maxima blindly.
    unsigned long long mine[SEL_PER_THREAD];
    bool ok[SEL_PER_THREAD];
#pragma unroll
    for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
        const uint32_t f = tid + u * nthreads;
        mine[u] = ~0ull;
        if (f < n_slots) mine[u] = ld_agent(&P.wg_cand[f]);
    }
    if (P.pos_to_row) {
#pragma unroll
        for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
            const uint32_t pos = (uint32_t)(mine[u] >> 32);
            if (pos != SLOT_INVALID) mine[u] = pack_cand((uint32_t)mine[u], P.pos_to_row[pos]);
        }
    }
    // The reducer servers keep the k-th largest published maximum in tau_g: a valid lower bound of the k-th best
    // score (slightly stale, never too high). It prunes what was appended while the threshold was converging.
    const uint32_t thr = P.use_gmax ? __hip_atomic_load(P.tau_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t novf = __hip_atomic_load(P.ovf_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    novf = novf < P.ovf_cap ? novf : P.ovf_cap;
    if (stamps && tid == 0) stamps[4] = __builtin_amdgcn_s_memtime() + (mine[0] & 1ull) * 0ull;  // after the loads returned
    if (dbg_flags & 256u) {  // timing aid: stop once the loads have landed
        if (mine[0] == 1234567ull && thr == 7654321u) P.out_idx[0] = novf;
        if (tid == 0) for (uint32_t c = 0; c < 9u; ++c) P.done_count[32u * c] = 0u;
        return;
    }
    if (tid == 0) {
        S.cnt = 0;
        S.total = 0;
    }
    __syncthreads();
    uint32_t spos[SEL_PER_THREAD];
    uint32_t wtot = 0;
#pragma unroll
    for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
        ok[u] = ((uint32_t)(mine[u] >> 32) != SLOT_INVALID) && (order_key(__uint_as_float((uint32_t)mine[u])) >= thr);
        const uint64_t bm = __ballot(ok[u]);
        spos[u] = wtot + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
        wtot += (uint32_t)__popcll(bm);
    }
    uint32_t wbase = 0;
    if (lane == 0 && wtot) wbase = atomicAdd(&S.total, wtot);
    wbase = __builtin_amdgcn_readfirstlane(wbase);
    __syncthreads();
    const uint32_t n_from_slots = S.total;
    // Overflow entries are pruned against the same threshold (waves that finish early flush against a threshold
    // that is not final yet, and a late threshold floods the list): count first, then place.
    {
        uint32_t c = 0;
        for (uint32_t i = tid; i < novf; i += nthreads)
            c += (order_key(__uint_as_float((uint32_t)ld_agent(&P.ovf_cand[i]))) >= thr) ? 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
        if (lane == 0 && c) atomicAdd(&S.cnt, c);
    }
    __syncthreads();
    const uint32_t total = n_from_slots + S.cnt;
    const bool small = total <= SEL_CAP;
    __syncthreads();  // everybody has read S.cnt before the general path reuses it
    unsigned long long *dst = small ? S.keys : P.scratch;
#pragma unroll
    for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
        if (ok[u]) dst[wbase + spos[u]] = make_ckey(mine[u]);
    }
    for (uint32_t i0 = 0; i0 < novf; i0 += nthreads) {  // wave-uniform trip count
        const uint32_t i = i0 + tid;
        unsigned long long v = i < novf ? ld_agent(&P.ovf_cand[i]) : 0ull;
        const bool keep = i < novf && order_key(__uint_as_float((uint32_t)v)) >= thr;
        if (keep && P.pos_to_row) v = pack_cand((uint32_t)v, P.pos_to_row[(uint32_t)(v >> 32)]);
        const uint64_t bm = __ballot(keep);
        uint32_t base = 0;
        if (lane == 0 && bm) base = atomicAdd(&S.total, (uint32_t)__popcll(bm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep)
            dst[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] =
                make_ckey(v);
    }
    uint32_t n_sel;

    if (small) {
        if (tid < 8) S.keys[total + tid] = 0ull;  // padding for the unrolled rank loop (0 is below every real key)
        __syncthreads();
        n_sel = total;
    } else {
        // General path (threshold exchange disabled or not converged): the keys went to global scratch; bisection
        // for the k-th largest composite key, then compaction of the keys >= it into LDS.
        __syncthreads();
        unsigned long long prefix = 0ull;
        if (total > P.k) {
            for (int bit = 63; bit >= 0; --bit) {
                const unsigned long long trial = prefix | (1ull << bit);
                uint32_t c = 0;
                for (uint32_t i = tid; i < total; i += nthreads) c += (P.scratch[i] >= trial);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
                if (tid == 0) S.cnt = 0;
                __syncthreads();
                if (lane == 0 && c) atomicAdd(&S.cnt, c);
                __syncthreads();
                if (S.cnt >= P.k) prefix = trial;
                __syncthreads();
            }
        }
        if (tid == 0) S.cnt = 0;
        __syncthreads();
        for (uint32_t i = tid; i < total; i += nthreads) {
            const unsigned long long kx = P.scratch[i];
            if (kx >= prefix) {
                const uint32_t pos = atomicAdd(&S.cnt, 1u);
                if (pos < SEL_CAP) S.keys[pos] = kx;
            }
        }
        __syncthreads();
        n_sel = S.cnt < SEL_CAP ? S.cnt : SEL_CAP;
        if (tid < 8) S.keys[n_sel + tid] = 0ull;
        __syncthreads();
    }
    if (stamps && tid == 0) stamps[5] = __builtin_amdgcn_s_memtime();  // keys in LDS

    // Rank by counting: keys are unique (distinct rows), rank r = number of larger keys. G threads share one key
    // (each counts a slice of the list, partial counts meet through quad/oct shuffles) so the whole workgroup works.
    uint32_t G = 1;
    while (G < 8u && n_sel * (G * 2u) <= nthreads) G *= 2u;
    const uint32_t n_pad = (n_sel + 7u) & ~7u;
    const uint32_t n_blocks = n_pad >> 3;  // blocks of 8 keys
    for (uint32_t base = 0; base < n_sel; base += nthreads / G) {
        const uint32_t i = base + tid / G, part = tid & (G - 1u);
        const bool active = i < n_sel;
        const unsigned long long kx = active ? S.keys[i] : ~0ull;
        uint32_t r = 0;
        for (uint32_t blk = part; blk < n_blocks; blk += G) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) r += (S.keys[blk * 8u + u] > kx);
        }
        for (uint32_t d = 1; d < G; d <<= 1) r += (uint32_t)__shfl_xor((int)r, (int)d);
        if (active && part == 0u && r < P.k) {
            P.out_idx[r] = (uint32_t)(kx & 0xFFFFFFFFull) + P.first_row;
            P.out_val[r] = key_to_float((uint32_t)(kx >> 32)) * out_scale;
        }
    }
    if (stamps && tid == 0) stamps[6] = __builtin_amdgcn_s_memtime();  // ranked
    for (uint32_t r = n_sel + tid; r < P.k; r += nthreads) {
        P.out_idx[r] = 0u;
        P.out_val[r] = 0.0f;
    }
    // Reset the exchange state for the next query (this is the last consumer of the query on the stream); last, so
    // that no barrier above has to wait for these stores. Slots: only the ones that held a survivor need a store
    // (the stream kernel writes a slot only when it has one).
#pragma unroll
    for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
        const uint32_t f = tid + u * nthreads;
        if (f < n_slots && (uint32_t)(mine[u] >> 32) != SLOT_INVALID) P.wg_cand[f] = pack_cand(0u, SLOT_INVALID);
    }
    for (uint32_t i = tid; i < P.n_groups_pub; i += nthreads) P.gmax[i] = 0u;
    if (tid == 0) {
        *P.ovf_count = 0u;
        *P.tau_g = 0u;
        for (uint32_t c = 0; c < 9u; ++c) P.done_count[32u * c] = 0u;
    }
    if (tid == 0 && P.stats) {  // TKSPMV_STATS=1 only: four dependent global read-modify-writes
        P.stats[0] += total;
        P.stats[1] += 1ull;
        if (total > P.stats[2]) P.stats[2] = total;
        if (!small) P.stats[3] += 1ull;
        P.stats[8] += novf;  // overflow-list entries before pruning
    }
}

__global__ void __launch_bounds__(SEL_THREADS) select_kernel(const SelectParams P) {
    __shared__ SelectShared S;
    select_body(P, threadIdx.x, blockDim.x, S);
}

// DPP lane movement (gfx950 keeps the GFX9 controls): lanes without a valid source receive 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_zero(float src) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, ROW_MASK, 0xF, true));
}
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138, DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;

// Wave-wide maximum with DPP (result uniform, returned through an SGPR).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float v) {  // lanes without a valid source keep their own value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_keep<DPP_ROW_SHR1, 0xF>(v));
    v = fmaxf(v, dpp_keep<DPP_ROW_SHR2, 0xF>(v));
    v = fmaxf(v, dpp_keep<DPP_ROW_SHR4, 0xF>(v));
    v = fmaxf(v, dpp_keep<DPP_ROW_SHR8, 0xF>(v));
    v = fmaxf(v, dpp_keep<DPP_ROW_BCAST15, 0xA>(v));
    v = fmaxf(v, dpp_keep<DPP_ROW_BCAST31, 0xC>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    auto mv = [](uint32_t a, uint32_t b) { return b < a ? b : a; };
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR1, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR2, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR4, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR8, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Threshold exchange, reader side. One wave: (1) issue the loads of the published maxima early, (2) much later
// stage them in LDS and reduce: tau = min over sets of (max over the set's groups).
struct TauRegs {
    uint32_t k[MAX_GM];
};
__device__ __forceinline__ void tau_issue(const StreamParams &P, uint32_t lane, TauRegs &t) {
    // gmax is allocated with MAX_GM * 64 entries (zero beyond n_groups_pub), so no bounds predicate is needed;
    // whole 256-B rows beyond the used part are skipped with a uniform branch.
#pragma unroll
    for (int i = 0; i < MAX_GM; ++i) {
        t.k[i] = 0u;
        if (64u * i < P.n_groups_pub)
            t.k[i] = __hip_atomic_load(&P.gmax[lane + 64u * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// tau = (lower bound within 2^-8 relative of) the k-th largest published maximum: the maxima are scores of distinct
// rows, so k of them at or above tau prove that the k-th best score overall is at least tau.
__device__ __forceinline__ float tau_from_maxima(const StreamParams &P, const TauRegs &t, const float min_units) {
    const uint32_t rows_used = (P.n_groups_pub + 63u) >> 6;
    uint32_t key;
    if (rows_used <= 1) key = kth_largest_prefix<1, 17>(t.k, P.k);
    else if (rows_used <= 2) key = kth_largest_prefix<2, 17>(t.k, P.k);
    else if (rows_used <= 4) key = kth_largest_prefix<4, 17>(t.k, P.k);
    else if (rows_used <= 8) key = kth_largest_prefix<8, 17>(t.k, P.k);
    else key = kth_largest_prefix<16, 17>(t.k, P.k);
    float tau = min_units;
    if (key != 0u) {
        const float f = key_to_float(key);
        tau = f > tau ? f : tau;
    }
    return tau;
}

// Writer side: lanes 0..gpw-1 of the calling wave push the workgroup's group maxima (kept in LDS) to gmax.
__device__ __forceinline__ void publish_group_max(const StreamParams &P, uint32_t bid, uint32_t lane, uint32_t *misc) {
    if (lane < P.gpw) {
        const uint32_t g = bid * P.gpw + lane;
        const uint32_t key = misc[MISC_GRPMAX + lane];
        if (g < P.n_groups_pub && key > misc[MISC_PUBLISHED + lane]) {
            misc[MISC_PUBLISHED + lane] = key;
            // single writer per slot (this workgroup): a write-through store, no memory-side read-modify-write
            __hip_atomic_store(&P.gmax[g], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The fused streaming kernel
// ------------------------------------------------------------------------------------------------------------
// ---- single-instruction helpers ---------------------------------------------------------------------------------
// hipcc turns mask arithmetic back into v_cmp + v_cndmask (+ s_nop hazards); these keep it at one VALU op each.
// All are plain VGPR -> VGPR VALU operations (no hazard besides the DPP one noted at `tail` below).
template <int BIT>
__device__ __forceinline__ uint32_t bit_mask(uint32_t w) {  // all ones iff bit BIT of w is set
    uint32_t r;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(r) : "v"(w), "n"(BIT));
    return r;
}
__device__ __forceinline__ float mask_select(uint32_t m, float if_set, float if_clear) {  // bitwise m ? a : b
    float r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(if_set), "v"(if_clear));
    return r;
}
__device__ __forceinline__ float mask_clear(uint32_t m, float a) {  // a where m is clear, +0.0 where set
    float r;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(r) : "v"(m), "v"(a));
    return r;
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Finished-row sums of one packet as seen by one lane. The flags stay where they are, in the packet's column
// words: entry j -> word j/2, bits 16*(j&1) (ROW_END) and 16*(j&1)+1 (SKIP).
template <int C>
struct RowSums {
    float rs[C];
    uint32_t cw[C / 2];
    float best_any;  // max over the lane's row ends, placeholders included: only the hot-path trigger uses it
    __device__ __forceinline__ bool end(int j) const { return (cw[j >> 1] >> (16 * (j & 1))) & 1u; }
    __device__ __forceinline__ bool valid(int j) const { return ((cw[j >> 1] >> (16 * (j & 1))) & 3u) == 1u; }
};

// Products, in-lane segmented sums, cross-lane segmented scan. Updates the packet carry.
// Arithmetic (mirrored statement for statement by oracle_packed_scores in oracle/oracle.c):
//   p_j = v_j * x[col_j];  p_0 += carry on lane 0;  s_0 = p_0,  s_j = (end_{j-1} ? +0 : s_{j-1}) + p_j
//   tail = end_{C-1} ? +0 : s_{C-1};   head = s at the lane's first row end
//   vv = clipped Kogge-Stone scan of tail over the 64 lanes (never across a lane that holds a row end)
//   row sum at the lane's first row end = vv[lane-1] + head, at its later row ends = s_j; carry' = vv[63]
// The reduction proper, from the C products of a lane (p) and its column words (cwv).
// INT: the C "floats" (and the carry) hold u32 fixed-point words; every sum is an integer add (wrapping at 2^32 = 2.0 in
// Q1.31: the reference's real_type sums wrap the same way), lane movement and masking are bitwise either way. At the end
// the row sums are converted to fp32 (round to nearest even, what C's (float)u32 does) so that thresholds, candidate
// lists and the selection see ordinary floats: "score units" of 2^-31.
template <bool INT>
__device__ __forceinline__ float add_rn(float a, float b) {
    if (INT) return __uint_as_float(__float_as_uint(a) + __float_as_uint(b));
    return __fadd_rn(a, b);
}
template <int C, bool INT = false>
__device__ __forceinline__ RowSums<C> reduce_core(float (&p)[C], const uint32_t (&cwv)[C / 2], float &carry) {
    uint32_t m[C];  // all-ones where entry j ends a row
#pragma unroll
    for (int j = 0; j < C; ++j) m[j] = (j & 1) ? bit_mask<16>(cwv[j >> 1]) : bit_mask<0>(cwv[j >> 1]);
    p[0] = __builtin_amdgcn_inverse_ballot_w64(1ull) ? add_rn<INT>(p[0], carry) : p[0];  // lane 0 only

    float s[C];
    uint32_t o[C];  // o_j = m_0 | ... | m_j
    s[0] = p[0];
    o[0] = m[0];
#pragma unroll
    for (int j = 1; j < C; ++j) {
        s[j] = add_rn<INT>(mask_clear(m[j - 1], s[j - 1]), p[j]);
        o[j] = o[j - 1] | m[j];
    }
    float head = s[C - 1];
#pragma unroll
    for (int j = C - 2; j >= 0; --j) head = mask_select(m[j], s[j], head);
    float tail;
    // (the DPP instruction that reads `tail` next needs two wait states after a VALU write; the compiler does
    //  not look inside asm, hence the explicit s_nop)
    asm("v_bfi_b32 %0, %1, 0, %2\n\ts_nop 1" : "=v"(tail) : "v"(m[C - 1]), "v"(s[C - 1]));

    // Lane masks of the clipped scan, computed once on the scalar unit from H = lanes holding a row end:
    //   M_d  : no row end in lanes (l-d, l]                     (steps row_shr:1,2,4,8)
    //   P16  : no row end in [first lane of l's 16-lane row, l]  (step row_bcast:15)
    //   P32  : no row end in [first lane of l's 32-lane half, l] (step row_bcast:31)
    const uint64_t H = __ballot(o[C - 1] != 0u);
    const uint64_t M1 = ~H;
    const uint64_t M2 = M1 & ((M1 << 1) | 0x1ull);
    const uint64_t M4 = M2 & ((M2 << 2) | 0x3ull);
    const uint64_t M8 = M4 & ((M4 << 4) | 0xFull);
    uint64_t P16 = M1 & ((M1 << 1) | 0x0001000100010001ull);
    P16 &= (P16 << 2) | 0x0003000300030003ull;
    P16 &= (P16 << 4) | 0x000F000F000F000Full;
    P16 &= (P16 << 8) | 0x00FF00FF00FF00FFull;
    // upper row of each half also needs the whole lower row clear: bit 15 / 47 of P16
    const uint64_t low_clear = ((P16 >> 15) & 0x0000000100000001ull) * 0xFFFF0000ull;
    const uint64_t P32 = P16 & (low_clear | 0x0000FFFF0000FFFFull);

    float vv = tail;
    {
        float t;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR1, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M1) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR2, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M2) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR4, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M4) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_SHR8, 0xF>(vv));
        vv = __builtin_amdgcn_inverse_ballot_w64(M8) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_BCAST15, 0xA>(vv));  // lane 15 -> row 1, lane 47 -> row 3
        vv = __builtin_amdgcn_inverse_ballot_w64(P16) ? t : vv;
        t = add_rn<INT>(vv, dpp_zero<DPP_ROW_BCAST31, 0xC>(vv));  // lane 31 -> rows 2 and 3
        vv = __builtin_amdgcn_inverse_ballot_w64(P32) ? t : vv;
    }
    const float cin = dpp_zero<DPP_WAVE_SHR1, 0xF>(vv);  // lane l-1's inclusive sum; 0 for lane 0
    const float S = add_rn<INT>(cin, head);
    carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vv), 63));

    RowSums<C> out;
    out.rs[0] = S;  // if entry 0 ends a row it is the lane's first row end
#pragma unroll
    for (int j = 1; j < C; ++j) out.rs[j] = mask_select(o[j - 1], s[j], S);  // an earlier end in the lane => s_j
    if (INT) {
#pragma unroll
        for (int j = 0; j < C; ++j) out.rs[j] = (float)__float_as_uint(out.rs[j]);  // fixed-point word -> score units
    }
#pragma unroll
    for (int j = 0; j < C / 2; ++j) out.cw[j] = cwv[j];
    const float NEG_INF = -__builtin_huge_valf();
    float e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) e[j] = mask_select(m[j], out.rs[j], NEG_INF);
    float best = max3(e[0], e[1], e[2]);
#pragma unroll
    for (int j = 3; j < C; j += 2) best = max3(best, e[j], (j + 1 < C) ? e[j + 1] : NEG_INF);
    out.best_any = best;
    return out;
}

// Products from a packet and the x vector staged in LDS, then the reduction.
template <int C, int QM>
__device__ __forceinline__ RowSums<C> reduce_packet(const Pkt<C, value_type_of(QM)> &cur, float &carry, const float *x_lds,
                                                    const uint32_t fixed_mask = 0u) {
    constexpr int VT = value_type_of(QM);
    float p[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint32_t word = cur.cw[j >> 1];
        const uint32_t off = (j & 1) ? ((word >> 16) & 0xFFFCu) : (word & 0xFFFCu);  // byte offset of x[col]
        if (VT == 1) {
            // x is staged as Q1.7 integers; product truncated to Q1.7 and wrapped to 8 bits, exact in fp32
            const uint32_t xq = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(x_lds) + off);
            const uint32_t vq = (cur.vq[VT == 1 ? (j >> 2) : 0] >> (8 * (j & 3))) & 255u;
            // both factors are below 2^8: the 24-bit multiply is exact (and full rate; v_mul_lo_u32 is quarter rate)
            const uint32_t t = __umul24(vq, xq);
            p[j] = (float)(QM == 2 ? (t >> 7) : ((t >> 7) & 255u));  // wide mode: no wrap
        } else if (VT == 2) {
            const float xv = *reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(x_lds) + off);
            const uint32_t hw = cur.vq[VT == 2 ? (j >> 1) : 0];
            const _Float16 hv = __builtin_bit_cast(_Float16, (uint16_t)((j & 1) ? (hw >> 16) : (hw & 0xFFFFu)));
            p[j] = __fmul_rn((float)hv, xv);  // the conversion is exact
        } else if (QM == 4) {
            // both factors are Q1.31 words: the 64-bit product is Q2.62; bits 31..62 are the product in Q1.31 (its integer
            // part wrapped to one bit, like an assignment to real_type), masked down to the W-1 fraction bits kept
            const uint32_t xq = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(x_lds) + off);
            const uint32_t vq = __float_as_uint(cur.v[VT == 0 ? j : 0]);
            if (fixed_mask & 0xFFu) {
                p[j] = __uint_as_float(__builtin_amdgcn_alignbit(__umulhi(vq, xq), vq * xq, 31) & fixed_mask);
            } else {
                // W <= 24: the low 8 bits of every word are zero and x was staged shifted down by 8, so both factors are
                // 24-bit integers (Q1.23) and the full-rate 24-bit multipliers give the 48-bit product (Q2.46), of which
                // bits 15..46 are the product in Q1.31 (v_mul_lo/hi_u32 run at quarter rate)
                uint32_t hi;
                const uint32_t v24 = vq >> 8;
                asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(v24), "v"(xq));
                p[j] = __uint_as_float(__builtin_amdgcn_alignbit(hi, __umul24(v24, xq), 15) & fixed_mask);
            }
        } else {
            const float xv = *reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(x_lds) + off);
            p[j] = __fmul_rn(cur.v[VT == 0 ? j : 0], xv);
        }
    }
    return reduce_core<C, QM == 4>(p, cur.cw, carry);
}

template <int C, int QM>
__device__ __forceinline__ float row_score(const RowSums<C> &R, int j) {  // strict Q1.7: the 8-bit wrap of the row sum
    return QM == 1 ? q17_wrap(R.rs[j]) : R.rs[j];
}
template <int C, int QM>
__device__ __forceinline__ float lane_best(const RowSums<C> &R) {  // placeholders excluded
    float best = -__builtin_huge_valf();
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const float sc = row_score<C, QM>(R, j);
        best = (R.valid(j) && sc > best) ? sc : best;
    }
    return best;
}

// Number of row ends in lower lanes (=> row id of this lane's first row end is rb + that).
template <int C>
__device__ __forceinline__ uint32_t ends_below(const RowSums<C> &R) {
    uint32_t below = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint64_t b = __ballot(R.end(j));
        below += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    }
    return below;
}

// Filter a wave's private candidate list against the threshold: lane l holds entries l, l+64, ... (EPL per lane);
// keep[] / pos[] tell which survive and where they go in the compacted order. Returns the number kept.
template <uint32_t EPL>
struct ListScan {
    uint2 e[EPL];
    uint32_t pos[EPL];
    bool keep[EPL];
};
template <uint32_t EPL>
__device__ __forceinline__ uint32_t scan_list(const uint2 *wcand, uint32_t n, float tau, uint32_t lane, ListScan<EPL> &L) {
    uint32_t total = 0;
#pragma unroll
    for (uint32_t u = 0; u < EPL; ++u) {
        const uint32_t i = lane + 64u * u;
        L.e[u] = make_uint2(0u, 0u);
        if (i < n) L.e[u] = wcand[i];
        L.keep[u] = i < n && __uint_as_float(L.e[u].x) >= tau;
        const uint64_t b = __ballot(L.keep[u]);
        L.pos[u] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        total += (uint32_t)__popcll(b);
    }
    return total;
}
// Drop what the (risen) threshold has made obsolete. All reads are issued before any write (LDS executes a wave's
// instructions in order), so writing the kept entries to the front cannot clobber an entry still to be read.
template <uint32_t EPL>
__device__ __forceinline__ uint32_t compact_list(uint2 *wcand, uint32_t n, float tau, uint32_t lane) {
    ListScan<EPL> L;
    const uint32_t kept = scan_list<EPL>(wcand, n, tau, lane, L);
#pragma unroll
    for (uint32_t u = 0; u < EPL; ++u)
        if (L.keep[u]) wcand[L.pos[u]] = L.e[u];
    return kept;
}

// Candidate path (rare once tau has converged). Every streaming wave owns a private list of WAVE_CAP entries in LDS
// (its length lives in an SGPR: no atomic, no other wave involved). A full list is first compacted against the
// current threshold; only what still does not fit goes to the shared overflow list in global memory, with ONE
// atomic per wave and packet. One LDS atomic raises the group maximum (the server wave pushes it to global memory).
template <int C, int QM, uint32_t WAVE_CAP>
__device__ __forceinline__ void offer_candidates(const StreamParams &P, const RowSums<C> &R, uint32_t rb, float tau,
                                                 uint32_t lane, uint32_t grp_local, bool publishes, uint2 *wcand,
                                                 uint32_t &wcnt, uint32_t *misc) {
    bool pass[C];
    uint32_t slot[C];
    uint32_t total = 0;
    const uint32_t below = ends_below<C>(R);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        pass[j] = R.valid(j) && row_score<C, QM>(R, j) >= tau;
        const uint64_t pb = __ballot(pass[j]);
        slot[j] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(pb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pb, 0u));
        total += (uint32_t)__popcll(pb);
    }
    const float best = lane_best<C, QM>(R);
    const float wmax = wave_max(best >= tau ? best : -__builtin_huge_valf());
    if (total == 0u) return;  // only placeholders of empty rows tripped the trigger
    if (lane == 0) {
        if (publishes)
            (void)__hip_atomic_fetch_max(&misc[MISC_GRPMAX + grp_local], order_key(wmax), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
        if (P.dbg) {  // TKSPMV_STATS=1: summed into global memory when the wave finishes
            atomicAdd(&misc[MISC_SLOW_CNT], 1u);
            atomicAdd(&misc[MISC_CAND_CNT], total);
        }
    }
    if (wcnt + total > WAVE_CAP) wcnt = compact_list<WAVE_CAP / 64u>(wcand, wcnt, tau, lane);
    const uint32_t base = wcnt;
    const uint32_t first_ovf = base < WAVE_CAP ? WAVE_CAP : base;  // list position of the first overflowing row
    uint32_t gbase = 0u;
    if (base + total > WAVE_CAP) {
        if (lane == 0) gbase = atomicAdd(P.ovf_count, base + total - first_ovf);
        gbase = __builtin_amdgcn_readfirstlane(gbase);
    }
    uint32_t r = rb + below;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        if (pass[j]) {
            const uint32_t pos = base + slot[j];
            if (pos < WAVE_CAP) {
                wcand[pos] = make_uint2(__float_as_uint(row_score<C, QM>(R, j)), r);
            } else {
                const uint32_t gp = gbase + (pos - first_ovf);
                if (gp < P.ovf_cap) st_agent(&P.ovf_cand[gp], pack_cand(__float_as_uint(row_score<C, QM>(R, j)), r));
            }
        }
        r += R.end(j) ? 1u : 0u;
    }
    wcnt = base + total < WAVE_CAP ? base + total : WAVE_CAP;
}

#ifndef TKSPMV_STREAM_PRIO
#define TKSPMV_STREAM_PRIO 2
#endif
#ifndef TKSPMV_REDUCER_SLEEP
#define TKSPMV_REDUCER_SLEEP 8
#endif
constexpr unsigned long long FLUSH_TAU_WAIT = 2000;  // x 10 ns: longest wait of a wave for a first threshold
#ifndef TKSPMV_DEFER_PACKETS
#define TKSPMV_DEFER_PACKETS 3
#endif
constexpr int DEFER = TKSPMV_DEFER_PACKETS;  // packets per wave whose rows are judged at the end (threshold exchange cold start)

// One static LDS object per workgroup. x sits at LDS offset 0, so that (column word & 0xFFFC) IS the ds_read address;
// the selection tail reuses the bytes of x and of the candidate list, which are dead by then. Static objects are
// addressed with ds_* instructions for certain: a pointer carved out of the dynamic region can degrade to flat_*
// accesses, and one flat access in the loop forces s_waitcnt vmcnt(0), draining the packet prefetch every iteration.
template <int XCOLS>
struct StreamLds {
    union {
        struct {
            float x[XCOLS];
            uint2 cand[ListGeom<XCOLS>::CAND_CAP];  // private candidate lists {score bits, row}
        } w;
        SelectShared sel;  // fused selection tail (last workgroup only)
    } u;
    uint32_t misc[MISC_WORDS];
};

#ifndef TKSPMV_NBUF
#define TKSPMV_NBUF 3
#endif
template <int C, bool SCORES, int XCOLS, int QM = 0, int NBUF = TKSPMV_NBUF>
__global__ void __launch_bounds__(576, (C == 8 && !SCORES) ? 6 : 5) stream_kernel(const StreamParams P, const SelectParams SP) {
    constexpr bool Q8 = QM == 1 || QM == 2;  // x staged as Q1.7 integers
    constexpr int VT = value_type_of(QM);
    // Deferred packets live in registers (C row sums + C / 2 flag words each): with 8 entries per lane one packet is held,
    // not three -- the same number of rows as two 4-entry packets, and the kernel stays at 80 registers (two workgroups
    // per CU; with three it needed 96 and a single query took 57 us instead of 36).
    constexpr int DEFER_C = C == 8 ? 1 : DEFER;
    __shared__ StreamLds<XCOLS> L;
    float *x_lds = L.u.w.x;
    uint2 *cand = L.u.w.cand;
    uint32_t *misc = L.misc;
    SelectShared &sel_sh = L.u.sel;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t bid = blockIdx.x, n_wg = gridDim.x;  // streaming workgroup id / count
    // TKSPMV_TRACE=1: 100 MHz wall-clock stamps per wave (kept in SGPRs, written once at the very end)
    unsigned long long *tr = (!SCORES && P.trace) ? P.trace + ((size_t)blockIdx.x * 9u + wave) * 8u : nullptr;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0;
    if (tr) tr0 = __builtin_amdgcn_s_memrealtime();
    if (!SCORES && P.deferred) {
        // The selection of the previous query rides along in workgroup 0 (SP.n_wg = 0: there is none); the others
        // stream. The launch has as many workgroups as fit the GPU at once (two per CU) and the matrix is cut into
        // one partition per streaming wave of grid - 1 workgroups, so nothing waits for a free slot: the selection
        // runs during the launch's start-up, when the memory system is still idle.
        if (bid == 0u) {
            if (SP.n_wg != 0u) select_body(SP, tid, blockDim.x, sel_sh);
            if (tr && lane == 0) {
                tr[0] = tr0;
                tr[5] = __builtin_amdgcn_s_memrealtime();
            }
            return;
        }
        bid -= 1u;
        n_wg -= 1u;
    }
    // The last wave of the workgroup is the exchange SERVER, the others stream. vmcnt retires in order, so a slow
    // remote access (the hot threshold word, the maxima of 512 workgroups) issued by a streaming wave would hold
    // back the visibility of every packet load behind it; the server keeps such traffic out of the stream.
    const uint32_t nwaves = (blockDim.x >> 6) - 1u;  // streaming waves
    const bool is_server = (wave == nwaves);
    // Streaming waves outrank the server waves at instruction issue: a reducer's k-th-largest search otherwise slows
    // the workgroups sharing its CU (they were the launch's stragglers by ~2 us).
    if (!is_server) __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);
    const uint32_t grp_local = is_server ? 0u : wave * P.gpw / nwaves;
    const uint32_t grp_global = bid * P.gpw + grp_local;
    const bool publishes = (P.n_sets != 0u) && (grp_global < P.n_groups_pub);
    const bool reducer = bid < P.n_reducers;

    // The first packets of this wave's partition are requested before anything else, so that staging x and the
    // barrier overlap with the first memory round trip instead of preceding it.
    const uint32_t total_waves = nwaves * n_wg;
    uint32_t q = is_server ? P.n_parts : wave * n_wg + bid;
    Pkt<C, VT> buf[NBUF];
    uint32_t rbs[NBUF];
    uint32_t p0 = 0, np = 0;
    if (q < P.n_parts) {
        p0 = P.part_first[q];
        np = P.part_count[q];
    }
    auto prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NBUF - 1; ++u) {  // NBUF-1 packets in flight
            rbs[u] = 0u;
            if (np > 0) {
                const uint32_t iu = ((uint32_t)u < np) ? (uint32_t)u : (np - 1);
                load_packet<C, VT>(P.packets + (size_t)(p0 + iu) * P.packet_bytes, lane, buf[u]);
                rbs[u] = P.pkt_row[p0 + iu];
            }
        }
        rbs[NBUF - 1] = 0u;
    };
    prologue();

    // Stage the dense query vector in LDS (reference: URAM copies, spmv_bscsr_top_k_multicore.cpp:87-140).
    // Scores travel in "units": 1 for fp32; 1/128 for strict Q1.7; 1/(128 * 2^s) in wide mode, where s is the
    // per-query block scale of x (largest s in [0,15] with max(x) * 2^s <= 255/128; every workgroup derives the
    // same s from the same x).
    if (tid < MISC_WORDS) misc[tid] = 0u;
    float x_scale = 1.0f;    // applied to x before quantisation (2^s)
    float unit_scale = 1.0f; // units per 1.0 of score
    if (QM == 2) {
        __syncthreads();
        float lm = 0.0f;
        for (uint32_t i = tid; i < P.cols; i += blockDim.x) lm = fmaxf(lm, P.x[i]);
        lm = wave_max(lm);
        if (lane == 0) atomicMax(&misc[MISC_XMAX], __float_as_uint(lm));  // non-negative floats order like their bits
        __syncthreads();
        const float xmax = __uint_as_float(misc[MISC_XMAX]);
        int sh = 0;
        if (xmax > 0.0f) {
            const float ratio = 1.9921875f / xmax;
            sh = (int)((__float_as_uint(ratio) >> 23) & 255u) - 127;
            sh = sh < 0 ? 0 : (sh > 15 ? 15 : sh);
        }
        x_scale = (float)(1u << sh);
        unit_scale = 128.0f * x_scale;
    } else if (QM == 1) {
        unit_scale = 128.0f;
    } else if (QM == 4) {
        unit_scale = 2147483648.0f;  // scores are Q1.31 words converted to fp32
    }
    const float inv_unit = 1.0f / unit_scale;             // exact: unit_scale is a power of two
    const float min_units = P.min_score * unit_scale;
    for (uint32_t i = tid; i < (uint32_t)XCOLS; i += blockDim.x) {
        const float xv = (i < P.cols) ? P.x[i] : 0.0f;
        if (Q8)
            reinterpret_cast<uint32_t *>(x_lds)[i] = to_q1_7_dev(xv * x_scale);  // x quantised like the matrix values
        else if (QM == 4)  // W <= 24: as a 24-bit integer (see reduce_packet)
            reinterpret_cast<uint32_t *>(x_lds)[i] = to_fixed_dev(xv, P.fixed_width) >> (P.fixed_width <= 24u ? 8 : 0);
        else
            x_lds[i] = xv;
    }
    if (tid == 0) misc[MISC_TAU] = __float_as_uint(min_units);
    __syncthreads();
    if (tr) tr1 = __builtin_amdgcn_s_memrealtime();

    if (is_server) {
        if (!SCORES && P.n_sets != 0u && !(P.dbg_flags & 4u)) {
            for (;;) {
                if (!(P.dbg_flags & 1u)) publish_group_max(P, bid, lane, misc);
                // Only a few servers read all published maxima (many readers of those 16 lines slow the whole
                // stream down: measured); the others read the one word the reducers keep up to date.
                float t;
                if (reducer) {
                    TauRegs tr_;
                    tau_issue(P, lane, tr_);
                    t = tau_from_maxima(P, tr_, min_units);
                    if (lane == 0 && t > min_units)
                        __hip_atomic_fetch_max(P.tau_g, order_key(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const uint32_t kx = __hip_atomic_load(P.tau_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t = kx ? key_to_float(kx) : min_units;
                }
                if (lane == 0) {
                    const float cur_tau = __uint_as_float(
                        __hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    if (t > cur_tau)
                        __hip_atomic_store(&misc[MISC_TAU], __float_as_uint(t), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // every streaming wave counts itself out; none of them ever waits, so this loop always ends
                const uint32_t done =
                    __hip_atomic_load(&misc[MISC_DONE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane(done) >= nwaves) break;
                if (reducer) __builtin_amdgcn_s_sleep(TKSPMV_REDUCER_SLEEP);
                else __builtin_amdgcn_s_sleep(8);
            }
        }
        // Last publication of this workgroup's maxima, now complete (fire and forget). Outside the fused tail the
        // server has no further part: every streaming wave flushes on its own, nobody waits for this wave.
        if (!SCORES && P.n_sets != 0u) publish_group_max(P, bid, lane, misc);
        if (!SCORES && !P.fused) return;
    }
    constexpr uint32_t WAVE_CAP = ListGeom<XCOLS>::WAVE_CAP;
    uint2 *wcand = cand + (is_server ? 0u : wave) * WAVE_CAP;  // this wave's private candidate list
    uint32_t wcnt = 0u;                                         // its length (wave-uniform)
    for (bool first_part = true; q < P.n_parts; q += total_waves, first_part = false) {
        if (!first_part) {  // more partitions than waves (not the case for engines built by tkspmv_create)
            p0 = P.part_first[q];
            np = P.part_count[q];
            prologue();
        }
        const uint8_t *pk = P.packets + (size_t)p0 * P.packet_bytes;
        float carry = 0.0f;

        RowSums<C> st[DEFER_C];  // deferred packets
        uint32_t st_rb[DEFER_C];
#pragma unroll
        for (int d = 0; d < DEFER_C; ++d) {
            st[d].best_any = -__builtin_huge_valf();
            st_rb[d] = 0u;
#pragma unroll
            for (int j = 0; j < C; ++j) st[d].rs[j] = 0.0f;
#pragma unroll
            for (int j = 0; j < C / 2; ++j) st[d].cw[j] = 0u;
        }

        // Two packets in flight behind the one being reduced. The buffers rotate by NAME (the loop is unrolled by
        // NBUF): copying a freshly loaded buffer into another would wait for the youngest load and drain the
        // prefetch queue every iteration.
        const uint32_t np_one = np;
        if (P.dbg_repeat > 1u) np *= P.dbg_repeat;  // experiment: what a persistent multi-query kernel would stream
        uint32_t ia_cur = NBUF - 1 < np_one ? NBUF - 1 : 0u, ia_rep = 0u;
        for (uint32_t i0 = 0; i0 < np; i0 += NBUF) {
#pragma unroll
            for (int u = 0; u < NBUF; ++u) {
                const uint32_t i = i0 + (uint32_t)u;
                if (i >= np) break;
                const Pkt<C, VT> &cur = buf[u];
                const uint32_t rb_cur = rbs[u];
                Pkt<C, VT> &ahead = buf[(u + NBUF - 1) % NBUF];
                uint32_t &rb_ahead = rbs[(u + NBUF - 1) % NBUF];
            {
                // Unconditional (index clamped to the last packet): a fixed number of younger loads lets the
                // compiler wait with a counted vmcnt instead of vmcnt(0).
                uint32_t ia = (i + (NBUF - 1) < np) ? (i + (NBUF - 1)) : (np - 1);
                const uint8_t *pk_a = pk;
                if (P.dbg_repeat > 1u) {
                    ia = ia_cur;
                    pk_a = P.rep_packets[ia_rep & 3u] + (size_t)p0 * P.packet_bytes;
                    if (i + (NBUF - 1) < np) {
                        ++ia_cur;
                        if (ia_cur == np_one) {
                            ia_cur = 0u;
                            ++ia_rep;
                        }
                    }
                }
                load_packet<C, VT>(pk_a + (size_t)ia * P.packet_bytes, lane, ahead);
                rb_ahead = P.pkt_row[p0 + ia];
            }
            float tau = 0.0f;
            if (!SCORES)
                tau = __uint_as_float(
                    __hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));

            const RowSums<C> R = reduce_packet<C, QM>(cur, carry, x_lds, P.fixed_mask);
            if (tr && i == 0u) tr2 = __builtin_amdgcn_s_memrealtime() + (__float_as_uint(R.best_any) & 0u);

            if (SCORES) {
                uint32_t r = rb_cur + ends_below<C>(R);
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    if (R.end(j)) {
                        if (R.valid(j)) P.scores[r] = row_score<C, QM>(R, j) * inv_unit;
                        ++r;
                    }
                }
            } else {
                if (i < (uint32_t)DEFER_C && P.n_sets != 0u) {
                    // Cold start of the threshold exchange: keep the sums in registers, only feed the maxima.
#pragma unroll
                    for (int d = 0; d < DEFER_C; ++d) {
                        if (i == (uint32_t)d) {
                            st[d] = R;
                            st_rb[d] = rb_cur;
                        }
                    }
                    const float wmax = wave_max(lane_best<C, QM>(R));
                    if (lane == 0 && publishes && wmax >= min_units)
                        (void)__hip_atomic_fetch_max(&misc[MISC_GRPMAX + grp_local], order_key(wmax), __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_WORKGROUP);
                } else if (__any(R.best_any >= tau) && !(P.dbg_flags & 2u)) {
                    offer_candidates<C, QM, ListGeom<XCOLS>::WAVE_CAP>(P, R, rb_cur, tau, lane, grp_local, publishes, wcand, wcnt, misc);
                }
            }
            }
        }
        if (tr) tr3 = __builtin_amdgcn_s_memrealtime();
        if (!SCORES && P.n_sets != 0u) {
            // A short partition (small matrix: a handful of packets per wave) is over before the exchange has produced
            // any threshold (~8 us); judging now would keep -- and dump to global memory -- every row, and the
            // selection would face the whole matrix (measured: 100 us per query at 200k rows). Give the exchange a
            // moment, bounded, and only where a threshold can form at all (>= k groups own rows). With long
            // partitions the threshold exists long before this point and the loop does not spin.
            if (P.tau_possible && first_part) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ==
                           __float_as_uint(min_units) &&
                       __builtin_amdgcn_s_memrealtime() - t0 < FLUSH_TAU_WAIT)
                    __builtin_amdgcn_s_sleep(4);
            }
            // The deferred packets, against the threshold as it stands now.
            const float tau =
                __uint_as_float(__hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
#pragma unroll
            for (int d = 0; d < DEFER_C; ++d) {
                if (np > (uint32_t)d && __any(st[d].best_any >= tau))
                    offer_candidates<C, QM, ListGeom<XCOLS>::WAVE_CAP>(P, st[d], st_rb[d], tau, lane, grp_local, publishes, wcand, wcnt, misc);
            }
        }
    }

    if (SCORES) return;
    if (tr) tr4 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long ts_stream_end = P.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (!is_server && lane == 0) atomicAdd(&misc[MISC_DONE], 1u);
    if (P.dbg_flags & 8u) return;

    // ---- flush: every wave on its own, no workgroup synchronisation. What still clears the (now much tighter)
    // threshold leaves the wave's private list: the first survivor to this wave's fixed slot, further ones to the
    // shared overflow list. Slots without a survivor are NOT written: the selection resets every slot it consumed,
    // so an untouched slot is invalid by construction. Write-through (sc1) stores: in fused mode another workgroup
    // of this launch reads them.
    if (!is_server) {
        const float tau = __uint_as_float(
            __hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        ListScan<WAVE_CAP / 64u> LS;
        const uint32_t surv = scan_list<WAVE_CAP / 64u>(wcand, wcnt, tau, lane, LS);
        if (surv != 0u) {
            uint32_t gbase = 0u;
            if (surv > 1u) {
                if (lane == 0) gbase = atomicAdd(P.ovf_count, surv - 1u);
                gbase = __builtin_amdgcn_readfirstlane(gbase);
            }
            unsigned long long *slot = P.wg_cand + (size_t)bid * WG_SLOTS + wave;
#pragma unroll
            for (uint32_t u = 0; u < WAVE_CAP / 64u; ++u) {
                if (LS.keep[u]) {
                    const unsigned long long v = pack_cand(LS.e[u].x, LS.e[u].y);
                    if (LS.pos[u] == 0u) st_agent(slot, v);
                    else if (gbase + LS.pos[u] - 1u < P.ovf_cap) st_agent(&P.ovf_cand[gbase + LS.pos[u] - 1u], v);
                }
            }
        }
        if (P.dbg && lane == 0 && wave == 0u) {  // TKSPMV_STATS=1 (approximate: waves still running are not counted)
            atomicAdd(&P.dbg[0], (unsigned long long)misc[MISC_SLOW_CNT]);
            atomicAdd(&P.dbg[1], (unsigned long long)misc[MISC_CAND_CNT]);
        }
    }
    if (tr && lane == 0) {
        tr[0] = tr0;
        tr[1] = tr1;
        tr[2] = tr2;
        tr[3] = tr3;
        tr[4] = tr4;
        tr[5] = __builtin_amdgcn_s_memrealtime();
        tr[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(63492);  // XCC_ID | HW_ID
    }
    if (!P.fused) {
        if (bid == 0u && tid == 0u && P.unit_inv_out) *P.unit_inv_out = inv_unit;
        return;
    }

    // ---- fused tail: the last workgroup to get here selects the final top-k -----------------------------------
    // Hand-off (cdna_hip_programming.md Guideline 16): every storing wave drains its write-through stores, the
    // workgroup barrier orders them before ONE agent-scope ticket add; the workgroup whose add came last takes an
    // agent-scope acquire, a barrier, and only then loads what the others stored.
    const unsigned long long ts_flush_issued = P.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long ts_flush_done = P.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (tid == 0) {
        // Two-level ticket: 8 group counters (blockIdx % 8) and a top counter, each on its own 128-B line, so the
        // workgroups that finish together do not serialise on one word. Which workgroups share a group is
        // irrelevant for correctness.
        const uint32_t g = bid & 7u;
        const uint32_t n_in_group = (n_wg - g + 7u) >> 3;
        const uint32_t n_groups = n_wg < 8u ? n_wg : 8u;
        uint32_t last = 0u;
        const uint32_t t1 =
            __hip_atomic_fetch_add(&SP.done_count[32u * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t1 == n_in_group - 1u) {
            const uint32_t t2 =
                __hip_atomic_fetch_add(&SP.done_count[32u * 8u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t2 == n_groups - 1u) ? 1u : 0u;
        }
        if (last && !(P.dbg_flags & 32u)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sel_sh.last = last;
    }
    __syncthreads();
    const unsigned long long ts_ticket = P.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (sel_sh.last && !(P.dbg_flags & 16u)) select_body(SP, tid, blockDim.x, sel_sh, P.dbg_flags, P.stamps, inv_unit);
    if (P.stamps && sel_sh.last && tid == 0) {
        P.stamps[0] = ts_stream_end;
        P.stamps[1] = ts_flush_issued;
        P.stamps[2] = ts_flush_done;
        P.stamps[3] = ts_ticket;
        P.stamps[7] = __builtin_amdgcn_s_memtime();
        P.stamps[8] = __builtin_amdgcn_s_memrealtime();
    }
}


// ------------------------------------------------------------------------------------------------------------
// Batch kernel: up to BATCH_MAX queries in ONE launch. Measured on the single-query kernel: a launch costs ~6.5 us
// beyond its steady-state streaming (launch turnaround, first-touch latency of every launch, end skew), and the
// streaming loop alone runs at ~6.4 TB/s once it is going (TKSPMV_DBG_REPEAT experiment). Here every streaming wave
// walks its partition once per query with ONE continuous packet prefetch pipeline across query boundaries; nobody
// waits for another workgroup:
//   * workgroups 1..grid-1 stream; per query the workgroup's server wave stages x into one of two LDS buffers
//     (x of query q+1 while the streaming waves are still in q), serves the threshold exchange of the newest query
//     through that query's own exchange-state set, and, when its 8 streaming waves have counted themselves out of a
//     query, copies their staged survivors to the query's slots, drains ITS stores and adds the workgroup's ticket
//     (release). Streaming waves never wait for global memory they do not need: their survivors go to LDS.
//   * workgroup 0 is the selector: for q = 0, 1, ... it waits until all tickets of q are in (acquire) and runs
//     select_body on q's state. It waits for the streaming workgroups; none of them ever waits for it or for each
//     other, so there is no cycle even if not all workgroups are resident at once.
// Waits inside a streaming workgroup are on its own LDS flags (x staged / waves done), set by waves of the same
// workgroup that never block on anything but memory.
// ------------------------------------------------------------------------------------------------------------
constexpr int BATCH_MAX = 32;
constexpr uint32_t STG_N = 8;  // survivors a wave can stage per query (64 lanes = 8 waves x 8 when the server copies)
#ifndef TKSPMV_ALL_SERVERS_PRIO
#define TKSPMV_ALL_SERVERS_PRIO 0
#endif
#ifndef TKSPMV_TAU_WAIT
#define TKSPMV_TAU_WAIT 3000
#endif
constexpr unsigned long long BATCH_TAU_WAIT = TKSPMV_TAU_WAIT;  // x 10 ns (s_memrealtime runs at 100 MHz)
constexpr int MISC_XREADY = 2, MISC_MINU = 3;  // batch kernel only: x staged for query (value - 1); min score in units

// Per query only what differs from query to query travels in the kernel arguments (32 bytes); the exchange-state set of
// query q is set 0 plus q strides (the sets are allocated as one block per field), so the argument block stays small (64 queries would fit the 4 KiB limit; 32 are used: longer batches measured no faster).
struct BatchIO {
    const float *x;
    const uint8_t *packets;
    uint32_t *out_idx;
    float *out_val;
};
// Exchange-state sets are allocated as one block per field: set s = set 0 plus s strides.
struct SetAddr {
    uint32_t *gmax0, *tau_g0, *ovf_count0;
    unsigned long long *wg_cand0, *ovf_cand0, *scratch;
    float *unit_inv0;
    uint32_t gmax_stride, word_stride, cand_stride;
    uint64_t ovf_stride;
    __device__ __forceinline__ uint32_t *gmax(uint32_t q) const { return gmax0 + (size_t)q * gmax_stride; }
    __device__ __forceinline__ uint32_t *tau_g(uint32_t q) const { return tau_g0 + (size_t)q * word_stride; }
    __device__ __forceinline__ uint32_t *ovf_count(uint32_t q) const { return ovf_count0 + (size_t)q * word_stride; }
    __device__ __forceinline__ float *unit_inv(uint32_t q) const { return unit_inv0 + (size_t)q * word_stride; }
    __device__ __forceinline__ unsigned long long *wg_cand(uint32_t q) const { return wg_cand0 + (size_t)q * cand_stride; }
    __device__ __forceinline__ unsigned long long *ovf_cand(uint32_t q) const { return ovf_cand0 + (size_t)q * ovf_stride; }
};
struct BatchParams : SetAddr {
    uint32_t n_q;
    uint32_t *tickets;  // [BATCH_MAX] counters, 32 words apart
    BatchIO io[BATCH_MAX];
};

template <int XCOLS, int C = 4>
struct BatchLds {
    union {
        struct {
            float x[2][XCOLS];                           // query vector, double-buffered by query parity
            uint2 cand[ListGeom<XCOLS>::CAND_CAP];       // private candidate lists of the streaming waves
        } w;
        SelectShared sel;  // selector workgroup only
    } u;
    uint32_t misc[2][MISC_WORDS];                        // per query parity
    unsigned long long stg[2][8][STG_N];                 // survivors staged by the streaming waves
    uint32_t stg_cnt[2][8];
    // Deferred packets (threshold exchange cold start) wait here, not in registers: row sums and packed row flags per
    // lane. No register cost, so more packets can be deferred (5 while x is small) and fewer rows are appended before
    // the threshold has arrived.
#ifndef TKSPMV_DEFER_B
#define TKSPMV_DEFER_B 2
#endif
    static constexpr int DEFER_B = C == 8 ? 1 : (XCOLS <= 1024 ? TKSPMV_DEFER_B : 2);
    float4 drs[8][DEFER_B][C / 4][64];
    uint32_t dfl[8][DEFER_B][64];
    uint32_t drb[8][DEFER_B];
};

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int C, int XCOLS, int QM>
__global__ void __launch_bounds__(576, 6) batch_kernel(const StreamParams P0, const SelectParams SP0, const BatchParams B) {
    constexpr bool Q8 = QM == 1 || QM == 2;  // x staged as Q1.7 integers
    constexpr int VT = value_type_of(QM);
    constexpr int NBUF = C == 8 ? 2 : 3;  // packets of 8 entries per lane are twice as large: one ahead is as many bytes
    constexpr uint32_t WAVE_CAP = ListGeom<XCOLS>::WAVE_CAP;
    __shared__ BatchLds<XCOLS, C> L;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nwaves = (blockDim.x >> 6) - 1u;  // streaming waves
    const bool is_server = (wave == nwaves);
    const uint32_t nq = B.n_q;

    if (blockIdx.x == 0u) {
        // ---- selector workgroup ------------------------------------------------------------------------------
        const uint32_t n_stream = gridDim.x - 1u;
        for (uint32_t q = 0; q < nq; ++q) {
            if (tid == 0) {
                uint32_t *t = B.tickets + 32u * q;
                // Polled with a compare-and-swap (which also resets the counter for the next launch): atomics execute
                // at the device-wide coherence point, whereas a load -- even agent-scope -- can keep hitting a stale
                // copy of the line in this XCD's L2 (seen: 33 ms on an otherwise idle L2).
                while (atomicCAS(t, n_stream, 0u) != n_stream) __builtin_amdgcn_s_sleep(32);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            SelectParams S = SP0;
            S.wg_cand = B.wg_cand(q);
            S.ovf_cand = B.ovf_cand(q);
            S.ovf_count = B.ovf_count(q);
            S.gmax = B.gmax(q);
            S.tau_g = B.tau_g(q);
            S.scratch = B.scratch;
            S.unit_inv_in = B.unit_inv(q);
            S.out_idx = B.io[q].out_idx;
            S.out_val = B.io[q].out_val;
            select_body(S, tid, blockDim.x, L.u.sel);
            __syncthreads();
            if (P0.trace && tid == 0 && q < 8u) P0.trace[q] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    const uint32_t bid = blockIdx.x - 1u, n_wg = gridDim.x - 1u;
    // traced queries: the first, the middle and the last of the batch
#define TRSLOT(q) ((q) == 0u ? 0u : ((q) == nq / 2u ? 1u : ((q) + 1u == nq ? 2u : 9u)))
    unsigned long long *trw = P0.trace ? P0.trace + ((size_t)blockIdx.x * 9u + wave) * 8u : nullptr;
    if (trw && lane == 0) trw[0] = __builtin_amdgcn_s_memrealtime();
    if (tid < 2u * MISC_WORDS) (&L.misc[0][0])[tid] = 0u;
    if (tid < 16u) (&L.stg_cnt[0][0])[tid] = 0u;
    __syncthreads();
    const uint32_t grp_local = is_server ? 0u : wave * P0.gpw / nwaves;
    const bool publishes = (bid * P0.gpw + grp_local) < P0.n_groups_pub;
    const bool reducer = bid < P0.n_reducers;
    // Streaming waves that own a partition (wave w streams partition w * n_wg + bid): only they take part in the
    // per-query protocol. Waves without one leave at once -- spinning at stream priority on every query's x flag, six of
    // them per workgroup on a small matrix, they starved the server wave (585 us per query at 50k rows).
    uint32_t n_active = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {  // the very test the waves apply to themselves below
        const uint32_t pw = w * n_wg + bid;
        if (pw < P0.n_parts && P0.part_count[pw] != 0u) ++n_active;
    }

    if (is_server) {
        // ---- server wave: x staging, threshold exchange of the newest query, finalisation of the oldest -------
        // The reducers' search must not starve: in a batch the streaming waves (priority 2) never pause, and a reducer at
        // the default priority got ONE pass per query (traced), i.e. the threshold arrived when the query was over.
#if TKSPMV_ALL_SERVERS_PRIO
        __builtin_amdgcn_s_setprio(3);
#else
        if (reducer) __builtin_amdgcn_s_setprio(3);
#endif
        uint32_t staged = 0u, tail = 0u;
        float inv_unit_q[2] = {1.0f, 1.0f}, min_units_q[2] = {0.0f, 0.0f};
        unsigned long long dbg_first_duty = 0ull;
        uint32_t dbg_iters = 0u;
        for (;;) {
            if (staged < nq && staged - tail < 2u) {
                const uint32_t par = staged & 1u;
                const float *xg = B.io[staged].x;
                float x_scale = 1.0f, unit_scale = 1.0f;
                if (QM == 2) {
                    float lm = 0.0f;
#pragma unroll 1
                    for (uint32_t b0 = 0; b0 < (uint32_t)XCOLS; b0 += 1024u) {  // 16 loads in flight per lane
                        float r[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const uint32_t i = b0 + lane + 64u * (uint32_t)u;
                            r[u] = (i < P0.cols) ? xg[i] : 0.0f;
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) lm = fmaxf(lm, r[u]);
                    }
                    const float xmax = wave_max(lm);
                    int sh = 0;
                    if (xmax > 0.0f) {
                        const float ratio = 1.9921875f / xmax;
                        sh = (int)((__float_as_uint(ratio) >> 23) & 255u) - 127;
                        sh = sh < 0 ? 0 : (sh > 15 ? 15 : sh);
                    }
                    x_scale = (float)(1u << sh);
                    unit_scale = 128.0f * x_scale;
                } else if (QM == 1) {
                    unit_scale = 128.0f;
                } else if (QM == 4) {
                    unit_scale = 2147483648.0f;
                }
                inv_unit_q[par] = 1.0f / unit_scale;
                min_units_q[par] = P0.min_score * unit_scale;
                float *xl = L.u.w.x[par];
#pragma unroll 1
                for (uint32_t b0 = 0; b0 < (uint32_t)XCOLS; b0 += 1024u) {  // 16 loads in flight per lane
                    float r[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const uint32_t i = b0 + lane + 64u * (uint32_t)u;
                        r[u] = (i < P0.cols) ? xg[i] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const uint32_t i = b0 + lane + 64u * (uint32_t)u;
                        if (Q8)
                            reinterpret_cast<uint32_t *>(xl)[i] = to_q1_7_dev(r[u] * x_scale);
                        else if (QM == 4)
                            reinterpret_cast<uint32_t *>(xl)[i] = to_fixed_dev(r[u], P0.fixed_width) >> (P0.fixed_width <= 24u ? 8 : 0);
                        else
                            xl[i] = r[u];
                    }
                }
                uint32_t *mp = L.misc[par];
                if (lane < (uint32_t)MISC_WORDS && lane != (uint32_t)MISC_XREADY) mp[lane] = 0u;
                if (lane < 8u) L.stg_cnt[par][lane] = 0u;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (lane == 0) {
                    mp[MISC_TAU] = __float_as_uint(min_units_q[par]);
                    mp[MISC_MINU] = __float_as_uint(min_units_q[par]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(&mp[MISC_XREADY], staged + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (trw && lane == 0 && TRSLOT(staged) < 3u) trw[1 + TRSLOT(staged)] = __builtin_amdgcn_s_memrealtime();
                ++staged;
            }
            // Threshold exchange of the query this workgroup's waves are streaming: the oldest unfinished one until
            // half of the waves have left it, then the next (whose waves need a threshold most).
            if (P0.n_sets != 0u && !(P0.dbg_flags & 4u)) {
                uint32_t hq = tail;
                if (tail + 1u < staged &&
                    2u * __builtin_amdgcn_readfirstlane(lds_load(&L.misc[tail & 1u][MISC_DONE])) >= n_active)
                    hq = tail + 1u;
                {
                    const uint32_t sq = hq;
                    StreamParams P = P0;
                    P.gmax = B.gmax(sq);
                    P.tau_g = B.tau_g(sq);
                    uint32_t *mp = L.misc[sq & 1u];
                    const float min_units = min_units_q[sq & 1u];
                    publish_group_max(P, bid, lane, mp);
                    float t;
                    if (reducer) {
                        TauRegs tr_;
                        tau_issue(P, lane, tr_);
                        t = tau_from_maxima(P, tr_, min_units);
                        if (lane == 0 && t > min_units)
                            __hip_atomic_fetch_max(P.tau_g, order_key(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        const uint32_t kx = __hip_atomic_load(P.tau_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        t = kx ? key_to_float(kx) : min_units;
                    }
                    if (lane == 0) {
                        const float cur_tau = __uint_as_float(lds_load(&mp[MISC_TAU]));
                        if (t > cur_tau)
                            __hip_atomic_store(&mp[MISC_TAU], __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (trw && TRSLOT(sq) == 1u) {
                            if (t > cur_tau && cur_tau <= min_units) trw[7] = __builtin_amdgcn_s_memrealtime();  // first threshold
                            if (dbg_first_duty == 0ull) dbg_first_duty = __builtin_amdgcn_s_memrealtime();
                            ++dbg_iters;
                        }
                    }
                }
            }
            // finalise the oldest query once its streaming waves have all counted themselves out
            {
                const uint32_t tp = tail & 1u;
                uint32_t *mp = L.misc[tp];
                if (tail < staged && __builtin_amdgcn_readfirstlane(lds_load(&mp[MISC_DONE])) >= n_active) {
                    asm volatile("" ::: "memory");
                    StreamParams P = P0;
                    P.gmax = B.gmax(tail);
                    if (P0.n_sets != 0u) publish_group_max(P, bid, lane, mp);  // complete maxima (fire and forget)
                    if (P0.dbg && lane == 0) {  // TKSPMV_STATS=1
                        atomicAdd(&P0.dbg[0], (unsigned long long)mp[MISC_SLOW_CNT]);
                        atomicAdd(&P0.dbg[1], (unsigned long long)mp[MISC_CAND_CNT]);
                    }
                    // lane l copies entry (l % 8) of wave (l / 8): the first goes to the wave's slot, others to the
                    // query's overflow list
                    const uint32_t w = lane >> 3, e = lane & 7u;
                    const uint32_t cnt = L.stg_cnt[tp][w];
                    const bool have = e < cnt;
                    const unsigned long long v = have ? L.stg[tp][w][e] : 0ull;
                    const bool extra = have && e > 0u;
                    const uint64_t bm = __ballot(extra);
                    uint32_t gbase = 0u;
                    if (bm) {
                        if (lane == 0) gbase = atomicAdd(B.ovf_count(tail), (uint32_t)__popcll(bm));
                        gbase = __builtin_amdgcn_readfirstlane(gbase);
                    }
                    if (have && e == 0u) st_agent(B.wg_cand(tail) + (size_t)bid * WG_SLOTS + w, v);
                    if (extra) {
                        const uint32_t gp = gbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        if (gp < P0.ovf_cap) st_agent(&B.ovf_cand(tail)[gp], v);
                    }
                    if (bid == 0u && lane == 0)
                        __hip_atomic_store(B.unit_inv(tail), inv_unit_q[tp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // Hand-off as in the fused tail (cdna_hip_programming.md Guideline 16): everything above is a
                    // write-through (sc1) store; drain them, then a RELAXED agent-scope add. A release-ordered atomic
                    // would write back the whole L2 (buffer_wbl2) once per workgroup and query: measured 4 ms/query.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        (void)__hip_atomic_fetch_add(B.tickets + 32u * tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (trw && lane == 0 && TRSLOT(tail) < 3u) trw[4 + TRSLOT(tail)] = __builtin_amdgcn_s_memrealtime();
                    if (trw && lane == 0 && TRSLOT(tail) == 1u) {
                        trw[3] = dbg_first_duty;
                        trw[6] = dbg_iters;  // (overwritten by the last query's finalise stamp; read when nq is small only)
                    }
                    ++tail;
                }
            }
            if (tail == nq) break;
            if (reducer) __builtin_amdgcn_s_sleep(TKSPMV_REDUCER_SLEEP);
            else __builtin_amdgcn_s_sleep(8);
        }
        return;
    }

    // ---- streaming waves ---------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);
    const uint32_t part = wave * n_wg + bid;
    uint32_t p0 = 0, np = 0;
    if (part < P0.n_parts) {
        p0 = P0.part_first[part];
        np = P0.part_count[part];
    }
    uint2 *wcand = L.u.w.cand + wave * WAVE_CAP;
    if (np == 0u) return;  // no partition (n_active does not count this wave)

    Pkt<C, VT> buf[NBUF];
    uint32_t rbs[NBUF];
    // Next packet to request: a running pointer into the stream copy of its query, a running pointer into pkt_row, and two
    // down-counters (requests left in the query, requests left in the launch): per request two pointer increments and a
    // compare; no multiply, nothing re-read from the kernel arguments (measured with rocprofv3 --pmc: the kernel issued
    // as many scalar as vector instructions, ~100 per packet, a third of them in this bookkeeping).
    uint32_t qa = 0u;
    const size_t part_off = (size_t)p0 * P0.packet_bytes;
    const uint8_t *pk_a = B.io[0].packets + part_off;
    const uint32_t *row_a = P0.pkt_row + p0;
    uint32_t left_q = np, left_all = np * nq;
#define TKSPMV_REQUEST(dst, rb_dst)                                                                                   \
    do {                                                                                                              \
        load_packet<C, VT>(pk_a, lane, dst);                                                                          \
        rb_dst = *row_a;                                                                                              \
        if (left_all > 1u) { /* past the end: the last packet is requested again (counted vmcnt) */                    \
            --left_all;                                                                                               \
            pk_a += P0.packet_bytes;                                                                                  \
            ++row_a;                                                                                                  \
            if (--left_q == 0u) {                                                                                     \
                left_q = np;                                                                                          \
                ++qa;                                                                                                 \
                pk_a = B.io[qa].packets + part_off;                                                                   \
                row_a = P0.pkt_row + p0;                                                                              \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) TKSPMV_REQUEST(buf[u], rbs[u]);
    rbs[NBUF - 1] = 0u;

    uint32_t qc = 0u, jc = 0u;  // packet being reduced
    float carry = 0.0f, min_units = 0.0f;
    uint32_t wcnt = 0u;
    bool waited = false;  // this wave has used its bounded wait for a threshold in the current query
    const bool long_partition = np * (uint32_t)(C / 4) >= 28u;  // ~14 rows finish per 256 entries: > 1.5 lists per query
    uint32_t *mp = L.misc[0];
    const float *xq = L.u.w.x[0];
    StreamParams P = P0;
    constexpr uint32_t DEFER_B = (uint32_t)BatchLds<XCOLS, C>::DEFER_B;
    static_assert(C == 4 || C == 8, "the batch kernel is built for 4 or 8 entries per lane");

    const uint32_t total = np * nq;
    for (uint32_t i0 = 0; i0 < total; i0 += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            if (i0 + (uint32_t)u >= total) break;
            const Pkt<C, VT> &cur = buf[u];
            const uint32_t rb_cur = rbs[u];
            TKSPMV_REQUEST(buf[(u + NBUF - 1) % NBUF], rbs[(u + NBUF - 1) % NBUF]);
            if (jc == 0u) {  // a new query starts: its x must have been staged
                mp = L.misc[qc & 1u];
                xq = L.u.w.x[qc & 1u];
                while (lds_load(&mp[MISC_XREADY]) != qc + 1u) __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
                min_units = __uint_as_float(lds_load(&mp[MISC_MINU]));
                if (trw && lane == 0 && TRSLOT(qc) < 3u) trw[1 + TRSLOT(qc)] = __builtin_amdgcn_s_memrealtime();
                P.ovf_cand = B.ovf_cand(qc);
                P.ovf_count = B.ovf_count(qc);
                carry = 0.0f;
                wcnt = 0u;
                waited = false;
            }
            const float tau = __uint_as_float(lds_load(&mp[MISC_TAU]));
            const RowSums<C> R = reduce_packet<C, QM>(cur, carry, xq, P0.fixed_mask);
            if (jc < DEFER_B && P0.n_sets != 0u) {
                uint32_t fl = 0u;
#pragma unroll
                for (int h = 0; h < C / 4; ++h) {
                    L.drs[wave][jc][h][lane] = make_float4(R.rs[4 * h], R.rs[4 * h + 1], R.rs[4 * h + 2], R.rs[4 * h + 3]);
                    fl |= ((R.cw[2 * h] & 0x00030003u) << (4 * h)) | ((R.cw[2 * h + 1] & 0x00030003u) << (4 * h + 2));
                }
                L.dfl[wave][jc][lane] = fl;
                if (lane == 0) L.drb[wave][jc] = rb_cur;
                const float wmax = wave_max(lane_best<C, QM>(R));
                if (lane == 0 && publishes && wmax >= min_units)
                    (void)__hip_atomic_fetch_max(&mp[MISC_GRPMAX + grp_local], order_key(wmax), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (__any(R.best_any >= tau) && !(P0.dbg_flags & 2u)) {
                float tau_now = tau;
                // Long partitions only (more rows per wave and query than its list holds: from ~1.5M rows on 256 CUs). A
                // wave that runs ahead of its workgroup's exchange has no threshold yet: every row passes, and once the
                // list is nearly full the rest would pour into the query's overflow list. It is ahead of the others anyway:
                // it waits for the threshold instead, bounded, once per query (2M rows: 40.4 against 42.1 us per query, 3M:
                // 58.5 against 60.7). On shorter partitions the list holds a whole query's rows and the wait only costs
                // the overlap of consecutive queries (1M rows, bench.py's conditions: 3-10 % slower), hence the condition.
                if (long_partition && wcnt + 2u * 64u > WAVE_CAP && tau <= min_units && P0.tau_possible && !waited) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (lds_load(&mp[MISC_TAU]) == __float_as_uint(min_units) && __builtin_amdgcn_s_memrealtime() - t0 < BATCH_TAU_WAIT)
                        __builtin_amdgcn_s_sleep(4);
                    waited = true;
                    tau_now = __uint_as_float(lds_load(&mp[MISC_TAU]));
                }
                if (tau_now == tau || __any(R.best_any >= tau_now))
                    offer_candidates<C, QM, WAVE_CAP>(P, R, rb_cur, tau_now, lane, grp_local, publishes, wcand, wcnt, mp);
            }
            if (jc + 1u == np) {  // the query ends for this wave
                if (P0.n_sets != 0u) {
                    // A workgroup that runs ahead of the others can get here before any threshold exists for this
                    // query; judging now would keep (and dump to global memory) every row it has seen. Give the
                    // exchange a moment -- bounded: after BATCH_TAU_WAIT the wave goes on without a threshold, so
                    // progress never depends on other workgroups being resident.
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (P0.tau_possible && lds_load(&mp[MISC_TAU]) == __float_as_uint(min_units) && !(P0.dbg_flags & 4u) &&
                           __builtin_amdgcn_s_memrealtime() - t0 < BATCH_TAU_WAIT)
                        __builtin_amdgcn_s_sleep(4);
                    const float tau2 = __uint_as_float(lds_load(&mp[MISC_TAU]));
                    const uint32_t nd = np < DEFER_B ? np : DEFER_B;
                    for (uint32_t d = 0; d < nd; ++d) {
                        const uint32_t c = L.dfl[wave][d][lane];
                        RowSums<C> S;
#pragma unroll
                        for (int h = 0; h < C / 4; ++h) {
                            const float4 v = L.drs[wave][d][h][lane];
                            S.rs[4 * h] = v.x;
                            S.rs[4 * h + 1] = v.y;
                            S.rs[4 * h + 2] = v.z;
                            S.rs[4 * h + 3] = v.w;
                            S.cw[2 * h] = (c >> (4 * h)) & 0x00030003u;
                            S.cw[2 * h + 1] = (c >> (4 * h + 2)) & 0x00030003u;
                        }
                        float best = -__builtin_huge_valf();
#pragma unroll
                        for (int j = 0; j < C; ++j) best = (S.end(j) && S.rs[j] > best) ? S.rs[j] : best;
                        S.best_any = best;
                        const uint32_t rb_d = __builtin_amdgcn_readfirstlane(L.drb[wave][d]);
                        if (__any(best >= tau2))
                            offer_candidates<C, QM, WAVE_CAP>(P, S, rb_d, tau2, lane, grp_local, publishes, wcand, wcnt, mp);
                    }
                }
                const float tau3 = __uint_as_float(lds_load(&mp[MISC_TAU]));
                ListScan<WAVE_CAP / 64u> LS;
                const uint32_t surv = scan_list<WAVE_CAP / 64u>(wcand, wcnt, tau3, lane, LS);
                uint32_t gbase = 0u;
                if (surv > STG_N) {  // rare: more survivors than the staging area holds go to global memory directly
                    if (lane == 0) gbase = atomicAdd(P.ovf_count, surv - STG_N);
                    gbase = __builtin_amdgcn_readfirstlane(gbase);
                }
#pragma unroll
                for (uint32_t e = 0; e < WAVE_CAP / 64u; ++e) {
                    if (LS.keep[e]) {
                        const unsigned long long v = pack_cand(LS.e[e].x, LS.e[e].y);
                        if (LS.pos[e] < STG_N) L.stg[qc & 1u][wave][LS.pos[e]] = v;
                        else if (gbase + LS.pos[e] - STG_N < P0.ovf_cap) st_agent(&P.ovf_cand[gbase + LS.pos[e] - STG_N], v);
                    }
                }
                if (surv > STG_N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // those stores precede the ticket
                if (lane == 0) L.stg_cnt[qc & 1u][wave] = surv < STG_N ? surv : STG_N;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&mp[MISC_DONE], 1u);
                if (trw && lane == 0 && TRSLOT(qc) < 3u) {
                    trw[4 + TRSLOT(qc)] = __builtin_amdgcn_s_memrealtime();
                    trw[7] = (TRSLOT(qc) == 0u ? 0ull : trw[7]) | ((unsigned long long)(((surv > 0xFFu ? 0xFFu : surv) << 8) | 0xFFu) << (16u * TRSLOT(qc))) |
                             ((unsigned long long)(tau3 <= min_units ? 1u : 0u) << (48u + TRSLOT(qc)));
                }
                ++qc;
                jc = 0u;
            } else {
                ++jc;
            }
        }
    }
#undef TKSPMV_REQUEST
}

// ------------------------------------------------------------------------------------------------------------
// Multi-query kernel: Q queries per pass over the matrix. The reference streams the matrix once per
// query vector (one x per run: host_spmv_bscsr.cpp:602-622, spmv_bscsr_top_k_multicore.cpp:87-140); here a chunk that
// has been loaded serves up to Q queries before the next one. It streams the wave-sliced ELL copy of the matrix
// (wsell.hpp): one lane owns one row, so a non-zero costs one LDS read, one multiply and one add per query and there is
// no cross-lane scan (measured first on the wave-BSCSR stream: there the segmented scan, ~12 VALU instructions per
// non-zero and query, made 4 queries per pass SLOWER per query than one query per pass -- 28 us against 21 us).
// Row sums are accumulated in the row's own entry order = the order of the reference's gold (rows of more than 64
// entries: in segments of 64): bit-identical scores.
// Q copies of x in LDS, Q accumulators per lane, Q thresholds, Q private candidate lists per wave, Q exchange-state
// sets. Launch structure = deferred selection: workgroups 0..7 select the top-k lists of the PREVIOUS group of queries
// (one query each) while workgroups 8.. stream the current one; a sequence runs as two independent chains of such
// launches on two streams. Cold start of the threshold exchange: the scores of a wave's leading slices (one float per
// lane, slice and query) wait in registers while no threshold has arrived, and are judged at the end of the partition;
// beyond that a wave waits (once, bounded) rather than judge 64 rows per slice without a threshold.
// Workgroup size: 8 streaming waves + the server; for Q = 8 (91 registers) 7 + the server, so that two workgroups fit a
// CU (the host packs the stream for that many partitions).
// ------------------------------------------------------------------------------------------------------------
constexpr int MULTI_Q_MAX = 8;
// entries of a wave's private candidate list, per query (LDS: 8 waves x Q lists)
template <int Q>
struct MultiGeom {
    static constexpr uint32_t WAVE_CAP = Q <= 4 ? 128u : 64u;
    // Slices whose scores a wave can hold back in registers (one float per lane, slice and query) while no threshold has
    // arrived yet
    static constexpr int HOLD = Q <= 2 ? 6 : 4;
};
struct MultiGroup {  // a group of queries sharing one pass; their exchange-state sets are set0 .. set0 + n_q - 1
    uint32_t n_q, set0;
    BatchIO io[MULTI_Q_MAX];
};
struct MultiParams {
    SetAddr A;
    MultiGroup cur, prev;  // prev.n_q == 0: no selection owed
    unsigned long long *scratch0;  // general-path scratch of selector q: scratch0 + q * scratch_stride
    uint64_t scratch_stride;
    const uint32_t *part_slice0;  // [n_parts] first slice of every partition
};
template <int Q>
struct MultiLds {
    union {
        struct {
            // x of the Q queries, + the two padding slots (wsell.hpp). Q = 8: interleaved, x[col][query], so that one
            // ds_read_b128 fetches a column value for four queries (a quarter of the LDS instructions)
            float x[Q * (SELL_XCOLS + 8)];
            uint2 cand[8][Q][MultiGeom<Q>::WAVE_CAP];
        } w;
        SelectShared sel;  // selector workgroup only
    } u;
    uint32_t misc[Q][MISC_WORDS];
};

__device__ __forceinline__ SelectParams select_params_of_set(const SelectParams &SP0, const SetAddr &A, uint32_t set,
                                                             const BatchIO &io) {
    SelectParams S = SP0;
    S.wg_cand = A.wg_cand(set);
    S.ovf_cand = A.ovf_cand(set);
    S.ovf_count = A.ovf_count(set);
    S.gmax = A.gmax(set);
    S.tau_g = A.tau_g(set);
    S.scratch = A.scratch;
    S.unit_inv_in = nullptr;
    S.out_idx = io.out_idx;
    S.out_val = io.out_val;
    return S;
}

// The selections still owed to the last group of a sequence: one workgroup per query, each with its own general-path
// scratch. Queries that share a result buffer (the engine-owned pair: "the last query wins") are selected one after the
// other by workgroup 0 instead.
__global__ void __launch_bounds__(SEL_THREADS) select_group_kernel(const SelectParams SP0, const SetAddr A, const MultiGroup G,
                                                                   unsigned long long *scratch0, uint64_t scratch_stride,
                                                                   uint32_t serial) {
    __shared__ SelectShared S;
    const uint32_t q0 = serial ? 0u : blockIdx.x, q1 = serial ? G.n_q : blockIdx.x + 1u;
    for (uint32_t q = q0; q < q1 && q < G.n_q; ++q) {
        SelectParams P = select_params_of_set(SP0, A, G.set0 + q, G.io[q]);
        P.scratch = scratch0 + (size_t)q * scratch_stride;
        select_body(P, threadIdx.x, blockDim.x, S);
        __syncthreads();
    }
}

// Candidate path of the multi-query kernel: one finished row per lane. Same
// list discipline as offer_candidates: private list, compaction against the current threshold when full, what still
// does not fit goes to the query's overflow list with one atomic per wave.
template <uint32_t WAVE_CAP>
__device__ __forceinline__ void offer_rows(const SetAddr &A, uint32_t set, uint32_t ovf_cap, float score, uint32_t pos_of_slice,
                                           float tau, uint32_t lane, uint32_t grp_local, bool publishes, uint2 *wcand,
                                           uint32_t &wcnt, uint32_t *misc, unsigned long long *dbg = nullptr) {
    // The candidate's id is its position in the stream; the selection translates it (SelectParams::pos_to_row), so this
    // path touches no global memory unless a list overflows. Lanes without a row (and the leading lanes of a row that
    // spans several) hold -inf.
    const uint32_t r = pos_of_slice + lane;
    const bool pass = score >= tau && score > -__builtin_huge_valf();
    const uint64_t pb = __ballot(pass);
    if (pb == 0ull) return;
    const uint32_t total = (uint32_t)__popcll(pb);
    const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(pb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pb, 0u));
    const float wmax = wave_max(pass ? score : -__builtin_huge_valf());
    if (lane == 0 && publishes)
        (void)__hip_atomic_fetch_max(&misc[MISC_GRPMAX + grp_local], order_key(wmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (dbg && lane == 0) {  // TKSPMV_STATS=1
        atomicAdd(&dbg[0], 1ull);
        atomicAdd(&dbg[1], (unsigned long long)total);
        if (tau <= 0.0f) atomicAdd(&dbg[2], (unsigned long long)total);
    }
    if (wcnt + total > WAVE_CAP) wcnt = compact_list<WAVE_CAP / 64u>(wcand, wcnt, tau, lane);
    const uint32_t base = wcnt;
    const uint32_t first_ovf = base < WAVE_CAP ? WAVE_CAP : base;
    uint32_t gbase = 0u;
    if (base + total > WAVE_CAP) {
        if (dbg && lane == 0) atomicAdd(&dbg[3], (unsigned long long)(base + total - first_ovf));
        if (lane == 0) gbase = atomicAdd(A.ovf_count(set), base + total - first_ovf);
        gbase = __builtin_amdgcn_readfirstlane(gbase);
    }
    if (pass) {
        const uint32_t pos = base + slot;
        if (pos < WAVE_CAP) {
            wcand[pos] = make_uint2(__float_as_uint(score), r);
        } else {
            const uint32_t gp = gbase + (pos - first_ovf);
            if (gp < ovf_cap) st_agent(&A.ovf_cand(set)[gp], pack_cand(__float_as_uint(score), r));
        }
    }
    wcnt = base + total < WAVE_CAP ? base + total : WAVE_CAP;
}

template <int Q>
__global__ void __launch_bounds__(576, Q >= 8 ? 4 : 6) multi_kernel(const StreamParams P0, const SelectParams SP0, const MultiParams M) {
    constexpr int C = 4, NBUF = 3, DEFER_S = MultiGeom<Q>::HOLD;
    constexpr uint32_t MULTI_WAVE_CAP = MultiGeom<Q>::WAVE_CAP;
    __shared__ MultiLds<Q> L;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x < (uint32_t)MULTI_Q_MAX) {
        // selector workgroups: workgroup q selects query q of the previous group (all of them at once: one after the other
        // in ONE workgroup they took longer than the pass they ride in)
        if (blockIdx.x < M.prev.n_q) {
            SelectParams S = select_params_of_set(SP0, M.A, M.prev.set0 + blockIdx.x, M.prev.io[blockIdx.x]);
            S.scratch = M.scratch0 + (size_t)blockIdx.x * M.scratch_stride;
            select_body(S, tid, blockDim.x, L.u.sel);
        }
        return;
    }
    const uint32_t bid = blockIdx.x - (uint32_t)MULTI_Q_MAX, n_wg = gridDim.x - (uint32_t)MULTI_Q_MAX;
    const uint32_t nwaves = (blockDim.x >> 6) - 1u;  // streaming waves
    const bool is_server = (wave == nwaves);
    const uint32_t nq = M.cur.n_q < (uint32_t)Q ? M.cur.n_q : (uint32_t)Q;
    const uint32_t set0 = M.cur.set0;
    const uint32_t grp_local = is_server ? 0u : wave * P0.gpw / nwaves;
    const bool publishes = (bid * P0.gpw + grp_local) < P0.n_groups_pub;
    const bool reducer = bid < P0.n_reducers;
    const float min_units = P0.min_score;  // fp32 values only: one score unit is 1.0
    // Every server wave outranks the streaming waves here: with several queries per chunk those hardly ever wait for
    // memory, and a server at a lower priority does not get to publish its workgroup's maxima (or to fetch the threshold)
    // before most of the pass is over. The servers sleep between rounds, so they take few issue slots.
    if (is_server) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);

    // This wave's partition of the wave-sliced ELL stream; its first chunks are requested before x is staged.
    uint32_t p0 = 0, np = 0, slice = 0;
    {
        const uint32_t part = is_server ? P0.n_parts : wave * n_wg + bid;
        if (part < P0.n_parts) {
            p0 = P0.part_first[part];
            np = P0.part_count[part];
            slice = M.part_slice0[part];
        }
    }
    const uint8_t *pk = M.cur.io[0].packets + (size_t)p0 * P0.packet_bytes;
    Pkt<C, 0> buf[NBUF];
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) {
        if (np > 0u) {
            const uint32_t iu = ((uint32_t)u < np) ? (uint32_t)u : (np - 1u);
            load_packet<C, 0>(pk + (size_t)iu * P0.packet_bytes, lane, buf[u]);
        }
    }

    for (uint32_t i = tid; i < (uint32_t)Q * MISC_WORDS; i += blockDim.x)
        (&L.misc[0][0])[i] = (i % MISC_WORDS) == (uint32_t)MISC_TAU ? __float_as_uint(min_units) : 0u;
    constexpr bool IL = Q >= 8;  // interleaved x: measured faster for 8 queries (5.99 against 7.09 us per query), slower for 4 (9.43 against 7.59)
    auto x_slot = [&](uint32_t q, uint32_t col) __attribute__((always_inline)) -> float & {
        return L.u.w.x[IL ? col * (uint32_t)Q + q : q * (SELL_XCOLS + 8u) + col];
    };
    for (uint32_t q = 0; q < (uint32_t)Q; ++q) {  // queries beyond nq (a partial group): zeros, their sums are never looked at
        const float *xg = M.cur.io[q < nq ? q : 0u].x;
        for (uint32_t i = tid; i < SELL_XCOLS; i += blockDim.x) x_slot(q, i) = (i < P0.cols && q < nq) ? xg[i] : 0.0f;
        if (tid == 0) {
            x_slot(q, SELL_PAD_NEUTRAL) = -0.0f;
            x_slot(q, SELL_PAD_ONE) = 1.0f;
        }
    }
    __syncthreads();

    if (is_server) {
        // Threshold exchange of all nq queries at once: lane l serves (query l / 8, local group l % 8), so a round costs one
        // store and one load round trip however many queries share the pass (query by query, a round took nq round trips
        // and a threshold needed three rounds -- publish, reduce, fetch -- to reach a workgroup: most of the pass). Each
        // reducer workgroup searches the k-th largest maximum of ONE query per round (~2.5 us).
        const uint32_t q_l = lane >> 3, g_l = lane & 7u;
        const bool q_ok = q_l < nq;
        uint32_t *mp_l = L.misc[q_ok ? q_l : 0u];
        const uint32_t grp = bid * P0.gpw + g_l;
        const bool pub_lane = q_ok && g_l < P0.gpw && grp < P0.n_groups_pub;
        uint32_t *gmax_l = M.A.gmax(set0 + (q_ok ? q_l : 0u));
        uint32_t *tau_g_l = M.A.tau_g(set0 + (q_ok ? q_l : 0u));
        const uint32_t rq = bid % nq;  // the query this workgroup reduces (if it is a reducer)
        auto publish_all = [&]() __attribute__((always_inline)) {
            if (pub_lane) {
                const uint32_t key = lds_load(&mp_l[MISC_GRPMAX + g_l]);
                if (key > mp_l[MISC_PUBLISHED + g_l]) {
                    mp_l[MISC_PUBLISHED + g_l] = key;
                    __hip_atomic_store(&gmax_l[grp], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // single writer per slot
                }
            }
        };
        for (;;) {
            publish_all();
            if (reducer) {
                StreamParams P = P0;
                P.gmax = M.A.gmax(set0 + rq);
                TauRegs tr_;
                tau_issue(P, lane, tr_);
                const float t = tau_from_maxima(P, tr_, min_units);
                if (lane == 0 && t > min_units)
                    __hip_atomic_fetch_max(M.A.tau_g(set0 + rq), order_key(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (q_ok && g_l == 0u) {
                const uint32_t kx = __hip_atomic_load(tau_g_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float t = kx ? key_to_float(kx) : min_units;
                const float cur_tau = __uint_as_float(lds_load(&mp_l[MISC_TAU]));
                if (t > cur_tau) __hip_atomic_store(&mp_l[MISC_TAU], __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // every streaming wave counts itself out and none of them waits for this wave: the loop always ends
            if (__builtin_amdgcn_readfirstlane(lds_load(&L.misc[0][MISC_DONE])) >= nwaves) break;
            __builtin_amdgcn_s_sleep(8);
        }
        publish_all();  // the workgroup's complete maxima (fire and forget)
        return;
    }

    // ---- streaming waves ---------------------------------------------------------------------------------------
    float acc[Q];              // this lane's row of the current slice, one running sum per query
    float held[DEFER_S][Q];    // scores of the partition's first slices, judged at the end
    uint32_t wcnt[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        acc[q] = 0.0f;
        wcnt[q] = 0u;
#pragma unroll
        for (int d = 0; d < DEFER_S; ++d) held[d][q] = -__builtin_huge_valf();
    }
    // true while some query of the pass has no threshold yet (its LDS word still holds the minimum score)
    auto no_tau = [&]() __attribute__((always_inline)) -> bool {
        bool missing = false;
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if ((uint32_t)q < nq) missing = missing || lds_load(&L.misc[q][MISC_TAU]) == __float_as_uint(min_units);
        return missing;
    };
    bool gave_up = false;
    // chunk the next prefetch reads: NBUF - 2 ahead of the last one requested above (a running pointer: no multiply per chunk)
    const uint8_t *pk_ahead = pk + (size_t)(np > (uint32_t)(NBUF - 2) ? (uint32_t)(NBUF - 2) : (np > 0u ? np - 1u : 0u)) * P0.packet_bytes;
    uint32_t n_done = 0u;  // slices finished by this wave
    uint32_t n_held = 0u;  // of which held back (the first n_held of the partition)
    for (uint32_t i0 = 0; i0 < np; i0 += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            const uint32_t i = i0 + (uint32_t)u;
            if (i >= np) break;
            const Pkt<C, 0> &cur = buf[u];
            {  // unconditional (pointer clamped to the last chunk): a fixed number of younger loads => counted vmcnt
                if (i + (NBUF - 1) < np) pk_ahead += P0.packet_bytes;
                load_packet<C, 0>(pk_ahead, lane, buf[(u + NBUF - 1) % NBUF]);
            }
            uint32_t off[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const uint32_t word = cur.cw[j >> 1];
                off[j] = (j & 1) ? ((word >> 16) & 0xFFFCu) : (word & 0xFFFCu);  // byte offset of x[col]
            }
            if (IL) {
                // Two queries per VALU instruction: v_pk_mul_f32 / v_pk_add_f32 work on a pair of fp32 lanes each, every
                // product and every sum rounded on its own exactly like the scalar forms (the file is built with
                // -ffp-contract=off: no fused multiply-add is formed).
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const unsigned char *xb = reinterpret_cast<const unsigned char *>(L.u.w.x);
                f32x2 a2[Q >= 2 ? Q / 2 : 1];
#pragma unroll
                for (int h = 0; h < Q / 2; ++h) a2[h] = f32x2{acc[2 * h], acc[2 * h + 1]};
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const f32x2 vv = {cur.v[j], cur.v[j]};
#pragma unroll
                    for (int h = 0; h < Q / 4; ++h) {
                        const float4 xv = *reinterpret_cast<const float4 *>(xb + off[j] * (uint32_t)Q + 16u * (uint32_t)h);
                        a2[2 * h + 0] = a2[2 * h + 0] + vv * f32x2{xv.x, xv.y};
                        a2[2 * h + 1] = a2[2 * h + 1] + vv * f32x2{xv.z, xv.w};
                    }
                }
#pragma unroll
                for (int h = 0; h < Q / 2; ++h) {
                    acc[2 * h] = a2[h].x;
                    acc[2 * h + 1] = a2[h].y;
                }
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if ((uint32_t)q < nq) {
                        const unsigned char *xb = reinterpret_cast<const unsigned char *>(L.u.w.x + (size_t)q * (SELL_XCOLS + 8u));
#pragma unroll
                        for (int j = 0; j < C; ++j)
                            acc[q] = __fadd_rn(acc[q], __fmul_rn(cur.v[j], *reinterpret_cast<const float *>(xb + off[j])));
                    }
                }
            }
            if (__builtin_amdgcn_readfirstlane(cur.cw[0]) & 1u) {  // last chunk of the slice: 64 rows are complete
                // A row of more than 64 entries spans adjacent lanes (segment index in the flag bits of this chunk, wsell.hpp):
                // its segment sums are added left to right and the score ends up on its last lane; rare.
                const uint32_t depth = ((cur.cw[0] >> 16) & 3u) | ((cur.cw[1] & 3u) << 2) | (((cur.cw[1] >> 16) & 3u) << 4);
                if (__ballot(depth != 0u) != 0ull) {
                    for (uint32_t d = 1; d < 64u; ++d) {
                        if (__ballot(depth == d) == 0ull) break;
#pragma unroll
                        for (int q = 0; q < Q; ++q) {
                            const float left = dpp_zero<DPP_WAVE_SHR1, 0xF>(acc[q]);
                            acc[q] = (depth == d) ? __fadd_rn(left, acc[q]) : acc[q];
                        }
                    }
                    const uint32_t depth_right = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)depth, DPP_WAVE_SHL1, 0xF, 0xF, true);
#pragma unroll
                    for (int q = 0; q < Q; ++q) acc[q] = depth_right != 0u ? -__builtin_huge_valf() : acc[q];  // not the row's last lane
                }
                // Hold the slice back while no threshold has arrived (cold start of the exchange) and registers are left;
                // only the partition's leading slices are held, so that their slice numbers stay implicit.
                if (n_held == n_done && n_held < (uint32_t)DEFER_S && no_tau()) {
                    ++n_held;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if ((uint32_t)q < nq) {
#pragma unroll
                            for (int d = 0; d < DEFER_S; ++d) held[d][q] = (n_done == (uint32_t)d) ? acc[q] : held[d][q];
                            const float wmax = wave_max(acc[q]);
                            if (lane == 0 && publishes && wmax >= min_units)
                                (void)__hip_atomic_fetch_max(&L.misc[q][MISC_GRPMAX + grp_local], order_key(wmax), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                } else {
                    // Nothing is judged without a threshold: 64 rows per slice and query would all pass, fill the list and
                    // pour into the overflow list. With several queries per chunk the pass is bound by instruction issue, so
                    // a wave that waits here leaves its issue slots to the others; bounded, so that progress never depends
                    // on the exchange.
                    if (P0.tau_possible && !gave_up && no_tau()) {
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        while (no_tau() && __builtin_amdgcn_s_memrealtime() - t0 < FLUSH_TAU_WAIT) __builtin_amdgcn_s_sleep(8);
                        gave_up = no_tau();  // one bounded wait per pass: a threshold that cannot form must not cost one per slice
                        if (P0.dbg && lane == 0) {
                            atomicAdd(&P0.dbg[8], 1ull);
                            atomicAdd(&P0.dbg[9], __builtin_amdgcn_s_memrealtime() - t0);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        if ((uint32_t)q < nq) {
                            const float tau = __uint_as_float(lds_load(&L.misc[q][MISC_TAU]));
                            if (__any(acc[q] >= tau) && !(P0.dbg_flags & 2u))
                                offer_rows<MULTI_WAVE_CAP>(M.A, set0 + q, P0.ovf_cap, acc[q], (slice + n_done) * 64u, tau,
                                                           lane, grp_local, publishes, L.u.w.cand[wave][q], wcnt[q], L.misc[q], P0.dbg);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[q] = 0.0f;
                ++n_done;
            }
        }
    }
    // The held slices. A short partition (small matrix) gets here before any threshold exists: give the exchange a moment,
    // bounded, and only where a threshold can form at all.
    if (np > 0u) {
        if (P0.tau_possible && !gave_up) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (no_tau() && __builtin_amdgcn_s_memrealtime() - t0 < FLUSH_TAU_WAIT) __builtin_amdgcn_s_sleep(4);
        }
#pragma unroll
        for (int d = 0; d < DEFER_S; ++d) {
            if ((uint32_t)d < n_held) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if ((uint32_t)q < nq) {
                        const float tau = __uint_as_float(lds_load(&L.misc[q][MISC_TAU]));
                        if (__any(held[d][q] >= tau) && !(P0.dbg_flags & 2u))
                            offer_rows<MULTI_WAVE_CAP>(M.A, set0 + q, P0.ovf_cap, held[d][q], (slice + (uint32_t)d) * 64u, tau,
                                                       lane, grp_local, publishes, L.u.w.cand[wave][q], wcnt[q], L.misc[q],
                                                       P0.dbg ? P0.dbg + 4 : nullptr);
                    }
                }
            }
        }
    }
    if (lane == 0) atomicAdd(&L.misc[0][MISC_DONE], 1u);

    // ---- flush: what still clears the final threshold leaves the wave's lists (first survivor to the wave's slot,
    // further ones to the query's overflow list); complete at the end of the launch, selected by the next launch.
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if ((uint32_t)q < nq) {
            const float tau = __uint_as_float(lds_load(&L.misc[q][MISC_TAU]));
            ListScan<MULTI_WAVE_CAP / 64u> LS;
            const uint32_t surv = scan_list<MULTI_WAVE_CAP / 64u>(L.u.w.cand[wave][q], wcnt[q], tau, lane, LS);
            if (surv != 0u) {
                uint32_t gbase = 0u;
                if (surv > 1u) {
                    if (lane == 0) gbase = atomicAdd(M.A.ovf_count(set0 + q), surv - 1u);
                    gbase = __builtin_amdgcn_readfirstlane(gbase);
                }
                unsigned long long *slot = M.A.wg_cand(set0 + q) + (size_t)bid * WG_SLOTS + wave;
                unsigned long long *ovf = M.A.ovf_cand(set0 + q);
#pragma unroll
                for (uint32_t e = 0; e < MULTI_WAVE_CAP / 64u; ++e) {
                    if (LS.keep[e]) {
                        const unsigned long long v = pack_cand(LS.e[e].x, LS.e[e].y);
                        if (LS.pos[e] == 0u) st_agent(slot, v);
                        else if (gbase + LS.pos[e] - 1u < P0.ovf_cap) st_agent(&ovf[gbase + LS.pos[e] - 1u], v);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Large k: scores + radix select. The threshold exchange needs k well below the number of publishing groups (at most
// 1024); beyond that the k-th largest group maximum is a weak bound (k = 500: 96 us per query, k = 1000: 207 us) and
// for k = 1023, 1024 it cannot form at all (every row becomes a candidate: 33 ms). Such engines take the reference GPU
// host's route (host_spmv_topk_csr_gpu.cu:171-231: full y, then a selection over all rows), with a selection that is
// not a sort: the SpMV-only variant of the stream kernel writes every row's score, four 8-bit histogram passes over the
// order keys find the k-th largest key T exactly, a filter pass appends the rows with key >= T (k of them plus ties) to
// the overflow list, and the ordinary selection kernel ranks those (score desc, row desc). Rows below min_score and
// rows without entries (their score slot keeps -inf) never count.
// ------------------------------------------------------------------------------------------------------------
struct RadixParams {
    const float *scores;  // [rows]; -inf where a row has no entry
    uint32_t rows, k;
    uint32_t kmin;        // order key of min_score: keys below it are not eligible
    uint32_t *hist;       // [4][256], zeroed before the first pass
    unsigned long long *ovf_cand;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
};
constexpr uint32_t RADIX_THREADS = 1024;

// From the histograms of passes 0 .. n_pass-1: the key prefix decided so far and how many keys of the next pass's bins
// are still wanted. Called by wave 0; take_all: fewer eligible keys than k exist (every eligible row is a result).
__device__ __forceinline__ void radix_decide(const RadixParams &R, int n_pass, uint32_t lane, uint32_t &prefix, uint32_t &k_rem,
                                             bool &take_all) {
    prefix = 0u;
    k_rem = R.k;
    take_all = false;
    for (int q = 0; q < n_pass; ++q) {
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = __hip_atomic_load(&R.hist[q * 256 + 4 * (int)lane + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t tot = h[0] + h[1] + h[2] + h[3];
        uint32_t above = tot;  // inclusive suffix sum over lanes >= this one ...
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_down((int)above, d);
            above += (lane + (uint32_t)d < 64u) ? o : 0u;
        }
        above -= tot;  // ... made exclusive: keys in the bins of higher lanes
        uint32_t hit_bin = 0xFFFFFFFFu, hit_above = 0u;
        uint32_t c = above;
#pragma unroll
        for (int j = 3; j >= 0; --j) {  // bins from the top
            if (hit_bin == 0xFFFFFFFFu && c < k_rem && c + h[j] >= k_rem) {
                hit_bin = 4u * lane + (uint32_t)j;
                hit_above = c;
            }
            c += h[j];
        }
        const uint64_t hb = __ballot(hit_bin != 0xFFFFFFFFu);
        if (hb == 0ull) {  // fewer than k_rem keys left (only possible in pass 0: the bins of a later pass hold >= k_rem)
            take_all = true;
            return;
        }
        const int src = __builtin_ctzll(hb);
        const uint32_t bin = (uint32_t)__shfl((int)hit_bin, src);
        k_rem -= (uint32_t)__shfl((int)hit_above, src);
        prefix = (prefix << 8) | bin;
    }
}

__global__ void __launch_bounds__(RADIX_THREADS) radix_hist_kernel(const RadixParams R, const int pass) {
    __shared__ uint32_t lh[256];
    __shared__ uint32_t sh_prefix, sh_take_all;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 256u) lh[tid] = 0u;
    if (tid < 64u) {
        uint32_t prefix, k_rem;
        bool take_all;
        radix_decide(R, pass, lane, prefix, k_rem, take_all);
        if (tid == 0u) {
            sh_prefix = prefix;
            sh_take_all = take_all ? 1u : 0u;
        }
    }
    __syncthreads();
    if (sh_take_all) return;
    const uint32_t prefix = sh_prefix, shift = 24u - 8u * (uint32_t)pass;
    for (uint32_t i = blockIdx.x * RADIX_THREADS + tid; i < R.rows; i += gridDim.x * RADIX_THREADS) {
        const float sc = R.scores[i];
        const uint32_t key = order_key(sc);
        // (a row without entries keeps -inf in its slot and is never eligible, whatever min_score is)
        if (key >= R.kmin && sc > -__builtin_huge_valf() && (pass == 0 || (key >> (shift + 8u)) == prefix))
            atomicAdd(&lh[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256u && lh[tid] != 0u) atomicAdd(&R.hist[pass * 256 + (int)tid], lh[tid]);
}

__global__ void __launch_bounds__(RADIX_THREADS) radix_filter_kernel(const RadixParams R) {
    __shared__ uint32_t sh_thr;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 64u) {
        uint32_t prefix, k_rem;
        bool take_all;
        radix_decide(R, 4, lane, prefix, k_rem, take_all);
        if (tid == 0u) sh_thr = (take_all || prefix < R.kmin) ? R.kmin : prefix;
    }
    __syncthreads();
    const uint32_t thr = sh_thr;
    const uint32_t n_iter = (R.rows + gridDim.x * RADIX_THREADS - 1u) / (gridDim.x * RADIX_THREADS);  // uniform trip count
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t i = (it * gridDim.x + blockIdx.x) * RADIX_THREADS + tid;
        const float sc = i < R.rows ? R.scores[i] : -__builtin_huge_valf();
        const bool keep = i < R.rows && order_key(sc) >= thr && sc > -__builtin_huge_valf();
        const uint64_t bm = __ballot(keep);
        uint32_t base = 0u;
        if (lane == 0u && bm) base = atomicAdd(R.ovf_count, (uint32_t)__popcll(bm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) {
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (pos < R.ovf_cap) R.ovf_cand[pos] = pack_cand(__float_as_uint(sc), i);
        }
    }
}

// Empty kernel with the stream kernel's geometry: calibrates what an event bracket adds around one launch.
__global__ void __launch_bounds__(576) null_kernel(const uint32_t *p) {
    if (p == nullptr && threadIdx.x == 123456u) __builtin_trap();
}

// ------------------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            err = std::string(#expr) + " failed: " + hipGetErrorString(_e);                             \
            return TKSPMV_ERR_DEVICE;                                                                   \
        }                                                                                               \
    } while (0)

struct EngineImpl {
    tkspmv_desc desc{};
    PackedMatrix pm;  // host copy is dropped after upload (only the small tables are kept)
    tkspmv_info info{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    // device buffers
    uint8_t *d_packets = nullptr;
    std::vector<uint8_t *> d_replicas;  // extra copies of the packet stream (cache-defeat mode)
    mutable uint64_t launch_counter = 0;
    uint32_t *d_pkt_row = nullptr, *d_part_first = nullptr, *d_part_count = nullptr;
    float *d_x = nullptr;
    const float *d_x_cur = nullptr;
    // Exchange state of one query in flight (published maxima, threshold word, survivor slots, overflow list).
    // Two sets: with deferred selection, launch q+1 streams into one set while its workgroup 0 selects query q from
    // the other. Everything else uses set 0.
    struct ExState {
        uint32_t *tau_g = nullptr, *gmax = nullptr, *ovf_count = nullptr;
        unsigned long long *wg_cand = nullptr, *ovf = nullptr, *scratch = nullptr;
        float *unit_inv = nullptr;
    };
    static constexpr int N_STATE = BATCH_MAX;  // deferred selection uses sets 0/1, the batch kernel one per query
    static constexpr uint32_t GMAX_WORDS = MAX_GM * 64, STATE_WORD_STRIDE = 64;  // set-to-set distances (elements)
    ExState st[N_STATE];
    mutable int cur_set = 0;                  // set the next deferred launch streams into
    mutable bool pending = false;             // a deferred selection is owed for ...
    mutable int pending_set = 0;              // ... this set, into ...
    mutable uint32_t *pending_idx = nullptr;  // ... these result buffers
    mutable float *pending_val = nullptr;
    uint32_t *d_out_idx = nullptr;
    uint32_t *d_done = nullptr;
    bool fused = true;
    bool can_defer = false;
    bool can_batch = false;         // batch kernel usable (exchange on, x double-buffered in LDS, 4 entries per lane)
    // Multi-query passes (multi_kernel, desc.multi_q): fp32 values, <= 1024 columns, exchange on. multi_q queries share one
    // pass over the wave-sliced ELL copy of the matrix (wsell.hpp); a group's selection is owed to the next launch (or to
    // drain()). Groups alternate between the exchange-state sets [0, MULTI_Q_MAX) and [MULTI_Q_MAX, 2 * MULTI_Q_MAX).
    bool can_multi = false;
    int multi_q = 0;
    // Large k (see radix_hist_kernel): every query = scores kernel + radix select + the selection kernel
    bool use_radix = false;
    float *d_rscores = nullptr;   // [rows], -inf where a row has no entry (never written by the scores kernel)
    uint32_t *d_rhist = nullptr;  // [4][256]
    uint8_t *d_sell_packets = nullptr;
    std::vector<uint8_t *> d_sell_replicas;
    uint32_t *d_sell_rows = nullptr, *d_sell_part_first = nullptr, *d_sell_part_count = nullptr, *d_sell_part_slice0 = nullptr;
    uint32_t sell_parts = 0, multi_stream_waves = 8;
    uint64_t sell_bytes = 0;
    uint32_t *d_multi_out_idx = nullptr;  // [2 * MULTI_Q_MAX][k] results of tkspmv_time_multi
    float *d_multi_out_val = nullptr;
    unsigned long long *d_multi_scratch = nullptr;  // [MULTI_Q_MAX] general-path scratches (the selectors of a group run at once)
    // Two independent chains of multi-query launches (TKSPMV_MULTI_CHAINS=1 switches the second off): chain c runs on its own
    // stream with its own exchange-state sets [16c, 16c + 16), so the start-up of one chain's launch fills the tail of the
    // other's (a launch still selects the previous group of ITS chain). Measured: 6.26 against 7.84 us per query at 4
    // queries per pass, 5.31 against 5.98 at 8.
    mutable MultiGroup pending_group[2]{};
    mutable int multi_parity[2] = {0, 0};
    int multi_chains = 2;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t *d_tickets = nullptr;  // [BATCH_MAX] x 32 words
    uint32_t groups_with_rows = 0;  // publishing groups that own at least one wave partition
    uint32_t n_reducers = 0;  // TKSPMV_REDUCERS (tuning): workgroups whose server derives tau from all maxima itself
    float *d_out_val = nullptr, *d_scores = nullptr;
    unsigned long long *d_stats = nullptr;
    uint32_t grid = 0, block = 0, gpw = 1, n_sets = 0, n_groups_pub = 0, cand_cap = 0, ovf_cap = 0, lds_bytes = 0,
             xcols = 1024;
    bool collect_stats = false;
    bool q8 = false;
    bool collect_stamps = false;
    unsigned long long *d_trace = nullptr;  // TKSPMV_TRACE=1: 4 launches x [grid+1][9][8] stamps
    size_t trace_words = 0;
    uint32_t dbg_flags = 0;
    uint32_t dbg_repeat = 0;
    bool have_query = false;
    bool ran = false;

    StreamParams stream_params(const float *x, int set = 0) const {
        StreamParams P{};
        const ExState &E = st[set];
        P.packets = d_replicas.empty() ? d_packets : d_replicas[launch_counter % d_replicas.size()];
        P.pkt_row = d_pkt_row;
        P.part_first = d_part_first;
        P.part_count = d_part_count;
        P.x = x;
        P.n_parts = (uint32_t)info.n_wave_partitions;
        P.cols = desc.cols;
        P.packet_bytes = info.packet_entries * (value_bytes(stream_precision(desc.precision)) + 2u);
        P.n_sets = n_sets;
        P.k = (uint32_t)desc.k;
        P.n_groups_pub = n_groups_pub;
        P.gpw = gpw;
        P.min_score = desc.min_score;  // converted to score units inside the kernel
        P.fixed_width = pm.fixed_width;
        P.fixed_mask = pm.fixed_width ? fixed_mask(pm.fixed_width) : 0u;
        P.gmax = E.gmax;
        P.tau_g = E.tau_g;
        P.n_reducers = n_reducers ? n_reducers : (grid < 8u ? grid : 8u);
        P.tau_possible = groups_with_rows >= (uint32_t)desc.k ? 1u : 0u;

        P.wg_cand = E.wg_cand;
        P.ovf_cand = E.ovf;
        P.ovf_count = E.ovf_count;
        P.ovf_cap = ovf_cap;
        P.scores = d_scores;
        P.fused = fused ? 1u : 0u;
        P.deferred = 0u;
        P.unit_inv_out = E.unit_inv;
        P.dbg = collect_stats ? d_stats + 4 : nullptr;
        P.stamps = collect_stamps ? d_stats + 16 : nullptr;
        P.trace = d_trace ? d_trace + (launch_counter % 4) * trace_words : nullptr;
        P.dbg_flags = dbg_flags;
        P.dbg_repeat = dbg_repeat;
        for (int r = 0; r < 4; ++r)
            P.rep_packets[r] = d_replicas.empty() ? d_packets : d_replicas[(launch_counter + r) % d_replicas.size()];
        return P;
    }
    SelectParams select_params(uint32_t *out_idx, float *out_val, int set = 0) const {
        SelectParams S{};
        const ExState &E = st[set];
        S.wg_cand = E.wg_cand;
        S.n_wg = grid;
        S.ovf_cand = E.ovf;
        S.ovf_count = E.ovf_count;
        S.ovf_cap = ovf_cap;
        S.k = (uint32_t)desc.k;
        S.first_row = desc.first_row;
        S.out_scale = desc.precision == TKSPMV_Q1_7 ? (1.0f / 128.0f)
                                                    : (desc.precision == TKSPMV_FIXED ? (1.0f / 2147483648.0f) : 1.0f);  // fused tail / unit_inv_in override it
        S.unit_inv_in = nullptr;
        S.out_idx = out_idx;
        S.out_val = out_val;
        S.gmax = E.gmax;
        S.tau_g = E.tau_g;
        S.done_count = d_done;
        S.n_groups_pub = n_groups_pub;
        S.use_gmax = (n_sets != 0u && n_groups_pub >= (uint32_t)desc.k) ? 1u : 0u;
        S.scratch = E.scratch;
        S.stats = collect_stats ? d_stats : nullptr;
        return S;
    }
    // The selection still owed to the last deferred launch, as its own small kernel.
    void drain(hipStream_t s) const {
        if (pending) {
            launch_select(pending_idx, pending_val, s, pending_set);
            pending = false;
        }
        drain_chain(0, s);
        if (pending_group[1].n_q != 0u) {  // (only between the fork and the join of launch_multi_sequence)
            drain_chain(1, side);
            (void)hipEventRecord(ev_join, side);
            (void)hipStreamWaitEvent(s, ev_join, 0);
        }
    }
    void drain_chain(int c, hipStream_t s) const {
        if (pending_group[c].n_q == 0u) return;
        SelectParams S = select_params(nullptr, nullptr, 0);
        S.pos_to_row = d_sell_rows;
        const MultiGroup &G = pending_group[c];
        const bool serial = G.n_q > 1u && G.io[0].out_idx == G.io[1].out_idx;
        hipLaunchKernelGGL(select_group_kernel, dim3(serial ? 1u : G.n_q), dim3(SEL_THREADS), 0, s, S, set_addr(0), G,
                           d_multi_scratch + (size_t)c * MULTI_Q_MAX * ((uint64_t)grid * WG_SLOTS + ovf_cap),
                           (uint64_t)grid * WG_SLOTS + ovf_cap, serial ? 1u : 0u);
        pending_group[c].n_q = 0u;
    }
    SetAddr set_addr(int s0) const {
        SetAddr A{};
        A.gmax0 = st[s0].gmax;
        A.tau_g0 = st[s0].tau_g;
        A.ovf_count0 = st[s0].ovf_count;
        A.wg_cand0 = st[s0].wg_cand;
        A.ovf_cand0 = st[s0].ovf;
        A.scratch = st[s0].scratch;
        A.unit_inv0 = st[s0].unit_inv;
        A.gmax_stride = GMAX_WORDS;
        A.word_stride = STATE_WORD_STRIDE;
        A.cand_stride = grid * WG_SLOTS;
        A.ovf_stride = ovf_cap;
        return A;
    }
    // n <= multi_q queries in ONE pass over the matrix; their selection is owed (pending_group) to the next multi launch
    // or to drain().
    void launch_multi(const float *const *xs, uint32_t *const *out_idx, float *const *out_val, int n, hipStream_t s, int chain = 0) const {
        if (pending) {  // a deferred single-query selection uses sets 0/1: settle it first
            launch_select(pending_idx, pending_val, s, pending_set);
            pending = false;
        }
        StreamParams P = stream_params(xs[0], 0);
        P.fused = 0u;
        P.part_first = d_sell_part_first;
        P.part_count = d_sell_part_count;
        P.n_parts = sell_parts;
        P.packet_bytes = SellMatrix::PACKET_BYTES;
        P.pkt_row = nullptr;
        MultiParams M{};
        M.A = set_addr(0);
        M.part_slice0 = d_sell_part_slice0;
        M.scratch0 = d_multi_scratch + (size_t)chain * MULTI_Q_MAX * ((uint64_t)grid * WG_SLOTS + ovf_cap);
        M.scratch_stride = (uint64_t)grid * WG_SLOTS + ovf_cap;
        M.prev = pending_group[chain];
        M.cur.n_q = (uint32_t)n;
        M.cur.set0 = (uint32_t)((2 * chain + multi_parity[chain]) * MULTI_Q_MAX);
        const uint8_t *pk = d_sell_replicas.empty() ? d_sell_packets : d_sell_replicas[launch_counter % d_sell_replicas.size()];
        for (int q = 0; q < n; ++q) {
            BatchIO &Q = M.cur.io[q];
            Q.x = xs[q];
            Q.packets = pk;
            Q.out_idx = out_idx[q];
            Q.out_val = out_val[q];
        }
        ++launch_counter;
        SelectParams S = select_params(nullptr, nullptr, 0);
        S.pos_to_row = d_sell_rows;
        const dim3 mblock(multi_stream_waves * 64u + 64u);
        if (multi_q <= 1) hipLaunchKernelGGL(multi_kernel<1>, dim3(grid), mblock, 0, s, P, S, M);
        else if (multi_q <= 2) hipLaunchKernelGGL(multi_kernel<2>, dim3(grid), mblock, 0, s, P, S, M);
        else if (multi_q <= 4) hipLaunchKernelGGL(multi_kernel<4>, dim3(grid), mblock, 0, s, P, S, M);
        else hipLaunchKernelGGL(multi_kernel<8>, dim3(grid), mblock, 0, s, P, S, M);
        pending_group[chain] = M.cur;
        multi_parity[chain] ^= 1;
    }
    // A sequence of queries in passes of multi_q; complete in stream order when this returns. Engines without the
    // multi-query kernel run the ordinary back-to-back sequence.
    void launch_multi_sequence(const float *const *xs, uint32_t *const *out_idx, float *const *out_val, int n, hipStream_t s) const {
        if (!can_multi) {
            launch_sequence(xs, out_idx, out_val, n, s);
            return;
        }
        // (two chains only with per-query result buffers: with the engine-owned pair "the last query wins" must hold)
        if (multi_chains < 2 || n <= 2 * multi_q || out_idx[0] == out_idx[1]) {
            for (int i = 0; i < n; i += multi_q) launch_multi(xs + i, out_idx + i, out_val + i, std::min(multi_q, n - i), s);
            drain(s);
            return;
        }
        drain(s);
        (void)hipEventRecord(ev_fork, s);
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        int g = 0;
        for (int i = 0; i < n; i += multi_q, ++g)
            launch_multi(xs + i, out_idx + i, out_val + i, std::min(multi_q, n - i), (g & 1) ? side : s, g & 1);
        drain(s);  // both chains' last selections, then the caller's stream waits for the side stream
    }
    // One query, its result complete in stream order right after these launches: the stream kernel and, unless
    // fused into its tail, the select kernel.
    void launch_query(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s) const {
        drain(s);
        if (use_radix) {
            launch_query_radix(x, out_idx, out_val, s);
            return;
        }
        launch_stream(x, out_idx, out_val, s);
        if (!fused) launch_select(out_idx, out_val, s);
    }
    typedef void (*batch_fn)(const StreamParams, const SelectParams, const BatchParams);
    batch_fn batch_kernel_for() const {  // can_batch: x of at most 1024 columns (it is held twice in LDS)
        if (desc.precision == TKSPMV_Q1_7) return &batch_kernel<4, 1024, 1>;
        if (desc.precision == TKSPMV_Q1_7_WIDE) return &batch_kernel<4, 1024, 2>;
        if (desc.precision == TKSPMV_F16) return &batch_kernel<4, 1024, 3>;
        if (desc.precision == TKSPMV_FIXED) return &batch_kernel<4, 1024, 4>;
        if (info.packet_entries == 512) return &batch_kernel<8, 1024, 0>;
        return &batch_kernel<4, 1024, 0>;
    }
    // n <= BATCH_MAX queries in one launch of the batch kernel; results complete in stream order after the launch.
    void launch_batch(const float *const *xs, uint32_t *const *out_idx, float *const *out_val, int n, hipStream_t s) const {
        drain(s);
        StreamParams P = stream_params(xs[0], 0);
        P.fused = 0u;
        SelectParams S = select_params(out_idx[0], out_val[0], 0);
        BatchParams B{};
        B.n_q = (uint32_t)n;
        B.tickets = d_tickets;
        static_cast<SetAddr &>(B) = set_addr(0);
        for (int q = 0; q < n; ++q) {
            BatchIO &Q = B.io[q];
            Q.x = xs[q];
            Q.packets = d_replicas.empty() ? d_packets : d_replicas[(launch_counter + q) % d_replicas.size()];
            Q.out_idx = out_idx[q];
            Q.out_val = out_val[q];
        }
        launch_counter += (uint64_t)n;
        hipLaunchKernelGGL(batch_kernel_for(), dim3(grid), dim3(block + 64), 0, s, P, S, B);
    }
    // A back-to-back sequence of queries given as pointer lists: batch kernel launches of up to BATCH_MAX queries
    // when it is available, else deferred selection.
    void launch_sequence(const float *const *xs, uint32_t *const *out_idx, float *const *out_val, int n, hipStream_t s) const {
        if (!can_batch) {
            for (int i = 0; i < n; ++i) launch_deferred(xs[i], out_idx[i], out_val[i], s);
            drain(s);
            return;
        }
        for (int i = 0; i < n; i += BATCH_MAX) launch_batch(xs + i, out_idx + i, out_val + i, std::min(BATCH_MAX, n - i), s);
    }
    // One query of a back-to-back sequence: its selection runs inside the NEXT deferred launch (or in drain()).
    void launch_deferred(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s) const {
        if (!can_defer) {
            launch_query(x, out_idx, out_val, s);
            return;
        }
        StreamParams P = stream_params(x, cur_set);
        P.fused = 0u;
        P.deferred = 1u;
        SelectParams S{};  // n_wg = 0: nothing owed
        if (pending) {
            S = select_params(pending_idx, pending_val, pending_set);
            S.unit_inv_in = st[pending_set].unit_inv;
        }
        ++launch_counter;
        hipLaunchKernelGGL(kernel_for(false), dim3(grid), dim3(block + 64), 0, s, P, S);
        pending = true;
        pending_set = cur_set;
        pending_idx = out_idx;
        pending_val = out_val;
        cur_set ^= 1;
    }
    typedef void (*stream_fn)(const StreamParams, const SelectParams);
    stream_fn kernel_for(bool scores) const {
        const bool c8 = info.packet_entries == 512;
        if (desc.precision == TKSPMV_Q1_7) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 1> : &stream_kernel<4, false, 1024, 1>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 1> : &stream_kernel<4, false, 4096, 1>;
            return scores ? &stream_kernel<4, true, 16384, 1> : &stream_kernel<4, false, 16384, 1>;
        }
        if (desc.precision == TKSPMV_F16) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 3> : &stream_kernel<4, false, 1024, 3>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 3> : &stream_kernel<4, false, 4096, 3>;
            return scores ? &stream_kernel<4, true, 16384, 3> : &stream_kernel<4, false, 16384, 3>;
        }
        if (desc.precision == TKSPMV_FIXED) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 4> : &stream_kernel<4, false, 1024, 4>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 4> : &stream_kernel<4, false, 4096, 4>;
            return scores ? &stream_kernel<4, true, 16384, 4> : &stream_kernel<4, false, 16384, 4>;
        }
        if (desc.precision == TKSPMV_Q1_7_WIDE) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 2> : &stream_kernel<4, false, 1024, 2>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 2> : &stream_kernel<4, false, 4096, 2>;
            return scores ? &stream_kernel<4, true, 16384, 2> : &stream_kernel<4, false, 16384, 2>;
        }
        if (c8) return scores ? &stream_kernel<8, true, 1024, 0, 2> : &stream_kernel<8, false, 1024, 0, 2>;  // (two packet buffers: 3 KB packets)
        if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024> : &stream_kernel<4, false, 1024>;
        if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096> : &stream_kernel<4, false, 4096>;
        return scores ? &stream_kernel<4, true, 16384> : &stream_kernel<4, false, 16384>;
    }
    void launch_stream(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s) const {
        StreamParams P = stream_params(x);
        SelectParams S = select_params(out_idx, out_val);
        ++launch_counter;
        hipLaunchKernelGGL(kernel_for(false), dim3(grid), dim3(block + 64), 0, s, P, S);
    }
    void launch_query_radix(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s) const {
        launch_scores(x, s, d_rscores);
        RadixParams R{};
        R.scores = d_rscores;
        R.rows = desc.rows;
        R.k = (uint32_t)desc.k;
        {  // order key of min_score (the device function's host twin)
            uint32_t u;
            std::memcpy(&u, &desc.min_score, 4);
            R.kmin = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        }
        R.hist = d_rhist;
        R.ovf_cand = st[0].ovf;
        R.ovf_count = st[0].ovf_count;
        R.ovf_cap = ovf_cap;
        (void)hipMemsetAsync(d_rhist, 0, 4 * 256 * 4, s);
        const uint32_t rgrid = std::max(1u, std::min(256u, (desc.rows + RADIX_THREADS * 4u - 1u) / (RADIX_THREADS * 4u)));
        for (int pass = 0; pass < 4; ++pass) hipLaunchKernelGGL(radix_hist_kernel, dim3(rgrid), dim3(RADIX_THREADS), 0, s, R, pass);
        hipLaunchKernelGGL(radix_filter_kernel, dim3(rgrid), dim3(RADIX_THREADS), 0, s, R);
        SelectParams S = select_params(out_idx, out_val, 0);
        S.use_gmax = 0u;  // no threshold word in this path
        S.out_scale = 1.0f;  // the scores kernel already wrote final scores
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SEL_THREADS), 0, s, S);
    }
    void launch_scores(const float *x, hipStream_t s, float *dst = nullptr) const {
        StreamParams P = stream_params(x);
        if (dst) P.scores = dst;
        SelectParams S = select_params(d_out_idx, d_out_val);
        hipLaunchKernelGGL(kernel_for(true), dim3(grid), dim3(block + 64), 0, s, P, S);
    }
    void launch_select(uint32_t *out_idx, float *out_val, hipStream_t s, int set = 0) const {
        SelectParams S = select_params(out_idx, out_val, set);
        S.unit_inv_in = st[set].unit_inv;  // written by the (unfused) stream kernel of that query
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SEL_THREADS), 0, s, S);
    }
};

uint64_t algorithmic_bytes(uint64_t nnz, uint32_t rows, uint32_t cols, uint32_t vbytes, int k) {
    // value + 16-bit column per nnz; 4 B of row-delimiting metadata per row; x once; k pairs out.
    return nnz * (uint64_t)(vbytes + 2) + (uint64_t)rows * 4 + (uint64_t)cols * vbytes + (uint64_t)k * 8;
}

void fill_info(const PackedMatrix &pm, int k, tkspmv_info *out) {
    std::memset(out, 0, sizeof(*out));
    out->rows = pm.rows;
    out->cols = pm.cols;
    out->nnz = pm.nnz;
    out->packed_entries = pm.packed_entries;
    out->packed_bytes = pm.stream_bytes() + pm.side_bytes();
    out->algorithmic_bytes = algorithmic_bytes(pm.nnz, pm.rows, pm.cols, value_bytes(pm.precision), k);
    out->n_packets = pm.n_packets;
    out->packet_entries = pm.packet_entries;
    out->n_wave_partitions = (uint32_t)pm.part_first.size();
    out->packets_per_partition = pm.packets_per_partition;
    out->k = k;
    out->precision = (int32_t)pm.precision;
    out->fixed_width = pm.fixed_width;
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

Engine::~Engine() {
    if (!impl_) return;
    EngineImpl &m = *impl_;
    (void)hipSetDevice(m.device);
    if (m.stream) (void)hipStreamSynchronize(m.stream);
    void *bufs[] = {m.d_packets, m.d_pkt_row, m.d_part_first, m.d_part_count, m.d_x,
                    m.d_out_idx, m.d_out_val, m.d_scores,     m.d_stats,      m.d_done, m.d_trace, m.d_tickets};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    {
        EngineImpl::ExState &E = m.st[0];  // the other sets point into these blocks
        void *eb[] = {E.tau_g, E.gmax, E.ovf_count, E.wg_cand, E.ovf, E.scratch, E.unit_inv};
        for (void *b : eb)
            if (b) (void)hipFree(b);
    }
    for (size_t r = 1; r < m.d_replicas.size(); ++r) (void)hipFree(m.d_replicas[r]);
    for (size_t r = 1; r < m.d_sell_replicas.size(); ++r) (void)hipFree(m.d_sell_replicas[r]);
    {
        void *sb[] = {m.d_sell_packets, m.d_sell_rows, m.d_sell_part_first, m.d_sell_part_count, m.d_sell_part_slice0, m.d_multi_scratch, m.d_multi_out_idx, m.d_multi_out_val, m.d_rscores, m.d_rhist};
        for (void *b : sb)
            if (b) (void)hipFree(b);
    }
    if (m.ev0) (void)hipEventDestroy(m.ev0);
    if (m.ev1) (void)hipEventDestroy(m.ev1);
    if (m.ev2) (void)hipEventDestroy(m.ev2);
    if (m.ev_fork) (void)hipEventDestroy(m.ev_fork);
    if (m.ev_join) (void)hipEventDestroy(m.ev_join);
    if (m.side) {
        (void)hipStreamSynchronize(m.side);
        (void)hipStreamDestroy(m.side);
    }
    if (m.stream) (void)hipStreamDestroy(m.stream);
    delete impl_;
}

// Exchange words that other XCDs poll (published maxima, threshold word, tickets) live in fine-grained device memory:
// it is not cached in the per-XCD L2s, so a poll never hits a stale copy (with ordinary memory an agent-scope load can
// keep returning the old value until the line happens to be evicted: milliseconds on an idle L2).
static hipError_t malloc_exchange(void **p, size_t bytes) {
    if (getenv("TKSPMV_COARSE_EXCHANGE") == nullptr &&
        hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) == hipSuccess)
        return hipSuccess;
    if (getenv("TKSPMV_DEBUG_OCC")) fprintf(stderr, "[tkspmv] fine-grained allocation unavailable, using hipMalloc\n");
    (void)hipGetLastError();
    return hipMalloc(p, bytes);
}

static int create_impl(const tkspmv_desc &d, EngineImpl &m, std::string &err, const PackedMatrix *prepacked = nullptr) {
    if (d.k < 1 || d.k > TKSPMV_MAX_K) {
        err = "k must be in [1, 1024]";
        return TKSPMV_ERR_INVALID;
    }
    if (d.cols == 0 || d.cols > TKSPMV_MAX_COLS) {
        err = "cols must be in [1, 16384]";
        return TKSPMV_ERR_INVALID;
    }
    if (d.precision != TKSPMV_F32 && d.precision != TKSPMV_Q1_7 && d.precision != TKSPMV_Q1_7_WIDE &&
        d.precision != TKSPMV_F16 && d.precision != TKSPMV_FIXED) {
        err = "unknown precision";
        return TKSPMV_ERR_INVALID;
    }
    if (d.precision == TKSPMV_FIXED ? (d.fixed_width != 0 && (d.fixed_width < 8 || d.fixed_width > 32)) : d.fixed_width != 0) {
        err = "fixed_width must be 0 (= 32) or in [8, 32] for TKSPMV_FIXED, and 0 for every other precision";
        return TKSPMV_ERR_INVALID;
    }
    if (d.precision != TKSPMV_F32 && d.nnz_per_lane == 8) {
        err = "the reduced precisions are built for nnz_per_lane = 4 only";
        return TKSPMV_ERR_UNSUPPORTED;
    }
    m.q8 = d.precision == TKSPMV_Q1_7 || d.precision == TKSPMV_Q1_7_WIDE;
    if (d.partitions > 1) {
        // The reference keeps k_per_partition (its compile-time K) candidates per row partition and merges them on
        // the host (host_spmv_bscsr.cpp:399-448). For k <= k_per_partition the union of the per-partition lists
        // contains the global top-k, so the partitioned result IS the exact one computed here. The lossy regime
        // (k > k_per_partition) is an accuracy knob of the FPGA design that is not reproduced.
        const int kpp = d.k_per_partition > 0 ? d.k_per_partition : d.k;
        if (d.k > kpp) {
            err = "partitions > 1 with k > k_per_partition (the reference's lossy regime) is not implemented";
            return TKSPMV_ERR_UNSUPPORTED;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        err = "no HIP device available (this engine has no CPU fallback)";
        return TKSPMV_ERR_DEVICE;
    }
    int dev = d.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) {
        err = "device ordinal out of range";
        return TKSPMV_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(dev));
    m.device = dev;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const uint32_t num_cus = (uint32_t)prop.multiProcessorCount;

    m.desc = d;
    m.desc.row = m.desc.col = nullptr;
    m.desc.val = nullptr;
    m.block = d.threads_per_wg > 0 ? (uint32_t)d.threads_per_wg : 512u;
    if (m.block % 64 || m.block > 512 || m.block < 64) {
        err = "threads_per_wg must be a multiple of 64, at most 512";
        return TKSPMV_ERR_INVALID;
    }
    const uint32_t waves_per_cu = d.waves_per_cu > 0 ? (uint32_t)d.waves_per_cu : 16u;
    const uint32_t waves_per_wg = m.block / 64;  // streaming waves; one more wave per workgroup serves the exchange
    m.grid = std::max(1u, num_cus * waves_per_cu / waves_per_wg);
    if ((uint64_t)m.grid * WG_SLOTS > (uint64_t)SEL_PER_THREAD * SEL_THREADS) {
        err = "launch geometry too large: waves_per_cu * num_cus / waves_per_wg must be <= 1024 workgroups";
        return TKSPMV_ERR_INVALID;
    }
    const uint32_t C = entries_per_lane_of(d);

    int kind = 0;
    // Deferred selection gives workgroup 0 of a back-to-back launch to the previous query's selection: one
    // partition per streaming wave of the remaining grid - 1 workgroups.
    const bool defer_capable = m.grid >= 2 && (uint64_t)m.grid * WG_SLOTS <= (uint64_t)SEL_PER_THREAD * (m.block + 64);
    const uint32_t n_stream_waves = (m.grid - (defer_capable ? 1u : 0u)) * waves_per_wg;
    if (prepacked) {
        // A matrix packed earlier (tkspmv_pack / a .tkspmv file): it must describe the same problem and must not have
        // more partitions than this launch geometry has streaming waves (the batch kernel gives every wave one).
        const PackedMatrix &q = *prepacked;
        if (q.rows != d.rows || q.cols != d.cols || q.precision != stream_precision(d.precision) ||
            q.C != C || q.fixed_width != fixed_width_of(d)) {
            err = "the packed matrix does not match the descriptor (rows, cols, precision or entries per lane)";
            return TKSPMV_ERR_INVALID;
        }
        if (q.part_first.size() > n_stream_waves) {
            err = "the packed matrix has more wave partitions than this GPU's launch geometry has streaming waves: pack it "
                  "again with tkspmv_pack(desc, " + std::to_string(n_stream_waves) + ", ...)";
            return TKSPMV_ERR_UNSUPPORTED;
        }
        if (q.packets.size() != q.stream_bytes() || q.pkt_row.size() != q.n_packets) {
            err = "the packed matrix is incomplete";
            return TKSPMV_ERR_INVALID;
        }
        m.pm = q;  // copy: the engine drops its host copy of the stream after the upload
        m.desc.nnz = q.nnz;
    } else {
        std::string perr = pack_wbscsr(d.rows, d.cols, d.nnz, d.row, d.col, d.val,
                                       stream_precision(d.precision), C, n_stream_waves, 4,
                                       m.pm, kind, fixed_width_of(d));
        if (!perr.empty()) {
            err = perr;
            return kind == 2 ? TKSPMV_ERR_NOT_SORTED : TKSPMV_ERR_INVALID;
        }
    }
    fill_info(m.pm, d.k, &m.info);

    // Threshold-exchange geometry: at least k publishing groups are needed (tau = k-th largest published maximum);
    // if one group per workgroup is not enough, the waves of a workgroup are split into up to 8 groups.
    // Counted on the workgroups that stream in a sequence launch (grid - 1: workgroup 0 selects). Aim for 4k groups
    // where the 1024-word limit of the exchange allows: with k close to the number of groups the k-th largest maximum
    // is a weak bound (k = 256 on 511 groups ran 2x slower than k = 100).
    const uint32_t n_pub_wg = m.grid - (defer_capable ? 1u : 0u);
    m.gpw = 1;
    while (n_pub_wg * m.gpw < 4u * (uint32_t)d.k && m.gpw < 8 && m.gpw < waves_per_wg && (waves_per_wg % (m.gpw * 2) == 0) &&
           m.grid * m.gpw * 2 <= (uint32_t)MAX_GM * 64)
        m.gpw *= 2;
    m.n_groups_pub = std::min<uint32_t>(m.grid * m.gpw, MAX_GM * 64);
    m.n_sets = (std::min<uint32_t>(n_pub_wg * m.gpw, MAX_GM * 64) >= (uint32_t)d.k) ? 1u : 0u;  // 0: exchange disabled, every row >= min_score is a candidate
    if (!m.n_sets) m.n_groups_pub = 1;
    m.ovf_cap = std::max<uint32_t>(d.rows, 1u);
    m.xcols = d.cols <= 1024 ? 1024u : (d.cols <= 4096 ? 4096u : 16384u);
    m.cand_cap = m.xcols <= 1024 ? 2048u : 1024u;  // ListGeom<XCOLS>::CAND_CAP
    {
        // groups (workgroup, local group) whose first wave owns a partition: wave w of streaming workgroup b streams
        // partition w * n_wg + b, where n_wg is the number of streaming workgroups of a sequence launch
        const uint32_t n_wg = m.grid - (defer_capable ? 1u : 0u), n_parts = (uint32_t)m.pm.part_first.size();
        m.groups_with_rows = 0;
        for (uint32_t b = 0; b < n_wg; ++b)
            for (uint32_t g = 0; g < m.gpw; ++g) {
                const uint32_t w0 = g * waves_per_wg / m.gpw;
                if ((uint64_t)w0 * n_wg + b < n_parts && (uint64_t)b * m.gpw + g < m.n_groups_pub) ++m.groups_with_rows;
            }
    }
    if (C == 8 && d.cols > 1024) {
        err = "nnz_per_lane = 8 is only built for cols <= 1024";
        return TKSPMV_ERR_UNSUPPORTED;
    }
    m.lds_bytes = (uint32_t)std::max<size_t>(sizeof(SelectShared), (size_t)m.xcols * 4 + (size_t)m.cand_cap * 8) +
                  MISC_WORDS * 4;  // all static

    HIP_TRY(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&m.ev0));
    HIP_TRY(hipEventCreate(&m.ev1));
    HIP_TRY(hipEventCreate(&m.ev2));

    const size_t stream_bytes = std::max<size_t>(m.pm.stream_bytes(), 256);
    HIP_TRY(hipMalloc((void **)&m.d_packets, stream_bytes));
    HIP_TRY(hipMalloc((void **)&m.d_pkt_row, std::max<size_t>(m.pm.pkt_row.size(), 1) * 4));
    HIP_TRY(hipMalloc((void **)&m.d_part_first, std::max<size_t>(m.pm.part_first.size(), 1) * 4));
    HIP_TRY(hipMalloc((void **)&m.d_part_count, std::max<size_t>(m.pm.part_count.size(), 1) * 4));
    if (m.pm.stream_bytes())
        HIP_TRY(hipMemcpy(m.d_packets, m.pm.packets.data(), m.pm.stream_bytes(), hipMemcpyHostToDevice));
    if (!m.pm.pkt_row.empty())
        HIP_TRY(hipMemcpy(m.d_pkt_row, m.pm.pkt_row.data(), m.pm.pkt_row.size() * 4, hipMemcpyHostToDevice));
    if (!m.pm.part_first.empty()) {
        HIP_TRY(hipMemcpy(m.d_part_first, m.pm.part_first.data(), m.pm.part_first.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m.d_part_count, m.pm.part_count.data(), m.pm.part_count.size() * 4, hipMemcpyHostToDevice));
    }
    if (d.stream_replicas > 1 && m.pm.stream_bytes()) {
        m.d_replicas.push_back(m.d_packets);
        for (int r = 1; r < d.stream_replicas; ++r) {
            uint8_t *p = nullptr;
            HIP_TRY(hipMalloc((void **)&p, stream_bytes));
            HIP_TRY(hipMemcpy(p, m.d_packets, m.pm.stream_bytes(), hipMemcpyDeviceToDevice));
            m.d_replicas.push_back(p);
        }
    }
    // The packed stream now lives in HBM; drop the host copy.
    std::vector<uint8_t>().swap(m.pm.packets);
    std::vector<uint32_t>().swap(m.pm.pkt_row);

    HIP_TRY(hipMalloc((void **)&m.d_x, (size_t)d.cols * 4));
    HIP_TRY(hipMalloc((void **)&m.d_out_idx, (size_t)d.k * 4));
    HIP_TRY(hipMalloc((void **)&m.d_out_val, (size_t)d.k * 4));
    HIP_TRY(hipMalloc((void **)&m.d_stats, 32 * 8));
    m.collect_stats = getenv("TKSPMV_STATS") != nullptr;
    if (const char *f = getenv("TKSPMV_DBG_FLAGS")) m.dbg_flags = (uint32_t)atoi(f);
    if (const char *f = getenv("TKSPMV_DBG_REPEAT")) m.dbg_repeat = (uint32_t)atoi(f);
    HIP_TRY(malloc_exchange((void **)&m.d_done, 9 * 128));
    HIP_TRY(hipMemset(m.d_done, 0, 9 * 128));
    // Fused tail / deferred selection: one workgroup (block + 64 threads) must hold every slot in SEL_PER_THREAD
    // registers per thread.
    m.fused = (uint64_t)m.grid * WG_SLOTS <= (uint64_t)SEL_PER_THREAD * (m.block + 64);
    m.can_defer = defer_capable;
    if (const char *f = getenv("TKSPMV_FUSED")) m.fused = m.fused && atoi(f) != 0;
    if (const char *f = getenv("TKSPMV_REDUCERS")) m.n_reducers = (uint32_t)atoi(f);
    if (const char *f = getenv("TKSPMV_DEFER")) m.can_defer = m.can_defer && atoi(f) != 0;
    m.can_batch = m.can_defer && m.n_sets != 0u && m.xcols <= 1024u && (C == 4u || (C == 8u && d.precision == TKSPMV_F32));  // larger x: two workgroups no longer fit a CU
    if (const char *f = getenv("TKSPMV_BATCH")) m.can_batch = m.can_batch && atoi(f) != 0;
    // Large k: the scores + radix-select path wherever the threshold exchange is off or next to useless (k above
    // 3/8 of the publishing groups: measured cross-over on the BASELINE matrix, tools/k_probe.py). TKSPMV_RADIX=0/1 forces.
    m.use_radix = m.n_sets == 0u || (uint64_t)d.k * 8u > (uint64_t)m.n_groups_pub * 3u;
    if (const char *f = getenv("TKSPMV_RADIX")) m.use_radix = atoi(f) != 0;
    if (m.use_radix) {
        m.can_defer = m.can_batch = false;
        m.fused = false;
        const size_t n = std::max<size_t>(d.rows, 1);
        HIP_TRY(hipMalloc((void **)&m.d_rscores, n * 4));
        std::vector<float> ninf(n, -std::numeric_limits<float>::infinity());
        HIP_TRY(hipMemcpy(m.d_rscores, ninf.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_rhist, 4 * 256 * 4));
    }
    // Multi-query passes (desc.multi_q; TKSPMV_MULTI_Q overrides): a second copy of the matrix in the wave-sliced ELL layout.
    {
        int mq = d.multi_q;
        if (const char *f = getenv("TKSPMV_MULTI_Q")) mq = atoi(f);
        if (mq != 0 && mq != 1 && mq != 2 && mq != 4 && mq != 8) {
            err = "multi_q must be 0 (off), 1, 2, 4 or 8";
            return TKSPMV_ERR_INVALID;
        }
        // A threshold is the k-th largest of the published group maxima. With k above a quarter of the groups it forms late
        // and stays weak; with 8 queries per pass the private lists are half as long (64 entries) and the held slices
        // fewer: measured at k = 500 (while that kernel still ran one workgroup per CU) every wave ran into its bounded
        // wait and then poured its rows into the overflow list, 2 ms per query. Such engines run 4 queries per pass.
        if (mq == 8 && (uint32_t)d.k * 4u > m.n_groups_pub) mq = 4;
        m.multi_q = mq;
        // (k above half of the groups: the threshold is next to useless -- k = 1000 on 1024 groups measured 3.6 ms per query
        // through the multi-query kernel against 0.2 ms one query per pass; such engines keep the ordinary sequence)
        m.can_multi = mq > 0 && !m.use_radix && m.can_defer && m.n_sets != 0u && d.cols <= SELL_XCOLS && d.precision == TKSPMV_F32 && m.pm.nnz > 0 &&
                      m.grid > 2u * (uint32_t)MULTI_Q_MAX && (uint32_t)d.k * 2u <= m.n_groups_pub;
    }
    if (m.can_multi) {
        // the first MULTI_Q_MAX workgroups of a multi-query launch are its selectors, the others stream
        // 8 queries per pass need 91 registers: with 9 waves per workgroup only one workgroup fits a CU (the dispatcher wants
        // 6 waves on one SIMD for two); with 8 waves -- 7 streaming + the server -- two fit at up to 128 registers.
        m.multi_stream_waves = (m.multi_q >= 8 && waves_per_wg == 8u) ? 7u : waves_per_wg;
        const uint32_t n_multi_waves = (m.grid - (uint32_t)MULTI_Q_MAX) * m.multi_stream_waves;
        SellMatrix sm;
        std::string perr;
        if (prepacked) {  // no COO at hand: decode the packed matrix
            std::vector<uint32_t> r, c;
            std::vector<float> v;
            decode_wbscsr(*prepacked, r, c, v);
            perr = pack_wsell(d.rows, d.cols, r.size(), r.data(), c.data(), v.data(), n_multi_waves, sm);
        } else {
            perr = pack_wsell(d.rows, d.cols, d.nnz, d.row, d.col, d.val, n_multi_waves, sm);
        }
        if (!perr.empty()) {
            err = perr;
            return TKSPMV_ERR_INVALID;
        }
        m.sell_parts = (uint32_t)sm.part_first.size();
        m.sell_bytes = sm.stream_bytes() + sm.slice_rows.size() * 4 + (uint64_t)m.sell_parts * 12;
        HIP_TRY(hipMalloc((void **)&m.d_sell_packets, std::max<size_t>(sm.stream_bytes(), 256)));
        HIP_TRY(hipMemcpy(m.d_sell_packets, sm.packets.data(), sm.stream_bytes(), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_sell_rows, sm.slice_rows.size() * 4));
        HIP_TRY(hipMemcpy(m.d_sell_rows, sm.slice_rows.data(), sm.slice_rows.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_sell_part_first, (size_t)m.sell_parts * 4));
        HIP_TRY(hipMalloc((void **)&m.d_sell_part_count, (size_t)m.sell_parts * 4));
        HIP_TRY(hipMalloc((void **)&m.d_sell_part_slice0, (size_t)m.sell_parts * 4));
        HIP_TRY(hipMemcpy(m.d_sell_part_first, sm.part_first.data(), (size_t)m.sell_parts * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m.d_sell_part_count, sm.part_count.data(), (size_t)m.sell_parts * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m.d_sell_part_slice0, sm.part_slice0.data(), (size_t)m.sell_parts * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_multi_scratch, 2 * (size_t)MULTI_Q_MAX * ((size_t)m.grid * WG_SLOTS + std::max<uint32_t>(d.rows, 1u)) * 8));
        HIP_TRY(hipMalloc((void **)&m.d_multi_out_idx, 2 * (size_t)MULTI_Q_MAX * d.k * 4));
        HIP_TRY(hipMalloc((void **)&m.d_multi_out_val, 2 * (size_t)MULTI_Q_MAX * d.k * 4));
        HIP_TRY(hipStreamCreateWithFlags(&m.side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m.ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m.ev_join, hipEventDisableTiming));
        if (const char *f = getenv("TKSPMV_MULTI_CHAINS")) m.multi_chains = atoi(f) >= 2 ? 2 : 1;
        if (d.stream_replicas > 1) {
            m.d_sell_replicas.push_back(m.d_sell_packets);
            for (int r = 1; r < d.stream_replicas; ++r) {
                uint8_t *p = nullptr;
                HIP_TRY(hipMalloc((void **)&p, sm.stream_bytes()));
                HIP_TRY(hipMemcpy(p, m.d_sell_packets, sm.stream_bytes(), hipMemcpyDeviceToDevice));
                m.d_sell_replicas.push_back(p);
            }
        }
    }
    HIP_TRY(malloc_exchange((void **)&m.d_tickets, BATCH_MAX * 32 * 4));
    HIP_TRY(hipMemset(m.d_tickets, 0, BATCH_MAX * 32 * 4));
    {
        // Exchange-state sets: one block per field, set s at s strides (the batch kernel addresses them that way).
        const int n_sets_alloc = (m.can_batch || m.can_multi) ? EngineImpl::N_STATE : (m.can_defer ? 2 : 1);
        const size_t ns = (size_t)n_sets_alloc;
        EngineImpl::ExState &E0 = m.st[0];
        HIP_TRY(malloc_exchange((void **)&E0.gmax, ns * EngineImpl::GMAX_WORDS * 4));
        HIP_TRY(hipMemset(E0.gmax, 0, ns * EngineImpl::GMAX_WORDS * 4));
        HIP_TRY(malloc_exchange((void **)&E0.tau_g, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMemset(E0.tau_g, 0, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMalloc((void **)&E0.ovf_count, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMemset(E0.ovf_count, 0, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMalloc((void **)&E0.wg_cand, ns * m.grid * WG_SLOTS * 8));
        HIP_TRY(hipMemset(E0.wg_cand, 0xFF, ns * m.grid * WG_SLOTS * 8));
        HIP_TRY(hipMalloc((void **)&E0.ovf, ns * m.ovf_cap * 8));
        // the scratch of the selection's general path is used by one selection at a time: shared by all sets
        HIP_TRY(hipMalloc((void **)&E0.scratch, ((size_t)m.grid * WG_SLOTS + m.ovf_cap) * 8));
        HIP_TRY(hipMalloc((void **)&E0.unit_inv, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        std::vector<float> ones(ns * EngineImpl::STATE_WORD_STRIDE, 1.0f);
        HIP_TRY(hipMemcpy(E0.unit_inv, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
        for (int si = 1; si < n_sets_alloc; ++si) {
            EngineImpl::ExState &E = m.st[si];
            E.gmax = E0.gmax + (size_t)si * EngineImpl::GMAX_WORDS;
            E.tau_g = E0.tau_g + (size_t)si * EngineImpl::STATE_WORD_STRIDE;
            E.ovf_count = E0.ovf_count + (size_t)si * EngineImpl::STATE_WORD_STRIDE;
            E.wg_cand = E0.wg_cand + (size_t)si * m.grid * WG_SLOTS;
            E.ovf = E0.ovf + (size_t)si * m.ovf_cap;
            E.scratch = E0.scratch;
            E.unit_inv = E0.unit_inv + (size_t)si * EngineImpl::STATE_WORD_STRIDE;
        }
    }
    HIP_TRY(hipMemset(m.d_stats, 0, 32 * 8));
    m.collect_stamps = getenv("TKSPMV_STAMPS") != nullptr;
    if (getenv("TKSPMV_TRACE")) {
        m.trace_words = (size_t)(m.grid + 1) * 9 * 8;
        HIP_TRY(hipMalloc((void **)&m.d_trace, m.trace_words * 4 * 8));
        HIP_TRY(hipMemset(m.d_trace, 0, m.trace_words * 4 * 8));
    }
    HIP_TRY(hipMemset(m.d_out_idx, 0, (size_t)d.k * 4));
    HIP_TRY(hipMemset(m.d_out_val, 0, (size_t)d.k * 4));

    if (getenv("TKSPMV_DEBUG_OCC")) {
        int n1 = -1, n2 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, reinterpret_cast<const void *>(m.kernel_for(false)), (int)m.block + 64, 0);
        if (m.can_batch)
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, reinterpret_cast<const void *>(m.batch_kernel_for()), (int)m.block + 64, 0);
        fprintf(stderr, "[tkspmv] workgroups per CU by the runtime's occupancy calculator: stream kernel %d, batch kernel %d; LDS per CU %zu\n",
                n1, n2, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    }
    m.info.grid = m.grid;
    m.info.block = m.block;
    m.info.n_groups = m.n_sets ? m.n_groups_pub : 0;
    m.info.lds_bytes = m.lds_bytes;
    m.info.partitions = d.partitions > 1 ? d.partitions : 1;
    m.info.k_per_partition = d.k_per_partition > 0 ? d.k_per_partition : d.k;
    m.info.device = dev;
    m.info.num_cus = num_cus;
    m.info.multi_q = m.can_multi ? (uint32_t)m.multi_q : 0u;
    m.info.multi_bytes = m.can_multi ? m.sell_bytes : 0u;
    HIP_TRY(hipDeviceSynchronize());
    return TKSPMV_OK;
}

Engine *Engine::create(const tkspmv_desc &desc, std::string &err, int &status, const PackedMatrix *prepacked) {
    Engine *e = new Engine();
    e->impl_ = new EngineImpl();
    status = create_impl(desc, *e->impl_, err, prepacked);
    if (status != TKSPMV_OK) {
        delete e;
        return nullptr;
    }
    return e;
}

void Engine::info(tkspmv_info *out) const { *out = impl_->info; }

int wave_partitions_for(const tkspmv_desc &d, uint32_t *out, std::string &err) {
    int dev = d.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const uint32_t block = d.threads_per_wg > 0 ? (uint32_t)d.threads_per_wg : 512u;
    const uint32_t waves_per_cu = d.waves_per_cu > 0 ? (uint32_t)d.waves_per_cu : 16u;
    const uint32_t waves_per_wg = block / 64;
    const uint32_t grid = std::max(1u, (uint32_t)prop.multiProcessorCount * waves_per_cu / waves_per_wg);
    const bool defer_capable = grid >= 2 && (uint64_t)grid * WG_SLOTS <= (uint64_t)SEL_PER_THREAD * (block + 64);
    *out = (grid - (defer_capable ? 1u : 0u)) * waves_per_wg;
    return TKSPMV_OK;
}

int Engine::set_query(const float *host_x, double *elapsed_ns, std::string &err) {
    EngineImpl &m = *impl_;
    if (!host_x) {
        err = "query vector is NULL";
        return TKSPMV_ERR_INVALID;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipMemcpyAsync(m.d_x, host_x, (size_t)m.desc.cols * 4, hipMemcpyHostToDevice, m.stream));
    HIP_TRY(hipStreamSynchronize(m.stream));
    m.d_x_cur = m.d_x;
    m.have_query = true;
    if (elapsed_ns)
        *elapsed_ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::high_resolution_clock::now() - t0)
                          .count();
    return TKSPMV_OK;
}

int Engine::set_query_device(const float *dev_x, std::string &err) {
    if (!dev_x) {
        err = "query vector is NULL";
        return TKSPMV_ERR_INVALID;
    }
    impl_->d_x_cur = dev_x;
    impl_->have_query = true;
    return TKSPMV_OK;
}

int Engine::enqueue(const float *dev_x, uint32_t *dev_idx, float *dev_val, void *stream, std::string &err) {
    EngineImpl &m = *impl_;
    const float *x = dev_x ? dev_x : m.d_x_cur;
    if (!x) {
        err = "no query vector installed (call tkspmv_set_query first)";
        return TKSPMV_ERR_STATE;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    m.launch_query(x, dev_idx ? dev_idx : m.d_out_idx, dev_val ? dev_val : m.d_out_val, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

// Pointer lists of a back-to-back sequence (query i = dev_xs + (i % n_x) * cols; results to out + i * stride).
static void sequence_lists(const EngineImpl &m, const float *dev_xs, int32_t n_x, int32_t count, uint32_t *idx, float *val,
                           size_t stride, std::vector<const float *> &xs, std::vector<uint32_t *> &oi,
                           std::vector<float *> &ov) {
    xs.resize(count);
    oi.resize(count);
    ov.resize(count);
    for (int i = 0; i < count; ++i) {
        xs[i] = dev_xs + (size_t)(i % n_x) * m.desc.cols;
        oi[i] = idx + (size_t)i * stride;
        ov[i] = val + (size_t)i * stride;
    }
}

int Engine::enqueue_many(const float *dev_xs, int32_t n_x, int32_t count, void *stream, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || n_x < 1 || count < 0) {
        err = "bad arguments to enqueue_many";
        return TKSPMV_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    std::vector<const float *> xs;
    std::vector<uint32_t *> oi;
    std::vector<float *> ov;
    sequence_lists(m, dev_xs, n_x, count, m.d_out_idx, m.d_out_val, 0, xs, oi, ov);
    m.launch_sequence(xs.data(), oi.data(), ov.data(), count, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::enqueue_batch(const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream,
                          std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || count < 0 || (dev_idx == nullptr) != (dev_val == nullptr)) {
        err = "bad arguments to enqueue_batch";
        return TKSPMV_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    std::vector<const float *> xs;
    std::vector<uint32_t *> oi;
    std::vector<float *> ov;
    sequence_lists(m, dev_xs, count > 0 ? count : 1, count, dev_idx ? dev_idx : m.d_out_idx, dev_val ? dev_val : m.d_out_val,
                   dev_idx ? (size_t)m.desc.k : 0, xs, oi, ov);
    m.launch_sequence(xs.data(), oi.data(), ov.data(), count, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::enqueue_list(const float *const *dev_xs, uint32_t *const *dev_idx, float *const *dev_val, int32_t count,
                         void *stream, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || !dev_idx || !dev_val || count < 0) {
        err = "bad arguments to enqueue_list";
        return TKSPMV_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    m.launch_sequence(dev_xs, dev_idx, dev_val, count, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::enqueue_multi(const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream,
                          std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs && count == 1) dev_xs = m.d_x_cur;  // the vector installed by tkspmv_set_query / set_query_device
    if (!dev_xs || count < 0 || (dev_idx == nullptr) != (dev_val == nullptr)) {
        err = "bad arguments to enqueue_multi";
        return TKSPMV_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    std::vector<const float *> xs;
    std::vector<uint32_t *> oi;
    std::vector<float *> ov;
    sequence_lists(m, dev_xs, count > 0 ? count : 1, count, dev_idx ? dev_idx : m.d_out_idx, dev_val ? dev_val : m.d_out_val,
                   dev_idx ? (size_t)m.desc.k : 0, xs, oi, ov);
    m.launch_multi_sequence(xs.data(), oi.data(), ov.data(), count, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::enqueue_multi_list(const float *const *dev_xs, uint32_t *const *dev_idx, float *const *dev_val, int32_t count,
                               void *stream, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || !dev_idx || !dev_val || count < 0) {
        err = "bad arguments to enqueue_multi_list";
        return TKSPMV_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    m.launch_multi_sequence(dev_xs, dev_idx, dev_val, count, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::time_multi(const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || n_x < 1 || iters < 1 || !ns_per_query) {
        err = "bad arguments to time_multi";
        return TKSPMV_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(hipEventRecord(m.ev0, m.stream));
    {
        std::vector<const float *> xs;
        std::vector<uint32_t *> oi;
        std::vector<float *> ov;
        sequence_lists(m, dev_xs, n_x, iters, m.d_out_idx, m.d_out_val, 0, xs, oi, ov);
        if (m.d_multi_out_idx) {  // a result buffer per query in flight (two groups): the two chains may run
            for (int i = 0; i < iters; ++i) {
                const size_t slot = (size_t)(i % (2 * MULTI_Q_MAX)) * (size_t)m.desc.k;
                oi[i] = m.d_multi_out_idx + slot;
                ov[i] = m.d_multi_out_val + slot;
            }
        }
        m.launch_multi_sequence(xs.data(), oi.data(), ov.data(), iters, m.stream);
    }
    HIP_TRY(hipEventRecord(m.ev1, m.stream));
    HIP_TRY(hipEventSynchronize(m.ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
    *ns_per_query = (double)ms * 1e6 / iters;
    if (m.collect_stats) {  // TKSPMV_STATS=1: candidate-path counters of the multi-query kernel, per query
        unsigned long long st[32];
        HIP_TRY(hipMemcpy(st, m.d_stats, sizeof(st), hipMemcpyDeviceToHost));
        const double n = (double)iters;
        fprintf(stderr, "[tkspmv multi stats per query] judged slices: offers %.1f rows %.1f (tau<=0: %.1f) overflowed %.1f | held slices: offers %.1f rows %.1f "
                        "(tau<=0: %.1f) overflowed %.1f | waits %.1f, %.2f us each\n",
                st[4] / n, st[5] / n, st[6] / n, st[7] / n, st[8] / n, st[9] / n, st[10] / n, st[11] / n, st[12] / n,
                st[12] ? st[13] * 0.01 / st[12] : 0.0);
        HIP_TRY(hipMemset(m.d_stats, 0, 32 * 8));
    }
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::enqueue_deferred(const float *dev_x, uint32_t *dev_idx, float *dev_val, void *stream, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_x) {
        err = "query vector is NULL";
        return TKSPMV_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    m.launch_deferred(dev_x, dev_idx ? dev_idx : m.d_out_idx, dev_val ? dev_val : m.d_out_val, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::drain(void *stream, std::string &err) {
    EngineImpl &m = *impl_;
    hipStream_t s = stream ? (hipStream_t)stream : m.stream;
    HIP_TRY(hipSetDevice(m.device));
    m.drain(s);
    HIP_TRY(hipGetLastError());
    return TKSPMV_OK;
}

int Engine::run(double *kernel_ns, std::string &err) {
    EngineImpl &m = *impl_;
    if (!m.have_query) {
        err = "no query vector installed (call tkspmv_set_query first)";
        return TKSPMV_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipEventRecord(m.ev0, m.stream));
    m.launch_query(m.d_x_cur, m.d_out_idx, m.d_out_val, m.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m.ev1, m.stream));
    HIP_TRY(hipEventSynchronize(m.ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
    if (kernel_ns) *kernel_ns = (double)ms * 1e6;
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::synchronize(std::string &err) {
    HIP_TRY(hipSetDevice(impl_->device));
    impl_->drain(impl_->stream);
    HIP_TRY(hipStreamSynchronize(impl_->stream));
    return TKSPMV_OK;
}

int Engine::read(uint32_t *idx, float *val, int32_t *n, std::string &err) {
    EngineImpl &m = *impl_;
    if (!m.ran) {
        err = "no query has been run";
        return TKSPMV_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    if (idx) HIP_TRY(hipMemcpy(idx, m.d_out_idx, (size_t)m.desc.k * 4, hipMemcpyDeviceToHost));
    if (val) HIP_TRY(hipMemcpy(val, m.d_out_val, (size_t)m.desc.k * 4, hipMemcpyDeviceToHost));
    if (n) *n = m.desc.k;
    return TKSPMV_OK;
}

int Engine::read_trace(unsigned long long *host, size_t max_words, size_t *words, std::string &err) {
    EngineImpl &m = *impl_;
    if (!m.d_trace) {
        err = "tracing is off (set TKSPMV_TRACE=1 before tkspmv_create)";
        return TKSPMV_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = std::min(max_words, m.trace_words * 4);
    HIP_TRY(hipMemcpy(host, m.d_trace, n * 8, hipMemcpyDeviceToHost));
    if (words) *words = n;
    return TKSPMV_OK;
}

int Engine::result_device(const uint32_t **dev_idx, const float **dev_val) {
    if (dev_idx) *dev_idx = impl_->d_out_idx;
    if (dev_val) *dev_val = impl_->d_out_val;
    return TKSPMV_OK;
}

int Engine::scores(float *host_y, std::string &err) {
    EngineImpl &m = *impl_;
    if (!m.have_query) {
        err = "no query vector installed";
        return TKSPMV_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(m.device));
    if (!m.d_scores) HIP_TRY(hipMalloc((void **)&m.d_scores, std::max<size_t>(m.desc.rows, 1) * 4));
    HIP_TRY(hipMemsetAsync(m.d_scores, 0, std::max<size_t>(m.desc.rows, 1) * 4, m.stream));
    m.launch_scores(m.d_x_cur, m.stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_y, m.d_scores, (size_t)m.desc.rows * 4, hipMemcpyDeviceToHost, m.stream));
    HIP_TRY(hipStreamSynchronize(m.stream));
    return TKSPMV_OK;
}

int Engine::time_queries(const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || n_x < 1 || iters < 1 || !ns_per_query) {
        err = "bad arguments to time_queries";
        return TKSPMV_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(hipEventRecord(m.ev0, m.stream));
    {
        std::vector<const float *> xs;
        std::vector<uint32_t *> oi;
        std::vector<float *> ov;
        sequence_lists(m, dev_xs, n_x, iters, m.d_out_idx, m.d_out_val, 0, xs, oi, ov);
        m.launch_sequence(xs.data(), oi.data(), ov.data(), iters, m.stream);
    }
    HIP_TRY(hipEventRecord(m.ev1, m.stream));
    HIP_TRY(hipEventSynchronize(m.ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
    *ns_per_query = (double)ms * 1e6 / iters;
    m.ran = true;
    return TKSPMV_OK;
}

int Engine::profile(const float *dev_xs, int32_t n_x, int32_t iters, tkspmv_timing *out, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || n_x < 1 || iters < 1 || !out) {
        err = "bad arguments to profile";
        return TKSPMV_ERR_INVALID;
    }
    std::memset(out, 0, sizeof(*out));
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    unsigned long long st0[8], st1[8];
    HIP_TRY(hipMemcpy(st0, m.d_stats, sizeof(st0), hipMemcpyDeviceToHost));
    const size_t stride = m.desc.cols;
    // (1) whole queries back-to-back
    HIP_TRY(hipEventRecord(m.ev0, m.stream));
    {
        std::vector<const float *> xs;
        std::vector<uint32_t *> oi;
        std::vector<float *> ov;
        sequence_lists(m, dev_xs, n_x, iters, m.d_out_idx, m.d_out_val, 0, xs, oi, ov);
        m.launch_sequence(xs.data(), oi.data(), ov.data(), iters, m.stream);
    }
    HIP_TRY(hipEventRecord(m.ev1, m.stream));
    HIP_TRY(hipEventSynchronize(m.ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
    out->query_ns = (double)ms * 1e6 / iters;
    HIP_TRY(hipMemcpy(st1, m.d_stats, sizeof(st1), hipMemcpyDeviceToHost));
    out->candidates_avg = (double)(st1[0] - st0[0]) / (double)std::max<unsigned long long>(1, st1[1] - st0[1]);
    out->slow_paths_avg = (double)(st1[4] - st0[4]) / (double)std::max<unsigned long long>(1, st1[1] - st0[1]);
    out->appended_avg = (double)(st1[5] - st0[5]) / (double)std::max<unsigned long long>(1, st1[1] - st0[1]);
    // (2) per-kernel: an event before and after every stream kernel, everything enqueued back to back and one
    // host sync at the end, so the GPU never idles between launches (an idle GPU adds the dispatch latency of the
    // next kernel to the interval).
    double t_stream = 0, t_select = 0;
    {
        std::vector<hipEvent_t> evs((size_t)iters * 2 + 1);
        for (auto &e : evs) HIP_TRY(hipEventCreate(&e));
        for (int i = 0; i < iters; ++i) {
            const float *x = dev_xs + (size_t)(i % n_x) * stride;
            HIP_TRY(hipEventRecord(evs[2 * i], m.stream));
            if (m.use_radix) m.launch_query(x, m.d_out_idx, m.d_out_val, m.stream);  // (scores + radix select + selection)
            else m.launch_stream(x, m.d_out_idx, m.d_out_val, m.stream);
            HIP_TRY(hipEventRecord(evs[2 * i + 1], m.stream));
            if (!m.fused && !m.use_radix) m.launch_select(m.d_out_idx, m.d_out_val, m.stream);
        }
        HIP_TRY(hipEventRecord(evs[2 * iters], m.stream));
        HIP_TRY(hipEventSynchronize(evs[2 * iters]));
        for (int i = 0; i < iters; ++i) {
            float a = 0, b = 0;
            HIP_TRY(hipEventElapsedTime(&a, evs[2 * i], evs[2 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&b, evs[2 * i + 1], evs[2 * i + 2]));
            t_stream += a;
            t_select += b;
        }
        for (auto &e : evs) (void)hipEventDestroy(e);
    }
    // (3) SpMV-only variant (hw_spmv_only_time of the reference's GPU host): full y written, no top-k
    {
        if (!m.d_scores) HIP_TRY(hipMalloc((void **)&m.d_scores, std::max<size_t>(m.desc.rows, 1) * 4));
        HIP_TRY(hipEventRecord(m.ev0, m.stream));
        for (int i = 0; i < iters; ++i) {
            m.launch_scores(dev_xs + (size_t)(i % n_x) * stride, m.stream);
        }
        HIP_TRY(hipEventRecord(m.ev1, m.stream));
        HIP_TRY(hipEventSynchronize(m.ev1));
        float ms2 = 0;
        HIP_TRY(hipEventElapsedTime(&ms2, m.ev0, m.ev1));
        out->scores_kernel_ns = (double)ms2 * 1e6 / iters;
    }
    // (4) what the event bracket itself costs around one launch: the same bracket around an empty kernel with
    // the same geometry, enqueued the same way (followed by the select kernel so the stream stays busy).
    double t_null = 0;
    {
        const int n = iters < 200 ? iters : 200;
        std::vector<hipEvent_t> evs((size_t)n * 2);
        for (auto &e : evs) HIP_TRY(hipEventCreate(&e));
        for (int i = 0; i < n; ++i) {
            HIP_TRY(hipEventRecord(evs[2 * i], m.stream));
            hipLaunchKernelGGL(null_kernel, dim3(m.grid), dim3(m.block + 64), 0, m.stream, m.st[0].gmax);
            HIP_TRY(hipEventRecord(evs[2 * i + 1], m.stream));
            if (!m.fused && !m.use_radix) m.launch_select(m.d_out_idx, m.d_out_val, m.stream);
        }
        HIP_TRY(hipStreamSynchronize(m.stream));
        for (int i = 0; i < n; ++i) {
            float a = 0;
            HIP_TRY(hipEventElapsedTime(&a, evs[2 * i], evs[2 * i + 1]));
            t_null += a;
        }
        for (auto &e : evs) (void)hipEventDestroy(e);
        t_null = t_null * 1e6 / n;
    }
    out->event_bracket_ns = t_null;
    out->stream_kernel_ns = t_stream * 1e6 / iters;
    out->select_kernel_ns = t_select * 1e6 / iters;
    out->n_queries = (uint32_t)iters;
    if (m.collect_stats) {
        unsigned long long sx[12];
        HIP_TRY(hipMemcpy(sx, m.d_stats, sizeof(sx), hipMemcpyDeviceToHost));
        fprintf(stderr, "[tkspmv stats] selections %llu, general-path selections %llu, max candidates %llu, overflow entries per selection %.1f\n",
                sx[1], sx[3], sx[2], (double)sx[8] / (double)std::max<unsigned long long>(1, sx[1]));
    }
    if (m.collect_stamps) {
        unsigned long long st[16];
        HIP_TRY(hipMemcpy(st, m.d_stats + 16, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "[tkspmv stamps, shader cycles since stream end of the last workgroup] flush_issued %lld flush_done %lld ticket %lld loads %lld keys %lld ranked %lld end %lld\n",
                (long long)(st[1] - st[0]), (long long)(st[2] - st[0]), (long long)(st[3] - st[0]), (long long)(st[4] - st[0]),
                (long long)(st[5] - st[0]), (long long)(st[6] - st[0]), (long long)(st[7] - st[0]));
    }
    m.ran = true;
    return TKSPMV_OK;
}

}  // namespace tkspmv
