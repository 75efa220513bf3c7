// kernels/select.hpp -- The exact selection over the surviving candidates (select_body, select_kernel) and the k-th-largest search the threshold exchange uses.
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include "common.hpp"

namespace tkspmv {

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
    // Multi-query kernel: candidates carry their POSITION in the wave-sliced ELL stream (slice * 64 + lane) instead of a
    // row id; the selection, off the streaming waves' path, looks the row ids up here. NULL: candidates carry row ids.
    const uint32_t *pos_to_row;
    unsigned long long *stats;    // [0] += candidates, [1] += queries, [2] = max candidates, [3] += general-path runs
    // Low-latency result hand-over of tkspmv_run (NULL: off): the k results are ALSO written to host-visible memory --
    // host_out[0..k) row ids, [k..2k) score bits -- followed by host_out[2k] = host_epoch, which the host polls. No
    // device-to-host copy, no stream synchronisation on the host's critical path.
    uint32_t *host_out;
    uint32_t host_epoch;
    // Batch kernel: the exchange-state sets and overflow lists are reused WITHIN one launch, so the resets at the end of a selection
    // must be written through (no kernel boundary flushes them) and drained before the next user is told. t_seen: s_memrealtime
    // stamp a caller may pass for host_out[2k + 1] (the device time of the query in 10 ns ticks; 0: the launch's own start stamp).
    uint32_t wt_reset;
    unsigned long long t_seen;
    // Fused single launch with the host-visible result (tkspmv_run): workgroup 0 stamps the launch's start here (s_memrealtime,
    // 100 MHz) and the selecting workgroup reports end - start in host_out[2k + 1] -- the kernel's own duration, so that
    // tkspmv_run needs no hipEvent pair around the launch (two records and a query: ~6 us of the 67 us a query took end to end).
    unsigned long long *t_start;
};

constexpr int MAX_GM = 16;  // n_groups_pub <= 1024 => at most 16 published maxima per lane
constexpr uint32_t SEL_THREADS = 1024;
constexpr uint32_t SEL_CAP = 2048;
constexpr uint32_t SEL_PER_THREAD = 8;  // slot entries held in registers per thread

struct SelectShared {
    unsigned long long keys[SEL_CAP + 8];
    uint32_t cnt, total, thr, last, sum;
};

// Checksum of a result block as the host recomputes it before it trusts the block (tkspmv_run): the payload stores and the flag
// are relaxed system-scope stores, ordered in practice by a drain (s_waitcnt) but not by the memory model -- so the word behind
// the flag carries what the payload must add up to (plus the epoch), and the host reads until it does.
__host__ __device__ inline uint32_t result_checksum_term(uint32_t idx, uint32_t val_bits, uint32_t r) {
    return (idx ^ ((val_bits << 13) | (val_bits >> 19))) * (2u * r + 1u);
}

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

// OVF_SPEC: the first OVF_SPEC x nthreads entries of the overflow list are loaded BLINDLY with the slots (before the list's length
// is known): the single-query launches, whose tail this selection is, save the two trips through memory that counting and
// placing a short overflow list cost otherwise (it holds ~100 entries at 1M rows). 0: the batch kernel (its selections run
// beside the stream, and it has no registers to spare).
template <int OVF_SPEC = 0>
__device__ __forceinline__ void select_body(const SelectParams &P, const uint32_t tid, const uint32_t nthreads,
                                            SelectShared &S,
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
        // (clamped address, masked value: as "if in range, load" the compiler put every load into a branch of its own behind an
        //  s_waitcnt vmcnt(0) in some instantiations -- eight trips through memory one after the other per selection)
        const unsigned long long w = ld_agent(&P.wg_cand[f < n_slots ? f : 0u]);
        mine[u] = f < n_slots ? w : ~0ull;
    }
    unsigned long long ospec[OVF_SPEC > 0 ? OVF_SPEC : 1];
    if (OVF_SPEC > 0) {
#pragma unroll
        for (int u = 0; u < OVF_SPEC; ++u) {
            const uint32_t i = tid + (uint32_t)u * nthreads;
            const unsigned long long w = ld_agent(&P.ovf_cand[i < P.ovf_cap ? i : 0u]);
            ospec[u] = i < P.ovf_cap ? w : 0ull;
        }
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
    if (tid == 0) {
        S.cnt = 0;
        S.total = 0;
        S.sum = 0;
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
        if (OVF_SPEC > 0) {
#pragma unroll
            for (int u = 0; u < OVF_SPEC; ++u)
                c += (tid + (uint32_t)u * nthreads < novf && order_key(__uint_as_float((uint32_t)ospec[u])) >= thr) ? 1u : 0u;
        }
        for (uint32_t i = tid + (uint32_t)OVF_SPEC * nthreads; i < novf; i += nthreads)
            c += (order_key(__uint_as_float((uint32_t)ld_agent(&P.ovf_cand[i]))) >= thr) ? 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
        if (lane == 0 && c) atomicAdd(&S.cnt, c);
    }
    __syncthreads();
    const uint32_t total = n_from_slots + S.cnt;
    const bool small = total <= SEL_CAP;
    __syncthreads();  // everybody has read S.cnt before the general path reuses it
    // An overflow entry as the selection sees it (entry i, or 0: not a candidate): pruned by the threshold, its position translated
    // to a row id where candidates carry positions.
    auto ovf_entry = [&](uint32_t i) __attribute__((always_inline)) -> unsigned long long {
        if (i >= novf) return 0ull;
        // (a plain load: the selecting workgroup took its acquire before it got here and nobody writes the list while it is
        //  being selected from, so the 64 passes of the bisection may come from the L2 -- as round 3's scratch copy did)
        unsigned long long v = P.ovf_cand[i];
        if (order_key(__uint_as_float((uint32_t)v)) < thr) return 0ull;
        if (P.pos_to_row) v = pack_cand((uint32_t)v, P.pos_to_row[(uint32_t)(v >> 32)]);
        return make_ckey(v);
    };
    uint32_t n_sel;
    if (small) {
#pragma unroll
        for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
            if (ok[u]) S.keys[wbase + spos[u]] = make_ckey(mine[u]);
        }
        for (uint32_t i0 = 0; i0 < novf; i0 += nthreads) {  // wave-uniform trip count
            const uint32_t i = i0 + tid;
            unsigned long long v = 0ull;
            if (OVF_SPEC > 0 && i0 < (uint32_t)OVF_SPEC * nthreads) {
#pragma unroll
                for (int u = 0; u < OVF_SPEC; ++u)
                    if (i0 == (uint32_t)u * nthreads) v = ospec[u];
            } else if (i < novf) {
                v = ld_agent(&P.ovf_cand[i]);
            }
            const bool keep = i < novf && order_key(__uint_as_float((uint32_t)v)) >= thr;
            if (keep && P.pos_to_row) v = pack_cand((uint32_t)v, P.pos_to_row[(uint32_t)(v >> 32)]);
            const uint64_t bm = __ballot(keep);
            uint32_t base = 0;
            if (lane == 0 && bm) base = atomicAdd(&S.total, (uint32_t)__popcll(bm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (keep)
                S.keys[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] =
                    make_ckey(v);
        }
        if (tid < 8) S.keys[total + tid] = 0ull;  // padding for the unrolled rank loop (0 is below every real key)
        __syncthreads();
        n_sel = total;
    } else {
        // General path (threshold exchange disabled or not converged, degenerate queries: up to every row is a candidate):
        // bisection for the k-th largest composite key straight over the sources -- this thread's slots (registers) and the
        // overflow list, re-read on every pass: as many bytes as a copy of the keys would cost, and no scratch of 8 bytes per row
        // and selector (round 3 kept one) --, then the keys at or above it into LDS (composite keys are unique: at most k).
        // (the slots' keys are recomputed on every pass rather than kept: 16 more registers would cost the single-query kernels
        //  their second workgroup per CU)
        auto sk = [&](uint32_t u) __attribute__((always_inline)) -> unsigned long long { return ok[u] ? make_ckey(mine[u]) : 0ull; };
        unsigned long long prefix = 0ull;
        if (total > P.k) {
            for (int bit = 63; bit >= 0; --bit) {
                const unsigned long long trial = prefix | (1ull << bit);
                uint32_t c = 0;
#pragma unroll
                for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) c += (sk(u) >= trial);
                for (uint32_t i = tid; i < novf; i += nthreads) c += (ovf_entry(i) >= trial);
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
        auto put = [&](unsigned long long kx) __attribute__((always_inline)) {
            if (kx != 0ull && kx >= prefix) {
                const uint32_t pos = atomicAdd(&S.cnt, 1u);
                if (pos < SEL_CAP) S.keys[pos] = kx;
            }
        };
#pragma unroll
        for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) put(sk(u));
        for (uint32_t i = tid; i < novf; i += nthreads) put(ovf_entry(i));
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
    uint32_t cs = 0u;  // (host-visible results: this thread's share of the block's checksum)
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
            const uint32_t oi = (uint32_t)(kx & 0xFFFFFFFFull) + P.first_row;
            const float ov = key_to_float((uint32_t)(kx >> 32)) * out_scale;
            P.out_idx[r] = oi;
            P.out_val[r] = ov;
            if (P.host_out) {  // system-scope stores: written through to host memory
                __hip_atomic_store(&P.host_out[r], oi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&P.host_out[P.k + r], __float_as_uint(ov), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                cs += result_checksum_term(oi, __float_as_uint(ov), r);
            }
        }
    }
    if (P.host_out) {  // (wave-uniform)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cs += (uint32_t)__shfl_xor((int)cs, d);
        if (lane == 0 && cs) atomicAdd(&S.sum, cs);
    }
    if (stamps && tid == 0) stamps[6] = __builtin_amdgcn_s_memtime();  // ranked
    for (uint32_t r = n_sel + tid; r < P.k; r += nthreads) {
        P.out_idx[r] = 0u;
        P.out_val[r] = 0.0f;
        if (P.host_out) {
            __hip_atomic_store(&P.host_out[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&P.host_out[P.k + r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    auto raise_host_flag = [&]() __attribute__((always_inline)) {
        // every writer drains its stores, the workgroup meets, one thread raises the flag (relaxed: the stores before it are
        // write-through stores already drained by their writers; a release would write back the whole L2 first). The memory
        // model does not order them, so the flag comes with a checksum of the payload that the host verifies (result_checksum_term).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(&P.host_out[2u * P.k + 4u], S.sum + P.host_epoch * 0x9E3779B1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (P.wt_reset)
                __hip_atomic_store(&P.host_out[2u * P.k + 1u], (uint32_t)(__builtin_amdgcn_s_memrealtime() - P.t_seen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else if (P.t_start)
                __hip_atomic_store(&P.host_out[2u * P.k + 1u],
                                   (uint32_t)(__builtin_amdgcn_s_memrealtime() - __hip_atomic_load(P.t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&P.host_out[2u * P.k], P.host_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    };
    if (P.host_out && !P.wt_reset) raise_host_flag();
    // Reset the exchange state for the next query (this is the last consumer of the query on the stream); last, so
    // that no barrier above has to wait for these stores. Slots: only the ones that held a survivor need a store
    // (the stream kernel writes a slot only when it has one).
#pragma unroll
    for (uint32_t u = 0; u < SEL_PER_THREAD; ++u) {
        const uint32_t f = tid + u * nthreads;
        if (f < n_slots && (uint32_t)(mine[u] >> 32) != SLOT_INVALID) {
            if (P.wt_reset) st_agent(&P.wg_cand[f], pack_cand(0u, SLOT_INVALID));
            else P.wg_cand[f] = pack_cand(0u, SLOT_INVALID);
        }
    }
    for (uint32_t i = tid; i < P.n_groups_pub; i += nthreads) {
        if (P.wt_reset) __hip_atomic_store(&P.gmax[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else P.gmax[i] = 0u;
    }
    if (tid == 0) {
        if (P.wt_reset) {
            __hip_atomic_store(P.ovf_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.tau_g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            *P.ovf_count = 0u;
            *P.tau_g = 0u;
        }
        for (uint32_t c = 0; c < 9u; ++c) P.done_count[32u * c] = 0u;
    }
    if (P.host_out && P.wt_reset) raise_host_flag();
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

}  // namespace tkspmv
