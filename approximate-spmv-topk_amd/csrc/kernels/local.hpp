// kernels/local.hpp -- workgroup-local thresholds with a checked selection: the shared pieces (staging a wave's best rows, the
// workgroup's eight ordered slots, select_local) and single_kernel, the one-query launch built on them (tkspmv_run).
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include <cstddef>
#include "packet_math.hpp"

namespace tkspmv {

// ------------------------------------------------------------------------------------------------------------
// Local thresholds in one paragraph (batch_kernel.hpp has the history). A workgroup's threshold comes from its OWN waves,
// through LDS: every streaming wave publishes the best (mode 1) or second best (mode 2) packet maximum it has seen -- scores of
// distinct rows --, the smallest of the waves' words is the workgroup's threshold, and it may START at the score of the 8th best
// row the workgroup delivered for its previous query. None of this proves anything about the matrix as a whole, so the
// selection CHECKS it. Every workgroup delivers a record: its 8 best rows in order (slot 0 = its best) and `used`, an order key
// above everything it dropped (the largest threshold it filtered with; one step above the best row that did not fit a list or a
// slot). The selection takes t0 = the largest `used`: if at least k delivered rows reach t0, the k-th best score does too and
// every dropped row lies strictly below it -- the list is exact; otherwise the query runs again through the exact path
// (device-wide threshold exchange, overflow lists). Nothing is appended to global memory in this mode: no overflow list.
// ------------------------------------------------------------------------------------------------------------
struct LocalParams {
    unsigned long long *slots;  // [n_wg][WG_SLOTS] {score bits | row << 32}, row SLOT_INVALID = empty; all 8 written by every workgroup
    uint32_t *used;             // [n_wg] order key above everything the workgroup dropped (0: it dropped nothing)
    float *wg_prior;            // [n_wg] what the workgroup's next query starts from, relative to sum |x| (reported-score units per unit of L1 norm; 0: nothing). NULL: no carrying
    uint32_t *prior_block;      // [0] selections to go without carried thresholds, [1] what a failure costs, [2] clean run, [3] failures so far
    uint32_t *status;           // device word: 1 = the check of this launch failed (a repair launch behind it reads it), else 0
    uint32_t mode;              // 1: a wave's word is its best packet maximum, 2: its second best
    float beta;                 // carried thresholds start at beta x the recorded score
    unsigned long long *trace;  // optional (TKSPMV_TRACE=1): [grid][8 waves][8] s_memrealtime stamps
    uint32_t shared_bookkeeping;  // 1: several selections run at once (batch kernel): prior_block is updated with atomics
    // single_kernel (round 5): workgroup 0 of the launch is the SELECTOR; a streaming workgroup that has stored its record and
    // drained the stores raises ready[its number] to the launch's epoch and leaves -- no ticket, no atomic round trip behind its
    // stream --, and the selector, which has polled the flags all along, loads each record as its flag comes up: when the last
    // workgroup is through, one record is left to load, not 4096 slots.
    uint32_t *ready;  // [n_wg]
    uint32_t epoch;   // != 0, different from launch to launch
};

constexpr uint32_t STG_N = 8;         // survivors a wave stages per query (64 lanes = 8 waves x 8 when one wave finalises)
constexpr uint32_t LSEL_CAP = 4096;   // 512 workgroups x 8 slots
constexpr uint32_t LSEL_RANK = 1024;  // candidates ranked by counting; more are first cut down to the k best exactly (bisection)
constexpr uint32_t LSEL_BUCKETS = 1024;
struct LocalSelectShared {
    unsigned long long keys[LSEL_CAP + 8];
    uint32_t hist[LSEL_BUCKETS];  // candidates per bucket of order keys
    uint32_t cnt, total, thr, t0, last, sum;
};

// Order key of the n-th largest (n >= 1) of the keys held EPL per lane (0 = no key), exact: 32-step bisection with ballots.
template <int EPL>
__device__ __forceinline__ uint32_t nth_largest_key(const uint32_t (&kk)[EPL], uint32_t n) {
    uint32_t prefix = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t trial = prefix | (1u << bit);
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < EPL; ++i) c += (uint32_t)__popcll(__ballot(kk[i] >= trial));
        if (c >= n) prefix = trial;
    }
    return prefix;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto mv = [](uint32_t a, uint32_t b) { return b > a ? b : a; };
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR1, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR2, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR4, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_SHR8, 0xF, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, false));
    v = mv(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// (fp32 sum of a wave in a FIXED order -- the DPP tree -- so that a value derived from it is the same in every run)
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_SHR1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_SHR2, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_SHR4, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_SHR8, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_BCAST15, 0xA, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_BCAST31, 0xC, 0xF, true));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {  // (DPP prefix sum; the total is read from lane 63)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR1, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR2, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR4, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR8, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, true);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// End of a wave's partition: what still clears the threshold is staged in LDS for the workgroup's finalisation -- at most STG_N
// rows; if more survive, the rest is dropped and goes on record (MISC_BOUND). BEST: the wave keeps its STG_N BEST rows (an exact
// bisection over the survivors' keys; single_kernel has the registers for it); else the first STG_N in list order, and the largest
// dropped score is what goes on record -- a recorded bound is a bound either way, a loose one just fails the check sooner.
// Lane 0 writes the count.
template <uint32_t WAVE_CAP, bool BEST = true>
__device__ __forceinline__ void stage_wave_local(const uint2 *wcand, uint32_t wcnt, float tau, uint32_t lane, unsigned long long *stg,
                                                 uint32_t *stg_cnt, uint32_t *misc) {
    constexpr uint32_t EPL = WAVE_CAP / 64u;
    ListScan<EPL> LS;
    uint32_t surv = scan_list<EPL>(wcand, wcnt, tau, lane, LS);
    if (surv > STG_N) {  // rare
        if (BEST) {
            // keep the STG_N best (ties: the first in list order), drop the rest under a recorded bound
            uint32_t kk[EPL];
#pragma unroll
            for (uint32_t u = 0; u < EPL; ++u) kk[u] = LS.keep[u] ? order_key(__uint_as_float(LS.e[u].x)) : 0u;
            const uint32_t t = nth_largest_key<(int)EPL>(kk, STG_N);
            uint32_t n_gt = 0;
#pragma unroll
            for (uint32_t u = 0; u < EPL; ++u) n_gt += (uint32_t)__popcll(__ballot(kk[u] > t));
            uint32_t eq_seen = 0, total = 0;
            bool dropped_eq = false;
#pragma unroll
            for (uint32_t u = 0; u < EPL; ++u) {
                const bool eq = LS.keep[u] && kk[u] == t;
                const uint64_t be = __ballot(eq);
                const uint32_t my_eq = eq_seen + __builtin_amdgcn_mbcnt_hi((uint32_t)(be >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)be, 0u));
                eq_seen += (uint32_t)__popcll(be);
                const bool sel = LS.keep[u] && (kk[u] > t || (eq && n_gt + my_eq < STG_N));
                dropped_eq = dropped_eq || (eq && !sel);
                LS.keep[u] = sel;
                const uint64_t bs = __ballot(sel);
                LS.pos[u] = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(bs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bs, 0u));
                total += (uint32_t)__popcll(bs);
            }
            // everything dropped has a key <= t (== t only where a tie was dropped): on record strictly above it
            const uint32_t bound = __ballot(dropped_eq) != 0ull ? t + 1u : t;
            if (lane == 0) (void)__hip_atomic_fetch_max(&misc[MISC_BOUND], bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            float dmax = -__builtin_huge_valf();
#pragma unroll
            for (uint32_t u = 0; u < EPL; ++u) {
                const bool dropped = LS.keep[u] && LS.pos[u] >= STG_N;
                dmax = dropped ? fmaxf(dmax, __uint_as_float(LS.e[u].x)) : dmax;
                LS.keep[u] = LS.keep[u] && !dropped;
            }
            dmax = wave_max(dmax);
            if (lane == 0) (void)__hip_atomic_fetch_max(&misc[MISC_BOUND], order_key(dmax) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        surv = STG_N;
    }
#pragma unroll
    for (uint32_t u = 0; u < EPL; ++u)
        if (LS.keep[u]) stg[LS.pos[u]] = pack_cand(LS.e[u].x, LS.e[u].y);
    if (lane == 0) *stg_cnt = surv;
}

// The bookkeeping of carried thresholds after a selection's check (thread 0 of the selecting workgroup): a failed check suspends
// them for the next 16 .. 4096 selections (word 0 counts down; word 1: what a failure costs -- doubled by every failure, halved by
// every 64 checks passed in a row with them, word 2); word 3 counts the failures (tkspmv_debug_counters).
__device__ __forceinline__ void prior_block_update(uint32_t *pb, bool bad, bool had_threshold) {
    if (bad) {
        (void)__hip_atomic_fetch_add(pb + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t len = __hip_atomic_load(pb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t nl = len < 16u ? 16u : (len >= 2048u ? 4096u : 2u * len);
        __hip_atomic_store(pb + 1, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pb + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pb, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (__hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        (void)__hip_atomic_fetch_sub(pb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (had_threshold) {
        const uint32_t ok_run = __hip_atomic_fetch_add(pb + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if ((ok_run & 63u) == 0u) {
            const uint32_t len = __hip_atomic_load(pb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (len > 16u) __hip_atomic_store(pb + 1, len / 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The selection of a query streamed with local thresholds: one workgroup, one trip through memory for all records. t0 = the
// largest `used` of the records. The check is a count: if at least k delivered rows reach t0, the k-th best score does too -- every
// row of the result reaches t0 and everything that was dropped lay strictly below it; if fewer do, the check has failed and the
// list does not matter (the query runs again through the exact path). Typically a quarter of the 4096 delivered rows reach t0, so
// the rows to rank are cut down with a histogram: 1024 buckets of order keys upwards from t0, a sum from the top by one wave, and
// everything from the bucket that holds the k-th largest upwards is ranked by counting (k plus a handful). Returns (every thread)
// whether the check failed.
// SP.host_out != NULL: the k results also go to host-visible memory with the epoch flag, the launch's own duration, the status
// of the check and the checksum.
// pre / pre_used (single_kernel's selector): the calling thread already holds eight slots and the largest `used` of the records it
// looked at -- nothing is loaded here.
__device__ __forceinline__ bool select_local(const LocalParams &G, const SelectParams &SP, const uint32_t n_wg, const uint32_t tid,
                                             const uint32_t nthreads, LocalSelectShared &S, const float out_scale,
                                             unsigned long long *stamps = nullptr, const unsigned long long *pre = nullptr,
                                             const uint32_t pre_used = 0u) {
    const uint32_t lane = tid & 63u;
    const bool HOST = SP.host_out != nullptr;
    const uint32_t n_slots = n_wg * WG_SLOTS;  // (host: n_wg <= 512, n_slots <= 8 * nthreads)
    unsigned long long mine[8];
    uint32_t my_used = pre_used;
    if (pre) {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) mine[u] = pre[u];
    } else {
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t f = tid + u * nthreads;
            mine[u] = f < n_slots ? ld_agent(&G.slots[f]) : ~0ull;  // (eight loads in flight as the compiler emits it: checked on the ISA, tests/test_kernel_resources.py)
        }
        for (uint32_t w = tid; w < n_wg; w += nthreads) {
            const uint32_t uw = __hip_atomic_load(&G.used[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            my_used = uw > my_used ? uw : my_used;
        }
    }
    // (the bookkeeping thread reads the suspension counters with the records: single writer -- one selection at a time --, so
    //  what follows the check is stores only, no trip through memory behind it)
    unsigned long long t_start = 0ull;  // (thread 0: the launch's start stamp, for the duration it reports with the flag)
    if (tid == 0 && HOST && SP.t_start) t_start = __hip_atomic_load(SP.t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t pbv[4] = {0u, 0u, 0u, 0u};
    if (tid == nthreads - 1u && G.prior_block && !G.shared_bookkeeping) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pbv[i] = __hip_atomic_load(G.prior_block + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        S.cnt = 0;
        S.total = 0;
        S.t0 = 0;
        S.sum = 0;
    }
    for (uint32_t i = tid; i < LSEL_BUCKETS; i += nthreads) S.hist[i] = 0u;
    // (zeroed while the loads are in flight: the rank loop reads whole blocks of 8 keys, and 0 is below every real key)
    for (uint32_t i = tid; i < LSEL_RANK + 8u; i += nthreads) S.keys[i] = 0ull;
    __syncthreads();
    {
        const uint32_t mu = wave_max_u32(my_used);  // (waits for the loads of `used`, issued behind the records')
        if (lane == 0 && mu != 0u) atomicMax(&S.t0, mu);
    }
    unsigned long long key[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) key[u] = (uint32_t)(mine[u] >> 32) != SLOT_INVALID ? make_ckey(mine[u]) : 0ull;
    // (the loads of the bookkeeping words and of the start stamp are consumed HERE, where the records' loads have been waited for
    //  anyway: a first use behind the result stores would wait for those stores -- the counter retires in order)
    asm volatile("" ::"v"(pbv[0]), "v"(pbv[1]), "v"(pbv[2]), "v"(pbv[3]), "v"((uint32_t)t_start), "v"((uint32_t)(t_start >> 32)));
    __syncthreads();
    if (stamps && tid == 0) stamps[0] = __builtin_amdgcn_s_memrealtime();  // the loads have returned
    const uint32_t t0 = S.t0;
    // The candidates: delivered rows that reach t0. Their bucket: (key - t0) >> 13 -- the order keys of one binade above t0 in
    // 1024 steps, everything from 2 x t0 upwards in the last bucket (monotone in the key, so "bucket >= b" is an upper set in score
    // order). No threshold on record at all (t0 = 0; rare): the positive floats' whole range in 1024 steps, coarse but bounded.
    const uint32_t lo = t0 != 0u ? t0 : 0x80000000u;
    const uint32_t shift = t0 != 0u ? 13u : 21u;
    bool in[8];
    uint32_t bkt[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
        const uint32_t kh = (uint32_t)(key[u] >> 32);
        in[u] = key[u] != 0ull && kh >= t0;
        bkt[u] = 0u;
        if (in[u]) {
            const uint32_t b = kh >= lo ? (kh - lo) >> shift : 0u;
            bkt[u] = b < LSEL_BUCKETS ? b : LSEL_BUCKETS - 1u;
            atomicAdd(&S.hist[bkt[u]], 1u);
        }
    }
    __syncthreads();
    if (tid < 64u) {
        // Lane l sums buckets [16 (63 - l), 16 (63 - l) + 16) -- the TOP buckets in lane 0 --, so that an inclusive prefix sum over
        // the lanes (DPP) counts the candidates from the top; the first lane whose sum reaches k holds the bucket of the k-th
        // largest, and every lane walks its own 16 registers for it (only that lane's answer counts).
        constexpr uint32_t PER = LSEL_BUCKETS / 64u;
        const uint32_t b0 = (63u - lane) * PER;
        uint32_t h[PER];
        uint32_t sl = 0u;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            h[j] = S.hist[b0 + j];
            sl += h[j];
        }
        uint32_t v = sl;
        {
            auto add_dpp = [](uint32_t a, uint32_t b) { return a + b; };
            v = add_dpp(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR1, 0xF, 0xF, true));
            v = add_dpp(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR2, 0xF, 0xF, true));
            v = add_dpp(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR4, 0xF, 0xF, true));
            v = add_dpp(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_SHR8, 0xF, 0xF, true));
            v = add_dpp(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST15, 0xA, 0xF, true));
            v = add_dpp(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_BCAST31, 0xC, 0xF, true));
        }
        const uint64_t reach = __ballot(v >= SP.k);
        if (lane == 63u) S.cnt = v;  // candidates in all (the check)
        if (reach == 0ull) {
            if (lane == 0) S.thr = 0u;  // fewer than k candidates: all of them
        } else {
            uint32_t acc = v - sl, bc = b0;
            bool found = false;
#pragma unroll
            for (int j = (int)PER - 1; j >= 0; --j) {
                acc += h[j];
                if (!found && acc >= SP.k) {
                    bc = b0 + (uint32_t)j;
                    found = true;
                }
            }
            if (lane == (uint32_t)__builtin_ctzll(reach)) S.thr = bc;
        }
    }
    __syncthreads();
    const uint32_t bcut = S.thr;
    // the check: do at least k of the delivered rows reach everything that was recorded?
    const bool bad = t0 != 0u && S.cnt < SP.k;
    if (stamps && tid == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();  // the cut is known
    bool ok[8];
    uint32_t spos[8];
    uint32_t wtot = 0;
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
        ok[u] = in[u] && bkt[u] >= bcut;
        const uint64_t bm = __ballot(ok[u]);
        spos[u] = wtot + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
        wtot += (uint32_t)__popcll(bm);
    }
    uint32_t wbase = 0;
    if (lane == 0 && wtot) wbase = atomicAdd(&S.total, wtot);
    wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u)
        if (ok[u]) S.keys[wbase + spos[u]] = key[u];
    __syncthreads();
    uint32_t n_sel = S.total;
    if (!bad && n_sel > LSEL_RANK) {
        // Rare (a bucket full of equal scores): the k best exactly -- bisection for the k-th largest composite key over the keys
        // in LDS, then the keys at or above it (keys are unique: exactly k).
        unsigned long long prefix = 0ull;
        for (int bit = 63; bit >= 0; --bit) {
            const unsigned long long trial = prefix | (1ull << bit);
            uint32_t c = 0;
            for (uint32_t i = tid; i < n_sel; i += nthreads) c += (S.keys[i] >= trial);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
            if (tid == 0) S.cnt = 0;
            __syncthreads();
            if (lane == 0 && c) atomicAdd(&S.cnt, c);
            __syncthreads();
            if (S.cnt >= SP.k) prefix = trial;
            __syncthreads();
        }
        unsigned long long kv[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t i = tid + u * nthreads;
            kv[u] = i < n_sel ? S.keys[i] : 0ull;
        }
        if (tid == 0) S.total = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u)
            if (kv[u] != 0ull && kv[u] >= prefix) S.keys[atomicAdd(&S.total, 1u)] = kv[u];
        __syncthreads();
        n_sel = S.total;
        if (tid < 8) S.keys[n_sel + tid] = 0ull;  // padding for the unrolled rank loop
        __syncthreads();
    }
    if (bad) n_sel = 0u;  // (nothing is ranked: the list of a failed check is never read)
    if (stamps && tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();  // keys in LDS
    // Rank by counting (keys are unique: rank r = number of larger keys); GT threads share one key and meet through shuffles.
    uint32_t cs = 0u;
    {
        uint32_t GT = 1;
        while (GT < 8u && n_sel * (GT * 2u) <= nthreads) GT *= 2u;
        const uint32_t n_blocks = ((n_sel + 7u) & ~7u) >> 3;
        for (uint32_t base = 0; base < n_sel; base += nthreads / GT) {
            const uint32_t i = base + tid / GT, part = tid & (GT - 1u);
            const bool active = i < n_sel;
            const unsigned long long kx = active ? S.keys[i] : ~0ull;
            uint32_t r = 0;
            for (uint32_t blk = part; blk < n_blocks; blk += GT) {
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) r += (S.keys[blk * 8u + u] > kx);
            }
            for (uint32_t d = 1; d < GT; d <<= 1) r += (uint32_t)__shfl_xor((int)r, (int)d);
            if (active && part == 0u && r < SP.k) {
                const uint32_t oi = (uint32_t)(kx & 0xFFFFFFFFull) + SP.first_row;
                const float ov = key_to_float((uint32_t)(kx >> 32)) * out_scale;
                SP.out_idx[r] = oi;
                SP.out_val[r] = ov;
                if (HOST) {
                    __hip_atomic_store(&SP.host_out[r], oi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(&SP.host_out[SP.k + r], __float_as_uint(ov), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    cs += result_checksum_term(oi, __float_as_uint(ov), r);
                }
            }
        }
    }
    // (a failed check writes NO list: the repair that follows writes the same buffers, possibly from another XCD, and plain stores
    //  of two workgroups of one launch reach memory in no particular order -- zeros landing last is what a test caught)
    for (uint32_t r = n_sel + tid; r < SP.k && !bad; r += nthreads) {
        SP.out_idx[r] = 0u;
        SP.out_val[r] = 0.0f;
        if (HOST) {
            __hip_atomic_store(&SP.host_out[r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&SP.host_out[SP.k + r], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();  // ranked, stores issued
    if (tid == nthreads - 1u) {  // the bookkeeping of the check (prior_block_update's rules, as plain stores of this single writer)
        if (G.status) __hip_atomic_store(G.status, bad ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (G.prior_block && G.shared_bookkeeping) {
            prior_block_update(G.prior_block, bad, t0 != 0u);
        } else if (G.prior_block) {
            uint32_t *pb = G.prior_block;
            if (bad) {
                const uint32_t nl = pbv[1] < 16u ? 16u : (pbv[1] >= 2048u ? 4096u : 2u * pbv[1]);
                __hip_atomic_store(pb + 3, pbv[3] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pb + 1, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pb + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(pb, nl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (pbv[0] != 0u) {
                __hip_atomic_store(pb, pbv[0] - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (t0 != 0u) {
                __hip_atomic_store(pb + 2, pbv[2] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((pbv[2] + 1u) & 63u) == 0u && pbv[1] > 16u) __hip_atomic_store(pb + 1, pbv[1] / 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (HOST) {
        // The flag does not wait for the payload's stores to drain: the memory model would not order them anyway, and the host trusts
        // the block only once it adds up to the checksum (which also covers the status of the check) -- result_block_complete.
        cs = wave_sum_u32(cs);
        if (lane == 0 && cs) atomicAdd(&S.sum, cs);
        __syncthreads();
        if (stamps && tid == 0) stamps[4] = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            uint32_t *tail = SP.host_out + 2u * SP.k;
            if (SP.t_start) __hip_atomic_store(&tail[1], (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_start), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&tail[5], bad ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&tail[4], S.sum + SP.host_epoch * 0x9E3779B1u + (bad ? 0xBADC0DE5u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&tail[0], SP.host_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    return bad;
}

// ------------------------------------------------------------------------------------------------------------
// single_kernel: ONE query per launch, the reference loop's operator() (host_spmv_bscsr.cpp:323-397) -- built for latency.
// Measured on this GPU (round 4): a launch that only loads the 1M-row stream takes 17.5 us (15.6 per pass in steady state), with
// the per-packet arithmetic 19 us; round 3's fused launch took 32.7, its median wave finishing at 21.8 us and its last at 25.5 --
// the device-wide threshold exchange (every server wave polling one hot word, a cold phase in which every row is a candidate,
// two dependent table loads before the first packet request) cost the stream 6 us, and the tail another 8. This kernel streams with
// workgroup-local thresholds carried from the previous query (no exchange traffic, no cold phase, no server wave: 8 waves per
// workgroup, two workgroups per CU), derives its partition from its number, requests x ahead of the packets, and ends with
// select_local in the workgroup that draws the last ticket. A failed check is reported (status word, host-visible flag) and the
// query is run again through stream_kernel, whose result is exact on its own.
// ------------------------------------------------------------------------------------------------------------
struct SingleLds {
    union {
        struct {
            float x[1024];                                // at LDS offset 0: (column word & 0xFFC) | xbase is the address of x[col]
            uint2 cand[ListGeom<1024>::CAND_CAP];         // private candidate lists of the 8 waves
        } w;
        LocalSelectShared sel;  // the selecting workgroup only, after its stream
    } u;
    uint32_t misc[MISC_WORDS];
    unsigned long long stg[8 * STG_N];
    uint32_t stg_cnt[8];
    uint32_t rank[64];  // ranks of the staged rows (the workgroup's record)
    float xnorm[8];     // per wave: sum |x| over the words it loaded (carried thresholds are kept relative to the query's L1 norm)
};

constexpr uint32_t SINGLE_REC_STRIDE = 16;  // 64-bit words between the records of consecutive workgroups when workgroup 0 selects (LocalParams::ready)
template <int QM>
__global__ void __launch_bounds__(512, 4) single_kernel(const StreamParams P, const SelectParams SP, const LocalParams G) {
    static_assert(QM == 0 || QM == 7, "single_kernel: fp32 values, 4 entries per lane, at most 1024 columns");
    constexpr int C = 4, NBUF = 3;
    constexpr int VT = value_type_of(QM);
    constexpr uint32_t WAVE_CAP = ListGeom<1024>::WAVE_CAP;
    __shared__ SingleLds L;
    static_assert(offsetof(SingleLds, u) == 0 && offsetof(decltype(SingleLds::u), w) == 0 && offsetof(decltype(SingleLds::u.w), x) == 0, "x must be the first member of the kernel's only LDS block (reduce_packet's address trick)");
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // G.ready != NULL (round 5): workgroup 0 is the launch's selector, workgroups 1 .. grid - 1 stream (LocalParams::ready)
    const bool dedicated = G.ready != nullptr;
    const uint32_t bid = dedicated ? blockIdx.x - 1u : blockIdx.x, n_wg = dedicated ? gridDim.x - 1u : gridDim.x;
    unsigned long long *tr = G.trace ? G.trace + ((size_t)blockIdx.x * 8u + wave) * 8u : nullptr;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0;
    if (tr) tr0 = __builtin_amdgcn_s_memrealtime();
    if (SP.t_start && blockIdx.x == 0u && tid == 0u)  // (workgroup 0 is dispatched first: the launch's start within a microsecond)
        __hip_atomic_store(SP.t_start, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dedicated && blockIdx.x == 0u) {
        // ---- the selector: thread t waits for the record of streaming workgroup t (its flag at this launch's epoch: the record's
        // stores were drained before the flag was raised) and loads it at once -- 64 contiguous bytes it has never touched in this
        // launch, so no cache of this XCD can hold an older copy. Flags are polled with an atomic (executed at the device-wide
        // coherence point; a load, even agent-scope, was seen to return a stale line for milliseconds: batch_kernel.hpp). Bounded:
        // 20 ms without the last flag and the launch reports a failed check (the host repeats the query through the exact kernel).
        unsigned long long rec[8];
        uint32_t my_used = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) rec[u] = ~0ull;
        bool have = tid >= n_wg;
        const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
        bool timed_out = false;
        for (;;) {
            if (!have) {
                const uint32_t f = __hip_atomic_fetch_or(&G.ready[tid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f == G.epoch) {
                    // (a record has a 128-byte line of its own -- 8 slots, then `used` --: no other thread's load can have brought
                    //  it into a cache before its flag was up)
                    const unsigned long long *src = G.slots + (size_t)tid * SINGLE_REC_STRIDE;
#pragma unroll
                    for (int u = 0; u < 8; ++u) rec[u] = ld_agent(src + u);
                    my_used = __hip_atomic_load(reinterpret_cast<const uint32_t *>(src + WG_SLOTS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    have = true;
                }
            }
            if (__ballot(!have) == 0ull) break;
            if (__builtin_amdgcn_s_memrealtime() - t_begin > 2000000ull) {
                timed_out = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        unsigned long long *sel_stamps = G.trace ? G.trace + (size_t)gridDim.x * 64u : nullptr;  // (behind the per-wave rows)
        if (sel_stamps && tid == 0) sel_stamps[7] = __builtin_amdgcn_s_memrealtime();  // this wave holds its 64 records
        if (timed_out) my_used = 0xFFFFFFFFu;  // (above every key: fewer than k rows reach it -- the check fails, nothing is written)
        (void)select_local(G, SP, n_wg, tid, blockDim.x, L.u.sel, 1.0f, sel_stamps, rec, my_used);
        if (sel_stamps && tid == 0) sel_stamps[5] = __builtin_amdgcn_s_memrealtime();
        return;
    }

    // x first (two words per thread: they return ahead of the packets), then this wave's first packets
    // (clamped addresses, masked values: two loads in flight, no branch around either)
    const float xa = P.x[tid < P.cols ? tid : 0u], xb = P.x[tid + 512u < P.cols ? tid + 512u : 0u];
    const float x0 = tid < P.cols ? xa : 0.0f;
    const float x1 = tid + 512u < P.cols ? xb : 0.0f;
    const uint32_t q = wave * n_wg + bid;
    uint32_t p0 = 0, np = 0;
    if (q < P.n_parts) TKSPMV_PARTITION_RANGE(P, q, p0, np);
    // (round 5, as in the batch kernel: buffer loads -- resource = this wave's partition, scalar byte offset per packet, the lane's
    //  share as a constant vector offset: no vector address arithmetic per packet --, and the row base of a packet looked up on the
    //  candidate path only)
    Pkt<C, VT> buf[NBUF];
    const uint8_t *pk = P.packets + (size_t)p0 * P.packet_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = stream_resource(pk, np * P.packet_bytes);
    const LaneOffsets lo = lane_offsets<C, VT>(lane);
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) {
        if (np > 0) {
            const uint32_t iu = ((uint32_t)u < np) ? (uint32_t)u : (np - 1);
            load_packet_buf<C, VT>(rsrc, iu * P.packet_bytes, lo, buf[u]);
        }
    }
    uint32_t req_off = (np > (uint32_t)(NBUF - 1) ? (uint32_t)(NBUF - 1) : (np > 0u ? np - 1u : 0u)) * P.packet_bytes;  // the next request's packet
    // which waves of this workgroup stream (lane w < 8 looks at wave w): the workgroup's threshold is the smallest of THEIR words
    bool wave_has = false;
    {
        const uint32_t pw = lane * n_wg + bid;
        wave_has = lane < 8u && pw < P.n_parts && (P.uni_ppp != 0u || P.part_count[pw] != 0u);
    }
    const float min_units = P.min_score;  // (fp32 scores: one unit = 1.0)
    // The carried threshold of this workgroup (unless a failed check has suspended carrying): what it delivered as its 8th best
    // row for the previous query, kept RELATIVE to that query's L1 norm -- scores are linear in x, so a query three times as
    // large, or a hundred times smaller, starts from a threshold that fits it (round 4; before, every change of scale was a failed
    // check or a useless threshold). The loads of the prior are issued here, ahead of x; the threshold itself needs sum |x|.
    float prior = 0.0f;
    uint32_t blocked = 1u;
    if (G.wg_prior) {
        prior = __uint_as_float(scalar_load(reinterpret_cast<const uint32_t *>(G.wg_prior) + bid));
        blocked = scalar_load(G.prior_block);
    }
    if (tid < (uint32_t)MISC_WORDS) L.misc[tid] = tid == (uint32_t)MISC_TAU ? __float_as_uint(min_units) : 0u;
    if (tid < 8u) L.stg_cnt[tid] = 0u;
    if (tid < 64u) L.rank[tid] = 0u;
    L.u.w.x[tid] = x0;
    L.u.w.x[tid + 512u] = x1;
    {
        const float wsum = wave_sum_f32(fabsf(x0) + fabsf(x1));
        if (lane == 0) L.xnorm[wave] = wsum;
    }
    __syncthreads();
    // (every wave adds the eight partial sums in the same order: the same value everywhere, in every run)
    float xnorm = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) xnorm += L.xnorm[w];
    if (blocked == 0u && prior > 0.0f) {
        // Every wave installs the (same) value before its own first packet -- its LDS operations execute in order, so it streams
        // against it from the start; the rule is the one thresholds are raised by further down (record the largest in MISC_TAUKEY,
        // store only what raises it), so a wave that starts late never lowers what faster waves have already formed.
        const float t0 = prior * xnorm * G.beta;
        if (t0 > min_units && lane == 0) {
            const uint32_t k0 = order_key(t0);
            const uint32_t old = __hip_atomic_fetch_max(&L.misc[MISC_TAUKEY], k0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k0 > old) __hip_atomic_store(&L.misc[MISC_TAU], __float_as_uint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (tr) tr1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t xbase = lds_addr_of(L.u.w.x);
    uint32_t *misc = L.misc;
    uint2 *wcand = L.u.w.cand + wave * WAVE_CAP;

    if (np != 0u) {
        // The workgroup that arrived SECOND on its CU (the dispatcher places workgroups 0 .. n/2-1 first, one per CU) loses the
        // arbitration to the older one and finished ~1.8 us later at equal priorities (traced, tools/single_trace.py): it streams at
        // the higher issue priority. 26.4 against 27.0 us per launch (the kernel's own stamp, p95 27.2 against 28.3; three alternating
        // runs on one box); priority 3 instead of 2, or a further step for the younger waves of a SIMD, gave the same. (Round 2's
        // alternating turns -- the batch kernel's way -- had lost on the single-query kernel: one query has no turn to alternate.)
        if (bid >= n_wg / 2u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
        float carry = 0.0f;
        uint32_t wcnt = 0u;
        float top1 = -__builtin_huge_valf(), top2 = -__builtin_huge_valf();
        const bool top1_mode = G.mode == 1u;
        const bool rows_in_lanes = np <= 64u;
        const uint32_t row_base_v = (rows_in_lanes && lane < np) ? P.pkt_row[p0 + lane] : 0u;
        for (uint32_t i0 = 0; i0 < np; i0 += NBUF) {
#pragma unroll
            for (int u = 0; u < NBUF; ++u) {
                const uint32_t i = i0 + (uint32_t)u;
                if (i >= np) break;
                const Pkt<C, VT> &cur = buf[u];
                {
                    // Unconditional (offset clamped to the last packet): a fixed number of younger loads lets the compiler wait
                    // with a counted vmcnt instead of vmcnt(0).
                    load_packet_buf<C, VT>(rsrc, req_off, lo, buf[(u + NBUF - 1) % NBUF]);
                    if (i + (uint32_t)NBUF < np) req_off += P.packet_bytes;
                }
                const float tau = __uint_as_float(__hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const Reduced<C> Rd = reduce_packet<C, QM>(cur, carry, xbase, 0u);
                if (tr && i == 0u) tr2 = __builtin_amdgcn_s_memrealtime() + (__float_as_uint(Rd.S) & 0u);
                if (__any(trigger_of<C, false>(Rd) >= tau)) {
                    const RowSums<C> R = expand<C, false>(Rd, packet_flags<C, QM>(cur));
                    // (row of the first row end of this packet: lane j has held packet j's since the wave's first loads, batch_kernel.hpp)
                    const uint32_t rb_cur = rows_in_lanes ? (uint32_t)__builtin_amdgcn_readlane((int)row_base_v, (int)i) : scalar_load(P.pkt_row + p0 + i);
                    const float wm = offer_candidates<C, QM, WAVE_CAP, false>(P, R, rb_cur, tau, lane, 0u, false, wcand, wcnt, misc, true);
                    if (wm > top2 && wm >= min_units) {
                        // (a wave with a single packet has no second maximum: it stands for one row)
                        top2 = wm > top1 ? top1 : wm;
                        top1 = wm > top1 ? wm : top1;
                        const float pub = (np >= 2u && !top1_mode) ? top2 : top1;
                        if (pub >= min_units) {
                            if (lane == 0)  // single writer: this wave's word
                                __hip_atomic_store(&misc[MISC_GRPMAX + wave], order_key(pub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            // The workgroup's threshold: the smallest word, once every streaming wave has one (this wave's LDS
                            // operations execute in order). Whoever raises it records it (MISC_TAUKEY, an atomic max: the largest
                            // threshold ever formed) and passes it on; a smaller value landing last in MISC_TAU is still a threshold.
                            const uint32_t key = wave_has ? __hip_atomic_load(&misc[MISC_GRPMAX + (lane & 7u)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xFFFFFFFFu;
                            if (__ballot(key == 0u) == 0ull) {
                                const uint32_t kmin = wave_min_u32(key);
                                if (lane == 0) {
                                    const uint32_t old = __hip_atomic_fetch_max(&misc[MISC_TAUKEY], kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (kmin > old)
                                        __hip_atomic_store(&misc[MISC_TAU], __float_as_uint(key_to_float(kmin)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (tr) tr3 = __builtin_amdgcn_s_memrealtime();
        if (wcnt != 0u) {
            const float tau3 = __uint_as_float(__hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            stage_wave_local<WAVE_CAP>(wcand, wcnt, tau3, lane, L.stg + wave * STG_N, &L.stg_cnt[wave], misc);
        }
    }
    __syncthreads();
    // ---- the workgroup's record: its 8 best staged rows IN ORDER. Rank by counting with every wave: thread t looks at entry
    // e = t % 64 (slot e % 8 of wave e / 8) and counts the larger ones among wave (t / 64)'s staged rows; composite keys (score,
    // row) are unique, so the ranks are 0 .. n - 1. (One wave picking maximum after maximum took 1.6 us of every launch's tail.)
    {
        const uint32_t e = tid & 63u, part = tid >> 6;
        const bool valid = (e & 7u) < L.stg_cnt[e >> 3];
        const unsigned long long mk = valid ? make_ckey(L.stg[e]) : 0ull;
        const uint32_t cnt_p = L.stg_cnt[part];
        uint32_t r = 0;
#pragma unroll
        for (uint32_t jj = 0; jj < STG_N; ++jj) r += (jj < cnt_p && make_ckey(L.stg[part * STG_N + jj]) > mk) ? 1u : 0u;
        if (valid && r != 0u) atomicAdd(&L.rank[e], r);
    }
    __syncthreads();
    unsigned long long tr5 = 0, tr6 = 0, tr7 = 0;
    if (wave == 0u) {
        if (tr) tr5 = __builtin_amdgcn_s_memrealtime();  // every wave of the workgroup has staged, the ranks are known
        const bool valid = (lane & 7u) < L.stg_cnt[lane >> 3];
        const uint32_t r = valid ? L.rank[lane] : 0xFFFFu;
        const unsigned long long v = L.stg[lane];
        const uint32_t n_valid = (uint32_t)__popcll(__ballot(valid));
        unsigned long long *my_slots = G.slots + (size_t)bid * (dedicated ? SINGLE_REC_STRIDE : WG_SLOTS);
        if (valid && r < WG_SLOTS) st_agent(my_slots + r, v);
        if (lane < WG_SLOTS && lane >= n_valid) st_agent(my_slots + lane, pack_cand(0u, SLOT_INVALID));
        // what did not fit the 8 slots goes on record one step up (a dropped row may tie with the best of them)
        uint32_t used = 0u;
        const uint64_t b8 = __ballot(valid && r == WG_SLOTS);
        if (b8 != 0ull) used = order_key(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)__builtin_ctzll(b8)))) + 1u;
        {
            const uint32_t k_thr = __hip_atomic_load(&misc[MISC_TAUKEY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t k_drop = __hip_atomic_load(&misc[MISC_BOUND], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t km = __builtin_amdgcn_readfirstlane(k_thr > k_drop ? k_thr : k_drop);
            used = used > km ? used : km;
        }
        // the next query of this workgroup starts from the score of its 8th best row; with fewer rows, from the threshold in force
        // (it let fewer than 8 rows through: high enough), a little lower
        float next_prior = -1.0f;
        const uint64_t b7 = __ballot(valid && r == WG_SLOTS - 1u);
        if (b7 != 0ull) next_prior = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)__builtin_ctzll(b7)));
        else {
            const float t_end = __uint_as_float(__hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (t_end > min_units) next_prior = t_end * 0.95f;
        }
        if (tr) tr6 = __builtin_amdgcn_s_memrealtime();
        if (lane == 8u) __hip_atomic_store(dedicated ? reinterpret_cast<uint32_t *>(my_slots + WG_SLOTS) : &G.used[bid], used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 9u && G.wg_prior && next_prior >= 0.0f && xnorm > 0.0f) G.wg_prior[bid] = next_prior / xnorm;  // (relative to sum |x|; read by the NEXT launch)
        // Hand-off (cdna_hip_programming.md Guideline 16): the record is made of write-through stores; drain them, then ONE
        // relaxed agent-scope ticket add (two levels: 8 group counters and a top counter on separate 128-byte lines); the workgroup
        // whose add came last takes an agent-scope acquire and only then loads what the others stored.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tr) tr7 = __builtin_amdgcn_s_memrealtime();  // record drained
        uint32_t last = 0u;
        if (dedicated) {  // (the selector polls this word: nothing to wait for, the workgroup is through)
            if (lane == 0) __hip_atomic_store(&G.ready[bid], G.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) L.u.sel.last = 0u;
        } else if (lane == 0) {
            const uint32_t g = bid & 7u;
            const uint32_t n_in_group = (n_wg - g + 7u) >> 3;
            const uint32_t n_groups = n_wg < 8u ? n_wg : 8u;
            const uint32_t t1 = __hip_atomic_fetch_add(&SP.done_count[32u * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t1 == n_in_group - 1u) {
                const uint32_t t2 = __hip_atomic_fetch_add(&SP.done_count[32u * 8u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = (t2 == n_groups - 1u) ? 1u : 0u;
            }
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            L.u.sel.last = last;  // (the union's other member is dead: every wave of the workgroup has left the stream)
        }
    }
    if (tr && lane == 0) {
        tr[0] = tr0;
        tr[1] = tr1;
        tr[2] = tr2;
        tr[3] = tr3;
        tr[4] = __builtin_amdgcn_s_memrealtime();
        if (wave == 0u) {
            tr[5] = tr5;
            tr[6] = tr6;
            tr[7] = tr7;
        }
    }
    __syncthreads();
    if (!L.u.sel.last) return;
    __syncthreads();  // (everybody has read the flag before the selection's shared state is initialised around it)
    unsigned long long *sel_stamps = G.trace ? G.trace + (size_t)n_wg * 64u : nullptr;  // (behind the per-wave rows)
    if (sel_stamps && tid == 0) sel_stamps[7] = __builtin_amdgcn_s_memrealtime();  // selection starts (ticket drawn, acquire done)
    (void)select_local(G, SP, n_wg, tid, blockDim.x, L.u.sel, 1.0f, sel_stamps);
    if (tid == 0) {
        for (uint32_t c = 0; c < 9u; ++c) SP.done_count[32u * c] = 0u;
        if (sel_stamps) sel_stamps[5] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace tkspmv
