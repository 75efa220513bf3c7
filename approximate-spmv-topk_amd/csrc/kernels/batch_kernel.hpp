// kernels/batch_kernel.hpp -- batch_kernel: up to 32 queries per launch, one pass over the wave-BSCSR stream per query (the headline path).
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include <cstddef>
#include "stream_kernel.hpp"
#include "local.hpp"

namespace tkspmv {

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
// Attribution ladder (tools/ladder.sh; never defined in the product build): what the kernel of local thresholds costs rung by rung,
// measured on timing-only builds of the library. 1: packets loaded, unpacked, gathered, multiplied, scanned, trigger formed -- and
// never taken; 2: + the candidate path, the thresholds and everything a workgroup does in LDS for its record (staging, ranking);
// 3: + the record stored and drained before the ticket; 4 and up: + the selections (the product; pacing is the PACE option).
// Rungs below 4 return no results.
#ifndef TKSPMV_LADDER
#define TKSPMV_LADDER 9
#endif
#ifndef TKSPMV_CLOCK_STRIDE
#define TKSPMV_CLOCK_STRIDE 2  // (a power of two; same box, sustained / the driver's 20-query launch: 1: 16.8-17.0 / 17.2, 2: 16.8-17.1 / 17.0, 4: 17.1-17.5 / 17.7, 8: 17.3-18.2 / 17.8 us per query)
#endif
#ifndef TKSPMV_ORIGIN_OFFSET
#define TKSPMV_ORIGIN_OFFSET 0
#endif
#ifndef TKSPMV_TAU_WAIT
#define TKSPMV_TAU_WAIT 3000
#endif
constexpr unsigned long long BATCH_TAU_WAIT = TKSPMV_TAU_WAIT;  // x 10 ns (s_memrealtime runs at 100 MHz)
// (a workgroup-local threshold forms through LDS within the workgroup's own query -- or never, when some wave of it holds no row
//  above min_score: 3 us are plenty, and a workload filtered by min_score must not wait 30 us per query in such workgroups)
constexpr unsigned long long LOCAL_TAU_WAIT = 300;
constexpr int MISC_DBG_WAITS = 26, MISC_DBG_WAIT_TICKS = 27;  // TKSPMV_STATS=1 only
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
    unsigned long long *wg_cand0, *ovf_cand0;
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
    // ---- workgroup-local thresholds, checked by the selection (kernels/local.hpp; round 3, reworked in round 4) ---------------
    // local = 1 / 2: the threshold of a workgroup comes from its OWN waves, through LDS (a wave's word: the best / the second
    // best packet maximum it has seen), it may start at what the workgroup delivered for its previous query (wg_prior), nothing
    // is appended to global memory -- what does not fit a list or the workgroup's 8 slots is dropped under a recorded bound --, and
    // every workgroup delivers one record per query (lslots / lused: set s at s strides). The selection (select_local) checks
    // the records and reports a failed check in the launch's VERDICT word; the host switches the mode on only where a failure
    // is a once-in-thousands event (engine.hip). 0: the device-wide exchange (exact on its own; larger matrices, repair
    // phases, launches behind a closed gate).
    uint32_t local;
    // Selector workgroups: blocks 0 .. n_selectors-1 select, selector s the queries s, s + n_selectors, ... One selection is a
    // chain of several trips through global memory: a single selector is the slowest stage of the launch as soon as a query
    // streams faster than that (below ~500k rows).
    uint32_t n_selectors;
    float *wg_prior;        // [n_wg] what a workgroup's next query starts from, kept between launches (NULL: no carrying)
    // Round 5: a carried threshold belongs to queries that LOOK like the one it came from. While the server wave stages x it also
    // forms the query's signature -- rho = sum x / sum |x| (a change of sign turns it round) and pr = (sum |x|)^2 / (cols sum x^2)
    // (the share of the columns that carry the query: a query concentrated on a few columns has a small one); both are invariant
    // under scaling, like the carried threshold itself. A workgroup remembers TWO priors with their signatures and starts a query
    // from the one that matches (|rho - rho'| <= 0.25, pr within a factor of 2), or from none: the query that changed direction runs
    // without a carried threshold -- slower by a few us, exact, its check passes -- where round 4 let it fail its check and then
    // suspended carrying for the next 16 .. 4096 selections; the stream's usual direction finds its own prior again right behind it.
    // [n_wg][8]: rho0, pr0 (signature of wg_prior) | prior1, rho1, pr1 (the other one) | unused.
    float *wg_sig;
    uint32_t *prior_block;  // [0..3] suspension of carried thresholds after a failed check (prior_block_update), [4..7] the gate
    // The gate of the local thresholds: a launch of which a quarter or more failed its checks closes it for 8, 16, ... 1024 launches
    // (the matrix keeps its best rows together: a workgroup's 8 slots cannot hold them, whatever the thresholds do); it counts down
    // by one per launch, and 16 clean launches in a row halve the next closure. "Launches to go" is kept twice, prior_block[4] and
    // [7]: a launch READS the copy of its parity (every workgroup, whenever it starts) and its boundary thread WRITES the other
    // one -- the next launch's --, so that no word is read and written within one launch. [5]: length of a closure, [6]: clean run.
    uint32_t gate_parity;
    float local_beta;
    unsigned long long *lslots;  // records of local mode: [BATCH_MAX][n_wg][WG_SLOTS]
    uint32_t *lused;             //                       [BATCH_MAX][n_wg]
    uint32_t lslots_stride, lused_stride;
    // Pacing by rank (pace_quads != 0; see the note at the ticket add): the per-packet pause of the workgroups that led the field
    // in the previous query: pace_quads units of s_sleep(2) = 128 cycles, times pace_levels, pace_levels - 1, ..., 1 for the first,
    // second, ... eighth of the field (pace_levels = 3: three eighths pause).
    uint32_t pace_quads, pace_levels;
    uint32_t pace_base;   // units every workgroup pauses per packet whatever its rank (a uniform throttle; PACE_BASE, tuning runs)
    // Pacing by the clock (pace_period != 0; replaces the pauses by rank): every streaming wave keeps a timetable -- packet j of its
    // query q is due at (its first query's start) + (q x packets + j) x pace_period / packets, in 10 ns ticks << 8 -- and sleeps off
    // whatever it is ahead of it; a wave behind its timetable never pauses. The whole field then asks for the stream at the rate the
    // memory system can give: nobody queues, so nobody is favoured, and the workgroups end a launch together.
    uint32_t pace_period;  // ticks << 8 per query (0: off)
    // A period that follows the GPU: [0] what the waves add to pace_period (ticks << 8, >= 0), [1] the waves of this launch that
    // started its last query more than a quarter of a period behind their timetable. The selection that completes a launch of 8+
    // queries lengthens the period by 1/64 when a quarter of the waves were that late (the GPU streams slower than when the period was
    // measured) and takes 1/256 back when next to none were; never below what tkspmv_create measured, never more than 1/8 above.
    uint32_t *pace_adapt;
    unsigned long long *wg_times;  // optional (option WG_TIMES): [BATCH_MAX + 1][n_wg] s_memrealtime at every hand-over (row q) and at the workgroup's entry (row BATCH_MAX)
    uint32_t *wg_pace;    // [n_wg] the pause a workgroup ended the previous launch with: its first query here starts from it (NULL: from none)
    // ---- the verdict of a launch's checks and the repair launch -----------------------------------------------------------------
    // Every checked selection of a LOCAL launch adds 1 | failed << (32 + q) to the launch's verdict word (one 64-bit atomic: the
    // count of finished selections and the set of failed queries travel together). The exact launch behind it (repair = 1) reads the
    // completed word and runs the named queries again with the device-wide exchange -- none, almost always.
    unsigned long long *verdict;       // this launch's word
    unsigned long long *verdict_next;  // the next launch's word: zeroed by this one
    // Round 5: the exact launch behind EVERY local launch (6 us per launch to find out that nothing failed: 1.6 % of the driver's
    // 20-query region) is gone from the stream once the host has seen clean verdicts: the selection that completes the launch's
    // verdict also stores it in host-visible memory (verdict_host), and the host looks at it when it next waits for the stream
    // (EngineImpl::settle) -- a flagged query is then repaired by an exact launch with repair = 2, which takes the queries from
    // repair_mask instead of the (long reused) device word. repair = 1: the in-stream repair launch as before (caller's streams,
    // the launches after an observed failure, engines that have seen nothing yet).
    unsigned long long *verdict_host;
    uint32_t repair, repair_mask;
    // ---- overflow lists of the exact mode: ovf_lists of them (2), used round robin by the queries of a phase under flow control --
    // A list must be able to hold EVERY row (x = 0 makes every row a candidate and the result must still be exact): 8 bytes per
    // row. Round 3 kept one per query of a launch (256 MB at 1M rows, 2.6 GB at 10M); now query j of a phase uses list
    // j % 4 and may append to it once the selections of its earlier users have finished (ovf_epoch[l] counts them; a wave
    // checks only when it actually has something to append: staged rows go to the workgroup's slots first, but a wave whose
    // private list fills up before the query's threshold has arrived -- routine from ~1M rows on -- appends, so the pipeline is
    // four deep: with two lists the waves of query q stood waiting for the selection of q - 2, measured 30-60 us per query).
    uint32_t *ovf_epoch;  // [ovf_lists] adjacent words
    uint32_t ovf_lists;
    __device__ __forceinline__ unsigned long long *ovf_list(uint32_t l) const { return ovf_cand0 + (size_t)l * ovf_stride; }
    __device__ __forceinline__ uint32_t *ovf_list_count(uint32_t l) const { return ovf_count0 + (size_t)l * word_stride; }
};

template <int XCOLS, int C = 4>
struct BatchLds {
    union {
        struct {
            float x[2][XCOLS];                           // query vector, double-buffered by query parity (at LDS offset 0: the gathers OR a column's byte offset into the copy's base)
            uint2 cand[ListGeom<XCOLS>::CAND_CAP];       // private candidate lists of the streaming waves
        } w;
        SelectShared sel;        // selector workgroups only: the exact mode's selection
        LocalSelectShared lsel;  //                           local mode's
    } u;
    uint32_t misc[2][MISC_WORDS];                        // per query parity
    unsigned long long stg[2][8][STG_N];                 // survivors staged by the streaming waves
    uint32_t stg_cnt[2][8];
    uint32_t ck[64];  // score keys of the staged rows while the server ranks them (local mode)
    uint32_t pace;  // pause per packet (bits 0-7: 0..3 units) and issue priority (bit 8) of the streaming waves in their next query, set by the server
    uint32_t rq[BATCH_MAX + 2];  // repair phase: the flagged queries in order, [BATCH_MAX] their number; [BATCH_MAX + 1]: verdict hand-over
    uint32_t epoch0[4];     // ovf_epoch as it stood when the phase began
    uint32_t epoch_now[4];  // ... and as the server wave saw it last (what the streaming waves look at)
#ifndef TKSPMV_ALTERNATE_PRIO
#define TKSPMV_ALTERNATE_PRIO 1
#endif
};

__device__ __forceinline__ uint32_t lds_load(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One wave finalises a query of its workgroup in local mode: of the up to 8 x STG_N staged rows (lane l looks at entry l % 8 of
// wave l / 8) the 8 best go to the workgroup's record -- all 8 slots are written: row SLOT_INVALID where there are fewer --, the
// rest is dropped. With 8 rows or fewer (about every second query under carried thresholds) nothing is ranked: the rows go to the
// slots as the lanes hold them. More: rank by counting on the score keys, ties by lane (every lane against all 64, 16 keys per
// round of LDS reads). `used`: the order key above everything this workgroup dropped -- the thresholds it filtered with
// (MISC_TAUKEY), what its waves dropped (MISC_BOUND), what did not fit the slots. `next_prior`: what its next query may start from,
// in this query's score units (negative: nothing new to go by): the score of its 8th best row, else the threshold in force (it
// let fewer than 8 rows through: high enough), a little lower.
__device__ __forceinline__ void finalize_local_wave(const unsigned long long *stg, const uint32_t *stg_cnt, uint32_t *ck,
                                                    const uint32_t *misc, uint32_t lane, float min_units, unsigned long long *slots,
                                                    uint32_t &used, float &next_prior) {
    const bool valid = (lane & 7u) < stg_cnt[lane >> 3];
    const unsigned long long v = valid ? stg[lane] : 0ull;
    const uint64_t bv = __ballot(valid);
    const uint32_t n_valid = (uint32_t)__popcll(bv);
    used = 0u;
    next_prior = -1.0f;
    if (n_valid <= WG_SLOTS) {  // (wave-uniform)
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bv, 0u));
        if (TKSPMV_LADDER >= 3 && valid) st_agent(slots + pos, v);
        if (TKSPMV_LADDER >= 3 && lane < WG_SLOTS && lane >= n_valid) st_agent(slots + lane, pack_cand(0u, SLOT_INVALID));
        if (n_valid == WG_SLOTS) next_prior = -wave_max(valid ? -__uint_as_float((uint32_t)v) : -__builtin_huge_valf());  // the smallest of the 8
    } else {
        const uint32_t mk = valid ? order_key(__uint_as_float((uint32_t)v)) : 0u;
        ck[lane] = mk;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations execute in order)
        // (16 keys per round, their LDS reads issued back to back before the first compare: left to itself the scheduler keeps two
        //  reads in flight and pays the LDS latency for each -- the server wave's chain per query is what bounds small matrices)
        uint32_t r = 0;
#pragma unroll 1
        for (uint32_t j0 = 0; j0 < 64; j0 += 16) {
            uint32_t o[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) o[j] = ck[j0 + j];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) r += (o[j] > mk || (o[j] == mk && j0 + j < lane)) ? 1u : 0u;
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 96, 0);
        }
        if (TKSPMV_LADDER >= 3 && valid && r < WG_SLOTS) st_agent(slots + r, v);
        const uint64_t b8 = __ballot(valid && r == WG_SLOTS);  // the best row that did not fit, one step up (a dropped row may tie with it)
        if (b8 != 0ull) used = order_key(__uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)__builtin_ctzll(b8)))) + 1u;
        const uint64_t b7 = __ballot(valid && r == WG_SLOTS - 1u);
        if (b7 != 0ull) next_prior = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)__builtin_ctzll(b7)));
    }
    {
        const uint32_t k_thr = __hip_atomic_load(&misc[MISC_TAUKEY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t k_drop = __hip_atomic_load(&misc[MISC_BOUND], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t km = __builtin_amdgcn_readfirstlane(k_thr > k_drop ? k_thr : k_drop);
        used = used > km ? used : km;
    }
    if (next_prior < 0.0f) {
        const float t_end = __uint_as_float(__hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (t_end > min_units) next_prior = t_end * 0.95f;
    }
}

// The gate's bookkeeping (BatchParams::gate_parity), by ONE thread per launch: `failed` checks among the launch's n_q.
__device__ __forceinline__ void gate_update(const BatchParams &B, uint32_t failed) {
    uint32_t *g = B.prior_block;
    const uint32_t cur = g[B.gate_parity ? 7 : 4];
    uint32_t next = cur;
    if (B.n_q >= 4u && 4u * failed >= B.n_q) {
        const uint32_t len = g[5] < 8u ? 8u : (g[5] >= 512u ? 1024u : 2u * g[5]);
        g[5] = len;
        g[6] = 0u;
        next = len;
    } else if (cur != 0u) {
        next = cur - 1u;
    } else if (failed == 0u && ++g[6] >= 16u) {
        g[6] = 0u;
        if (g[5] > 8u) g[5] /= 2u;
    }
    g[B.gate_parity ? 4 : 7] = next;
}

// DBG = false (production): the tracing / statistics / ablation hooks of StreamParams (trace, dbg, stamps) are compiled
// out -- no per-packet compare of a tracing word or an ablation flag, and the scalar registers they held are free. The
// engine launches the DBG = true instantiation only when TKSPMV_TRACE / TKSPMV_STATS / TKSPMV_DBG_FLAGS ask for it.
//
// One PHASE of a launch: phase 0 = the launch's n_q queries in the mode the host chose (B.local, the gate permitting); phase 1 =
// the queries whose check failed in phase 0 (L.rq), with the device-wide exchange. Returning from here ends the phase for the
// calling wave; the kernel below puts the phases together.
template <int C, int XCOLS, int QM, bool DBG, bool LOCAL>
__device__ __forceinline__ void batch_phase(const StreamParams &P0, const SelectParams &SP0, const BatchParams &B, const bool repair,
                                            BatchLds<XCOLS, C> &L) {
    constexpr bool Q8 = QM == 1 || QM == 2;  // x staged as Q1.7 integers
    constexpr int VT = value_type_of(QM);
    constexpr int NBUF = C == 8 ? 2 : 3;  // packets of 8 entries per lane are twice as large: one ahead is as many bytes
    constexpr uint32_t WAVE_CAP = ListGeom<XCOLS>::WAVE_CAP;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nwaves = (blockDim.x >> 6) - 1u;  // streaming waves
    const bool is_server = (wave == nwaves);
    const uint32_t nq = repair ? L.rq[BATCH_MAX] : B.n_q;
    // query q of THIS phase in the launch's argument block (repair: the q-th flagged query)
    auto qx = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t { return repair ? L.rq[q] : q; };
    // exchange-state set / ticket counter of query q
    auto set_of = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t { return qx(q); };
    constexpr bool local = LOCAL;  // (compile-time: the two modes are two kernels, neither carries the other's code or registers)
    const bool local_top1 = B.local == 1u;  // a wave's word is its best packet maximum (1) or its second best (2)
    const uint32_t pace_q = LOCAL ? B.pace_quads : 0u;
    // Overflow list of query q of this phase and the value its epoch word must show before anything may be appended: the
    // selections of the list's earlier users in this phase have finished (L.epoch0: the words as the phase found them).
    // (BatchParams::ovf_lists: 4, or 2 where the exact kernel only repairs behind the kernel of local thresholds -- a power of two)
    const uint32_t n_lists = local ? 4u : B.ovf_lists, lists_shift = n_lists >= 4u ? 2u : (n_lists >> 1);
    auto list_of = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t { return q & (n_lists - 1u); };

    const uint32_t nsel = B.n_selectors;  // (>= 1)
    if (blockIdx.x < nsel) {
        // ---- selector workgroups -----------------------------------------------------------------------------
        const uint32_t n_stream = gridDim.x - nsel;
        // (the lists' epoch words as the phase found them -- read before any selection of the phase can have finished)
        if (!local && tid < 4u) L.epoch0[tid] = tid < n_lists ? __hip_atomic_load(B.ovf_epoch + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        for (uint32_t q = blockIdx.x; q < nq; q += nsel) {
            const bool tr_sel = P0.trace && q == 4u;  // (tools/batch_trace.py, NQ=8: the phases of one selection)
            if (tr_sel && tid == 0) {
                P0.trace[8] = __builtin_amdgcn_s_memrealtime();
                P0.trace[9] = __builtin_amdgcn_s_memtime();
            }
            if (tid == 0) {
                uint32_t *t = B.tickets + 32u * set_of(q);
                // Polled with a compare-and-swap (which also resets the counter for the next use): atomics execute at the
                // device-wide coherence point, whereas a load -- even agent-scope -- can keep hitting a stale copy of the line
                // in this XCD's L2 (seen: 33 ms on an otherwise idle L2).
                // (the last query's selection is the launch's tail: poll it faster)
                while (atomicCAS(t, n_stream, 0u) != n_stream) {
                    if (local || q + nsel >= nq) __builtin_amdgcn_s_sleep(4);
                    else __builtin_amdgcn_s_sleep(32);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (LOCAL && B.wg_times) B.wg_times[(size_t)set_of(q) * gridDim.x + n_stream] = __builtin_amdgcn_s_memrealtime();  // (option WG_TIMES: all tickets seen)
            }
            __syncthreads();
            SelectParams S = SP0;
            S.out_idx = B.io[qx(q)].out_idx;
            S.out_val = B.io[qx(q)].out_val;
            if (local) {
                // the workgroups' records of this query, checked: a failed check goes into the launch's verdict
                LocalParams G{};
                G.slots = B.lslots + (size_t)set_of(q) * B.lslots_stride;
                G.used = B.lused + (size_t)set_of(q) * B.lused_stride;
                G.prior_block = B.wg_prior ? B.prior_block : nullptr;
                G.shared_bookkeeping = 1u;  // (several selectors at once)
                S.host_out = nullptr;
                const float out_scale = __hip_atomic_load(B.unit_inv(set_of(q)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tr_sel && tid == 0) P0.trace[10] = __builtin_amdgcn_s_memrealtime();
                const bool bad = TKSPMV_LADDER < 4 ? false : select_local(G, S, n_stream, tid, blockDim.x, L.u.lsel, out_scale, tr_sel ? P0.trace + 11 : nullptr);
                __syncthreads();
                if (tid == 0 && B.verdict) {
                    const unsigned long long add = 1ull | ((bad ? 1ull : 0ull) << (32u + q));
                    const unsigned long long done = __hip_atomic_fetch_add(B.verdict, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
                    // (the selection that completes the word tells the host; read there only after the launch has ended)
                    if (B.verdict_host && (uint32_t)done == B.n_q) __hip_atomic_store(B.verdict_host, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (LOCAL && B.pace_adapt && B.pace_period != 0u && (uint32_t)done == B.n_q && B.n_q >= 8u) {
                        // (every streaming wave has long passed the start of its last query: the count is complete)
                        const uint32_t late = __hip_atomic_exchange(B.pace_adapt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t n_waves = n_stream * 8u, adj = B.pace_adapt[0];
                        uint32_t next = adj;
                        if (4u * late > n_waves) next = adj + (B.pace_period >> 6);
                        else if (64u * late < n_waves) next = adj > (B.pace_period >> 8) ? adj - (B.pace_period >> 8) : 0u;
                        if (next > (B.pace_period >> 3)) next = B.pace_period >> 3;
                        if (next != adj) B.pace_adapt[0] = next;
                    }
                    if (LOCAL && B.wg_times) B.wg_times[(size_t)set_of(q) * gridDim.x + n_stream + 1u] = __builtin_amdgcn_s_memrealtime();  // (option WG_TIMES: selection done)
                }
            } else {
                const uint32_t l = list_of(q);
                // Fewer lists than selector workgroups: the list's previous user may be with ANOTHER selector, still selecting --
                // its entries and its count are in the list until that selection has reset them (with as many lists as selectors
                // a list always comes back to the same workgroup, and this is true at once). Nobody has appended for q meanwhile:
                // appending waits for the same word.
                if (tid == 0) {
                    const uint32_t need = L.epoch0[l] + (q >> lists_shift);
                    while (__hip_atomic_load(B.ovf_epoch + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != need) __builtin_amdgcn_s_sleep(8);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                S.wg_cand = B.wg_cand(set_of(q));
                S.ovf_cand = B.ovf_list(l);
                S.ovf_count = B.ovf_list_count(l);
                S.gmax = B.gmax(set_of(q));
                S.tau_g = B.tau_g(set_of(q));
                S.unit_inv_in = B.unit_inv(set_of(q));
                // The sets and lists are reused INSIDE a launch (a repair phase behind phase 0, the lists round robin): the resets at the end of a selection are written through and drained before anybody is told.
                S.wt_reset = 1u;
                if (tr_sel && tid == 0) P0.trace[10] = __builtin_amdgcn_s_memrealtime();
                select_body<0>(S, tid, blockDim.x, L.u.sel, tr_sel ? P0.trace + 8 : nullptr);
                // the list is free for its next user: its count was reset (written through) and drained above
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) (void)__hip_atomic_fetch_add(B.ovf_epoch + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tr_sel && tid == 0) P0.trace[15] = __builtin_amdgcn_s_memrealtime();
            if (P0.trace && tid == 0 && q < 8u) P0.trace[q] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    const uint32_t bid = blockIdx.x - nsel, n_wg = gridDim.x - nsel;
// traced queries: the first, the middle one and (batches of 8 or more) the one after it, else the last
#define TRSLOT(q) ((q) == 0u ? 0u : ((q) == nq / 2u ? 1u : ((q) == (nq >= 8u ? nq / 2u + 1u : nq - 1u) ? 2u : 9u)))
    unsigned long long *trw = P0.trace ? P0.trace + ((size_t)blockIdx.x * 9u + wave) * 8u : nullptr;
    if (trw && lane == 0 && !repair) trw[0] = __builtin_amdgcn_s_memrealtime();
    if (tid < 2u * MISC_WORDS) (&L.misc[0][0])[tid] = 0u;
    if (tid < 16u) (&L.stg_cnt[0][0])[tid] = 0u;
    if (tid == 0u) L.pace = (LOCAL && B.wg_pace && B.pace_quads != 0u && !repair) ? B.wg_pace[blockIdx.x - B.n_selectors] : 0u;
    if (!local && tid < 4u) {  // (local mode appends nothing to global memory: no lists, no epochs -- one trip through memory less at the head of the launch)
        const uint32_t e = tid < n_lists ? __hip_atomic_load(B.ovf_epoch + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        L.epoch0[tid] = e;
        L.epoch_now[tid] = e;
    }
    __syncthreads();
    const uint32_t grp_local = is_server ? 0u : wave * P0.gpw / nwaves;
    const bool publishes = !local && (bid * P0.gpw + grp_local) < P0.n_groups_pub;  // (local: the waves publish by themselves)
    const bool reducer = bid < P0.n_reducers;
    // Streaming waves that own a partition (wave w streams partition w * n_wg + bid): only they take part in the
    // per-query protocol. Waves without one leave the phase at once -- spinning at stream priority on every query's x flag, six
    // of them per workgroup on a small matrix, they starved the server wave (585 us per query at 50k rows).
    // (the server counts them with ONE load instruction -- lane w looks at wave w's partition)
    uint32_t n_active = 0;
    bool wave_has = false;  // (lane w: wave w streams a partition -- the server counts them, local thresholds need to know who takes part)
    if (is_server || local) {
        const uint32_t pw = lane * n_wg + bid;
        wave_has = lane < nwaves && pw < P0.n_parts && (P0.uni_ppp != 0u || P0.part_count[pw] != 0u);
        n_active = (uint32_t)__popcll(__ballot(wave_has));
    }
    // what the epoch word of query q's overflow list must show before anything is appended to the list
    auto ovf_need = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t { return L.epoch0[list_of(q)] + (q >> lists_shift); };
    // (the streaming waves look at the server wave's LDS copy of the epoch words, refreshed once per turn of its loop)
    auto ovf_wait = [&](uint32_t q) __attribute__((always_inline)) {
        const uint32_t need = ovf_need(q);
        while (__builtin_amdgcn_readfirstlane(lds_load(&L.epoch_now[list_of(q)])) != need) {
            if (is_server && lane < n_lists) L.epoch_now[lane] = __hip_atomic_load(B.ovf_epoch + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(8);
        }
    };

    if (is_server) {
        // ---- server wave: x staging, threshold exchange of the newest query, finalisation of the oldest -------
        // The reducers' search must not starve: in a batch the streaming waves (priority 2) never pause, and a reducer at
        // the default priority got ONE pass per query (traced), i.e. the threshold arrived when the query was over.
        // (local thresholds: the query is short and the workgroup's waves wait for this wave's next x: it must not queue behind them)
        if (reducer || local) __builtin_amdgcn_s_setprio(3);
        const unsigned long long t_wg_entry = (LOCAL && B.wg_times) ? __builtin_amdgcn_s_memrealtime() : 0ull;  // (option WG_TIMES)
        uint32_t staged = 0u, tail = 0u;
        uint32_t gate_seen = 0u;  // (exact mode: queries [0, gate_seen) have had their overflow list seen free by this wave)
        const bool carry_local = local && B.wg_prior != nullptr;
        float wg_prior = carry_local ? B.wg_prior[bid] : 0.0f;  // (reported-score units per unit of the query's L1 norm; 0: none)
        float xnorm_q[2] = {0.0f, 0.0f};  // sum |x| of the queries in flight (carried thresholds are relative to it: scores are linear in x)
        // the two remembered priors with their signatures (slot 0: wg_prior, the most recently used), and per query in flight: its
        // signature and the slot it started from (2: none)
        // (signatures are formed from the fp32 copy of x in LDS: the integer-staged value types keep round 4's behaviour)
        const bool sigs = carry_local && B.wg_sig != nullptr && !Q8 && QM != 4 && QM != 6 && QM != 8;
        float p_rho[2] = {0.0f, 0.0f}, p_pr[2] = {0.0f, 0.0f}, prior1 = 0.0f;
        // (wave-uniform values: kept in scalar registers -- as vector registers they cost the kernel 60 spilled registers)
        auto uni = [](float v) __attribute__((always_inline)) -> float { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
        if (sigs) {
            const uint32_t *g = reinterpret_cast<const uint32_t *>(B.wg_sig + (size_t)bid * 8u);
            p_rho[0] = __uint_as_float(scalar_load(g + 0));
            p_pr[0] = __uint_as_float(scalar_load(g + 1));
            prior1 = __uint_as_float(scalar_load(g + 2));
            p_rho[1] = __uint_as_float(scalar_load(g + 3));
            p_pr[1] = __uint_as_float(scalar_load(g + 4));
        }
        float q_rho[2] = {0.0f, 0.0f}, q_pr[2] = {0.0f, 0.0f};
        uint32_t q_slot[2] = {2u, 2u};

        float inv_unit_q[2] = {1.0f, 1.0f}, min_units_q[2] = {0.0f, 0.0f};
        unsigned long long dbg_first_duty = 0ull;
        uint32_t dbg_iters = 0u;
        for (;;) {
            if (staged < nq && staged - tail < 2u) {
                const uint32_t par = staged & 1u;
                // (wave-uniform, and said so: a resource descriptor the compiler cannot prove uniform gets a loop of readfirstlanes around
                //  EVERY load -- the sixteen loads of x one by one again)
                const float *xg;
                {
                    const uint64_t xa = (uint64_t)(uintptr_t)B.io[qx(staged)].x;
                    // (the builtin returns int: through uint32_t, or a low word with its top bit set sign-extends into the high one)
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xa >> 32));
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xa);
                    xg = reinterpret_cast<const float *>((uintptr_t)(((uint64_t)hi << 32) | (uint64_t)lo));
                }
                const uint32_t prior_blocked = carry_local ? __hip_atomic_load(B.prior_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
                const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xg), 0, (int)(P0.cols * 4u), 0x00020000);
                auto x_at = [&](uint32_t i) __attribute__((always_inline)) -> float {  // x[i], 0 beyond the columns (the resource's bounds check)
                    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xres, i * 4u, 0, 0));
                };
                float x_scale = 1.0f, unit_scale = 1.0f;
                if (QM == 2) {
                    float lm = 0.0f;
#pragma unroll 1
                    for (uint32_t b0 = 0; b0 < (uint32_t)XCOLS; b0 += 1024u) {  // 16 loads in flight per lane
                        float r[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const uint32_t i = b0 + lane + 64u * (uint32_t)u;
                            r[u] = x_at(i);
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) lm = fmaxf(lm, (b0 + lane + 64u * (uint32_t)u) < P0.cols ? r[u] : 0.0f);
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
                } else if (QM == 4 || QM == 6 || QM == 8) {
                    unit_scale = 2147483648.0f;
                }
                inv_unit_q[par] = 1.0f / unit_scale;
                min_units_q[par] = P0.min_score * unit_scale;
                float *xl = L.u.w.x[par];
                float xabs = 0.0f, xsum = 0.0f, xsq = 0.0f;
#pragma unroll 1
                for (uint32_t b0 = 0; b0 < (uint32_t)XCOLS; b0 += 1024u) {  // 16 loads in flight per lane
                    float r[16];
                    // (Round 4 found these sixteen loads issued one by one -- "in range ? load : 0" put each into a branch of its own behind
                    //  an s_waitcnt vmcnt(0): 6-16 us per staging -- and cured it with clamped addresses and masked values, an address pair
                    //  per load: registers the exact kernel did not have. Round 5: BUFFER loads -- the resource carries x's length, a load
                    //  beyond it returns 0 by itself; one lane offset, sixteen immediate offsets, sixteen loads in flight whatever the
                    //  number of columns, in both kernels.)
#pragma unroll
                    for (int u = 0; u < 16; ++u) r[u] = x_at(b0 + lane + 64u * (uint32_t)u);
                    if (carry_local) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) xabs += fabsf(r[u]);
                    }

#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const uint32_t i = b0 + lane + 64u * (uint32_t)u;
                        if (Q8)
                            reinterpret_cast<uint32_t *>(xl)[i] = to_q1_7_dev(r[u] * x_scale);
                        else if (QM == 6)
                            reinterpret_cast<uint32_t *>(xl)[i] = to_fixed_dev(r[u], P0.fixed_width) >> 12;
                        else if (QM == 4 || QM == 8)
                            reinterpret_cast<uint32_t *>(xl)[i] = to_fixed_dev(r[u], P0.fixed_width) >> (P0.fixed_width <= 24u ? 8 : 0);
                        else
                            xl[i] = QM == 5 ? r[u] * Q17_UNIT : r[u];
                    }
                }
                uint32_t *mp = L.misc[par];
                if (lane < (uint32_t)MISC_WORDS && lane != (uint32_t)MISC_XREADY) mp[lane] = 0u;
                if (lane < 8u) L.stg_cnt[par][lane] = 0u;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                float tau_init = min_units_q[par];
                uint32_t carried_key = 0u;
                if (carry_local) xnorm_q[par] = wave_sum_f32(xabs);  // (a fixed summation order: the same value in every run)
                float use_prior = wg_prior;
                if (sigs) {
                    // (the signature's sums come from the staged copy in LDS, behind the loads -- inside the staging loop they cost the
                    //  kernel 60 spilled registers --, and from a QUARTER of the columns: four 64-column blocks, 256 values; the chain
                    //  from the last load to the x-ready flag is what the workgroup's fastest waves wait for)
                    float xabs_s = 0.0f;
#pragma unroll
                    for (uint32_t i = lane; i < (uint32_t)XCOLS; i += (uint32_t)XCOLS / 4u) {
                        const float v = xl[i];
                        xsum += v;
                        xabs_s += fabsf(v);
                        xsq += v * v;
                    }
                    const float s1 = wave_sum_f32(xsum), s2 = wave_sum_f32(xsq), n1 = wave_sum_f32(xabs_s);
                    const float n_cols = (float)(P0.cols < (uint32_t)XCOLS ? (P0.cols + 3u) / 4u : (uint32_t)XCOLS / 4u);
                    const float rho = uni(n1 > 0.0f ? s1 / n1 : 0.0f);
                    const float pr = uni(s2 > 0.0f ? n1 * n1 / (n_cols * s2) : 0.0f);
                    q_rho[par] = rho;
                    q_pr[par] = pr;
                    auto like = [&](float r2, float p2) __attribute__((always_inline)) -> bool {
                        return n1 > 0.0f && fabsf(rho - r2) <= 0.25f && pr <= 2.0f * p2 && p2 <= 2.0f * pr;
                    };
                    if (wg_prior > 0.0f && like(p_rho[0], p_pr[0])) q_slot[par] = 0u;
                    else if (prior1 > 0.0f && like(p_rho[1], p_pr[1])) {
                        q_slot[par] = 1u;
                        use_prior = prior1;
                    } else {
                        q_slot[par] = 2u;
                        use_prior = 0.0f;  // (a query unlike both remembered ones starts without a carried threshold)
                    }
                }
                if (prior_blocked == 0u && use_prior > 0.0f) {
                    const float t0 = use_prior * xnorm_q[par] * unit_scale * B.local_beta;
                    if (t0 > tau_init) {
                        tau_init = t0;
                        carried_key = order_key(t0);  // (on record with the thresholds the waves form: MISC_TAUKEY)
                    }
                }
                if (lane == 0) {
                    mp[MISC_TAU] = __float_as_uint(tau_init);
                    mp[MISC_MINU] = __float_as_uint(min_units_q[par]);
                    if (carry_local) mp[MISC_TAUKEY] = carried_key;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(&mp[MISC_XREADY], staged + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (trw && lane == 0 && TRSLOT(staged) < 3u) trw[1 + TRSLOT(staged)] = __builtin_amdgcn_s_memrealtime();
                if (LOCAL && B.wg_times && staged == 0u && lane == 0) {  // the entry stamp, and in its top byte how long the first x took (0.1 us)
                    const unsigned long long d = (__builtin_amdgcn_s_memrealtime() - t_wg_entry) / 10ull;
                    B.wg_times[(size_t)BATCH_MAX * gridDim.x + bid] = (t_wg_entry & 0x00FFFFFFFFFFFFFFull) | ((d > 255ull ? 255ull : d) << 56);
                }
                ++staged;
            }
            // Threshold exchange of the query this workgroup's waves are streaming: the oldest unfinished one until
            // half of the waves have left it, then the next (whose waves need a threshold most).
            if (local) {
                // (workgroup-local thresholds are formed by the streaming waves themselves, in LDS: nothing to do here)
            } else if (P0.n_sets != 0u) {
                // The overflow lists' epoch words for the streaming waves (L.epoch_now): loaded with this turn's exchange traffic and
                // stored below -- but only while the list of a query in flight has not been SEEN free yet (gate_seen: the queries, in
                // order, whose lists this wave has seen free). In the steady state that is one load per query: 512 servers polling
                // the four words every turn doubled the traffic on the exchange's memory channel, thresholds arrived late, and a
                // query of 19 packets per wave took 27-36 us instead of 18 (round 4, 1M rows with the device-wide exchange).
                while (gate_seen < staged && (int32_t)(__builtin_amdgcn_readfirstlane(lds_load(&L.epoch_now[list_of(gate_seen)])) - ovf_need(gate_seen)) >= 0) ++gate_seen;
                const bool poll_epochs = gate_seen < staged;
                const uint32_t e_now = (poll_epochs && lane < n_lists) ? __hip_atomic_load(B.ovf_epoch + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                uint32_t hq = tail;
                if (tail + 1u < staged &&
                    2u * __builtin_amdgcn_readfirstlane(lds_load(&L.misc[tail & 1u][MISC_DONE])) >= n_active)
                    hq = tail + 1u;
                {
                    const uint32_t sq = hq;
                    StreamParams P = P0;
                    P.gmax = B.gmax(set_of(sq));
                    P.tau_g = B.tau_g(set_of(sq));
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
                    if (poll_epochs && lane < n_lists) L.epoch_now[lane] = e_now;
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
                bool deliver = tail < staged && __builtin_amdgcn_readfirstlane(lds_load(&mp[MISC_DONE])) >= n_active;
                // Exact mode: staged rows beyond the workgroup's 8 slots go to the query's overflow list, which it shares with the
                // queries 4, 8, ... before it -- their selections must have finished (flow control). This wave never WAITS for
                // that: a waiting server stops publishing its workgroup's maxima for the younger query, everybody's threshold for
                // that query forms later, more rows survive everywhere, more workgroups need the lists -- 27-36 us per query at 1M
                // rows where 18-20 is the norm (round 4). It puts the hand-over off to a later turn and keeps the exchange going.
                uint32_t n_have = 0u;
                if (!local && deliver) {
                    const uint32_t c = lane < nwaves ? (L.stg_cnt[tp][lane] < 8u ? L.stg_cnt[tp][lane] : 8u) : 0u;  // (as the hand-over below counts them)
                    n_have = wave_sum_u32(c);
                    if (n_have > WG_SLOTS && __builtin_amdgcn_readfirstlane(lds_load(&L.epoch_now[list_of(tail)])) != ovf_need(tail)) deliver = false;
                }
                if (deliver) {
                    asm volatile("" ::: "memory");
                    StreamParams P = P0;
                    P.gmax = B.gmax(set_of(tail));
                    if (P0.dbg && lane == 0) {  // TKSPMV_STATS=1
                        atomicAdd(&P0.dbg[0], (unsigned long long)mp[MISC_SLOW_CNT]);
                        atomicAdd(&P0.dbg[1], (unsigned long long)mp[MISC_CAND_CNT]);
                        atomicAdd(&P0.dbg[5], (unsigned long long)mp[MISC_DBG_WAIT_TICKS]);
                        atomicAdd(&P0.dbg[6], (unsigned long long)mp[MISC_DBG_WAITS]);
                    }
                    if (local) {
                        // the workgroup's record of this query: its 8 best staged rows in order + what everything dropped lay below
                        uint32_t used = 0u;
                        float next_prior = -1.0f;
                        finalize_local_wave(&L.stg[tp][0][0], L.stg_cnt[tp], L.ck, mp, lane, min_units_q[tp],
                                            B.lslots + (size_t)set_of(tail) * B.lslots_stride + (size_t)bid * WG_SLOTS, used, next_prior);
                        if (TKSPMV_LADDER >= 3 && lane == 8u)
                            __hip_atomic_store(B.lused + (size_t)set_of(tail) * B.lused_stride + bid, used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (carry_local && next_prior >= 0.0f && xnorm_q[tp] > 0.0f) {
                            const float np_rel = sigs ? uni(next_prior * inv_unit_q[tp] / xnorm_q[tp]) : next_prior * inv_unit_q[tp] / xnorm_q[tp];
                            if (!sigs) {
                                wg_prior = np_rel;
                            } else {
                                // The refreshed prior becomes slot 0 (the most recently used) with the query's own signature; if the
                                // query started from slot 1 or from neither, the old slot 0 moves down (the other one is forgotten).
                                // The other query in flight chose its slot by the old numbering: renumbered with the slots -- only
                                // which slot it will push out depends on that, a value always travels with its signature.
                                const uint32_t sl = q_slot[tp];
                                if (sl != 0u) {
                                    prior1 = wg_prior;
                                    p_rho[1] = p_rho[0];
                                    p_pr[1] = p_pr[0];
                                    const uint32_t o = q_slot[tp ^ 1u];
                                    q_slot[tp ^ 1u] = o == 0u ? 1u : (o == 1u ? (sl == 1u ? 0u : 2u) : 2u);
                                }
                                wg_prior = np_rel;
                                p_rho[0] = q_rho[tp];
                                p_pr[0] = q_pr[tp];
                            }
                        }
                    } else {
                        if (P0.n_sets != 0u) publish_group_max(P, bid, lane, mp);  // complete maxima (fire and forget)
                        // lane l looks at entry (l % 8) of wave (l / 8): the staged rows go to the workgroup's 8 slots, in any
                        // order; only a ninth and later ones go to the query's overflow list (under its flow control)
                        const uint32_t w = lane >> 3, e = lane & 7u;
                        const bool have = e < L.stg_cnt[tp][w];
                        const unsigned long long v = have ? L.stg[tp][w][e] : 0ull;
                        const uint64_t bh = __ballot(have);
                        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(bh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bh, 0u));
                        if (have && pos < WG_SLOTS) st_agent(B.wg_cand(set_of(tail)) + (size_t)bid * WG_SLOTS + pos, v);
                        if (n_have > WG_SLOTS) {  // (the list was seen free above)
                            uint32_t gbase = 0u;
                            if (lane == 0) gbase = atomicAdd(B.ovf_list_count(list_of(tail)), n_have - WG_SLOTS);
                            gbase = __builtin_amdgcn_readfirstlane(gbase);
                            const uint32_t gp = gbase + pos - WG_SLOTS;
                            if (have && pos >= WG_SLOTS && gp < P0.ovf_cap) st_agent(&B.ovf_list(list_of(tail))[gp], v);
                        }
                    }
                    if (bid == 0u && lane == 0)
                        __hip_atomic_store(B.unit_inv(set_of(tail)), inv_unit_q[tp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // Hand-off as in the fused tail (cdna_hip_programming.md Guideline 16): everything above is a
                    // write-through (sc1) store; drain them, then a RELAXED agent-scope add. A release-ordered atomic
                    // would write back the whole L2 (buffer_wbl2) once per workgroup and query: measured 4 ms/query.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // The ticket's value tells this workgroup where it stands in the field: among the first to deliver a
                    // query, it is ahead of the others; among the last, behind. Workgroups are not equally fast -- the two on a
                    // CU do not share it evenly (the older one wins the arbitration), and the less a wave does per packet the
                    // more that shows: with the candidate path switched off the median wave streams a query in 11.5 us while a
                    // tenth of the workgroups take 26-28 us, and the launch waits for them (tools/batch_trace.py). The rank is
                    // the feedback: an early workgroup pauses a little per packet, a late one gets the higher issue priority.
                    uint32_t paced_units = 0u;  // (option WG_TIMES: the pause chosen here travels in the stamp's top byte)
                    if (pace_q != 0u) {
                        uint32_t rank = 0u;
                        if (lane == 0) rank = __hip_atomic_fetch_add(B.tickets + 32u * set_of(tail), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        rank = __builtin_amdgcn_readfirstlane(rank);
                        // the pause per packet of the next query (bits 0-7, units of pace_quads x 128 cycles): 3, 2, 1 for the first
                        // three eighths of the field; bit 8: the last third gets the higher issue priority
                        const uint32_t e8 = 8u * rank / n_wg;  // 0..7
                        uint32_t units = (e8 < B.pace_levels ? (B.pace_levels - e8) * pace_q : 0u) + B.pace_base;  // (per packet, units of s_sleep(2) = 128 cycles)
                        units = units > 255u ? 255u : units;
                        const uint32_t lvl = units | (3u * rank >= 2u * n_wg ? 256u : 0u);
                        paced_units = units;
                        if (lane == 0) __hip_atomic_store(&L.pace, lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    } else if (lane == 0)
                        (void)__hip_atomic_fetch_add(B.tickets + 32u * set_of(tail), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (LOCAL && B.wg_times && lane == 0)
                        B.wg_times[(size_t)set_of(tail) * gridDim.x + bid] = (__builtin_amdgcn_s_memrealtime() & 0x00FFFFFFFFFFFFFFull) | ((unsigned long long)paced_units << 56);
                    if (trw && lane == 0 && TRSLOT(tail) < 3u) trw[4 + TRSLOT(tail)] = __builtin_amdgcn_s_memrealtime();
                    if (trw && lane == 0 && TRSLOT(tail) == 1u) {
                        trw[3] = dbg_first_duty;
                        trw[6] = dbg_iters;  // (overwritten by the last query's finalise stamp; read when nq is small only)
                    }
                    ++tail;
                }
            }
            if (tail == nq) {
                if (carry_local && lane == 0) B.wg_prior[bid] = wg_prior;
                if (sigs && lane == 0) {
                    float *g = B.wg_sig + (size_t)bid * 8u;
                    g[0] = p_rho[0];
                    g[1] = p_pr[0];
                    g[2] = prior1;
                    g[3] = p_rho[1];
                    g[4] = p_pr[1];
                }
                if (pace_q != 0u && B.wg_pace && lane == 0 && !repair) B.wg_pace[bid] = lds_load(&L.pace);
                break;
            }
            if (local) __builtin_amdgcn_s_sleep(2);
            else if (reducer) __builtin_amdgcn_s_sleep(TKSPMV_REDUCER_SLEEP);
            else __builtin_amdgcn_s_sleep(8);
        }
        return;
    }

    // ---- streaming waves ---------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);
    const uint32_t part = wave * n_wg + bid;
    uint32_t p0 = 0, np = 0;
    if (part < P0.n_parts) TKSPMV_PARTITION_RANGE(P0, part, p0, np);
    uint2 *wcand = L.u.w.cand + wave * WAVE_CAP;
    if (np == 0u) return;  // no partition (n_active does not count this wave)
    static_assert(C == 4 || C == 8, "the batch kernel is built for 4 or 8 entries per lane");
    constexpr bool INT = int_sums<QM>();

    Pkt<C, VT> buf[NBUF];
    // Requests run NBUF - 1 packets ahead of the reduction, through the queries of the phase in order. Per request (round 5: 5
    // scalar instructions where round 4 spent ~25 and three 64-bit vector adds): a down-counter, one scalar add on the packet's byte
    // offset; the query's stream copy changes behind the counter's zero. Past the end of the phase the last packet is requested
    // again (step 0: a fixed number of younger loads lets the compiler wait with a counted vmcnt). fp32 streams are fetched with
    // buffer loads (load_packet_buf): resource = this wave's partition in the query's stream copy. The row base of a packet is
    // looked up on the candidate path only (pkt_row[p0 + jc], a scalar load where round 4 carried one per packet).
    constexpr bool BUF = C == 4 && (VT == 0 || VT == 4);
    auto stream_of = [&](uint32_t q) __attribute__((always_inline)) -> const uint8_t * {
        return B.io[qx(q)].packets;
    };
    const size_t part_off = (size_t)p0 * P0.packet_bytes;
    const uint32_t part_bytes = np * P0.packet_bytes;
    const uint8_t *pk_a = stream_of(0u) + part_off;  // first byte of this wave's partition in the stream copy of the query being requested
    __amdgpu_buffer_rsrc_t rsrc = stream_resource(pk_a, part_bytes);
    LaneOffsets lo{0u, 0u};
    if constexpr (BUF) lo = lane_offsets<C, VT>(lane);
    uint32_t qa = 0u, req_left = np, req_off = 0u, req_step = P0.packet_bytes;
#define TKSPMV_REQUEST(dst)                                                                                           \
    do {                                                                                                              \
        if (req_left == 0u) { /* the previous request was the last of its query */                                     \
            ++qa;                                                                                                     \
            if (qa == nq) { /* the phase's last packet, again and again */                                             \
                req_off -= req_step;                                                                                  \
                req_step = 0u;                                                                                        \
                req_left = 0x7FFFFFFFu;                                                                               \
            } else {                                                                                                  \
                req_left = np;                                                                                        \
                req_off = 0u;                                                                                         \
                pk_a = stream_of(qa) + part_off;                                                                      \
                if constexpr (BUF) rsrc = stream_resource(pk_a, part_bytes);                                          \
            }                                                                                                         \
        }                                                                                                             \
        if constexpr (BUF) load_packet_buf<C, VT>(rsrc, req_off, lo, dst);                                            \
        else load_packet<C, VT>(pk_a + req_off, lane, dst);                                                           \
        --req_left;                                                                                                   \
        req_off += req_step;                                                                                          \
    } while (0)
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) TKSPMV_REQUEST(buf[u]);

    uint32_t qc = 0u, jc = 0u;  // query / packet being reduced
    const bool rows_in_lanes = LOCAL && np <= 64u;
    const uint32_t row_base_v = (rows_in_lanes && lane < np) ? P0.pkt_row[p0 + lane] : 0u;
    float carry = 0.0f, min_units = 0.0f;
    float top1 = 0.0f, top2 = 0.0f;  // local thresholds: the two largest packet maxima of this wave in the current query
    uint32_t wcnt = 0u;
    uint32_t pace = 0u;  // this query's pause per packet, units of pace_quads x 128 cycles (the server: from the workgroup's rank in the previous query)
    const uint32_t wave_entry_fp = (LOCAL && B.pace_period != 0u) ? ((uint32_t)__builtin_amdgcn_s_memrealtime() << 8) : 0u;
    const uint32_t period_fp = (LOCAL && B.pace_period != 0u) ? B.pace_period + (B.pace_adapt ? scalar_load(B.pace_adapt) : 0u) : 0u;
    const uint32_t tpkt_fp = (period_fp != 0u && np != 0u) ? (uint32_t)((float)period_fp / (float)np) : 0u;  // a packet's slot on the timetable
    uint32_t sched_fp = 0u;  // when the packet being reduced is due (ticks << 8, low 32 bits)
    uint32_t pace_rank = 0u;  // (timetable: the pause by rank, which takes over while the wave is more than half a query behind)
    bool behind = false;
    bool waited = false;  // this wave has used its bounded wait for a threshold in the current query (long partitions)
    const bool long_partition = np * (uint32_t)(C / 4) >= 28u;  // ~14 rows finish per 256 entries: > 1.5 lists per query
    uint32_t *mp = L.misc[0];
    uint32_t xbase = 0u;
    StreamParams P = P0;

    for (;;) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            const Pkt<C, VT> &cur = buf[u];
            TKSPMV_REQUEST(buf[(u + NBUF - 1) % NBUF]);
            if (jc == 0u) {  // a new query starts: its x must have been staged
                carry = 0.0f;  // (a partition starts on a row boundary)
                mp = L.misc[qc & 1u];
                xbase = lds_addr_of(L.u.w.x[qc & 1u]);
                for (;;) {
                    const uint32_t xr_ = lds_load(&mp[MISC_XREADY]);
                    if (xr_ == qc + 1u) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                asm volatile("" ::: "memory");
                min_units = __uint_as_float(lds_load(&mp[MISC_MINU]));
                if (trw && lane == 0 && TRSLOT(qc) < 3u) trw[1 + TRSLOT(qc)] = __builtin_amdgcn_s_memrealtime();
                if (!local) {  // (local mode appends nothing to global memory)
                    P.ovf_cand = B.ovf_list(list_of(qc));
                    P.ovf_count = B.ovf_list_count(list_of(qc));
                    P.ovf_gate_lds = lds_addr_of(&L.epoch_now[list_of(qc)]);
                    P.ovf_need = ovf_need(qc);
                }
                wcnt = 0u;
                top1 = top2 = -__builtin_huge_valf();
                waited = false;
                // The timetable starts where the wave ENTERED the kernel, not where the first x became ready 3-5 us later: the wave is behind
                // it from its first packet and runs unpaused until it has caught up (a query or two), and the launch -- which ends a
                // fixed number of periods behind the timetable's start -- ends that much earlier: 16.5-16.7 against 16.8-16.9 us per
                // query sustained, a launch of 20 queries 16.8-16.9 against 17.1 on one box (TKSPMV_ORIGIN_OFFSET ticks: 0; +200 as before,
                // -300 / -600 measured too).
                if (tpkt_fp != 0u && qc == 0u) sched_fp = wave_entry_fp + (uint32_t)((int32_t)TKSPMV_ORIGIN_OFFSET * 256);
                if (tpkt_fp != 0u && (np & (uint32_t)(TKSPMV_CLOCK_STRIDE - 1)) != 0u)  // (the query's last look covers fewer packets than it books)
                    sched_fp -= tpkt_fp * ((uint32_t)TKSPMV_CLOCK_STRIDE - (np & (uint32_t)(TKSPMV_CLOCK_STRIDE - 1)));
                if (pace_q != 0u) {
                    const uint32_t pw = __builtin_amdgcn_readfirstlane(lds_load(&L.pace));
                    pace = pace_rank = pw & 255u;
                    if (pw & 256u) __builtin_amdgcn_s_setprio(2);
                    else __builtin_amdgcn_s_setprio(1);
                } else {
#if TKSPMV_ALTERNATE_PRIO
                    // The two workgroups of a CU do not share it evenly at equal priority: the older one wins the arbitration
                    // (traced over a 32-query launch: 17.2 against 21.5 us per query), runs ahead, finishes early and leaves
                    // the CU half empty while the launch waits for the slower half. They take turns instead, query by query:
                    // 18.7 against 19.8-20.3 us per query on one box (tools/ab_variants.sh; turns of 8, 12 or 32 packets: the same; of 2
                    // or 4: less). Partitions twice as long (2M rows, 512 x 40) neither gain nor lose.
                    if (((qc ^ (bid >= n_wg / 2u ? 1u : 0u)) & 1u) != 0u) __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);
                    else __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO - 1);
#endif
                }
            }
            // a workgroup ahead of the field yields: fewer requests from it, more bandwidth for the XCDs that lag
            // (the pause's bits, one s_sleep each -- units of 128 cycles: a loop of s_sleep(2) spent four scalar instructions per unit,
            //  a quarter of the kernel's scalar instructions with six eighths of the field pausing)
            // (the clock is asked for here and looked at behind the packet's arithmetic: a wave that is behind its timetable -- the one the
            //  launch waits for -- must not stand still for the answer)
            const bool look = tpkt_fp != 0u && (jc & (uint32_t)(TKSPMV_CLOCK_STRIDE - 1)) == 0u;
            uint32_t clk_now = 0u;
            if (look) clk_now = (uint32_t)__builtin_amdgcn_s_memrealtime();
            const uint32_t tau_bits = lds_load(&mp[MISC_TAU]);
            const float tau = __uint_as_float(tau_bits);
            const Reduced<C> Rd = reduce_packet<C, QM>(cur, carry, xbase, P0.fixed_mask);
            const float trig = trigger_of<C, INT>(Rd);
            if (tpkt_fp != 0u) {
                // (the clock is read every TKSPMV_CLOCK_STRIDE packets and what the wave is ahead by slept off in one go: half the scalar
                //  instructions of a look at every packet; requests in bursts of four packets or more measure slower)
                pace = behind ? pace_rank : 0u;
                if (look) {
                    sched_fp += tpkt_fp * (uint32_t)TKSPMV_CLOCK_STRIDE;  // (a query of an odd number of packets: evened out where the query starts)
                    const int32_t ahead = (int32_t)(sched_fp - (clk_now << 8));  // ticks << 8
                    // The wave sleeps in steps of 512 cycles (21 ticks at 2.4 GHz; what is left over the next look sees: the timetable is
                    // absolute, nothing adds up) -- one short loop instead of a test per bit of the count.
                    // (never longer than one period per look: a wave cannot be that far ahead of a timetable it follows -- whatever says
                    //  so, a clock that wrapped, a period of another matrix, costs a bounded pause and not a launch that stands still)
                    if (!behind) {
                        const int32_t cap = (int32_t)B.pace_period;
#pragma unroll 1
                        for (int32_t z = ahead < cap ? ahead : cap; z > (int32_t)(11u << 8); z -= (int32_t)(21u << 8)) __builtin_amdgcn_s_sleep(8);
                        if (ahead > 2 * cap) sched_fp = clk_now << 8;
                    }
                    // A timetable nobody can keep (the GPU streams slower than when the period was measured: a change of power state,
                    // the first milliseconds after an idle period) would leave the field unpaced: a wave more than half a query behind
                    // (looked at where a query starts) paces by its workgroup's rank as if there were no timetable, and its debt stops
                    // growing at one query.
                    if (jc == 0u) {
                        behind = ahead < -(int32_t)(B.pace_period >> 1);
                        if (ahead < -(int32_t)B.pace_period) sched_fp -= (uint32_t)(ahead + (int32_t)B.pace_period);
                        if (behind) pace = pace_rank;
                        // (the launch's last query: a wave a quarter of a period behind says so -- BatchParams::pace_adapt)
                        if (B.pace_adapt && qc + 1u == nq && nq >= 8u && ahead < -(int32_t)(B.pace_period >> 2) && lane == 0)
                            (void)__hip_atomic_fetch_add(B.pace_adapt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (pace != 0u) {
                if (pace & 1u) __builtin_amdgcn_s_sleep(2);
                if (pace & 2u) __builtin_amdgcn_s_sleep(4);
                if (pace & 4u) __builtin_amdgcn_s_sleep(8);
                if (pace & 8u) __builtin_amdgcn_s_sleep(16);
                if (pace & 16u) __builtin_amdgcn_s_sleep(32);
                if (pace & 32u) __builtin_amdgcn_s_sleep(64);
#pragma unroll 1
                for (uint32_t z = pace >> 6; z != 0u; --z) __builtin_amdgcn_s_sleep(127);
            }
            if (TKSPMV_LADDER < 2) {  // (timing-only build: the trigger is formed and kept alive, never taken)
                top1 = max2(top1, trig);
            } else if (__any(trig >= tau)) {
                float tau_now = tau;
                // Long partitions only (more rows per wave and query than its list holds: from ~1.5M rows on 256 CUs). A
                // wave that runs ahead of its workgroup's exchange has no threshold yet: every row passes, and once the
                // list is nearly full the rest would pour into the query's overflow list. It is ahead of the others anyway:
                // it waits for the threshold instead, bounded, once per query (2M rows: 40.4 against 42.1 us per query, 3M:
                // 58.5 against 60.7; 10M rows: 191 against 206). On shorter partitions the list holds a whole query's rows
                // and the wait only costs the overlap of consecutive queries, hence the condition.
                if (long_partition && wcnt + 2u * 64u > WAVE_CAP && tau_bits == __float_as_uint(min_units) && P0.tau_possible && !waited) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (lds_load(&mp[MISC_TAU]) == __float_as_uint(min_units) && __builtin_amdgcn_s_memrealtime() - t0 < BATCH_TAU_WAIT)
                        __builtin_amdgcn_s_sleep(4);
                    waited = true;
                    tau_now = __uint_as_float(lds_load(&mp[MISC_TAU]));
                }
                if (tau_now == tau || __any(trig >= tau_now)) {
                    const RowSums<C> R = expand<C, INT>(Rd, packet_flags<C, QM>(cur));
                    // (row of the first row end of this packet: lane j holds packet j's since the launch began -- a partition of up to 64
                    //  packets; the load from the side table costs the candidate path a trip through the scalar cache otherwise)
                    const uint32_t rb_cur = rows_in_lanes ? (uint32_t)__builtin_amdgcn_readlane((int)row_base_v, (int)jc) : scalar_load(P0.pkt_row + p0 + jc);
                    // (local: what does not fit the list is dropped under a recorded bound, not appended to global memory)
                    const float wm = offer_candidates<C, QM, WAVE_CAP>(P, R, rb_cur, tau_now, lane, grp_local, publishes, wcand, wcnt, mp, local);
                    if (local && wm > top2 && wm >= min_units) {
                        // (a wave with a single packet has no second maximum: it stands for one row)
                        top2 = wm > top1 ? top1 : wm;
                        top1 = wm > top1 ? wm : top1;
                        const float pub = (np >= 2u && !local_top1) ? top2 : top1;
                        if (pub >= min_units) {
                            if (lane == 0)  // single writer: this wave's word
                                __hip_atomic_store(&mp[MISC_GRPMAX + wave], order_key(pub), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            // The workgroup's threshold: the smallest word, once every streaming wave has one (lane w looks at wave
                            // w's; this wave's LDS operations execute in order). Whoever raises it records it (MISC_TAUKEY, an
                            // atomic max: the largest threshold ever formed) and passes it on; a smaller value landing last in
                            // MISC_TAU is still a threshold.
                            const uint32_t key = wave_has ? lds_load(&mp[MISC_GRPMAX + (lane & 7u)]) : 0xFFFFFFFFu;
                            if (__ballot(key == 0u) == 0ull) {
                                const uint32_t kmin = wave_min_u32(key);
                                if (lane == 0) {
                                    const uint32_t old = __hip_atomic_fetch_max(&mp[MISC_TAUKEY], kmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    if (kmin > old)
                                        __hip_atomic_store(&mp[MISC_TAU], __float_as_uint(key_to_float(kmin)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                        }
                    }
                }
            }
            if (jc + 1u == np) {  // the query ends for this wave
                if ((local || (P0.n_sets != 0u && P0.tau_possible)) && wcnt != 0u) {
                    // A wave that runs ahead of the others gets here before any threshold exists for this query; flushing
                    // now would dump every row it has seen to global memory. Give the exchange a moment -- bounded: after
                    // BATCH_TAU_WAIT the wave goes on without one, so progress never depends on other workgroups being
                    // resident. (On a small matrix every wave is in that position: 100+ us per query without this wait.)
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (lds_load(&mp[MISC_TAU]) == __float_as_uint(min_units) &&
                           __builtin_amdgcn_s_memrealtime() - t0 < (local ? LOCAL_TAU_WAIT : BATCH_TAU_WAIT))
                        __builtin_amdgcn_s_sleep(4);
                    if (DBG && P0.dbg && lane == 0) {  // TKSPMV_STATS=1
                        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
                        if (dt > 50ull) {
                            atomicAdd(&mp[MISC_DBG_WAIT_TICKS], (uint32_t)dt);
                            atomicAdd(&mp[MISC_DBG_WAITS], 1u);
                        }
                    }
                }
                if (wcnt != 0u) {
                    const float tau3 = __uint_as_float(lds_load(&mp[MISC_TAU]));
                    if (local) {
                        // the wave's best STG_N survivors are staged; more are dropped under a recorded bound
                        stage_wave_local<WAVE_CAP, true>(wcand, wcnt, tau3, lane, &L.stg[qc & 1u][wave][0], &L.stg_cnt[qc & 1u][wave], mp);
                    } else {
                        ListScan<WAVE_CAP / 64u> LS;
                        const uint32_t surv = scan_list<WAVE_CAP / 64u>(wcand, wcnt, tau3, lane, LS);
                        uint32_t gbase = 0u;
                        if (surv > STG_N) {  // rare: more survivors than the staging area holds go to the overflow list directly
                            if (P.ovf_gate_lds != 0u) ovf_wait(qc);
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
                    }
                }
                if (TKSPMV_LADDER < 2 && top1 == 12345.0f) L.ck[lane] = __float_as_uint(top1);  // (keeps rung 1's arithmetic alive)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&mp[MISC_DONE], 1u);
                if (trw && lane == 0 && TRSLOT(qc) < 3u) trw[4 + TRSLOT(qc)] = __builtin_amdgcn_s_memrealtime();
                ++qc;
                if (qc == nq) return;
                jc = 0u;
            } else {
                ++jc;
            }
        }
    }
#undef TKSPMV_REQUEST
#undef TRSLOT
}

// The kernel's arguments as ONE block.
struct BatchArgs {
    StreamParams P;
    SelectParams S;
    BatchParams B;
};

// Two kernels, one per mode (round 4 first had both modes, both selections and an in-launch repair phase in ONE kernel: the
// streaming loop, which has no register to spare -- 80: two 576-thread workgroups per CU --, paid for the other paths' scalar
// state with reloads from scratch memory, twice the time per query; moved out of line, the rare paths still cost every launch
// its scratch set-up and 3-8 % -- measured A/B on one box, tools/steps_probe.py):
//   LOCAL = true : the checked local thresholds. Streams, delivers the workgroups' records, select_local checks them and adds to
//                  the launch's verdict. Behind a closed gate it does nothing but report every query as failed.
//   LOCAL = false: the device-wide exchange, exact on its own. B.repair = 0: the launch's n_q queries (engines that do not use
//                  local thresholds); B.repair = 1: the queries the verdict of the LOCAL launch before it names -- none, almost
//                  always: every workgroup reads one word and leaves -- and the gate's bookkeeping.
template <int C, int XCOLS, int QM, bool DBG = false, bool LOCAL = false>
__global__ void __launch_bounds__(576, 6) batch_kernel(const BatchArgs A) {
    StreamParams P0 = A.P;
    const SelectParams &SP0 = A.S;
    const BatchParams &B = A.B;
    if (!DBG) {
        P0.trace = nullptr;
        P0.dbg = nullptr;
        P0.stamps = nullptr;
    }
    __shared__ BatchLds<XCOLS, C> L;
    // (reduce_packet forms LDS addresses of x as (word & 0xFFC) | base: x must sit on a 4 KiB boundary -- this object is the
    //  kernel's ONLY __shared__ block, so it starts at LDS address 0, and x is its first member)
    using LdsBlock = BatchLds<XCOLS, C>;
    static_assert(offsetof(LdsBlock, u) == 0 && offsetof(decltype(LdsBlock::u), w) == 0 && offsetof(decltype(LdsBlock::u.w), x) == 0, "x must be the first member of the kernel's LDS block");
    const uint32_t tid = threadIdx.x;
    if (LOCAL) {
        if (blockIdx.x == 0u && tid == 0u && B.verdict_next) __hip_atomic_store(B.verdict_next, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (B.prior_block && B.prior_block[B.gate_parity ? 7 : 4] != 0u) {
            // the gate is closed: nothing is streamed here; every query goes through the exact launch that follows
            if (blockIdx.x == 0u && tid == 0u) {
                const unsigned long long all = (unsigned long long)B.n_q | ((B.n_q >= 32u ? 0xFFFFFFFFull : ((1ull << B.n_q) - 1ull)) << 32);
                __hip_atomic_store(B.verdict, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (B.verdict_host) __hip_atomic_store(B.verdict_host, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        batch_phase<C, XCOLS, QM, DBG, true>(P0, SP0, B, false, L);
        return;
    }
    if (B.repair != 0u) {
        // ---- which queries of the LOCAL launch failed their check? (that launch is over: the word is complete) ------------------
        // (repair = 2: a late repair from the host's side of the verdict -- the device word has been reused since; the gate's
        //  bookkeeping stays with the in-stream launches that follow an observed failure)
        const bool late = B.repair == 2u;
        const unsigned long long v = late ? 0ull : __hip_atomic_load(B.verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t mask = late ? B.repair_mask : (uint32_t)(v >> 32);
        const bool gate_was_closed = !late && B.prior_block && B.prior_block[B.gate_parity ? 7 : 4] != 0u;
        if (tid < 64u) {
            const bool f = tid < B.n_q && ((mask >> tid) & 1u) != 0u;
            const uint64_t bm = __ballot(f);
            if (f) L.rq[__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = tid;
            if (tid == 0) L.rq[BATCH_MAX] = (uint32_t)__popcll(bm);
        }
        __syncthreads();
        // (behind a closed gate the queries did not fail, they were never tried: the closure counts down)
        if (!late && blockIdx.x == 0u && tid == 0u && B.prior_block) gate_update(B, gate_was_closed ? 0u : L.rq[BATCH_MAX]);
        if (L.rq[BATCH_MAX] == 0u) return;
    }
    batch_phase<C, XCOLS, QM, DBG, false>(P0, SP0, B, B.repair != 0u, L);
}

}  // namespace tkspmv
