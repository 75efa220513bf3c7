// kernels/multi_kernel.hpp -- multi_kernel<Q>: Q queries per pass over the wave-sliced ELL stream (tkspmv_enqueue_multi).
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include "batch_kernel.hpp"
#include "../wsell.hpp"

namespace tkspmv {

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
constexpr int MISC_NEED = 28;  // multi-query kernel: a streaming wave of the workgroup has been waiting for a threshold for NEED_AFTER
constexpr unsigned long long NEED_AFTER = 1000;  // x 10 ns
#ifndef TKSPMV_SELL_BYTE_NBUF
#define TKSPMV_SELL_BYTE_NBUF 5
#endif
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
    const uint32_t *part_slice0;  // [n_parts] first slice of every partition
    uint32_t n_sel;               // selector workgroups at the head of the grid = the engine's queries per pass (1, 2, 4 or 8)
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
    S.unit_inv_in = nullptr;
    S.out_idx = io.out_idx;
    S.out_val = io.out_val;
    return S;
}

// The selections still owed to the last group of a sequence: one workgroup per query, each with its own general-path
// scratch. Queries that share a result buffer (the engine-owned pair: "the last query wins") are selected one after the
// other by workgroup 0 instead.
__global__ void __launch_bounds__(SEL_THREADS) select_group_kernel(const SelectParams SP0, const SetAddr A, const MultiGroup G,
                                                                   uint32_t serial) {
    __shared__ SelectShared S;
    const uint32_t q0 = serial ? 0u : blockIdx.x, q1 = serial ? G.n_q : blockIdx.x + 1u;
    for (uint32_t q = q0; q < q1 && q < G.n_q; ++q) {
        SelectParams P = select_params_of_set(SP0, A, G.set0 + q, G.io[q]);
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

// The value of entry j of a lane's chunk as a float: fp32 chunks hold it; byte chunks (VT = 1, Q1.7 rounded to nearest,
// TKSPMV_Q1_7_F32) convert it with one v_cvt_f32_ubyteN -- x in LDS is pre-scaled by 2^-7, so the product is the same
// fp32 number as (byte / 128) * x.
template <int VT>
__device__ __forceinline__ float chunk_value(const Pkt<4, VT> &p, int j) {
    if (VT == 1 || VT == 5) return ubyte_to_float(p.vq[0], j);
    return p.v[VT == 0 ? j : 0];
}

// VT: 0 = fp32 chunks (1536 B), 1 = byte chunks (768 B; four of them in flight behind the one being reduced instead of two).
template <int Q, int VT = 0>
__global__ void __launch_bounds__(576, Q >= 8 ? 4 : 6) multi_kernel(const StreamParams P0, const SelectParams SP0, const MultiParams M) {
    // (byte chunks: four in flight behind the one being reduced; one query per pass -- BASELINE configs[4] -- seven: 19.0 against 19.6 us
    //  per query, while 4 queries per pass lose 1 % by it and ten or more chunks spill: 89 us)
    constexpr int C = 4, NBUF = (VT == 1 || VT == 5) ? (Q == 1 ? 8 : TKSPMV_SELL_BYTE_NBUF) : 3, DEFER_S = MultiGeom<Q>::HOLD;
    constexpr bool BYTES = VT == 1 || VT == 5;  // VT 5: byte values with 12-bit column words (640-byte chunks, padding slots 1022 / 1023)
    constexpr uint32_t PAD_NEUTRAL = VT == 5 ? 1022u : SELL_PAD_NEUTRAL, PAD_ONE = VT == 5 ? 1023u : SELL_PAD_ONE;
    constexpr uint32_t MULTI_WAVE_CAP = MultiGeom<Q>::WAVE_CAP;
    __shared__ MultiLds<Q> L;
    const uint32_t tid0 = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    if (blockIdx.x < M.n_sel) {
        // selector workgroups: workgroup q selects query q of the previous group (all of them at once: one after the other
        // in ONE workgroup they took longer than the pass they ride in)
        if (blockIdx.x < M.prev.n_q) {
            SelectParams S = select_params_of_set(SP0, M.A, M.prev.set0 + blockIdx.x, M.prev.io[blockIdx.x]);
            select_body(S, tid0, blockDim.x, L.u.sel);
        }
        return;
    }
    const uint32_t bid = blockIdx.x - M.n_sel, n_wg = gridDim.x - M.n_sel;
    const uint32_t nwaves = (blockDim.x >> 6) - 1u;  // streaming waves
    const bool is_server = (wave == nwaves);
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
    // A launch makes ceil(M.cur.n_q / Q) PASSES over the matrix, Q queries each (the last one maybe fewer): pass p serves queries
    // p Q .. p Q + Q - 1 of the group, exchange-state sets set0 + p Q + q, M.cur.io[p Q + q] (round 4, for BASELINE configs[4] at one
    // query per pass: the launch's ramp, its partition lookup and its tail are paid once per eight queries; the next pass's first
    // chunks and its x are requested before the barrier that ends the current one).
    // (4 and 8 queries per pass: ONE pass per launch, known at compile time -- their register budgets have no room for a loop
    //  around the pass: 16 and 57 spilled registers when tried)
    const uint32_t n_pass = Q >= 4 ? 1u : (M.cur.n_q + (uint32_t)Q - 1u) / (uint32_t)Q;
    auto nq_of = [&](uint32_t pass) __attribute__((always_inline)) -> uint32_t {
        const uint32_t left = M.cur.n_q - pass * (uint32_t)Q;
        return left < (uint32_t)Q ? left : (uint32_t)Q;
    };
    Pkt<C, VT> buf[NBUF];
    constexpr bool IL = Q >= 8;  // interleaved x: measured faster for 8 queries (5.99 against 7.09 us per query), slower for 4 (9.43 against 7.59)
    auto x_slot = [&](uint32_t q, uint32_t col) __attribute__((always_inline)) -> float & {
        return L.u.w.x[IL ? col * (uint32_t)Q + q : q * (SELL_XCOLS + 8u) + col];
    };
    // x of the pass's queries: ALL loads first (Q x 2 per thread at 512+ threads: 2 x blockDim covers the 1024 columns), the LDS
    // writes later: written as "load, convert, store" per element the compiler waited for every load before it issued the next
    // one -- 2 Q trips through memory one after the other (round 4 found the same in the batch kernel's staging). Addresses are
    // clamped and values masked, so no load sits in a branch. Queries beyond nq (a partial group): zeros, their sums are never
    // looked at.
    constexpr uint32_t XI = 2u;  // (the host launches this kernel only with 2 x blockDim >= SELL_XCOLS: engine.hip, can_multi)
    float xr[Q][XI];
    auto request = [&](uint32_t pass, uint32_t tid) __attribute__((always_inline)) {  // the pass's first chunks, then its x
        const uint32_t lane = tid & 63u, nq = nq_of(pass);
        const uint8_t *pkp = M.cur.io[pass * (uint32_t)Q].packets + (size_t)p0 * P0.packet_bytes;
#pragma unroll
        for (int u = 0; u < NBUF - 1; ++u) {
            if (np > 0u) {
                const uint32_t iu = ((uint32_t)u < np) ? (uint32_t)u : (np - 1u);
                load_packet<C, VT>(pkp + (size_t)iu * P0.packet_bytes, lane, buf[u]);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)Q; ++q) {
            const float *xg = M.cur.io[pass * (uint32_t)Q + (q < nq ? q : 0u)].x;
#pragma unroll
            for (uint32_t it = 0; it < XI; ++it) {
                const uint32_t i = tid + it * blockDim.x;
                xr[q][it] = xg[i < P0.cols ? i : 0u];
            }
        }
    };
    request(0u, tid0);
    // The head of a pass: the workgroup's words, then x from the registers request() filled.
    auto commit = [&](uint32_t tid, uint32_t nq) __attribute__((always_inline)) {
        for (uint32_t i = tid; i < (uint32_t)Q * MISC_WORDS; i += blockDim.x)
            (&L.misc[0][0])[i] = (i % MISC_WORDS) == (uint32_t)MISC_TAU ? __float_as_uint(min_units) : 0u;
#pragma unroll
        for (uint32_t q = 0; q < (uint32_t)Q; ++q) {
#pragma unroll
            for (uint32_t it = 0; it < XI; ++it) {
                const uint32_t i = tid + it * blockDim.x;
                // (12-bit column words keep the two padding slots INSIDE the 1024 columns: they are thread 0's to write, below)
                if (i < SELL_XCOLS && i != PAD_NEUTRAL && i != PAD_ONE)
                    x_slot(q, i) = (i < P0.cols && q < nq) ? (BYTES ? xr[q][it] * Q17_UNIT : xr[q][it]) : 0.0f;
            }
            if (tid == 0) {
                x_slot(q, PAD_NEUTRAL) = -0.0f;
                x_slot(q, PAD_ONE) = BYTES ? -__builtin_huge_valf() : 1.0f;  // byte chunks: a lane without a row starts with byte 1
            }
        }
    };
    // The server wave and the streaming waves run their OWN loops over the passes (the same two barriers per pass in both): in one
    // loop with a branch inside, the server's registers were allocated across the streaming loop and the kernel spilled 40.
    if (is_server) {
#pragma unroll 1
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            uint32_t tid = tid0;
            asm volatile("" : "+v"(tid));  // (as in the streaming waves' loop below)
            const uint32_t lane = tid & 63u;
            const uint32_t setb = set0 + pass * (uint32_t)Q, nq = nq_of(pass);
            commit(tid, nq);
            __syncthreads();
            // Threshold exchange of all nq queries at once: lane l serves (query l / 8, local group l % 8), so a round costs one
            // store and one load round trip however many queries share the pass (query by query, a round took nq round trips
            // and a threshold needed three rounds -- publish, reduce, fetch -- to reach a workgroup: most of the pass). Each
            // reducer workgroup searches the k-th largest maximum of ONE query per round (~2.5 us).
            const uint32_t q_l = lane >> 3, g_l = lane & 7u;
            const bool q_ok = q_l < nq;
            uint32_t *mp_l = L.misc[q_ok ? q_l : 0u];
            const uint32_t grp = bid * P0.gpw + g_l;
            const bool pub_lane = q_ok && g_l < P0.gpw && grp < P0.n_groups_pub;
            uint32_t *gmax_l = M.A.gmax(setb + (q_ok ? q_l : 0u));
            uint32_t *tau_g_l = M.A.tau_g(setb + (q_ok ? q_l : 0u));
            const uint32_t rq = bid % nq;  // the query this workgroup reduces (if it is a reducer)
            uint32_t *misc0 = L.misc[0];
            auto publish_all = [&]() __attribute__((always_inline)) {
                if (pub_lane) {
                    const uint32_t key = lds_load(&mp_l[MISC_GRPMAX + g_l]);
                    if (key > mp_l[MISC_PUBLISHED + g_l]) {
                        mp_l[MISC_PUBLISHED + g_l] = key;
                        __hip_atomic_store(&gmax_l[grp], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // single writer per slot
                    }
                }
            };
            // Passes are not synchronised across workgroups. The reducers are the first workgroups dispatched, the leaders of the
            // launch; a workgroup behind them enters a pass after they have left it and finds the threshold of their LAST search
            // there -- none at all if fewer than k groups had published by then (measured: whole runs at 50 or 150 us per query on
            // most boxes; every wave ran into its bounded wait at the end of the pass and then judged its rows without a threshold,
            // a million candidates per query for the selection). Two remedies, both below: a reducer keeps searching for the passes
            // it has left; and, as the net under that, a workgroup one of whose waves has WAITED 10 us for a threshold (MISC_NEED)
            // searches it itself from whatever has been published for the pass -- valid as any other, and shared through tau_g.
            // Tried instead: reducers spread over the dispatch order (4, 8, 16, 56 of them: 19.3 / 19.4 / 19.6 / 21.0 us per query
            // where this scheme runs 18.5-18.9 -- a reducer's workgroup streams more slowly, and only the leaders can afford that);
            // every workgroup searching after five rounds (+9 us), or as soon as a wave waits (+3 us).
            uint32_t rounds = 0u;
            for (;;) {
                publish_all();
                uint32_t rq_now = rq;
                bool reduce_now = reducer;
                if (Q <= 2 && !reducer && P0.tau_possible && lds_load(&misc0[MISC_NEED]) != 0u) {  // (4 and 8 per pass: one pass per launch)
                    const uint64_t lack = __ballot(q_ok && g_l == 0u && lds_load(&mp_l[MISC_TAU]) == __float_as_uint(min_units));
                    if (lack != 0ull) {
                        reduce_now = true;
                        rq_now = (uint32_t)__builtin_ctzll(lack) >> 3;  // (lane 8 q asks for query q)
                    }
                }
                if (reduce_now) {
                    StreamParams P = P0;
                    P.gmax = M.A.gmax(setb + rq_now);
                    TauRegs tr_;
                    tau_issue(P, lane, tr_);
                    const float t = tau_from_maxima(P, tr_, min_units);
                    if (lane == 0 && t > min_units)
                        __hip_atomic_fetch_max(M.A.tau_g(setb + rq_now), order_key(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // The reducers are the first workgroups dispatched, the leaders of the launch: the workgroups behind them enter a
                // pass after the reducers have left it, and find the threshold of the reducers' LAST search there -- none at all if
                // fewer than k groups had published by then. So a reducer keeps searching for the passes it has left, one of them
                // per round in turn (the maxima published since are in gmax): every set of the launch is revisited until it ends.
                if (Q <= 2 && reducer && pass != 0u) {  // (every round: on alternate rounds only, 18.9-19.1 against 18.5-18.9 us per query)
                    const uint32_t older = set0 + (rounds % pass) * (uint32_t)Q + rq % (uint32_t)Q;
                    StreamParams P = P0;
                    P.gmax = M.A.gmax(older);
                    TauRegs tr_;
                    tau_issue(P, lane, tr_);
                    const float t = tau_from_maxima(P, tr_, min_units);
                    if (lane == 0 && t > min_units)
                        __hip_atomic_fetch_max(M.A.tau_g(older), order_key(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ++rounds;
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
            if (pass + 1u < n_pass) {
                request(pass + 1u, tid);  // (x only: the server has no partition)
                __syncthreads();
            }
        }
        return;
    }
#pragma unroll 1
    for (uint32_t pass = 0; pass < n_pass; ++pass) {
    // (the thread's index is taken afresh in every pass: whatever depends on it is then computed inside the pass, where it is
    //  needed, instead of ahead of the pass loop and kept in registers through the streaming loop -- measured: 40 spilled registers)
    uint32_t tid = tid0;
    asm volatile("" : "+v"(tid));
    const uint32_t lane = tid & 63u;
    const uint32_t setb = set0 + pass * (uint32_t)Q, nq = nq_of(pass);
    const uint8_t *pk = M.cur.io[pass * (uint32_t)Q].packets + (size_t)p0 * P0.packet_bytes;
    commit(tid, nq);
    __syncthreads();

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
            const Pkt<C, VT> &cur = buf[u];
            {  // unconditional (pointer clamped to the last chunk): a fixed number of younger loads => counted vmcnt
                if (i + (NBUF - 1) < np) pk_ahead += P0.packet_bytes;
                load_packet<C, VT>(pk_ahead, lane, buf[(u + NBUF - 1) % NBUF]);
            }
            // Byte offsets of x[col] for the lane's four entries, and (f0, f1, f2, f3) the words that carry the flag bits in their low two
            // bits. 16-bit words: two dwords of two. 12-bit words: 48 bits of the two dwords loaded, from bit 0 on even lanes and from
            // bit 16 on odd ones -- taken apart straight into offsets (9 instructions; via an intermediate pair of 16-bit words: 15).
            uint32_t off[C], fw[C];
            if (VT == 5) {
                const uint32_t odd16 = (threadIdx.x & 1u) << 4;
                const uint32_t lo = __builtin_amdgcn_alignbit(cur.cw[1], cur.cw[0], odd16), hi = cur.cw[1] >> odd16;
                const uint32_t mid = __builtin_amdgcn_alignbit(hi, lo, 24);
                fw[0] = lo;
                fw[1] = lo >> 12;
                fw[2] = mid;
                fw[3] = mid >> 12;
#pragma unroll
                for (int j = 0; j < C; ++j) off[j] = fw[j] & 0xFFCu;
            } else {
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    fw[j] = (j & 1) ? (cur.cw[j >> 1] >> 16) : cur.cw[j >> 1];
                    off[j] = fw[j] & 0xFFFCu;
                }
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
                    const float vj = chunk_value<VT>(cur, j);
                    const f32x2 vv = {vj, vj};
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
                            acc[q] = __fadd_rn(acc[q], __fmul_rn(chunk_value<VT>(cur, j), *reinterpret_cast<const float *>(xb + off[j])));
                    }
                }
            }
            if (__builtin_amdgcn_readfirstlane(fw[0]) & 1u) {  // last chunk of the slice: 64 rows are complete
                // A row of more than 64 entries spans adjacent lanes (segment index in the flag bits of this chunk, wsell.hpp):
                // its segment sums are added left to right and the score ends up on its last lane; rare.
                const uint32_t depth = (fw[1] & 3u) | ((fw[2] & 3u) << 2) | ((fw[3] & 3u) << 4);
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
                        while (no_tau() && __builtin_amdgcn_s_memrealtime() - t0 < FLUSH_TAU_WAIT) {
                            if (Q <= 2 && lane == 0 && __builtin_amdgcn_s_memrealtime() - t0 > NEED_AFTER)
                                __hip_atomic_store(&L.misc[0][MISC_NEED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __builtin_amdgcn_s_sleep(8);
                        }
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
                            if (__any(acc[q] >= tau))
                                offer_rows<MULTI_WAVE_CAP>(M.A, setb + q, P0.ovf_cap, acc[q], (slice + n_done) * 64u, tau,
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
    // (what follows runs once per pass: its lane-dependent addresses are computed here, not ahead of the streaming loop)
    uint32_t lane_t = lane;
    asm volatile("" : "+v"(lane_t));
    // The held slices. A short partition (small matrix) gets here before any threshold exists: give the exchange a moment,
    // bounded, and only where a threshold can form at all.
    if (np > 0u) {
        if (P0.tau_possible && !gave_up) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (no_tau() && __builtin_amdgcn_s_memrealtime() - t0 < FLUSH_TAU_WAIT) {
                if (Q <= 2 && lane_t == 0 && __builtin_amdgcn_s_memrealtime() - t0 > NEED_AFTER)
                    __hip_atomic_store(&L.misc[0][MISC_NEED], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_s_sleep(4);
            }
        }
#pragma unroll
        for (int d = 0; d < DEFER_S; ++d) {
            if ((uint32_t)d < n_held) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    if ((uint32_t)q < nq) {
                        const float tau = __uint_as_float(lds_load(&L.misc[q][MISC_TAU]));
                        if (__any(held[d][q] >= tau))
                            offer_rows<MULTI_WAVE_CAP>(M.A, setb + q, P0.ovf_cap, held[d][q], (slice + (uint32_t)d) * 64u, tau,
                                                       lane_t, grp_local, publishes, L.u.w.cand[wave][q], wcnt[q], L.misc[q],
                                                       P0.dbg ? P0.dbg + 4 : nullptr);
                    }
                }
            }
        }
    }
    if (lane_t == 0) atomicAdd(&L.misc[0][MISC_DONE], 1u);

    // ---- flush: what still clears the final threshold leaves the wave's lists (first survivor to the wave's slot,
    // further ones to the query's overflow list); complete at the end of the launch, selected by the next launch.
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if ((uint32_t)q < nq) {
            const float tau = __uint_as_float(lds_load(&L.misc[q][MISC_TAU]));
            ListScan<MULTI_WAVE_CAP / 64u> LS;
            const uint32_t surv = scan_list<MULTI_WAVE_CAP / 64u>(L.u.w.cand[wave][q], wcnt[q], tau, lane_t, LS);
            if (surv != 0u) {
                uint32_t gbase = 0u;
                if (surv > 1u) {
                    if (lane_t == 0) gbase = atomicAdd(M.A.ovf_count(setb + q), surv - 1u);
                    gbase = __builtin_amdgcn_readfirstlane(gbase);
                }
                unsigned long long *slot = M.A.wg_cand(setb + q) + (size_t)bid * WG_SLOTS + wave;
                unsigned long long *ovf = M.A.ovf_cand(setb + q);
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
    if (pass + 1u < n_pass) {
        request(pass + 1u, tid);
        __syncthreads();  // every wave is through with this pass's x, lists and words
    }
    }  // (passes)
}

}  // namespace tkspmv
