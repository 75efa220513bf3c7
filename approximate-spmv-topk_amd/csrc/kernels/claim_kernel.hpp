// kernels/claim_kernel.hpp -- claim_kernel: up to 32 queries per launch like batch_kernel, with the matrix DEALT OUT dynamically (round 3; the headline path).
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include "batch_kernel.hpp"

namespace tkspmv {

// ------------------------------------------------------------------------------------------------------------
// Why. batch_kernel gives every streaming wave ONE partition of the matrix for the whole launch. Traced (tools/batch_trace.py,
// tools/claim_probe.py): when every wave asks for all the memory system can give, the eight XCDs do not get equal shares --
// with the candidate path switched off the median wave streams a query in 11.5 us, the workgroups of four XCDs reach query 16
// after 216 us and those of the other four after 340 us; even a pure read kernel with static partitions runs its last 16 %
// with fewer than half of its waves. A fixed share per wave turns that into idle bandwidth: the launch ends with its slowest
// workgroup. (With the candidate path on, the extra work of the cold phase happens to act as a governor -- everybody
// equally slow.)
//
// How. The matrix is cut into SETS of 8 wave partitions (one per streaming wave of a workgroup; ~10 packets each instead of
// 19) and a launch becomes a list of ITEMS (query q, set s). A workgroup's server wave CLAIMS items one ahead of its
// streaming waves from 8 sharded counters (shard = s % 8; a workgroup starts with the shard of its XCD and moves on when
// that one is exhausted: no counter sees more than a few claims per microsecond, and the stealing at the end of the launch
// is what keeps the tail short); a fast workgroup simply claims more items. To the rest of the kernel an ASSIGNMENT (one
// claimed item) is what a query is to batch_kernel: the server stages x(q) into one of two LDS buffers while the waves are
// still in the previous assignment, serves the threshold exchange of q, and finalises the assignment when its 8 waves have
// counted themselves out -- survivors to q's candidate list, then one add to q's DONE counter. The selector workgroup
// selects q when all of q's sets are done. Items of a shard are claimed in order (q major), so a workgroup sees
// non-decreasing q until it starts stealing; nothing depends on that.
//
// What it also buys: an assignment that is claimed after its query's threshold has formed starts WITH that threshold (the
// server reads the query's threshold word when it stages the assignment), so only the first assignments of a query go
// through a cold phase; and the launch's tail is one assignment long, not one query.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t CLAIM_SHARDS = 8;
#ifndef TKSPMV_CLAIM_RING
#define TKSPMV_CLAIM_RING 4
#endif
constexpr uint32_t CLAIM_RING = TKSPMV_CLAIM_RING;  // assignments a workgroup's server may stage ahead of their finalisation (LDS slots; a power of two)
static_assert((CLAIM_RING & (CLAIM_RING - 1u)) == 0u && CLAIM_RING >= 2u, "CLAIM_RING must be a power of two");
constexpr uint32_t CLAIM_TERMINATE = 0xFFFFFFFFu;

struct ClaimParams : SetAddr {
    uint32_t n_q;
    uint32_t n_sets;       // sets per query; set s = wave partitions [8 s, 8 s + 8)
    uint32_t *claim;       // [CLAIM_SHARDS] item counters of THIS launch, 32 words apart, zero at launch
    uint32_t *claim_other; // the counters of the other launch parity: the selector zeroes them for the next launch
    uint32_t *done;        // [BATCH_MAX][CLAIM_SHARDS] finished sets per query and shard, 32 words apart; the selector resets them
    BatchIO io[BATCH_MAX];
};

template <int XCOLS>
struct ClaimLds {
    union {
        struct {
            float x[CLAIM_RING][XCOLS];              // query vector of every staged assignment
            uint2 cand[ListGeom<XCOLS>::CAND_CAP];   // private candidate lists of the streaming waves
        } w;
        SelectShared sel;  // selector workgroup only
    } u;
    uint32_t misc[CLAIM_RING][MISC_WORDS];          // per assignment slot (assignment % CLAIM_RING)
    unsigned long long stg[CLAIM_RING][8][STG_N];   // survivors staged by the streaming waves
    uint32_t stg_cnt[CLAIM_RING][8];
    struct Asg {
        uint32_t q, shard, n_active, pad;
        uint32_t p0[8], np[8];  // first packet / packets of the partition of each streaming wave
    } asg[CLAIM_RING];
};

template <int XCOLS, int QM>
__global__ void __launch_bounds__(576, 6) claim_kernel(const StreamParams P0, const SelectParams SP0, const ClaimParams B) {
    constexpr int C = 4;
    constexpr int VT = value_type_of(QM);
    constexpr int NBUF = 3;
    constexpr bool INT = int_sums<QM>();
    constexpr uint32_t WAVE_CAP = ListGeom<XCOLS>::WAVE_CAP;
    static_assert(QM == 0 || QM == 7, "the claim kernel is built for the fp32 streams");
    __shared__ ClaimLds<XCOLS> L;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nwaves = (blockDim.x >> 6) - 1u;  // streaming waves (8)
    const bool is_server = (wave == nwaves);
    const uint32_t nq = B.n_q;
    auto sets_of_shard = [&](uint32_t h) -> uint32_t { return h < B.n_sets ? (B.n_sets - h + CLAIM_SHARDS - 1u) / CLAIM_SHARDS : 0u; };

    if (blockIdx.x == 0u) {
        // ---- selector workgroup: query q is complete when all of its sets are done -----------------------------------
        if (tid < CLAIM_SHARDS) B.claim_other[32u * tid] = 0u;  // (the launch after this one uses them; nobody else touches them now)
        for (uint32_t q = 0; q < nq; ++q) {
            if (wave == 0u) {
                // lanes 0..7 poll one shard each (relaxed agent-scope loads), the wave adds them up
                for (;;) {
                    uint32_t v = 0u;
                    if (lane < CLAIM_SHARDS) v = __hip_atomic_load(&B.done[(q * CLAIM_SHARDS + lane) * 32u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int d = 4; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor((int)v, d);
                    if (__builtin_amdgcn_readfirstlane(v) >= B.n_sets) break;
                    __builtin_amdgcn_s_sleep(16);
                }
                if (lane < CLAIM_SHARDS) __hip_atomic_store(&B.done[(q * CLAIM_SHARDS + lane) * 32u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            SelectParams S = SP0;
            S.n_wg = 0u;  // no per-workgroup slots: every survivor is in the query's candidate list
            S.ovf_cand = B.ovf_cand(q);
            S.ovf_count = B.ovf_count(q);
            S.gmax = B.gmax(q);
            S.tau_g = B.tau_g(q);
            S.scratch = B.scratch;
            S.unit_inv_in = nullptr;
            S.out_idx = B.io[q].out_idx;
            S.out_val = B.io[q].out_val;
            select_body(S, tid, blockDim.x, L.u.sel);
            __syncthreads();
        }
        return;
    }
    const uint32_t bid = blockIdx.x - 1u;
    if (tid < CLAIM_RING * MISC_WORDS) (&L.misc[0][0])[tid] = 0u;
    if (tid < CLAIM_RING * 8u) (&L.stg_cnt[0][0])[tid] = 0u;
    __syncthreads();
    const uint32_t grp_local = is_server ? 0u : wave * P0.gpw / nwaves;
    const bool publishes = (bid * P0.gpw + grp_local) < P0.n_groups_pub;
    const bool reducer = bid < P0.n_reducers;
    const float min_units = P0.min_score;  // fp32 scores: units of 1

    if (is_server) {
        // ---- server wave: claims, x staging, threshold exchange, finalisation ---------------------------------------
        if (reducer) __builtin_amdgcn_s_setprio(3);
        uint32_t staged = 0u, tail = 0u;  // assignments staged / finalised
        uint32_t shard_try = 0u;          // shards found exhausted so far (CLAIM_SHARDS: nothing left to claim)
        const uint32_t home = (uint32_t)__builtin_amdgcn_s_getreg(63508) & 7u;  // HW_REG_XCC_ID (speed only: any value is correct)
        for (;;) {
            if (shard_try < CLAIM_SHARDS && staged - tail < CLAIM_RING) {
                // claim the next item: the home shard first, then the others in turn
                uint32_t q = 0u, set = 0u, shard = 0u;
                bool got = false;
                while (shard_try < CLAIM_SHARDS) {
                    shard = (home + shard_try) & (CLAIM_SHARDS - 1u);
                    const uint32_t n_h = sets_of_shard(shard);
                    uint32_t i = 0u;
                    if (n_h != 0u) {
                        if (lane == 0) i = __hip_atomic_fetch_add(&B.claim[32u * shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        i = __builtin_amdgcn_readfirstlane(i);
                    }
                    if (n_h != 0u && i < nq * n_h) {
                        q = i / n_h;
                        set = (i - q * n_h) * CLAIM_SHARDS + shard;
                        got = true;
                        break;
                    }
                    ++shard_try;
                }
                if (got) {
                    const uint32_t par = staged & (CLAIM_RING - 1u);
                    typename ClaimLds<XCOLS>::Asg &A = L.asg[par];
                    // the set's partitions, x of its query, the query's threshold as it stands: one round trip
                    uint32_t my_p0 = 0u, my_np = 0u;
                    const uint32_t pidx = set * 8u + lane;
                    if (lane < 8u && pidx < P0.n_parts) {
                        my_p0 = P0.part_first[pidx];
                        my_np = P0.part_count[pidx];
                    }
                    const uint32_t tau_key = P0.n_sets != 0u ? __hip_atomic_load(B.tau_g(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    const float *xg = B.io[q].x;
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
                        for (int u = 0; u < 16; ++u) xl[b0 + lane + 64u * (uint32_t)u] = r[u];
                    }
                    uint32_t *mp = L.misc[par];
                    if (lane < (uint32_t)MISC_WORDS && lane != (uint32_t)MISC_XREADY) mp[lane] = 0u;
                    if (lane < 8u) {
                        L.stg_cnt[par][lane] = 0u;
                        A.p0[lane] = my_p0;
                        A.np[lane] = my_np;
                    }
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    float tau_init = min_units;
                    if (tau_key != 0u && key_to_float(tau_key) > tau_init) tau_init = key_to_float(tau_key);
                    if (lane == 0) {
                        A.q = q;
                        A.shard = shard;
                        A.n_active = nwaves;  // (every wave counts itself out of every assignment)
                        mp[MISC_TAU] = __float_as_uint(tau_init);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(&mp[MISC_XREADY], staged + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    ++staged;
                }
            }
            // Threshold exchange of the assignment this workgroup's waves are streaming: the oldest unfinished one until
            // half of its waves have left it, then the next.
            if (P0.n_sets != 0u && tail < staged) {
                uint32_t ha = tail;
                if (tail + 1u < staged &&
                    2u * __builtin_amdgcn_readfirstlane(lds_load(&L.misc[tail & (CLAIM_RING - 1u)][MISC_DONE])) >= L.asg[tail & (CLAIM_RING - 1u)].n_active)
                    ha = tail + 1u;
                const uint32_t q = L.asg[ha & (CLAIM_RING - 1u)].q;
                StreamParams P = P0;
                P.gmax = B.gmax(q);
                P.tau_g = B.tau_g(q);
                uint32_t *mp = L.misc[ha & (CLAIM_RING - 1u)];
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
                }
            }
            // finalise the oldest assignment once its streaming waves have all counted themselves out
            if (tail < staged) {
                const uint32_t tp = tail & (CLAIM_RING - 1u);
                uint32_t *mp = L.misc[tp];
                const typename ClaimLds<XCOLS>::Asg &A = L.asg[tp];
                if (__builtin_amdgcn_readfirstlane(lds_load(&mp[MISC_DONE])) >= A.n_active) {
                    asm volatile("" ::: "memory");
                    const uint32_t q = A.q;
                    StreamParams P = P0;
                    P.gmax = B.gmax(q);
                    if (P0.n_sets != 0u) publish_group_max(P, bid, lane, mp);  // complete maxima (fire and forget)
                    // lane l copies entry (l % 8) of wave (l / 8) to the query's candidate list
                    const uint32_t w = lane >> 3, e = lane & 7u;
                    const bool have = e < L.stg_cnt[tp][w];
                    const unsigned long long v = have ? L.stg[tp][w][e] : 0ull;
                    const uint64_t bm = __ballot(have);
                    if (bm) {
                        uint32_t gbase = 0u;
                        if (lane == 0) gbase = atomicAdd(B.ovf_count(q), (uint32_t)__popcll(bm));
                        gbase = __builtin_amdgcn_readfirstlane(gbase);
                        const uint32_t gp = gbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                        if (have && gp < P0.ovf_cap) st_agent(&B.ovf_cand(q)[gp], v);
                    }
                    // Hand-off (cdna_hip_programming.md Guideline 16): everything above is a write-through (sc1) store; drain
                    // them, then a RELAXED agent-scope add (a release would write back the whole L2 per assignment).
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0)
                        (void)__hip_atomic_fetch_add(&B.done[(q * CLAIM_SHARDS + A.shard) * 32u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ++tail;
                }
            }
            if (shard_try >= CLAIM_SHARDS && tail == staged) {  // nothing left anywhere: tell the streaming waves and leave
                if (lane < CLAIM_RING) __hip_atomic_store(&L.misc[lane][MISC_XREADY], CLAIM_TERMINATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
            }
            if (reducer) __builtin_amdgcn_s_sleep(TKSPMV_REDUCER_SLEEP);
            else __builtin_amdgcn_s_sleep(8);
        }
    }

    // ---- streaming waves: assignment after assignment, one continuous prefetch ring ----------------------------------
    __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);
    uint2 *wcand = L.u.w.cand + wave * WAVE_CAP;

    Pkt<C, VT> buf[NBUF];
    uint32_t rbs[NBUF];
    // What a ring slot holds: SLOT_BUBBLE = nothing (the next assignment was not staged yet when the request was due: the last
    // packet was requested again, so that the number of loads in flight stays fixed); 0 = a further packet of the segment being
    // reduced; else the FIRST packet of the wave's partition in an assignment: assignment << 16 | packets.
    constexpr uint32_t SLOT_BUBBLE = 0xFFFFFFFFu;
    uint32_t info[NBUF];
    // Request side: the next assignment to look at (ra), packets of the current segment still to request (req_left), real
    // packets requested and not yet reduced (pend). Every wave counts itself out of EVERY assignment (the server recycles an
    // assignment's LDS slot only then); where the wave has no partition -- the last, partial set -- the request side does it on
    // the spot. The request side never waits for the server while the reduction side still owes it something (pend != 0): the
    // server stages assignment a + 2 only after a has been finalised, which may take this very wave's remaining packets.
    uint32_t ra = 0u, req_left = 0u, pend = 0u, seg_first = 0u;
    bool req_done = false;
    const uint8_t *pk_a = B.io[0].packets;
    const uint32_t *row_a = P0.pkt_row;
    auto try_next_segment = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            uint32_t *mpa = L.misc[ra & (CLAIM_RING - 1u)];
            const uint32_t xr_ = lds_load(&mpa[MISC_XREADY]);
            if (xr_ == CLAIM_TERMINATE) {
                req_done = true;
                return false;
            }
            if (xr_ != ra + 1u) return false;  // not staged yet
            asm volatile("" ::: "memory");
            const uint32_t np = __builtin_amdgcn_readfirstlane(lds_load(&L.asg[ra & (CLAIM_RING - 1u)].np[wave]));
            if (np != 0u) {
                const uint32_t p0 = __builtin_amdgcn_readfirstlane(lds_load(&L.asg[ra & (CLAIM_RING - 1u)].p0[wave]));
                const uint32_t q = __builtin_amdgcn_readfirstlane(lds_load(&L.asg[ra & (CLAIM_RING - 1u)].q));
                pk_a = B.io[q].packets + (size_t)p0 * P0.packet_bytes;
                row_a = P0.pkt_row + p0;
                req_left = np;
                seg_first = (ra << 16) | np;
                ++ra;
                return true;
            }
            if (lane == 0) atomicAdd(&mpa[MISC_DONE], 1u);  // no partition of this wave in the set
            ++ra;
        }
    };
#define TKSPMV_REQUEST(dst, rb_dst, info_dst)                                                                         \
    do {                                                                                                              \
        if (req_left == 0u && !req_done) {                                                                            \
            bool ok_ = try_next_segment();                                                                            \
            while (!ok_ && !req_done && pend == 0u) { /* nothing owed: wait for the server */                          \
                __builtin_amdgcn_s_sleep(2);                                                                          \
                ok_ = try_next_segment();                                                                             \
            }                                                                                                         \
        }                                                                                                             \
        load_packet<C, VT>(pk_a, lane, dst);                                                                          \
        rb_dst = scalar_load(row_a);                                                                                              \
        if (req_left != 0u) {                                                                                         \
            info_dst = seg_first;                                                                                     \
            seg_first = 0u;                                                                                           \
            ++pend;                                                                                                   \
            if (--req_left != 0u) {                                                                                   \
                pk_a += P0.packet_bytes;                                                                              \
                ++row_a;                                                                                              \
            }                                                                                                         \
        } else {                                                                                                      \
            info_dst = SLOT_BUBBLE;                                                                                   \
        }                                                                                                             \
    } while (0)
#pragma unroll
    for (int u = 0; u < NBUF - 1; ++u) TKSPMV_REQUEST(buf[u], rbs[u], info[u]);
    rbs[NBUF - 1] = 0u;
    info[NBUF - 1] = SLOT_BUBBLE;
    if (req_done && pend == 0u) return;  // this workgroup never got an item

    uint32_t ac = 0u, jc = 0u, np_c = 0u;  // assignment / packet being reduced / packets of the wave's partition in it
    float carry = 0.0f;
    uint32_t wcnt = 0u;
    uint32_t *mp = L.misc[0];
    uint32_t xbase = 0u;
    StreamParams P = P0;

    for (;;) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            const Pkt<C, VT> &cur = buf[u];
            const uint32_t rb_cur = rbs[u];
            const uint32_t info_cur = info[u];
            TKSPMV_REQUEST(buf[(u + NBUF - 1) % NBUF], rbs[(u + NBUF - 1) % NBUF], info[(u + NBUF - 1) % NBUF]);
            if (info_cur == SLOT_BUBBLE) {
                if (req_done && pend == 0u) return;
                continue;
            }
            if (info_cur != 0u) {  // the first packet of the wave's partition in a new assignment
                ac = info_cur >> 16;
                np_c = info_cur & 0xFFFFu;
                mp = L.misc[ac & (CLAIM_RING - 1u)];
                xbase = lds_addr_of(L.u.w.x[ac & (CLAIM_RING - 1u)]);
                const uint32_t q = __builtin_amdgcn_readfirstlane(lds_load(&L.asg[ac & (CLAIM_RING - 1u)].q));
                P.ovf_cand = B.ovf_cand(q);
                P.ovf_count = B.ovf_count(q);
                carry = 0.0f;  // (a partition starts on a row boundary)
                wcnt = 0u;
                jc = 0u;
            }
            const uint32_t tau_bits = lds_load(&mp[MISC_TAU]);
            const float tau = __uint_as_float(tau_bits);
            const Reduced<C> Rd = reduce_packet<C, QM>(cur, carry, xbase, P0.fixed_mask);
            const float trig = trigger_of<C, INT>(Rd);
            if (__any(trig >= tau)) {
                const RowSums<C> R = expand<C, INT>(Rd, packet_flags<C, QM>(cur));
                offer_candidates<C, QM, WAVE_CAP, false>(P, R, rb_cur, tau, lane, grp_local, publishes, wcand, wcnt, mp);
            }
            --pend;
            if (++jc == np_c) {  // the assignment ends for this wave
                if (wcnt != 0u) {
                    if (P0.n_sets != 0u && P0.tau_possible) {
                        // no threshold yet (the first assignments of a query; a small matrix): give the exchange a moment rather
                        // than dumping every row to global memory -- bounded, progress never depends on other workgroups
                        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                        while (lds_load(&mp[MISC_TAU]) == __float_as_uint(min_units) && __builtin_amdgcn_s_memrealtime() - t0 < BATCH_TAU_WAIT)
                            __builtin_amdgcn_s_sleep(4);
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
                            if (LS.pos[e] < STG_N) L.stg[ac & (CLAIM_RING - 1u)][wave][LS.pos[e]] = v;
                            else if (gbase + LS.pos[e] - STG_N < P0.ovf_cap) st_agent(&P.ovf_cand[gbase + LS.pos[e] - STG_N], v);
                        }
                    }
                    if (surv > STG_N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // those stores precede the DONE add
                    if (lane == 0) L.stg_cnt[ac & (CLAIM_RING - 1u)][wave] = surv < STG_N ? surv : STG_N;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) atomicAdd(&mp[MISC_DONE], 1u);
                if (req_done && pend == 0u) return;
            }
        }
    }
#undef TKSPMV_REQUEST
}

}  // namespace tkspmv
