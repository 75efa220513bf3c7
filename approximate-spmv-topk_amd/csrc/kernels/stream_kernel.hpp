// kernels/stream_kernel.hpp -- stream_kernel: one query per launch (fused selection tail, deferred selection, or SpMV-only scores).
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include <cstddef>
#include "packet_math.hpp"

namespace tkspmv {

#ifndef TKSPMV_STREAM_TURNS
#define TKSPMV_STREAM_TURNS 2
#endif
#ifndef TKSPMV_STREAM_PRIO
#define TKSPMV_STREAM_PRIO 2
#endif
#ifndef TKSPMV_REDUCER_SLEEP
#define TKSPMV_REDUCER_SLEEP 8
#endif
constexpr unsigned long long FLUSH_TAU_WAIT = 2000;  // x 10 ns: longest wait of a wave for a first threshold
#ifndef TKSPMV_DEFER_PACKETS
#define TKSPMV_DEFER_PACKETS 2
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
// DBG = false (production): tracing / statistics / ablation hooks compiled out (see batch_kernel).
template <int C, bool SCORES, int XCOLS, int QM = 0, int NBUF = TKSPMV_NBUF, bool DBG = false>
__global__ void __launch_bounds__(576, ((C == 8 && !SCORES) || QM == 7) ? 6 : 5) stream_kernel(const StreamParams P, const SelectParams SP) {
    // (constants, not a modified copy of P: a copy that is passed on by reference ends up in scratch memory)
    unsigned long long *const dbg_trace = DBG ? P.trace : nullptr;
    unsigned long long *const dbg_stamps = DBG ? P.stamps : nullptr;
    unsigned long long *const dbg_counters = DBG ? P.dbg : nullptr;
    constexpr bool Q8 = QM == 1 || QM == 2;  // x staged as Q1.7 integers
    constexpr int VT = value_type_of(QM);
    constexpr bool INT = int_sums<QM>();
    // Deferred packets live in registers (C row sums + C / 2 flag words each): with 8 entries per lane one packet is held,
    // not three -- the same number of rows as two 4-entry packets, and the kernel stays at 80 registers (two workgroups
    // per CU; with three it needed 96 and a single query took 57 us instead of 36).
    // (at most 1024 columns: two -- the third would push the 16-bit layouts past 80 registers; the 12-bit layout keeps one)
    constexpr int DEFER_C = (C == 8 || QM == 7) ? 1 : (XCOLS <= 1024 ? DEFER : DEFER + 1);
    __shared__ StreamLds<XCOLS> L;
    // (reduce_packet forms LDS addresses of x as (word & 0xFFC) | base: x must sit on a 4 KiB boundary -- this object is the
    //  kernel's ONLY __shared__ block, so it starts at LDS address 0, and x is its first member)
    using LdsBlock = StreamLds<XCOLS>;
    static_assert(offsetof(LdsBlock, u) == 0 && offsetof(decltype(LdsBlock::u), w) == 0 && offsetof(decltype(LdsBlock::u.w), x) == 0, "x must be the first member of the kernel's LDS block");
    float *x_lds = L.u.w.x;
    const uint32_t xbase = lds_addr_of(L.u.w.x);  // (0: the object is the kernel's only LDS block and x its first member)
    uint2 *cand = L.u.w.cand;
    uint32_t *misc = L.misc;
    SelectShared &sel_sh = L.u.sel;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t bid = blockIdx.x, n_wg = gridDim.x;  // streaming workgroup id / count
    // TKSPMV_TRACE=1: 100 MHz wall-clock stamps per wave (kept in SGPRs, written once at the very end)
    unsigned long long *tr = (!SCORES && dbg_trace) ? dbg_trace + ((size_t)blockIdx.x * 9u + wave) * 8u : nullptr;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0, tr3 = 0, tr4 = 0;
    if (tr) tr0 = __builtin_amdgcn_s_memrealtime();
    if (!SCORES && SP.t_start && blockIdx.x == 0u && tid == 0u)  // (workgroup 0 is dispatched first: the launch's start within a microsecond)
        __hip_atomic_store(SP.t_start, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!SCORES && P.deferred) {
        // The selection of the previous query rides along in workgroup 0 (SP.n_wg = 0: there is none); the others
        // stream. The launch has as many workgroups as fit the GPU at once (two per CU) and the matrix is cut into
        // one partition per streaming wave of grid - 1 workgroups, so nothing waits for a free slot: the selection
        // runs during the launch's start-up, when the memory system is still idle.
        if (bid == 0u) {
            if (SP.n_wg != 0u) select_body<1>(SP, tid, blockDim.x, sel_sh);
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
    if (q < P.n_parts) TKSPMV_PARTITION_RANGE(P, q, p0, np);
    auto prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NBUF - 1; ++u) {  // NBUF-1 packets in flight
            rbs[u] = 0u;
            if (np > 0) {
                const uint32_t iu = ((uint32_t)u < np) ? (uint32_t)u : (np - 1);
                load_packet<C, VT>(P.packets + (size_t)(p0 + iu) * P.packet_bytes, lane, buf[u]);
                rbs[u] = SCORES ? P.pkt_row[p0 + iu] : scalar_load(P.pkt_row + p0 + iu);  // (the SpMV-only variant needs it for every packet)
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
    } else if (QM == 4 || QM == 6 || QM == 8) {
        unit_scale = 2147483648.0f;  // scores are Q1.31 words converted to fp32
    }
    const float inv_unit = 1.0f / unit_scale;             // exact: unit_scale is a power of two
    const float min_units = P.min_score * unit_scale;
    for (uint32_t i = tid; i < (uint32_t)XCOLS; i += blockDim.x) {
        const float xw = P.x[i < P.cols ? i : 0u];  // (clamped address, masked value: the loads are issued back to back)
        const float xv = (i < P.cols) ? xw : 0.0f;
        if (Q8)
            reinterpret_cast<uint32_t *>(x_lds)[i] = to_q1_7_dev(xv * x_scale);  // x quantised like the matrix values
        else if (QM == 6)  // bit-packed narrow fixed point: x as a 20-bit integer
            reinterpret_cast<uint32_t *>(x_lds)[i] = to_fixed_dev(xv, P.fixed_width) >> 12;
        else if (QM == 4 || QM == 8)  // W <= 24: as a 24-bit integer (see reduce_packet)
            reinterpret_cast<uint32_t *>(x_lds)[i] = to_fixed_dev(xv, P.fixed_width) >> (P.fixed_width <= 24u ? 8 : 0);
        else
            x_lds[i] = QM == 5 ? xv * Q17_UNIT : xv;
    }
    if (tid == 0) misc[MISC_TAU] = __float_as_uint(min_units);
    __syncthreads();
    if (tr) tr1 = __builtin_amdgcn_s_memrealtime();

    if (is_server) {
        if (!SCORES && P.n_sets != 0u) {
            for (;;) {
                publish_group_max(P, bid, lane, misc);
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
    // Flush of the wave's list: what clears the threshold goes to global memory -- the first survivor of the launch to this
    // wave's fixed slot, further ones to the shared overflow list (write-through stores: in fused mode another workgroup of
    // this launch reads them) -- and the list is empty again. Slots without a survivor are NOT written: the selection resets
    // every slot it consumed, so an untouched slot is invalid by construction.
    bool slot_used = false;
    auto flush_list = [&](float tau_f) __attribute__((always_inline)) {
        ListScan<WAVE_CAP / 64u> LS;
        const uint32_t surv = scan_list<WAVE_CAP / 64u>(wcand, wcnt, tau_f, lane, LS);
        wcnt = 0u;
        if (surv == 0u) return;
        const uint32_t to_slot = slot_used ? 0u : 1u;
        uint32_t gbase = 0u;
        if (surv > to_slot) {
            if (lane == 0) gbase = atomicAdd(P.ovf_count, surv - to_slot);
            gbase = __builtin_amdgcn_readfirstlane(gbase);
        }
        unsigned long long *slot = P.wg_cand + (size_t)bid * WG_SLOTS + wave;
#pragma unroll
        for (uint32_t u = 0; u < WAVE_CAP / 64u; ++u) {
            if (LS.keep[u]) {
                const unsigned long long v = pack_cand(LS.e[u].x, LS.e[u].y);
                if (LS.pos[u] < to_slot) st_agent(slot, v);
                else if (gbase + LS.pos[u] - to_slot < P.ovf_cap) st_agent(&P.ovf_cand[gbase + LS.pos[u] - to_slot], v);
            }
        }
        slot_used = true;
    };
    // (Flushing EARLY -- four packets before the end of the partition, so that the stores' ~3 us trip to memory rides under the
    //  remaining packets instead of sitting between the last workgroup's last packet and its ticket -- was measured and is not
    //  done: the threshold of that moment lets thousands of rows through that the final one stops (the final flush delivers
    //  ~110 candidates per query at 1M rows), the overflow list and the selection grow: 32-35 us per launch against 30.9.)
    for (bool first_part = true; q < P.n_parts; q += total_waves, first_part = false) {
        if (!first_part) {  // more partitions than waves (not the case for engines built by tkspmv_create)
            TKSPMV_PARTITION_RANGE(P, q, p0, np);
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
            st[d].fl = 0u;
        }

        // Two packets in flight behind the one being reduced. The buffers rotate by NAME (the loop is unrolled by
        // NBUF): copying a freshly loaded buffer into another would wait for the youngest load and drain the
        // prefetch queue every iteration.
        for (uint32_t i0 = 0; i0 < np; i0 += NBUF) {
#if TKSPMV_STREAM_TURNS
            // The two workgroups of a CU take turns at the higher priority (see batch_kernel.hpp): at equal priority the
            // older one wins the arbitration, finishes ~3 us early and leaves the CU to the other for the rest of the launch.
            // SpMV-only variant: 22.9 against 23.75 us. The top-k variant loses by it (35.4-36.0 against 34.5 us per single
            // launch, whatever the turn length): it keeps equal priorities.
            if (SCORES && !is_server) {
                if ((((i0 / (uint32_t)(NBUF * TKSPMV_STREAM_TURNS)) ^ (bid >= n_wg / 2u ? 1u : 0u)) & 1u) != 0u) __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO);
                else __builtin_amdgcn_s_setprio(TKSPMV_STREAM_PRIO - 1);
            }
#endif
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
                const uint32_t ia = (i + (NBUF - 1) < np) ? (i + (NBUF - 1)) : (np - 1);
                load_packet<C, VT>(pk + (size_t)ia * P.packet_bytes, lane, ahead);
                rb_ahead = SCORES ? P.pkt_row[p0 + ia] : scalar_load(P.pkt_row + p0 + ia);
            }
            float tau = 0.0f;
            if (!SCORES)
                tau = __uint_as_float(
                    __hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));

            const Reduced<C> Rd = reduce_packet<C, QM>(cur, carry, xbase, P.fixed_mask);
            if (tr && i == 0u) tr2 = __builtin_amdgcn_s_memrealtime() + (__float_as_uint(Rd.S) & 0u);

            if (SCORES) {
                const RowSums<C> R = expand<C, INT>(Rd, packet_flags<C, QM>(cur));
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
                    const RowSums<C> R = expand<C, INT>(Rd, packet_flags<C, QM>(cur));
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
                } else if (__any(trigger_of<C, INT>(Rd) >= tau)) {
                    // (the trigger bounds every finished row of its lane from above: packet_math.hpp)
                    const RowSums<C> R = expand<C, INT>(Rd, packet_flags<C, QM>(cur));
                    offer_candidates<C, QM, ListGeom<XCOLS>::WAVE_CAP, DBG>(P, R, rb_cur, tau, lane, grp_local, publishes, wcand, wcnt, misc);
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
                    offer_candidates<C, QM, ListGeom<XCOLS>::WAVE_CAP, DBG>(P, st[d], st_rb[d], tau, lane, grp_local, publishes, wcand, wcnt, misc);
            }
        }
    }

    if (SCORES) return;
    if (tr) tr4 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long ts_stream_end = dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (!is_server && lane == 0) atomicAdd(&misc[MISC_DONE], 1u);

    // ---- flush: every wave on its own, no workgroup synchronisation. What still clears the (now much tighter)
    // threshold leaves the wave's private list: the first survivor to this wave's fixed slot, further ones to the
    // shared overflow list. Slots without a survivor are NOT written: the selection resets every slot it consumed,
    // so an untouched slot is invalid by construction. Write-through (sc1) stores: in fused mode another workgroup
    // of this launch reads them.
    if (!is_server) {
        const float tau = __uint_as_float(
            __hip_atomic_load(&misc[MISC_TAU], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (wcnt != 0u) flush_list(tau);
        if (dbg_counters && lane == 0 && wave == 0u) {  // TKSPMV_STATS=1 (approximate: waves still running are not counted)
            atomicAdd(&dbg_counters[0], (unsigned long long)misc[MISC_SLOW_CNT]);
            atomicAdd(&dbg_counters[1], (unsigned long long)misc[MISC_CAND_CNT]);
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
    const unsigned long long ts_flush_issued = dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long ts_flush_done = dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;
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
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sel_sh.last = last;
    }
    __syncthreads();
    const unsigned long long ts_ticket = dbg_stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    if (sel_sh.last) select_body<1>(SP, tid, blockDim.x, sel_sh, dbg_stamps, inv_unit);
    if (dbg_stamps && sel_sh.last && tid == 0) {
        dbg_stamps[0] = ts_stream_end;
        dbg_stamps[1] = ts_flush_issued;
        dbg_stamps[2] = ts_flush_done;
        dbg_stamps[3] = ts_ticket;
        dbg_stamps[7] = __builtin_amdgcn_s_memtime();
        dbg_stamps[8] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace tkspmv
