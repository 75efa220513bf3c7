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
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <string>
#include <vector>

#include "device_pack.hpp"
#include "engine.hpp"
#include "options.hpp"
#include "wsell.hpp"

#include "kernels/common.hpp"
#include "kernels/select.hpp"
#include "kernels/packet_math.hpp"
#include "kernels/stream_kernel.hpp"
#include "kernels/local.hpp"
#include "kernels/batch_kernel.hpp"
#include "kernels/multi_kernel.hpp"
#include "kernels/radix_select.hpp"
#include "kernels/read_probe.hpp"

namespace tkspmv {

// Empty kernel with the stream kernel's geometry: calibrates what an event bracket adds around one launch.
__global__ void __launch_bounds__(576) null_kernel(const uint32_t *p) {
    if (p == nullptr && threadIdx.x == 123456u) __builtin_trap();
}

// What a KERNEL sees of words the CPU stored into device memory through the PCIe BAR (tkspmv_create's check of the BAR path of
// tkspmv_set_query): plain loads, through the same caches a streaming kernel's loads of x go through.
__global__ void bar_check_kernel(const uint32_t *x, uint32_t last, uint32_t *out) {
    if (threadIdx.x == 0) {
        out[0] = x[0];
        out[1] = x[last];
    }
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
    // Low-latency host boundary of the reference loop (reset -> operator() -> read_result, host_spmv_bscsr.cpp:602-632):
    // set_query copies x into pinned host memory and enqueues the 4 KiB upload without waiting for it; run() launches the
    // fused kernel, whose selection tail also writes the k results and an epoch flag to pinned host memory, and polls that
    // flag; read() copies from the pinned block. No stream synchronisation, no device-to-host copy on the critical path.
    float *h_x = nullptr;            // pinned staging copy of x
    uint32_t *h_res = nullptr;       // pinned: [k] row ids, [k] score bits, [2k] epoch flag
    uint32_t *h_res_dev = nullptr;   // the same block as the device sees it
    mutable uint32_t host_epoch = 0;
    mutable bool last_on_host = false;  // the most recent result is complete in h_res
    bool x_pending = false;             // an upload from h_x may still be in flight
    // Large BAR (every MI355X server has it; checked at create): tkspmv_set_query writes x into device memory with plain CPU stores
    // through the PCIe aperture -- 0.5 us for 4 KiB, no copy kernel ahead of the query's launch (that copy, 3-10 us of device time
    // plus a dispatch, stood between set_query and every tkspmv_run). Posted writes stay ordered with the launch's doorbell.
    bool bar_x = false;
    // CPU stores through the BAR pass the host data path (HDP) on their way to memory: the documented protocol ends them with a
    // write to the device's HDP_MEM_COHERENCY_FLUSH_CNTL register and a read back of it (what ROCclr does behind its own large-BAR
    // copies); the register's address comes from hipDeviceAttributeHdpMemFlushCntl. No address, no BAR path.
    volatile uint32_t *hdp_flush = nullptr;
    void flush_hdp() const {
        __builtin_ia32_sfence();
        *hdp_flush = 1u;
        (void)*hdp_flush;  // (the read completes only behind the posted writes before it)
    }
    int host_path = 1;                  // TKSPMV_HOST_PATH=0: the plain path (stream synchronisation + copies)
    float *h_x_dev = nullptr;           // h_x as the device sees it (TKSPMV_HOST_X=direct: kernels read x from host memory)
    bool host_x_direct = false;
    bool run_events = false;            // TKSPMV_RUN_EVENTS=1: tkspmv_run brackets the fused launch with a hipEvent pair instead of taking the kernel's own duration
    unsigned long long *d_tstart = nullptr;  // the fused launch's start stamp (SelectParams::t_start)
    // Exchange state of one query in flight (published maxima, threshold word, survivor slots, overflow list).
    // Two sets: with deferred selection, launch q+1 streams into one set while its workgroup 0 selects query q from
    // the other. Everything else uses set 0.
    struct ExState {
        uint32_t *tau_g = nullptr, *gmax = nullptr, *ovf_count = nullptr;
        unsigned long long *wg_cand = nullptr, *ovf = nullptr;
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
    uint32_t *d_alias_idx = nullptr;  // [BATCH_MAX][k] where the earlier queries of a launch write when they share the last one's buffer
    float *d_alias_val = nullptr;
    uint32_t *d_done = nullptr;
    bool fused = true;
    bool can_defer = false;
    bool can_batch = false;         // batch kernel usable (exchange on, x double-buffered in LDS, 4 entries per lane)
    // Queries per launch of the batch kernel = exchange-state sets allocated. Every set carries an overflow list that must be
    // able to hold EVERY row (a degenerate query -- x = 0, all scores equal -- makes every row a candidate, and the result
    // must still be exact), 8 B per row: 256 MB at 1M rows, 2.6 GB at 10M rows -- under 1 % of this GPU's 288 GB either way;
    // 32 sets up to 64M rows, fewer beyond (the lists stay within 16 GiB).
    int batch_max = BATCH_MAX;
    // Multi-query passes (multi_kernel, desc.multi_q): fp32 values, <= 1024 columns, exchange on. multi_q queries share one
    // pass over the wave-sliced ELL copy of the matrix (wsell.hpp); a group's selection is owed to the next launch (or to
    // drain()). Groups alternate between the exchange-state sets [0, MULTI_Q_MAX) and [MULTI_Q_MAX, 2 * MULTI_Q_MAX).
    bool can_multi = false;
    int multi_q = 0;
    // queries per multi-query LAUNCH (= its selector workgroups): multi_q, or with one query per pass up to MULTI_Q_MAX passes
    int multi_group = 0;
    // Large k (see radix_hist_kernel): every query = scores kernel + radix select + the selection kernel
    bool use_radix = false;
    bool approx_parts = false;    // partitions > 1 with k > k_per_partition: the reference's lossy per-partition lists
    float *d_rscores = nullptr;   // [rows], -inf where a row has no entry (never written by the scores kernel)
    uint32_t *d_rhist = nullptr;  // [4][256]
    uint8_t *d_sell_packets = nullptr;
    std::vector<uint8_t *> d_sell_replicas;
    uint32_t *d_sell_rows = nullptr, *d_sell_part_first = nullptr, *d_sell_part_count = nullptr, *d_sell_part_slice0 = nullptr;
    uint32_t sell_parts = 0, multi_stream_waves = 8, sell_packet_bytes = 1536;
    bool sell_byte_values = false;  // TKSPMV_Q1_7_F32: 768-byte chunks of Q1.7 bytes
    bool sell_c12 = false;          // ... with 12-bit column words: 640-byte chunks (at most 1022 columns)
    uint64_t sell_bytes = 0;
    uint32_t *d_multi_out_idx = nullptr;  // [2 * MULTI_Q_MAX][k] results of tkspmv_time_multi
    uint64_t multi_stat_queries = 0, multi_waits = 0, multi_wait_ticks = 0, multi_rows_offered = 0, multi_rows_overflowed = 0;  // option STATS, summed over tkspmv_time_multi calls
    float *d_multi_out_val = nullptr;
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
    // checked local thresholds of the batch kernel (BatchParams): the launches' verdict words (two, by launch parity), the workgroups'
    // records per exchange-state set, the epoch words of the overflow lists' flow control
    unsigned long long *d_verdict = nullptr;
    // The host's side of the verdicts (round 5; BatchParams::verdict_host): a ring of host-visible words, one per batch launch on
    // the engine's own stream, and the launches whose word has not been looked at yet. While the verdicts the host has SEEN are
    // clean, a local launch goes out alone and settle() -- called wherever the host has just waited for the engine's stream --
    // repairs what a check flagged after all (an exact launch of the flagged queries, then one more wait). An observed failure
    // puts the in-stream repair launch back behind the next DISTRUST_LAUNCHES launches (it also keeps the gate's books).
    // Launches on a caller's stream are never trusted: nobody tells the engine when the caller has waited.
    static constexpr uint32_t VERDICT_RING = 8192, DISTRUST_LAUNCHES = 64;  // (launches the host may enqueue between two waits and still look at every verdict)
    unsigned long long *h_verdict = nullptr, *h_verdict_dev = nullptr;
    struct PendingCheck {
        BatchArgs A;
        uint32_t slot;
        bool trusted;  // no repair launch went out behind it
    };
    mutable std::deque<PendingCheck> pending_checks;
    mutable uint64_t verdict_seq = 0, late_repairs = 0, trusted_launches = 0;
    mutable uint32_t clean_seen = 0, distrust_left = 0;
    // Two launches in flight (round 5): a launch of 32 queries pays ~35 us beyond its queries -- 5 us before the first packet is
    // reduced, the wait for its last workgroup, the last selection, the gap to the next launch -- and an in-order stream pays them
    // once per launch. Consecutive launches of ONE sequence alternate between the engine's stream and `bside`: the workgroups of the
    // next launch are dispatched as the CUs of the previous one fall free, its ramp fills the other's tail. Only while launches are
    // trusted (no exact launch behind them: two of those would share the exchange state); per-launch state (records, tickets,
    // scale words, scratch results, verdict words) exists once per parity.
    hipStream_t bside = nullptr;
    hipEvent_t ev_bfork = nullptr, ev_bjoin = nullptr;
    bool overlap_launches = false;  // option OVERLAP=1 (measured slower with rank pacing as it is: DESIGN.md section 9)
    bool repair_by_host = true;  // option REPAIR=stream: the exact launch always follows in the stream (round 4's behaviour)
    unsigned long long *d_rec_slots = nullptr;  // [batch_max][grid][WG_SLOTS]
    uint32_t *d_rec_used = nullptr;             // [batch_max][grid]
    uint32_t *d_ovf_epoch = nullptr;            // [ovf_lists] x 32 words
    uint32_t ovf_lists = 1;                     // overflow lists allocated (batch engines: 4, or 2 with local thresholds; shared by the queries of a launch under flow control)
    uint32_t use_local = 0;  // workgroup-local thresholds (BatchParams::local: 0 off, 1 / 2: a wave's best / second best packet maximum)
    uint32_t single_mode = 0;  // the same choice for ONE query per launch (single_kernel): by the failure estimate alone, whatever the size
    float *d_wg_prior = nullptr;  // [grid] + the countdown word (BatchParams::wg_prior / prior_block)
    float *d_wg_sig = nullptr;    // [grid][8] the signatures of the remembered priors and the second prior (BatchParams::wg_sig; option SIGNATURES=0: none)
    float local_beta = 1.0f;
    uint32_t pace_tuned_us = 0;  // 0: the pacing is the static default (or an option); else what tkspmv_create's measurement took
    uint32_t pace_tune_launches = 0;  // ... and the batch launches it made (synthetic queries: a kernel trace of a run holds them)
    uint32_t pace_quads = 0, pace_levels = 3, pace_base = 0; // pacing by rank (BatchParams::pace_quads, pace_levels, pace_base)
    unsigned long long *d_wg_times = nullptr;  // option WG_TIMES: BatchParams::wg_times of the LAST batch launch (read through tkspmv_debug_trace)
    bool pace_carry = true;  // option PACE_CARRY=0: every launch starts unpaced
    uint32_t *d_wg_pace = nullptr;  // [grid] the pause every workgroup ended its last launch with (BatchParams::wg_pace)
    // time_queries: the events of its bracket ride on the region's first and last kernel (hipExtLaunchKernelGGL: the dispatch's own
    // start and end stamps -- what rocprofv3 reports) instead of being recorded around them: a start event recorded on an idle stream
    // is stamped before the host has written the dispatch packet (1.6 us of host time inside a 350 us bracket, tools/probes/event_probe.cpp)
    mutable hipEvent_t ext_start = nullptr, ext_stop = nullptr;
    mutable bool ext_last = false;  // (launch_sequence: the launch_batch being made is the region's last)
    mutable bool ext_record = false;  // the same two events RECORDED right in front of the first and right behind the last launch (EXT_EVENTS=2)
    uint32_t *d_pace_adapt = nullptr;  // [0] what launches add to the period, [1] late waves of the launch in flight (BatchParams::pace_adapt)
    bool pace_adapt_on = true;         // option PACE_ADAPT=0: the period stays what tkspmv_create measured
    uint32_t pace_period_ns = 0;  // pacing by the clock (BatchParams::pace_period): ns per query of every wave's timetable; 0: pacing by rank
    mutable uint64_t batch_launches = 0;
    uint32_t n_sel_wg = 1;   // selector workgroups of a batch launch (BatchParams::n_selectors): 4 on small matrices
    uint32_t groups_with_rows = 0;  // publishing groups that own at least one wave partition
    float *d_out_val = nullptr, *d_scores = nullptr;
    unsigned long long *d_stats = nullptr;
    uint32_t grid = 0, block = 0, gpw = 1, n_sets = 0, n_groups_pub = 0, cand_cap = 0, ovf_cap = 0, lds_bytes = 0,
             xcols = 1024;
    bool collect_stats = false;
    // TKSPMV_TRACE / TKSPMV_STATS / TKSPMV_STAMPS / TKSPMV_DBG_FLAGS / TKSPMV_DBG_REPEAT: launch the instantiations that
    // carry the tracing and ablation hooks (fp32 values, 4 entries per lane, <= 1024 columns; elsewhere the hooks do not exist)
    bool dbg_kernels = false;
    bool q8 = false;
    bool collect_stamps = false;
    unsigned long long *d_trace = nullptr;  // TKSPMV_TRACE=1: 4 launches x [grid+1][9][8] stamps
    size_t trace_words = 0;
    bool have_query = false;
    bool ran = false;
    bool packed_on_device = false;  // the stream was built by device_pack.hip
    uint32_t sell_pack_us = 0;      // packing of the wave-sliced ELL copy (multi-query engines)
    uint32_t *d_coo_col = nullptr;  // the COO's columns and values, kept in HBM between the two device packers (create only)
    float *d_coo_val = nullptr;
    uint32_t pack_us = 0;           // time of the packing step of tkspmv_create (upload of the COO included when on the device)
    // single_kernel (kernels/local.hpp): the launch of tkspmv_run on engines that stream with workgroup-local thresholds
    bool can_single = false;
    unsigned long long *d_lslots = nullptr;  // [grid][WG_SLOTS] the workgroups' records
    uint32_t *d_lused = nullptr;             // [grid]
    float *d_lprior = nullptr;               // [grid] carried thresholds | 32 words: the suspension counters (LocalParams::prior_block)
    uint32_t *d_lstatus = nullptr;           // [0] 1: the last single launch failed its check
    uint32_t *d_lready = nullptr;            // [grid] the streaming workgroups' flags (LocalParams::ready; NULL: the last workgroup selects)
    mutable uint64_t single_launches = 0, single_repairs = 0;
    uint32_t uni_ppp = 0, uni_last = 0;  // uniform partition table (StreamParams::uni_ppp): partition q = packets [q * uni_ppp, ...)

    StreamParams stream_params(const float *x, int set = 0) const {
        StreamParams P{};
        const ExState &E = st[set];
        P.packets = d_replicas.empty() ? d_packets : d_replicas[launch_counter % d_replicas.size()];
        P.pkt_row = d_pkt_row;
        P.part_first = d_part_first;
        P.part_count = d_part_count;
        P.x = x;
        P.n_parts = (uint32_t)info.n_wave_partitions;
        P.uni_ppp = uni_ppp;
        P.uni_last = uni_last;
        P.cols = desc.cols;
        P.packet_bytes = pm.packet_bytes;
        P.n_sets = n_sets;
        P.k = (uint32_t)desc.k;
        P.n_groups_pub = n_groups_pub;
        P.gpw = gpw;
        P.min_score = desc.min_score;  // converted to score units inside the kernel
        P.fixed_width = pm.fixed_width;
        P.fixed_mask = pm.fixed_width ? fixed_mask(pm.fixed_width) : 0u;
        P.gmax = E.gmax;
        P.tau_g = E.tau_g;
        P.n_reducers = grid < 8u ? grid : 8u;
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
        S.stats = collect_stats ? d_stats : nullptr;
        S.host_out = nullptr;
        S.host_epoch = 0u;
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
        hipLaunchKernelGGL(select_group_kernel, dim3(serial ? 1u : G.n_q), dim3(SEL_THREADS), 0, s, S, set_addr(0), G, serial ? 1u : 0u);
        pending_group[c].n_q = 0u;
    }
    SetAddr set_addr(int s0) const {
        SetAddr A{};
        A.gmax0 = st[s0].gmax;
        A.tau_g0 = st[s0].tau_g;
        A.ovf_count0 = st[s0].ovf_count;
        A.wg_cand0 = st[s0].wg_cand;
        A.ovf_cand0 = st[s0].ovf;
        A.unit_inv0 = st[s0].unit_inv;
        A.gmax_stride = GMAX_WORDS;
        A.word_stride = STATE_WORD_STRIDE;
        A.cand_stride = grid * WG_SLOTS;
        A.ovf_stride = ovf_cap;
        return A;
    }
    // n <= multi_group queries in ONE launch (multi_q per pass over the matrix); their selection is owed (pending_group) to the next multi launch
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
        P.uni_ppp = 0u;  // (the row-per-lane stream has its own partition tables)
        P.packet_bytes = sell_packet_bytes;
        P.pkt_row = nullptr;
        MultiParams M{};
        M.A = set_addr(0);
        M.part_slice0 = d_sell_part_slice0;
        M.n_sel = (uint32_t)multi_group;
        M.prev = pending_group[chain];
        M.cur.n_q = (uint32_t)n;
        M.cur.set0 = (uint32_t)((2 * chain + multi_parity[chain]) * MULTI_Q_MAX);
        for (int q = 0; q < n; ++q) {  // (one copy of the stream per PASS: the queries of a pass share it)
            BatchIO &Q = M.cur.io[q];
            Q.x = xs[q];
            Q.packets = d_sell_replicas.empty() ? d_sell_packets : d_sell_replicas[(launch_counter + (uint64_t)(q / multi_q)) % d_sell_replicas.size()];
            Q.out_idx = out_idx[q];
            Q.out_val = out_val[q];
        }
        launch_counter += (uint64_t)((n + multi_q - 1) / multi_q);
        SelectParams S = select_params(nullptr, nullptr, 0);
        S.pos_to_row = d_sell_rows;
        const dim3 mblock(multi_stream_waves * 64u + 64u);
        typedef void (*multi_fn)(const StreamParams, const SelectParams, const MultiParams);
        const int qi = multi_q <= 1 ? 0 : (multi_q <= 2 ? 1 : (multi_q <= 4 ? 2 : 3));
        static const multi_fn fns[3][4] = {{&multi_kernel<1, 0>, &multi_kernel<2, 0>, &multi_kernel<4, 0>, &multi_kernel<8, 0>},
                                           {&multi_kernel<1, 1>, &multi_kernel<2, 1>, &multi_kernel<4, 1>, &multi_kernel<8, 1>},
                                           {&multi_kernel<1, 5>, &multi_kernel<2, 5>, &multi_kernel<4, 5>, &multi_kernel<8, 5>}};
        hipLaunchKernelGGL(fns[sell_byte_values ? (sell_c12 ? 2 : 1) : 0][qi], dim3(grid), mblock, 0, s, P, S, M);
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
        if (multi_chains < 2 || n <= 2 * multi_group || out_idx[0] == out_idx[1]) {
            for (int i = 0; i < n; i += multi_group) launch_multi(xs + i, out_idx + i, out_val + i, std::min(multi_group, n - i), s);
            drain(s);
            return;
        }
        drain(s);
        (void)hipEventRecord(ev_fork, s);
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        int g = 0;
        for (int i = 0; i < n; i += multi_group, ++g)
            launch_multi(xs + i, out_idx + i, out_val + i, std::min(multi_group, n - i), (g & 1) ? side : s, g & 1);
        drain(s);  // both chains' last selections, then the caller's stream waits for the side stream
    }
    // One query, its result complete in stream order right after these launches: the stream kernel and, unless
    // fused into its tail, the select kernel.
    void launch_query(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s) const {
        if (desc.impl == TKSPMV_IMPL_ROW_PER_LANE && can_multi) {  // the row-per-lane variant, one query per pass
            launch_multi_sequence(&x, &out_idx, &out_val, 1, s);
            return;
        }
        drain(s);
        if (use_radix) {
            launch_query_radix(x, out_idx, out_val, s);
            return;
        }
        launch_stream(x, out_idx, out_val, s);
        if (!fused) launch_select(out_idx, out_val, s);
    }
    typedef void (*batch_fn)(const BatchArgs);
    // can_batch: x of at most 1024 columns (it is held twice in LDS). local: the kernel of the checked local thresholds.
    template <bool LOCAL>
    batch_fn batch_kernel_of() const {
        if (desc.precision == TKSPMV_Q1_7) return &batch_kernel<4, 1024, 1, false, LOCAL>;
        if (desc.precision == TKSPMV_Q1_7_WIDE) return &batch_kernel<4, 1024, 2, false, LOCAL>;
        if (desc.precision == TKSPMV_F16) return &batch_kernel<4, 1024, 3, false, LOCAL>;
        if (desc.precision == TKSPMV_FIXED)
            return pm.precision == Precision::FIXED20 ? &batch_kernel<4, 1024, 6, false, LOCAL>
                                                      : (pm.precision == Precision::FIXED26 ? &batch_kernel<4, 1024, 8, false, LOCAL> : &batch_kernel<4, 1024, 4, false, LOCAL>);
        if (desc.precision == TKSPMV_Q1_7_F32) return &batch_kernel<4, 1024, 5, false, LOCAL>;
        if (info.packet_entries == 512) return &batch_kernel<8, 1024, 0, false, LOCAL>;
        if (pm.precision == Precision::F32C12) return dbg_kernels ? &batch_kernel<4, 1024, 7, true, LOCAL> : &batch_kernel<4, 1024, 7, false, LOCAL>;
        return dbg_kernels ? &batch_kernel<4, 1024, 0, true, LOCAL> : &batch_kernel<4, 1024, 0, false, LOCAL>;
    }
    batch_fn batch_kernel_for(bool local = false) const { return local ? batch_kernel_of<true>() : batch_kernel_of<false>(); }
    // n <= BATCH_MAX queries in one launch of the batch kernel; results complete in stream order after the launch.
    void launch_batch(const float *const *xs, uint32_t *const *out_idx, float *const *out_val, int n, hipStream_t s, uint32_t parity = 0u,
                      bool final_launch = true) const {
        // (first: the region's start event goes with this kernel, if one is waiting; last: the stop event, if this is the region's final launch)
        auto go = [&](void (*fn)(BatchArgs), const BatchArgs &args, bool last) {
            hipEvent_t a = ext_start, b = (last && ext_last) ? ext_stop : nullptr;
            ext_start = nullptr;
            if (b) ext_stop = nullptr;
            if (ext_record) {
                if (a) (void)hipEventRecord(a, s);
                hipLaunchKernelGGL(fn, dim3(grid), dim3(block + 64), 0, s, args);
                if (b) (void)hipEventRecord(b, s);
            } else if (a || b) {
                hipExtLaunchKernelGGL(fn, dim3(grid), dim3(block + 64), 0, s, a, b, 0, args);
            } else {
                hipLaunchKernelGGL(fn, dim3(grid), dim3(block + 64), 0, s, args);
            }
        };
        if (!bside || s != bside) drain(s);
        StreamParams P = stream_params(xs[0], 0);
        P.fused = 0u;
        SelectParams S = select_params(out_idx[0], out_val[0], 0);
        BatchParams B{};
        B.n_q = (uint32_t)n;
        B.tickets = d_tickets + (size_t)parity * BATCH_MAX * 32;
        static_cast<SetAddr &>(B) = set_addr(0);
        B.unit_inv0 += (size_t)parity * BATCH_MAX * STATE_WORD_STRIDE;
        for (int q = 0; q < n; ++q) {
            BatchIO &Q = B.io[q];
            Q.x = xs[q];
            Q.packets = d_replicas.empty() ? d_packets : d_replicas[(launch_counter + q) % d_replicas.size()];
            Q.out_idx = out_idx[q];
            Q.out_val = out_val[q];
            // Result buffers shared by several queries of the launch (the engine-owned pair: "the last query wins"): selections run
            // concurrently and a repair phase writes after the fact, so only the LAST query may keep such a buffer -- the earlier
            // ones write to a scratch block nobody reads.
            // (a launch that overlaps with its successor: its last query too, if it names the engine's own pair)
            if ((q + 1 < n && out_idx[q] == out_idx[n - 1]) || (!final_launch && out_idx[q] == d_out_idx)) {
                Q.out_idx = d_alias_idx + ((size_t)parity * BATCH_MAX + q) * desc.k;
                Q.out_val = d_alias_val + ((size_t)parity * BATCH_MAX + q) * desc.k;
            }
        }
        launch_counter += (uint64_t)n;
        B.n_selectors = n_sel_wg;
        B.local = use_local;
        B.pace_quads = pace_quads;
        B.pace_levels = pace_levels;
        B.pace_base = pace_base;
        B.pace_period = (uint32_t)std::min<uint64_t>(0xFFFFFF00ull, (uint64_t)pace_period_ns * 256u / 10u);
        // (the waves keep the timetable in 32 bits of ticks << 8: 168 ms to the wrap. A launch that could last a quarter of that -- 32
        //  queries of 1.3 ms: 80M rows -- paces by rank instead)
        if ((uint64_t)pace_period_ns * (uint64_t)n > 40000000ull) B.pace_period = 0u;
        B.pace_adapt = pace_adapt_on ? d_pace_adapt : nullptr;
        B.wg_pace = pace_carry ? d_wg_pace : nullptr;
        B.wg_times = d_wg_times;
        B.prior_block = reinterpret_cast<uint32_t *>(d_wg_prior + grid);
        B.gate_parity = (uint32_t)(batch_launches & 1u);
        // (four words: a launch zeroes the word of the launch after the next -- its successor on the same stream --, so the scheme
        //  holds with one launch at a time and with two in flight)
        B.verdict = d_verdict + 16 * (batch_launches & 3u);
        B.verdict_next = d_verdict + 16 * ((batch_launches + 2u) & 3u);
        ++batch_launches;
        if (use_local) {
            B.wg_prior = d_wg_prior;
            B.wg_sig = d_wg_sig;
            B.local_beta = local_beta;
        }
        B.lslots = d_rec_slots + (size_t)parity * BATCH_MAX * grid * WG_SLOTS;
        B.lused = d_rec_used + (size_t)parity * BATCH_MAX * grid;
        B.lslots_stride = grid * WG_SLOTS;
        B.lused_stride = grid;
        B.ovf_epoch = d_ovf_epoch;
        B.ovf_lists = std::min(ovf_lists, 4u);  // (engines with one list per set -- the multi-query ones -- lend the batch kernel their first four)
        if (use_local && (s == stream || (bside && s == bside)) && h_verdict && pending_checks.size() + 1u < VERDICT_RING) {
            const uint32_t slot = (uint32_t)(verdict_seq++ % VERDICT_RING);
            h_verdict[slot] = 0ull;
            B.verdict_host = h_verdict_dev + slot;
            const bool trusted = repair_by_host && clean_seen != 0u && distrust_left == 0u;
            BatchArgs A{P, S, B};
            go(batch_kernel_for(true), A, trusted);
            if (!trusted) {
                BatchArgs R = A;
                R.B.repair = 1u;
                go(batch_kernel_for(false), R, true);
                if (distrust_left != 0u) --distrust_left;
            } else {
                ++trusted_launches;
            }
            pending_checks.push_back(PendingCheck{A, slot, trusted});
            return;
        }
        BatchArgs A{P, S, B};
        if (use_local) {
            // the kernel of the checked local thresholds, then the exact kernel for whatever failed its check (the launch is
            // empty -- every workgroup reads the verdict word and leaves -- unless a query was unlike the ones before it)
            go(batch_kernel_for(true), A, false);
            A.B.repair = 1u;
        }
        go(batch_kernel_for(false), A, true);
    }
    // The host has just waited for the engine's stream: look at the verdicts of the launches enqueued since it last did. A flagged
    // query of a TRUSTED launch (no repair launch behind it) is repaired now -- exact launch of the flagged queries, one more wait --;
    // any failure brings the in-stream repair launch back for a while. Returns hipSuccess, or the error of the repair's wait.
    hipError_t settle() const {
        bool late = false;
        const size_t n_pending = pending_checks.size();
        for (size_t i = 0; i < n_pending; ++i) {
            PendingCheck &pc = pending_checks[i];
            const unsigned long long v = const_cast<volatile unsigned long long *>(h_verdict)[pc.slot];
            const uint32_t mask = (uint32_t)(v >> 32);
            // (a word that is not complete: the launch was not waited for after all -- a caller polling an event of its own; such a
            //  launch counts as a failure of trust, not of a check: everything of it is run again below if nobody repaired it)
            const bool complete = (uint32_t)v == pc.A.B.n_q;
            if (complete && mask == 0u) {
                if (clean_seen != 0xFFFFFFFFu) ++clean_seen;
                continue;
            }
            clean_seen = 0u;
            distrust_left = DISTRUST_LAUNCHES;
            if (!pc.trusted) continue;  // (its in-stream repair launch has dealt with it)
            BatchArgs R = pc.A;
            R.B.repair = 2u;
            R.B.repair_mask = complete ? mask : (R.B.n_q >= 32u ? 0xFFFFFFFFu : ((1u << R.B.n_q) - 1u));
            R.B.verdict_host = nullptr;
            // "the last query wins" for the engine-owned result pair: a launch that was not the last one enqueued must not write
            // there again after the fact
            if (i + 1u < n_pending)
                for (uint32_t q = 0; q < R.B.n_q; ++q)
                    if (R.B.io[q].out_idx == d_out_idx) {
                        R.B.io[q].out_idx = d_alias_idx + (size_t)q * desc.k;
                        R.B.io[q].out_val = d_alias_val + (size_t)q * desc.k;
                    }
            hipLaunchKernelGGL(batch_kernel_for(false), dim3(grid), dim3(block + 64), 0, stream, R);
            ++late_repairs;
            late = true;
        }
        pending_checks.clear();
        return late ? hipStreamSynchronize(stream) : hipSuccess;
    }
    // A back-to-back sequence of queries given as pointer lists: batch kernel launches of up to BATCH_MAX queries
    // when it is available, else deferred selection.
    void launch_sequence(const float *const *xs, uint32_t *const *out_idx, float *const *out_val, int n, hipStream_t s) const {
        if (!can_batch) {
            for (int i = 0; i < n; ++i) launch_deferred(xs[i], out_idx[i], out_val[i], s);
            drain(s);
            return;
        }
        // Two launches in flight where every launch of the sequence goes out trusted (see bside): fork, alternate, join -- the
        // caller's view of the engine's stream is unchanged (everything enqueued here is complete when the stream is).
        const int n_launches = (n + batch_max - 1) / batch_max;
        const bool overlap = overlap_launches && bside && use_local && s == stream && n_launches >= 2 && repair_by_host && h_verdict &&
                             clean_seen != 0u && distrust_left == 0u && pending_checks.size() + (size_t)n_launches + 1u < VERDICT_RING;
        if (overlap) {
            drain(s);
            (void)hipEventRecord(ev_bfork, s);
            (void)hipStreamWaitEvent(bside, ev_bfork, 0);
        }
        int l = 0;
        for (int i = 0; i < n; i += batch_max, ++l) {
            const bool odd = overlap && (l & 1);
            ext_last = i + batch_max >= n;
            launch_batch(xs + i, out_idx + i, out_val + i, std::min(batch_max, n - i), odd ? bside : s, odd ? 1u : 0u, !overlap || i + batch_max >= n);
        }
        if (overlap) {
            (void)hipEventRecord(ev_bjoin, bside);
            (void)hipStreamWaitEvent(s, ev_bjoin, 0);
        }
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
        if (desc.precision == TKSPMV_FIXED && pm.precision == Precision::FIXED20)  // bit-packed: at most 1024 columns
            return scores ? &stream_kernel<4, true, 1024, 6> : &stream_kernel<4, false, 1024, 6>;
        if (desc.precision == TKSPMV_FIXED && pm.precision == Precision::FIXED26)  // five bytes per entry: at most 1024 columns
            return scores ? &stream_kernel<4, true, 1024, 8> : &stream_kernel<4, false, 1024, 8>;
        if (desc.precision == TKSPMV_FIXED) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 4> : &stream_kernel<4, false, 1024, 4>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 4> : &stream_kernel<4, false, 4096, 4>;
            return scores ? &stream_kernel<4, true, 16384, 4> : &stream_kernel<4, false, 16384, 4>;
        }
        if (desc.precision == TKSPMV_Q1_7_F32) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 5> : &stream_kernel<4, false, 1024, 5>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 5> : &stream_kernel<4, false, 4096, 5>;
            return scores ? &stream_kernel<4, true, 16384, 5> : &stream_kernel<4, false, 16384, 5>;
        }
        if (desc.precision == TKSPMV_Q1_7_WIDE) {
            if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024, 2> : &stream_kernel<4, false, 1024, 2>;
            if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096, 2> : &stream_kernel<4, false, 4096, 2>;
            return scores ? &stream_kernel<4, true, 16384, 2> : &stream_kernel<4, false, 16384, 2>;
        }
        if (c8) return scores ? &stream_kernel<8, true, 1024, 0, 2> : &stream_kernel<8, false, 1024, 0, 2>;  // (two packet buffers: 3 KB packets)
        if (pm.precision == Precision::F32C12) {  // 12-bit column words: at most 1024 columns, 4 entries per lane
            if (dbg_kernels && !scores) return &stream_kernel<4, false, 1024, 7, TKSPMV_NBUF, true>;
            return scores ? &stream_kernel<4, true, 1024, 7> : &stream_kernel<4, false, 1024, 7>;
        }
        if (xcols <= 1024 && dbg_kernels && !scores) return &stream_kernel<4, false, 1024, 0, TKSPMV_NBUF, true>;
        if (xcols <= 1024) return scores ? &stream_kernel<4, true, 1024> : &stream_kernel<4, false, 1024>;
        if (xcols <= 4096) return scores ? &stream_kernel<4, true, 4096> : &stream_kernel<4, false, 4096>;
        return scores ? &stream_kernel<4, true, 16384> : &stream_kernel<4, false, 16384>;
    }
    void launch_stream(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s, bool to_host = false) const {
        StreamParams P = stream_params(x);
        SelectParams S = select_params(out_idx, out_val);
        if (to_host) {
            S.host_out = h_res_dev;
            S.host_epoch = ++host_epoch;
            S.t_start = d_tstart;
        }
        ++launch_counter;
        hipLaunchKernelGGL(kernel_for(false), dim3(grid), dim3(block + 64), 0, s, P, S);
    }
    // One query through single_kernel (local thresholds, checked): the result is exact iff the status word / host flag says the
    // check passed; otherwise the caller runs launch_stream, whose result is exact on its own.
    void launch_single(const float *x, uint32_t *out_idx, float *out_val, hipStream_t s, bool to_host) const {
        StreamParams P = stream_params(x);
        SelectParams S = select_params(out_idx, out_val);
        if (to_host) {
            S.host_out = h_res_dev;
            S.host_epoch = ++host_epoch;
            S.t_start = d_tstart;
        }
        LocalParams G{};
        G.slots = d_lslots;
        G.used = d_lused;
        G.wg_prior = d_lprior;
        G.prior_block = reinterpret_cast<uint32_t *>(d_lprior + grid);
        G.status = d_lstatus;
        G.ready = d_lready;
        G.epoch = (uint32_t)(single_launches + 1u) | 0x80000000u;  // (never 0, never the previous launch's)
        G.mode = single_mode;
        G.beta = local_beta;
        G.trace = d_trace ? d_trace + (launch_counter % 4) * trace_words : nullptr;
        ++launch_counter;
        ++single_launches;
        if (pm.precision == Precision::F32C12) hipLaunchKernelGGL((single_kernel<7>), dim3(grid), dim3(512), 0, s, P, S, G);
        else hipLaunchKernelGGL((single_kernel<0>), dim3(grid), dim3(512), 0, s, P, S, G);
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
        if (approx_parts) {
            PartitionParams Q{};
            Q.scores = d_rscores;
            Q.rows = desc.rows;
            Q.per = (desc.rows + (uint32_t)desc.partitions - 1u) / (uint32_t)desc.partitions;  // ceil(N / P): host_spmv_bscsr.cpp:136
            Q.k_part = (uint32_t)(desc.k_per_partition > 0 ? desc.k_per_partition : desc.k);
            Q.kmin = R.kmin;
            Q.ovf_cand = R.ovf_cand;
            Q.ovf_count = R.ovf_count;
            Q.ovf_cap = ovf_cap;
            hipLaunchKernelGGL(partition_topk_kernel, dim3((uint32_t)desc.partitions), dim3(RADIX_THREADS), 0, s, Q);
        } else {
            (void)hipMemsetAsync(d_rhist, 0, 4 * 256 * 4, s);
            const uint32_t rgrid = std::max(1u, std::min(256u, (desc.rows + RADIX_THREADS * 4u - 1u) / (RADIX_THREADS * 4u)));
            for (int pass = 0; pass < 4; ++pass) hipLaunchKernelGGL(radix_hist_kernel, dim3(rgrid), dim3(RADIX_THREADS), 0, s, R, pass);
            hipLaunchKernelGGL(radix_filter_kernel, dim3(rgrid), dim3(RADIX_THREADS), 0, s, R);
        }
        SelectParams S = select_params(out_idx, out_val, 0);
        S.use_gmax = 0u;  // no threshold word in this path
        S.out_scale = 1.0f;  // the scores kernel already wrote final scores
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SEL_THREADS), 0, s, S);
    }
    void launch_scores(const float *x, hipStream_t s, float *dst = nullptr) const {
        StreamParams P = stream_params(x);
        if (dst) P.scores = dst;
        SelectParams S = select_params(d_out_idx, d_out_val);
        ++launch_counter;  // the stream copies rotate per query here too (cache-defeat mode)
        hipLaunchKernelGGL(kernel_for(true), dim3(grid), dim3(block + 64), 0, s, P, S);
    }
    void launch_select(uint32_t *out_idx, float *out_val, hipStream_t s, int set = 0) const {
        SelectParams S = select_params(out_idx, out_val, set);
        S.unit_inv_in = st[set].unit_inv;  // written by the (unfused) stream kernel of that query
        hipLaunchKernelGGL(select_kernel, dim3(1), dim3(SEL_THREADS), 0, s, S);
    }
    // Does the host-visible result block add up to the checksum its writer left behind the flag (result_checksum_term)? The block
    // is written with relaxed system-scope stores; the flag alone proves nothing by the memory model.
    bool result_block_complete(uint32_t epoch) const {
        const size_t k = (size_t)desc.k;
        const volatile uint32_t *r = h_res;
        // (word 2k + 5: the status of single_kernel's check. A FAILED check writes no payload -- the block still holds the previous
        //  query's list -- and its checksum covers the epoch and the status alone: summing the stale payload here never matched,
        //  and every failed check cost tkspmv_run its 2 s flag timeout before the repair)
        if (r[2 * k + 5] != 0u) return r[2 * k + 4] == epoch * 0x9E3779B1u + 0xBADC0DE5u;
        uint32_t sum = epoch * 0x9E3779B1u;
        for (size_t i = 0; i < k; ++i) sum += result_checksum_term(r[i], r[k + i], (uint32_t)i);
        return r[2 * k + 4] == sum;
    }
    // tkspmv_set_query enqueues the upload of x on the engine's stream and does not wait for it. A launch that reads d_x from
    // ANOTHER stream (tkspmv_enqueue with a caller's stream and dev_x = NULL) has no ordering against that copy -- the
    // engine's stream is non-blocking -- so it waits for the event recorded behind the copy.
    hipError_t order_x(const float *x, hipStream_t s) const {
        if (x != d_x || !x_pending || s == stream || bar_x) return hipSuccess;  // (bar_x: the host wrote x itself, nothing is enqueued)
        return hipStreamWaitEvent(s, ev2, 0);
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
    // (the bit-packed narrow fixed-point stream is a layout of TKSPMV_FIXED, not a precision of the API)
    out->precision = (pm.precision == Precision::FIXED20 || pm.precision == Precision::FIXED26) ? (int32_t)Precision::FIXED
                                                       : (pm.precision == Precision::F32C12 ? (int32_t)Precision::F32 : (int32_t)pm.precision);
    out->fixed_width = pm.fixed_width;
}

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int use_device(int device, std::string &err) {
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    return TKSPMV_OK;
}

Engine::~Engine() {
    if (!impl_) return;
    EngineImpl &m = *impl_;
    (void)hipSetDevice(m.device);
    if (m.stream) (void)hipStreamSynchronize(m.stream);
    void *bufs[] = {m.d_packets, m.d_pkt_row, m.d_part_first, m.d_part_count, m.d_x,
                    m.d_out_idx, m.d_out_val, m.d_scores,     m.d_stats,      m.d_done, m.d_trace, m.d_tickets,
                    m.d_tstart, m.d_verdict, m.d_wg_prior, m.d_rec_slots, m.d_rec_used, m.d_ovf_epoch, m.d_alias_idx, m.d_alias_val,
                    m.d_lslots,  m.d_lused, m.d_lprior, m.d_lstatus, m.d_wg_pace, m.d_wg_times, m.d_wg_sig, m.d_lready, m.d_pace_adapt};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    {
        EngineImpl::ExState &E = m.st[0];  // the other sets point into these blocks
        void *eb[] = {E.tau_g, E.gmax, E.ovf_count, E.wg_cand, E.ovf, E.unit_inv};
        for (void *b : eb)
            if (b) (void)hipFree(b);
    }
    if (m.h_x) (void)hipHostFree(m.h_x);
    if (m.h_res) (void)hipHostFree(m.h_res);
    if (m.h_verdict) (void)hipHostFree(m.h_verdict);
    for (size_t r = 1; r < m.d_replicas.size(); ++r) (void)hipFree(m.d_replicas[r]);
    for (size_t r = 1; r < m.d_sell_replicas.size(); ++r) (void)hipFree(m.d_sell_replicas[r]);
    {
        void *sb[] = {m.d_coo_col, m.d_coo_val, m.d_sell_packets, m.d_sell_rows, m.d_sell_part_first, m.d_sell_part_count, m.d_sell_part_slice0, m.d_multi_out_idx, m.d_multi_out_val, m.d_rscores, m.d_rhist};
        for (void *b : sb)
            if (b) (void)hipFree(b);
    }
    if (m.ev0) (void)hipEventDestroy(m.ev0);
    if (m.ev1) (void)hipEventDestroy(m.ev1);
    if (m.ev2) (void)hipEventDestroy(m.ev2);
    if (m.ev_fork) (void)hipEventDestroy(m.ev_fork);
    if (m.ev_join) (void)hipEventDestroy(m.ev_join);
    if (m.bside) {
        (void)hipStreamSynchronize(m.bside);
        (void)hipStreamDestroy(m.bside);
    }
    if (m.ev_bfork) (void)hipEventDestroy(m.ev_bfork);
    if (m.ev_bjoin) (void)hipEventDestroy(m.ev_bjoin);
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
    if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) == hipSuccess) return hipSuccess;
    if (opt("DEBUG_OCC")) fprintf(stderr, "[tkspmv] fine-grained allocation unavailable, using hipMalloc\n");
    (void)hipGetLastError();
    return hipMalloc(p, bytes);
}

// Small matrices (the shards of a strong-scaled run): a query streams in less time than one selection takes and than a
// device-wide threshold needs to form. They get 4 selector workgroups, partitions from 1-2 packets up (every wave streams:
// twice the loads in flight; wbscsr.hpp: min_packets_per_partition_for) and workgroup-local thresholds. With pacing by rank
// (create_impl) the same settings win up to LOCAL_MATRIX_PACKETS -- the headline's 1M rows included; beyond, the device-wide
// exchange is as fast and checks nothing. Returns the selector workgroups of a batch launch for this matrix and geometry
// (TKSPMV_SELECTORS overrides; the matrix unknown -- nnz = 0 --: 1). TKSPMV_SMALL_PACKETS: both limits (0: round 2's behaviour).
// (fp32 and fp16 values: every size, wbscsr.hpp; byte and fixed-point values keep round 4's limit -- their kernels are bound by
//  arithmetic, and there the local thresholds lose beyond it: Q1.7 bytes at 1M x 512 x 40, 154k packets: 33.0 against 24.7 us)
static uint64_t local_matrix_packets(int32_t precision) {
    if (const char *f = opt("SMALL_PACKETS")) return (uint64_t)atoll(f);
    return (precision == TKSPMV_F32 || precision == TKSPMV_F16) ? LOCAL_MATRIX_PACKETS : 100000ull;
}
static uint32_t small_matrix_settings(const tkspmv_desc &d, uint32_t grid, bool defer_capable, uint32_t C, bool *small_out) {
    const uint64_t packets_lb = d.nnz / (64u * (uint64_t)std::max(C, 1u));
    // (partitions with k <= k_per_partition: the union of the per-partition lists holds the global top-k -- the ordinary kernels'
    //  result IS the partitioned one, create_impl --, so such engines take the settings of an unpartitioned one: BASELINE configs[2])
    const bool plain_topk = d.partitions <= 1 || d.k <= (d.k_per_partition > 0 ? d.k_per_partition : d.k);
    const bool small = defer_capable && grid >= 64u && d.nnz != 0 && d.cols <= 1024u && d.impl == TKSPMV_IMPL_STREAM && d.multi_q == 0 &&
                       plain_topk && packets_lb <= local_matrix_packets(d.precision) && !opt("MULTI_Q");
    if (small_out) *small_out = small;
    uint32_t n = small ? 4u : 1u;
    if (const char *f = opt("SELECTORS")) n = (uint32_t)std::max(1, std::min(8, atoi(f)));
    if (!defer_capable || grid < 2u * n) n = 1u;
    return n;
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
        d.precision != TKSPMV_F16 && d.precision != TKSPMV_FIXED && d.precision != TKSPMV_Q1_7_F32) {
        err = "unknown precision";
        return TKSPMV_ERR_INVALID;
    }
    if (d.impl == 3) {  // (rounds 2-4: a resident kernel serving tkspmv_run; slower than a launch per query since round 4, removed in round 5)
        err = "impl 3 (the resident kernel) was removed: tkspmv_run through impl 0 (single_kernel) is faster";
        return TKSPMV_ERR_UNSUPPORTED;
    }
    if (d.impl != TKSPMV_IMPL_STREAM && d.impl != TKSPMV_IMPL_ROW_PER_LANE && d.impl != TKSPMV_IMPL_SCORES_SELECT) {
        err = "unknown impl (0 = stream, 1 = row per lane, 2 = scores + select)";
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
        // contains the global top-k, so the partitioned result IS the exact one computed by the ordinary kernels. With
        // k > k_per_partition (the FPGA design's lossy regime: 32 partitions x K = 8 lists answering k = 100) the engine
        // reproduces the reference's approximate union: partition_topk_kernel.
        const int kpp = d.k_per_partition > 0 ? d.k_per_partition : d.k;
        if (kpp > TKSPMV_MAX_K || d.partitions > 65535) {
            err = "partitions must be <= 65535 and k_per_partition <= 1024";
            return TKSPMV_ERR_INVALID;
        }
        m.approx_parts = d.k > kpp;
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
    // Small matrices (the shards of a strong-scaled run) get 4 selector workgroups, shorter partitions and -- decided below, once
    // the partitions are known -- workgroup-local thresholds: small_matrix_settings().
    bool small_matrix = false;
    m.n_sel_wg = small_matrix_settings(d, m.grid, defer_capable, C, &small_matrix);
    if (prepacked && prepacked->part_first.size() > (size_t)(m.grid - m.n_sel_wg) * waves_per_wg) m.n_sel_wg = 1u;
    const uint32_t n_stream_waves = (m.grid - (defer_capable ? m.n_sel_wg : 0u)) * waves_per_wg;
    // (TKSPMV_PARTITIONS_HINT: measurement aid of the load-only probe -- more partitions than waves; such an engine runs no queries)
    const uint32_t n_parts_hint = opt("PARTITIONS_HINT") ? (uint32_t)atoi(opt("PARTITIONS_HINT")) : n_stream_waves;
    if (prepacked) {
        // A matrix packed earlier (tkspmv_pack / a .tkspmv file): it must describe the same problem and must not have
        // more partitions than this launch geometry has streaming waves (the batch kernel gives every wave one).
        const PackedMatrix &q = *prepacked;
        const bool fixed_either = d.precision == TKSPMV_FIXED && (q.precision == Precision::FIXED || q.precision == Precision::FIXED20 || q.precision == Precision::FIXED26);
        const bool f32_either = d.precision == TKSPMV_F32 && (q.precision == Precision::F32 || q.precision == Precision::F32C12);
        if (q.rows != d.rows || q.cols != d.cols || (!fixed_either && !f32_either && q.precision != stream_precision(d.precision)) ||
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
        // The packer runs on the device by default (device_pack.hip: same bytes as the host packer, which stays available
        // with TKSPMV_DEVICE_PACK=0 and serves tkspmv_pack / the .tkspmv files).
        bool on_device = d.nnz > 0;
        if (const char *f = opt("DEVICE_PACK")) on_device = on_device && atoi(f) != 0;
        const uint32_t min_packets = min_packets_per_partition_for(d.nnz, C, d.cols);
        const auto t_pack = std::chrono::steady_clock::now();
        std::string perr;
        if (on_device) {
            DevicePacked dp;
            // (a multi-query engine packs the same COO a second time below: leave its columns and values in HBM until then)
            dp.keep_coo = d.cols <= SELL_XCOLS && (d.multi_q != 0 || d.impl == TKSPMV_IMPL_ROW_PER_LANE || opt("MULTI_Q"));
            perr = pack_wbscsr_device(d.rows, d.cols, d.nnz, d.row, d.col, d.val, stream_precision_of(d), C,
                                      n_parts_hint, min_packets, fixed_width_of(d), dp, kind);
            if (perr.empty()) {
                m.pm = std::move(dp.meta);
                m.d_packets = dp.d_packets;
                m.d_pkt_row = dp.d_pkt_row;
                m.d_coo_col = dp.d_col;
                m.d_coo_val = dp.d_val;
                m.packed_on_device = true;
            }
        } else {
            perr = pack_wbscsr(d.rows, d.cols, d.nnz, d.row, d.col, d.val, stream_precision_of(d), C, n_parts_hint, min_packets,
                               m.pm, kind, fixed_width_of(d));
        }
        if (!perr.empty()) {
            err = perr;
            return kind == 2 ? TKSPMV_ERR_NOT_SORTED : (perr.find("failed:") != std::string::npos ? TKSPMV_ERR_DEVICE : TKSPMV_ERR_INVALID);
        }
        m.pack_us = (uint32_t)std::min<long long>(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_pack).count(), 0xFFFFFFFFll);
    }
    fill_info(m.pm, d.k, &m.info);
    {
        // The packer's usual outcome is a uniform table -- every partition but the last holds packets_per_partition packets, back
        // to back --: the kernels then derive a wave's range from its partition number (no table loads at the head of a launch).
        const size_t np = m.pm.part_first.size();
        const uint32_t ppp = m.pm.packets_per_partition;
        bool uniform = np > 0 && ppp > 0 && (uint64_t)np * ppp <= 0xFFFFFFFFull && m.pm.part_count[np - 1] >= 1u && m.pm.part_count[np - 1] <= ppp;
        for (size_t p = 0; uniform && p < np; ++p)
            uniform = m.pm.part_first[p] == (uint32_t)(p * ppp) && (p + 1 == np || m.pm.part_count[p] == ppp);
        m.uni_ppp = uniform ? ppp : 0u;
        m.uni_last = uniform ? m.pm.part_count[np - 1] : 0u;
    }

    // Threshold-exchange geometry: at least k publishing groups are needed (tau = k-th largest published maximum);
    // if one group per workgroup is not enough, the waves of a workgroup are split into up to 8 groups.
    // Counted on the workgroups that stream in a sequence launch (grid - 1: workgroup 0 selects). Aim for 4k groups
    // where the 1024-word limit of the exchange allows: with k close to the number of groups the k-th largest maximum
    // is a weak bound (k = 256 on 511 groups ran 2x slower than k = 100).
    const uint32_t n_pub_wg = m.grid - (defer_capable ? m.n_sel_wg : 0u);
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
        const uint32_t n_wg = m.grid - (defer_capable ? m.n_sel_wg : 0u), n_parts = (uint32_t)m.pm.part_first.size();
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
    if (!m.packed_on_device) {  // (the device packer has left the stream and its row table in HBM already)
        HIP_TRY(hipMalloc((void **)&m.d_packets, stream_bytes));
        HIP_TRY(hipMalloc((void **)&m.d_pkt_row, std::max<size_t>(m.pm.pkt_row.size(), 1) * 4));
        if (m.pm.stream_bytes())
            HIP_TRY(hipMemcpy(m.d_packets, m.pm.packets.data(), m.pm.stream_bytes(), hipMemcpyHostToDevice));
        if (!m.pm.pkt_row.empty())
            HIP_TRY(hipMemcpy(m.d_pkt_row, m.pm.pkt_row.data(), m.pm.pkt_row.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc((void **)&m.d_part_first, std::max<size_t>(m.pm.part_first.size(), 1) * 4));
    HIP_TRY(hipMalloc((void **)&m.d_part_count, std::max<size_t>(m.pm.part_count.size(), 1) * 4));
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
    if (const char *f = opt("HOST_PATH")) m.host_path = atoi(f);
    {
        // Host stores into device memory: only where the runtime reports a large BAR, and only after a round trip has shown that a
        // value stored by the CPU is the value a device-side copy reads back.
        int large_bar = 0;
        if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess) large_bar = 0;
        (void)hipGetLastError();
        bool want = large_bar != 0 && m.host_path != 0;
        if (const char *f = opt("BAR_X")) want = want && atoi(f) != 0;
        if (want) {
            uint32_t *reg = nullptr;  // (the attribute is returned through the int pointer as an address)
            if (hipDeviceGetAttribute(reinterpret_cast<int *>(&reg), hipDeviceAttributeHdpMemFlushCntl, dev) != hipSuccess) reg = nullptr;
            (void)hipGetLastError();
            m.hdp_flush = reg;
            want = reg != nullptr;
        }
        if (want) {
            // Four rounds with changing values: the CPU stores two words and flushes the HDP, a KERNEL loads them (a copy engine
            // reading them back proves nothing about what a kernel's loads see behind its caches) -- after a kernel of an
            // earlier round has had the old values in its caches.
            uint32_t *d_back = nullptr;
            HIP_TRY(hipMalloc((void **)&d_back, 8));
            HIP_TRY(hipMemset(m.d_x, 0, (size_t)d.cols * 4));
            HIP_TRY(hipDeviceSynchronize());
            volatile uint32_t *px = reinterpret_cast<volatile uint32_t *>(m.d_x);
            bool ok = true;
            for (uint32_t round = 0; round < 4u && ok; ++round) {
                const uint32_t a = 0x5A17C0DEu + 0x01010101u * round, b = (0xC0DE5A17u ^ d.cols) + 0x00010001u * round;
                px[0] = a;
                px[d.cols - 1] = b;
                m.flush_hdp();
                hipLaunchKernelGGL(bar_check_kernel, dim3(1), dim3(64), 0, nullptr, reinterpret_cast<const uint32_t *>(m.d_x), d.cols - 1u, d_back);
                uint32_t back[2] = {0, 0};
                HIP_TRY(hipMemcpy(back, d_back, 8, hipMemcpyDeviceToHost));
                ok = back[0] == (d.cols == 1 ? b : a) && back[1] == b;
            }
            m.bar_x = ok;
            (void)hipFree(d_back);
            HIP_TRY(hipMemset(m.d_x, 0, (size_t)d.cols * 4));
            HIP_TRY(hipDeviceSynchronize());
        }
    }
    if (m.host_path) {  // pinned staging copy of x and the host-visible result block (optional: the plain path needs neither)
        if (const char *f = opt("HOST_X")) m.host_x_direct = std::string(f) == "direct" || std::string(f) == "direct_nc";
        if (const char *f = opt("RUN_EVENTS")) m.run_events = atoi(f) != 0;
        {
            const char *f = opt("HOST_X");
            const unsigned flags = (f && std::string(f) == "direct_nc") ? (hipHostMallocMapped | hipHostMallocNonCoherent) : hipHostMallocMapped;
            if (hipHostMalloc((void **)&m.h_x, (size_t)d.cols * 4, flags) != hipSuccess) m.h_x = nullptr;
            if (m.h_x && hipHostGetDevicePointer((void **)&m.h_x_dev, m.h_x, 0) != hipSuccess) m.host_x_direct = false;
        }
        if (hipHostMalloc((void **)&m.h_res, ((size_t)2 * d.k + 16) * 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void **)&m.h_res_dev, m.h_res, 0) != hipSuccess) {
            if (m.h_res) (void)hipHostFree(m.h_res);
            m.h_res = m.h_res_dev = nullptr;
        }
        (void)hipGetLastError();
        if (m.h_res) std::memset(m.h_res, 0, ((size_t)2 * d.k + 16) * 4);
    }
    HIP_TRY(hipMalloc((void **)&m.d_out_idx, (size_t)d.k * 4));
    HIP_TRY(hipMalloc((void **)&m.d_out_val, (size_t)d.k * 4));
    HIP_TRY(hipMalloc((void **)&m.d_stats, 32 * 8));
    m.collect_stats = opt("STATS") != nullptr;
    HIP_TRY(malloc_exchange((void **)&m.d_tstart, 128));
    HIP_TRY(hipMemset(m.d_tstart, 0, 128));
    HIP_TRY(malloc_exchange((void **)&m.d_done, 9 * 128));
    HIP_TRY(hipMemset(m.d_done, 0, 9 * 128));
    // Fused tail / deferred selection: one workgroup (block + 64 threads) must hold every slot in SEL_PER_THREAD
    // registers per thread.
    m.fused = (uint64_t)m.grid * WG_SLOTS <= (uint64_t)SEL_PER_THREAD * (m.block + 64);
    m.can_defer = defer_capable;
    if (const char *f = opt("FUSED")) m.fused = m.fused && atoi(f) != 0;
    m.can_batch = m.can_defer && m.n_sets != 0u && m.xcols <= 1024u && (C == 4u || (C == 8u && d.precision == TKSPMV_F32));  // larger x: two workgroups no longer fit a CU
    if (const char *f = opt("BATCH")) m.can_batch = m.can_batch && atoi(f) != 0;
    // Large k: the scores + radix-select path wherever the threshold exchange is off or next to useless (k above
    // 3/8 of the publishing groups: measured cross-over on the BASELINE matrix, tools/k_probe.py). TKSPMV_RADIX=0/1 forces.
    m.use_radix = m.n_sets == 0u || (uint64_t)d.k * 8u > (uint64_t)m.n_groups_pub * 3u || d.impl == TKSPMV_IMPL_SCORES_SELECT;
    if (const char *f = opt("RADIX")) m.use_radix = atoi(f) != 0;
    if (m.approx_parts) m.use_radix = true;  // scores, then the per-partition selection instead of the radix select
    if (m.use_radix) {
        m.can_defer = m.can_batch = false;
        m.fused = false;
        const size_t n = std::max<size_t>(d.rows, 1);
        HIP_TRY(hipMalloc((void **)&m.d_rscores, n * 4));
        std::vector<float> ninf(n, -std::numeric_limits<float>::infinity());
        HIP_TRY(hipMemcpy(m.d_rscores, ninf.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_rhist, 4 * 256 * 4));
    }
    // (every streaming wave of a sequence launch owns ONE partition: an engine packed with more -- the load-only probe's
    //  TKSPMV_PARTITIONS_HINT -- would silently skip the rest)
    const bool partitions_fit = m.pm.part_first.size() <= (size_t)n_stream_waves;
    if (!partitions_fit) m.can_batch = m.can_defer = false;
    // Multi-query passes (desc.multi_q; TKSPMV_MULTI_Q overrides): a second copy of the matrix in the wave-sliced ELL layout.
    {
        int mq = d.multi_q;
        if (d.impl == TKSPMV_IMPL_ROW_PER_LANE && mq == 0) mq = 1;
        if (const char *f = opt("MULTI_Q")) mq = atoi(f);
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
        // One or two queries per pass: a launch makes several passes (MULTI_PASSES; default: as many as the MULTI_Q_MAX queries of
        // a group allow -- 8 or 4), each pass with its own queries and exchange-state sets. (4 and 8 per pass: one pass per launch.)
        m.multi_group = mq;
        if (mq == 1 || mq == 2) {
            int passes = MULTI_Q_MAX / mq;
            if (const char *f = opt("MULTI_PASSES")) passes = std::min(std::max(atoi(f), 1), passes);
            m.multi_group = passes * mq;
        }
        // 8 queries per pass need 91 registers: with 9 waves per workgroup only one workgroup fits a CU (the dispatcher wants
        // 6 waves on one SIMD for two); with 8 waves -- 7 streaming + the server -- two fit at up to 128 registers.
        m.multi_stream_waves = (m.multi_q >= 8 && waves_per_wg == 8u) ? 7u : waves_per_wg;
        // (k above half of the groups: the threshold is next to useless -- k = 1000 on 1024 groups measured 3.6 ms per query
        // through the multi-query kernel against 0.2 ms one query per pass; such engines keep the ordinary sequence)
        // The selector workgroups of a multi-query launch have multi_stream_waves * 64 + 64 threads and hold SEL_PER_THREAD
        // slots each in registers: every slot of the grid must fit (with 7 streaming waves that is 4096 slots = 512
        // workgroups, fewer than the 576 the ordinary launch geometry allows).
        m.can_multi = mq > 0 && !m.use_radix && m.can_defer && m.n_sets != 0u && d.cols <= SELL_XCOLS &&
                      (d.precision == TKSPMV_F32 || d.precision == TKSPMV_Q1_7_F32) && m.pm.nnz > 0 &&
                      m.grid > 2u * (uint32_t)MULTI_Q_MAX && (uint32_t)d.k * 2u <= m.n_groups_pub &&
                      (uint64_t)m.grid * WG_SLOTS <= (uint64_t)SEL_PER_THREAD * (m.multi_stream_waves * 64u + 64u) &&
                      // (the kernel stages x with TWO words per thread -- multi_kernel's XI --: a workgroup narrower than 512 threads
                      //  would leave the columns from 2 x blockDim on unwritten in LDS and score against garbage, ADVICE r4;
                      //  such geometries -- threads_per_wg <= 384 -- keep the ordinary sequence)
                      2u * (m.multi_stream_waves * 64u + 64u) >= SELL_XCOLS;
    }
    if (m.can_multi) {
        // the first multi_group workgroups of a multi-query launch are its selectors (one per query of the previous launch), the others stream
        const uint32_t n_multi_waves = (m.grid - (uint32_t)m.multi_group) * m.multi_stream_waves;
        SellMatrix sm;
        std::string perr;
        const auto t_sell = std::chrono::steady_clock::now();
        const SellValues sv = d.precision == TKSPMV_Q1_7_F32 ? SellValues::Q1_7_RND : SellValues::F32;
        if (prepacked) {  // no COO at hand: decode the packed matrix (byte values decode to exactly representable floats)
            std::vector<uint32_t> r, c;
            std::vector<float> v;
            decode_wbscsr(*prepacked, r, c, v);
            perr = pack_wsell(d.rows, d.cols, r.size(), r.data(), c.data(), v.data(), n_multi_waves, sm, sv);
        } else if (m.packed_on_device) {  // plan on the host, fill on the device (device_pack.hip; same bytes)
            DeviceSell ds;
            perr = pack_wsell_device(d.rows, d.cols, d.nnz, d.row, d.col, d.val, n_multi_waves, sv, m.d_coo_col, m.d_coo_val, ds);
            sm = std::move(ds.meta);
            m.d_sell_packets = ds.d_packets;
        } else {
            perr = pack_wsell(d.rows, d.cols, d.nnz, d.row, d.col, d.val, n_multi_waves, sm, sv);
        }
        m.sell_pack_us = (uint32_t)std::min<long long>(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_sell).count(), 0xFFFFFFFFll);
        m.sell_packet_bytes = sm.packet_bytes;
        m.sell_byte_values = sv == SellValues::Q1_7_RND;
        m.sell_c12 = sm.cw_bits == 12;
        if (!perr.empty()) {
            err = perr;
            return perr.find("failed:") != std::string::npos ? TKSPMV_ERR_DEVICE : TKSPMV_ERR_INVALID;
        }
        m.sell_parts = (uint32_t)sm.part_first.size();
        m.sell_bytes = sm.stream_bytes() + sm.slice_rows.size() * 4 + (uint64_t)m.sell_parts * 12;
        if (!m.d_sell_packets) {
            HIP_TRY(hipMalloc((void **)&m.d_sell_packets, std::max<size_t>(sm.stream_bytes(), 256)));
            HIP_TRY(hipMemcpy(m.d_sell_packets, sm.packets.data(), sm.stream_bytes(), hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc((void **)&m.d_sell_rows, sm.slice_rows.size() * 4));
        HIP_TRY(hipMemcpy(m.d_sell_rows, sm.slice_rows.data(), sm.slice_rows.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_sell_part_first, (size_t)m.sell_parts * 4));
        HIP_TRY(hipMalloc((void **)&m.d_sell_part_count, (size_t)m.sell_parts * 4));
        HIP_TRY(hipMalloc((void **)&m.d_sell_part_slice0, (size_t)m.sell_parts * 4));
        HIP_TRY(hipMemcpy(m.d_sell_part_first, sm.part_first.data(), (size_t)m.sell_parts * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m.d_sell_part_count, sm.part_count.data(), (size_t)m.sell_parts * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(m.d_sell_part_slice0, sm.part_slice0.data(), (size_t)m.sell_parts * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&m.d_multi_out_idx, 2 * (size_t)MULTI_Q_MAX * d.k * 4));
        HIP_TRY(hipMalloc((void **)&m.d_multi_out_val, 2 * (size_t)MULTI_Q_MAX * d.k * 4));
        HIP_TRY(hipStreamCreateWithFlags(&m.side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m.ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m.ev_join, hipEventDisableTiming));
        if (const char *f = opt("MULTI_CHAINS")) m.multi_chains = atoi(f) >= 2 ? 2 : 1;
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
    if (m.d_coo_col) (void)hipFree(m.d_coo_col);
    if (m.d_coo_val) (void)hipFree(m.d_coo_val);
    m.d_coo_col = nullptr;
    m.d_coo_val = nullptr;
    HIP_TRY(malloc_exchange((void **)&m.d_tickets, 2 * BATCH_MAX * 32 * 4));  // (one block per launch parity: EngineImpl::bside)
    HIP_TRY(hipMemset(m.d_tickets, 0, 2 * BATCH_MAX * 32 * 4));
    HIP_TRY(malloc_exchange((void **)&m.d_verdict, 512));
    HIP_TRY(hipMemset(m.d_verdict, 0, 512));
    // the host's side of the verdicts (EngineImpl::settle); without it every local launch keeps its repair launch in the stream
    if (const char *f = opt("REPAIR")) m.repair_by_host = std::string(f) != "stream";
    if (hipHostMalloc((void **)&m.h_verdict, (size_t)EngineImpl::VERDICT_RING * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void **)&m.h_verdict_dev, m.h_verdict, 0) != hipSuccess) {
        if (m.h_verdict) (void)hipHostFree(m.h_verdict);
        m.h_verdict = m.h_verdict_dev = nullptr;
        (void)hipGetLastError();
    } else {
        std::memset(m.h_verdict, 0, (size_t)EngineImpl::VERDICT_RING * 8);
    }
    {
        // Workgroup-local thresholds pay when they practically never fail the selection's check (a failure costs the query a
        // second pass). The workgroup's threshold is the smallest of its waves' words; it exceeds the k-th best score when EVERY
        // streaming wave's word does (a = waves with a partition per workgroup, at least). A wave's word is its best packet
        // maximum (mode 1: lambda = k / partitions of the k best rows sit in a wave: ~lambda) or its second best (mode 2: two of
        // them in different packets, ~lambda^2 / 2; a single-packet partition has no second one). Mode 1 forms the threshold a
        // packet earlier: preferred where it is safe. And a workgroup keeps its 8 best rows: 9 of the k best in one workgroup
        // fail too (Poisson tail at k / workgroups).
        const double n_parts = (double)std::max<size_t>(m.pm.part_first.size(), 1), n_wg = (double)(m.grid - m.n_sel_wg);
        const double lam = std::min(1.0, (double)d.k / n_parts);
        const double per_part = (double)m.pm.n_packets / n_parts;
        const double a = std::max(1.0, std::floor(n_parts / n_wg));
        const double lw = (double)d.k / n_wg;
        double tail9 = 1.0, term = std::exp(-lw);
        for (int i = 0; i < 9; ++i) { tail9 -= term; term *= lw / (i + 1); }
        tail9 = std::max(tail9, 0.0);
        const double p1 = n_wg * (std::pow(lam, a) + tail9);
        const double p2 = n_wg * (std::pow(per_part < 1.5 ? lam : std::min(1.0, 0.5 * lam * lam), a) + tail9);
        m.single_mode = m.grid > 512u ? 0u : (p1 <= 1e-4 ? 1u : (p2 <= 1e-3 ? 2u : 0u));  // (512: select_local's record count)
        // Back-to-back queries: beyond ~1.3M rows the device-wide exchange is as fast and checks nothing (small_matrix). ONE query per
        // launch has no such cross-over -- without a previous query in flight the exchange's threshold arrives a third into the
        // partition, a carried local one is there from the first packet: 58.4 against 67.5 us at 3M rows -- so single_kernel serves
        // tkspmv_run wherever the failure estimate allows it.
        m.use_local = small_matrix ? m.single_mode : 0u;
        if (opt("DEBUG_OCC"))
            fprintf(stderr, "[tkspmv] small matrix %d: %u selector workgroups, %.0f partitions of %.1f packets, %.0f per workgroup; local thresholds fail with p = %.2e (mode 1) / %.2e (mode 2): mode %u\n",
                    (int)small_matrix, m.n_sel_wg, n_parts, per_part, a, p1, p2, m.use_local);
    }
    HIP_TRY(malloc_exchange((void **)&m.d_wg_prior, ((size_t)m.grid + 32) * 4));  // priors | countdown words
    HIP_TRY(hipMemset(m.d_wg_prior, 0, ((size_t)m.grid + 32) * 4));
    if (!opt("SIGNATURES") || atoi(opt("SIGNATURES")) != 0) {
        HIP_TRY(hipMalloc((void **)&m.d_wg_sig, (size_t)m.grid * 8 * 4));
        HIP_TRY(hipMemset(m.d_wg_sig, 0, (size_t)m.grid * 8 * 4));
    }
    if (const char *f = opt("LOCAL_BETA")) m.local_beta = (float)atof(f);
    if (const char *f = opt("LOCAL")) m.use_local = m.single_mode = m.grid > 512u ? 0u : (uint32_t)std::max(0, std::min(2, atoi(f)));
    // Pacing by rank, with local thresholds only (with the device-wide exchange the cold phase of every query is governor enough, §3.0):
    // the longer the partitions, the longer the pause (size sweeps on two boxes, tools/ab_rank.sh: best at 0 / 1 / 2 units of 256 cycles
    // up to 575k / 830k / 1.3M rows of 20 non-zeros).
    // The pause follows the packet's size (fp32 values with 12-bit columns, 1408 bytes: 2 units; fp16, 896 bytes: 1 -- 15.4 us per query
    // at 1M rows against 16.3 at 2 units and 16.7 with the device-wide exchange).
    // (units of 128 cycles per level)
    // (round 5, with the pause behind the packet's arithmetic: from ~330k rows up a pause of ONE unit per level pays -- 500k rows: 9.03
    //  against 9.86 us per query unpaced, 10.0 at two units; 250k rows: 5.8 either way -- and tkspmv_create's measurement may still
    //  choose none)
    if (m.use_local) m.pace_quads = m.pm.n_packets <= 25000u ? 0u : (m.pm.n_packets <= 45000u ? 1u : (m.pm.n_packets <= 65000u ? 2u : 2u * std::max(1u, (m.pm.packet_bytes + 352u) / 704u)));
    if (const char *f = opt("PACE_LEVELS")) m.pace_levels = (uint32_t)std::max(1, std::min(8, atoi(f)));
    if (const char *f = opt("PACE")) m.pace_quads = (uint32_t)std::max(0, std::min(32, atoi(f)));
    if (const char *f = opt("PACE_BASE")) m.pace_base = (uint32_t)std::max(0, std::min(64, atoi(f)));
    if (const char *f = opt("PACE_CARRY")) m.pace_carry = atoi(f) != 0;
    if (const char *f = opt("PACE_ADAPT")) m.pace_adapt_on = atoi(f) != 0;
    if (const char *f = opt("PACE_PERIOD")) m.pace_period_ns = (uint32_t)std::max(0, std::min(10000000, atoi(f)));
    if (m.use_local && m.pace_quads != 0u) {
        HIP_TRY(hipMalloc((void **)&m.d_wg_pace, (size_t)m.grid * 4));
        HIP_TRY(malloc_exchange((void **)&m.d_pace_adapt, 128));
        HIP_TRY(hipMemset(m.d_pace_adapt, 0, 128));
        HIP_TRY(hipMemset(m.d_wg_pace, 0, (size_t)m.grid * 4));
    }
    // single_kernel serves tkspmv_run where the engine streams with local thresholds: fp32 values, 4 entries per lane, x of at most
    // 1024 columns, at most 512 workgroups (select_local's first cut), one partition per wave of ITS launch (8 waves x grid).
    m.can_single = m.single_mode != 0u && d.precision == TKSPMV_F32 && C == 4u && m.xcols <= 1024u && m.grid <= 512u && m.block == 512u &&
                   d.impl == TKSPMV_IMPL_STREAM && !m.use_radix && m.fused && m.host_path && m.h_res != nullptr &&
                   m.pm.part_first.size() <= (size_t)m.grid * 8u;
    if (const char *f = opt("SINGLE")) m.can_single = m.can_single && atoi(f) != 0;
    if (m.can_single) {
        HIP_TRY(hipMalloc((void **)&m.d_lslots, (size_t)m.grid * SINGLE_REC_STRIDE * 8));  // (a 128-byte line per workgroup: single_kernel's selector)
        HIP_TRY(hipMemset(m.d_lslots, 0xFF, (size_t)m.grid * SINGLE_REC_STRIDE * 8));
        // workgroup 0 selects (LocalParams::ready) where the partitions fit the grid's other workgroups; option SINGLE_SELECTOR=0:
        // round 4's scheme (every workgroup streams, the one that draws the last ticket selects)
        if (m.pm.part_first.size() <= (size_t)(m.grid - 1u) * 8u && m.grid >= 2u && (!opt("SINGLE_SELECTOR") || atoi(opt("SINGLE_SELECTOR")) != 0)) {
            HIP_TRY(malloc_exchange((void **)&m.d_lready, (size_t)m.grid * 4));
            HIP_TRY(hipMemset(m.d_lready, 0, (size_t)m.grid * 4));
        }
        HIP_TRY(hipMalloc((void **)&m.d_lused, (size_t)m.grid * 4));
        HIP_TRY(hipMemset(m.d_lused, 0, (size_t)m.grid * 4));
        HIP_TRY(malloc_exchange((void **)&m.d_lprior, ((size_t)m.grid + 32) * 4));
        HIP_TRY(hipMemset(m.d_lprior, 0, ((size_t)m.grid + 32) * 4));
        HIP_TRY(malloc_exchange((void **)&m.d_lstatus, 128));
        HIP_TRY(hipMemset(m.d_lstatus, 0, 128));
    }
    {
        // Exchange-state sets: one block per field, set s at s strides (the batch kernel addresses them that way). The batch kernel
        // runs one set per query of a launch; every set is small -- published maxima, a threshold word, one slot per wave --
        // except for the overflow list, which must be able to hold EVERY row (a degenerate query -- x = 0, all scores equal --
        // makes every row a candidate and the result must still be exact): 8 bytes per row. Round 3 kept one list per set: 256 MB
        // at 1M rows, 2.6 GB at 10M. Engines of the batch kernel now keep FOUR, which the queries of a launch share round robin
        // under flow control (BatchParams::ovf_epoch), and the general path of the selection needs no scratch copy any more
        // (select_body): 32 MB at 1M rows, 320 MB at 10M. Multi-query engines keep one list per set (their groups' selections are
        // owed across launches).
        if (const char *f = opt("BATCH_MAX")) m.batch_max = std::max(1, std::min(BATCH_MAX, atoi(f)));
        const int n_sets_alloc = m.can_multi ? EngineImpl::N_STATE : (m.can_batch ? m.batch_max : (m.can_defer ? 2 : 1));
        const size_t ns = (size_t)n_sets_alloc;
        // (engines that stream with checked local thresholds use the lists for repairs and behind a closed gate only: two; a power of two)
        int want_lists = m.use_local ? 2 : 4;
        if (const char *f = opt("OVF_LISTS")) want_lists = atoi(f);
        // (the deferred scheme streams query i + 1 into one set while workgroup 0 selects query i from the other: its two sets need
        //  private lists -- with OVF_LISTS=1 they shared one, the stream appended while the selection read and reset it)
        if (m.can_defer) want_lists = std::max(want_lists, 2);
        want_lists = std::min(std::min(want_lists, 4), n_sets_alloc);
        m.ovf_lists = m.can_multi ? (uint32_t)n_sets_alloc : (want_lists >= 4 ? 4u : (want_lists >= 2 ? 2u : 1u));
        const size_t nl = m.ovf_lists;
        EngineImpl::ExState &E0 = m.st[0];
        HIP_TRY(malloc_exchange((void **)&E0.gmax, ns * EngineImpl::GMAX_WORDS * 4));
        HIP_TRY(hipMemset(E0.gmax, 0, ns * EngineImpl::GMAX_WORDS * 4));
        HIP_TRY(malloc_exchange((void **)&E0.tau_g, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMemset(E0.tau_g, 0, ns * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMalloc((void **)&E0.ovf_count, nl * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMemset(E0.ovf_count, 0, nl * EngineImpl::STATE_WORD_STRIDE * 4));
        HIP_TRY(hipMalloc((void **)&E0.wg_cand, ns * m.grid * WG_SLOTS * 8));
        HIP_TRY(hipMemset(E0.wg_cand, 0xFF, ns * m.grid * WG_SLOTS * 8));
        HIP_TRY(hipMalloc((void **)&E0.ovf, nl * m.ovf_cap * 8));
        const size_t n_unit = std::max(ns, 2 * (size_t)BATCH_MAX);  // (the batch kernel's scale words: one block per launch parity)
        HIP_TRY(hipMalloc((void **)&E0.unit_inv, n_unit * EngineImpl::STATE_WORD_STRIDE * 4));
        std::vector<float> ones(n_unit * EngineImpl::STATE_WORD_STRIDE, 1.0f);
        HIP_TRY(hipMemcpy(E0.unit_inv, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
        for (int si = 1; si < n_sets_alloc; ++si) {
            EngineImpl::ExState &E = m.st[si];
            E.gmax = E0.gmax + (size_t)si * EngineImpl::GMAX_WORDS;
            E.tau_g = E0.tau_g + (size_t)si * EngineImpl::STATE_WORD_STRIDE;
            E.ovf_count = E0.ovf_count + ((size_t)si % nl) * EngineImpl::STATE_WORD_STRIDE;
            E.wg_cand = E0.wg_cand + (size_t)si * m.grid * WG_SLOTS;
            E.ovf = E0.ovf + ((size_t)si % nl) * m.ovf_cap;
            E.unit_inv = E0.unit_inv + (size_t)si * EngineImpl::STATE_WORD_STRIDE;
        }
        HIP_TRY(malloc_exchange((void **)&m.d_ovf_epoch, 128));
        HIP_TRY(hipMemset(m.d_ovf_epoch, 0, 128));
        if (m.can_batch) {
            // the workgroups' records of local mode, one block per set; the scratch results of queries that share the last one's buffer
            // (twice BATCH_MAX sets each: one block per launch parity, EngineImpl::bside)
            const size_t nr = 2 * (size_t)BATCH_MAX;
            HIP_TRY(hipMalloc((void **)&m.d_rec_slots, nr * m.grid * WG_SLOTS * 8));
            HIP_TRY(hipMemset(m.d_rec_slots, 0xFF, nr * m.grid * WG_SLOTS * 8));
            HIP_TRY(hipMalloc((void **)&m.d_rec_used, nr * m.grid * 4));
            HIP_TRY(hipMemset(m.d_rec_used, 0, nr * m.grid * 4));
            HIP_TRY(hipMalloc((void **)&m.d_alias_idx, nr * d.k * 4));
            HIP_TRY(hipMalloc((void **)&m.d_alias_val, nr * d.k * 4));
            if (m.use_local) {
                if (const char *f = opt("OVERLAP")) m.overlap_launches = atoi(f) != 0;
                if (m.overlap_launches) {  // (the second stream exists only where it is used: every stream is one more for a device-wide wait to visit)
                    HIP_TRY(hipStreamCreateWithFlags(&m.bside, hipStreamNonBlocking));
                    HIP_TRY(hipEventCreateWithFlags(&m.ev_bfork, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&m.ev_bjoin, hipEventDisableTiming));
                }
            }
        }
        m.info.state_bytes = ns * (EngineImpl::GMAX_WORDS * 4 + 2 * EngineImpl::STATE_WORD_STRIDE * 4 + (uint64_t)m.grid * WG_SLOTS * 8) +
                             nl * ((uint64_t)m.ovf_cap * 8 + EngineImpl::STATE_WORD_STRIDE * 4) +
                             (m.can_batch ? 2 * (uint64_t)BATCH_MAX * ((uint64_t)m.grid * WG_SLOTS * 8 + (uint64_t)m.grid * 4) : 0);
    }
    HIP_TRY(hipMemset(m.d_stats, 0, 32 * 8));
    m.collect_stamps = opt("STAMPS") != nullptr;
    if (opt("WG_TIMES") && m.use_local) {
        HIP_TRY(hipMalloc((void **)&m.d_wg_times, (size_t)(BATCH_MAX + 1) * m.grid * 8));
        HIP_TRY(hipMemset(m.d_wg_times, 0, (size_t)(BATCH_MAX + 1) * m.grid * 8));
    }
    if (opt("TRACE")) {
        m.trace_words = (size_t)(m.grid + 1) * 9 * 8;
        HIP_TRY(hipMalloc((void **)&m.d_trace, m.trace_words * 4 * 8));
        HIP_TRY(hipMemset(m.d_trace, 0, m.trace_words * 4 * 8));
    }
    HIP_TRY(hipMemset(m.d_out_idx, 0, (size_t)d.k * 4));
    HIP_TRY(hipMemset(m.d_out_val, 0, (size_t)d.k * 4));
    m.dbg_kernels = m.collect_stats || m.collect_stamps || m.d_trace != nullptr;
    if (m.dbg_kernels && (d.precision != TKSPMV_F32 || C != 4u || m.xcols > 1024u))
        fprintf(stderr, "[tkspmv] tracing / statistics hooks exist in the fp32, 4-entries-per-lane, <= 1024-column kernels only: "
                        "this engine runs without them\n");

    if (opt("DEBUG_OCC")) {
        int n1 = -1, n2 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, reinterpret_cast<const void *>(m.kernel_for(false)), (int)m.block + 64, 0);
        if (m.can_batch)
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, reinterpret_cast<const void *>(m.batch_kernel_for()), (int)m.block + 64, 0);
        fprintf(stderr, "[tkspmv] workgroups per CU by the runtime's occupancy calculator: stream kernel %d, batch kernel %d; LDS per CU %zu\n",
                n1, n2, (size_t)prop.maxSharedMemoryPerMultiProcessor);
    }
    m.info.precision = d.precision;  // the API's enum (fill_info reports the value type of the stream)
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
    m.info.pack_us = m.pack_us;
    m.info.multi_pack_us = m.sell_pack_us;
    m.info.pack_on_device = m.packed_on_device ? 1u : 0u;
    m.info.claim_sets = 0u;
    // ---- pacing, tuned on THIS box for THIS matrix (round 5) ----------------------------------------------------------------------
    // The pause per packet of the workgroups that lead the field (BatchParams::pace_quads x pace_levels) decides whether the eight
    // XCDs share the memory system evenly -- unpaced, four of them stream a query in 12 us and the other four in 22-26
    // (tools/wg_times.py) -- and its best setting moves with the box: quantum 2 over 6 eighths on the pool's fast boxes (16.3 us per
    // query, 17.6 at round 4's 4 over 3), 2 over 7 on its slow ones (17.4 against 18.9 and 20.0). So the engine measures: a handful of
    // settings, three launches of 32 synthetic queries each, twice through, behind 64 launches of warm-up: ~55 ms of tkspmv_create.
    if (m.use_local && m.can_batch && m.pace_quads != 0u && !opt("PACE") && !opt("PACE_LEVELS") && !opt("PACE_BASE") &&
        (!opt("AUTOTUNE") || atoi(opt("AUTOTUNE")) != 0)) {  // (every packet size since the pause sits behind the arithmetic: fp16 and the fixed-point streams gain as fp32 does)
        // (launches of 32 queries up to the headline's size; fewer queries per launch on larger matrices, so that the measurement stays
        //  at ~150 ms of streaming: 10M rows -- 5.3 ms per launch of 32 -- would spend 1.3 s on it)
        const int nq = std::max(4, std::min(std::min((int)BATCH_MAX, m.batch_max), (int)(32ull * 80000ull / std::max<uint64_t>(m.pm.n_packets, 1))));
        std::vector<float> hx((size_t)nq * d.cols);
        uint32_t lcg = 0x1234567u;
        for (float &v : hx) {
            lcg = lcg * 1664525u + 1013904223u;
            v = (float)(lcg >> 8) * (1.0f / 16777216.0f);
        }
        float *d_tx = nullptr;
        HIP_TRY(hipMalloc((void **)&d_tx, hx.size() * 4));
        HIP_TRY(hipMemcpy(d_tx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        std::vector<const float *> xs(nq);
        std::vector<uint32_t *> oi(nq, m.d_out_idx);
        std::vector<float *> ov(nq, m.d_out_val);
        for (int i = 0; i < nq; ++i) xs[i] = d_tx + (size_t)i * d.cols;
        static const uint32_t cand[][2] = {{2, 6}, {2, 7}, {3, 5}, {4, 4}, {2, 5}, {4, 3}, {1, 6}, {1, 4}, {0, 3}};  // ({0, .}: unpaced)
        constexpr int NC = (int)(sizeof(cand) / sizeof(cand[0]));
        float best_ms[NC];
        for (float &b : best_ms) b = 1e30f;
        const auto t_tune = std::chrono::steady_clock::now();
        const bool adapt_was = m.pace_adapt_on;
        m.pace_adapt_on = false;  // (the periods are measured as they are given)
        // (a GPU that has idled streams 10-15 % slower for its first ~20 ms -- 19.4 us per query falling to 16.7 over thirty launches,
        //  tools/launch_series.py --: the settings are compared on a GPU that has been busy for 64 launches)
        m.pace_quads = cand[0][0];
        m.pace_levels = cand[0][1];
        for (int w = 0; w < 64; ++w) m.launch_batch(xs.data(), oi.data(), ov.data(), nq, m.stream);
        HIP_TRY(hipStreamSynchronize(m.stream));
        HIP_TRY(m.settle());
        for (int pass = 0; pass < 2; ++pass)
            for (int c = 0; c < NC; ++c) {
                m.pace_quads = cand[c][0];
                m.pace_levels = cand[c][1];
                m.launch_batch(xs.data(), oi.data(), ov.data(), nq, m.stream);  // (settles the pauses the setting leads to)
                HIP_TRY(hipEventRecord(m.ev0, m.stream));
                m.launch_batch(xs.data(), oi.data(), ov.data(), nq, m.stream);
                m.launch_batch(xs.data(), oi.data(), ov.data(), nq, m.stream);
                HIP_TRY(hipEventRecord(m.ev1, m.stream));
                HIP_TRY(hipEventSynchronize(m.ev1));
                HIP_TRY(m.settle());
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
                best_ms[c] = std::min(best_ms[c], ms);
            }
        int best = 0;
        for (int c = 1; c < NC; ++c)
            if (best_ms[c] < best_ms[best]) best = c;
        m.pace_quads = cand[best][0];
        m.pace_levels = cand[best][1];
        // ---- the timetable (pacing by the clock, BatchParams::pace_period) ------------------------------------------------------
        // Its period must be what THIS box sustains with THIS kernel: a per cent too short and the waves fall behind, nobody pauses
        // and the XCDs are back to sharing unevenly (+5 %); too long costs what it is too long by. The best period sits 5-6 % under
        // the time per query of the pauses by rank just measured (15.8 us on the pool's fast boxes, 16.0 on a medium one): a coarse
        // grid around that, then half a per cent either side of its best; kept only if it beats the pauses by rank.
        float period_ms = 1e30f;
        uint32_t period_ns = 0;
        if (!opt("PACE_PERIOD")) {
            // (ms per TWO launches, the mean over `launches` of them in one go, the better of `passes` such series)
            // (a HIP call that fails leaves `measure_failed` set: the timetable is then not taken, the pauses by rank stay)
            bool measure_failed = false;
            auto measure = [&](uint32_t ns, int passes, int launches) -> float {
                m.pace_period_ns = ns;
                float b = 1e30f;
                for (int pass = 0; pass < passes; ++pass) {
                    m.launch_batch(xs.data(), oi.data(), ov.data(), nq, m.stream);
                    bool ok = hipEventRecord(m.ev0, m.stream) == hipSuccess;
                    for (int l = 0; l < launches; ++l) m.launch_batch(xs.data(), oi.data(), ov.data(), nq, m.stream);
                    ok = ok && hipEventRecord(m.ev1, m.stream) == hipSuccess && hipEventSynchronize(m.ev1) == hipSuccess && m.settle() == hipSuccess;
                    float ms = 0;
                    ok = ok && hipEventElapsedTime(&ms, m.ev0, m.ev1) == hipSuccess && hipGetLastError() == hipSuccess;
                    if (!ok) {
                        measure_failed = true;
                        return 1e30f;
                    }
                    b = std::min(b, ms * 2.0f / (float)launches);
                }
                return b;
            };
            const double rank_ns = (double)best_ms[best] * 1e6 / (2.0 * nq);  // per query
            static const double grid[] = {0.90, 0.915, 0.93, 0.945, 0.96, 0.975};
            for (double g : grid) {
                const uint32_t ns = (uint32_t)(rank_ns * g);
                const float ms = measure(ns, 2, 2);
                if (ms < period_ms) {
                    period_ms = ms;
                    period_ns = ns;
                }
            }
            // The grid's best is the best of single launches. Back to back, a period that close to what the kernel sustains has
            // every fourth launch or so run 8-10 % longer (for ~100 us the memory system gives less, one XCD's workgroups take the whole
            // shortfall, fall 40-80 us behind and the launch waits for them: tools/spike_probe.py); 2 % more period and those launches
            // are gone (same box: 16.7 us per query with spikes to 18.5 at 15.8 us, 16.9 without any at 16.1). So the fine search
            // measures series of eight launches -- their mean --, from the grid's best upwards.
            const uint32_t centre = period_ns;
            period_ms = 1e30f;
            for (double g : {1.03, 1.02, 1.01, 1.0, 0.99}) {
                const uint32_t ns = (uint32_t)((double)centre * g);
                const float ms = measure(ns, 2, 8);
                // (from the longest period down: a shorter one must measure 0.3 % better to be taken -- a single launch at a period
                //  close to what the kernel sustains runs 4-5 % long one time in eight, which a mean over eight launches half hides)
                if (ms < period_ms * 0.997f) {
                    period_ms = ms;
                    period_ns = ns;
                }
            }
            m.pace_period_ns = (!measure_failed && period_ms < best_ms[best] * 0.999f) ? period_ns : 0u;
        }
        const uint32_t tune_us = (uint32_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_tune).count();
        // the engine starts as if nothing had run: no carried thresholds, no pauses, no counters, no trust
        HIP_TRY(hipMemset(m.d_wg_prior, 0, ((size_t)m.grid + 32) * 4));
        if (m.d_wg_pace) HIP_TRY(hipMemset(m.d_wg_pace, 0, (size_t)m.grid * 4));
        if (m.d_wg_sig) HIP_TRY(hipMemset(m.d_wg_sig, 0, (size_t)m.grid * 8 * 4));
        m.clean_seen = m.distrust_left = 0u;
        m.trusted_launches = m.late_repairs = 0;
        const uint64_t tune_launches = m.batch_launches;
        m.batch_launches = 0;
        HIP_TRY(hipMemset(m.d_verdict, 0, 512));
        HIP_TRY(hipMemset(m.d_out_idx, 0, (size_t)d.k * 4));
        HIP_TRY(hipMemset(m.d_out_val, 0, (size_t)d.k * 4));
        (void)hipFree(d_tx);
        m.pace_tuned_us = tune_us ? tune_us : 1u;
        m.pace_tune_launches = (uint32_t)tune_launches;
        m.pace_adapt_on = adapt_was;
        if (opt("DEBUG_OCC")) {
            fprintf(stderr, "[tkspmv] pacing tuned in %u us:", tune_us);
            for (int c = 0; c < NC; ++c) fprintf(stderr, " %ux%u %.2f us/q%s", cand[c][0], cand[c][1], best_ms[c] * 1e3 / (2 * nq), c == best ? "*" : "");
            fprintf(stderr, "; timetable %u ns: %.2f us/q%s\n", period_ns, period_ms * 1e3 / (2 * nq), m.pace_period_ns ? " (kept)" : " (pauses by rank kept)");
        }
    }
    m.info.batch_mode = (m.can_batch ? (m.n_sel_wg | (m.use_local << 8)) : 0u) | (std::min<uint32_t>(n_parts_hint, 0xFFFFu) << 16);
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
    // (with the matrix described -- rows, cols, nnz, precision --: the hint tkspmv_create itself would pack this matrix with;
    //  without: the largest count any engine of this geometry accepts)
    *out = (grid - (defer_capable ? small_matrix_settings(d, grid, defer_capable, entries_per_lane_of(d), nullptr) : 0u)) * waves_per_wg;
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
    if (m.bar_x) {
        // (a launch enqueued earlier may still be reading d_x: tkspmv_run clears x_pending, the asynchronous entry points do not)
        if (m.x_pending) HIP_TRY(hipStreamSynchronize(m.stream));
        std::memcpy(m.d_x, host_x, (size_t)m.desc.cols * 4);
        m.flush_hdp();  // (sfence, HDP flush register written and read back: the stores are in memory before the launch's doorbell)
        m.x_pending = true;
        m.d_x_cur = m.d_x;
        m.have_query = true;
        if (elapsed_ns)
            *elapsed_ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        return TKSPMV_OK;
    }
    if (m.host_path && m.h_x) {
        // (an earlier upload from the staging copy must have been consumed before it is overwritten)
        if (m.x_pending) HIP_TRY(hipStreamSynchronize(m.stream));
        std::memcpy(m.h_x, host_x, (size_t)m.desc.cols * 4);
        if (!m.host_x_direct) {
            HIP_TRY(hipMemcpyAsync(m.d_x, m.h_x, (size_t)m.desc.cols * 4, hipMemcpyHostToDevice, m.stream));
            HIP_TRY(hipEventRecord(m.ev2, m.stream));  // a launch on another stream waits for this upload (order_x)
        }
        m.x_pending = true;
        m.d_x_cur = m.host_x_direct ? m.h_x_dev : m.d_x;
        m.have_query = true;
        if (elapsed_ns)
            *elapsed_ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        return TKSPMV_OK;
    } else {
        HIP_TRY(hipMemcpyAsync(m.d_x, host_x, (size_t)m.desc.cols * 4, hipMemcpyHostToDevice, m.stream));
        HIP_TRY(hipStreamSynchronize(m.stream));
    }
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
    HIP_TRY(m.order_x(x, s));
    m.launch_query(x, dev_idx ? dev_idx : m.d_out_idx, dev_val ? dev_val : m.d_out_val, s);
    HIP_TRY(hipGetLastError());
    m.ran = true;
    m.last_on_host = false;
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
    m.last_on_host = false;
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
    m.last_on_host = false;
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
    m.last_on_host = false;
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
    m.last_on_host = false;
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
    m.last_on_host = false;
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
    HIP_TRY(m.settle());
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
        // (kept for tkspmv_debug_counters: the guard of the passes' threshold exchange is a COUNT -- waves that ran into their
        //  bounded wait, rows that overflowed a list -- not a wall clock)
        m.multi_stat_queries += (uint64_t)iters;
        m.multi_waits += st[12];
        m.multi_wait_ticks += st[13];
        m.multi_rows_offered += st[5] + st[9];
        m.multi_rows_overflowed += st[7] + st[11];
        HIP_TRY(hipMemset(m.d_stats, 0, 32 * 8));
    }
    m.ran = true;
    m.last_on_host = false;
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
    m.last_on_host = false;
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
    // The fused single launch can hand its result to the host itself (see h_res); the other launch schemes (radix select,
    // row per lane, unfused selection) complete in stream order and are waited for with the event.
    const bool to_host = m.host_path && m.h_res && m.fused && !m.use_radix && !(m.desc.impl == TKSPMV_IMPL_ROW_PER_LANE && m.can_multi);
    const bool events = kernel_ns && (m.run_events || !to_host);
    const auto t_host0 = std::chrono::steady_clock::now();
    if (events) HIP_TRY(hipEventRecord(m.ev0, m.stream));
    // The result flag of the launch with the current epoch, polled (never longer than 2 s), and the block behind it verified.
    auto wait_flag = [&]() -> bool {
        volatile uint32_t *flag = m.h_res + 2 * (size_t)m.desc.k;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spins = 0;; ++spins) {
            if (*flag == m.host_epoch && m.result_block_complete(m.host_epoch)) return true;
            __builtin_ia32_pause();
            if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return false;
        }
    };
    bool seen = false;
    double own_ticks = 0.0;  // the launches' own durations (100 MHz ticks: first workgroup's entry -> result flag)
    if (to_host) {
        m.drain(m.stream);
        if (m.can_single) {
            // workgroup-local thresholds, checked by the selection (kernels/local.hpp); a failed check -- a query unlike the ones
            // before it -- sends the query through the exact launch below
            m.launch_single(m.d_x_cur, m.d_out_idx, m.d_out_val, m.stream, true);
            HIP_TRY(hipGetLastError());
            seen = wait_flag();
            if (seen) own_ticks += (double)m.h_res[2 * (size_t)m.desc.k + 1];
        }
        if (!m.can_single || !seen || m.h_res[2 * (size_t)m.desc.k + 5] != 0u) {
            if (m.can_single) ++m.single_repairs;
            m.h_res[2 * (size_t)m.desc.k + 5] = 0u;  // (only single_kernel writes the status word: the exact launch's block has none)
            m.launch_stream(m.d_x_cur, m.d_out_idx, m.d_out_val, m.stream, true);
            HIP_TRY(hipGetLastError());
            seen = wait_flag();
            if (seen) own_ticks += (double)m.h_res[2 * (size_t)m.desc.k + 1];
        }
    } else {
        m.launch_query(m.d_x_cur, m.d_out_idx, m.d_out_val, m.stream);
    }
    HIP_TRY(hipGetLastError());
    if (events) HIP_TRY(hipEventRecord(m.ev1, m.stream));
    if (!seen) HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(m.settle());
    m.x_pending = false;  // the kernel has read x: the staging copy is free again
    if (kernel_ns && !events) {
        // the kernels' own durations, or the host clock if the flag never came
        if (seen) *kernel_ns = own_ticks * 10.0;
        else *kernel_ns = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_host0).count();
    }
    if (events) {
        if (seen) {  // the kernel is in its last instructions: the end event follows within a microsecond or two
            hipError_t q;
            while ((q = hipEventQuery(m.ev1)) == hipErrorNotReady) __builtin_ia32_pause();
            if (q != hipSuccess) {
                err = std::string("hipEventQuery failed: ") + hipGetErrorString(q);
                return TKSPMV_ERR_DEVICE;
            }
        }
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
        *kernel_ns = (double)ms * 1e6;
    }
    m.ran = true;
    m.last_on_host = seen;
    return TKSPMV_OK;
}

int Engine::synchronize(std::string &err) {
    HIP_TRY(hipSetDevice(impl_->device));
    impl_->drain(impl_->stream);
    HIP_TRY(hipStreamSynchronize(impl_->stream));
    HIP_TRY(impl_->settle());
    return TKSPMV_OK;
}

int Engine::read(uint32_t *idx, float *val, int32_t *n, std::string &err) {
    EngineImpl &m = *impl_;
    if (!m.ran) {
        err = "no query has been run";
        return TKSPMV_ERR_STATE;
    }
    if (m.last_on_host) {  // tkspmv_run left the result in pinned host memory
        if (idx) std::memcpy(idx, m.h_res, (size_t)m.desc.k * 4);
        if (val) std::memcpy(val, m.h_res + m.desc.k, (size_t)m.desc.k * 4);
        if (n) *n = m.desc.k;
        return TKSPMV_OK;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(m.settle());
    m.x_pending = false;
    if (idx) HIP_TRY(hipMemcpy(idx, m.d_out_idx, (size_t)m.desc.k * 4, hipMemcpyDeviceToHost));
    if (val) HIP_TRY(hipMemcpy(val, m.d_out_val, (size_t)m.desc.k * 4, hipMemcpyDeviceToHost));
    if (n) *n = m.desc.k;
    return TKSPMV_OK;
}

// out[0] = selections whose threshold check failed so far (each sent its query through the repair launch), out[1] = the current
// suspension length of carried thresholds, out[2] = selections to go until they are used again, out[3] = batch launches so far,
// out[4] = launches the local thresholds as a whole stay switched off for, out[5] = the length of that closure.
int Engine::debug_counters(unsigned long long *out, int n, std::string &err) {
    EngineImpl &m = *impl_;
    if (!out || n < 6) {
        err = "debug_counters needs room for 6 values";
        return TKSPMV_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m.device));
    if (m.stream) HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(m.settle());
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (m.d_wg_prior) HIP_TRY(hipMemcpy(w, m.d_wg_prior + m.grid, sizeof(w), hipMemcpyDeviceToHost));
    out[0] = w[3];
    out[1] = w[1];
    out[2] = w[0];
    out[3] = m.batch_launches;
    out[4] = w[(m.batch_launches & 1u) ? 7 : 4];  // launches the gate of the local thresholds stays closed for (the next launch's copy)
    out[5] = w[5];  // length of its latest closure
    if (n >= 10) {  // tkspmv_run through single_kernel: launches, queries sent on through the exact launch, failed checks, suspension
        uint32_t s[4] = {0, 0, 0, 0};
        if (m.d_lprior) HIP_TRY(hipMemcpy(s, m.d_lprior + m.grid, sizeof(s), hipMemcpyDeviceToHost));
        out[6] = m.single_launches;
        out[7] = m.single_repairs;
        out[8] = s[3];
        out[9] = s[0];
    }
    if (n >= 12) {  // round 5: batch launches that went out without a repair launch behind them, and the late repairs they cost
        out[10] = m.trusted_launches;
        out[11] = m.late_repairs;
    }
    if (n >= 14) {  // the pacing in force: quantum | levels << 8 | base << 16, and what tkspmv_create's measurement of it took (us; 0: not measured)
        out[12] = m.pace_quads | (m.pace_levels << 8) | (m.pace_base << 16) | ((uint64_t)m.pace_period_ns << 32);  // (upper half: the timetable's period, ns per query; 0: pauses by rank)
        out[13] = m.pace_tuned_us | ((uint64_t)m.pace_tune_launches << 32);  // (upper half: the batch launches the measurement made)
    }
    if (n >= 19) {  // option STATS, summed over the tkspmv_time_multi calls so far: queries, bounded waits for a threshold (count, ticks of 10 ns), rows offered / overflowed
        out[14] = m.multi_stat_queries;
        out[15] = m.multi_waits;
        out[16] = m.multi_wait_ticks;
        out[17] = m.multi_rows_offered;
        out[18] = m.multi_rows_overflowed;
    }
    return TKSPMV_OK;
}

int Engine::read_trace(unsigned long long *host, size_t max_words, size_t *words, std::string &err) {
    EngineImpl &m = *impl_;
    if (!m.d_trace && m.d_wg_times) {  // option WG_TIMES: [BATCH_MAX + 1][grid] hand-over stamps of the last batch launch
        HIP_TRY(hipSetDevice(m.device));
        HIP_TRY(hipDeviceSynchronize());
        const size_t n = std::min(max_words, (size_t)(BATCH_MAX + 1) * m.grid);
        HIP_TRY(hipMemcpy(host, m.d_wg_times, n * 8, hipMemcpyDeviceToHost));
        if (words) *words = n;
        return TKSPMV_OK;
    }
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
    HIP_TRY(m.settle());
    return TKSPMV_OK;
}

int Engine::time_queries(const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || n_x < 1 || iters < 1 || !ns_per_query) {
        err = "bad arguments to time_queries";
        return TKSPMV_ERR_INVALID;
    }
    const bool host_times = opt("HOST_TIMES") != nullptr;  // (diagnostic: where the host's microseconds around the region go, to stderr)
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double h0 = host_times ? now_us() : 0.0;
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    const double h1 = host_times ? now_us() : 0.0;
    // (batch launches on one stream: the bracket's events travel with the first and the last kernel; any other path records them)
    const int ext_mode = opt("EXT_EVENTS") ? atoi(opt("EXT_EVENTS")) : 1;
    const bool ext = m.can_batch && !(m.overlap_launches && m.bside) && ext_mode != 0;
    m.ext_record = ext && ext_mode == 2;
    if (ext) {
        m.ext_start = m.ev0;
        m.ext_stop = m.ev1;
    } else {
        HIP_TRY(hipEventRecord(m.ev0, m.stream));
    }
    {
        std::vector<const float *> xs;
        std::vector<uint32_t *> oi;
        std::vector<float *> ov;
        sequence_lists(m, dev_xs, n_x, iters, m.d_out_idx, m.d_out_val, 0, xs, oi, ov);
        m.launch_sequence(xs.data(), oi.data(), ov.data(), iters, m.stream);
    }
    if (!ext) HIP_TRY(hipEventRecord(m.ev1, m.stream));
    const double h2 = host_times ? now_us() : 0.0;
    if (m.ext_start || m.ext_stop) {  // (cannot happen: every batch path launches through launch_batch)
        m.ext_start = m.ext_stop = nullptr;
        err = "time_queries: the region's events were not attached to its launches";
        return TKSPMV_ERR_STATE;
    }
    // (polled: a blocking wait adds 10-20 us of wake-up latency to a region that may be as short as 400 us)
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(m.ev1);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(q);
        if (spins > (1u << 22)) {  // seconds of polling: something is wrong, let the runtime wait
            HIP_TRY(hipEventSynchronize(m.ev1));
            break;
        }
    }
    const double h3 = host_times ? now_us() : 0.0;
    HIP_TRY(m.settle());  // (the end event has passed: the stream is idle; a flagged query is repaired outside the event bracket)
    const double h4 = host_times ? now_us() : 0.0;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
    if (host_times)
        fprintf(stderr, "[tkspmv] time_queries(%d): entry + wait for the idle stream %.1f us, lists + launches %.1f, until the end event is seen %.1f (the region itself %.1f), verdicts %.1f, elapsed time %.1f\n",
                iters, h1 - h0, h2 - h1, h3 - h2, (double)ms * 1e3, h4 - h3, now_us() - h4);
    *ns_per_query = (double)ms * 1e6 / iters;
    m.ran = true;
    m.last_on_host = false;
    return TKSPMV_OK;
}

// `reps` repetitions of `iters` back-to-back queries, ALL enqueued before the first wait, one hipEvent between consecutive
// repetitions: ns_per_query[r] = the r-th repetition's time / iters. The GPU never idles between repetitions, so the figures
// are the kernel's under sustained load -- what tkspmv_time_queries measures when it is called in a loop is the kernel right
// after a host-side gap, 10-25 % slower for about a millisecond (power management: DESIGN.md).
int Engine::time_query_batches(const float *dev_xs, int32_t n_x, int32_t iters, int32_t reps, double *ns_per_query, std::string &err) {
    EngineImpl &m = *impl_;
    if (!dev_xs || n_x < 1 || iters < 1 || reps < 1 || reps > 4096 || !ns_per_query) {
        err = "bad arguments to time_query_batches";
        return TKSPMV_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(m.device));
    HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(m.settle());
    std::vector<hipEvent_t> evs((size_t)reps + 1);
    for (auto &e : evs) HIP_TRY(hipEventCreate(&e));
    std::vector<const float *> xs;
    std::vector<uint32_t *> oi;
    std::vector<float *> ov;
    sequence_lists(m, dev_xs, n_x, iters, m.d_out_idx, m.d_out_val, 0, xs, oi, ov);
    HIP_TRY(hipEventRecord(evs[0], m.stream));
    for (int r = 0; r < reps; ++r) {
        m.launch_sequence(xs.data(), oi.data(), ov.data(), iters, m.stream);
        HIP_TRY(hipEventRecord(evs[(size_t)r + 1], m.stream));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(evs[(size_t)reps]));
    HIP_TRY(m.settle());
    for (int r = 0; r < reps; ++r) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, evs[(size_t)r], evs[(size_t)r + 1]));
        ns_per_query[r] = (double)ms * 1e6 / iters;
    }
    for (auto &e : evs) (void)hipEventDestroy(e);
    m.ran = true;
    m.last_on_host = false;
    return TKSPMV_OK;
}

int Engine::time_stream_read(int32_t passes, double *ns_per_pass, std::string &err) {
    EngineImpl &m = *impl_;
    if (passes < 1 || !ns_per_pass) {
        err = "bad arguments to time_stream_read";
        return TKSPMV_ERR_INVALID;
    }
    if (m.pm.n_packets == 0) {
        *ns_per_pass = 0.0;
        return TKSPMV_OK;
    }
    HIP_TRY(hipSetDevice(m.device));
    const uint32_t stream_block = m.block;  // the streaming waves of a workgroup (the server wave of a sequence launch aside)
    ReadProbeParams R{};
    R.n_replicas = m.d_replicas.empty() ? 1u : (uint32_t)std::min<size_t>(m.d_replicas.size(), 8);
    for (uint32_t r = 0; r < 8u; ++r) R.replicas[r] = m.d_replicas.empty() ? m.d_packets : m.d_replicas[r % m.d_replicas.size()];
    R.part_first = m.d_part_first;
    R.part_count = m.d_part_count;
    R.n_parts = (uint32_t)m.pm.part_first.size();
    R.n_pass = (uint32_t)passes;
    if (const char *f = opt("READ_PROBE_MAP")) R.map = (uint32_t)atoi(f);
    if (const char *f = opt("READ_PROBE_PERIOD")) R.period = (uint32_t)std::min<uint64_t>(0xFFFFFF00ull, (uint64_t)std::max(0, atoi(f)) * 256u / 10u);  // ns per pass: the probe on a timetable
    struct SinkGuard {  // freed on every return path
        uint32_t *p = nullptr;
        ~SinkGuard() {
            if (p) (void)hipFree(p);
        }
    } sink;
    const size_t n_end = (size_t)m.grid * 16;
    HIP_TRY(hipMalloc((void **)&sink.p, (size_t)m.grid * 16 * 4 + (size_t)(passes + 2) * 128 + n_end * 8));
    R.sink = sink.p;
    R.claim = sink.p + (size_t)m.grid * 16;
    const bool want_ends = opt("READ_PROBE_ENDS") != nullptr;  // (tuning runs: when did the waves of each XCD finish?)
    R.t_end = want_ends ? reinterpret_cast<unsigned long long *>(R.claim + (size_t)(passes + 2) * 32) : nullptr;
    void (*fn)(ReadProbeParams) = nullptr;
    switch (m.pm.packet_bytes / 64u) {
        case 22: fn = read_probe_kernel<22>; break;
        case 24: fn = read_probe_kernel<24>; break;
        case 48: fn = read_probe_kernel<48>; break;
        case 16: fn = read_probe_kernel<16>; break;
        case 20: fn = read_probe_kernel<20>; break;
        case 12: fn = read_probe_kernel<12>; break;
        default: break;
    }
    if (const char *f = opt("READ_PROBE")) {  // "depth,work" (tuning runs; fp32 packets only)
        int depth = 8, work = 0;
        sscanf(f, "%d,%d", &depth, &work);
        if (m.pm.packet_bytes == 1536u) {
#define RP(D, W) if (depth == D && work == W) fn = read_probe_kernel<24, D, W>;
            if (depth == 22) fn = read_probe_kernel<22, 8, 1>;  // what 1408-byte packets would cost, read out of this 1536-byte stream
            RP(3, -1) RP(3, -2) RP(4, -2) RP(2, 0) RP(3, 0) RP(4, 0) RP(6, 0) RP(12, 0) RP(16, 0)
            RP(3, 1) RP(3, 32) RP(3, 64) RP(3, 96) RP(3, 128) RP(4, 64) RP(4, 96) RP(6, 64) RP(6, 96) RP(8, 64) RP(8, 96) RP(8, 1)
#undef RP
        }
    }
    if (!fn) {
        err = "no read probe for this packet size";
        return TKSPMV_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipStreamSynchronize(m.stream));
    HIP_TRY(m.settle());
    R.n_pass = 2;  // (warm-up: code object, clocks)
    HIP_TRY(hipMemsetAsync(R.claim, 0, (size_t)(passes + 2) * 128, m.stream));
    hipLaunchKernelGGL(fn, dim3(m.grid), dim3(stream_block), 0, m.stream, R);
    R.n_pass = (uint32_t)passes;
    HIP_TRY(hipMemsetAsync(R.claim, 0, (size_t)(passes + 2) * 128, m.stream));
    HIP_TRY(hipEventRecord(m.ev0, m.stream));
    hipLaunchKernelGGL(fn, dim3(m.grid), dim3(stream_block), 0, m.stream, R);
    HIP_TRY(hipEventRecord(m.ev1, m.stream));
    HIP_TRY(hipEventSynchronize(m.ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, m.ev0, m.ev1));
    HIP_TRY(hipGetLastError());
    if (want_ends) {
        std::vector<unsigned long long> te(n_end);
        HIP_TRY(hipMemcpy(te.data(), R.t_end, n_end * 8, hipMemcpyDeviceToHost));
        const uint32_t nw = stream_block / 64u;
        unsigned long long t0 = ~0ull, t1 = 0ull;
        for (uint32_t b = 0; b < m.grid; ++b)
            for (uint32_t w = 0; w < nw; ++w) {
                t0 = std::min(t0, te[(size_t)b * nw + w]);
                t1 = std::max(t1, te[(size_t)b * nw + w]);
            }
        fprintf(stderr, "[read probe] wave end times relative to the last one, by blockIdx %% 8 (mean / earliest, us):");
        for (uint32_t x = 0; x < 8u; ++x) {
            double sum = 0, mn = 1e30;
            uint32_t n = 0;
            for (uint32_t b = x; b < m.grid; b += 8u)
                for (uint32_t w = 0; w < nw; ++w, ++n) {
                    const double d = (double)(t1 - te[(size_t)b * nw + w]) * 0.01;
                    sum += d;
                    mn = std::min(mn, -d);
                }
            fprintf(stderr, " %.1f / %.1f", -sum / std::max(1u, n), mn);
        }
        fprintf(stderr, "  (launch %.1f us)\n", (double)ms * 1e3);
        std::vector<double> lead;
        for (size_t i = 0; i < n_end; ++i)
            if (te[i] != 0ull && (double)(t1 - te[i]) * 0.01 < (double)ms * 1e3 * 0.9) lead.push_back((double)(t1 - te[i]) * 0.01);
        std::sort(lead.begin(), lead.end());
        if (!lead.empty()) {
            fprintf(stderr, "[read probe] waves still running x us before the end of the launch:");
            for (double x : {1.0, 5.0, 10.0, 20.0, 40.0, 80.0, 160.0, 320.0})
                fprintf(stderr, " %g us: %zu", x, (size_t)(std::lower_bound(lead.begin(), lead.end(), x) - lead.begin()));
            fprintf(stderr, " (of %zu)\n", lead.size());
        }
    }
    *ns_per_pass = (double)ms * 1e6 / passes;
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
    HIP_TRY(m.settle());
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
            else if (m.can_single) m.launch_single(x, m.d_out_idx, m.d_out_val, m.stream, false);  // (what tkspmv_run launches)
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
        HIP_TRY(m.settle());
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
        unsigned long long sx[14];
        HIP_TRY(hipMemcpy(sx, m.d_stats, sizeof(sx), hipMemcpyDeviceToHost));
        const double nsel = (double)std::max<unsigned long long>(1, sx[1]);
        fprintf(stderr, "[tkspmv stats] selections %llu, general-path selections %llu, max candidates %llu, overflow entries per selection %.1f; "
                        "per selection: waves that waited for a threshold at the end of their partition %.1f (%.1f us in all)\n",
                sx[1], sx[3], sx[2], (double)sx[8] / nsel, (double)sx[10] / nsel, (double)sx[9] / 100.0 / nsel);
    }
    if (m.collect_stamps) {
        unsigned long long st[16];
        HIP_TRY(hipMemcpy(st, m.d_stats + 16, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "[tkspmv stamps, shader cycles since stream end of the last workgroup] flush_issued %lld flush_done %lld ticket %lld loads %lld keys %lld ranked %lld end %lld\n",
                (long long)(st[1] - st[0]), (long long)(st[2] - st[0]), (long long)(st[3] - st[0]), (long long)(st[4] - st[0]),
                (long long)(st[5] - st[0]), (long long)(st[6] - st[0]), (long long)(st[7] - st[0]));
    }
    m.ran = true;
    m.last_on_host = false;
    return TKSPMV_OK;
}

// The reference's loop -- reset(x) from host memory, operator(), read_result() (host_spmv_bscsr.cpp:602-632) -- `iters` times in
// native code, cycling over n_x host vectors: loop_ns[i] = the host's steady clock around the three calls of iteration i,
// kernel_ns[i] = what tkspmv_run returned for it. What a C++ host of the reference sees per iteration; a Python caller adds a
// ctypes transition per call to it.
int Engine::time_host_loop(const float *host_xs, int32_t n_x, int32_t iters, double *loop_ns, double *kernel_ns, std::string &err) {
    if (!host_xs || n_x < 1 || iters < 1 || !loop_ns) {
        err = "time_host_loop: NULL or empty arguments";
        return TKSPMV_ERR_INVALID;
    }
    EngineImpl &m = *impl_;
    std::vector<uint32_t> idx((size_t)m.desc.k);
    std::vector<float> val((size_t)m.desc.k);
    for (int32_t i = 0; i < iters; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        double set_ns = 0.0, k_ns = 0.0;
        int32_t n = 0;
        int st = set_query(host_xs + (size_t)(i % n_x) * m.desc.cols, &set_ns, err);
        if (st == TKSPMV_OK) st = run(&k_ns, err);
        if (st == TKSPMV_OK) st = read(idx.data(), val.data(), &n, err);
        if (st != TKSPMV_OK) return st;
        loop_ns[i] = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
        if (kernel_ns) kernel_ns[i] = k_ns;
    }
    return TKSPMV_OK;
}

}  // namespace tkspmv
