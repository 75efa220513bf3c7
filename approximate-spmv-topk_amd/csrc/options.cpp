#include "options.hpp"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>

namespace tkspmv {

static const OptionDef k_options[] = {
    // ---- behaviour: which path answers a query ----
    {"LOCAL", "behaviour", "0 | 1 | 2 (default: by matrix size)",
     "threshold scheme of the batch and single-query kernels: 0 = device-wide exchange (exact by construction); 1 / 2 = workgroup-local "
     "thresholds from each wave's best / second-best packet maximum, carried from query to query and CHECKED by the selection (a failed "
     "check repairs the query through the exact kernel). Ignored above 512 workgroups."},
    {"SIGNATURES", "behaviour", "0 | 1 (default 1)", "1: a carried workgroup threshold is used only for a query whose signature (sum x / sum |x|, share of the columns that carry it) matches the query it came from, two priors remembered per workgroup; 0: round 4's behaviour (every query starts from the last prior; a failed check suspends carrying)"},
    {"LOCAL_BETA", "tuning", "float (default 1.0)", "factor applied to a carried workgroup threshold before the next query uses it"},
    {"SINGLE_SELECTOR", "behaviour", "0 | 1 (default 1)", "1: workgroup 0 of a single-query launch only selects -- it polls the other workgroups' flags and loads each record as it is delivered; 0: round 4's scheme (every workgroup streams, the one that draws the last ticket selects)"},
    {"SINGLE", "behaviour", "0 | 1 (default 1)", "0: tkspmv_run uses the stream kernel with the device-wide exchange instead of the single-query kernel with local thresholds"},
    {"BATCH", "behaviour", "0 | 1 (default 1)", "0: tkspmv_enqueue_batch / _many launch one kernel per query"},
    {"BATCH_MAX", "tuning", "1..32 (default 32)", "queries per batch launch"},
    {"SELECTORS", "tuning", "1..8 (default 4 up to LOCAL_MATRIX_PACKETS, else 1)", "selection workgroups a batch launch keeps in flight"},
    {"OVF_LISTS", "tuning", "1 | 2 | 4 (default 4; 2 for engines that stream with local thresholds; at least 2 wherever the deferred scheme can run)", "overflow lists of the exact kernel (8 bytes per row each), shared round robin by the queries of a launch under flow control"},
    {"PACE", "tuning", "0..32 (default by size)", "pacing quantum of the batch kernel's workgroups by rank (s_sleep units); 0 = none"},
    {"PACE_LEVELS", "tuning", "1..8", "number of distinct pacing ranks"},
    {"AUTOTUNE", "behaviour", "0 | 1 (default 1)", "1: tkspmv_create measures a handful of pacing settings on the matrix it has just packed (engines of checked local thresholds that pace at all; ~10 ms) and keeps the fastest; PACE / PACE_LEVELS / PACE_BASE switch it off"},
    {"PACE_CARRY", "tuning", "0 | 1 (default 1)", "1: a workgroup starts a launch with the pause it ended the previous launch with; 0: every launch starts unpaced"},
    {"HOST_TIMES", "diagnostic", "set = on", "tkspmv_time_queries prints to stderr where the host's microseconds around the timed region go"},
    {"EXT_EVENTS", "diagnostic", "0 | 1 | 2 (default 1; 2 = the pair recorded right in front of the first and right behind the last launch: 7 us more per region than 1, and the device-wide wait behind the call 11 us cheaper)", "tkspmv_time_queries: 1 = the event pair travels with the region's first and last kernel (hipExtLaunchKernelGGL: the dispatches' own start and end stamps, what rocprofv3 reports); 0 = hipEventRecord before and after (the start stamp then precedes the host's writing of the first dispatch packet: +1.6 us)"},
    {"PACE_PERIOD", "tuning", "ns per query (default: measured at create; 0 = pacing by rank)", "pacing by the clock: every streaming wave keeps a timetable of this many ns per query and sleeps off what it is ahead of it (replaces the pauses by rank)"},
    {"BALANCED_CUTS", "layout", "0 | 1 | 2 (default 1; 2 = already where the uniform cut misses the count by 1/32: tuning runs)", "1: where partitions of equal capacity come to fewer than the waves asked for by more than 1/8, the packets are dealt out over exactly that many partitions (floor or ceil of the mean each): every workgroup of the batch kernel then streams the same number of partitions; 0: equal capacities always (rounds 1-4)"},
    {"PACE_ADAPT", "behaviour", "0 | 1 (default 1)", "1: the timetable's period lengthens by 1/64 per launch while a quarter of the waves start a launch's last query more than a quarter of a period late (the GPU streams slower than when tkspmv_create measured), and comes back by 1/256 per launch when next to none do; 0: the period stays as measured"},
    {"PACE_BASE", "tuning", "0..64 (default 0)", "pause per packet (s_sleep units) of EVERY workgroup of the batch kernel, whatever its rank: a uniform throttle (tuning runs)"},
    {"REPAIR", "behaviour", "host | stream (default host)",
     "who looks at the verdict of a launch of checked local thresholds: host = the launch goes out alone once the verdicts the host has seen are clean, "
     "and tkspmv_synchronize / tkspmv_read (any engine call that waits for the engine's stream) repair a flagged query with an exact launch; "
     "stream = the exact launch follows every local launch in the stream (always so on a caller's stream and for 64 launches after an observed failure)"},
    {"OVERLAP", "behaviour", "0 | 1 (default 1)", "1: consecutive launches of one sequence of checked local thresholds alternate between two streams while they go out trusted (REPAIR=host): the next launch's ramp fills the previous one's tail; 0: one launch at a time"},
    {"FUSED", "behaviour", "0 | 1 (default 1 where the selection fits one workgroup)", "0: stream and selection as two launches"},
    {"RADIX", "behaviour", "0 | 1 (default: k above 3/8 of the publishing groups)", "1: scores + radix select instead of thresholded streaming"},
    {"MULTI_Q", "behaviour", "0 | 1 | 3 | 5 | 8 (default by size; desc.multi_q wins)", "queries per pass of the small-matrix kernel (multi_kernel); 0 = off"},
    {"MULTI_CHAINS", "tuning", "1 | 2 (default 2)", "independent launch chains tkspmv_enqueue_multi alternates between"},
    {"MULTI_PASSES", "tuning", "1..8 (default 8)", "passes a launch of the row-per-lane kernel makes at one or two queries per pass (8 / 4 by default)"},
    {"SMALL_PACKETS", "tuning", "packets (default LOCAL_MATRIX_PACKETS)", "size up to which a matrix gets the small-matrix settings (4 selectors, 1-2 packet partitions, local thresholds); 0 = round 2's behaviour"},
    {"MIN_PACKETS", "tuning", ">= 1", "minimum packets per wave partition"},
    {"READ_PROBE_PERIOD", "diagnostic", "ns per pass (0 = off)", "tkspmv_time_stream_read: the load-only probe on a timetable (every wave sleeps off what it is ahead of it) -- the floor bench.py reports as read_only.paced_us is the smallest time over a handful of periods"},
    {"PARTITIONS_HINT", "diagnostic", "count", "wave partitions to pack (read probe experiments; an engine whose partitions exceed its streaming waves does not batch)"},
    {"DEVICE_PACK", "behaviour", "0 | 1 (default 1)", "0: pack the matrix on the host instead of on the device"},
    {"HOST_PATH", "behaviour", "0 | 1 (default 1)", "0: no host-visible result block / mapped x (tkspmv_run and tkspmv_read go through hipMemcpy)"},
    {"BAR_X", "behaviour", "0 | 1 (default 1 where the device reports a large BAR and its HDP flush register, and a kernel reads back what the CPU stored)", "0: tkspmv_set_query stages x through pinned memory instead of storing into device memory"},
    {"HOST_X", "behaviour", "copy | direct | direct_nc (default copy)", "direct: kernels read x from mapped host memory (direct_nc: non-coherent mapping)"},
    {"RUN_EVENTS", "behaviour", "0 | 1 (default 0)", "1: tkspmv_run returns a hipEvent bracket instead of the kernel's own device-clock span"},
    {"HOST_THREADS", "tuning", "count", "threads of the host-side packer and generators"},
    // ---- layout ----
    {"F32_C12", "layout", "0 | 1 (default 1 for <= 4096 columns)", "fp32 values with 12-bit column words (1408-byte packets)"},
    {"SELL_C12", "layout", "0 | 1", "12-bit column words in the row-per-lane (SELL) layout"},
    {"FIXED_UNPACKED", "layout", "set = on", "fixed-point values as one u32 per entry instead of the packed 20..26-bit streams"},
    // ---- multi-GPU ----
    {"DIST_NO_NCCL", "behaviour", "set = on", "shard merge through the host instead of RCCL (CPU rehearsal, tests)"},
    {"DIST_FORCE_NCCL", "behaviour", "set = on", "RCCL even at world size 1"},
    {"DIST_BATCH", "tuning", "queries", "queries per exchange of the sharded engine"},
    // ---- diagnostics (DBG instantiations of the kernels; never on by default) ----
    {"STATS", "diagnostic", "set = on", "per-launch counters (offers, triggers) in tkspmv_debug_counters"},
    {"STAMPS", "diagnostic", "set = on", "device-clock stamps of the kernel phases"},
    {"WG_TIMES", "diagnostic", "set = on", "the batch kernel of local thresholds stamps every workgroup's hand-overs of its last launch ([33][grid] ticks of 10 ns, read through tkspmv_debug_trace): who leads, who lags (tools/wg_times.py)"},
    {"TRACE", "diagnostic", "set = on", "per-wave trace buffer for tkspmv_debug_trace"},
    {"DEBUG_OCC", "diagnostic", "set = on", "print launch geometry and occupancy at creation"},
    {"READ_PROBE", "diagnostic", "depth,work", "load-only probe: loads in flight and arithmetic per packet"},
    {"READ_PROBE_MAP", "diagnostic", "0..3", "load-only probe: workgroup -> partition map"},
    {"READ_PROBE_ENDS", "diagnostic", "set = on", "load-only probe: record when each XCD's waves finished"},
};

static std::mutex &lock() {
    static std::mutex m;
    return m;
}
// (values are interned in a node-based set that only grows: a pointer opt() / tkspmv_get_option has handed out stays valid
//  whatever another thread sets afterwards)
static std::set<std::string> &value_pool() {
    static std::set<std::string> p;
    return p;
}
static std::map<std::string, const std::string *> &overrides() {
    static std::map<std::string, const std::string *> m;
    return m;
}

int option_count() { return (int)(sizeof(k_options) / sizeof(k_options[0])); }
const OptionDef *option_def(int i) { return i >= 0 && i < option_count() ? &k_options[i] : nullptr; }

static const OptionDef *find(const char *name) {
    if (!name) return nullptr;
    for (const OptionDef &o : k_options)
        if (std::strcmp(o.name, name) == 0) return &o;
    return nullptr;
}

const char *opt(const char *name) {
    if (!find(name)) {
        std::fprintf(stderr, "[tkspmv] internal error: option %s is not in the table of options.cpp\n", name ? name : "(null)");
        std::abort();
    }
    {
        std::lock_guard<std::mutex> g(lock());
        auto it = overrides().find(name);
        if (it != overrides().end()) return it->second->c_str();  // (interned: stable for the life of the process)
    }
    return std::getenv((std::string("TKSPMV_") + name).c_str());
}

int set_option(const char *name, const char *value) {
    if (!find(name)) return -1;
    std::lock_guard<std::mutex> g(lock());
    if (value) overrides()[name] = &*value_pool().insert(value).first;
    else overrides().erase(name);
    return 0;
}

}  // namespace tkspmv
