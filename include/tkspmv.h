/*
 * tkspmv.h -- C ABI of the MI355X-native Top-K SpMV engine.
 *
 * This is the drop-in boundary for the ONE hot path of AlbertoParravicini/approximate-spmv-topk:
 * "given a fixed sparse matrix A (row-sorted COO, <= 16384 columns) and a fresh dense query x,
 *  return the K rows with the largest A.x".
 *
 * The reference has no FFI; the boundary it re-implements per back-end is the `struct SpMV`
 * engine concept (4 verbs) plus the loader/options helpers every `main` calls. Each entry point
 * below cites the reference interface it replaces (paths relative to the reference repo root):
 *
 *   tkspmv_create        <- SpMV::SpMV + packet_coo + setup   src/fpga/src/host_spmv_bscsr.cpp:104-131,133-321
 *                                                             src/gpu/host_spmv_topk_csr_gpu.cu:95-169
 *   tkspmv_set_query     <- SpMV::reset(vec)                  src/fpga/src/host_spmv_bscsr.cpp:450-485
 *                                                             src/gpu/host_spmv_topk_csr_gpu.cu:241-263
 *   tkspmv_run           <- SpMV::operator()(debug) + wait    src/fpga/src/host_spmv_bscsr.cpp:323-397
 *                                                             src/gpu/host_spmv_topk_csr_gpu.cu:171-231
 *   tkspmv_read          <- SpMV::read_result                 src/fpga/src/host_spmv_bscsr.cpp:399-448
 *                                                             src/gpu/host_spmv_topk_csr_gpu.cu:233-239
 *   tkspmv_mtx_read      <- readMtx / readTuples / mm_read_*  src/common/utils/utils.hpp:372-404,474-520
 *                                                             src/common/utils/mmio.hpp:124-230
 *   tkspmv_sample_vector <- create_sample_vector              src/common/utils/utils.hpp:234-267
 *   tkspmv_options_parse <- Options::Options(argc, argv)      src/common/utils/options.hpp:62-132
 *   tkspmv_generate      <- create_sparse_matrix              src/resources/python/create_matrices.py:58-128
 *
 * Plain pointers and sizes only; no C++/torch types cross this boundary. All functions return a
 * tkspmv_status (0 = OK) unless stated; tkspmv_last_error() returns a thread-local message.
 * The engine NEVER falls back to a CPU path: without a usable HIP device tkspmv_create fails with
 * TKSPMV_ERR_DEVICE.
 */
#ifndef TKSPMV_H
#define TKSPMV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TKSPMV_VERSION 1
#define TKSPMV_MAX_COLS 16384u /* 14 column bits per packed entry (reference MAX_COLS = 1024, types.hpp:55) */
#define TKSPMV_MAX_K 1024      /* reference GPU path: get_topk<<<1,1024>>> => k <= 1024 */

typedef enum {
    TKSPMV_OK = 0,
    TKSPMV_ERR_INVALID = 1,     /* bad argument / descriptor */
    TKSPMV_ERR_NOT_SORTED = 2,  /* COO rows not non-decreasing (reference requires row-major input) */
    TKSPMV_ERR_DEVICE = 3,      /* HIP runtime error or no device */
    TKSPMV_ERR_NOMEM = 4,
    TKSPMV_ERR_IO = 5,          /* file not found / bad MatrixMarket banner */
    TKSPMV_ERR_UNSUPPORTED = 6,
    TKSPMV_ERR_STATE = 7        /* e.g. run before set_query */
} tkspmv_status;

typedef enum {
    TKSPMV_F32 = 0,  /* fp32 values, fp32 x, fp32 accumulate (USE_FLOAT build / GPU hosts) */
    TKSPMV_Q1_7 = 1, /* unsigned fixed point, 1 integer + 7 fraction bits: values, x, products AND sums in 8 bits
                        (the FPGA's ap_ufixed<FIXED_WIDTH,1> real_type with FIXED_WIDTH = 8), see DESIGN.md */
    TKSPMV_Q1_7_WIDE = 2, /* Q1.7 values and products, x block-scaled by a power of two per query, exact (non-wrapping)
                             accumulation: the same 3 B/nnz stream with usable ranking quality */
    TKSPMV_F16 = 3,       /* fp16 values (round to nearest even), fp32 x, fp32 products and sums: 4 B/nnz. The CUDA
                             comparator's half mode (-a, host_spmv_topk_csr_gpu.cu:132-136,152-160) */
    TKSPMV_Q1_7_F32 = 5,  /* BASELINE configs[4] done properly: values stored as Q1.7 bytes (ap_ufixed<8,1,AP_RND,AP_SAT>: rounded
                             to nearest, 3 B/nnz with the column word), x in fp32, fp32 products and sums -- the reduced-
                             precision VALUE STREAM of the FPGA design (fpga_types.hpp:16-23) with the arithmetic of the fp32
                             path. The only error is the quantisation of the values: precision@100 against the fp32 gold 0.97
                             on configs[4] (host_spmv_bscsr.cpp:646-650 is the reference's acceptance metric). Scores are bit-
                             identical to the fp32 engine's on the de-quantised values. */
    TKSPMV_FIXED = 4      /* the FPGA's real_type for any FIXED_WIDTH (types.hpp:20; builds tested by the reference:
                             20/21/25/26/32 bits, test_spmv_topk.py:42-47): ap_ufixed<W,1,AP_TRN_ZERO> with
                             W = desc.fixed_width in [8, 32] -- values, x, every product and every partial sum truncated
                             to W-1 fraction bits, sums wrap at 2.0 (fpga_types.hpp:20,
                             spmv_bscsr_top_k_multicore.hpp:121-141). Values travel as one u32 each (6 B/nnz); ranking and
                             output use the fixed-point score converted to fp32. W = 8 reproduces TKSPMV_Q1_7. */
} tkspmv_precision;

/* Engine variants, the counterpart of the reference's -i/--gpu_impl selector (options.hpp:35, enum GPU_IMPL {CSR,
 * CSR_LIGHTSPMV, COO}: three ways of running the same query on the GPU). All return identical index lists; scores may
 * differ in the last bits (summation order). */
typedef enum {
    TKSPMV_IMPL_STREAM = 0,        /* default: fused streaming kernel over wave-BSCSR packets (batch kernel for sequences) */
    TKSPMV_IMPL_ROW_PER_LANE = 1,  /* one row per lane over the wave-sliced ELL copy (the multi-query kernel with one query
                                      per pass; scores in the gold's summation order). Engines it does not apply to
                                      (reduced precisions, > 1024 columns, ...) run the default. */
    TKSPMV_IMPL_SCORES_SELECT = 2  /* full y = A.x, then an exact radix select over all rows: the structure of the
                                      reference's GPU host (cusparseSpMV + sort, host_spmv_topk_csr_gpu.cu:171-231) */
    /* (3 was rounds 2-4's resident kernel -- one launch serving the reset / run / read loop through pinned memory; since round 4 a launch
       per query through single_kernel is faster on the device and end to end: removed in round 5, tkspmv_create answers TKSPMV_ERR_UNSUPPORTED) */
} tkspmv_impl;

typedef struct tkspmv_engine tkspmv_t;

/* Engine descriptor. Caller owns row/col/val; they are only read during tkspmv_create
 * (the reference keeps raw pointers for the engine lifetime; this engine copies what it needs). */
typedef struct {
    uint32_t rows;            /* logical row count (ids returned are < first_row + rows) */
    uint32_t cols;            /* length of the query vector, <= TKSPMV_MAX_COLS */
    uint64_t nnz;
    const uint32_t *row;      /* [nnz] COO row ids, non-decreasing (reference `x`) */
    const uint32_t *col;      /* [nnz] COO column ids (reference `y`) */
    const float *val;         /* [nnz] values; NULL => all ones (reference -v / ignore_matrix_values) */
    int32_t k;                /* results per query, 1..TKSPMV_MAX_K (reference -k, default 20) */
    int32_t partitions;       /* logical row partitions, reference SPMV_PARTITIONS; 0/1 => one (exact top-k) */
    int32_t k_per_partition;  /* candidates kept per logical partition, reference K (types.hpp:49); 0 => k */
    int32_t precision;        /* tkspmv_precision */
    int32_t device;           /* HIP device ordinal; -1 => current device */
    uint32_t first_row;       /* added to every returned row id (row-sharded multi-GPU) */
    float min_score;          /* rows scoring below this are never returned; gold uses 0 (gold_algorithms.hpp:200) */
    /* tuning knobs, 0 = auto */
    int32_t waves_per_cu;
    int32_t threads_per_wg;
    int32_t nnz_per_lane;     /* 4 or 8 entries per lane per packet; 0 = 4 */
    int32_t stream_replicas;  /* measurement aid: keep R copies of the packet stream in HBM and rotate them per query so
                                 that consecutive queries cannot be served from the 256 MiB Infinity Cache; 0/1 = off */
    int32_t fixed_width;      /* TKSPMV_FIXED: bits per value, 8..32 (0 => 32, the reference's default FIXED_WIDTH,
                                 types.hpp:20); must be 0 for the other precisions */
    int32_t multi_q;          /* queries per matrix pass of tkspmv_enqueue_multi: 0 = off (default), 1, 2, 4 or 8. When set, the
                                 engine keeps a second copy of the matrix in the wave-sliced ELL layout (info.multi_bytes);
                                 info.multi_q tells what the engine uses (8 becomes 4 when k exceeds a quarter of the
                                 threshold groups, 0 when the kernel does not apply) */
    int32_t impl;             /* tkspmv_impl; 0 = default */
    int32_t reserved[1];
} tkspmv_desc;

typedef struct {
    uint32_t rows, cols;
    uint64_t nnz;
    uint64_t packed_entries;      /* nnz + placeholders for empty rows + tail padding */
    uint64_t packed_bytes;        /* bytes of the packet stream + side tables resident in HBM */
    uint64_t algorithmic_bytes;   /* nnz*(sizeof(val)+2) + rows*4 + cols*sizeof(val) + k*8 */
    uint32_t n_packets;
    uint32_t packet_entries;      /* entries per packet = 64 lanes * nnz_per_lane */
    uint32_t n_wave_partitions;   /* physical row ranges: one per wave, or (claim_sets != 0) 8 per claimable set */
    uint32_t packets_per_partition;
    uint32_t grid, block;         /* launch geometry of the streaming kernel */
    uint32_t n_groups;            /* threshold groups publishing maxima */
    uint32_t lds_bytes;
    int32_t k, partitions, k_per_partition, precision, device;
    uint32_t num_cus;
    uint32_t fixed_width;         /* TKSPMV_FIXED: bits per value; 0 otherwise */
    uint32_t multi_q;             /* queries per matrix pass of tkspmv_enqueue_multi; 0 = this engine has no multi-query kernel */
    uint32_t multi_pack_us;       /* tkspmv_create: microseconds spent packing the wave-sliced ELL copy (0 without it) */
    uint64_t multi_bytes;         /* bytes of the wave-sliced ELL copy the multi-query kernel streams (0 without it) */
    uint32_t pack_us;             /* tkspmv_create: microseconds spent packing (on the device: upload of the COO included) */
    uint32_t pack_on_device;      /* 1: the stream was packed by the device packer (default), 0: by the host packer */
    uint32_t claim_sets;          /* always 0 (round 3's dynamically claimed partition sets are gone; the field keeps the layout) */
    uint32_t batch_mode;          /* back-to-back queries: bits 0-7 = selector workgroups of a launch (4 on small matrices), bits 8-15 =
                                     workgroup-local thresholds (0: the device-wide exchange; 1 / 2: see DESIGN.md 3.0b); bits 16-31 = the
                                     n_wave_partitions_hint this engine packed with (tkspmv_pack with the same hint cuts the same partitions) */
    uint64_t state_bytes;         /* device memory of the exchange state of back-to-back queries: the per-query sets (published maxima,
                                     threshold word, one slot per wave, the workgroups' records) and the overflow lists (8 B per row each;
                                     two per engine since round 4, one per query of a launch before) */
} tkspmv_info;

typedef struct {
    double stream_kernel_ns;   /* average hipEvent bracket around the fused streaming kernel (includes event_bracket_ns) */
    double select_kernel_ns;   /* average device time of the final candidate-select kernel */
    double query_ns;           /* average device time per query, back-to-back enqueue (all kernels) */
    double candidates_avg;     /* average number of candidates reaching the select stage */
    double scores_kernel_ns;   /* average device time of the SpMV-only variant (writes the full y; no top-k) */
    double slow_paths_avg;     /* TKSPMV_STATS=1: wave-packets per query that took the candidate path */
    double appended_avg;       /* TKSPMV_STATS=1: rows per query appended to the in-LDS candidate lists */
    double event_bracket_ns;   /* the same event bracket around an empty kernel of the same geometry */
    uint32_t n_queries;
    uint32_t reserved[1];
} tkspmv_timing;

/* ---- engine ------------------------------------------------------------------------------- */
int tkspmv_create(tkspmv_t **out, const tkspmv_desc *desc);
void tkspmv_destroy(tkspmv_t *e);
int tkspmv_get_info(const tkspmv_t *e, tkspmv_info *info);

/* Install a new query vector (host pointer, `cols` floats) -- the reference's reset(vec), host_spmv_bscsr.cpp:354-358. Where the
 * device exposes its memory through a large PCIe BAR (checked by a round trip at create time; option BAR_X) x is stored straight
 * into device memory with CPU stores and a store fence; else it goes through a pinned staging copy and an asynchronous upload.
 * Returns ns in *elapsed_ns if non-NULL. */
int tkspmv_set_query(tkspmv_t *e, const float *host_x, double *elapsed_ns);
/* Same, but x already lives in device memory (no copy; pointer must stay valid until the run completes). */
int tkspmv_set_query_device(tkspmv_t *e, const float *dev_x);

/* Run one query on the current vector and wait. *kernel_ns = device time of the launch(es): where the result travels through
 * host-visible memory (the default) the launch's OWN span -- its s_memrealtime clock, 100 MHz: 10 ns per tick, from the first
 * workgroup's entry to the raising of the result flag; dispatch latency and the instructions behind the flag are outside it --,
 * a hipEvent bracket otherwise (TKSPMV_RUN_EVENTS=1 asks for the bracket; it costs ~6 us around an empty kernel). Engines that
 * stream with workgroup-local thresholds (tkspmv_info.batch_mode bits 8-15) launch single_kernel here; if its selection's
 * check fails -- a query unlike the ones before it -- the query runs again through the exact launch and both spans are added. */
int tkspmv_run(tkspmv_t *e, double *kernel_ns);
/* Enqueue one query on `stream` (hipStream_t cast to void*; NULL => engine stream), no host sync.
 * dev_idx/dev_val: optional device output buffers of k entries (NULL => engine-owned result buffers). */
int tkspmv_enqueue(tkspmv_t *e, const float *dev_x, uint32_t *dev_idx, float *dev_val, void *stream);
/* Enqueue `count` queries back to back (query i uses dev_xs + (i % n_x) * cols), no host sync; the engine-owned
 * result buffers end up holding the last query's top-k. Launch scheme: the batch kernel, up to 32 queries per launch,
 * selections running beside the stream inside the launch (wide x or large k: one launch per query with
 * the selection of query i inside the launch of query i+1 and a small closing launch); on a caller's stream the sequence is
 * complete in stream order when the call returns, on the engine's stream after tkspmv_synchronize (see there). */
int tkspmv_enqueue_many(tkspmv_t *e, const float *dev_xs, int32_t n_x, int32_t count, void *stream);
/* A batch of `count` queries (the loop of the reference's drivers, host_spmv_bscsr.cpp main: for each test vector
 * reset -> operator() -> read_result): query i = dev_xs + i * cols, its k results go to dev_idx + i * k and
 * dev_val + i * k (both NULL => engine-owned buffers, last query wins). Same launch scheme as enqueue_many. */
int tkspmv_enqueue_batch(tkspmv_t *e, const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val,
                         void *stream);
/* Several queries per pass over the matrix (an extension: the reference streams its matrix once per
 * query vector, host_spmv_bscsr.cpp:602-622). Same arguments and result contract as tkspmv_enqueue_batch. Needs
 * desc.multi_q != 0 at create time: info.multi_q queries share every chunk of the wave-sliced ELL copy of the matrix that
 * is loaded (x is held multi_q times in LDS; every query has its own accumulators, threshold, candidate lists and
 * exchange state). One lane owns one row and sums it in the row's own entry order, i.e. in the order of the reference's
 * gold (gold_algorithms.hpp:188-246): scores are bit-identical to the gold's sequential fp32 sums (tkspmv_run's differ
 * from those in the last bits: its sums follow the packet layout). Engines without the kernel (info.multi_q == 0:
 * desc.multi_q = 0, reduced precisions, more than 1024 columns, fewer publishing groups than k) run the ordinary
 * back-to-back sequence. dev_xs = NULL with count = 1: the vector installed by tkspmv_set_query.
 * At multi_q = 1 or 2 a launch makes several passes (8 / 4; option MULTI_PASSES), each over its own queries: invisible to the
 * caller, results and their order are the same. */
int tkspmv_enqueue_multi(tkspmv_t *e, const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream);
/* tkspmv_time_queries for the multi-query path: *ns_per_query = time of the whole sequence / iters. */
int tkspmv_time_multi(tkspmv_t *e, const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query);
/* Waits for the engine's stream. Results of queries enqueued with stream = NULL (the engine's own stream) are FINAL when this call
 * (or tkspmv_read, or any other engine call that waits for that stream) has returned: engines that stream with checked
 * workgroup-local thresholds (tkspmv_info.batch_mode bits 8-15) look at their launches' verdicts here and run a query whose check
 * failed again through the exact kernel -- the device-side equivalent of the reference's wait() after operator()
 * (host_spmv_bscsr.cpp:354-358). The query vectors of such launches must stay unchanged until then. (Option REPAIR=stream keeps the
 * exact launch behind every local launch in the stream, as it always is on a caller's stream: results are then final in stream
 * order, for 1-2 % of the time per query.) */
int tkspmv_synchronize(tkspmv_t *e);

/* Copy back the k results of the last completed query, sorted by (score desc, row desc)
 * (sort_tuples order, evaluation_utils.hpp:40-62). *n receives the count (always k; entries past the
 * number of qualifying rows are (0, 0.0f), as the gold's zero-initialised list). */
int tkspmv_read(tkspmv_t *e, uint32_t *idx, float *val, int32_t *n);
/* Device pointers of the engine-owned result buffers (k entries each). */
int tkspmv_result_device(tkspmv_t *e, const uint32_t **dev_idx, const float **dev_val);
/* Debug/verification: full score vector y = A.x of the current query (rows floats, host). */
int tkspmv_scores(tkspmv_t *e, float *host_y);

/* Diagnostics (TKSPMV_TRACE=1 at create time): per-wave 100 MHz wall-clock stamps of the last four launches,
 * [4][grid+1][9 waves][8] words: entry, x staged, first packet reduced, stream loop done, deferred packets judged,
 * flush done. tools/timeline.py turns them into a timeline. TKSPMV_ERR_STATE when tracing is off. */
int tkspmv_debug_trace(tkspmv_t *e, uint64_t *host, uint64_t max_words, uint64_t *words);
/* Diagnostics of the checked thresholds of back-to-back queries (tkspmv_info.batch_mode; no reference counterpart): out[0] =
 * selections whose check failed so far (each sent its query through the repair launch), out[1] = current suspension length of
 * carried thresholds (selections), out[2] = selections to go until they are used again, out[3] = batch launches so far, out[4] =
 * launches for which the workgroup-local thresholds as a whole stay switched off (a launch of which a quarter failed closes them
 * for 8 .. 1024 launches: data that keeps its best rows together), out[5] = the length of the latest such closure. n >= 6.
 * With n >= 10 also tkspmv_run's single-query kernel (workgroup-local thresholds, checked by its selection): out[6] = its
 * launches, out[7] = queries run again through the exact launch because the check failed, out[8] = failed checks as the device
 * counted them, out[9] = selections to go until carried thresholds are used again. With n >= 12: out[10] = batch launches that
 * went out without a repair launch behind them (see tkspmv_synchronize), out[11] = the late repairs that had to follow after all.
 * With n >= 14: out[12] = the pacing of back-to-back queries in force (pause quantum | levels << 8 | uniform pause << 16; units of
 * 128 cycles per packet), out[13] = microseconds tkspmv_create spent measuring it on this box (option AUTOTUNE; 0: static default).
 * With n >= 19 and option STATS, summed over the tkspmv_time_multi calls so far: out[14] = queries, out[15] = waves of the multi-query
 * kernel that ran into their bounded wait for a threshold, out[16] = the ticks (10 ns) spent there, out[17] / out[18] = rows offered
 * to / overflowed from the candidate lists.
 * Synchronises the engine's stream. */
int tkspmv_debug_counters(tkspmv_t *e, uint64_t *out, int32_t n);

/* `iters` queries back to back on the engine stream (cycling over n_x device-resident vectors), ONE hipEvent pair
 * around the batch: *ns_per_query = batch time / iters. Nothing else is launched (for profiler runs). */
int tkspmv_time_queries(tkspmv_t *e, const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query);
/* `reps` such batches back to back, all enqueued before the first wait, a hipEvent between consecutive ones: ns_per_query[r] =
 * time of batch r / iters. The GPU does not idle between batches (a batch that follows a host-side gap runs 10-25 % slower for
 * about a millisecond: power management), so median and p95 of these are the kernel's under sustained load. reps <= 4096. */
int tkspmv_time_query_batches(tkspmv_t *e, const float *dev_xs, int32_t n_x, int32_t iters, int32_t reps, double *ns_per_query);
/* Measurement aid: the reference's loop -- reset(x) from host memory, operator(), read_result(): host_spmv_bscsr.cpp:602-632 --
 * run `iters` times in native code over n_x host vectors (stride cols floats): loop_ns[i] = the host's steady clock around
 * tkspmv_set_query + tkspmv_run + tkspmv_read of iteration i (what the reference reports as hw_full_exec_time + its reset),
 * kernel_ns[i] (may be NULL) = tkspmv_run's own figure. A caller that goes through a foreign-function layer pays that layer's
 * transitions on top (three per iteration). */
int tkspmv_time_host_loop(tkspmv_t *e, const float *host_xs, int32_t n_x, int32_t iters, double *loop_ns, double *kernel_ns);
/* Measurement aid: `passes` passes over the engine's packet stream (rotating its stream copies) by a kernel with the
 * engine's launch geometry that only LOADS the packets -- no x, no arithmetic, no selection -- in ONE launch, inside one
 * hipEvent pair: *ns_per_pass = what moving the stream from HBM into registers costs on this GPU. bench.py prints the
 * streaming kernels' time against it next to the fraction of the 8 TB/s specification. */
int tkspmv_time_stream_read(tkspmv_t *e, int32_t passes, double *ns_per_pass);

/* Benchmark helper: run `iters` queries cycling over `n_x` device-resident vectors (stride cols floats)
 * back-to-back on the engine stream, timed with hipEvents on that stream. */
int tkspmv_profile(tkspmv_t *e, const float *dev_xs, int32_t n_x, int32_t iters, tkspmv_timing *out);

/* Engine options (no reference counterpart: the reference fixes its variants at compile time, types.hpp / the Makefile's -D flags).
 * Every switch that changes what an engine does is one row of a documented table (csrc/options.cpp; tkspmv_option_info lists it at
 * run time: `bin/approximate-spmv-mi355x-topk --options` prints it): threshold scheme, batching, layouts, host hand-off, multi-GPU
 * merge, diagnostics. An option is read when an engine, a packed matrix or a communicator is CREATED; tkspmv_set_option(name, value)
 * sets it for the creations that follow in this process (value NULL: back to unset = the engine's own default), and the
 * environment variable TKSPMV_<NAME> is the same option for shell-driven runs (a value set through this call wins).
 * Returns TKSPMV_ERR_INVALID for a name that is not in the table. tkspmv_get_option: the current value or NULL.
 * tkspmv_option_info(i, ...): row i of the table (0 <= i < tkspmv_option_count()); any out pointer may be NULL. */
int tkspmv_set_option(const char *name, const char *value);
const char *tkspmv_get_option(const char *name);
int tkspmv_option_count(void);
int tkspmv_option_info(int32_t i, const char **name, const char **kind, const char **values, const char **doc);

const char *tkspmv_last_error(void);
int tkspmv_device_count(void);

/* ---- row-sharded multi-GPU step (one process per GPU) ---------------------------------------------------------
 * The reference merges its row partitions on the host (src/fpga/src/host_spmv_bscsr.cpp:399-448, global id =
 * local + first_row at :415). One level up: every rank owns an engine over its row shard (desc.first_row = first
 * global row), per query the local fused kernel, ONE RCCL all-gather of k (row, score) pairs per rank and a merge
 * kernel. Queries are exchanged in batches (default 32, TKSPMV_DIST_BATCH / tkspmv_dist_set_batch; 1 = every query on
 * its own): the local step of a batch is launched as one back-to-back sequence when the batch closes (passes of several
 * queries each if the engine was created with desc.multi_q), then one
 * all-gather and one merge launch per batch on a side stream, overlapping the local step of the next batch (two buffer
 * sets). synchronize / read flush an open batch; a query vector passed to tkspmv_dist_enqueue must stay valid until then. Every rank must issue the same call sequence. RCCL is
 * loaded with dlopen inside tkspmv_dist_create / tkspmv_dist_unique_id: TKSPMV_ERR_UNSUPPORTED if it cannot be loaded. */
typedef struct tkspmv_dist tkspmv_dist_t;
/* rank 0 creates the 128-byte RCCL unique id and ships it to the other ranks (any transport). */
int tkspmv_dist_unique_id(uint8_t *out128);
/* id128 may be NULL when world == 1. The engine must outlive the returned object. */
int tkspmv_dist_create(tkspmv_dist_t **out, tkspmv_t *engine, const uint8_t *id128, int32_t rank, int32_t world);
int tkspmv_dist_set_batch(tkspmv_dist_t *d, int32_t batch);                    /* 1..32 queries per exchange */
int tkspmv_dist_enqueue(tkspmv_dist_t *d, const float *dev_x);                 /* one query, asynchronous */
int tkspmv_dist_run_many(tkspmv_dist_t *d, const float *dev_xs, int32_t n_x, int32_t count);
int tkspmv_dist_synchronize(tkspmv_dist_t *d);
/* merged (global) top-k of the most recently enqueued query (host buffers of k entries); waits for it */
int tkspmv_dist_read(tkspmv_dist_t *d, uint32_t *idx, float *val, int32_t *n);
/* The exchange step alone, for measurement (collective: every rank calls it with the same arguments): `iters` times the
 * all-gather of one full batch (batch x 2k words per rank) + the merge launch, back to back on the communication stream,
 * one hipEvent pair around them. */
int tkspmv_dist_read_batch(tkspmv_dist_t *d, uint32_t *idx, float *val, int32_t *n_q);  /* every list of the last exchanged batch: [n_q][k] */
int tkspmv_dist_time_exchange(tkspmv_dist_t *d, int32_t iters, double *ns_per_exchange);
void tkspmv_dist_destroy(tkspmv_dist_t *d);
const char *tkspmv_dist_last_error(void);
/* The merge step alone: dev_gathered = [world][2][k] u32 (row ids, then score bits) -> k best, sort_tuples order. */
/* The merge of an exchange batch as the pipelined step launches it: dev_gathered [world][n_q][2][k], one block per query,
 * results [n_q][k] (n_q <= 32). Lifts the host-side merge of host_spmv_bscsr.cpp:399-448 for a batch of queries. */
int tkspmv_merge_topk_batch(const uint32_t *dev_gathered, int32_t world, int32_t n_q, int32_t k, uint32_t *dev_idx,
                            float *dev_val, void *stream);
/* Rehearsal without RCCL (which refuses two ranks on one device): the all-gather of the pipelined step goes through host
 * buffers and this callback (send: this rank's bytes_per_rank bytes; recv: world such blocks, rank-major; 0 = success);
 * everything else -- batches, buffer rotation, events, merge launch -- is the real step. With TKSPMV_DIST_NO_NCCL=1
 * tkspmv_dist_create builds no communicator for world > 1 and the callback is mandatory. */
typedef int (*tkspmv_host_allgather_fn)(const void *send, void *recv, uint64_t bytes_per_rank, void *user);
int tkspmv_dist_set_host_exchange(tkspmv_dist_t *d, tkspmv_host_allgather_fn fn, void *user);
int tkspmv_merge_topk(const uint32_t *dev_gathered, int32_t world, int32_t k, uint32_t *dev_idx, float *dev_val,
                      void *stream);

/* ---- host-side helpers (no GPU needed) --------------------------------------------------------- */
typedef struct {
    uint32_t rows, cols;   /* from the size line */
    uint64_t nnz;
    uint32_t *row, *col;   /* malloc'd, release with tkspmv_mtx_free */
    float *val;
    uint32_t num_rows_coo; /* max(row)+1, as coo_t derives it (coo_matrix.hpp:21-27) */
    int32_t index_base;    /* index base actually applied: 0 or 1 */
    int32_t symmetric;     /* banner said symmetric */
} tkspmv_coo;

/* index_base: 0 = file is zero-indexed (the reference's compiled-in behaviour, zero_indexed_file=true),
 *             1 = file is one-indexed (create_matrices.py output), -1 = auto-detect (min index == 0 ? 0 : 1).
 * read_values = 0 => all values 1.0 (reference -v). sort = non-zero => (row,col) sort like customSort. */
int tkspmv_mtx_read(const char *path, int32_t index_base, int32_t read_values, int32_t sort, tkspmv_coo *out);
void tkspmv_mtx_free(tkspmv_coo *m);
int tkspmv_mtx_write(const char *path, uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row,
                     const uint32_t *col, const float *val, int32_t index_base, int32_t precision);

/* create_sample_vector(vec,size,random,sum_to_one,norm_one,seed): seed==0 => std::random_device. */
int tkspmv_sample_vector(float *vec, int32_t size, int32_t random, int32_t sum_to_one, int32_t norm_one, int32_t seed);

/* Synthetic matrix with the distributions of create_matrices.py (dist: 0 = uniform, 1 = gamma). Own PRNG. */
int tkspmv_generate(uint32_t rows, uint32_t cols, uint32_t avg_nnz, int32_t dist, uint64_t seed, tkspmv_coo *out);
/* Rows [row_begin, row_end) of that matrix with LOCAL row ids (row - row_begin): every row has its own PRNG streams, so a
 * slice equals the corresponding rows of the whole matrix -- what one rank of a row-sharded job builds (BASELINE configs[3]:
 * 10M rows over 8 GPUs; the 10M-row COO never exists in one process). tkspmv_generate_degrees: the row lengths alone
 * (deg[row_end - row_begin]), from which nnz-balanced shard bounds are computed without generating any entry. */
int tkspmv_generate_rows(uint32_t row_begin, uint32_t row_end, uint32_t cols, uint32_t avg_nnz, int32_t dist, uint64_t seed,
                         tkspmv_coo *out);
int tkspmv_generate_degrees(uint32_t row_begin, uint32_t row_end, uint32_t avg_nnz, int32_t dist, uint64_t seed,
                            uint32_t *deg);

typedef struct {
    char matrix_path[1024];
    int32_t use_sample_matrix, reset, num_tests, debug, ignore_matrix_values, top_k_value;
    char xclbin_path[1024];
    int32_t gpu_impl, use_half_precision_gpu, block_size_1d, block_size_2d, num_blocks;
} tkspmv_options;
int tkspmv_options_parse(int argc, char **argv, tkspmv_options *out);

/* Layout introspection for tests: pack on the host, decode back to COO (+ placeholders dropped). */
typedef struct tkspmv_packed tkspmv_packed;
int tkspmv_pack(const tkspmv_desc *desc, uint32_t n_wave_partitions_hint, tkspmv_packed **out);
int tkspmv_packed_info(const tkspmv_packed *p, tkspmv_info *info);
/* Decodes into caller arrays sized >= nnz; returns the number of decoded entries in *n. */
int tkspmv_packed_decode(const tkspmv_packed *p, uint32_t *row, uint32_t *col, float *val, uint64_t *n);
/* Raw views (host memory owned by p). */
int tkspmv_packed_raw(const tkspmv_packed *p, const void **packets, uint64_t *packet_bytes, const uint32_t **pkt_row,
                      const uint32_t **part_first, const uint32_t **part_count, uint32_t *n_parts);
void tkspmv_packed_free(tkspmv_packed *p);
/* The DEVICE packer (what tkspmv_create uses by default): packs desc's COO with HIP kernels on desc->device
 * and copies the result back into a tkspmv_packed, so that it can be compared byte for byte with tkspmv_pack's
 * (tkspmv_packed_raw) or saved. ms[0] = upload of the COO, ms[1] = packing kernels (ms may be NULL). Needs a GPU. */
int tkspmv_pack_device(const tkspmv_desc *desc, uint32_t n_wave_partitions_hint, tkspmv_packed **out, double *ms);
/* The same for the wave-sliced ELL layout of the multi-query kernel (wsell.hpp): packs desc's COO for
 * n_wave_partitions_hint waves and decodes it again into caller arrays sized >= nnz (entries grouped by row, rows in
 * stream order). info[0..5] = slices, chunks, padded entries, partitions, stream bytes, most chunks in one partition. */
int tkspmv_sell_roundtrip(const tkspmv_desc *desc, uint32_t n_wave_partitions_hint, uint32_t *row, uint32_t *col, float *val,
                          uint64_t *n, uint64_t *info);
/* Packs desc's COO into that layout twice -- on the host and with the device packer tkspmv_create uses (plan on the
 * host, fill kernel on desc->device) -- and compares the two byte for byte, side tables included.
 * info[0] = 1 if identical, info[1] = stream bytes, info[2] = chunks; ms[0] = host packer, ms[1..3] = device packer:
 * plan, uploads (the COO included), fill kernel (ms may be NULL). Needs a GPU. */
int tkspmv_sell_pack_device_check(const tkspmv_desc *desc, uint32_t n_wave_partitions_hint, uint64_t *info, double *ms);

/* ---- packed-matrix cache ------------------------------------------------------------------------
 * The reference parses the MatrixMarket text (utils.hpp:380-388, minutes at 10^7 rows) and packs
 * (host_spmv_bscsr.cpp:133-248, `hw_setup_time_ms`) on every run. Here the packed matrix can be written once
 * (".tkspmv": 128-byte header, packet stream, side tables, checksum) and an engine created straight from it.
 * tkspmv_wave_partitions: how many streaming waves a launch on desc->device has (pass it to tkspmv_pack as the hint
 * so that the file suits that GPU; a file with MORE partitions is rejected with TKSPMV_ERR_UNSUPPORTED unless the engine
 * deals the partitions out dynamically -- fp32 values, at most 1024 columns: tkspmv_info.claim_sets -- where any count
 * works and tkspmv_create itself cuts ~2 sets of 8 per workgroup; fewer is fine everywhere). tkspmv_packed_load: TKSPMV_ERR_IO for a missing, truncated,
 * inconsistent or corrupted file. tkspmv_create_packed: rows/cols/nnz and the value type come from the packed matrix,
 * everything else (k, device, min_score, first_row, stream_replicas, TKSPMV_Q1_7 vs TKSPMV_Q1_7_WIDE) from desc. */
int tkspmv_wave_partitions(const tkspmv_desc *desc, uint32_t *n);
int tkspmv_packed_save(const tkspmv_packed *p, const char *path);
int tkspmv_packed_load(const char *path, tkspmv_packed **out);
int tkspmv_create_packed(tkspmv_t **out, const tkspmv_packed *p, const tkspmv_desc *desc);

#ifdef __cplusplus
}
#endif
#endif /* TKSPMV_H */
