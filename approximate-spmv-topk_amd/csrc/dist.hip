// dist.hip -- row-sharded multi-GPU step in native code: per query, the local fused kernel, ONE RCCL all-gather of
// k (row, score) pairs per rank over xGMI, and a merge kernel; queries are exchanged in batches (default 8: one all-gather and one merge
// launch per batch) on a side stream while the stream kernels of the next batch already run (two buffer sets, events
// both ways).
//
// The reference is single-device: its row partitions are merged on the host (src/fpga/src/host_spmv_bscsr.cpp:399-448,
// `local + first_row` at :415). This is that merge one level up. RCCL is loaded with dlopen at
// tkspmv_dist_create, so the library has no link-time dependency on it and a missing/incompatible RCCL is an error
// code, not a crash (bench.py then falls back to torch.distributed for the exchange).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tkspmv.h"
#include "engine.hpp"
#include "options.hpp"

namespace tkspmv {

// ---- RCCL, loaded at run time. Types and signatures come from rccl.h itself (decltype of its declarations): nothing of
// its ABI is restated here; only the symbols are resolved with dlsym instead of at link time. --------------------------
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string &err) {
        if (handle) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            err = std::string("cannot load RCCL: ") + dlerror();
            return false;
        }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(handle, "ncclAllGather"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy) {
            err = "RCCL is missing an expected symbol";
            handle = nullptr;
            return false;
        }
        return true;
    }
};
static Rccl g_rccl;
static_assert(sizeof(ncclUniqueId) == 128, "tkspmv_dist_unique_id ships 128 bytes");

// ---- merge kernel ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_key_d(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// 256 threads and as much LDS as the batch needs (8 bytes per gathered pair): a merge block must fit on a CU BESIDE two workgroups of
// the persistent batch kernel (18 of its 32 wave slots) -- with 1024 threads and a fixed 64 KB it could not, the merge of batch b
// waited for the local kernel of batch b+1 to end, and that kernel's successor for the merge (rocprofv3: 113 us per merge launch).
constexpr uint32_t MERGE_THREADS = 256;
constexpr uint32_t MERGE_MAX = 8184;  // world * k (+ 8 padding keys: 64 KB of dynamic LDS)

// gathered: [world][n_q][2][k] u32 (row ids, then score bits); block q merges query q of the batch. Output: the k best
// by (score desc, row desc); equal keys (the (0, 0.0) fillers several shards may contribute) are ordered by position,
// so ranks stay unique.
__global__ void __launch_bounds__(MERGE_THREADS) merge_kernel(const uint32_t *__restrict__ gathered, uint32_t world,
                                                              uint32_t k, uint32_t *__restrict__ out_idx,
                                                              float *__restrict__ out_val) {
    extern __shared__ unsigned long long keys[];  // [world * k + 8]
    const uint32_t n = world * k, tid = threadIdx.x, q = blockIdx.x, n_q = gridDim.x;
    out_idx += (size_t)q * k;
    out_val += (size_t)q * k;
    for (uint32_t i = tid; i < n; i += MERGE_THREADS) {
        const uint32_t r = i / k, j = i % k;
        const uint32_t *src = gathered + ((size_t)r * n_q + q) * 2 * k;
        keys[i] = ((unsigned long long)order_key_d(__uint_as_float(src[k + j])) << 32) | src[j];
    }
    if (tid < 8) keys[n + tid] = 0ull;
    __syncthreads();
    const uint32_t n_pad = (n + 7u) & ~7u;
    for (uint32_t i = tid; i < n; i += MERGE_THREADS) {
        const unsigned long long kx = keys[i];
        uint32_t r = 0;
        for (uint32_t j = 0; j < n_pad; j += 8) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const unsigned long long o = keys[j + u];
                r += (o > kx) || (o == kx && (j + u) < i);
            }
        }
        if (r < k) {
            const uint32_t key32 = (uint32_t)(kx >> 32);
            const uint32_t u = (key32 & 0x80000000u) ? (key32 & 0x7FFFFFFFu) : ~key32;
            out_idx[r] = (uint32_t)(kx & 0xFFFFFFFFull);
            out_val[r] = __uint_as_float(u);
        }
    }
}

static inline size_t merge_lds(uint32_t world, uint32_t k) { return ((size_t)world * k + 8u) * sizeof(unsigned long long); }

constexpr int MAX_BATCH = 32;

struct Dist {
    Engine *engine = nullptr;
    int device = 0, rank = 0, world = 1, k = 0;
    bool use_nccl = false;  // world > 1, or TKSPMV_DIST_FORCE_NCCL=1 (exercises the RCCL calls with one rank)
    ncclComm_t comm = nullptr;
    hipStream_t compute = nullptr, comm_stream = nullptr;
    hipEvent_t ev_comp[2] = {nullptr, nullptr}, ev_merge[2] = {nullptr, nullptr};
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // tkspmv_dist_time_exchange
    // Queries are exchanged in batches of up to `batch`: one all-gather and one merge launch per batch, so the
    // collective's latency and the host's enqueue cost are paid once per batch. Two buffer sets alternate.
    int batch = MAX_BATCH;
    uint32_t *local[2] = {nullptr, nullptr};     // [batch][2k]: k row ids, k score bits per query
    uint32_t *gathered[2] = {nullptr, nullptr};  // [world][n_q][2k]
    uint32_t *out_idx[2] = {nullptr, nullptr};   // [batch][k]
    float *out_val[2] = {nullptr, nullptr};
    uint64_t steps = 0;    // queries enqueued
    uint64_t flushes = 0;  // batches exchanged
    int fill = 0;          // queries in the open batch (buffer set flushes & 1)
    const float *pend_x[MAX_BATCH] = {};  // their query vectors: the local kernels are launched when the batch closes
    uint32_t *pend_idx[MAX_BATCH] = {};
    float *pend_val[MAX_BATCH] = {};
    int last_set = -1, last_slot = -1;  // where the most recent query's merged result lands
    int last_batch_set = -1, last_batch_n = 0;  // the most recently exchanged batch (tkspmv_dist_read_batch)
    tkspmv_host_allgather_fn host_fn = nullptr;  // rehearsal: the all-gather through host buffers (tkspmv_dist_set_host_exchange)
    void *host_user = nullptr;
    std::vector<uint32_t> host_send, host_recv;
};

}  // namespace tkspmv

using namespace tkspmv;

struct tkspmv_dist {
    Dist d;
};

static thread_local std::string g_dist_err;
static int dfail(int code, const std::string &msg) {
    g_dist_err = msg;
    return code;
}
#define DHIP(expr)                                                                                          \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) return dfail(TKSPMV_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

extern "C" {

void tkspmv_dist_destroy(tkspmv_dist_t *h);

const char *tkspmv_dist_last_error(void) { return g_dist_err.c_str(); }

int tkspmv_dist_unique_id(uint8_t *out128) {
    if (!out128) return dfail(TKSPMV_ERR_INVALID, "NULL argument");
    std::string err;
    if (!g_rccl.load(err)) return dfail(TKSPMV_ERR_UNSUPPORTED, err);
    ncclUniqueId id;
    const ncclResult_t rc = g_rccl.GetUniqueId(&id);
    if (rc != ncclSuccess) return dfail(TKSPMV_ERR_DEVICE, "ncclGetUniqueId failed");
    std::memcpy(out128, id.internal, 128);
    return TKSPMV_OK;
}

int tkspmv_merge_topk(const uint32_t *dev_gathered, int32_t world, int32_t k, uint32_t *dev_idx, float *dev_val,
                      void *stream) {
    if (!dev_gathered || !dev_idx || !dev_val || world < 1 || k < 1 || (uint64_t)world * k > MERGE_MAX)
        return dfail(TKSPMV_ERR_INVALID, "bad arguments to tkspmv_merge_topk (world * k must be <= 8192)");
    hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(MERGE_THREADS), merge_lds((uint32_t)world, (uint32_t)k), (hipStream_t)stream, dev_gathered, (uint32_t)world,
                       (uint32_t)k, dev_idx, dev_val);
    DHIP(hipGetLastError());
    return TKSPMV_OK;
}

// The merge of a whole exchange batch as the pipelined step launches it: dev_gathered is [world][n_q][2][k] (what the
// all-gather of n_q * 2k words per rank leaves behind), one block per query, results to dev_idx / dev_val [n_q][k].
int tkspmv_merge_topk_batch(const uint32_t *dev_gathered, int32_t world, int32_t n_q, int32_t k, uint32_t *dev_idx, float *dev_val,
                            void *stream) {
    if (!dev_gathered || !dev_idx || !dev_val || world < 1 || k < 1 || n_q < 1 || n_q > MAX_BATCH || (uint64_t)world * k > MERGE_MAX)
        return dfail(TKSPMV_ERR_INVALID, "bad arguments to tkspmv_merge_topk_batch (world * k must be <= 8192, n_q in [1, 32])");
    hipLaunchKernelGGL(merge_kernel, dim3((uint32_t)n_q), dim3(MERGE_THREADS), merge_lds((uint32_t)world, (uint32_t)k), (hipStream_t)stream, dev_gathered, (uint32_t)world,
                       (uint32_t)k, dev_idx, dev_val);
    DHIP(hipGetLastError());
    return TKSPMV_OK;
}

// Rehearsal of the pipelined step without RCCL (which refuses two ranks on one device): the all-gather of dist_flush is replaced
// by a device-to-host copy of this rank's block, a callback that exchanges host buffers between the ranks (e.g. torch.distributed
// over gloo), and a host-to-device copy of the gathered blocks -- buffer rotation, events, partial batches and the merge launch
// are the real ones. send: this rank's n_q * 2k words; recv: room for world such blocks, rank-major.
int tkspmv_dist_set_host_exchange(tkspmv_dist_t *h, tkspmv_host_allgather_fn fn, void *user) {
    if (!h) return dfail(TKSPMV_ERR_INVALID, "NULL argument");
    if (h->d.fill != 0) return dfail(TKSPMV_ERR_STATE, "a batch is open: synchronize first");
    h->d.host_fn = fn;
    h->d.host_user = user;
    return TKSPMV_OK;
}

int tkspmv_dist_create(tkspmv_dist_t **out, tkspmv_t *engine, const uint8_t *id128, int32_t rank, int32_t world) {
    if (!out || !engine || world < 1 || rank < 0 || rank >= world) return dfail(TKSPMV_ERR_INVALID, "bad arguments");
    *out = nullptr;
    tkspmv_info info;
    engine->e->info(&info);
    if ((uint64_t)world * info.k > MERGE_MAX) return dfail(TKSPMV_ERR_INVALID, "world * k must be <= 8184");
    tkspmv_dist *h = new tkspmv_dist();
    Dist &d = h->d;
    d.engine = engine->e;
    d.device = info.device;
    d.rank = rank;
    d.world = world;
    d.k = info.k;
    // every failure below releases what has been created so far (communicator, streams, events, buffers)
    auto bail = [&](int code, const std::string &msg) {
        tkspmv_dist_destroy(h);
        return dfail(code, msg);
    };
#define DHIP_OR_BAIL(expr)                                                                                    \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return bail(TKSPMV_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)
    DHIP_OR_BAIL(hipSetDevice(d.device));
    // (TKSPMV_DIST_NO_NCCL=1: no communicator -- the exchange goes through tkspmv_dist_set_host_exchange's callback)
    d.use_nccl = (world > 1 && opt("DIST_NO_NCCL") == nullptr) || opt("DIST_FORCE_NCCL") != nullptr;
    if (d.use_nccl) {
        if (!id128) return bail(TKSPMV_ERR_INVALID, "world > 1 needs the unique id of rank 0");
        std::string err;
        if (!g_rccl.load(err)) return bail(TKSPMV_ERR_UNSUPPORTED, err);
        ncclUniqueId id;
        std::memcpy(id.internal, id128, 128);
        const ncclResult_t rc = g_rccl.CommInitRank(&d.comm, world, id, rank);
        if (rc != ncclSuccess) {
            d.comm = nullptr;
            return bail(TKSPMV_ERR_DEVICE, std::string("ncclCommInitRank failed: ") +
                                               (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
        }
    }
    if (const char *e = opt("DIST_BATCH")) d.batch = atoi(e);
    if (d.batch < 1) d.batch = 1;
    if (d.batch > MAX_BATCH) d.batch = MAX_BATCH;
    DHIP_OR_BAIL(hipStreamCreateWithFlags(&d.compute, hipStreamNonBlocking));
    DHIP_OR_BAIL(hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking));
    DHIP_OR_BAIL(hipEventCreate(&d.ev_t0));
    DHIP_OR_BAIL(hipEventCreate(&d.ev_t1));
    for (int b = 0; b < 2; ++b) {
        DHIP_OR_BAIL(hipEventCreateWithFlags(&d.ev_comp[b], hipEventDisableTiming));
        DHIP_OR_BAIL(hipEventCreateWithFlags(&d.ev_merge[b], hipEventDisableTiming));
        DHIP_OR_BAIL(hipMalloc((void **)&d.local[b], (size_t)MAX_BATCH * 2 * d.k * 4));
        DHIP_OR_BAIL(hipMalloc((void **)&d.gathered[b], (size_t)world * MAX_BATCH * 2 * d.k * 4));
        DHIP_OR_BAIL(hipMalloc((void **)&d.out_idx[b], (size_t)MAX_BATCH * d.k * 4));
        DHIP_OR_BAIL(hipMalloc((void **)&d.out_val[b], (size_t)MAX_BATCH * d.k * 4));
        DHIP_OR_BAIL(hipMemset(d.local[b], 0, (size_t)MAX_BATCH * 2 * d.k * 4));
        DHIP_OR_BAIL(hipMemset(d.out_idx[b], 0, (size_t)MAX_BATCH * d.k * 4));
        DHIP_OR_BAIL(hipMemset(d.out_val[b], 0, (size_t)MAX_BATCH * d.k * 4));
    }
    DHIP_OR_BAIL(hipDeviceSynchronize());
#undef DHIP_OR_BAIL
    *out = h;
    return TKSPMV_OK;
}

// The exchange step alone (collective: every rank calls it with the same arguments): `iters` times the all-gather of one
// full batch (batch x 2k words per rank) and the merge launch, back to back on the communication stream, one hipEvent pair
// around them. bench.py reports it beside the whole step (which overlaps the exchange with the next batch's local kernels).
int tkspmv_dist_time_exchange(tkspmv_dist_t *h, int32_t iters, double *ns_per_exchange) {
    if (!h || iters < 1 || !ns_per_exchange) return dfail(TKSPMV_ERR_INVALID, "bad arguments");
    Dist &d = h->d;
    if (d.fill != 0) return dfail(TKSPMV_ERR_STATE, "a batch is open: synchronize first");
    DHIP(hipSetDevice(d.device));
    DHIP(hipStreamSynchronize(d.compute));
    DHIP(hipStreamSynchronize(d.comm_stream));
    const int n_q = d.batch;
    DHIP(hipEventRecord(d.ev_t0, d.comm_stream));
    for (int i = 0; i < iters; ++i) {
        if (d.use_nccl) {
            const ncclResult_t rc = g_rccl.AllGather(d.local[0], d.gathered[0], (size_t)n_q * 2 * d.k, ncclInt32, d.comm, d.comm_stream);
            if (rc != ncclSuccess) return dfail(TKSPMV_ERR_DEVICE, "ncclAllGather failed");
        } else {
            DHIP(hipMemcpyAsync(d.gathered[0], d.local[0], (size_t)n_q * 2 * d.k * 4, hipMemcpyDeviceToDevice, d.comm_stream));
        }
        hipLaunchKernelGGL(merge_kernel, dim3(n_q), dim3(MERGE_THREADS), merge_lds((uint32_t)d.world, (uint32_t)d.k), d.comm_stream, d.gathered[0], (uint32_t)d.world,
                           (uint32_t)d.k, d.out_idx[0], d.out_val[0]);
    }
    DHIP(hipGetLastError());
    DHIP(hipEventRecord(d.ev_t1, d.comm_stream));
    DHIP(hipEventSynchronize(d.ev_t1));
    float ms = 0;
    DHIP(hipEventElapsedTime(&ms, d.ev_t0, d.ev_t1));
    *ns_per_exchange = (double)ms * 1e6 / iters;
    return TKSPMV_OK;
}

int tkspmv_dist_set_batch(tkspmv_dist_t *h, int32_t batch) {
    if (!h || batch < 1 || batch > MAX_BATCH) return dfail(TKSPMV_ERR_INVALID, "batch must be in [1, 32]");
    if (h->d.fill != 0) return dfail(TKSPMV_ERR_STATE, "a batch is open: synchronize first");
    h->d.batch = batch;
    return TKSPMV_OK;
}

// Exchange the open batch: all-gather of fill * 2k words per rank and one merge block per query, on the comm stream.
static int dist_flush(Dist &d) {
    if (d.fill == 0) return TKSPMV_OK;
    const int b = (int)(d.flushes & 1);
    const int n_q = d.fill;
    {
        // The local step of the whole batch as ONE back-to-back sequence (the engine's batch kernel when available; passes
        // of several queries each when the engine was created with desc.multi_q): its launch overhead is paid once per
        // exchange batch, like the collective's.
        if (d.flushes >= 2) DHIP(hipStreamWaitEvent(d.compute, d.ev_merge[b], 0));  // buffer set b is free again
        std::string err;
        int st = d.engine->enqueue_multi_list(d.pend_x, d.pend_idx, d.pend_val, n_q, d.compute, err);
        if (st != TKSPMV_OK) return dfail(st, err);
    }
    DHIP(hipEventRecord(d.ev_comp[b], d.compute));
    DHIP(hipStreamWaitEvent(d.comm_stream, d.ev_comp[b], 0));
    if (d.host_fn) {
        const size_t words = (size_t)n_q * 2 * d.k;
        d.host_send.resize(words);
        d.host_recv.assign(words * d.world, 0u);
        DHIP(hipMemcpyAsync(d.host_send.data(), d.local[b], words * 4, hipMemcpyDeviceToHost, d.comm_stream));
        DHIP(hipStreamSynchronize(d.comm_stream));
        if (d.host_fn(d.host_send.data(), d.host_recv.data(), (uint64_t)words * 4, d.host_user) != 0)
            return dfail(TKSPMV_ERR_DEVICE, "the host exchange callback failed");
        DHIP(hipMemcpyAsync(d.gathered[b], d.host_recv.data(), words * 4 * d.world, hipMemcpyHostToDevice, d.comm_stream));
        DHIP(hipStreamSynchronize(d.comm_stream));  // (host_recv is reused by the next flush)
    } else if (d.use_nccl) {
        const ncclResult_t rc = g_rccl.AllGather(d.local[b], d.gathered[b], (size_t)n_q * 2 * d.k, ncclInt32, d.comm, d.comm_stream);
        if (rc != ncclSuccess)
            return dfail(TKSPMV_ERR_DEVICE, std::string("ncclAllGather failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"));
    } else {
        DHIP(hipMemcpyAsync(d.gathered[b], d.local[b], (size_t)n_q * 2 * d.k * 4, hipMemcpyDeviceToDevice, d.comm_stream));
    }
    hipLaunchKernelGGL(merge_kernel, dim3(n_q), dim3(MERGE_THREADS), merge_lds((uint32_t)d.world, (uint32_t)d.k), d.comm_stream, d.gathered[b], (uint32_t)d.world,
                       (uint32_t)d.k, d.out_idx[b], d.out_val[b]);
    DHIP(hipGetLastError());
    DHIP(hipEventRecord(d.ev_merge[b], d.comm_stream));
    d.last_batch_set = b;
    d.last_batch_n = n_q;
    d.fill = 0;
    ++d.flushes;
    return TKSPMV_OK;
}

// One query, asynchronously: it joins the open batch; a full batch is launched on the compute stream and exchanged on
// the comm stream. Every rank must issue the same sequence of enqueue / synchronize / read calls (they are collective).
int tkspmv_dist_enqueue(tkspmv_dist_t *h, const float *dev_x) {
    if (!h || !dev_x) return dfail(TKSPMV_ERR_INVALID, "NULL argument");
    Dist &d = h->d;
    const int b = (int)(d.flushes & 1);
    DHIP(hipSetDevice(d.device));
    uint32_t *dst = d.local[b] + (size_t)d.fill * 2 * d.k;
    d.pend_x[d.fill] = dev_x;  // must stay valid until the batch has been flushed (batch full, synchronize or read)
    d.pend_idx[d.fill] = dst;
    d.pend_val[d.fill] = reinterpret_cast<float *>(dst + d.k);
    d.last_set = b;
    d.last_slot = d.fill;
    ++d.fill;
    ++d.steps;
    if (d.fill == d.batch) return dist_flush(d);
    return TKSPMV_OK;
}

int tkspmv_dist_run_many(tkspmv_dist_t *h, const float *dev_xs, int32_t n_x, int32_t count) {
    if (!h || !dev_xs || n_x < 1 || count < 0) return dfail(TKSPMV_ERR_INVALID, "bad arguments");
    tkspmv_info info;
    h->d.engine->info(&info);
    for (int i = 0; i < count; ++i) {
        int st = tkspmv_dist_enqueue(h, dev_xs + (size_t)(i % n_x) * info.cols);
        if (st != TKSPMV_OK) return st;
    }
    return dist_flush(h->d);
}

int tkspmv_dist_synchronize(tkspmv_dist_t *h) {
    if (!h) return dfail(TKSPMV_ERR_INVALID, "NULL argument");
    DHIP(hipSetDevice(h->d.device));
    int st = dist_flush(h->d);
    if (st != TKSPMV_OK) return st;
    DHIP(hipStreamSynchronize(h->d.compute));
    DHIP(hipStreamSynchronize(h->d.comm_stream));
    return TKSPMV_OK;
}

// Merged (global) top-k of the most recently enqueued query; waits for it.
int tkspmv_dist_read(tkspmv_dist_t *h, uint32_t *idx, float *val, int32_t *n) {
    if (!h) return dfail(TKSPMV_ERR_INVALID, "NULL argument");
    Dist &d = h->d;
    if (d.steps == 0) return dfail(TKSPMV_ERR_STATE, "no query has been enqueued");
    int st = tkspmv_dist_synchronize(h);
    if (st != TKSPMV_OK) return st;
    const int b = d.last_set;
    const size_t off = (size_t)d.last_slot * d.k;
    if (idx) DHIP(hipMemcpy(idx, d.out_idx[b] + off, (size_t)d.k * 4, hipMemcpyDeviceToHost));
    if (val) DHIP(hipMemcpy(val, d.out_val[b] + off, (size_t)d.k * 4, hipMemcpyDeviceToHost));
    if (n) *n = d.k;
    return TKSPMV_OK;
}

// Every merged list of the most recently exchanged batch (the open batch is flushed first): idx / val [n_q][k], *n_q queries.
int tkspmv_dist_read_batch(tkspmv_dist_t *h, uint32_t *idx, float *val, int32_t *n_q) {
    if (!h || !n_q) return dfail(TKSPMV_ERR_INVALID, "NULL argument");
    Dist &d = h->d;
    if (d.steps == 0) return dfail(TKSPMV_ERR_STATE, "no query has been enqueued");
    int st = tkspmv_dist_synchronize(h);
    if (st != TKSPMV_OK) return st;
    const int b = d.last_batch_set;
    if (idx) DHIP(hipMemcpy(idx, d.out_idx[b], (size_t)d.last_batch_n * d.k * 4, hipMemcpyDeviceToHost));
    if (val) DHIP(hipMemcpy(val, d.out_val[b], (size_t)d.last_batch_n * d.k * 4, hipMemcpyDeviceToHost));
    *n_q = d.last_batch_n;
    return TKSPMV_OK;
}

void tkspmv_dist_destroy(tkspmv_dist_t *h) {
    if (!h) return;
    Dist &d = h->d;
    (void)hipSetDevice(d.device);
    if (d.compute) (void)hipStreamSynchronize(d.compute);
    if (d.comm_stream) (void)hipStreamSynchronize(d.comm_stream);
    if (d.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(d.comm);
    if (d.ev_t0) (void)hipEventDestroy(d.ev_t0);
    if (d.ev_t1) (void)hipEventDestroy(d.ev_t1);
    for (int b = 0; b < 2; ++b) {
        if (d.ev_comp[b]) (void)hipEventDestroy(d.ev_comp[b]);
        if (d.ev_merge[b]) (void)hipEventDestroy(d.ev_merge[b]);
        if (d.local[b]) (void)hipFree(d.local[b]);
        if (d.gathered[b]) (void)hipFree(d.gathered[b]);
        if (d.out_idx[b]) (void)hipFree(d.out_idx[b]);
        if (d.out_val[b]) (void)hipFree(d.out_val[b]);
    }
    if (d.compute) (void)hipStreamDestroy(d.compute);
    if (d.comm_stream) (void)hipStreamDestroy(d.comm_stream);
    delete h;
}

}  // extern "C"
