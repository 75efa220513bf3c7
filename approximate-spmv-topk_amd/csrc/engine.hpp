// engine.hpp -- device engine behind the C ABI (include/tkspmv.h). Mirrors the reference's per-back-end
// `struct SpMV` (host_spmv_bscsr.cpp:79-485, host_spmv_topk_csr_gpu.cu:44-263): setup once, then per
// query reset(vec) -> operator()() -> read_result().
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tkspmv.h"
#include "wbscsr.hpp"

namespace tkspmv {

struct EngineImpl;  // HIP state lives in engine.hip

class Engine {
   public:
    // Returns nullptr and fills err/status on failure.
    // prepacked: a matrix packed earlier (then desc.row/col/val are not read)
    static Engine *create(const tkspmv_desc &desc, std::string &err, int &status, const PackedMatrix *prepacked = nullptr);
    ~Engine();

    int set_query(const float *host_x, double *elapsed_ns, std::string &err);
    int set_query_device(const float *dev_x, std::string &err);
    int run(double *kernel_ns, std::string &err);
    int enqueue(const float *dev_x, uint32_t *dev_idx, float *dev_val, void *stream, std::string &err);
    int enqueue_many(const float *dev_xs, int32_t n_x, int32_t count, void *stream, std::string &err);
    int enqueue_batch(const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream,
                      std::string &err);
    // Queries in passes of several per matrix pass (multi_kernel); same contract as enqueue_batch.
    int enqueue_multi(const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream, std::string &err);
    int time_multi(const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query, std::string &err);
    // enqueue_list through the multi-query path when the engine has one (desc.multi_q), else the ordinary sequence
    int enqueue_multi_list(const float *const *dev_xs, uint32_t *const *dev_idx, float *const *dev_val, int32_t count,
                           void *stream, std::string &err);
    // A back-to-back sequence given as lists of device pointers (count entries each); complete in stream order.
    int enqueue_list(const float *const *dev_xs, uint32_t *const *dev_idx, float *const *dev_val, int32_t count,
                     void *stream, std::string &err);
    // Deferred selection (back-to-back sequences on ONE stream): the top-k of a query is selected inside the next
    // enqueue_deferred launch, or by drain(). Results are complete in stream order only after drain().
    int enqueue_deferred(const float *dev_x, uint32_t *dev_idx, float *dev_val, void *stream, std::string &err);
    int drain(void *stream, std::string &err);
    int synchronize(std::string &err);
    int read(uint32_t *idx, float *val, int32_t *n, std::string &err);
    int result_device(const uint32_t **dev_idx, const float **dev_val);
    int scores(float *host_y, std::string &err);
    int read_trace(unsigned long long *host, size_t max_words, size_t *words, std::string &err);
    int debug_counters(unsigned long long *out, int n, std::string &err);
    int time_queries(const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query, std::string &err);
    int time_host_loop(const float *host_xs, int32_t n_x, int32_t iters, double *loop_ns, double *kernel_ns, std::string &err);
    int time_query_batches(const float *dev_xs, int32_t n_x, int32_t iters, int32_t reps, double *ns_per_query, std::string &err);
    int time_stream_read(int32_t passes, double *ns_per_pass, std::string &err);
    int profile(const float *dev_xs, int32_t n_x, int32_t iters, tkspmv_timing *out, std::string &err);
    void info(tkspmv_info *out) const;

   private:
    Engine() = default;
    EngineImpl *impl_ = nullptr;
};

void fill_info(const PackedMatrix &pm, int k, tkspmv_info *out);
uint64_t algorithmic_bytes(uint64_t nnz, uint32_t rows, uint32_t cols, uint32_t value_bytes, int k);
int device_count();
int use_device(int device, std::string &err);  // hipSetDevice(device), or the current device for -1
// Entries per lane and packet: desc.nnz_per_lane, by default 4. (8, where those kernels exist -- fp32 values, at most
// 1024 columns -- is opt-in: 3-4 % faster on the BASELINE matrix in short probes, 10-15 % slower in bench.py's
// conditions (64 query vectors, 4 stream copies), 3x slower at 2-3M rows, where its longer partitions overflow the
// private candidate lists.)
inline uint32_t entries_per_lane_of(const tkspmv_desc &d) {
    return d.nnz_per_lane > 0 ? (uint32_t)d.nnz_per_lane : 4u;
}
// Bits per value of a descriptor: desc.fixed_width for TKSPMV_FIXED (0 => 32, the reference's default FIXED_WIDTH), 0 for
// every other precision (where a non-zero fixed_width is rejected by the packer).
inline uint32_t fixed_width_of(const tkspmv_desc &d) {
    if (d.precision != TKSPMV_FIXED) return (uint32_t)d.fixed_width;
    return d.fixed_width == 0 ? 32u : (uint32_t)d.fixed_width;
}
// Value type of the packet stream a descriptor asks for (at most 1024 columns: narrow fixed point travels bit-packed, FIXED20,
// and fp32 values with 12-bit column words, F32C12).
inline Precision stream_precision_of(const tkspmv_desc &d) {
    return stream_precision(d.precision, fixed_width_of(d), d.cols, entries_per_lane_of(d));
}
// Wave partitions tkspmv_create would cut the matrix into on desc.device (= streaming waves of its launch geometry).
int wave_partitions_for(const tkspmv_desc &desc, uint32_t *out, std::string &err);

}  // namespace tkspmv

// The opaque handle of the C ABI (include/tkspmv.h: `typedef struct tkspmv_engine tkspmv_t`). Defined once, here.
struct tkspmv_engine {
    tkspmv::Engine *e;
};
