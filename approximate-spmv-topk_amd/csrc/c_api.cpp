// c_api.cpp -- extern "C" surface declared in include/tkspmv.h (plain pointers and sizes only).
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>

#include "../../include/tkspmv.h"
#include "device_pack.hpp"
#include "engine.hpp"
#include "host_utils.hpp"
#include "options.hpp"
#include "wbscsr.hpp"
#include "wsell.hpp"

using namespace tkspmv;

struct tkspmv_packed {
    PackedMatrix pm;
    int k;
};

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

extern "C" {

const char *tkspmv_last_error(void) { return g_err.c_str(); }

int tkspmv_device_count(void) { return device_count(); }

int tkspmv_create(tkspmv_t **out, const tkspmv_desc *desc) {
    if (!out || !desc) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    *out = nullptr;
    std::string err;
    int status = TKSPMV_OK;
    Engine *e = nullptr;
    try {
        e = Engine::create(*desc, err, status);
    } catch (const std::bad_alloc &) {
        return fail(TKSPMV_ERR_NOMEM, "out of host memory while packing");
    }
    if (!e) return fail(status, err);
    *out = new tkspmv_engine{e};
    return TKSPMV_OK;
}

void tkspmv_destroy(tkspmv_t *h) {
    if (!h) return;
    delete h->e;
    delete h;
}

#define ENGINE_CALL(call)                                        \
    if (!h) return fail(TKSPMV_ERR_INVALID, "NULL engine");      \
    std::string err;                                             \
    int st = h->e->call;                                         \
    if (st != TKSPMV_OK) g_err = err;                            \
    return st;

int tkspmv_get_info(const tkspmv_t *h, tkspmv_info *info) {
    if (!h || !info) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    h->e->info(info);
    return TKSPMV_OK;
}
int tkspmv_set_query(tkspmv_t *h, const float *host_x, double *elapsed_ns) { ENGINE_CALL(set_query(host_x, elapsed_ns, err)) }
int tkspmv_set_query_device(tkspmv_t *h, const float *dev_x) { ENGINE_CALL(set_query_device(dev_x, err)) }
int tkspmv_run(tkspmv_t *h, double *kernel_ns) { ENGINE_CALL(run(kernel_ns, err)) }
int tkspmv_enqueue(tkspmv_t *h, const float *dev_x, uint32_t *dev_idx, float *dev_val, void *stream) {
    ENGINE_CALL(enqueue(dev_x, dev_idx, dev_val, stream, err))
}
int tkspmv_enqueue_many(tkspmv_t *h, const float *dev_xs, int32_t n_x, int32_t count, void *stream) {
    ENGINE_CALL(enqueue_many(dev_xs, n_x, count, stream, err))
}
int tkspmv_enqueue_batch(tkspmv_t *h, const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream) {
    ENGINE_CALL(enqueue_batch(dev_xs, count, dev_idx, dev_val, stream, err))
}
int tkspmv_synchronize(tkspmv_t *h) { ENGINE_CALL(synchronize(err)) }
int tkspmv_read(tkspmv_t *h, uint32_t *idx, float *val, int32_t *n) { ENGINE_CALL(read(idx, val, n, err)) }
int tkspmv_result_device(tkspmv_t *h, const uint32_t **dev_idx, const float **dev_val) {
    if (!h) return fail(TKSPMV_ERR_INVALID, "NULL engine");
    return h->e->result_device(dev_idx, dev_val);
}
int tkspmv_scores(tkspmv_t *h, float *host_y) { ENGINE_CALL(scores(host_y, err)) }
int tkspmv_debug_trace(tkspmv_t *h, uint64_t *host, uint64_t max_words, uint64_t *words) {
    size_t w = 0;
    if (!h) return fail(TKSPMV_ERR_INVALID, "NULL engine");
    std::string err;
    int st = h->e->read_trace(reinterpret_cast<unsigned long long *>(host), (size_t)max_words, &w, err);
    if (st != TKSPMV_OK) g_err = err;
    if (words) *words = w;
    return st;
}
int tkspmv_debug_counters(tkspmv_t *h, uint64_t *out, int32_t n) { ENGINE_CALL(debug_counters(reinterpret_cast<unsigned long long *>(out), n, err)) }
int tkspmv_enqueue_multi(tkspmv_t *h, const float *dev_xs, int32_t count, uint32_t *dev_idx, float *dev_val, void *stream) {
    ENGINE_CALL(enqueue_multi(dev_xs, count, dev_idx, dev_val, stream, err))
}
int tkspmv_time_multi(tkspmv_t *h, const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query) {
    ENGINE_CALL(time_multi(dev_xs, n_x, iters, ns_per_query, err))
}
int tkspmv_set_option(const char *name, const char *value) {
    if (tkspmv::set_option(name, value) != 0) return fail(TKSPMV_ERR_INVALID, std::string("no such option: ") + (name ? name : "(null)"));
    return TKSPMV_OK;
}
const char *tkspmv_get_option(const char *name) {
    for (int i = 0; i < tkspmv::option_count(); ++i)
        if (name && std::strcmp(tkspmv::option_def(i)->name, name) == 0) return tkspmv::opt(name);
    return nullptr;
}
int tkspmv_option_count(void) { return tkspmv::option_count(); }
int tkspmv_option_info(int32_t i, const char **name, const char **kind, const char **values, const char **doc) {
    const tkspmv::OptionDef *o = tkspmv::option_def(i);
    if (!o) return fail(TKSPMV_ERR_INVALID, "option index out of range");
    if (name) *name = o->name;
    if (kind) *kind = o->kind;
    if (values) *values = o->values;
    if (doc) *doc = o->doc;
    return TKSPMV_OK;
}
int tkspmv_time_query_batches(tkspmv_t *h, const float *dev_xs, int32_t n_x, int32_t iters, int32_t reps, double *ns_per_query) {
    ENGINE_CALL(time_query_batches(dev_xs, n_x, iters, reps, ns_per_query, err))
}
int tkspmv_time_host_loop(tkspmv_t *h, const float *host_xs, int32_t n_x, int32_t iters, double *loop_ns, double *kernel_ns) {
    ENGINE_CALL(time_host_loop(host_xs, n_x, iters, loop_ns, kernel_ns, err))
}
int tkspmv_time_queries(tkspmv_t *h, const float *dev_xs, int32_t n_x, int32_t iters, double *ns_per_query) {
    ENGINE_CALL(time_queries(dev_xs, n_x, iters, ns_per_query, err))
}
int tkspmv_time_stream_read(tkspmv_t *h, int32_t passes, double *ns_per_pass) {
    ENGINE_CALL(time_stream_read(passes, ns_per_pass, err))
}
int tkspmv_profile(tkspmv_t *h, const float *dev_xs, int32_t n_x, int32_t iters, tkspmv_timing *out) {
    ENGINE_CALL(profile(dev_xs, n_x, iters, out, err))
}

// ---- host helpers ------------------------------------------------------------------------------------
static int coo_to_c(CooMatrix &m, tkspmv_coo *out) {
    std::memset(out, 0, sizeof(*out));
    out->rows = m.rows;
    out->cols = m.cols;
    out->nnz = m.nnz();
    out->num_rows_coo = m.num_rows_coo;
    out->index_base = m.index_base;
    out->symmetric = m.symmetric ? 1 : 0;
    size_t n = m.row.size();
    out->row = (uint32_t *)malloc(std::max<size_t>(n, 1) * 4);
    out->col = (uint32_t *)malloc(std::max<size_t>(n, 1) * 4);
    out->val = (float *)malloc(std::max<size_t>(n, 1) * 4);
    if (!out->row || !out->col || !out->val) {
        free(out->row);
        free(out->col);
        free(out->val);
        std::memset(out, 0, sizeof(*out));
        return fail(TKSPMV_ERR_NOMEM, "out of memory");
    }
    if (n) {
        std::memcpy(out->row, m.row.data(), n * 4);
        std::memcpy(out->col, m.col.data(), n * 4);
        std::memcpy(out->val, m.val.data(), n * 4);
    }
    return TKSPMV_OK;
}

int tkspmv_mtx_read(const char *path, int32_t index_base, int32_t read_values, int32_t sort, tkspmv_coo *out) {
    if (!path || !out) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    CooMatrix m;
    IoError e = read_mtx(path, index_base, read_values != 0, sort != 0, m);
    if (e.code) return fail(e.code, e.message);
    return coo_to_c(m, out);
}

void tkspmv_mtx_free(tkspmv_coo *m) {
    if (!m) return;
    free(m->row);
    free(m->col);
    free(m->val);
    std::memset(m, 0, sizeof(*m));
}

int tkspmv_mtx_write(const char *path, uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row,
                     const uint32_t *col, const float *val, int32_t index_base, int32_t precision) {
    if (!path || (nnz && (!row || !col))) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    IoError e = write_mtx(path, rows, cols, nnz, row, col, val, index_base, precision);
    if (e.code) return fail(e.code, e.message);
    return TKSPMV_OK;
}

int tkspmv_sample_vector(float *vec, int32_t size, int32_t random, int32_t sum_to_one, int32_t norm_one,
                         int32_t seed) {
    if (!vec || size < 0) return fail(TKSPMV_ERR_INVALID, "bad arguments");
    sample_vector(vec, size, random != 0, sum_to_one != 0, norm_one != 0, seed);
    return TKSPMV_OK;
}

int tkspmv_generate(uint32_t rows, uint32_t cols, uint32_t avg_nnz, int32_t dist, uint64_t seed, tkspmv_coo *out) {
    if (!out || cols == 0 || avg_nnz == 0) return fail(TKSPMV_ERR_INVALID, "bad arguments");
    if (dist != DIST_UNIFORM && dist != DIST_GAMMA) return fail(TKSPMV_ERR_INVALID, "dist must be 0 (uniform) or 1 (gamma)");
    CooMatrix m;
    try {
        generate_matrix(rows, cols, avg_nnz, dist, seed, m);
    } catch (const std::bad_alloc &) {
        return fail(TKSPMV_ERR_NOMEM, "out of memory");
    }
    return coo_to_c(m, out);
}

int tkspmv_generate_rows(uint32_t row_begin, uint32_t row_end, uint32_t cols, uint32_t avg_nnz, int32_t dist, uint64_t seed,
                         tkspmv_coo *out) {
    if (!out || cols == 0 || avg_nnz == 0 || row_end < row_begin) return fail(TKSPMV_ERR_INVALID, "bad arguments");
    if (dist != DIST_UNIFORM && dist != DIST_GAMMA) return fail(TKSPMV_ERR_INVALID, "dist must be 0 (uniform) or 1 (gamma)");
    CooMatrix m;
    try {
        generate_matrix_rows(row_begin, row_end, cols, avg_nnz, dist, seed, m);
    } catch (const std::bad_alloc &) {
        return fail(TKSPMV_ERR_NOMEM, "out of memory");
    }
    return coo_to_c(m, out);
}

int tkspmv_generate_degrees(uint32_t row_begin, uint32_t row_end, uint32_t avg_nnz, int32_t dist, uint64_t seed,
                            uint32_t *deg) {
    if (!deg || avg_nnz == 0 || row_end < row_begin) return fail(TKSPMV_ERR_INVALID, "bad arguments");
    if (dist != DIST_UNIFORM && dist != DIST_GAMMA) return fail(TKSPMV_ERR_INVALID, "dist must be 0 (uniform) or 1 (gamma)");
    generate_degrees(row_begin, row_end, avg_nnz, dist, seed, deg);
    return TKSPMV_OK;
}

int tkspmv_options_parse(int argc, char **argv, tkspmv_options *out) {
    if (!out || (argc > 0 && !argv)) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    Options o(argc, argv);
    std::memset(out, 0, sizeof(*out));
    std::strncpy(out->matrix_path, o.matrix_path.c_str(), sizeof(out->matrix_path) - 1);
    std::strncpy(out->xclbin_path, o.xclbin_path.c_str(), sizeof(out->xclbin_path) - 1);
    out->use_sample_matrix = o.use_sample_matrix;
    out->reset = o.reset;
    out->num_tests = (int32_t)o.num_tests;
    out->debug = o.debug;
    out->ignore_matrix_values = o.ignore_matrix_values;
    out->top_k_value = o.top_k_value;
    out->gpu_impl = o.gpu_impl;
    out->use_half_precision_gpu = o.use_half_precision_gpu;
    out->block_size_1d = o.block_size_1d;
    out->block_size_2d = o.block_size_2d;
    out->num_blocks = o.num_blocks;
    return TKSPMV_OK;
}

int tkspmv_pack(const tkspmv_desc *d, uint32_t n_wave_partitions_hint, tkspmv_packed **out) {
    if (!d || !out) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    *out = nullptr;
    tkspmv_packed *p = new tkspmv_packed();
    int kind = 0;
    uint32_t C = entries_per_lane_of(*d);
    const Precision sp = stream_precision_of(*d);
    std::string err = pack_wbscsr(d->rows, d->cols, d->nnz, d->row, d->col, d->val, sp, C,
                                  n_wave_partitions_hint ? n_wave_partitions_hint : 4096u, min_packets_per_partition_for(d->nnz, C, d->cols), p->pm, kind,
                                  fixed_width_of(*d));
    if (!err.empty()) {
        delete p;
        return fail(kind == 2 ? TKSPMV_ERR_NOT_SORTED : TKSPMV_ERR_INVALID, err);
    }
    p->k = d->k;
    *out = p;
    return TKSPMV_OK;
}

int tkspmv_pack_device(const tkspmv_desc *d, uint32_t n_wave_partitions_hint, tkspmv_packed **out, double *ms) {
    if (!d || !out) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (device_count() < 1) return fail(TKSPMV_ERR_DEVICE, "no HIP device available (the device packer needs one)");
    std::string serr;
    if (use_device(d->device, serr) != TKSPMV_OK) return fail(TKSPMV_ERR_DEVICE, serr);
    DevicePacked dp;
    int kind = 0;
    std::string err = pack_wbscsr_device(d->rows, d->cols, d->nnz, d->row, d->col, d->val, stream_precision_of(*d),
                                         entries_per_lane_of(*d), n_wave_partitions_hint ? n_wave_partitions_hint : 4096u,
                                         min_packets_per_partition_for(d->nnz, entries_per_lane_of(*d), d->cols),
                                         fixed_width_of(*d), dp, kind);
    if (err.empty()) err = download_device_packed(dp);
    free_device_packed(dp);
    if (!err.empty()) return fail(kind == 2 ? TKSPMV_ERR_NOT_SORTED : (err.find("failed:") != std::string::npos ? TKSPMV_ERR_DEVICE : TKSPMV_ERR_INVALID), err);
    tkspmv_packed *p = new tkspmv_packed();
    p->pm = std::move(dp.meta);
    p->k = d->k;
    if (ms) {
        ms[0] = dp.upload_ms;
        ms[1] = dp.kernels_ms;
    }
    *out = p;
    return TKSPMV_OK;
}

int tkspmv_packed_info(const tkspmv_packed *p, tkspmv_info *info) {
    if (!p || !info) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    fill_info(p->pm, p->k, info);
    return TKSPMV_OK;
}

int tkspmv_packed_decode(const tkspmv_packed *p, uint32_t *row, uint32_t *col, float *val, uint64_t *n) {
    if (!p || !n) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    std::vector<uint32_t> r, c;
    std::vector<float> v;
    decode_wbscsr(p->pm, r, c, v);
    *n = r.size();
    if (row) std::memcpy(row, r.data(), r.size() * 4);
    if (col) std::memcpy(col, c.data(), c.size() * 4);
    if (val) std::memcpy(val, v.data(), v.size() * 4);
    return TKSPMV_OK;
}

int tkspmv_packed_raw(const tkspmv_packed *p, const void **packets, uint64_t *packet_bytes, const uint32_t **pkt_row,
                      const uint32_t **part_first, const uint32_t **part_count, uint32_t *n_parts) {
    if (!p) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    if (packets) *packets = p->pm.packets.data();
    if (packet_bytes) *packet_bytes = p->pm.packet_bytes;
    if (pkt_row) *pkt_row = p->pm.pkt_row.data();
    if (part_first) *part_first = p->pm.part_first.data();
    if (part_count) *part_count = p->pm.part_count.data();
    if (n_parts) *n_parts = (uint32_t)p->pm.part_first.size();
    return TKSPMV_OK;
}

void tkspmv_packed_free(tkspmv_packed *p) { delete p; }

int tkspmv_sell_roundtrip(const tkspmv_desc *d, uint32_t n_wave_partitions_hint, uint32_t *row, uint32_t *col, float *val,
                          uint64_t *n, uint64_t *info) {
    if (!d || !n) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    SellMatrix sm;
    const std::string err = pack_wsell(d->rows, d->cols, d->nnz, d->row, d->col, d->val,
                                       n_wave_partitions_hint ? n_wave_partitions_hint : 4088u, sm,
                                       d->precision == TKSPMV_Q1_7_F32 ? SellValues::Q1_7_RND : SellValues::F32);
    if (!err.empty()) return fail(TKSPMV_ERR_INVALID, err);
    std::vector<uint32_t> r, c;
    std::vector<float> v;
    decode_wsell(sm, r, c, v);
    *n = r.size();
    if (row) std::memcpy(row, r.data(), r.size() * 4);
    if (col) std::memcpy(col, c.data(), c.size() * 4);
    if (val) std::memcpy(val, v.data(), v.size() * 4);
    if (info) {
        uint32_t most = 0;
        for (uint32_t pc : sm.part_count) most = std::max(most, pc);
        info[0] = sm.n_slices;
        info[1] = sm.n_chunks;
        info[2] = sm.padded_entries;
        info[3] = sm.part_first.size();
        info[4] = sm.stream_bytes();
        info[5] = most;
    }
    return TKSPMV_OK;
}

int tkspmv_sell_pack_device_check(const tkspmv_desc *d, uint32_t n_wave_partitions_hint, uint64_t *info, double *ms) {
    if (!d || !info) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    if (device_count() < 1) return fail(TKSPMV_ERR_DEVICE, "no HIP device available (the device packer needs one)");
    std::string serr;
    if (use_device(d->device, serr) != TKSPMV_OK) return fail(TKSPMV_ERR_DEVICE, serr);
    const uint32_t hint = n_wave_partitions_hint ? n_wave_partitions_hint : 4088u;
    const SellValues sv = d->precision == TKSPMV_Q1_7_F32 ? SellValues::Q1_7_RND : SellValues::F32;
    SellMatrix host;
    const auto t0 = std::chrono::steady_clock::now();
    const std::string herr = pack_wsell(d->rows, d->cols, d->nnz, d->row, d->col, d->val, hint, host, sv);
    const auto t1 = std::chrono::steady_clock::now();
    DeviceSell ds;
    std::string derr = pack_wsell_device(d->rows, d->cols, d->nnz, d->row, d->col, d->val, hint, sv, nullptr, nullptr, ds);
    if (herr != derr) {
        free_device_sell(ds);
        return fail(TKSPMV_ERR_STATE, "the two packers disagree on the input: host '" + herr + "', device '" + derr + "'");
    }
    if (!herr.empty()) return fail(TKSPMV_ERR_INVALID, herr);
    derr = download_device_sell(ds);
    free_device_sell(ds);
    if (!derr.empty()) return fail(TKSPMV_ERR_DEVICE, derr);
    const SellMatrix &dev = ds.meta;
    info[0] = (host.n_slices == dev.n_slices && host.n_chunks == dev.n_chunks && host.packet_bytes == dev.packet_bytes &&
               host.padded_entries == dev.padded_entries && host.packets == dev.packets && host.slice_rows == dev.slice_rows &&
               host.part_first == dev.part_first && host.part_count == dev.part_count && host.part_slice0 == dev.part_slice0)
                  ? 1u : 0u;
    info[1] = host.stream_bytes();
    info[2] = host.n_chunks;
    if (ms) {
        ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        ms[1] = ds.plan_ms;
        ms[2] = ds.upload_ms;
        ms[3] = ds.kernels_ms;
    }
    return TKSPMV_OK;
}

int tkspmv_packed_save(const tkspmv_packed *p, const char *path) {
    if (!p || !path) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    const std::string err = save_packed(p->pm, path);
    if (!err.empty()) return fail(TKSPMV_ERR_IO, err);
    return TKSPMV_OK;
}

int tkspmv_packed_load(const char *path, tkspmv_packed **out) {
    if (!path || !out) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    *out = nullptr;
    tkspmv_packed *p = new tkspmv_packed();
    const std::string err = load_packed(path, p->pm);
    if (!err.empty()) {
        delete p;
        return fail(TKSPMV_ERR_IO, err);
    }
    p->k = 0;
    *out = p;
    return TKSPMV_OK;
}

int tkspmv_wave_partitions(const tkspmv_desc *desc, uint32_t *n) {
    if (!desc || !n) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    std::string err;
    int st = wave_partitions_for(*desc, n, err);
    if (st != TKSPMV_OK) g_err = err;
    return st;
}

int tkspmv_create_packed(tkspmv_t **out, const tkspmv_packed *p, const tkspmv_desc *desc) {
    if (!out || !p || !desc) return fail(TKSPMV_ERR_INVALID, "NULL argument");
    *out = nullptr;
    tkspmv_desc d = *desc;
    d.rows = p->pm.rows;
    d.cols = p->pm.cols;
    d.nnz = p->pm.nnz;
    // desc.precision chooses among the arithmetic modes of the packed value type (only Q1.7 has two)
    const bool fixed_either = desc->precision == TKSPMV_FIXED && (p->pm.precision == Precision::FIXED || p->pm.precision == Precision::FIXED20 || p->pm.precision == Precision::FIXED26);
    const bool f32_either = desc->precision == TKSPMV_F32 && (p->pm.precision == Precision::F32 || p->pm.precision == Precision::F32C12);
    if (!fixed_either && !f32_either && stream_precision(desc->precision) != p->pm.precision)
        d.precision = (p->pm.precision == Precision::F32 || p->pm.precision == Precision::F32C12)
                          ? TKSPMV_F32
                          : (p->pm.precision == Precision::F16
                                 ? TKSPMV_F16
                                 : ((p->pm.precision == Precision::FIXED || p->pm.precision == Precision::FIXED20 || p->pm.precision == Precision::FIXED26)
                                        ? TKSPMV_FIXED
                                        : (p->pm.precision == Precision::Q1_7_RND ? TKSPMV_Q1_7_F32 : TKSPMV_Q1_7)));
    d.fixed_width = (int32_t)p->pm.fixed_width;  // a property of the packed values
    d.nnz_per_lane = (int32_t)p->pm.C;
    std::string err;
    int status = TKSPMV_OK;
    Engine *e = nullptr;
    try {
        e = Engine::create(d, err, status, &p->pm);
    } catch (const std::bad_alloc &) {
        return fail(TKSPMV_ERR_NOMEM, "out of host memory");
    }
    if (!e) return fail(status, err);
    *out = new tkspmv_engine{e};
    return TKSPMV_OK;
}

}  // extern "C"
