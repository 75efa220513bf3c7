// device_pack.hip -- the wave-BSCSR packer on the GPU.
//
// The reference parses and packs on the host on every run (src/common/utils/utils.hpp:380-388,
// src/fpga/src/host_spmv_bscsr.cpp:133-248: `hw_setup_time_ms`); so does wbscsr.cpp, single-threaded. Here the row-sorted
// COO is uploaded once and packed where the stream is going to live:
//
//   count_kernel     row lengths (one atomic per entry), validation of ordering and ranges
//   scan_*           exclusive prefix sums of the row lengths and of the placeholder-expanded lengths (empty row -> 1 entry)
//   cuts_kernel      the greedy partition cuts of fill_partitions() -- a chain of P dependent searches, done by ONE wave with
//                    64-ary searches over the prefix sums (4 probes per partition instead of 20 binary steps), including the
//                    loop that grows the partition capacity until the cuts fit the wave count
//   scatter_kernel   every COO entry finds its packet slot (row start + offset -> partition by binary search in LDS) and
//                    writes its value (converted by the SAME functions as the host packer, wbscsr.hpp) and column word
//   holes_kernel     placeholder entries of empty rows
//
// Output bytes are identical to pack_wbscsr()'s (tests/test_gpu_device_pack.py compares streams and side tables).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "device_pack.hpp"
#include "wsell.hpp"

namespace tkspmv {

namespace {

constexpr uint32_t SCAN_BLOCK = 256, SCAN_ITEMS = 4, SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

// flags[0]: 1 = rows not sorted, 2 = row id out of range, 4 = column id out of range
// len[] has n_rows = row[nnz - 1] + 1 entries: a row id beyond it can only occur in an unsorted COO (it exceeds the last
// entry's row) and must not be counted -- the atomic would land outside the allocation.
__global__ void __launch_bounds__(256) count_kernel(const uint32_t *__restrict__ row, const uint32_t *__restrict__ col, uint64_t nnz,
                                                    uint32_t rows, uint32_t n_rows, uint32_t cols, uint32_t *__restrict__ len,
                                                    uint32_t *__restrict__ flags) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t bad = 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += stride) {
        const uint32_t r = row[i];
        if (i > 0 && r < row[i - 1]) bad |= 1u;
        if (r >= rows) {
            bad |= 2u;
            continue;
        }
        if (col[i] >= cols) bad |= 4u;
        if (r >= n_rows) {
            bad |= 1u;
            continue;
        }
        atomicAdd(&len[r], 1u);
    }
    if (bad) atomicOr(flags, bad);
}

struct Pair {
    unsigned long long a, b;  // a: entries of the COO, b: entries of the placeholder-expanded stream
};
__device__ __forceinline__ Pair pair_of(uint32_t L) { return Pair{L, L ? L : 1u}; }
__device__ __forceinline__ Pair operator+(Pair x, Pair y) { return Pair{x.a + y.a, x.b + y.b}; }

__device__ __forceinline__ Pair block_reduce(Pair v, Pair *sh) {
    const uint32_t tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (uint32_t s = SCAN_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = sh[tid] + sh[tid + s];
        __syncthreads();
    }
    const Pair r = sh[0];
    __syncthreads();
    return r;
}

// tile sums of the two length sequences
__global__ void __launch_bounds__(SCAN_BLOCK) scan_tile_sums(const uint32_t *__restrict__ len, uint32_t n, Pair *__restrict__ tile_sum) {
    __shared__ Pair sh[SCAN_BLOCK];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    Pair v{0, 0};
    for (uint32_t u = 0; u < SCAN_ITEMS; ++u)
        if (base + u < n) v = v + pair_of(len[base + u]);
    const Pair t = block_reduce(v, sh);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = t;
}
// exclusive scan of the tile sums by one block (sequential over chunks of SCAN_BLOCK); tile_sum[n_tiles] = grand total
__global__ void __launch_bounds__(SCAN_BLOCK) scan_tiles(Pair *__restrict__ tile_sum, uint32_t n_tiles) {
    __shared__ Pair sh[SCAN_BLOCK];
    Pair carry{0, 0};
    for (uint32_t c0 = 0; c0 < n_tiles; c0 += SCAN_BLOCK) {
        const uint32_t i = c0 + threadIdx.x;
        const Pair mine = i < n_tiles ? tile_sum[i] : Pair{0, 0};
        sh[threadIdx.x] = mine;
        __syncthreads();
        for (uint32_t d = 1; d < SCAN_BLOCK; d <<= 1) {  // Hillis-Steele inclusive scan
            Pair t{0, 0};
            if (threadIdx.x >= d) t = sh[threadIdx.x - d];
            __syncthreads();
            sh[threadIdx.x] = sh[threadIdx.x] + t;
            __syncthreads();
        }
        const Pair incl = sh[threadIdx.x];
        const Pair total = sh[SCAN_BLOCK - 1];
        if (i < n_tiles) tile_sum[i] = Pair{carry.a + incl.a - mine.a, carry.b + incl.b - mine.b};
        carry = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[n_tiles] = carry;
}
// coo_start[r] = entries before row r in the COO; exp_start[r] = entries before row r in the expanded stream; both [n + 1]
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply(const uint32_t *__restrict__ len, uint32_t n, const Pair *__restrict__ tile_sum,
                                                         unsigned long long *__restrict__ coo_start,
                                                         unsigned long long *__restrict__ exp_start) {
    __shared__ Pair sh[SCAN_BLOCK];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    Pair item[SCAN_ITEMS], v{0, 0};
    for (uint32_t u = 0; u < SCAN_ITEMS; ++u) {
        item[u] = base + u < n ? pair_of(len[base + u]) : Pair{0, 0};
        v = v + item[u];
    }
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_BLOCK; d <<= 1) {
        Pair t{0, 0};
        if (threadIdx.x >= d) t = sh[threadIdx.x - d];
        __syncthreads();
        sh[threadIdx.x] = sh[threadIdx.x] + t;
        __syncthreads();
    }
    Pair run = tile_sum[blockIdx.x] + Pair{sh[threadIdx.x].a - v.a, sh[threadIdx.x].b - v.b};
    for (uint32_t u = 0; u < SCAN_ITEMS; ++u) {
        if (base + u < n) {
            coo_start[base + u] = run.a;
            exp_start[base + u] = run.b;
        }
        run = run + item[u];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) {
        coo_start[n] = tile_sum[gridDim.x].a;
        exp_start[n] = tile_sum[gridDim.x].b;
    }
}

// Smallest index f in (lo, hi] with S[f] > target, given S[lo] <= target < S[hi] (S non-decreasing), by 64-ary search: the
// lanes of one wave probe evenly spaced indices strictly inside the range, a ballot narrows it to one step.
__device__ __forceinline__ uint32_t first_above(const unsigned long long *__restrict__ S, uint32_t lo, uint32_t hi,
                                                unsigned long long target, uint32_t lane) {
    while (hi - lo > 1u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + (lane + 1u) * step;
        const bool le = idx < hi && S[idx] <= target;
        const uint32_t n_le = (uint32_t)__popcll(__ballot(le));  // S is monotone: the lanes that pass form a prefix
        lo += n_le * step;                                       // S[lo] <= target still holds
        hi = (lo + step < hi) ? lo + step : hi;                  // the first probe that failed (or the old bound)
    }
    return hi;
}

struct CutsOut {
    uint32_t n_parts, m, n_packets_lo, n_packets_hi, overflow;
};

// The greedy cuts of fill_partitions(): a partition takes rows while their expanded entries fit `cap`; at least one row.
// part tables are written for up to P_max partitions; returns the number of partitions the capacity leads to.
// bal_B != 0: balanced cuts (wbscsr.cpp, fill_partitions_balanced) -- partition p may take PE x (floor((p + 1) B / P) - floor(p B / P))
// entries instead of `cap`.
__device__ uint32_t cut_pass(const unsigned long long *__restrict__ S, uint32_t n_rows, unsigned long long cap, uint32_t P_max,
                             uint32_t *__restrict__ part_row0, uint32_t lane, unsigned long long bal_B = 0ull, unsigned long long PE = 0ull) {
    uint32_t a = 0, parts = 0;
    while (a < n_rows) {
        if (parts < P_max && lane == 0) part_row0[parts] = a;
        if (bal_B != 0ull) {
            const unsigned long long p = parts, P = P_max;
            cap = PE * (p < P ? ((p + 1ull) * bal_B) / P - (p * bal_B) / P : (bal_B + P - 1ull) / P);
        }
        ++parts;
        if (parts > P_max) return parts;  // too many: the caller grows the capacity
        const unsigned long long target = S[a] + cap;
        // rows a .. b-1 with S[b] - S[a] <= cap, b >= a + 1: b = (first index in (a, n] with S[idx] > target) - 1, or n
        uint32_t b;
        if (S[n_rows] <= target) {
            b = n_rows;
        } else {
            const uint32_t f = first_above(S, a, n_rows, target, lane);  // S[f] > target, S[f - 1] <= target
            b = f - 1u;
            if (b <= a) b = a + 1u;  // a row longer than the capacity gets a partition of its own
        }
        a = b;
    }
    return parts;
}

__global__ void __launch_bounds__(64) cuts_kernel(const unsigned long long *__restrict__ S, uint32_t n_rows, uint32_t PE,
                                                  uint32_t P_hint, uint32_t min_packets, uint32_t balanced, uint32_t *__restrict__ part_row0,
                                                  uint32_t *__restrict__ part_rows, uint32_t *__restrict__ part_first,
                                                  uint32_t *__restrict__ part_count, CutsOut *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const unsigned long long E = S[n_rows];
    const unsigned long long total_packets_lb = (E + PE - 1) / PE;
    unsigned long long max_parts = total_packets_lb / min_packets;
    if (max_parts < 1) max_parts = 1;
    const uint32_t P = (uint32_t)(P_hint < max_parts ? P_hint : max_parts);
    unsigned long long m = (E + (unsigned long long)P * PE - 1) / ((unsigned long long)P * PE);
    if (m < 1) m = 1;
    uint32_t used;
    for (;;) {
        used = cut_pass(S, n_rows, m * PE, P, part_row0, lane);
        if (used <= P) break;
        ++m;  // padding pushed the cuts over the wave count: allow one more packet per partition
    }
    // balanced cuts where the uniform ones miss P by more than 1/8 (wbscsr.cpp: the same rule, the same arithmetic)
    if (balanced != 0u && P >= 2u && total_packets_lb >= 2ull * P &&
        (balanced == 2u ? (unsigned long long)used * 32ull < (unsigned long long)P * 31ull : (unsigned long long)used * 8ull < (unsigned long long)P * 7ull)) {
        unsigned long long B = total_packets_lb > P ? total_packets_lb : P;
        for (;;) {
            used = cut_pass(S, n_rows, 0ull, P, part_row0, lane, B, PE);
            if (used <= P) break;
            B += (B / 64ull) > 1ull ? (B / 64ull) : 1ull;
        }
        m = (B + P - 1ull) / P;
    }
    __threadfence();  // lane 0's table writes are read by the other lanes below
    __syncthreads();
    unsigned long long n_packets = 0;
    for (uint32_t p0 = 0; p0 < used; p0 += 64u) {  // packet counts: an exclusive scan over the partitions, 64 at a time
        const uint32_t p = p0 + lane;
        unsigned long long pk = 0;
        uint32_t r0 = 0, r1 = 0;
        if (p < used) {
            r0 = part_row0[p];
            r1 = p + 1 < used ? part_row0[p + 1] : n_rows;
            pk = (S[r1] - S[r0] + PE - 1) / PE;
        }
        unsigned long long incl = pk;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(incl, d);
            if ((int)lane >= d) incl += t;
        }
        if (p < used) {
            part_rows[p] = r1 - r0;
            part_first[p] = (uint32_t)(n_packets + incl - pk);
            part_count[p] = (uint32_t)pk;
        }
        n_packets += __shfl(incl, 63);
    }
    if (lane == 0) {
        out->n_parts = used;
        out->m = (uint32_t)m;
        out->n_packets_lo = (uint32_t)n_packets;
        out->n_packets_hi = (uint32_t)(n_packets >> 32);
        out->overflow = n_packets > 0xFFFFFFFFull ? 1u : 0u;
    }
}

struct ScatterParams {
    const uint32_t *row, *col;
    const float *val;  // NULL: all ones
    uint64_t nnz;
    const uint32_t *len;
    const unsigned long long *coo_start, *exp_start;
    const uint32_t *part_row0, *part_first;
    uint32_t n_parts, n_rows;
    uint8_t *packets;
    uint32_t *pkt_row;
    uint32_t C, PE, packet_bytes, vb, precision, fixed_width;
};

constexpr uint32_t PART_LDS = 8192;  // partitions whose first rows are searched in LDS (beyond: in global memory)

__device__ __forceinline__ uint32_t partition_of(uint32_t r, const uint32_t *pr0, uint32_t n_parts) {
    uint32_t lo = 0, hi = n_parts;  // last p with pr0[p] <= r
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pr0[mid] <= r) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void place(const ScatterParams &P, uint32_t r, unsigned long long e, const uint32_t *pr0, uint16_t cw, float v) {
    const uint32_t p = partition_of(r, pr0, P.n_parts);
    const unsigned long long e_local = e - P.exp_start[pr0[p]];
    const uint32_t pk = P.part_first[p] + (uint32_t)(e_local / P.PE);
    const uint32_t ss = (uint32_t)(e_local % P.PE);
    const uint32_t slot = slot_to_index(ss, P.C);
    uint8_t *pkt = P.packets + (size_t)pk * P.packet_bytes;
    // the first row that ends in a packet is the row of its first entry (wbscsr.cpp)
    if (ss == 0u) P.pkt_row[pk] = r;
    if ((Precision)P.precision == Precision::FIXED20) {
        *reinterpret_cast<uint32_t *>(pkt + (size_t)slot * 4) = fixed20_word(to_fixed(v, P.fixed_width), (uint32_t)(cw >> COLW_COL_SHIFT), cw & 3u);
        return;
    }
    if ((Precision)P.precision == Precision::FIXED26) {  // (zeroed stream: the lane's E collects 6 bits from each of its 4 entries)
        const uint32_t colv = (uint32_t)(cw >> COLW_COL_SHIFT);
        *reinterpret_cast<uint32_t *>(pkt + (size_t)slot * 4) = fixed26_d(to_fixed(v, P.fixed_width), colv, cw & 3u);
        const uint32_t e = fixed26_e(slot & 3u, colv);
        if (e) atomicOr(reinterpret_cast<uint32_t *>(pkt + (size_t)P.PE * 4) + (slot >> 2), e);
        return;
    }
    if ((Precision)P.precision == Precision::F32C12) {
        *reinterpret_cast<float *>(pkt + (size_t)slot * 4) = v;
        // split 12-bit plane (wbscsr.hpp colw12s_*): the entry's bits are OR-ed into the lane's dword A and halfword B of the
        // zeroed stream (four entries share them; B through the dword it shares with the pair's other lane)
        const uint32_t t = slot & 255u, lane = t >> 2;
        uint32_t *plane = reinterpret_cast<uint32_t *>(pkt + (size_t)P.PE * 4 + (size_t)(slot >> 8) * 384u);
        uint32_t a, b;
        colw12s_bits(t & 3u, cw, a, b);
        if (a) atomicOr(&plane[colw12s_a_offset(lane) >> 2], a);
        if (b) atomicOr(&plane[(lane >> 1) * 3u + 1u], b << ((lane & 1u) * 16u));
        return;
    }
    switch ((Precision)P.precision) {
        case Precision::F32: *reinterpret_cast<float *>(pkt + (size_t)slot * 4) = v; break;
        case Precision::F16: *reinterpret_cast<uint16_t *>(pkt + (size_t)slot * 2) = to_half(v); break;
        case Precision::FIXED: *reinterpret_cast<uint32_t *>(pkt + (size_t)slot * 4) = to_fixed(v, P.fixed_width); break;
        case Precision::Q1_7_RND: pkt[slot] = to_q1_7_rnd(v); break;
        default: pkt[slot] = to_q1_7(v); break;
    }
    *reinterpret_cast<uint16_t *>(pkt + (size_t)P.PE * P.vb + (size_t)slot * 2) = cw;
}

__global__ void __launch_bounds__(256) scatter_kernel(const ScatterParams P) {
    __shared__ uint32_t pr0_lds[PART_LDS];
    const bool in_lds = P.n_parts <= PART_LDS;
    if (in_lds)
        for (uint32_t i = threadIdx.x; i < P.n_parts; i += blockDim.x) pr0_lds[i] = P.part_row0[i];
    __syncthreads();
    const uint32_t *pr0 = in_lds ? pr0_lds : P.part_row0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.nnz; i += stride) {
        const uint32_t r = P.row[i];
        const uint32_t j = (uint32_t)(i - P.coo_start[r]);
        uint16_t cw = (uint16_t)(P.col[i] << COLW_COL_SHIFT);
        if (j + 1u == P.len[r]) cw |= COLW_ROW_END;
        place(P, r, P.exp_start[r] + j, pr0, cw, P.val ? P.val[i] : 1.0f);
    }
}

__global__ void __launch_bounds__(256) holes_kernel(const ScatterParams P) {
    __shared__ uint32_t pr0_lds[PART_LDS];
    const bool in_lds = P.n_parts <= PART_LDS;
    if (in_lds)
        for (uint32_t i = threadIdx.x; i < P.n_parts; i += blockDim.x) pr0_lds[i] = P.part_row0[i];
    __syncthreads();
    const uint32_t *pr0 = in_lds ? pr0_lds : P.part_row0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < P.n_rows; r += stride)
        if (P.len[r] == 0u) place(P, r, P.exp_start[r], pr0, (uint16_t)(COLW_SKIP | COLW_ROW_END), 0.0f);
}

struct DevBuf {  // frees what it owns unless released
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
    void *release() {
        void *q = p;
        p = nullptr;
        return q;
    }
};

}  // namespace

#define DP_TRY(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return std::string(#expr) + " failed: " + hipGetErrorString(_e); \
    } while (0)

std::string pack_wbscsr_device(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                               const float *val, Precision precision, uint32_t C, uint32_t n_partitions_hint,
                               uint32_t min_packets_per_partition, uint32_t fixed_width, DevicePacked &out, int &kind) {
    kind = 1;
    if (C != 4 && C != 8) return "nnz_per_lane must be 4 or 8";
    if (precision == Precision::FIXED26 ? (fixed_width < 8 || fixed_width > FIXED26_MAX_WIDTH || cols > FIXED26_MAX_COLS || C != 4)
        : precision == Precision::FIXED20 ? (fixed_width < 8 || fixed_width > FIXED20_MAX_WIDTH || cols > FIXED20_MAX_COLS)
                                        : (precision == Precision::FIXED ? (fixed_width < 8 || fixed_width > 32) : fixed_width != 0))
        return "fixed_width must be in [8, 32] for fixed-point values (bit-packed: at most 20 bits and 1024 columns) and 0 otherwise";
    if (cols == 0 || cols > MAX_COLS) return "cols must be in [1, 16384]";
    if (precision == Precision::F32C12 && (cols > F32C12_MAX_COLS || C != 4)) return "12-bit column words need at most 1024 columns and 4 entries per lane";
    if (nnz > 0 && (!row || !col)) return "row/col arrays are NULL";
    if (n_partitions_hint == 0) n_partitions_hint = 1;
    if (min_packets_per_partition == 0) min_packets_per_partition = 1;
    const auto t_begin = std::chrono::steady_clock::now();

    PackedMatrix &pm = out.meta;
    pm = PackedMatrix();
    pm.rows = rows;
    pm.cols = cols;
    pm.nnz = nnz;
    pm.precision = precision;
    pm.fixed_width = fixed_width;
    pm.C = C;
    pm.packet_entries = WAVE * C;
    pm.packet_bytes = packet_bytes_for(precision, pm.packet_entries);
    out.d_packets = nullptr;
    out.d_pkt_row = nullptr;
    if (nnz == 0) {  // empty matrix: no packets, no partitions
        kind = 0;
        return "";
    }
    const uint32_t last_row = row[nnz - 1];  // (validated on the device together with the ordering)
    if (last_row >= rows) return "row id out of range (>= rows)";
    const uint32_t n_rows = last_row + 1;  // rows [0, last_row] take part in the stream (empty ones as placeholders)

    // ---- upload the COO -------------------------------------------------------------------------------------------
    DevBuf d_row, d_col, d_val, d_len, d_flags, d_tiles, d_coo_start, d_exp_start, d_pr0, d_prows, d_pfirst, d_pcount, d_cuts;
    DP_TRY(hipMalloc(&d_row.p, nnz * 4));
    DP_TRY(hipMalloc(&d_col.p, nnz * 4));
    if (val) DP_TRY(hipMalloc(&d_val.p, nnz * 4));
    DP_TRY(hipMemcpy(d_row.p, row, nnz * 4, hipMemcpyHostToDevice));
    DP_TRY(hipMemcpy(d_col.p, col, nnz * 4, hipMemcpyHostToDevice));
    if (val) DP_TRY(hipMemcpy(d_val.p, val, nnz * 4, hipMemcpyHostToDevice));
    const auto t_uploaded = std::chrono::steady_clock::now();

    // ---- row lengths, validation ------------------------------------------------------------------------------------
    DP_TRY(hipMalloc(&d_len.p, (size_t)n_rows * 4));
    DP_TRY(hipMemset(d_len.p, 0, (size_t)n_rows * 4));
    DP_TRY(hipMalloc(&d_flags.p, 64));
    DP_TRY(hipMemset(d_flags.p, 0, 64));
    const uint32_t grid_nnz = (uint32_t)std::min<uint64_t>((nnz + 255) / 256, 8192);
    hipLaunchKernelGGL(count_kernel, dim3(grid_nnz), dim3(256), 0, 0, d_row.as<uint32_t>(), d_col.as<uint32_t>(), nnz, rows, n_rows,
                       cols, d_len.as<uint32_t>(), d_flags.as<uint32_t>());
    uint32_t flags = 0;
    DP_TRY(hipMemcpy(&flags, d_flags.p, 4, hipMemcpyDeviceToHost));
    if (flags & 1u) {
        kind = 2;
        return "COO rows are not sorted in non-decreasing order";
    }
    if (flags & 2u) return "row id out of range (>= rows)";
    if (flags & 4u) return "column id out of range (>= cols)";

    // ---- prefix sums ------------------------------------------------------------------------------------------------------
    const uint32_t n_tiles = (n_rows + SCAN_TILE - 1) / SCAN_TILE;
    DP_TRY(hipMalloc(&d_tiles.p, ((size_t)n_tiles + 1) * sizeof(Pair)));
    DP_TRY(hipMalloc(&d_coo_start.p, ((size_t)n_rows + 1) * 8));
    DP_TRY(hipMalloc(&d_exp_start.p, ((size_t)n_rows + 1) * 8));
    hipLaunchKernelGGL(scan_tile_sums, dim3(n_tiles), dim3(SCAN_BLOCK), 0, 0, d_len.as<uint32_t>(), n_rows, d_tiles.as<Pair>());
    hipLaunchKernelGGL(scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, 0, d_tiles.as<Pair>(), n_tiles);
    hipLaunchKernelGGL(scan_apply, dim3(n_tiles), dim3(SCAN_BLOCK), 0, 0, d_len.as<uint32_t>(), n_rows, d_tiles.as<Pair>(),
                       d_coo_start.as<unsigned long long>(), d_exp_start.as<unsigned long long>());
    unsigned long long E = 0;
    DP_TRY(hipMemcpy(&E, d_exp_start.as<unsigned long long>() + n_rows, 8, hipMemcpyDeviceToHost));
    pm.placeholders = E - nnz;

    // ---- partition cuts ---------------------------------------------------------------------------------------------------
    const uint32_t PE = pm.packet_entries;
    const uint64_t total_packets_lb = (E + PE - 1) / PE;
    const uint32_t P_cap = (uint32_t)std::min<uint64_t>(n_partitions_hint, std::max<uint64_t>(1, total_packets_lb / min_packets_per_partition));
    DP_TRY(hipMalloc(&d_pr0.p, (size_t)P_cap * 4));
    DP_TRY(hipMalloc(&d_prows.p, (size_t)P_cap * 4));
    DP_TRY(hipMalloc(&d_pfirst.p, (size_t)P_cap * 4));
    DP_TRY(hipMalloc(&d_pcount.p, (size_t)P_cap * 4));
    DP_TRY(hipMalloc(&d_cuts.p, sizeof(CutsOut)));
    const uint32_t balanced = opt("BALANCED_CUTS") ? (uint32_t)std::max(0, std::min(2, atoi(opt("BALANCED_CUTS")))) : 1u;
    hipLaunchKernelGGL(cuts_kernel, dim3(1), dim3(64), 0, 0, d_exp_start.as<unsigned long long>(), n_rows, PE, n_partitions_hint,
                       min_packets_per_partition, balanced, d_pr0.as<uint32_t>(), d_prows.as<uint32_t>(), d_pfirst.as<uint32_t>(),
                       d_pcount.as<uint32_t>(), d_cuts.as<CutsOut>());
    CutsOut cuts{};
    DP_TRY(hipMemcpy(&cuts, d_cuts.p, sizeof(cuts), hipMemcpyDeviceToHost));
    if (cuts.overflow) return "matrix too large (packet count overflows 32 bits)";
    const uint32_t n_parts = cuts.n_parts;
    pm.packets_per_partition = cuts.m;
    pm.n_packets = cuts.n_packets_lo;
    pm.packed_entries = (uint64_t)pm.n_packets * PE;
    pm.part_first.resize(n_parts);
    pm.part_count.resize(n_parts);
    pm.part_row0.resize(n_parts);
    pm.part_rows.resize(n_parts);
    DP_TRY(hipMemcpy(pm.part_first.data(), d_pfirst.p, (size_t)n_parts * 4, hipMemcpyDeviceToHost));
    DP_TRY(hipMemcpy(pm.part_count.data(), d_pcount.p, (size_t)n_parts * 4, hipMemcpyDeviceToHost));
    DP_TRY(hipMemcpy(pm.part_row0.data(), d_pr0.p, (size_t)n_parts * 4, hipMemcpyDeviceToHost));
    DP_TRY(hipMemcpy(pm.part_rows.data(), d_prows.p, (size_t)n_parts * 4, hipMemcpyDeviceToHost));

    // ---- the stream ---------------------------------------------------------------------------------------------------------
    DevBuf d_packets, d_pkt_row;
    const size_t stream_bytes = std::max<size_t>((size_t)pm.n_packets * pm.packet_bytes, 256);
    DP_TRY(hipMalloc(&d_packets.p, stream_bytes));
    DP_TRY(hipMalloc(&d_pkt_row.p, std::max<size_t>(pm.n_packets, 1) * 4));
    DP_TRY(hipMemset(d_packets.p, 0, stream_bytes));  // padding entries: value 0, column word 0
    DP_TRY(hipMemset(d_pkt_row.p, 0, std::max<size_t>(pm.n_packets, 1) * 4));
    ScatterParams S{};
    S.row = d_row.as<uint32_t>();
    S.col = d_col.as<uint32_t>();
    S.val = val ? d_val.as<float>() : nullptr;
    S.nnz = nnz;
    S.len = d_len.as<uint32_t>();
    S.coo_start = d_coo_start.as<unsigned long long>();
    S.exp_start = d_exp_start.as<unsigned long long>();
    S.part_row0 = d_pr0.as<uint32_t>();
    S.part_first = d_pfirst.as<uint32_t>();
    S.n_parts = n_parts;
    S.n_rows = n_rows;
    S.packets = d_packets.as<uint8_t>();
    S.pkt_row = d_pkt_row.as<uint32_t>();
    S.C = C;
    S.PE = PE;
    S.packet_bytes = pm.packet_bytes;
    S.vb = value_bytes(precision);
    S.precision = (uint32_t)precision;
    S.fixed_width = fixed_width;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_nnz), dim3(256), 0, 0, S);
    if (pm.placeholders)
        hipLaunchKernelGGL(holes_kernel, dim3(std::min<uint32_t>((n_rows + 255) / 256, 4096)), dim3(256), 0, 0, S);
    DP_TRY(hipGetLastError());
    DP_TRY(hipDeviceSynchronize());
    const auto t_end = std::chrono::steady_clock::now();
    out.upload_ms = std::chrono::duration<double, std::milli>(t_uploaded - t_begin).count();
    out.kernels_ms = std::chrono::duration<double, std::milli>(t_end - t_uploaded).count();
    out.d_packets = static_cast<uint8_t *>(d_packets.release());
    out.d_pkt_row = static_cast<uint32_t *>(d_pkt_row.release());
    if (out.keep_coo) {  // the caller goes on to pack_wsell_device with the same COO
        out.d_col = static_cast<uint32_t *>(d_col.release());
        out.d_val = static_cast<float *>(d_val.release());
    }
    kind = 0;
    return "";
}

std::string download_device_packed(DevicePacked &dp) {
    PackedMatrix &pm = dp.meta;
    pm.packets.resize((size_t)pm.n_packets * pm.packet_bytes);
    pm.pkt_row.resize(pm.n_packets);
    if (pm.n_packets) {
        DP_TRY(hipMemcpy(pm.packets.data(), dp.d_packets, pm.packets.size(), hipMemcpyDeviceToHost));
        DP_TRY(hipMemcpy(pm.pkt_row.data(), dp.d_pkt_row, (size_t)pm.n_packets * 4, hipMemcpyDeviceToHost));
    }
    return "";
}

void free_device_packed(DevicePacked &dp) {
    if (dp.d_packets) (void)hipFree(dp.d_packets);
    if (dp.d_pkt_row) (void)hipFree(dp.d_pkt_row);
    if (dp.d_col) (void)hipFree(dp.d_col);
    if (dp.d_val) (void)hipFree(dp.d_val);
    dp.d_packets = nullptr;
    dp.d_pkt_row = nullptr;
    dp.d_col = nullptr;
    dp.d_val = nullptr;
}

// ------------------------------------------------------------------------------------------------------------
// The wave-sliced ELL layout (wsell.hpp). The layout decisions depend on the row lengths alone (length classes, the
// slices, their longest-first deal to the partitions): plan_wsell makes them on the host in a few passes over the row
// ids, exactly as the host packer does. What costs the host packer its time is the fill -- 256 slots per chunk, every
// one a scattered copy -- and that is one kernel here: a wave per slice, lane l writes the 16 (or 4) value bytes and
// the 8 column-word bytes of its slots in every chunk of the slice, reading its row's entries from the COO in HBM.
// Same bytes as fill_wsell_host.
// ------------------------------------------------------------------------------------------------------------
namespace {

struct SellLaneRec {         // one lane of one slice, in STREAM order
    unsigned long long src;  // COO offset of the lane's first entry
    uint32_t n;              // entries of the lane (0: none)
    uint32_t info;           // bits 0..7: index of the segment in its row; bit 31: lane without a row
};
struct SellScatterParams {
    const uint32_t *col;
    const float *val;  // NULL: every value is 1.0
    const SellLaneRec *lanes;
    const uint32_t *chunk0, *n_chunks_of;
    uint8_t *packets;
    uint32_t n_slices, vb, packet_bytes;
    uint32_t cw_bits, pad_neutral, pad_one;  // 12-bit column words: the padding slots sit at columns 1022 / 1023
};

__global__ void __launch_bounds__(64) sell_scatter_kernel(const SellScatterParams P) {
    const uint32_t so = blockIdx.x, l = threadIdx.x;
    const SellLaneRec ln = P.lanes[(size_t)so * 64 + l];
    const uint32_t nc = P.n_chunks_of[so], chunk = P.chunk0[so];
    const bool have = (ln.info >> 31) == 0u;
    const uint32_t depth = ln.info & 255u;
    for (uint32_t c = 0; c < nc; ++c) {
        uint8_t *pkt = P.packets + (size_t)(chunk + c) * P.packet_bytes;
        float v[4];
        uint8_t qv[4];
        uint16_t cw[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t e = 4 * c + j;
            if (have && e < ln.n) {
                v[j] = P.val ? P.val[ln.src + e] : 1.0f;
                qv[j] = to_q1_7_rnd(v[j]);
                cw[j] = (uint16_t)(P.col[ln.src + e] << 2);
            } else if (!have && e == 0) {
                v[j] = -__builtin_huge_valf();
                qv[j] = 1;
                cw[j] = (uint16_t)(P.pad_one << 2);
            } else {
                v[j] = 0.0f;
                qv[j] = 0;
                cw[j] = (uint16_t)(P.pad_neutral << 2);
            }
            if (c + 1 == nc) {
                if (j == 0) cw[j] |= SELL_LAST_CHUNK;
                if (j >= 1) cw[j] |= (uint16_t)((depth >> (2 * (j - 1))) & 3u);
            }
        }
        if (P.vb == 4u) *reinterpret_cast<float4 *>(pkt + (size_t)l * 16) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<uint32_t *>(pkt + (size_t)l * 4) = (uint32_t)qv[0] | ((uint32_t)qv[1] << 8) | ((uint32_t)qv[2] << 16) | ((uint32_t)qv[3] << 24);
        if (P.cw_bits == 12u) {
            // the lane's four 12-bit words are 48 bits; two lanes share three dwords: the even lane writes the first two (the
            // second one completed with the odd lane's low 16 bits), the odd lane the third
            const uint32_t lo = (uint32_t)cw[0] | ((uint32_t)cw[1] << 12) | ((uint32_t)cw[2] << 24);
            const uint32_t hi = ((uint32_t)cw[2] >> 8) | ((uint32_t)cw[3] << 4);  // bits 32..47
            const uint32_t other_lo = (uint32_t)__shfl_xor((int)lo, 1);
            uint32_t *d = reinterpret_cast<uint32_t *>(pkt + 256u * P.vb + (size_t)(l >> 1) * 12);
            if ((l & 1u) == 0u) {
                d[0] = lo;
                d[1] = hi | (other_lo << 16);
            } else {
                d[2] = (lo >> 16) | (hi << 16);
            }
        } else {
            *reinterpret_cast<uint2 *>(pkt + 256u * P.vb + (size_t)l * 8) =
                make_uint2((uint32_t)cw[0] | ((uint32_t)cw[1] << 16), (uint32_t)cw[2] | ((uint32_t)cw[3] << 16));
        }
    }
}

}  // namespace

std::string pack_wsell_device(uint32_t rows, uint32_t cols, uint64_t nnz, const uint32_t *row, const uint32_t *col,
                              const float *val, uint32_t n_partitions_hint, SellValues values, const uint32_t *d_col_in,
                              const float *d_val_in, DeviceSell &out) {
    const auto t0 = std::chrono::steady_clock::now();
    out.d_packets = nullptr;
    SellPlan plan;
    const std::string perr = plan_wsell(rows, cols, nnz, row, col, n_partitions_hint, values, plan, out.meta);
    if (!perr.empty() || nnz == 0) return perr;
    const SellMatrix &sm = out.meta;
    std::vector<SellLaneRec> recs((size_t)sm.n_slices * 64);
    for (uint32_t so = 0; so < sm.n_slices; ++so) {
        const SellLane *src = plan.lanes.data() + (size_t)plan.stream_slice[so] * 64;
        SellLaneRec *dst = recs.data() + (size_t)so * 64;
        for (uint32_t l = 0; l < 64; ++l) {
            const SellLane &ln = src[l];
            if (ln.row == SELL_NO_ROW) dst[l] = SellLaneRec{0ull, 0u, 0x80000000u};
            else dst[l] = SellLaneRec{plan.start[ln.row] + ln.first, ln.n, ln.depth & 255u};
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    DevBuf d_col, d_val, d_recs, d_chunk0, d_nc, d_packets;
    const uint32_t *dc = d_col_in;
    const float *dv = d_val_in;
    if (!dc) {  // the COO is not in HBM yet
        DP_TRY(hipMalloc(&d_col.p, nnz * 4));
        DP_TRY(hipMemcpy(d_col.p, col, nnz * 4, hipMemcpyHostToDevice));
        dc = d_col.as<uint32_t>();
        if (val) {
            DP_TRY(hipMalloc(&d_val.p, nnz * 4));
            DP_TRY(hipMemcpy(d_val.p, val, nnz * 4, hipMemcpyHostToDevice));
            dv = d_val.as<float>();
        }
    }
    if (!val) dv = nullptr;
    DP_TRY(hipMalloc(&d_recs.p, recs.size() * sizeof(SellLaneRec)));
    DP_TRY(hipMemcpy(d_recs.p, recs.data(), recs.size() * sizeof(SellLaneRec), hipMemcpyHostToDevice));
    DP_TRY(hipMalloc(&d_chunk0.p, (size_t)sm.n_slices * 4));
    DP_TRY(hipMemcpy(d_chunk0.p, plan.chunk0.data(), (size_t)sm.n_slices * 4, hipMemcpyHostToDevice));
    DP_TRY(hipMalloc(&d_nc.p, (size_t)sm.n_slices * 4));
    DP_TRY(hipMemcpy(d_nc.p, plan.n_chunks_of.data(), (size_t)sm.n_slices * 4, hipMemcpyHostToDevice));
    DP_TRY(hipMalloc(&d_packets.p, std::max<size_t>(sm.stream_bytes(), 256)));
    const auto t2 = std::chrono::steady_clock::now();
    SellScatterParams S{};
    S.col = dc;
    S.val = dv;
    S.lanes = d_recs.as<SellLaneRec>();
    S.chunk0 = d_chunk0.as<uint32_t>();
    S.n_chunks_of = d_nc.as<uint32_t>();
    S.packets = d_packets.as<uint8_t>();
    S.n_slices = sm.n_slices;
    S.vb = (uint32_t)sm.values;
    S.packet_bytes = sm.packet_bytes;
    S.cw_bits = sm.cw_bits;
    S.pad_neutral = sm.pad_neutral;
    S.pad_one = sm.pad_one;
    hipLaunchKernelGGL(sell_scatter_kernel, dim3(sm.n_slices), dim3(64), 0, 0, S);
    DP_TRY(hipGetLastError());
    DP_TRY(hipDeviceSynchronize());
    const auto t3 = std::chrono::steady_clock::now();
    out.plan_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    out.upload_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
    out.kernels_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    out.d_packets = static_cast<uint8_t *>(d_packets.release());
    return "";
}

std::string download_device_sell(DeviceSell &ds) {
    ds.meta.packets.resize(ds.meta.stream_bytes());
    if (ds.meta.stream_bytes()) DP_TRY(hipMemcpy(ds.meta.packets.data(), ds.d_packets, ds.meta.stream_bytes(), hipMemcpyDeviceToHost));
    return "";
}

void free_device_sell(DeviceSell &ds) {
    if (ds.d_packets) (void)hipFree(ds.d_packets);
    ds.d_packets = nullptr;
}

}  // namespace tkspmv
