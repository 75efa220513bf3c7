// kernels/radix_select.hpp -- Large k: scores + radix select (radix_hist_kernel, radix_filter_kernel).
// Part of engine.hip (one translation unit: included there in this order; device code only).
#pragma once
#include "select.hpp"

namespace tkspmv {

// ------------------------------------------------------------------------------------------------------------
// Large k: scores + radix select. The threshold exchange needs k well below the number of publishing groups (at most
// 1024); beyond that the k-th largest group maximum is a weak bound (k = 500: 96 us per query, k = 1000: 207 us) and
// for k = 1023, 1024 it cannot form at all (every row becomes a candidate: 33 ms). Such engines take the reference GPU
// host's route (host_spmv_topk_csr_gpu.cu:171-231: full y, then a selection over all rows), with a selection that is
// not a sort: the SpMV-only variant of the stream kernel writes every row's score, four 8-bit histogram passes over the
// order keys find the k-th largest key T exactly, a filter pass appends the rows with key >= T (k of them plus ties) to
// the overflow list, and the ordinary selection kernel ranks those (score desc, row desc). Rows below min_score and
// rows without entries (their score slot keeps -inf) never count.
// ------------------------------------------------------------------------------------------------------------
struct RadixParams {
    const float *scores;  // [rows]; -inf where a row has no entry
    uint32_t rows, k;
    uint32_t kmin;        // order key of min_score: keys below it are not eligible
    uint32_t *hist;       // [4][256], zeroed before the first pass
    unsigned long long *ovf_cand;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
};
constexpr uint32_t RADIX_THREADS = 1024;

// From the histograms of passes 0 .. n_pass-1: the key prefix decided so far and how many keys of the next pass's bins
// are still wanted. Called by wave 0; take_all: fewer eligible keys than k exist (every eligible row is a result).
__device__ __forceinline__ void radix_decide(const RadixParams &R, int n_pass, uint32_t lane, uint32_t &prefix, uint32_t &k_rem,
                                             bool &take_all) {
    prefix = 0u;
    k_rem = R.k;
    take_all = false;
    for (int q = 0; q < n_pass; ++q) {
        uint32_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = __hip_atomic_load(&R.hist[q * 256 + 4 * (int)lane + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t tot = h[0] + h[1] + h[2] + h[3];
        uint32_t above = tot;  // inclusive suffix sum over lanes >= this one ...
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_down((int)above, d);
            above += (lane + (uint32_t)d < 64u) ? o : 0u;
        }
        above -= tot;  // ... made exclusive: keys in the bins of higher lanes
        uint32_t hit_bin = 0xFFFFFFFFu, hit_above = 0u;
        uint32_t c = above;
#pragma unroll
        for (int j = 3; j >= 0; --j) {  // bins from the top
            if (hit_bin == 0xFFFFFFFFu && c < k_rem && c + h[j] >= k_rem) {
                hit_bin = 4u * lane + (uint32_t)j;
                hit_above = c;
            }
            c += h[j];
        }
        const uint64_t hb = __ballot(hit_bin != 0xFFFFFFFFu);
        if (hb == 0ull) {  // fewer than k_rem keys left (only possible in pass 0: the bins of a later pass hold >= k_rem)
            take_all = true;
            return;
        }
        const int src = __builtin_ctzll(hb);
        const uint32_t bin = (uint32_t)__shfl((int)hit_bin, src);
        k_rem -= (uint32_t)__shfl((int)hit_above, src);
        prefix = (prefix << 8) | bin;
    }
}

__global__ void __launch_bounds__(RADIX_THREADS) radix_hist_kernel(const RadixParams R, const int pass) {
    __shared__ uint32_t lh[256];
    __shared__ uint32_t sh_prefix, sh_take_all;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 256u) lh[tid] = 0u;
    if (tid < 64u) {
        uint32_t prefix, k_rem;
        bool take_all;
        radix_decide(R, pass, lane, prefix, k_rem, take_all);
        if (tid == 0u) {
            sh_prefix = prefix;
            sh_take_all = take_all ? 1u : 0u;
        }
    }
    __syncthreads();
    if (sh_take_all) return;
    const uint32_t prefix = sh_prefix, shift = 24u - 8u * (uint32_t)pass;
    for (uint32_t i = blockIdx.x * RADIX_THREADS + tid; i < R.rows; i += gridDim.x * RADIX_THREADS) {
        const float sc = R.scores[i];
        const uint32_t key = order_key(sc);
        // (a row without entries keeps -inf in its slot and is never eligible, whatever min_score is)
        if (key >= R.kmin && sc > -__builtin_huge_valf() && (pass == 0 || (key >> (shift + 8u)) == prefix))
            atomicAdd(&lh[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256u && lh[tid] != 0u) atomicAdd(&R.hist[pass * 256 + (int)tid], lh[tid]);
}

__global__ void __launch_bounds__(RADIX_THREADS) radix_filter_kernel(const RadixParams R) {
    __shared__ uint32_t sh_thr;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < 64u) {
        uint32_t prefix, k_rem;
        bool take_all;
        radix_decide(R, 4, lane, prefix, k_rem, take_all);
        if (tid == 0u) sh_thr = (take_all || prefix < R.kmin) ? R.kmin : prefix;
    }
    __syncthreads();
    const uint32_t thr = sh_thr;
    const uint32_t n_iter = (R.rows + gridDim.x * RADIX_THREADS - 1u) / (gridDim.x * RADIX_THREADS);  // uniform trip count
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t i = (it * gridDim.x + blockIdx.x) * RADIX_THREADS + tid;
        const float sc = i < R.rows ? R.scores[i] : -__builtin_huge_valf();
        const bool keep = i < R.rows && order_key(sc) >= thr && sc > -__builtin_huge_valf();
        const uint64_t bm = __ballot(keep);
        uint32_t base = 0u;
        if (lane == 0u && bm) base = atomicAdd(R.ovf_count, (uint32_t)__popcll(bm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) {
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (pos < R.ovf_cap) R.ovf_cand[pos] = pack_cand(__float_as_uint(sc), i);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The FPGA design's K-lists-per-partition approximation ("approximation C"): the reference cuts the rows
// into SPMV_PARTITIONS ranges (p = row / ceil(N / P), host_spmv_bscsr.cpp:133-141), every core keeps only K candidates
// (types.hpp:49) and the host merges the P x K of them (host_spmv_bscsr.cpp:399-448); a row of the true top-k that is
// not among the K best of its partition is lost (topk_errors.py:29-42 models the loss). Engines created with
// partitions = P and k > k_per_partition run the STRICT form of that scheme -- the one topk_errors.py models: the SpMV-only
// kernel writes every score, one workgroup per partition selects the k_per_partition best rows of its range exactly (score
// desc, row desc) and appends them to the candidate list, and the ordinary selection kernel ranks the union. The HLS core
// itself is looser: it keeps LIMITED_FINISHED_ROWS independent K-lists per partition (res_local[4][K],
// spmv_bscsr_top_k_multicore.hpp:466) and its host merges K x packet-size entries per partition
// (host_spmv_bscsr.cpp:410-417), so its candidate set is a superset of this one; oracle/hls_model.c restates that
// dataflow and experiments.py reports both.
// ------------------------------------------------------------------------------------------------------------
struct PartitionParams {
    const float *scores;  // [rows]; -inf where a row has no entry
    uint32_t rows, per, k_part;
    uint32_t kmin;        // order key of min_score
    unsigned long long *ovf_cand;
    uint32_t *ovf_count;
    uint32_t ovf_cap;
};

__global__ void __launch_bounds__(RADIX_THREADS) partition_topk_kernel(const PartitionParams R) {
    __shared__ uint32_t cnt;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t a = blockIdx.x * R.per;
    const uint32_t b = (a + R.per < R.rows && a + R.per > a) ? a + R.per : R.rows;
    if (a >= b) return;
    // number of rows of the range for which pred(key, row) holds (uniform result)
    auto count = [&](auto pred) -> uint32_t {
        if (tid == 0) cnt = 0u;
        __syncthreads();
        uint32_t c = 0;
        for (uint32_t i = a + tid; i < b; i += RADIX_THREADS) {
            const float sc = R.scores[i];
            const uint32_t key = order_key(sc);
            c += (key >= R.kmin && sc > -__builtin_huge_valf() && pred(key, i)) ? 1u : 0u;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d);
        if (lane == 0 && c) atomicAdd(&cnt, c);
        __syncthreads();
        const uint32_t r = cnt;
        __syncthreads();
        return r;
    };
    // T = the k_part-th largest eligible key of the range (bisection on the key bits); fewer eligible rows: all of them
    uint32_t T = R.kmin, R0 = a;
    const uint32_t eligible = count([](uint32_t, uint32_t) { return true; });
    if (eligible > R.k_part) {
        uint32_t prefix = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t trial = prefix | (1u << bit);
            if (count([trial](uint32_t key, uint32_t) { return key >= trial; }) >= R.k_part) prefix = trial;
        }
        T = prefix;
        const uint32_t n_gt = count([T](uint32_t key, uint32_t) { return key > T; });
        const uint32_t n_eq = count([T](uint32_t key, uint32_t) { return key == T; });
        const uint32_t need = R.k_part - n_gt;  // >= 1
        if (n_eq > need) {  // equal scores across the cut: the higher row ids win (sort_tuples order)
            uint32_t rp = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t trial = rp | (1u << bit);
                if (count([T, trial](uint32_t key, uint32_t row) { return key == T && row >= trial; }) >= need) rp = trial;
            }
            R0 = rp;
        }
    }
    for (uint32_t i0 = a; i0 < b; i0 += RADIX_THREADS) {  // uniform trip count
        const uint32_t i = i0 + tid;
        const float sc = i < b ? R.scores[i] : -__builtin_huge_valf();
        const uint32_t key = order_key(sc);
        const bool keep = i < b && sc > -__builtin_huge_valf() && key >= R.kmin && (key > T || (key == T && i >= R0));
        const uint64_t bm = __ballot(keep);
        uint32_t base = 0u;
        if (lane == 0u && bm) base = atomicAdd(R.ovf_count, (uint32_t)__popcll(bm));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) {
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (pos < R.ovf_cap) R.ovf_cand[pos] = pack_cand(__float_as_uint(sc), i);
        }
    }
}

}  // namespace tkspmv
