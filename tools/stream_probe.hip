// stream_probe.hip -- calibration only (not part of the product): how fast can ANY kernel read the 121 MB of one
// query's packet stream on this GPU, launched back to back over 4 rotating copies (cache-defeated)?
// Build: hipcc -O3 --offload-arch=gfx950 -o bin/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

// each wave owns a contiguous chunk; U independent 1-KiB wave loads in flight
template <int U, bool NT>
__global__ void __launch_bounds__(512) chunk_kernel(const float4v *__restrict__ src, size_t n_vec, float *out) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const size_t per_wave = ((n_vec / 64 + n_waves - 1) / n_waves);  // in 64-vector (1 KiB) units
    size_t i = (size_t)wave * per_wave, end = i + per_wave;
    if (end > n_vec / 64) end = n_vec / 64;
    float4v acc = {0, 0, 0, 0};
    for (; i + U <= end; i += U) {
        float4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4v *p = src + (i + u) * 64 + lane;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < end; ++i) acc += src[i * 64 + lane];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[wave] = acc.x;
}

// grid-stride: consecutive waves read consecutive KiB
template <int U, bool NT>
__global__ void __launch_bounds__(512) stride_kernel(const float4v *__restrict__ src, size_t n_vec, float *out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n_thr = (size_t)gridDim.x * blockDim.x;
    float4v acc = {0, 0, 0, 0};
    size_t i = tid;
    for (; i + (U - 1) * n_thr < n_vec; i += U * n_thr) {
        float4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4v *p = src + i + u * n_thr;
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n_vec; i += n_thr) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[tid] = acc.x;
}

template <typename K>
static void run(const char *name, K kernel, int grid, int block, std::vector<float4v *> &bufs, size_t n_vec, float *out,
                hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int iters = 400;
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, bufs[i % bufs.size()], n_vec, out);
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, bufs[i % bufs.size()], n_vec, out);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = 1e3 * ms / iters;
    printf("%-28s grid %5d x %4d  replicas %zu : %7.2f us  %6.2f TB/s\n", name, grid, block, bufs.size(), us,
           n_vec * 16.0 / us * 1e-6);
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
}

int main(int argc, char **argv) {
    const size_t bytes = argc > 1 ? strtoull(argv[1], 0, 10) : 117660000ull;
    const size_t n_vec = bytes / 16 / 64 * 64;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<float4v *> cold(4), warm(1);
    for (auto &p : cold) {
        CK(hipMalloc((void **)&p, n_vec * 16));
        CK(hipMemset(p, 0, n_vec * 16));
    }
    warm[0] = cold[0];
    float *out;
    CK(hipMalloc((void **)&out, 1 << 24));
    CK(hipDeviceSynchronize());
    printf("bytes per launch: %zu\n", n_vec * 16);
    for (int pass = 0; pass < 2; ++pass) {
        auto &b = pass == 0 ? cold : warm;
        printf("---- %s ----\n", pass == 0 ? "cold (4 rotating copies)" : "warm (one copy, Infinity Cache)");
        for (int grid : {512, 1024, 2048}) {
            run("chunk U=2", chunk_kernel<2, false>, grid, 512, b, n_vec, out, s);
            run("chunk U=4", chunk_kernel<4, false>, grid, 512, b, n_vec, out, s);
            run("chunk U=8", chunk_kernel<8, false>, grid, 512, b, n_vec, out, s);
            run("chunk U=4 nontemporal", chunk_kernel<4, true>, grid, 512, b, n_vec, out, s);
        }
        for (int grid : {512, 1024, 2048, 4096, 8192}) {
            run("stride U=2", stride_kernel<2, false>, grid, 512, b, n_vec, out, s);
            run("stride U=4", stride_kernel<4, false>, grid, 512, b, n_vec, out, s);
            run("stride U=8", stride_kernel<8, false>, grid, 512, b, n_vec, out, s);
            run("stride U=4 nontemporal", stride_kernel<4, true>, grid, 512, b, n_vec, out, s);
        }
        run("stride U=4 block 256", stride_kernel<4, false>, 4096, 256, b, n_vec, out, s);
        run("stride U=4 block 1024", stride_kernel<4, false>, 1024, 1024, b, n_vec, out, s);
    }
    return 0;
}
