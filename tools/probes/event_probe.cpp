// What an event pair around ONE kernel launch measures beyond the kernel (hipEventRecord on an idle stream: the start event's
// timestamp is taken before the host has even written the dispatch packet), against hipExtLaunchKernelGGL's start / stop events
// (the dispatch's own timestamps).   hipcc -O2 --offload-arch=gfx950 -o /tmp/event_probe tools/probes/event_probe.cpp
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned long long ticks, unsigned long long *out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    unsigned long long *d;
    hipMalloc(&d, 8);
    for (unsigned long long ticks : {0ull, 10000ull, 35000ull}) {  // 0, 100, 350 us
        std::vector<float> rec, ext;
        for (int i = 0; i < 40; ++i) {
            hipStreamSynchronize(s);
            hipEventRecord(a, s);
            hipLaunchKernelGGL(spin, dim3(512), dim3(576), 0, s, ticks, d);
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            rec.push_back(ms * 1e3f);
            hipStreamSynchronize(s);
            hipExtLaunchKernelGGL(spin, dim3(512), dim3(576), 0, s, a, b, 0, ticks, d);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            ext.push_back(ms * 1e3f);
        }
        std::sort(rec.begin(), rec.end());
        std::sort(ext.begin(), ext.end());
        printf("spin %6.1f us: hipEventRecord pair median %7.2f (min %7.2f)   hipExtLaunchKernelGGL events median %7.2f (min %7.2f)\n",
               ticks * 0.01, rec[20], rec[0], ext[20], ext[0]);
    }
    return 0;
}
