// lds_probe.hip -- calibration: how much LDS per workgroup still lets TWO 576-thread workgroups share a CU on this GPU?
// Launches 2 x CUs workgroups that each hold `bytes` of dynamic LDS and spin ~60 us; if both fit, all of them enter
// within a few microseconds, otherwise half of them enter when the first half leaves.
// Build: hipcc -O3 --offload-arch=gfx950 -o bin/lds_probe tools/lds_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VGPRS>
__global__ void __launch_bounds__(576) hold(unsigned long long *entry, unsigned long long ticks) {
    extern __shared__ unsigned char lds[];
    if (VGPRS == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");  // make the kernel occupy 72 VGPRs like the engine's
    if (VGPRS == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        lds[0] = 1;
        entry[blockIdx.x] = t0;
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = 2 * prop.multiProcessorCount;
    unsigned long long *d;
    CK(hipMalloc((void **)&d, grid * 8));
    std::vector<unsigned long long> h(grid);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(hold<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(hold<72>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(hold<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int vg : {8, 72, 96})
    for (int kb : {16, 40, 46, 48, 52, 56, 64, 80}) {
        if (vg == 8) hipLaunchKernelGGL(hold<8>, dim3(grid), dim3(576), kb * 1024, 0, d, 6000ull);
        else if (vg == 72) hipLaunchKernelGGL(hold<72>, dim3(grid), dim3(576), kb * 1024, 0, d, 6000ull);
        else hipLaunchKernelGGL(hold<96>, dim3(grid), dim3(576), kb * 1024, 0, d, 6000ull);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost));
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        int late = 0;
        for (auto t : h) late += (t - t0) > 2000;  // entered more than 20 us after the first
        printf("VGPRs %2d, LDS %3d KiB per workgroup: %3d of %d workgroups entered late -> %s\n", vg, kb, late, grid,
               late == 0 ? "two per CU" : "one per CU");
    }
    return 0;
}
