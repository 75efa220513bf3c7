// Can the host write device memory directly (large BAR)? A SIGSEGV/SIGBUS handler jumps back if the pointer is not host-accessible.
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <signal.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static sigjmp_buf jb;
static void on_fault(int) { siglongjmp(jb, 1); }
__global__ void sum_kernel(const float *x, float *out) { float s = 0; for (int i = 0; i < 1024; ++i) s += x[i]; *out = s; }
int main() {
    int large_bar = -1;
    CK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    for (int kind = 0; kind < 2; ++kind) {
        float *d = nullptr, *out = nullptr;
        if (kind == 0) CK(hipMalloc((void **)&d, 4096));
        else CK(hipExtMallocWithFlags((void **)&d, 4096, hipDeviceMallocFinegrained));
        CK(hipMalloc((void **)&out, 4));
        CK(hipMemset(d, 0, 4096));
        CK(hipDeviceSynchronize());
        fflush(stdout);
        signal(SIGSEGV, on_fault);
        signal(SIGBUS, on_fault);
        if (sigsetjmp(jb, 1) == 0) {
            volatile float *p = d;
            for (int i = 0; i < 1024; ++i) p[i] = 1.0f;
            __builtin_ia32_sfence();
        } else {
            printf("kind %d (%s): host store FAULTED\n", kind, kind ? "fine-grained" : "hipMalloc");
            continue;
        }
        signal(SIGSEGV, SIG_DFL);
        signal(SIGBUS, SIG_DFL);
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1), 0, 0, d, out);
        float h = -1;
        CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
        printf("kind %d (%s): host stores accepted; the GPU sums them to %.1f (1024 expected)\n", kind, kind ? "fine-grained" : "hipMalloc", h);
        if (h == 1024.0f) {  // time 4 KiB of host stores + sfence from THIS process
            std::vector<float> src(1024, 2.0f);
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 1000; ++r) { memcpy(d, src.data(), 4096); __builtin_ia32_sfence(); }
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000;
            hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1), 0, 0, d, out);
            CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
            printf("   4 KiB memcpy + sfence into it: %.2f us each; the GPU then sums %.1f (2048 expected)\n", us, h);
        }
    }
    return 0;
}
