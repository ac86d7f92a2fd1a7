// Lab: what a kernel launch costs on this GPU as a function of its resources (dynamic LDS, scratch, registers, kernarg size).
// Back-to-back launches on one stream, HIP events around 2000 of them: time per launch = duration + launch gap.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long long a[32]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_lds(int* p) { extern __shared__ char s[]; if (p && threadIdx.x == 9999) *p = s[0]; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = (int)b.a[5]; }
__global__ void k_scratch(int* p, int n) {
    volatile int a[64];
    if (n == 12345) { for (int i = 0; i < 64; ++i) a[i] = i * n; *p = a[n & 63]; }
}
__global__ __launch_bounds__(256, 4) void k_regs(float* p, int n) {       // many live registers
    float v[100];
    if (n == 12345) {
        for (int i = 0; i < 100; ++i) v[i] = p[i];
        for (int j = 0; j < n; ++j) for (int i = 0; i < 100; ++i) v[i] = v[i] * 1.01f + v[(i + 1) % 100];
        for (int i = 0; i < 100; ++i) p[i] = v[i];
    }
}
template <typename F> static float timeit(const char* name, F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 2000; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.2f us/launch\n", name, ms * 1e3f / 2000);
    return ms;
}
int main() {
    int* d; hipMalloc(&d, 4096); float* fp = (float*)d;
    Big b{};
    for (int grid : {48, 192, 1024}) {
        printf("grid %d x 256 threads\n", grid);
        timeit("empty", [&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0, d); });
        timeit("dynamic LDS 32 KB", [&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 32768, 0, d); });
        timeit("dynamic LDS 64 KB", [&] { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 65536, 0, d); });
        timeit("256-B kernarg", [&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, 0, b, d); });
        timeit("scratch 256 B/lane", [&] { hipLaunchKernelGGL(k_scratch, dim3(grid), dim3(256), 0, 0, d, 1); });
        timeit("128 VGPRs", [&] { hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), 0, 0, fp, 1); });
    }
    return 0;
}
