// Lab: time per launch of smx_gemm for a tiny problem, called back to back from C (no Python between launches).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../speechmix_amd/csrc/gemm_common.h"
typedef int (*gemm_fn)(const SmxGemmParams*, int, hipStream_t);
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    gemm_fn gemm = (gemm_fn)dlsym(h, "smx_gemm");
    const int M = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 768;
    printf("M=%d N=%d\n", M, N);
    void *A, *B, *C;
    hipMalloc(&A, (size_t)M * 4096 * 2); hipMalloc(&B, (size_t)N * 4096 * 2); hipMalloc(&C, (size_t)M * N * 4);
    hipMemset(A, 0, (size_t)M * 4096 * 2); hipMemset(B, 0, (size_t)N * 4096 * 2);
    hipStream_t st; hipStreamCreate(&st);
    for (int K : {64, 256, 768, 3072}) {
        for (unsigned lab : {0u, 0xdead0001u, 0xdead0002u}) {
            SmxGemmParams p; memset(&p, 0, sizeof p);
            p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
            p.a.ld = K; p.b.ld = K; p.c.ld = N; p.e = p.c;
            p.nbatch = 1; p.split_k = 1; p.tr_mode = 1; p.alpha = 1.f; p.drop_seed = lab;
            for (int i = 0; i < 20; ++i) gemm(&p, 1, st);
            hipStreamSynchronize(st);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, st);
            for (int i = 0; i < 200; ++i) gemm(&p, 1, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("K=%4d %s: %.2f us/launch\n", K, lab == 0 ? "full        " : lab == 0xdead0001u ? "return at top" : "no epilogue ", ms * 5.0f);
        }
    }
    return 0;
}
