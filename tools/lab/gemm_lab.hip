// GEMM laboratory (not part of the product library): includes the production kernels and times stripped variants of
// the 128x128x64 LDS-DMA kernel to see which resource bounds it.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../speechmix_amd/csrc gemm_lab.hip ../../speechmix_amd/csrc/gemm_pp.hip -o gemm_lab
#include "../../speechmix_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

// MODE 0: full loop, no epilogue   1: loads only   2: LDS reads + MFMA only (no global loads)
template <int MODE, bool A_RC = false, bool B_RC = false>
__global__ __launch_bounds__(256, 4) void lab_kernel(SmxGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    float sink = 0.f;
    for (int lin = blockIdx.x; lin < nwg; lin += gridDim.x) {
        int wg = lin;
        {
            const int q = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        }
        int tm, tn;
        {
            const int per_group = GROUP_N * ntm;
            const int grp = wg / per_group, rem = wg - grp * per_group;
            const int first = grp * GROUP_N;
            const int gsz = min(ntn - first, GROUP_N);
            tm = rem / gsz;
            tn = first + (rem - tm * gsz);
        }
        const int m0 = tm * BM, n0 = tn * BN;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int kst = (p.K + BK - 1) / BK;
        const int per = (kst + p.split_k - 1) / p.split_k;
        const int ksb = blockIdx.z * per, ks1 = min(kst, ksb + per);
        DmaLoader<A_RC> la;
        DmaLoader<B_RC> lb;
        la.init(A, p.a, m0, p.M, ksb * BK, tid);
        lb.init(B, p.b, n0, p.N, ksb * BK, tid);
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        char* tA = smem;
        char* tB = smem + 16384;
        for (int ks = ksb; ks < ks1; ++ks) {
            if (MODE != 2) {
                la.issue(tA, p.a, m0, p.M, ks * BK, p.K, tid);
                lb.issue(tB, p.b, n0, p.N, ks * BK, p.K, tid);
            }
            __syncthreads();
            if (MODE != 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8_t fa[4], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i] = load_frag<A_RC>(tA, wm * 64 + i * 16, kk, lane, 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_RC>(tB, wn * 64 + j * 16, kk, lane, 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    if (sink == 1234.5678f) reinterpret_cast<float*>(p.C)[tid] = sink;
}

// 256 x 128 x 64 tile, 4 waves of 128 x 64 (acc 8x4), one 48-KB LDS buffer, 3 workgroups / CU.  MODE as above.
template <int MODE>
__global__ __launch_bounds__(256, 3) void lab256_kernel(SmxGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + 127) / 128, ntm = (p.M + 255) / 256;
    const int nwg = ntn * ntm;
    float sink = 0.f;
    for (int lin = blockIdx.x; lin < nwg; lin += gridDim.x) {
        int wg = lin;
        {
            const int q = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        }
        int tm, tn;
        {
            const int per_group = GROUP_N * ntm;
            const int grp = wg / per_group, rem = wg - grp * per_group;
            const int first = grp * GROUP_N;
            const int gsz = min(ntn - first, GROUP_N);
            tm = rem / gsz;
            tn = first + (rem - tm * gsz);
        }
        const int m0 = tm * 256, n0 = tn * 128;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int ks1 = (p.K + BK - 1) / BK;
        DmaLoader<false> la0, la1, lb;
        la0.init(A, p.a, m0, p.M, 0, tid);
        la1.init(A, p.a, m0 + 128, p.M, 0, tid);
        lb.init(B, p.b, n0, p.N, 0, tid);
        f32x4_t acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        char* tA = smem;
        char* tB = smem + 32768;
        for (int ks = 0; ks < ks1; ++ks) {
            if (MODE != 2) {
                la0.issue(tA, p.a, m0, p.M, ks * BK, p.K, tid);
                la1.issue(tA + 16384, p.a, m0 + 128, p.M, ks * BK, p.K, tid);
                lb.issue(tB, p.b, n0, p.N, ks * BK, p.K, tid);
            }
            __syncthreads();
            if (MODE != 1) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8_t fa[8], fb[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) fa[i] = load_frag<false>(tA + wm * 16384, i * 16, kk, lane, 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = load_frag<false>(tB, wn * 64 + j * 16, kk, lane, 1);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    if (sink == 1234.5678f) reinterpret_cast<float*>(p.C)[tid] = sink;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F>
static float time_us(F&& f, int n = 20) {
    for (int i = 0; i < 3; ++i) f();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / n;
}

int main(int argc, char** argv) {
    const bool stripped = argc > 1 && atoi(argv[1]) == 1;
    const int shapes[][3] = {{15968, 3072, 768}, {15968, 768, 3072}, {15968, 768, 768}, {16384, 4096, 1024}, {1024, 768, 768},
                             {1024, 768, 3072}, {1024, 3072, 768}, {7968, 768, 768}, {511968, 512, 1536}, {15968, 2304, 768}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        bf16_t *A, *B, *C;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(B, 0x3c, (size_t)N * K * 2));
        const double fl = 2.0 * M * N * K;
        int tiles = ((M + 127) / 128) * ((N + 127) / 128);
        SmxGemmParams p = {};
        p.A = A; p.B = B; p.C = C;
        p.c = SmxRowView{0, N, 0, 0, 0}; p.e = p.c;
        p.M = M; p.N = N; p.K = K; p.nbatch = 1; p.split_k = 1; p.alpha = 1.f;
        if (stripped) {
            p.a = SmxRowView{0, K, 0, 0, 0}; p.b = SmxRowView{0, K, 0, 0, 0}; p.tr_mode = 1;
            dim3 grid(tiles > 1024 ? 1024 : tiles);
            float t_full = time_us([&] { smx_gemm(&p, SMX_BF16, 0); });
            float t0 = time_us([&] { hipLaunchKernelGGL(lab_kernel<0>, grid, dim3(256), 32768, 0, p); });
            float t1 = time_us([&] { hipLaunchKernelGGL(lab_kernel<1>, grid, dim3(256), 32768, 0, p); });
            float t2 = time_us([&] { hipLaunchKernelGGL(lab_kernel<2>, grid, dim3(256), 32768, 0, p); });
            {
                int t256 = ((M + 255) / 256) * ((N + 127) / 128);
                dim3 g2(t256 > 768 ? 768 : t256);
                (void)hipFuncSetAttribute((const void*)lab256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
                (void)hipFuncSetAttribute((const void*)lab256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
                (void)hipFuncSetAttribute((const void*)lab256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
                float u0 = time_us([&] { hipLaunchKernelGGL(lab256_kernel<0>, g2, dim3(256), 49152, 0, p); });
                float u1 = time_us([&] { hipLaunchKernelGGL(lab256_kernel<1>, g2, dim3(256), 49152, 0, p); });
                float u2 = time_us([&] { hipLaunchKernelGGL(lab256_kernel<2>, g2, dim3(256), 49152, 0, p); });
                printf("   256x128 tile (%d tiles): no-epilogue %.1f us (%.0f TF) | loads-only %.1f us | compute-only %.1f us (%.0f TF)\n", t256, u0,
                       fl / u0 / 1e6, u1, u2, fl / u2 / 1e6);
            }
            const double bytes = (double)tiles * ((K + 63) / 64) * 32768.0;
            printf("M=%d N=%d K=%d tiles=%d: production %.1f us (%.0f TF) | no-epilogue %.1f us (%.0f TF) | loads-only %.1f us (%.2f TB/s L2->LDS) | "
                   "compute-only %.1f us (%.0f TF)\n", M, N, K, tiles, t_full, fl / t_full / 1e6, t0, fl / t0 / 1e6, t1, bytes / t1 / 1e6, t2, fl / t2 / 1e6);
        } else {
            printf("M=%d N=%d K=%d tiles=%d:", M, N, K, tiles);
            const int rcs[3][2] = {{0, 0}, {0, 1}, {1, 1}};
            for (auto& rc : rcs) {
                p.a_rc = rc[0]; p.b_rc = rc[1];
                p.a = SmxRowView{0, rc[0] ? M : K, 0, 0, 0};
                p.b = SmxRowView{0, rc[1] ? N : K, 0, 0, 0};
                printf("  [a_rc=%d b_rc=%d]", rc[0], rc[1]);
                for (int tr : {1}) {
                    p.tr_mode = tr;
                    float t = time_us([&] { if (smx_gemm(&p, SMX_BF16, 0)) { printf("launch failed\n"); exit(1); } });
                    printf(" tr%d %.1fus(%.0fTF)", tr, t, fl / t / 1e6);
                }
            }
            printf("\n");
        }
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    // weight-gradient shapes: C[M,N] = A^T B with both operands rows-contiguous ([K, M] and [K, N]), split-K slabs
    const int wshapes[][4] = {{768, 3072, 15968, 7}, {768, 3072, 15968, 6}, {2304, 768, 15968, 9}, {2304, 768, 15968, 8}, {512, 1536, 511968, 21}, {512, 1536, 511968, 16}, {512, 1536, 511968, 12}, {768, 768, 15968, 28}, {768, 768, 15968, 21}, {768, 768, 15968, 14}, {50265, 768, 1024, 1}};
    for (auto& s : wshapes) {
        const int M = s[0], N = s[1], K = s[2], split = s[3];
        bf16_t *A, *B; float* C;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4 * split));
        CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(B, 0x3c, (size_t)N * K * 2));
        const double fl = 2.0 * M * N * K;
        const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
        SmxGemmParams p = {};
        p.A = A; p.B = B; p.C = C;
        p.a = SmxRowView{0, M, 0, 0, 0}; p.b = SmxRowView{0, N, 0, 0, 0}; p.c = SmxRowView{0, N, 0, 0, 0}; p.e = p.c;
        p.M = M; p.N = N; p.K = K; p.nbatch = 1; p.split_k = split; p.alpha = 1.f; p.a_rc = 1; p.b_rc = 1; p.out_f32 = 1;
        p.split_stride = (long long)M * N; p.tr_mode = 1;
        dim3 grid(tiles > 1024 ? 1024 : tiles, 1, split);
        float t_full = time_us([&] { smx_gemm(&p, SMX_BF16, 0); });
        float t0 = time_us([&] { hipLaunchKernelGGL((lab_kernel<0, true, true>), grid, dim3(256), 32768, 0, p); });
        float t1 = time_us([&] { hipLaunchKernelGGL((lab_kernel<1, true, true>), grid, dim3(256), 32768, 0, p); });
        float t2 = time_us([&] { hipLaunchKernelGGL((lab_kernel<2, true, true>), grid, dim3(256), 32768, 0, p); });
        p.split_k = 1; p.a_rc = 0; p.b_rc = 0; p.a = SmxRowView{0, K, 0, 0, 0}; p.b = SmxRowView{0, K, 0, 0, 0};
        p.split_k = split;
        float t3 = time_us([&] { hipLaunchKernelGGL((lab_kernel<0, false, false>), grid, dim3(256), 32768, 0, p); });
        printf("wgrad M=%d N=%d K=%d split=%d tiles=%d: production %.1f us (%.0f TF) | no-epilogue %.1f us (%.0f TF) | loads-only %.1f us | "
               "compute-only %.1f us (%.0f TF) | same shape with K-contiguous operands, no epilogue %.1f us (%.0f TF)\n", M, N, K, split, tiles,
               t_full, fl / t_full / 1e6, t0, fl / t0 / 1e6, t1, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    return 0;
}
