// GEMM laboratory (not part of the product library): the 128x128 LDS-DMA kernel's wave-level work (16 waves per CU, 64 x 64
// wave tiles, 4 waves per SIMD) arranged as ONE 1024-thread workgroup per CU that shares a 256 x 256 output tile, so the
// L2 -> LDS operand traffic per flop halves (the stripped variants of gemm_lab show the 128x128 kernel's fills alone take
// longer than its MFMA work).  Two 64-KB LDS stages, one barrier per K tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../speechmix_amd/csrc gemm16w_lab.hip ../../speechmix_amd/csrc/gemm_pp.hip -o gemm16w_lab
#include "../../speechmix_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define W16_STAGE 65536
__device__ unsigned long long w16_clk[4];
#define W16_GROUP_N 4

template <bool A_RC, bool B_RC, int EPI>
__global__ __launch_bounds__(1024, 1) void w16_kernel(SmxGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int q = wave >> 2, tq = tid & 255;              // loader role: quarter q fills sub-tile q (A0, A1, B0, B1)
    const int ntn = (p.N + 255) / 256, ntm = (p.M + 255) / 256;
    const int nwg = ntn * ntm;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int lin = blockIdx.x; lin < nwg; lin += gridDim.x) {
        int wg = lin;
        {
            const int qq = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
            wg = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + y;
        }
        int tm, tn;
        {
            const int per_group = W16_GROUP_N * ntm;
            const int grp = wg / per_group, rem = wg - grp * per_group;
            const int first = grp * W16_GROUP_N;
            const int gsz = min(ntn - first, W16_GROUP_N);
            tm = rem / gsz;
            tn = first + (rem - tm * gsz);
        }
        const int m0 = tm * 256, n0 = tn * 256;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
        const int ks1 = (p.K + BK - 1) / BK;
        DmaLoader<A_RC> la;
        DmaLoader<B_RC> lb;
        const int lrow0 = (q < 2 ? m0 : n0) + (q & 1) * 128;
        if (q < 2) la.init(A, p.a, lrow0, p.M, 0, tq);
        else lb.init(B, p.b, lrow0, p.N, 0, tq);
        auto issue = [&](int stage, int ks) {
            char* dst = smem + stage * W16_STAGE + q * 16384;
            if (q < 2) la.template issue<true>(dst, p.a, lrow0, p.M, ks * BK, p.K, tq);
            else lb.template issue<true>(dst, p.b, lrow0, p.N, ks * BK, p.K, tq);
        };
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        issue(0, 0);
        for (int ks = 0; ks < ks1; ++ks) {
            // my pieces of tile ks landed; after the barrier everyone's did, and everyone is past the reads of tile ks - 1
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (ks + 1 < ks1) issue((ks + 1) & 1, ks + 1);
            const char* st = smem + (ks & 1) * W16_STAGE;
            const char* tA = st + (wm >> 1) * 16384;
            const char* tB = st + 32768 + (wn >> 1) * 16384;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = load_frag<A_RC>(tA, (wm & 1) * 64 + i * 16, kk, lane, 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_RC>(tB, (wn & 1) * 64 + j * 16, kk, lane, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // tile memory becomes the waves' transposition slices
        if (p.drop_seed == 0xdead0002u) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.C)[tid] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3]; continue; }
        {
            int lane_e = lane, wave_e = wave;
            asm volatile("" : "+v"(lane_e), "+v"(wave_e));
            wave_e = __builtin_amdgcn_readfirstlane(wave_e);
            auto ka = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const SmxGemmParams& pe = *(const SmxGemmParams*)ka;
            if constexpr (EPI >= 0)
                epilogue_staged_fast<EPI, 0>(pe, acc, smem + wave_e * 8192, m0 + (wave_e >> 2) * 64, n0 + (wave_e & 3) * 64, 0, 0, 0, lane_e);
            else
                epilogue_staged(pe, acc, smem + wave_e * 8192, m0 + (wave_e >> 2) * 64, n0 + (wave_e & 3) * 64, 0, 0, 0, lane_e);
        }
        if (lin + (int)gridDim.x < nwg) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (blockIdx.x == 0 && tid == 0) {
        w16_clk[0] = __builtin_readcyclecounter() - c0;
        w16_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// ---- eight waves on a 256 x 128 tile, TWO workgroups per CU (single 48-KB stage each; 64 KB of LDS for the epilogue slices):
// the 128x128 kernel's structure - independent workgroups hide each other's fills - with 25 % fewer bytes into LDS per flop.
// Its K loop time matches the fill of 128 KB per CU per K step at ~42 B/clk exactly, so the fill volume is what to cut.
template <int EPI>
__global__ __launch_bounds__(512, 2) void w8_kernel(SmxGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int q = wave >> 2, tq = tid & 255;              // loader role: half q fills A rows 128 q .. and B rows 64 q ..
    const int ntn = (p.N + 127) / 128, ntm = (p.M + 255) / 256;
    const int nwg = ntn * ntm;
    for (int lin = blockIdx.x; lin < nwg; lin += gridDim.x) {
        int wg = lin;
        {
            const int qq = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
            wg = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + y;
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
        DmaLoader<false> la;
        DmaLoader<false, 2> lb;
        la.init(A, p.a, m0 + q * 128, p.M, 0, tq);
        lb.init(B, p.b, n0 + q * 64, p.N, 0, tq);
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        char* tA = smem + (wm >> 1) * 16384;
        char* tB = smem + 32768;
        for (int ks = 0; ks < ks1; ++ks) {
            la.issue(smem + q * 16384, p.a, m0 + q * 128, p.M, ks * BK, p.K, tq);
            lb.issue(tB + q * 8192, p.b, n0 + q * 64, p.N, ks * BK, p.K, tq);
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = load_frag<false>(tA, (wm & 1) * 64 + i * 16, kk, lane, 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = load_frag<false>(tB, wn * 64 + j * 16, kk, lane, 1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        if (p.drop_seed == 0xdead0002u) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.C)[tid] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3]; continue; }
        {
            int lane_e = lane, wave_e = wave;
            asm volatile("" : "+v"(lane_e), "+v"(wave_e));
            wave_e = __builtin_amdgcn_readfirstlane(wave_e);
            auto ka = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const SmxGemmParams& pe = *(const SmxGemmParams*)ka;
            epilogue_staged_fast<EPI, 0>(pe, acc, smem + wave_e * 8192, m0 + (wave_e >> 1) * 64, n0 + (wave_e & 1) * 64, 0, 0, 0, lane_e);
        }
        if (lin + (int)gridDim.x < nwg) __syncthreads();
    }
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

static float bf2f_h(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2bf_h(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

template <bool AR, bool BR, int E>
static void launch16(const SmxGemmParams& p) {
    static bool once = [] { CK(hipFuncSetAttribute((const void*)w16_kernel<AR, BR, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W16_STAGE)); return true; }();
    (void)once;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    hipLaunchKernelGGL((w16_kernel<AR, BR, E>), dim3(tiles > 256 ? 256 : tiles), dim3(1024), 2 * W16_STAGE, 0, p);
}

int main(int argc, char** argv) {
    // {M, N, K, layout}: layout 0 = forward (A [M,K], W [N,K]), 1 = data gradient (A [M,K], W [K,N]), 2 = weight gradient (A [K,M], B [K,N], fp32 out)
    const int shapes[][4] = {{15968, 3072, 768, 0}, {15968, 768, 3072, 0}, {15968, 768, 768, 0}, {15968, 2304, 768, 0}, {16384, 4096, 1024, 0},
                             {7968, 768, 768, 0}, {511968, 512, 1536, 0}, {300, 200, 192, 0},
                             {15968, 768, 3072, 1}, {15968, 3072, 768, 1}, {15968, 768, 2304, 1}, {15968, 768, 768, 1}, {300, 200, 192, 1}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2], lay = s[3];
        bf16_t *A, *B, *C, *C2;
        float* bias;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMalloc(&C2, (size_t)M * N * 2)); CK(hipMalloc(&bias, (size_t)N * 4));
        {
            std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K);
            std::vector<float> hbias(N);
            unsigned x = 12345u;
            auto rnd = [&] { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xffff) / 65536.f - 0.5f; };
            for (auto& v : ha) v = f2bf_h(rnd());
            for (auto& v : hb) v = f2bf_h(rnd());
            for (auto& v : hbias) v = rnd();
            CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice));
        }
        const double fl = 2.0 * M * N * K;
        SmxGemmParams p = {};
        p.A = A; p.B = B; p.C = C2; p.bias = bias;
        p.c = SmxRowView{0, N, 0, 0, 0}; p.e = p.c;
        p.a = SmxRowView{0, K, 0, 0, 0};
        p.b = SmxRowView{0, lay == 1 ? N : K, 0, 0, 0};
        p.b_rc = lay == 1;
        p.M = M; p.N = N; p.K = K; p.nbatch = 1; p.split_k = 1; p.alpha = 1.f;
        p.tr_mode = 1;
        float t1 = time_us([&] { smx_gemm(&p, SMX_BF16, 0); });
        p.tr_mode = 8;
        float t8 = time_us([&] { smx_gemm(&p, SMX_BF16, 0); });
        p.tr_mode = 1;
        smx_gemm(&p, SMX_BF16, 0);
        SmxGemmParams p2 = p;
        p2.C = C;
        const bool aligned = !(N & 7);
        auto go = [&](const SmxGemmParams& pp) {
            if (lay == 0) { if (aligned) launch16<false, false, PP_EPI_LINEAR>(pp); else launch16<false, false, -1>(pp); }
            else { if (aligned) launch16<false, true, PP_EPI_LINEAR>(pp); else launch16<false, true, -1>(pp); }
        };
        float u0 = time_us([&] { go(p2); });
        SmxGemmParams p3 = p2;
        p3.drop_seed = 0xdead0002u;
        float u1 = time_us([&] { go(p3); });
        float v0 = -1.f, v1 = -1.f;
        bool w8ok = false;
        if (lay == 0 && aligned) {
            static bool once8 = [] { CK(hipFuncSetAttribute((const void*)w8_kernel<PP_EPI_LINEAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)); return true; }();
            (void)once8;
            const int t8n = ((M + 255) / 256) * ((N + 127) / 128);
            auto go8 = [&](const SmxGemmParams& pp) { hipLaunchKernelGGL((w8_kernel<PP_EPI_LINEAR>), dim3(t8n > 512 ? 512 : t8n), dim3(512), 65536, 0, pp); };
            v0 = time_us([&] { go8(p2); });
            v1 = time_us([&] { go8(p3); });
            CK(hipMemset(C, 0, (size_t)M * N * 2));
            go8(p2);
            CK(hipDeviceSynchronize());
            const size_t nc = (size_t)M * N < (size_t)1 << 24 ? (size_t)M * N : (size_t)1 << 24;
            std::vector<unsigned short> g1(nc), g2(nc);
            CK(hipMemcpy(g1.data(), C, nc * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(g2.data(), C2, nc * 2, hipMemcpyDeviceToHost));
            w8ok = memcmp(g1.data(), g2.data(), nc * 2) == 0;
            printf("   [8 waves x 2 workgroups/CU, 256x128 tile: %.1f us (%.0f TF), no epilogue %.1f us (%.0f TF; %.2f us per 256^2-equivalent K tile); %s]\n",
                   v0, fl / v0 / 1e6, v1, fl / v1 / 1e6, v1 / (((t8n + 511) / 512)) / ((K + 63) / 64), w8ok ? "bit-identical to the 128x128 kernel" : "MISMATCH");
        }
        go(p2);
        CK(hipDeviceSynchronize());
        const size_t ncmp = (size_t)M * N < (size_t)1 << 24 ? (size_t)M * N : (size_t)1 << 24;
        std::vector<unsigned short> h1(ncmp), h2(ncmp);
        CK(hipMemcpy(h1.data(), C, ncmp * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), C2, ncmp * 2, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < ncmp; ++i) {
            const double a = bf2f_h(h1[i]), b = bf2f_h(h2[i]);
            md = fmax(md, fabs(a - b)); mx = fmax(mx, fabs(b));
        }
        unsigned long long clk[4];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(w16_clk), sizeof(clk)));
        printf("   [shader clock during the 16-wave launch: %.0f MHz]\n", (double)clk[0] / ((double)clk[1] / 100.0));
        const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
        const int rounds = (tiles + 255) / 256;
        printf("M=%d N=%d K=%d layout %d: 128x128 %.1f us (%.0f TF) | pingpong %.1f us (%.0f TF) | 16-wave 256x256 (%d tiles, %d rounds): %.1f us (%.0f TF), "
               "no epilogue %.1f us (%.0f TF; %.2f us per K tile per round) | max diff %.3g of %.3g\n",
               M, N, K, lay, t1, fl / t1 / 1e6, t8, fl / t8 / 1e6, tiles, rounds, u0, fl / u0 / 1e6, u1, fl / u1 / 1e6,
               u1 / rounds / ((K + 63) / 64), md, mx);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2)); CK(hipFree(bias));
    }
    return 0;
}
