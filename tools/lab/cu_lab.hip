// CU laboratory (not part of the product library): what one CU's LDS and LDS-DMA path deliver, alone and together.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 cu_lab.hip -o cu_lab
// Every kernel runs one 160-KB workgroup per CU on all CUs and reports bytes per shader clock per CU (s_memtime) and
// per wall second (HIP events).  Roles per wave:  R = ds_read_b128 stream (conflict-free, 16 reads per wait),
// D = LDS-DMA stream (buffer_load_dwordx4 ... lds, `inflight` instructions kept outstanding), M = MFMA stream, - = idle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) int rsrc_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, %2 offen lds"
                 :: "s"(rsrc), "v"(voff), "s"(soff), "s"(lds_wave_base) : "memory");
}

struct Res { unsigned long long cyc; unsigned long long rd_bytes, dma_bytes, mfma; };

// roles: 2 bits per wave (0 idle, 1 R, 2 D, 3 M), wave w at bits 2w
// src_span: bytes of global memory each CU cycles through (small: L2 hits; large: MALL / HBM)
template <int INFLIGHT, int MF32, int RTR = 0>     // RTR 1: the R role issues ds_read_b64_tr_b16 (rows-contiguous fragment reads) instead of ds_read_b128
__global__ __launch_bounds__(1024) void cu_kernel(const char* src, unsigned roles, int budget, unsigned src_span, Res* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) const char* lds_cp_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = (roles >> (2 * wave)) & 3;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cp_t)smem);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long rd = 0, dm = 0, mf = 0;
    if (role == 1) {
        // 16 x ds_read_b128 per batch over a 16-KB window (lane-linear: conflict-free), windows rotate over 128 KB
        unsigned acc = 0;
        int it = 0;
        for (;; ++it) {
            if ((it & 3) == 3 && (long long)(__builtin_amdgcn_s_memtime() - t0) > budget) break;
            uint4 v[16];
            if (RTR == 0) {
                const unsigned base = lds0 + ((unsigned)(it + wave) & 7) * 16384u + lane * 16u;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "n"(j * 1024));
            } else {
                // the GEMM's rows-contiguous fragment pattern (gemm_common.h load_frag / rc_addr): lane (g, q, c) reads 8 B of
                // k-row 8 g + q, 16-column group swizzled by the row; 32 reads of 512 B per batch (same bytes as 16 x b128)
                const int g = lane >> 4, q = (lane >> 2) & 3, c = lane & 3, kb = 8 * g + q;
                const unsigned sw = (unsigned)((kb & 3) | (((kb >> 3) & 1) << 2));
                const unsigned win = lds0 + ((unsigned)(it + wave) & 7) * 16384u + kb * 256u + c * 8u;
                unsigned long long* w = reinterpret_cast<unsigned long long*>(v);
#pragma unroll
                for (int j = 0; j < 32; ++j) {           // (16-column group j & 7, k sub-step (j >> 3) & 1, half j >> 4) of a [64 k][128 col] tile
                    const unsigned addr = win + (unsigned)((((j >> 3) & 1) * 32 + (j >> 4) * 4) * 256) + ((((unsigned)j & 7u) ^ sw) << 5);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(w[j]) : "v"(addr));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) acc ^= v[j].x;
        }
        rd = (unsigned long long)it * 16 * 1024;
        if (acc == 0x12345678u) out[0].mfma = acc;
    } else if (role == 2) {
        const rsrc_t rs = make_rsrc(src + (size_t)blockIdx.x * src_span);
        const unsigned mask = src_span - 1;      // power of two
        unsigned off = (unsigned)wave * 8192u;
        int it = 0;
        for (;; ++it) {
            if ((it & 3) == 3 && (long long)(__builtin_amdgcn_s_memtime() - t0) > budget) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dma16(rs, ((off + j * 1024u) & mask) + lane * 16u, 0, lds0 + 131072u + ((unsigned)wave & 1) * 8192u + j * 1024u);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT) : "memory");
            }
            off += 16u * 8192u;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dm = (unsigned long long)it * 8 * 1024;
    } else if (role == 3) {
        bf16x8_t x = {1, 2, 3, 4, 5, 6, 7, 8}, y = {8, 7, 6, 5, 4, 3, 2, 1};
        asm volatile("" : "+v"(x), "+v"(y));
        int it = 0;
        if constexpr (MF32 != 0) {
            f32x16_t a0 = {}, a1 = {}, a2 = {};
            for (;; ++it) {
                if ((it & 3) == 3 && (long long)(__builtin_amdgcn_s_memtime() - t0) > budget) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
                }
            }
            mf = (unsigned long long)it * 12 * 32;      // pipe cycles at 32 per instruction
            if (a0[0] + a1[0] + a2[0] == 1234.5f) out[0].mfma = 1;
        } else {
            f32x4_t a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};
            for (;; ++it) {
                if ((it & 3) == 3 && (long long)(__builtin_amdgcn_s_memtime() - t0) > budget) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a4) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a5) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a6) : "v"(x), "v"(y));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a7) : "v"(x), "v"(y));
                }
            }
            mf = (unsigned long long)it * 32 * 16;
            if (a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0] == 1234.5f) out[0].mfma = 1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // per-role maximum duration and byte totals of this CU
    __shared__ unsigned long long sh[4];
    if (tid < 4) sh[tid] = 0;
    __syncthreads();
    if (lane == 0) {
        atomicMax(&sh[0], t1 - t0);
        atomicAdd(&sh[1], rd);
        atomicAdd(&sh[2], dm);
        atomicAdd(&sh[3], mf);
    }
    __syncthreads();
    if (tid == 0) { out[blockIdx.x].cyc = sh[0]; out[blockIdx.x].rd_bytes = sh[1]; out[blockIdx.x].dma_bytes = sh[2]; out[blockIdx.x].mfma = sh[3]; }
}


// Store burst: every wave of the workgroup stores NST x 1 KB (16 rows x 64 B per instruction, row stride ld bytes) of a
// 256 x 256 bf16 tile (8 waves x 16 instructions = 128 KB per workgroup), as a GEMM epilogue does.  Reports per CU: cycles until
// the last store was ISSUED and until all were acknowledged (vmcnt(0)); `active_mod` / `active_rem` select the workgroups
// that store (block b runs on XCD b % 8): 1/0 = all, 8/0 = XCD 0 only.
struct StRes { unsigned long long issue, done; };
__global__ __launch_bounds__(512) void st_kernel(char* dst, int ld, int active_mod, int active_rem, StRes* out, int rounds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned long long sh[2];
    if (tid < 2) sh[tid] = 0;
    __syncthreads();
    const bool act = (int)(blockIdx.x % active_mod) == active_rem;
    unsigned long long ti = 0, td = 0;
    for (int r = 0; r < rounds; ++r) {
        // tile (blockIdx, round): 256 rows x 512 B; wave w owns rows (w >> 2) * 128 .. +128, byte columns (w & 3) * 128 .. +128
        const int tpr = ld / 512, t = r * gridDim.x + blockIdx.x;
        char* tile = dst + (size_t)(t / tpr) * 256 * (size_t)ld + (t % tpr) * 512;
        char* p = tile + (size_t)((wave >> 2) * 128 + (lane & 15)) * ld + (wave & 3) * 128 + (lane >> 4) * 16;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (act) {
            typedef __attribute__((ext_vector_type(4))) unsigned u4;
            const u4 v = {(unsigned)tid, 1u, 2u, (unsigned)r};
#pragma unroll
            for (int j = 0; j < 16; ++j)
                asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p + (size_t)((j >> 1) * 16) * ld + (j & 1) * 64), "v"(v) : "memory");
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (r == rounds - 1) { ti = t1 - t0; td = t2 - t0; }
    }
    if (lane == 0) { atomicMax(&sh[0], ti); atomicMax(&sh[1], td); }
    __syncthreads();
    if (tid == 0) { out[blockIdx.x].issue = sh[0]; out[blockIdx.x].done = sh[1]; }
}

static void run_st(char* dst, int ld, int mod, int rem, StRes* dres, int rounds, const char* what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(st_kernel, dim3(256), dim3(512), 0, 0, dst, ld, mod, rem, dres, rounds);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<StRes> h(256);
    CK(hipMemcpy(h.data(), dres, sizeof(StRes) * 256, hipMemcpyDeviceToHost));
    double si = 0, sd = 0, mi = 0, md = 0; int n = 0;
    for (int b = 0; b < 256; ++b) if (b % mod == rem) { si += h[b].issue; sd += h[b].done; mi = fmax(mi, (double)h[b].issue); md = fmax(md, (double)h[b].done); ++n; }
    printf("store burst %-22s ld %6d rounds %d: %3d CUs x 128 KB | issued after avg %7.0f max %7.0f cyc | acknowledged after avg %7.0f max %7.0f cyc | kernel %.1f us\n",
           what, ld, rounds, n, si / n, mi, sd / n, md, ms * 1e3);
}

static unsigned parse_roles(const char* s, int& nw) {
    unsigned r = 0;
    nw = (int)strlen(s);
    for (int w = 0; w < nw; ++w) r |= (unsigned)(s[w] == 'R' ? 1 : s[w] == 'D' ? 2 : s[w] == 'M' ? 3 : 0) << (2 * w);
    return r;
}

template <int INFLIGHT, int MF32 = 0>
static void run(const char* roles_s, int iters_r, int iters_d, int iters_m, unsigned span, const char* src, Res* dres, int ncu, int mfma32 = MF32, int rtr = 0) {
    int nw;
    const unsigned roles = parse_roles(roles_s, nw);
    // one iteration count per launch: pick per dominant role so that all roles run for a similar time (fixed by hand below)
    (void)iters_d; (void)iters_m;
    CK(hipFuncSetAttribute((const void*)cu_kernel<INFLIGHT, MF32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (rtr) hipLaunchKernelGGL((cu_kernel<INFLIGHT, MF32, 1>), dim3(ncu), dim3(64 * nw), 160 * 1024 - 64, 0, src, roles, iters_r, span, dres);
        else hipLaunchKernelGGL((cu_kernel<INFLIGHT, MF32>), dim3(ncu), dim3(64 * nw), 160 * 1024 - 64, 0, src, roles, iters_r, span, dres);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Res> h(ncu);
    CK(hipMemcpy(h.data(), dres, sizeof(Res) * ncu, hipMemcpyDeviceToHost));
    double cyc = 0, rd = 0, dm = 0, mf = 0;
    for (auto& r : h) { cyc += (double)r.cyc; rd += (double)r.rd_bytes; dm += (double)r.dma_bytes; mf += (double)r.mfma; }
    cyc /= ncu;
    if (rtr) printf("[R = ds_read_b64_tr_b16] ");
    printf("%-18s inflight %2d span %8u KB mfma%d: %8.0f cyc (%.1f us, %.2f GHz) | read %6.1f B/clk/CU %6.1f TB/s | dma %6.1f B/clk/CU %6.2f TB/s | mfma pipe %.2f of 4 SIMDs\n",
           roles_s, INFLIGHT, span >> 10, mfma32 ? 32 : 16, cyc, ms * 1e3, cyc / (ms * 1e3) * 1e-3, rd / ncu / cyc, rd / (ms * 1e-3) * 1e-12, dm / ncu / cyc,
           dm / (ms * 1e-3) * 1e-12, mf / ncu / cyc);
}

int main(int argc, char** argv) {
    const bool only_tr = argc > 1 && std::string(argv[1]) == "tr";
    int ncu = 256;
    const size_t bytes = (size_t)1 << 30;
    char* src;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    Res* dres;
    CK(hipMalloc(&dres, sizeof(Res) * 1024));

    // 0. store bursts (a GEMM epilogue's 128 KB per CU): issue vs acknowledge, all CUs / one XCD / one CU
    if (!only_tr) {
        StRes* sres;
        CK(hipMalloc(&sres, sizeof(StRes) * 256));
        for (int ld : {1536, 6144}) {
            for (int rounds : {1, 4}) {
                run_st(src, ld, 1, 0, sres, rounds, "all CUs");
                run_st(src, ld, 8, 0, sres, rounds, "XCD 0 only (32 CUs)");
                run_st(src, ld, 256, 0, sres, rounds, "one CU");
                run_st(src, ld, 2, 0, sres, rounds, "even XCDs (128 CUs)");
            }
        }
    }
    const int IT = 400000;          // shader cycles per launch
    const unsigned L2S = 64u << 10, BIG = 4u << 20;          // per-CU source span: 64 KB (L2-resident) / 4 MB (1 GB chip-wide: HBM)
    // 5. the rows-contiguous operands' transposing reads (8 B per lane): alone, beside MFMA streams and LDS-DMA
    for (const char* r : {"R", "RRRR", "RRRRRRRR", "RRRRRRRRRRRRRRRR"}) run<8>(r, IT, 0, 0, L2S, src, dres, ncu, 0, 1);
    for (const char* r : {"MMMMRRRR", "MMMMMMMMRRRR", "MMMMRRRRRRRRDDDD", "MMMMMMMMRRRRDDDD"}) run<7, 0>(r, IT, 0, 0, L2S, src, dres, ncu, 0, 1);
    if (only_tr) return 0;
    // 1. LDS reads alone
    for (const char* r : {"R", "RRRR", "RRRRRRRR", "RRRRRRRRRRRRRRRR"}) run<8>(r, IT, 0, 0, L2S, src, dres, ncu, 0);
    // 2. DMA alone: waves, in-flight depth, source
    for (const char* r : {"D", "DD", "DDDD", "DDDDDDDD"}) {
        run<0>(r, IT, 0, 0, L2S, src, dres, ncu, 0);
        run<2>(r, IT, 0, 0, L2S, src, dres, ncu, 0);
        run<4>(r, IT, 0, 0, L2S, src, dres, ncu, 0);
        run<7>(r, IT, 0, 0, L2S, src, dres, ncu, 0);
        run<7>(r, IT, 0, 0, BIG, src, dres, ncu, 0);
    }
    // 3. both: readers and DMA waves together (same iteration count: compare each rate with its solo run)
    for (const char* r : {"RRRRDDDD", "RRRRRRRRDDDD", "RRRRRRRRD", "RRRRRRRRDD", "RRRRDD"}) {
        run<7>(r, IT, 0, 0, L2S, src, dres, ncu, 0);
        run<7>(r, IT, 0, 0, BIG, src, dres, ncu, 0);
    }
    // 4. MFMA alone and beside readers / DMA
    for (const char* r : {"MMMM", "MMMMMMMM", "MMMMRRRR", "MMMMMMMMRRRR", "MMMMDDDD", "MMMMMMMMDDDD", "MMMMMMMMRRRRDDDD", "MMMMRRRRRRRRDDDD"}) {
        run<7, 0>(r, IT, 0, 0, L2S, src, dres, ncu);
        run<7, 1>(r, IT, 0, 0, L2S, src, dres, ncu);
    }
    return 0;
}
