// Epilogue access patterns (round 5): how fast one CU ISSUES the 16-byte stores / loads of a 256 x 256 bf16 output tile, by how the
// 64 lanes of an instruction map to (row, 16-B chunk).  8 waves x 16 instructions x 1 KB = 128 KB per workgroup, one workgroup per CU,
// row stride `ld` bytes (an [M, 768] / [M, 3072] bf16 tensor: 1536 / 6144).
//   pat 0  lane & 15 -> row, lane >> 4 -> chunk        16 rows x 64 B   (what the MFMA accumulator layout gives: a quad of lanes = 4 rows)
//   pat 1  lane >> 2 -> row, lane & 3 -> chunk         16 rows x 64 B   (a quad of lanes = 64 contiguous bytes)
//   pat 2  lane >> 3 -> row, lane & 7 -> chunk          8 rows x 128 B  (8 lanes = one full line)
//   pat 3  (lane & 7) | rows interleaved: lane & 7 -> row, lane >> 3 -> chunk   8 rows x 128 B, lanes of a row 8 apart
//   pat 4  1 KB contiguous
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/st_lab tools/lab/st_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Res { unsigned long long issue, done; };
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int PAT, bool LOAD>
__global__ __launch_bounds__(512) void burst(char* dst, int ld, Res* out, int rounds, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ unsigned long long sh[2];
    if (tid < 2) sh[tid] = 0;
    __syncthreads();
    unsigned long long ti = 0, td = 0;
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        const int tpr = ld / 512, t = r * gridDim.x + blockIdx.x;
        char* tile = dst + (size_t)(t / tpr) * 256 * (size_t)ld + (t % tpr) * 512;
        // wave w owns rows (w >> 2) * 128 .. +128, byte columns (w & 3) * 128 .. +128; instruction j of 16 covers 1 KB of it
        char* wbase = tile + (size_t)((wave >> 2) * 128) * ld + (wave & 3) * 128;
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        u4 v = {(unsigned)tid, 1u, 2u, (unsigned)r};
        u4 x[16];          // (every load keeps its own registers until the wait: the data arrives asynchronously)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            char* p;
            if (PAT == 0) p = wbase + (size_t)((j >> 1) * 16 + (lane & 15)) * ld + (j & 1) * 64 + (lane >> 4) * 16;
            else if (PAT == 1) p = wbase + (size_t)((j >> 1) * 16 + (lane >> 2)) * ld + (j & 1) * 64 + (lane & 3) * 16;
            else if (PAT == 2) p = wbase + (size_t)(j * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
            else if (PAT == 3) p = wbase + (size_t)(j * 8 + (lane & 7)) * ld + (lane >> 3) * 16;
            else p = wbase + (size_t)(j * 8) * ld + lane * 16;          // (crosses rows: only the rate matters)
            if (LOAD) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(x[j]) : "v"(p) : "memory");
            } else {
                asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (LOAD) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { asm volatile("" : "+v"(x[j])); acc += x[j][0]; }
        }
        if (r == rounds - 1) { ti = t1 - t0; td = t2 - t0; }
    }
    if (LOAD && acc == 0x12345u) sink[0] = acc;
    if (lane == 0) { atomicMax(&sh[0], ti); atomicMax(&sh[1], td); }
    __syncthreads();
    if (tid == 0) { out[blockIdx.x].issue = sh[0]; out[blockIdx.x].done = sh[1]; }
}

template <int PAT, bool LOAD>
static int run(char* dst, int ld, Res* dres, unsigned* sink, int ncu, int rounds) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((burst<PAT, LOAD>), dim3(ncu), dim3(512), 0, 0, dst, ld, dres, rounds, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Res> h(ncu);
    CK(hipMemcpy(h.data(), dres, sizeof(Res) * ncu, hipMemcpyDeviceToHost));
    double si = 0, sd = 0;
    for (auto& r : h) { si += r.issue; sd += r.done; }
    fflush(stdout);
    printf("%s pat %d ld %5d %3d CUs rounds %d: issued after %6.0f cyc (%5.1f B/clk/CU) | complete after %6.0f cyc | kernel %.1f us\n", LOAD ? "load " : "store", PAT, ld, ncu, rounds,
           si / ncu, 131072.0 / (si / ncu), sd / ncu, ms * 1e3);
    return 0;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* buf;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
    Res* dres;
    CK(hipMalloc(&dres, sizeof(Res) * 256));
    unsigned* sink;
    CK(hipMalloc(&sink, 64));
    for (int ld : {1536, 6144})
        for (int ncu : {256, 1})
            for (int rounds : {1, 3}) {
#define ALL(L) run<0, L>(buf, ld, dres, sink, ncu, rounds); run<1, L>(buf, ld, dres, sink, ncu, rounds); run<2, L>(buf, ld, dres, sink, ncu, rounds); \
               run<3, L>(buf, ld, dres, sink, ncu, rounds); run<4, L>(buf, ld, dres, sink, ncu, rounds);
                ALL(false)
                ALL(true)
            }
    return 0;
}
