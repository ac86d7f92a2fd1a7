// Small HBM-bound kernels around the GEMMs: dtype casts, conv-weight re-layouts, token embedding
// gather / scatter, bias-gradient column sums, fused cross-entropy (+argmax, +dlogits), SpecAugment row
// masking, element-wise adds.  All fp32 statistics; 16-B accesses where layouts allow.
#include <cstdlib>
#include "smx_common.h"
#include <algorithm>

// ---------------------------------------------------------------- cast fp32 -> T
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long stride = (long long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += stride) {
        float v[8];
        load8(src + i, v);
        store8(dst + i, v);
    }
    if (i < n && i + 8 > n)
        for (long long j = i; j < n; ++j) Cvt<T>::st(dst + j, src[j]);
}
extern "C" int smx_cast_from_f32(const float* src, void* dst, long long n, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (n <= 0) return SMX_OK;
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, n);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks), dim3(256), 0, stream, src, (float*)dst, n);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// T -> fp32 (outputs handed back to PyTorch callers: raw logits, hidden states)
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = Cvt<T>::ld(src + i);
}
extern "C" int smx_cast_to_f32(const void* src, float* dst, long long n, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (n <= 0) return SMX_OK;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)src, dst, n);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)src, dst, n);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- conv weight re-layout
// nn.Conv1d weight [Co, Ci, k] fp32  ->  tap-major GEMM operand [Co, k*Ci] (dtype T): the K index of the
// GEMM then walks a contiguous channels-last input window (TF:...wav2vec2.py:254-273 conv as GEMM).
template <typename T>
__global__ void pack_conv_w_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int k) {
    const long long n = (long long)Co * Ci * k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ci = i % Ci, t = (i / Ci) % k, co = i / ((long long)Ci * k);
        Cvt<T>::st(out + i, w[((long long)co * Ci + ci) * k + t]);
    }
}
extern "C" int smx_pack_conv_w(const float* w, void* out, int Co, int Ci, int k, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    const long long n = (long long)Co * Ci * k;
    int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(pack_conv_w_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, w, (bf16_t*)out, Co, Ci, k);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(pack_conv_w_kernel<float>, dim3(blocks), dim3(256), 0, stream, w, (float*)out, Co, Ci, k);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}
// Data-gradient operands of a strided Conv1d (round 4).  The gradient wrt input position t = u s + r (residue r) sums the taps
// j s + r: dX[t, ci] = sum_j sum_co dPre[u - j, co] w[co, ci, j s + r] - a GEMM whose A rows are the contiguous runs
// dPre[u - (nj - 1) .. u] (nj = taps of that residue) and whose B operand, K-CONTIGUOUS, is
//     Wd_r[ci, c Co + co] = w[co, ci, r + (nj - 1 - c) s],   c = 0 .. nj - 1.
// out = the s matrices [Ci, nj_r Co] back to back (residue r starts at Ci Co x taps of residues < r).  With them the data
// gradients run in the forward's (K-contiguous, K-contiguous) layout on the 256-wide kernels instead of reading the
// tap-major forward weight rows-contiguous through a batched view.
template <typename T>
__global__ void pack_conv_w_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int k, int s) {
    const long long n = (long long)Co * Ci * k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int t = i % k, ci = (i / k) % Ci, co = i / ((long long)Ci * k);
        const int r = t % s, j = t / s;
        const int nj = (k - r + s - 1) / s;
        int before = 0;                       // taps of the residues below r
        for (int q = 0; q < r; ++q) before += (k - q + s - 1) / s;
        const long long dst = (long long)Ci * Co * before + ((long long)ci * nj + (nj - 1 - j)) * Co + co;
        Cvt<T>::st(out + dst, w[i]);
    }
}
extern "C" int smx_pack_conv_w_dgrad(const float* w, void* out, int Co, int Ci, int k, int s, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (!w || !out || Co <= 0 || Ci <= 0 || k <= 0 || s <= 0 || s > k) return SMX_EINVAL;
    const long long n = (long long)Co * Ci * k;
    int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(pack_conv_w_dgrad_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, w, (bf16_t*)out, Co, Ci, k, s);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(pack_conv_w_dgrad_kernel<float>, dim3(blocks), dim3(256), 0, stream, w, (float*)out, Co, Ci, k, s);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}
// dW (tap-major fp32 [Co, k*Ci]) accumulated into the parameter-layout gradient [Co, Ci, k]
__global__ void unpack_conv_dw_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci, int k) {
    const long long n = (long long)Co * Ci * k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ci = i % Ci, t = (i / Ci) % k, co = i / ((long long)Ci * k);
        dw[((long long)co * Ci + ci) * k + t] += dwp[i];
    }
}
extern "C" int smx_unpack_conv_dw(const float* dwp, float* dw, int Co, int Ci, int k, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    const long long n = (long long)Co * Ci * k;
    int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(unpack_conv_dw_kernel, dim3(blocks), dim3(256), 0, stream, dwp, dw, Co, Ci, k);
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- token embedding
// out[r,:] = table[ids[r],:] * scale   (TF:models/bart/modeling_bart.py:101-113)
template <typename T>
__global__ void embed_fwd_kernel(const long long* __restrict__ ids, const T* __restrict__ table, T* __restrict__ out,
                                 int M, int D, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const T* src = table + ids[row] * (long long)D;
    T* dst = out + (long long)row * D;
    for (int c = lane * 8; c < D; c += 512) {
        float v[8];
        load8(src + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= scale;
        store8(dst + c, v);
    }
}
extern "C" int smx_embed_fwd(const long long* ids, const void* table, void* out, int M, int D, float scale, int dtype,
                             hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (M <= 0 || (D & 7)) return SMX_EINVAL;
    dim3 grid((M + 3) / 4);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, grid, dim3(256), 0, stream, ids, (const bf16_t*)table, (bf16_t*)out, M, D, scale);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(embed_fwd_kernel<float>, grid, dim3(256), 0, stream, ids, (const float*)table, (float*)out, M, D, scale);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}
template <typename T>
__global__ void embed_bwd_kernel(const long long* __restrict__ ids, const T* __restrict__ dy, float* __restrict__ dtable,
                                 int M, int D, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float* dst = dtable + ids[row] * (long long)D;
    const T* src = dy + (long long)row * D;
    for (int c = lane * 8; c < D; c += 512) {
        float v[8];
        load8(src + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(dst + c + e, v[e] * scale);
    }
}
extern "C" int smx_embed_bwd(const long long* ids, const void* dy, float* dtable, int M, int D, float scale, int dtype,
                             hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (M <= 0 || (D & 7)) return SMX_EINVAL;
    dim3 grid((M + 3) / 4);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(256), 0, stream, ids, (const bf16_t*)dy, dtable, M, D, scale);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), 0, stream, ids, (const float*)dy, dtable, M, D, scale);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- column sums (bias gradients)
// out[n] += alpha * sum_m x[m*ld + n]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int M, int N,
                                                     long long ld, float alpha) {
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < N) {
        const int step = gridDim.y * 4;
        if (c + 8 <= N) {
            int m = blockIdx.y * 4 + w;
            for (; m + 3 * step < M; m += 4 * step) {      // 4 independent 16-B loads in flight per lane
                float v0[8], v1[8], v2[8], v3[8];
                load8(x + (long long)m * ld + c, v0);
                load8(x + (long long)(m + step) * ld + c, v1);
                load8(x + (long long)(m + 2 * step) * ld + c, v2);
                load8(x + (long long)(m + 3 * step) * ld + c, v3);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
            }
            for (; m < M; m += step) {
                float v[8];
                load8(x + (long long)m * ld + c, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        } else {
            for (int m = blockIdx.y * 4 + w; m < M; m += step) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += c + e < N ? Cvt<T>::ld(x + (long long)m * ld + c + e) : 0.f;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w][lane][e] = acc[e];
    __syncthreads();
    if (w == 0 && c < N) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (c + e < N) atomicAdd(out + c + e, alpha * (red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e]));
    }
}
extern "C" int smx_colsum(const void* x, float* out, int M, int N, long long ld, float alpha, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (M <= 0 || N <= 0 || (ld & 7)) return SMX_EINVAL;
    const int gx = (N + 511) / 512;
    int gy = (M + 63) / 64;      // few blocks per column strip: the final fp32 atomics serialise in L2
    const int cap = max(1, 128 / gx);
    if (gy > cap) gy = cap;
    dim3 grid(gx, gy);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, out, M, N, ld, alpha);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, out, M, N, ld, alpha);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// Two-stage column sum for tall matrices (the bias gradients of the encoder: M = 16 k rows).  The atomic version above
// is limited by its handful of resident waves (the per-address fp32 atomics cap the number of row slices); here every
// row slice writes one partial row into a workspace (no atomics), so the grid can cover the chip with 8 x 16-B loads
// in flight per lane, and a second tiny launch folds the partial rows into out.
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, float* __restrict__ ws, int M, int N,
                                                          long long ld, int Np) {
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < N) {                                   // N % 8 == 0 on this path
        const int step = gridDim.y * 4;
        int m = blockIdx.y * 4 + w;
        for (; m + 7 * step < M; m += 8 * step) {
            float v[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) load8(x + (long long)(m + u * step) * ld + c, v[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc[e] += ((v[0][e] + v[1][e]) + (v[2][e] + v[3][e])) + ((v[4][e] + v[5][e]) + (v[6][e] + v[7][e]));
        }
        for (; m < M; m += step) {
            float v[8];
            load8(x + (long long)m * ld + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w][lane][e] = acc[e];
    __syncthreads();
    if (w == 0 && c < N) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
        store8(ws + (long long)blockIdx.y * Np + c, o);
    }
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ ws, float* __restrict__ out, int parts, int N,
                                                          int Np, float alpha) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float a = 0.f;
    if (c < N)
        for (int r = w; r < parts; r += 4) a += ws[(long long)r * Np + c];
    red[w][lane] = a;
    __syncthreads();
    if (w == 0 && c < N) out[c] += alpha * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}
// out[n] += alpha * sum_m x[m, n].  ws: >= smx_colsum_ws_floats(M, N) floats of scratch, or null (atomic single-stage form).
#ifndef SMX_COLSUM_MINM
#define SMX_COLSUM_MINM 4096
#endif
static int colsum_slices(int M, int N) {
    const int gx = (N + 511) / 512;
#ifdef SMX_COLSUM_GY
    int gy = SMX_COLSUM_GY;
#else
    // measured (tools/gpu_colsum_bench.py): 10^5-row inputs want ~4 workgroups per CU; at the encoder's 16 k rows more
    // than 64-128 row slices only make the fold pass longer
    int gy = M >= 131072 ? max(1, 1024 / gx) : (gx <= 2 ? 128 : 64);
#endif
    if (gy > (M + 31) / 32) gy = (M + 31) / 32;
    return gy;
}
extern "C" long long smx_colsum_ws_floats(int M, int N) { return (long long)colsum_slices(M, N) * ((N + 7) / 8 * 8); }
extern "C" int smx_colsum_ws(const void* x, float* out, int M, int N, long long ld, float alpha, int dtype, float* ws,
                             hipStream_t stream) {
    // short inputs: the second launch costs more than the atomics it removes
    if (!ws || (N & 7) || M < SMX_COLSUM_MINM) return smx_colsum(x, out, M, N, ld, alpha, dtype, stream);
    (void)hipGetLastError();
    if (M <= 0 || N <= 0 || (ld & 7)) return SMX_EINVAL;
    const int gx = (N + 511) / 512, Np = (N + 7) / 8 * 8;
    const int gy = colsum_slices(M, N);
    dim3 grid(gx, gy);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(colsum_part_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, ws, M, N, ld, Np);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(colsum_part_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, ws, M, N, ld, Np);
    else return SMX_EINVAL;
    // out == null: the caller folds the partial rows later (smx_fold_many): smx_colsum_slices(M, N) rows of ceil8(N) floats
    if (out) hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, ws, out, gy, N, Np, alpha);
    SMX_CHECK_LAUNCH();
}
extern "C" int smx_colsum_slices(int M, int N) { return colsum_slices(M, N); }
extern "C" int smx_colsum_min_rows(void) { return SMX_COLSUM_MINM; }

// Many second-stage column reductions in ONE launch: dst[c] += alpha * sum_r ws[r * ld + c] for every table entry.
// The two-stage reductions of backward (bias gradients, LayerNorm gamma / beta gradients) each used to end in their own
// 8-us launch of a few workgroups (~140 per step); the engine now queues them and folds a whole stage's worth at its end.
// Entries with many partial rows are cut into row slices that add with fp32 atomics (as the LayerNorm finaliser did).
#define SMX_FOLD_MAX 48
struct SmxFoldEntry {
    const float* ws;
    float* dst;
    int nrows, ncols;
    long long ld;
    float alpha;
    int pad;
};
struct SmxFoldTable {
    int n, pad;
    SmxFoldEntry e[SMX_FOLD_MAX];
};
#define FOLD_ROWS_PER_SLICE 64
__global__ __launch_bounds__(256) void fold_many_kernel(SmxFoldTable t) {
    __shared__ float red[4][64];
    const SmxFoldEntry& en = t.e[blockIdx.y];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.z * FOLD_ROWS_PER_SLICE, r1 = min(en.nrows, r0 + FOLD_ROWS_PER_SLICE);
    if (blockIdx.x * 64 >= en.ncols || r0 >= en.nrows) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < en.ncols) {
        const float* src = en.ws + c;
        int r = r0 + w;
        for (; r + 12 < r1; r += 16) {                 // four independent loads in flight per thread
            a0 += src[(long long)r * en.ld];
            a1 += src[(long long)(r + 4) * en.ld];
            a2 += src[(long long)(r + 8) * en.ld];
            a3 += src[(long long)(r + 12) * en.ld];
        }
        for (; r < r1; r += 4) a0 += src[(long long)r * en.ld];
    }
    red[w][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (w == 0 && c < en.ncols) {
        const float v = en.alpha * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        atomicAdd(en.dst + c, v);      // (one adder per address unless the entry has several row slices or two entries share a
                                       // destination: deterministic in the common case, never a lost update)
    }
}
extern "C" int smx_fold_many(const SmxFoldTable* tp, hipStream_t stream) {
    (void)hipGetLastError();
    if (!tp || tp->n < 0 || tp->n > SMX_FOLD_MAX) return SMX_EINVAL;
    if (tp->n == 0) return SMX_OK;
    int maxc = 0, maxr = 0;
    for (int i = 0; i < tp->n; ++i) {
        const SmxFoldEntry& e = tp->e[i];
        if (!e.ws || !e.dst || e.nrows <= 0 || e.ncols <= 0 || e.ld < e.ncols) return SMX_EINVAL;
        maxc = max(maxc, e.ncols);
        maxr = max(maxr, e.nrows);
    }
    dim3 grid((maxc + 63) / 64, tp->n, (maxr + FOLD_ROWS_PER_SLICE - 1) / FOLD_ROWS_PER_SLICE);
    hipLaunchKernelGGL(fold_many_kernel, grid, dim3(256), 0, stream, *tp);
    SMX_CHECK_LAUNCH();
}
extern "C" int smx_sizeof_SmxFoldTable(void) { return (int)sizeof(SmxFoldTable); }
// ---- batched 2-D transposes of 16-bit matrices (round 6) ----
// dst[cols][rows] = src[rows][cols] for up to SMX_TR_MAX matrices per launch (rows, cols multiples of 8; 16-B accesses on both sides).
// Used for the K-contiguous copies of the Linear weights that the data-gradient GEMMs read (engine.py `dgrad`): with W^T stored
// [in][out] the data gradient is a (KC, KC) launch - 16-byte fragment reads instead of two transposing 8-byte reads per fragment
// (K tile of the 256-wide kernels 1.18 -> 1.28 us with a rows-contiguous B, profiles/r05_fr_timeline.txt) - for one pass over the
// weights per optimizer step (85 MB of the 470 MB of config 2: ~40 us).
#define SMX_TR_MAX 64
struct SmxTrEntry {
    const unsigned short* src;
    unsigned short* dst;
    int rows, cols;
    int tile0, tcols;          // first tile of this matrix in the launch; tiles per row of tiles
};
struct SmxTrTable {
    int n, tiles;
    SmxTrEntry e[SMX_TR_MAX];
};
__global__ __launch_bounds__(256) void transpose_many_kernel(SmxTrTable t) {
    __shared__ unsigned short tile[64][72];          // (144-B rows: the 16-B row writes and the 2-B column reads both spread over the banks)
    int lo = 0, hi = t.n - 1;                        // last entry whose first tile is <= my tile
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.e[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const SmxTrEntry& en = t.e[lo];
    const int tl = blockIdx.x - en.tile0;
    const int r0 = (tl / en.tcols) * 64, c0 = (tl % en.tcols) * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int r = ps * 32 + (tid >> 3), c = (tid & 7) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + r < en.rows && c0 + c < en.cols) v = *reinterpret_cast<const uint4*>(en.src + (long long)(r0 + r) * en.cols + c0 + c);
        *reinterpret_cast<uint4*>(&tile[r][c]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int c = ps * 32 + (tid >> 3), r = (tid & 7) * 8;          // output row c0 + c, 8 consecutive source rows
        if (c0 + c < en.cols && r0 + r < en.rows) {
            union { uint4 v; unsigned short s[8]; } o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.s[e] = tile[r + e][c];
            *reinterpret_cast<uint4*>(en.dst + (long long)(c0 + c) * en.rows + r0 + r) = o.v;
        }
    }
}
extern "C" int smx_transpose_many(const SmxTrTable* tp, hipStream_t stream) {
    (void)hipGetLastError();
    if (!tp || tp->n < 0 || tp->n > SMX_TR_MAX) return SMX_EINVAL;
    if (tp->n == 0) return SMX_OK;
    SmxTrTable t = *tp;
    int tiles = 0;
    for (int i = 0; i < t.n; ++i) {
        SmxTrEntry& e = t.e[i];
        if (!e.src || !e.dst || e.rows <= 0 || e.cols <= 0 || (e.rows & 7) || (e.cols & 7) || ((size_t)e.src & 15) || ((size_t)e.dst & 15)) return SMX_EINVAL;
        e.tile0 = tiles;
        e.tcols = (e.cols + 63) / 64;
        tiles += e.tcols * ((e.rows + 63) / 64);
    }
    t.tiles = tiles;
    hipLaunchKernelGGL(transpose_many_kernel, dim3(tiles), dim3(256), 0, stream, t);
    SMX_CHECK_LAUNCH();
}
// x[i] = T(float(x[i]) * *scale) in place: the seed of backward scaled by the DEVICE scalar autograd hands to `loss.backward()`
// (1 / k under gradient accumulation) in fp32 - one rounding per element, no host read of the scalar (model.py _StepFn.backward)
template <typename T>
__global__ __launch_bounds__(256) void scale_dev_kernel(T* __restrict__ x, long long n, const float* __restrict__ scale) {
    const float s = *scale;
    const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i0 + 8 <= n) {
        float v[8];
        load8(x + i0, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= s;
        store8(x + i0, v);
    } else {
        for (long long i = i0; i < n; ++i) Cvt<T>::st(x + i, Cvt<T>::ld(x + i) * s);
    }
}
extern "C" int smx_scale_dev(void* x, long long n, const float* scale, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (!x || !scale || n < 0 || ((size_t)x & 15)) return SMX_EINVAL;
    if (n == 0) return SMX_OK;
    const unsigned grid = (unsigned)((n + 2047) / 2048);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(scale_dev_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (bf16_t*)x, n, scale);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(scale_dev_kernel<float>, dim3(grid), dim3(256), 0, stream, (float*)x, n, scale);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}
extern "C" int smx_sizeof_SmxTrTable(void) { return (int)sizeof(SmxTrTable); }
extern "C" int smx_tr_max(void) { return SMX_TR_MAX; }

extern "C" int smx_fold_max(void) { return SMX_FOLD_MAX; }

// ---------------------------------------------------------------- cross entropy over the vocabulary
// CrossEntropyLoss(ignore_index=-100, mean)  (TF:models/bart/modeling_bart.py:942-946) fused with
// argmax (ref:speechmix/model.py:174) and the logits gradient.  One block per token row.
struct SmxCEParams {
    const float* logits;       // [M, ldl] fp32
    const long long* labels;   // [M] or null (no loss: argmax only)
    float* loss;               // scalar, atomically accumulated: sum_rows loss_row / n_valid
    long long* argmax;         // [M] or null
    void* dlogits;             // [M, ldd] dtype T (pad columns zeroed) or null
    float* lse;                // [M] optional: log-sum-exp per row
    int M, V;
    long long ldl, ldd;
    float gscale;              // multiplies dlogits (upstream gradient)
    // SpeechMixSelf (ref:speechmix/model.py:257-259): KLDivLoss(batchmean)(log_softmax(logits), softmax(logits_t))
    const float* logits_t;     // teacher logits [M, ldl] or null
    float* kld;                // scalar, atomically accumulated (already divided by the batch size)
    float kld_scale;           // 1 / batch size
    // Row-chunked use (the LM head streamed over row chunks so that [B L, V] logits are never materialised, Engine.lm_losses):
    // the mean's denominator is the number of valid labels of the WHOLE batch - count over count_labels[0 .. count_M) when set
    const long long* count_labels;
    int count_M;
};
template <typename T>
__global__ __launch_bounds__(256) void ce_kernel(SmxCEParams p) {
    __shared__ float sh[16];
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* z = p.logits + (long long)row * p.ldl;
    float nvalid = 0.f;
    if (p.labels) {
        const long long* cl = p.count_labels ? p.count_labels : p.labels;
        const int cm = p.count_labels ? p.count_M : p.M;
        for (int i = tid; i < cm; i += 256) nvalid += cl[i] != -100 ? 1.f : 0.f;
        nvalid = block_sum(nvalid, sh);
    }
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int j = tid; j < p.V; j += 256) {
        const float v = z[j];
        if (v > mx) { mx = v; mi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(mi, o, 64);
        if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { shv[tid >> 6] = mx; shi[tid >> 6] = mi; }
    __syncthreads();
    mx = shv[0]; mi = shi[0];
    for (int w = 1; w < 4; ++w)
        if (shv[w] > mx || (shv[w] == mx && shi[w] < mi)) { mx = shv[w]; mi = shi[w]; }
    if (p.argmax && tid == 0) p.argmax[row] = mi;
    if (!p.labels && !p.lse) return;
    float se = 0.f;
    for (int j = tid; j < p.V; j += 256) se += __expf(z[j] - mx);
    se = block_sum(se, sh);
    const float lse = mx + __logf(se);
    if (p.lse && tid == 0) p.lse[row] = lse;
    if (!p.labels) return;
    const long long lab = p.labels[row];
    const bool valid = lab != -100;
    if (valid && tid == 0) atomicAdd(p.loss, (lse - z[lab]) / nvalid);
    float lse_t = 0.f;
    const float* zt = p.logits_t ? p.logits_t + (long long)row * p.ldl : nullptr;
    if (zt) {
        float mt = -INFINITY;
        for (int j = tid; j < p.V; j += 256) mt = fmaxf(mt, zt[j]);
        mt = block_max(mt, sh);
        float st = 0.f;
        for (int j = tid; j < p.V; j += 256) st += __expf(zt[j] - mt);
        st = block_sum(st, sh);
        lse_t = mt + __logf(st);
        float kl = 0.f;
        for (int j = tid; j < p.V; j += 256) {
            const float lpt = zt[j] - lse_t;
            kl += __expf(lpt) * (lpt - (z[j] - lse));
        }
        kl = block_sum(kl, sh);
        if (tid == 0 && p.kld) atomicAdd(p.kld, kl * p.kld_scale);
    }
    if (p.dlogits) {
        T* d = reinterpret_cast<T*>(p.dlogits) + (long long)row * p.ldd;
        const float coef = valid ? p.gscale / nvalid : 0.f;
        const float kcoef = p.gscale * p.kld_scale;
        for (int j = tid; j < p.ldd; j += 256) {
            float g = 0.f;
            if (j < p.V) {
                const float ps = __expf(z[j] - lse);
                if (valid) g = (ps - (j == lab ? 1.f : 0.f)) * coef;
                if (zt) g += (ps - __expf(zt[j] - lse_t)) * kcoef;
            }
            Cvt<T>::st(d + j, g);
        }
    }
}
// Same results for the plain loss (labels, no teacher logits) in two passes over the row instead of three, 16-B accesses:
// pass 1 keeps a running (max, sum of exp, first arg max) per thread and merges them across the block, pass 2 writes the
// gradient.  The 200-KB fp32 row of the BART vocabulary stays in L2 between the passes.
struct CeRun {
    float m, s;
    int i;
};
__device__ __forceinline__ void ce_merge(CeRun& a, float om, float os, int oi) {
    if (om == -INFINITY) return;
    if (a.m == -INFINITY) { a.m = om; a.s = os; a.i = oi; return; }
    const float M = fmaxf(a.m, om);
    a.s = a.s * __expf(a.m - M) + os * __expf(om - M);
    if (om > a.m || (om == a.m && oi < a.i)) a.i = oi;
    a.m = M;
}
__device__ __forceinline__ void ce_push(CeRun& a, float v, int idx) {
    if (v > a.m) {                      // strictly greater: the first index of the maximum is kept
        a.s = a.s * __expf(a.m - v) + 1.f;        // (exp(-inf) = 0 on the first element)
        a.m = v;
        a.i = idx;
    } else {
        a.s += __expf(v - a.m);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_fast_kernel(SmxCEParams p) {
    __shared__ float sh[16];
    __shared__ float shm[4], shs[4];
    __shared__ int shi[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* z = p.logits + (long long)row * p.ldl;
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const int nv4 = p.V >> 2;
    float nvalid = 0.f;
    const long long* cl = p.count_labels ? p.count_labels : p.labels;
    const int cm = p.count_labels ? p.count_M : p.M;
    for (int i = tid; i < cm; i += 256) nvalid += cl[i] != -100 ? 1.f : 0.f;
    nvalid = block_sum(nvalid, sh);
    CeRun a = {-INFINITY, 0.f, 0x7fffffff};
    for (int j4 = tid; j4 < nv4; j4 += 256) {
        const float4 v = z4[j4];
        const float m4 = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
        if (m4 > a.m) {
            const int k = v.x == m4 ? 0 : v.y == m4 ? 1 : v.z == m4 ? 2 : 3;
            a.s *= __expf(a.m - m4);
            a.m = m4;
            a.i = j4 * 4 + k;
        }
        a.s += (__expf(v.x - a.m) + __expf(v.y - a.m)) + (__expf(v.z - a.m) + __expf(v.w - a.m));
    }
    if (tid == 0)
        for (int j = nv4 * 4; j < p.V; ++j) ce_push(a, z[j], j);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(a.m, o, 64), os = __shfl_xor(a.s, o, 64);
        const int oi = __shfl_xor(a.i, o, 64);
        ce_merge(a, om, os, oi);
    }
    if (lane == 0) { shm[w] = a.m; shs[w] = a.s; shi[w] = a.i; }
    __syncthreads();
    a.m = shm[0]; a.s = shs[0]; a.i = shi[0];
    for (int k = 1; k < 4; ++k) ce_merge(a, shm[k], shs[k], shi[k]);
    if (p.argmax && tid == 0) p.argmax[row] = a.i;
    const float lse = a.m + __logf(a.s);
    if (p.lse && tid == 0) p.lse[row] = lse;
    const long long lab = p.labels[row];
    const bool valid = lab != -100;
    if (valid && tid == 0) atomicAdd(p.loss, (lse - z[lab]) / nvalid);
    if (!p.dlogits) return;
    T* d = reinterpret_cast<T*>(p.dlogits) + (long long)row * p.ldd;
    const float coef = valid ? p.gscale / nvalid : 0.f;
    const int lab4 = valid ? (int)(lab >> 2) : -1, labk = (int)(lab & 3);
    for (int j4 = tid; j4 < nv4; j4 += 256) {
        const float4 v = z4[j4];
        float g[4] = {__expf(v.x - lse) * coef, __expf(v.y - lse) * coef, __expf(v.z - lse) * coef, __expf(v.w - lse) * coef};
        if (j4 == lab4) g[labk] -= coef;
        if (sizeof(T) == 2) *reinterpret_cast<uint2*>(d + j4 * 4) = make_uint2(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]));
        else *reinterpret_cast<float4*>(d + j4 * 4) = make_float4(g[0], g[1], g[2], g[3]);
    }
    for (int j = nv4 * 4 + tid; j < p.ldd; j += 256) {          // ragged end of the vocabulary, then the zeroed pad columns
        float g = 0.f;
        if (j < p.V) g = (__expf(z[j] - lse) - ((valid && j == lab) ? 1.f : 0.f)) * coef;
        Cvt<T>::st(d + j, g);
    }
}
extern "C" int smx_cross_entropy(const SmxCEParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxCEParams p = *pp;
    if (p.M <= 0 || p.V <= 0 || p.ldl < p.V) return SMX_EINVAL;
    if (p.dlogits && p.ldd < p.V) return SMX_EINVAL;
    if (dtype != SMX_BF16 && dtype != SMX_F32) return SMX_EINVAL;
    // plain loss on 16-B aligned rows: the two-pass kernel (SMX_CE_FAST=0: off)
    static const bool fast_ok = !(getenv("SMX_CE_FAST") && getenv("SMX_CE_FAST")[0] == '0');
    const bool fast = fast_ok && p.labels && p.loss && !p.logits_t && !(p.ldl & 3) && !((size_t)p.logits & 15) &&
                      (!p.dlogits || (!(p.ldd & 3) && !((size_t)p.dlogits & 15)));
    if (fast) {
        if (dtype == SMX_BF16) hipLaunchKernelGGL(ce_fast_kernel<bf16_t>, dim3(p.M), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(ce_fast_kernel<float>, dim3(p.M), dim3(256), 0, stream, p);
        SMX_CHECK_LAUNCH();
    }
    if (dtype == SMX_BF16) hipLaunchKernelGGL(ce_kernel<bf16_t>, dim3(p.M), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ce_kernel<float>, dim3(p.M), dim3(256), 0, stream, p);
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- element-wise: out = a + b
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long stride = (long long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += stride) {
        float x[8], y[8];
        load8(a + i, x);
        load8(b + i, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        store8(out + i, x);
    }
}
extern "C" int smx_add(const void* a, const void* b, void* out, long long n, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (n <= 0 || (n & 7)) return SMX_EINVAL;
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)a, (const float*)b, (float*)out, n);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- zeroing of scattered ranges of the flat gradient
// one block per table row (offset, count <= 65536): 16-B stores over the aligned interior, scalar head / tail
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* base, const long long* table) {
    const long long off = table[2 * blockIdx.x], cnt = table[2 * blockIdx.x + 1];
    float* p = base + off;
    const long long head = min(cnt, (long long)((4 - (off & 3)) & 3));
    if ((long long)threadIdx.x < head) p[threadIdx.x] = 0.f;
    const long long nvec = (cnt - head) >> 2;
    float4* v = reinterpret_cast<float4*>(p + head);
    for (long long i = threadIdx.x; i < nvec; i += 256) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long done = head + (nvec << 2);
    if (done + threadIdx.x < cnt) p[done + threadIdx.x] = 0.f;
}
extern "C" int smx_zero_ranges(float* base, const long long* table, int n, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0) return 0;
    if (!base || !table || ((size_t)base & 15)) return SMX_EINVAL;
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(n), dim3(256), 0, stream, base, table);
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- SpecAugment time masking
// rows listed in `rows` are overwritten with the learned masked_spec_embed
// (TF:models/wav2vec2/modeling_wav2vec2.py:1272-1316; indices are drawn on the host).
template <typename T>
__global__ void mask_rows_kernel(T* __restrict__ x, const int* __restrict__ rows, int nrows, const float* __restrict__ emb, int D) {
    const int r = blockIdx.x;
    if (r >= nrows || rows[r] < 0) return;          // (negative: padding of a fixed-capacity row list - captured steps)
    T* dst = x + (long long)rows[r] * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) Cvt<T>::st(dst + c, emb[c]);
}
extern "C" int smx_mask_rows(void* x, const int* rows, int nrows, const float* emb, int D, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (nrows <= 0) return SMX_OK;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(mask_rows_kernel<bf16_t>, dim3(nrows), dim3(256), 0, stream, (bf16_t*)x, rows, nrows, emb, D);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(mask_rows_kernel<float>, dim3(nrows), dim3(256), 0, stream, (float*)x, rows, nrows, emb, D);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}
// backward: d_emb += sum over masked rows of dx; dx rows zeroed
template <typename T>
__global__ void mask_rows_bwd_kernel(T* __restrict__ dx, const int* __restrict__ rows, int nrows, float* __restrict__ demb, int D) {
    const int r = blockIdx.x;
    if (r >= nrows || rows[r] < 0) return;
    T* src = dx + (long long)rows[r] * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        if (demb) atomicAdd(demb + c, Cvt<T>::ld(src + c));
        Cvt<T>::st(src + c, 0.f);
    }
}
extern "C" int smx_mask_rows_bwd(void* dx, const int* rows, int nrows, float* demb, int D, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (nrows <= 0) return SMX_OK;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(mask_rows_bwd_kernel<bf16_t>, dim3(nrows), dim3(256), 0, stream, (bf16_t*)dx, rows, nrows, demb, D);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(mask_rows_bwd_kernel<float>, dim3(nrows), dim3(256), 0, stream, (float*)dx, rows, nrows, demb, D);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- per-step dropout key (smx_common.h)
// One launch rewrites the key word of every translation unit whose kernels hash dropout masks; stream-ordered, so a step's
// kernels (eager or replayed from a captured graph) read the key that was set ahead of them.
SMX_STEP_KEY_TU(misc)
extern "C" int smx_step_key_addr_gemm(void**);
extern "C" int smx_step_key_addr_gemm_pp(void**);
extern "C" int smx_step_key_addr_gemm_fr(void**);
extern "C" int smx_step_key_addr_gemm_ws(void**);
extern "C" int smx_step_key_addr_norm(void**);
extern "C" int smx_step_key_addr_attention(void**);
struct SmxKeyAddrs { unsigned* a[8]; int n; };
__global__ void set_step_key_kernel(SmxKeyAddrs t, unsigned key) {
    if ((int)threadIdx.x < t.n) *t.a[threadIdx.x] = key;
}
extern "C" int smx_set_step_key(unsigned key, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    static SmxKeyAddrs tab[16];
    static bool done[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -5;
    dev &= 15;
    if (!done[dev]) {          // (first call per device: must not happen inside a stream capture - the engine sets a key eagerly first)
        int (*fns[])(void**) = {smx_step_key_addr_misc, smx_step_key_addr_gemm, smx_step_key_addr_gemm_pp, smx_step_key_addr_gemm_fr, smx_step_key_addr_gemm_ws,
                                smx_step_key_addr_norm, smx_step_key_addr_attention};
        tab[dev].n = 0;
        for (auto fn : fns) {
            void* a = nullptr;
            if (fn(&a) != SMX_OK || !a) return -5;
            tab[dev].a[tab[dev].n++] = (unsigned*)a;
        }
        done[dev] = true;
    }
    hipLaunchKernelGGL(set_step_key_kernel, dim3(1), dim3(64), 0, stream, tab[dev], key);
    SMX_CHECK_LAUNCH();
}

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxCEParams(void) { return (int)sizeof(SmxCEParams); }

// ---------------------------------------------------------------- dx = dy * act'(pre), rows written through a view
// (lets the result land inside a zero-padded per-clip buffer that the conv dgrad GEMMs window over)
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, int M, int N,
                               SmxRowView ov, int act) {
    const int cv = N / 8;
    const long long n = (long long)M * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int m = i / cv, c = (i % cv) * 8;
        float d[8], x[8];
        load8(dy + (long long)m * N + c, d);
        if (pre) {                                   // pre == null: plain copy of dy through the output view
            load8(pre + (long long)m * N + c, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] *= act_grad(x[e], act);   // (scalar form: this kernel is HBM-bound)
        }
        store8(dx + view_off(ov, m) + c, d);
    }
}
extern "C" int smx_act_bwd(const void* dy, const void* pre, void* dx, int M, int N, const SmxRowView* ov, int act, int dtype,
                           hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (M <= 0 || (N & 7)) return SMX_EINVAL;
    const long long n = (long long)M * (N / 8);
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)pre, (bf16_t*)dx, M, N, *ov, act);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dy, (const float*)pre, (float*)dx, M, N, *ov, act);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- SpeechMixSelf hidden-state matching (fp32, tiny)
// ref:speechmix/model.py:247-255: attn = softmax(bmm(H_text, H_speech.view(B,d,-1)) / sqrt(d)); MSE(bmm(attn,H_speech), H_text)
__global__ void softmax_rows_kernel(float* __restrict__ x, int R, int Cn) {   // in place, one wave per row
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    float* r = x + (long long)row * Cn;
    float m = -INFINITY;
    for (int j = lane; j < Cn; j += 64) m = fmaxf(m, r[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < Cn; j += 64) s += __expf(r[j] - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int j = lane; j < Cn; j += 64) r[j] = __expf(r[j] - m) * inv;
}
extern "C" int smx_softmax_rows(float* x, int R, int Cn, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, x, R, Cn);
    SMX_CHECK_LAUNCH();
}
// dx = scale * p * (dp - sum_j p_j dp_j)
__global__ void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ dx,
                                        int R, int Cn, float scale) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* pr = p + (long long)row * Cn;
    const float* dr = dp + (long long)row * Cn;
    float s = 0.f;
    for (int j = lane; j < Cn; j += 64) s += pr[j] * dr[j];
    s = wave_sum(s);
    for (int j = lane; j < Cn; j += 64) dx[(long long)row * Cn + j] = scale * pr[j] * (dr[j] - s);
}
extern "C" int smx_softmax_rows_bwd(const float* p, const float* dp, float* dx, int R, int Cn, float scale, hipStream_t stream) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, p, dp, dx, R, Cn, scale);
    SMX_CHECK_LAUNCH();
}
// loss += mean((a-b)^2);  da = gscale * 2 (a-b) / n
__global__ void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, float* loss, float* __restrict__ da,
                           long long n, float gscale) {
    __shared__ float sh[16];
    float s = 0.f;
    const float inv = 1.f / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        s += d * d;
        if (da) da[i] = gscale * 2.f * d * inv;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(loss, s * inv);
}
extern "C" int smx_mse(const float* a, const float* b, float* loss, float* da, long long n, float gscale, hipStream_t stream) {
    (void)hipGetLastError();
    long long blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, stream, a, b, loss, da, n, gscale);
    SMX_CHECK_LAUNCH();
}
// dst (dtype T) += src (fp32)
template <typename T>
__global__ void add_f32_into_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Cvt<T>::st(dst + i, Cvt<T>::ld(dst + i) + src[i]);
}
extern "C" int smx_add_f32_into(const float* src, void* dst, long long n, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(add_f32_into_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, n);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(add_f32_into_kernel<float>, dim3(blocks), dim3(256), 0, stream, src, (float*)dst, n);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- layer-weighted sum of the speech-encoder hidden states
// ref:speechmix/hf_model.py:411-423: out = sum_l softmax(w)_l * h_l over the L+1 hidden states (HBM-bound, L+1 reads).
#define WS_MAXL 40
struct SmxWsumParams {
    const void* h[WS_MAXL];   // L+1 hidden states [n] (dtype T)
    const float* w;           // raw weights [L+1] (softmax applied here)
    void* out;                // fwd: [n] output          | bwd: unused
    const void* dy;           // bwd: upstream gradient [n]
    float* dots;              // bwd: [L+1] fp32 scratch (zeroed here): sum dy * h_l
    float* dw;                // bwd: gradient of the raw weights [L+1] (accumulated)
    float* sw;                // softmax(w) written out [L+1] (fwd) / read (bwd)
    long long n;
    int L1;                   // number of states
};
template <typename T>
__global__ void wsum_fwd_kernel(SmxWsumParams p) {
    __shared__ float sw[WS_MAXL];
    if (threadIdx.x == 0) {
        float m = -INFINITY, s = 0.f;
        for (int l = 0; l < p.L1; ++l) m = fmaxf(m, p.w[l]);
        for (int l = 0; l < p.L1; ++l) { sw[l] = __expf(p.w[l] - m); s += sw[l]; }
        for (int l = 0; l < p.L1; ++l) { sw[l] /= s; if (blockIdx.x == 0 && p.sw) p.sw[l] = sw[l]; }
    }
    __syncthreads();
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long step = (long long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= p.n; i += step) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int l = 0; l < p.L1; ++l) {
            float v[8];
            load8(reinterpret_cast<const T*>(p.h[l]) + i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(sw[l], v[e], acc[e]);
        }
        store8(reinterpret_cast<T*>(p.out) + i, acc);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void wsum_dots_kernel(SmxWsumParams p) {
    __shared__ float sh[16];
    long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long step = (long long)gridDim.x * blockDim.x * 8;
    for (int l = 0; l < p.L1; ++l) {
        float s = 0.f;
        for (long long i = i0; i + 8 <= p.n; i += step) {
            float a[8], b[8];
            load8(reinterpret_cast<const T*>(p.dy) + i, a);
            load8(reinterpret_cast<const T*>(p.h[l]) + i, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(a[e], b[e], s);
        }
        s = block_sum(s, sh);
        if (threadIdx.x == 0) atomicAdd(p.dots + l, s);
    }
}
__global__ void wsum_zero_kernel(float* dots, int n) {
    if ((int)threadIdx.x < n) dots[threadIdx.x] = 0.f;
}
__global__ void wsum_dw_kernel(SmxWsumParams p) {   // softmax backward on L+1 values
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float dot = 0.f;
    for (int l = 0; l < p.L1; ++l) dot += p.sw[l] * p.dots[l];
    for (int l = 0; l < p.L1; ++l) p.dw[l] += p.sw[l] * (p.dots[l] - dot);
}
extern "C" int smx_sizeof_SmxWsumParams(void) { return (int)sizeof(SmxWsumParams); }
extern "C" int smx_weighted_sum_fwd(const SmxWsumParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    SmxWsumParams p = *pp;
    if (p.L1 < 1 || p.L1 > WS_MAXL || (p.n & 7)) return SMX_EINVAL;
    long long blocks = (p.n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(wsum_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, p);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(wsum_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, p);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}
extern "C" int smx_weighted_sum_bwd(const SmxWsumParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    SmxWsumParams p = *pp;
    if (p.L1 < 1 || p.L1 > WS_MAXL || (p.n & 7) || !p.dots || !p.dw || !p.sw) return SMX_EINVAL;
    // (zeroed by a KERNEL: inside a stream capture hipMemsetAsync becomes a memset node, and the replayed node did not clear the 20-byte scratch
    //  - dots[0] kept whatever the pool block held, which is what "one replayed hidden state holds garbage" of round 5 was: the weighted-sum
    //  model is the only captured path that reached a memset; tests/test_gpu_r5.py with SMX_CAPTURE_GC_GUARD=collect failed 5 of 5 before, 0 after)
    hipLaunchKernelGGL(wsum_zero_kernel, dim3(1), dim3(64), 0, stream, p.dots, p.L1);
    long long blocks = (p.n / 8 + 255) / 256;
    if (blocks > 512) blocks = 512;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(wsum_dots_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, p);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(wsum_dots_kernel<float>, dim3(blocks), dim3(256), 0, stream, p);
    else return SMX_EINVAL;
    hipLaunchKernelGGL(wsum_dw_kernel, dim3(1), dim3(64), 0, stream, p);
    SMX_CHECK_LAUNCH();
}
// y += a[idx] * x   (a: device scalar array; adds the weighted-sum gradient share to a hidden state's gradient)
template <typename T>
__global__ void axpy_dev_kernel(T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ a, int idx, long long n, int init) {
    const float s = a[idx];
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long step = (long long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += step) {
        float u[8], v[8];
        load8(x + i, v);
        if (init) {
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = s * v[e];
        } else {
            load8(y + i, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = fmaf(s, v[e], u[e]);
        }
        store8(y + i, u);
    }
}
extern "C" int smx_axpy_dev(void* y, const void* x, const float* a, int idx, long long n, int init, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (n & 7) return SMX_EINVAL;
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(axpy_dev_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (bf16_t*)y, (const bf16_t*)x, a, idx, n, init);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(axpy_dev_kernel<float>, dim3(blocks), dim3(256), 0, stream, (float*)y, (const float*)x, a, idx, n, init);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- out = x * dropout_mask(seed) (elementwise, flat index)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, T* __restrict__ out, long long n, float p, unsigned seed) {
    seed = smx_dseed(p, seed);        // + the step key (smx_common.h), read once
    const unsigned th = smx_thresh24(p);
    const float inv = 1.0f / (1.0f - p);
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    const long long step = (long long)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += step) {
        float v[8];
        load8(x + i, v);
        smx_drop_mul8(seed, (unsigned)i, th, inv, v);              // i % 8 == 0
        store8(out + i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {      // ragged tail (only odd-sized test tensors get here)
        const long long t = (n & ~7ll) + threadIdx.x;
        Cvt<T>::st(out + t, Cvt<T>::ld(x + t) * smx_drop_mul(seed, (unsigned)t, th, inv));
    }
}
// out = dropout(x) (same mask function and flat index as dropout_kernel) fused with the column sums of out: the backward
// of a Linear whose output was dropped needs both the masked gradient (operand of its dgrad / wgrad GEMMs) and that
// gradient's column sums (its bias gradient) - one pass over the tensor instead of two.  Partial rows + colsum_fold_kernel.
template <typename T>
__global__ __launch_bounds__(256) void dropout_colsum_kernel(const T* __restrict__ x, T* __restrict__ out, float* __restrict__ ws,
                                                             int M, int N, int Np, float p, unsigned seed) {
    seed = smx_dseed(p, seed);        // + the step key (smx_common.h), read once
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * 8;
    const unsigned th = smx_thresh24(p);
    const float inv = 1.0f / (1.0f - p);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < N) {                                   // N % 8 == 0
        const int step = gridDim.y * 4;
        int m = blockIdx.y * 4 + w;
        for (; m + 3 * step < M; m += 4 * step) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8(x + (long long)(m + u * step) * N + c, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned idx = (unsigned)((long long)(m + u * step) * N + c);
                smx_drop_mul8(seed, idx, th, inv, v[u]);             // idx % 8 == 0 (N % 8 == 0)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[u][e] = rt(v[u][e], out);
                    acc[e] += v[u][e];
                }
                store8(out + (long long)(m + u * step) * N + c, v[u]);
            }
        }
        for (; m < M; m += step) {
            float v[8];
            load8(x + (long long)m * N + c, v);
            const unsigned idx = (unsigned)((long long)m * N + c);
            smx_drop_mul8(seed, idx, th, inv, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = rt(v[e], out);
                acc[e] += v[e];
            }
            store8(out + (long long)m * N + c, v);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[w][lane][e] = acc[e];
    __syncthreads();
    if (w == 0 && c < N) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
        store8(ws + (long long)blockIdx.y * Np + c, o);
    }
}
// ws: >= smx_colsum_ws_floats(M, N) floats.  colsum[n] += alpha * sum_m out[m, n]
extern "C" int smx_dropout_colsum_slices(int M, int N) {
    int gy = colsum_slices(M, N);
    if (gy < 64 && M >= 4096) gy = 64;
    return gy;
}
extern "C" int smx_dropout_colsum(const void* x, void* out, int M, int N, float p, unsigned seed, float* colsum, float alpha,
                                  float* ws, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (M <= 0 || N <= 0 || (N & 7) || p < 0.f || p >= 1.f || !ws) return SMX_EINVAL;
    const int gx = (N + 511) / 512, Np = N;
    const int gy = smx_dropout_colsum_slices(M, N);
    dim3 grid(gx, gy);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(dropout_colsum_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, ws, M, N, Np, p, seed);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(dropout_colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (float*)out, ws, M, N, Np, p, seed);
    else return SMX_EINVAL;
    // colsum == null: the caller folds the gy partial rows of N floats later (smx_fold_many)
    if (colsum) hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, ws, colsum, gy, N, Np, alpha);
    SMX_CHECK_LAUNCH();
}

extern "C" int smx_dropout(const void* x, void* out, long long n, float p, unsigned seed, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || p < 0.f || p >= 1.f) return SMX_EINVAL;
    long long blocks = (n / 8 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (dtype == SMX_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)out, n, p, seed);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)out, n, p, seed);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// ---------------------------------------------------------------- on-box peak probes (bench.py: `peaks_measured`)
// What THIS box's matrix pipes and HBM deliver, next to the datasheet values the roofline fractions are quoted against
// (SURVEY.md section 8d "state both").  MFMA: eight waves per CU (two per SIMD), four independent 32x32x16 bf16 accumulator
// chains each on non-trivial operands (zero operands clock ~20 % higher, MI355X_MICROARCH.md DVFS note).  HBM: a 16-B-per-lane
// streaming copy (read + write bytes counted).
typedef __attribute__((ext_vector_type(16))) float probe_f32x16_t;
__global__ __launch_bounds__(256) void probe_mfma_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8_t x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        x[e] = (short)(0x3f80 + ((lane * 37 + e * 11) & 0x7f) - ((lane + e) & 1) * 0x8000);       // +-[1, 2) bf16 patterns
        y[e] = (short)(0x3f00 + ((lane * 13 + e * 29) & 0x7f) - ((lane * 3 + e) & 1) * 0x8000);
    }
    probe_f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    if (s == 1234.5678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;       // keeps the chains live, never true in practice
}
// The same loop with a choice of operands and the shader clock it ran at (the box's probe reads ~1 985 TF/s where the
// guide measured 2 495): zero = 1 multiplies zeros - the chip then clocks ~20 % higher (DVFS, MI355X_MICROARCH.md); clk[0] / clk[1]
// receive block 0's shader cycles (s_memtime) and its 100-MHz real-time ticks over the loop, i.e. GHz = clk[0] / (10 clk[1]).
__global__ __launch_bounds__(256) void probe_mfma_clk_kernel(float* out, int iters, int zero, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8_t x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        x[e] = zero ? (short)0 : (short)(0x3f80 + ((lane * 37 + e * 11) & 0x7f) - ((lane + e) & 1) * 0x8000);
        y[e] = zero ? (short)0 : (short)(0x3f00 + ((lane * 13 + e * 29) & 0x7f) - ((lane * 3 + e) & 1) * 0x8000);
    }
    probe_f32x16_t a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    if (s == 1234.5678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" double smx_probe_mfma_clk(float* out, int blocks, int iters, int zero, unsigned long long* clk, hipStream_t stream) {
    (void)hipGetLastError();
    if (blocks <= 0 || iters <= 0 || !out) return -1.0;
    hipLaunchKernelGGL(probe_mfma_clk_kernel, dim3(blocks), dim3(256), 0, stream, out, iters, zero, clk);
    if (hipGetLastError() != hipSuccess) return -1.0;
    return (double)blocks * 4.0 * (double)iters * 16.0 * 2.0 * 32.0 * 32.0 * 16.0;
}
// -> flops issued by one launch (the caller times it with events): blocks x 4 waves x iters x 16 MFMAs x 2 * 32 * 32 * 16
extern "C" double smx_probe_mfma(float* out, int blocks, int iters, hipStream_t stream) {
    (void)hipGetLastError();
    if (blocks <= 0 || iters <= 0 || !out) return -1.0;
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, stream, out, iters);
    if (hipGetLastError() != hipSuccess) return -1.0;
    return (double)blocks * 4.0 * (double)iters * 16.0 * 2.0 * 32.0 * 32.0 * 16.0;
}
__global__ __launch_bounds__(256) void probe_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
// dst[0 .. bytes) = src[0 .. bytes) as ONE kernel on `stream` (16-byte aligned pointers).  A replayed step moves its inputs and a
// LayerDrop-dropped layer's pass-through with this instead of hipMemcpyAsync: the runtime's copy path put ~150 us of idle GPU
// time in front of every copy (kernel trace, round 5).  src may be pinned host memory (read through the host mapping).
__global__ __launch_bounds__(256) void copy_bytes_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long n16, int tail) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail)
        reinterpret_cast<unsigned char*>(dst + n16)[threadIdx.x] = reinterpret_cast<const unsigned char*>(src + n16)[threadIdx.x];
}
extern "C" int smx_copy_bytes(const void* src, void* dst, long long bytes, hipStream_t stream) {
    (void)hipGetLastError();
    if (bytes <= 0) return SMX_OK;
    if (!src || !dst || (((size_t)src | (size_t)dst) & 15)) return SMX_EINVAL;
    const long long n16 = bytes / 16;
    const int blocks = (int)std::min<long long>(256 * 8, std::max<long long>(1, (n16 + 255) / 256));
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(blocks), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, n16, (int)(bytes & 15));
    SMX_CHECK_LAUNCH();
}
extern "C" int smx_probe_copy(const void* src, void* dst, long long bytes, hipStream_t stream) {
    (void)hipGetLastError();
    if (bytes < 16 || !src || !dst) return SMX_EINVAL;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes / 16);
    SMX_CHECK_LAUNCH();
}
