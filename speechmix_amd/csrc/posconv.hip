// Helpers that turn wav2vec2's positional convolution  x + GELU(Conv1d(d, d, k=128, pad=64, groups=16)(x))
// with weight_norm(dim=2)  (TF:models/wav2vec2/modeling_wav2vec2.py:326-379) into batched MFMA GEMMs:
//
//  * smx_group_pack:   [B, T, C] channels-last  ->  group-major zero-padded [G, B, T+K-1, Cg].  In that layout
//    the K*Cg inputs of one output frame are ONE contiguous run, so the grouped conv is a GEMM over an
//    overlapping row view (row stride Cg) - no im2col.  The same pack (different front pad) feeds dgrad.
//  * smx_wn_fwd:       weight-norm g*v/||v|| evaluated once per step, emitted directly in the two GEMM
//    operand layouts: forward  Wp[g][co][k*Cg+ci]  and data-gradient  Wf[g][ci][k'*Cg+co] (taps flipped).
//  * smx_wn_bwd:       gradient of the packed weight back to (g, v).
#include "smx_common.h"

template <typename T>
__global__ void group_pack_kernel(const T* __restrict__ x, T* __restrict__ xg, int B, int Tt, int C, int G, int Tp,
                                  int pad_front) {
    const int Cg = C / G, cv = Cg / 8;
    const long long n = (long long)G * B * Tp * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = i % cv;
        const int tau = (i / cv) % Tp;
        const int b = (i / ((long long)cv * Tp)) % B;
        const int g = i / ((long long)cv * Tp * B);
        const int t = tau - pad_front;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= 0 && t < Tt) load8(x + ((long long)b * Tt + t) * C + g * Cg + c8 * 8, v);
        store8(xg + i * 8, v);
    }
}
extern "C" int smx_group_pack(const void* x, void* xg, int B, int T, int C, int G, int K, int pad_front, int dtype,
                              hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (C % G || (C / G) % 8) return SMX_EINVAL;
    const int Tp = T + K - 1;
    const long long n = (long long)G * B * Tp * (C / G / 8);
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(group_pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)xg, B, T, C, G, Tp, pad_front);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(group_pack_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)xg, B, T, C, G, Tp, pad_front);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// norm[k] = || v[:, :, k] ||_2   (v: [C, Cg, K]); one block per tap
// Norms over (co, ci) for every tap k of v[co][ci][k] (k fastest).  A block owns WN_ROWS (co, ci) rows, its threads walk
// along k (coalesced 512-B reads; one block per k read every 128th float of the whole tensor instead) and leave one partial
// row [K]; a second tiny launch adds the partial rows in order (deterministic) and takes the root.
#define WN_ROWS 64
extern "C" int smx_wn_partial_blocks(int C, int Cg) { return (C * Cg + WN_ROWS - 1) / WN_ROWS; }
// s_part[blk][k] = sum over the block's rows of a[row][k] * b[row][k]   (b == a: squares); K <= 256, 256 threads
__global__ __launch_bounds__(256) void wn_partial_kernel(const float* __restrict__ v, const float* __restrict__ dwp, float* __restrict__ part,
                                                         int rows, int Cg, int K) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int lanes_k = K <= 128 ? 128 : 256;               // threads along k; the rest of the block takes other rows
    const int k = tid % lanes_k, h = tid / lanes_k, nh = 256 / lanes_k;
    const int r0 = blockIdx.x * WN_ROWS, r1 = min(rows, r0 + WN_ROWS);
    float s = 0.f;
    if (k < K) {
#pragma unroll 8
        for (int r = r0 + h; r < r1; r += nh) {             // (independent trips: several loads in flight)
            const float a = v[(long long)r * K + k];
            float b = a;
            if (dwp) {                                       // backward: dW in the forward-pack layout [co][k*Cg+ci]
                const int ci = r % Cg, co = r / Cg;
                b = dwp[((long long)co * K + k) * Cg + ci];
            }
            s += a * b;
        }
    }
    red[tid] = s;
    __syncthreads();
    if (h == 0 && k < K) {
        float a = red[k];
        for (int j = 1; j < nh; ++j) a += red[j * lanes_k + k];
        part[(long long)blockIdx.x * K + k] = a;
    }
}
// out[k] = (root of) the sum of the nb partial rows, one block per k, partial rows added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void wn_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, int K, int root) {
    __shared__ float sh[16];
    const int k = blockIdx.x;
    float a = 0.f;
    for (int j = threadIdx.x; j < nb; j += 256) a += part[(long long)j * K + k];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) out[k] = root ? sqrtf(a) : a;
}
template <typename T>
__global__ void wn_pack_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ norm,
                               T* __restrict__ wp, T* __restrict__ wf, int C, int Cg, int K) {
    const long long n = (long long)C * Cg * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = i % K, ci = (i / K) % Cg, co = i / ((long long)K * Cg);
        const int grp = co / Cg, col = co % Cg;
        const float w = v[i] * g[k] / norm[k];
        Cvt<T>::st(wp + ((long long)co * K + k) * Cg + ci, w);                                   // [G][col][k*Cg+ci]
        if (wf) Cvt<T>::st(wf + (((long long)grp * Cg + ci) * K + (K - 1 - k)) * Cg + col, w);            // [G][ci][k'*Cg+col]
    }
}
extern "C" int smx_wn_fwd(const float* v, const float* g, void* wp, void* wf, float* norm, int C, int Cg, int K, int dtype,
                          hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (K > 256 || K <= 0) return SMX_EINVAL;
    const int nb = smx_wn_partial_blocks(C, Cg);             // norm: [K] norms followed by nb x K floats of scratch
    hipLaunchKernelGGL(wn_partial_kernel, dim3(nb), dim3(256), 0, stream, v, (const float*)nullptr, norm + K, C * Cg, Cg, K);
    hipLaunchKernelGGL(wn_finish_kernel, dim3(K), dim3(256), 0, stream, norm + K, norm, nb, K, 1);
    const long long n = (long long)C * Cg * K;
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(wn_pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, v, g, norm, (bf16_t*)wp, (bf16_t*)wf, C, Cg, K);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(wn_pack_kernel<float>, dim3(blocks), dim3(256), 0, stream, v, g, norm, (float*)wp, (float*)wf, C, Cg, K);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// dwp: fp32 [C][k*Cg+ci] (the forward-pack layout).  s[k] = sum dW*v;  dg[k] += s/norm;
// dv += g/norm * dW - g*s/norm^3 * v
__global__ void wn_bwd_apply_kernel(const float* __restrict__ dwp, const float* __restrict__ v, const float* __restrict__ g,
                                    const float* __restrict__ norm, const float* __restrict__ s, float* __restrict__ dg,
                                    float* __restrict__ dv, int C, int Cg, int K) {
    const long long n = (long long)C * Cg * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = i % K, ci = (i / K) % Cg, co = i / ((long long)K * Cg);
        const float nk = norm[k], gk = g[k];
        const float dw = dwp[((long long)co * K + k) * Cg + ci];
        dv[i] += gk / nk * dw - gk * s[k] / (nk * nk * nk) * v[i];
        if (i < K) dg[i] += s[i] / norm[i];
    }
}
extern "C" int smx_wn_bwd(const float* dwp, const float* v, const float* g, const float* norm, float* scratch_s, float* dg,
                          float* dv, int C, int Cg, int K, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    if (K > 256 || K <= 0) return SMX_EINVAL;
    const int nb = smx_wn_partial_blocks(C, Cg);             // scratch_s: [K] dots followed by nb x K floats
    hipLaunchKernelGGL(wn_partial_kernel, dim3(nb), dim3(256), 0, stream, v, dwp, scratch_s + K, C * Cg, Cg, K);
    hipLaunchKernelGGL(wn_finish_kernel, dim3(K), dim3(256), 0, stream, scratch_s + K, scratch_s, nb, K, 0);
    const long long n = (long long)C * Cg * K;
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, stream, dwp, v, g, norm, scratch_s, dg, dv, C, Cg, K);
    SMX_CHECK_LAUNCH();
}
