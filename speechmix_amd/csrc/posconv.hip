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
__global__ __launch_bounds__(256) void wn_norm_kernel(const float* __restrict__ v, float* __restrict__ norm, int C, int Cg, int K) {
    __shared__ float sh[16];
    const int k = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < C * Cg; i += 256) {
        const float a = v[(long long)i * K + k];
        s += a * a;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) norm[k] = sqrtf(s);
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
    hipLaunchKernelGGL(wn_norm_kernel, dim3(K), dim3(256), 0, stream, v, norm, C, Cg, K);
    const long long n = (long long)C * Cg * K;
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(wn_pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, v, g, norm, (bf16_t*)wp, (bf16_t*)wf, C, Cg, K);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(wn_pack_kernel<float>, dim3(blocks), dim3(256), 0, stream, v, g, norm, (float*)wp, (float*)wf, C, Cg, K);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// dwp: fp32 [C][k*Cg+ci] (the forward-pack layout).  s[k] = sum dW*v;  dg[k] += s/norm;
// dv += g/norm * dW - g*s/norm^3 * v
__global__ __launch_bounds__(256) void wn_bwd_dot_kernel(const float* __restrict__ dwp, const float* __restrict__ v,
                                                         float* __restrict__ s, int C, int Cg, int K) {
    __shared__ float sh[16];
    const int k = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < C * Cg; i += 256) {
        const int ci = i % Cg, co = i / Cg;
        a += dwp[((long long)co * K + k) * Cg + ci] * v[(long long)i * K + k];
    }
    a = block_sum(a, sh);
    if (threadIdx.x == 0) s[k] = a;
}
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
    hipLaunchKernelGGL(wn_bwd_dot_kernel, dim3(K), dim3(256), 0, stream, dwp, v, scratch_s, C, Cg, K);
    const long long n = (long long)C * Cg * K;
    int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(wn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, stream, dwp, v, g, norm, scratch_s, dg, dv, C, Cg, K);
    SMX_CHECK_LAUNCH();
}
