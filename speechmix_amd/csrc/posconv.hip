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


// ---------------------------------------------------------------- time-blocked positional convolution (round 4)
// As a batched GEMM the grouped conv has N = Cg = 48 outputs per group: 62 % of a 128-wide tile (and all of a 256-wide one beyond
// column 48) multiplies nothing.  Blocking J consecutive frames into ONE GEMM row - row (b, t') = frames J t' .. J t' + J - 1 of
// clip b, A = the (K + J - 1) Cg inputs they see together (one contiguous run of the group-major padded layout, rows J Cg apart),
// B = the taps shifted J times - gives N = J Cg outputs at (K + J - 1) / K of the flops, and the output [G][B][J T'][Cg] is again the
// group-major layout.  J = 4: N = 192, K' = 6288; forward 295 -> 150 us, data gradient 447 -> 150 us, weight gradient 475 -> 300 us
// at config 2 (tools/gpu_posconv_probe.py).
//
//   smx_posconv_pack_w:   out[g][(j, a)][(kk, b)] = 0 <= kk - j < K ? (flip ? wp[g][b][K - 1 - (kk - j)][a] : wp[g][a][kk - j][b]) : 0
//                         (wp = smx_wn_fwd's forward pack [G][co][k Cg + ci]; flip = 1: the data-gradient operand, taps reversed and
//                         the roles of ci / co swapped)
//   smx_posconv_unpack:   group-major fp32 [G][B][Tq][Cg] -> token-major [B T][C] with the epilogue the fused GEMM had:
//                         v = tmp + bias; pre = v; y = act(v) + resid      (data gradient: y = tmp + resid)
//   smx_posconv_fold_dw:  dwp[g][co][k][ci] = sum_j dwJ[g][(j, co)][(k + j)][ci]     (the J shifted diagonals of the blocked gradient)
template <typename T>
__global__ void posconv_pack_w_kernel(const T* __restrict__ wp, T* __restrict__ out, int G, int Cg, int K, int J, int flip) {
    const int Kp = K + J - 1;
    const long long n = (long long)G * J * Cg * Kp * Cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = i % Cg;
        const int kk = (i / Cg) % Kp;
        const int a = (i / ((long long)Cg * Kp)) % Cg;
        const int j = (i / ((long long)Cg * Kp * Cg)) % J;
        const int g = i / ((long long)Cg * Kp * Cg * J);
        const int k = kk - j;
        T v = T(0);
        if (k >= 0 && k < K)
            v = flip ? wp[(((long long)g * Cg + b) * K + (K - 1 - k)) * Cg + a] : wp[(((long long)g * Cg + a) * K + k) * Cg + b];
        out[i] = v;
    }
}
// flip = 0 fast path: output row (g, j, a) = [j Cg zeros][row (g, a) of the source: K Cg contiguous elements][(J - 1 - j) Cg zeros]
// - a shifted row copy, 16 B per thread.  (The data-gradient operand is the same copy of smx_wn_fwd's flipped pack `wf`.)
template <typename T>
__global__ void posconv_shift_rows_kernel(const T* __restrict__ src, T* __restrict__ out, int rows, int Cg, int K, int J) {
    const int Kp = K + J - 1, cv = Kp * Cg / 8, kv = K * Cg / 8;
    const long long n = (long long)rows * J * cv;                 // rows = G * Cg source rows
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const long long rj = i / cv;                              // = (g * J + j) * Cg + a
        const int a = (int)(rj % Cg), j = (int)((rj / Cg) % J);
        const long long g = rj / ((long long)Cg * J);
        const int sc = c - j * (Cg / 8);
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (sc >= 0 && sc < kv) load8(src + (g * Cg + a) * (long long)K * Cg + sc * 8, v);
        store8(out + i * 8, v);
    }
}
extern "C" int smx_posconv_pack_w(const void* wp, void* out, int G, int Cg, int K, int J, int flip, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (!wp || !out || G <= 0 || Cg <= 0 || K <= 0 || J <= 0) return SMX_EINVAL;
    const long long n = (long long)G * J * Cg * (K + J - 1) * Cg;
    if (!flip && !(Cg & 7)) {
        const long long nv = n / 8;
        int blocks = (int)((nv + 255) / 256 > 8192 ? 8192 : (nv + 255) / 256);
        if (dtype == SMX_BF16) hipLaunchKernelGGL(posconv_shift_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)wp, (bf16_t*)out, G * Cg, Cg, K, J);
        else if (dtype == SMX_F32) hipLaunchKernelGGL(posconv_shift_rows_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)wp, (float*)out, G * Cg, Cg, K, J);
        else return SMX_EINVAL;
        SMX_CHECK_LAUNCH();
    }
    int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(posconv_pack_w_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)wp, (bf16_t*)out, G, Cg, K, J, flip);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(posconv_pack_w_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)wp, (float*)out, G, Cg, K, J, flip);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

template <typename T>
__global__ void posconv_unpack_kernel(const float* __restrict__ tmp, const float* __restrict__ bias, const T* __restrict__ resid,
                                      T* __restrict__ pre, T* __restrict__ y, int B, int Tt, int C, int G, int Tq, int act) {
    const int Cg = C / G, cv = C / 8;
    const long long n = (long long)B * Tt * cv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (i % cv) * 8;
        const int t = (i / cv) % Tt;
        const int b = i / ((long long)cv * Tt);
        const int g = c / Cg, co = c % Cg;
        const float* src = tmp + (((long long)g * B + b) * Tq + t) * Cg + co;
        float v[8], r[8];
        const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        if (bias) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[c + e];
        }
        const long long o = ((long long)b * Tt + t) * C + c;
        if (pre) store8(pre + o, v);
        if (act != SMX_ACT_NONE) act_fwd8(v, act);
        if (resid) {
            load8(resid + o, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        store8(y + o, v);
    }
}
extern "C" int smx_posconv_unpack(const float* tmp, const float* bias, const void* resid, void* pre, void* y, int B, int T, int C, int G,
                                  int Tq, int act, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (!tmp || !y || C % G || (C / G) % 8 || Tq < T) return SMX_EINVAL;
    const long long n = (long long)B * T * (C / 8);
    int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    if (dtype == SMX_BF16) hipLaunchKernelGGL(posconv_unpack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, tmp, bias, (const bf16_t*)resid, (bf16_t*)pre, (bf16_t*)y, B, T, C, G, Tq, act);
    else if (dtype == SMX_F32) hipLaunchKernelGGL(posconv_unpack_kernel<float>, dim3(blocks), dim3(256), 0, stream, tmp, bias, (const float*)resid, (float*)pre, (float*)y, B, T, C, G, Tq, act);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

__global__ void posconv_fold_dw_kernel(const float* __restrict__ dwJ, float* __restrict__ dwp, int G, int Cg, int K, int J) {
    const int Kp = K + J - 1;
    const long long n = (long long)G * Cg * K * Cg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int ci = i % Cg;
        const int k = (i / Cg) % K;
        const int co = (i / ((long long)Cg * K)) % Cg;
        const int g = i / ((long long)Cg * K * Cg);
        float s = 0.f;
        for (int j = 0; j < J; ++j)                              // (fixed order: deterministic)
            s += dwJ[((((long long)g * J + j) * Cg + co) * Kp + (k + j)) * Cg + ci];
        dwp[i] = s;
    }
}
extern "C" int smx_posconv_fold_dw(const float* dwJ, float* dwp, int G, int Cg, int K, int J, hipStream_t stream) {
    (void)hipGetLastError();
    if (!dwJ || !dwp || G <= 0 || Cg <= 0 || K <= 0 || J <= 0) return SMX_EINVAL;
    const long long n = (long long)G * Cg * K * Cg;
    int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(posconv_fold_dw_kernel, dim3(blocks), dim3(256), 0, stream, dwJ, dwp, G, Cg, K, J);
    SMX_CHECK_LAUNCH();
}
