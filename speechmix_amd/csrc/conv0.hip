// Layer 0 of the wav2vec2 / HuBERT feature extractor: Conv1d(1 -> C, k, stride) on the raw waveform,
// fused with GroupNorm(C groups: per-(clip,channel) statistics over time) + affine + GELU
// (TF:models/wav2vec2/modeling_wav2vec2.py:301-323), or emitted plain for the "layer" extractor variant
// (conv(+bias) -> LayerNorm over C -> GELU, :275-299; the LN+GELU is smx_norm_fwd with act=GELU).
//
// HBM-bound: C_in = 1 so the 10-tap conv is recomputed from the waveform (0.64 MB / 10 s clip, LDS-staged)
// instead of ever being stored; the only large traffic is the channels-last output [B, T0, C]
// (32.8 MB / clip in bf16), written once with 16-B stores.  Statistics are accumulated in fp64 atomics.
// Thread <-> 8 consecutive channels, wave <-> time steps.
#include "smx_common.h"

#define C0_MAXK 16
#define C0_TT 64     // time steps per block

struct SmxConv0Params {
    const float* wave;    // [B, N] fp32
    const float* w;       // [C, k] fp32
    const float* cbias;   // [C] or null
    const float* gamma;   // [C] GroupNorm affine (group mode)
    const float* beta;
    double* stats;        // [B, C, 2] sum, sumsq of the conv output (group mode)
    void* y;              // [B, T0, C] output (dtype T)
    const void* dy;       // backward: grad wrt output [B, T0, C]
    double* bstats;       // backward: [B, C, 2]  sum dz, sum dz*xhat
    float* dw;            // [C, k] fp32 (atomic)
    float* dcbias;        // [C] or null (plain mode)
    float* dgamma;        // [C]
    float* dbeta;
    int B, N, C, k, stride, T0;
    int group;            // 1: GroupNorm+GELU fused, 0: plain conv
    float eps;
    int tiles_per_block;  // set by the launchers: consecutive time tiles handled by one block (reduction kernels)
};

__device__ __forceinline__ void load_w8(const SmxConv0Params& p, int c0, float w[8][C0_MAXK], float cb[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        cb[j] = p.cbias ? p.cbias[c0 + j] : 0.f;
#pragma unroll
        for (int t = 0; t < C0_MAXK; ++t) w[j][t] = t < p.k ? p.w[(c0 + j) * p.k + t] : 0.f;
    }
}

__device__ __forceinline__ void stage_wave(const SmxConv0Params& p, float* sx, int b, int t0) {
    const int n0 = t0 * p.stride, cnt = C0_TT * p.stride + C0_MAXK;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int n = n0 + i;
        sx[i] = n < p.N ? p.wave[(long long)b * p.N + n] : 0.f;
    }
}

// pass 1 (group mode): per-(b,c) sum / sumsq of u = conv(x).  A block walks `tiles_per_block` time tiles with
// register accumulators, reduces its 4 waves through LDS and issues ONE fp64 atomic pair per channel.
__global__ __launch_bounds__(256) void conv0_stats_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    __shared__ float red[4][64][16];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = lane * 8;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    float s[8], q[8], w[8][C0_MAXK], cb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    if (c0 < p.C) load_w8(p, c0, w, cb);
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (c0 < p.C) {
            for (int tt = wv; tt < C0_TT && t0 + tt < p.T0; tt += 4) {
                const float* x = sx + tt * p.stride;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float u = cb[j];
#pragma unroll
                    for (int t = 0; t < C0_MAXK; ++t) u = fmaf(w[j][t], x[t], u);
                    s[j] += u;
                    q[j] += u * u;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wv][lane][j] = s[j]; red[wv][lane][8 + j] = q[j]; }
    __syncthreads();
    if (wv == 0 && c0 < p.C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ss = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
            const float qq = red[0][lane][8 + j] + red[1][lane][8 + j] + red[2][lane][8 + j] + red[3][lane][8 + j];
            atomicAdd(p.stats + ((long long)b * p.C + c0 + j) * 2, (double)ss);
            atomicAdd(p.stats + ((long long)b * p.C + c0 + j) * 2 + 1, (double)qq);
        }
    }
}

__device__ __forceinline__ void mean_rstd(const SmxConv0Params& p, int b, int c, float& mean, float& rstd) {
    const double s = p.stats[((long long)b * p.C + c) * 2], q = p.stats[((long long)b * p.C + c) * 2 + 1];
    const double m = s / p.T0;
    double var = q / p.T0 - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)p.eps));
}

// pass 2: y = GELU(gamma * (u - mean) * rstd + beta)  (group)   or   y = u  (plain)
template <typename T>
__global__ __launch_bounds__(256) void conv0_apply_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, t0 = blockIdx.x * C0_TT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    stage_wave(p, sx, b, t0);
    __syncthreads();
    T* Y = reinterpret_cast<T*>(p.y) + (long long)b * p.T0 * p.C;
    const int c0 = lane * 8;
    if (c0 >= p.C) return;
    float w[8][C0_MAXK], cb[8], mu[8], rs[8], gm[8], bt[8];
    load_w8(p, c0, w, cb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = 0.f; rs[j] = 1.f; gm[j] = 1.f; bt[j] = 0.f;
        if (p.group) {
            mean_rstd(p, b, c0 + j, mu[j], rs[j]);
            gm[j] = p.gamma[c0 + j];
            bt[j] = p.beta[c0 + j];
        }
    }
    for (int tt = wv; tt < C0_TT && t0 + tt < p.T0; tt += 4) {
        const float* x = sx + tt * p.stride;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float u = cb[j];
#pragma unroll
            for (int t = 0; t < C0_MAXK; ++t) u = fmaf(w[j][t], x[t], u);
            if (p.group) u = act_fwd((u - mu[j]) * rs[j] * gm[j] + bt[j], SMX_ACT_GELU);
            o[j] = u;
        }
        store8(Y + (long long)(t0 + tt) * p.C + c0, o);
    }
}

// backward pass 1 (group): S1 = sum_t dz, S2 = sum_t dz * xhat  with dz = dy * gelu'(z)
template <typename T>
__global__ __launch_bounds__(256) void conv0_bwd_stats_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    __shared__ float red[4][64][16];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = lane * 8;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    const T* dY = reinterpret_cast<const T*>(p.dy) + (long long)b * p.T0 * p.C;
    float s1[8], s2[8], w[8][C0_MAXK], cb[8], mu[8], rs[8], gm[8], bt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    if (c0 < p.C) {
        load_w8(p, c0, w, cb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mean_rstd(p, b, c0 + j, mu[j], rs[j]);
            gm[j] = p.gamma[c0 + j];
            bt[j] = p.beta[c0 + j];
        }
    }
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (c0 < p.C) {
            for (int tt = wv; tt < C0_TT && t0 + tt < p.T0; tt += 4) {
                const float* x = sx + tt * p.stride;
                float d[8];
                load8(dY + (long long)(t0 + tt) * p.C + c0, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float u = cb[j];
#pragma unroll
                    for (int t = 0; t < C0_MAXK; ++t) u = fmaf(w[j][t], x[t], u);
                    const float xh = (u - mu[j]) * rs[j];
                    const float dz = d[j] * act_grad(xh * gm[j] + bt[j], SMX_ACT_GELU);
                    s1[j] += dz;
                    s2[j] += dz * xh;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[wv][lane][j] = s1[j]; red[wv][lane][8 + j] = s2[j]; }
    __syncthreads();
    if (wv == 0 && c0 < p.C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
            const float c = red[0][lane][8 + j] + red[1][lane][8 + j] + red[2][lane][8 + j] + red[3][lane][8 + j];
            atomicAdd(p.bstats + ((long long)b * p.C + c0 + j) * 2, (double)a);
            atomicAdd(p.bstats + ((long long)b * p.C + c0 + j) * 2 + 1, (double)c);
        }
    }
}

// dgamma / dbeta from the per-clip sums
__global__ void conv0_bwd_affine_kernel(SmxConv0Params p) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    double a = 0, g = 0;
    for (int b = 0; b < p.B; ++b) {
        a += p.bstats[((long long)b * p.C + c) * 2];
        g += p.bstats[((long long)b * p.C + c) * 2 + 1];
    }
    if (p.dbeta) atomicAdd(p.dbeta + c, (float)a);
    if (p.dgamma) atomicAdd(p.dgamma + c, (float)g);
}

// backward pass 2: du (through GroupNorm) then dW[c][t] += sum du * x[stride*t' + t].  Register accumulators
// across `tiles_per_block` tiles, LDS reduction over the 4 waves, one fp32 atomic per (channel, tap) per block.
template <typename T>
__global__ __launch_bounds__(256) void conv0_bwd_w_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    __shared__ float red[4][64][8];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = lane * 8;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    const T* dY = reinterpret_cast<const T*>(p.dy) + (long long)b * p.T0 * p.C;
    const float invT = 1.0f / (float)p.T0;
    float w[8][C0_MAXK], acc[8][C0_MAXK], cb[8], mu[8], rs[8], gm[8], bt[8], m1[8], m2[8], accb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = 0.f; rs[j] = 1.f; gm[j] = 1.f; bt[j] = 0.f; m1[j] = m2[j] = 0.f; accb[j] = 0.f; cb[j] = 0.f;
#pragma unroll
        for (int t = 0; t < C0_MAXK; ++t) { acc[j][t] = 0.f; w[j][t] = 0.f; }
    }
    if (c0 < p.C) {
        load_w8(p, c0, w, cb);
        if (p.group) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mean_rstd(p, b, c0 + j, mu[j], rs[j]);
                gm[j] = p.gamma[c0 + j]; bt[j] = p.beta[c0 + j];
                m1[j] = (float)(p.bstats[((long long)b * p.C + c0 + j) * 2] * invT);
                m2[j] = (float)(p.bstats[((long long)b * p.C + c0 + j) * 2 + 1] * invT);
            }
        }
    }
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (c0 < p.C) {
            for (int tt = wv; tt < C0_TT && t0 + tt < p.T0; tt += 4) {
                const float* x = sx + tt * p.stride;
                float d[8];
                load8(dY + (long long)(t0 + tt) * p.C + c0, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float du = d[j];
                    if (p.group) {
                        float u = cb[j];
#pragma unroll
                        for (int t = 0; t < C0_MAXK; ++t) u = fmaf(w[j][t], x[t], u);
                        const float xh = (u - mu[j]) * rs[j];
                        const float dz = du * act_grad(xh * gm[j] + bt[j], SMX_ACT_GELU);
                        du = gm[j] * rs[j] * (dz - m1[j] - xh * m2[j]);
                    }
                    accb[j] += du;
#pragma unroll
                    for (int t = 0; t < C0_MAXK; ++t) acc[j][t] = fmaf(du, x[t], acc[j][t]);
                }
            }
        }
    }
    // reduce the 4 waves (same channels, different time steps), one tap at a time
#pragma unroll
    for (int t = 0; t <= C0_MAXK; ++t) {
        if (t < C0_MAXK && t >= p.k) continue;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wv][lane][j] = t < C0_MAXK ? acc[j][t] : accb[j];
        __syncthreads();
        if (wv == 0 && c0 < p.C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
                if (t < C0_MAXK) atomicAdd(p.dw + (c0 + j) * p.k + t, v);
                else if (p.dcbias && !p.group) atomicAdd(p.dcbias + c0 + j, v);
            }
        }
    }
}

static int conv0_check(const SmxConv0Params& p) {
    if (p.B <= 0 || p.C <= 0 || (p.C & 7) || p.C > 512 || p.k > C0_MAXK || p.k <= 0 || p.stride <= 0 || p.stride > 8) return SMX_EINVAL;
    if (p.T0 != (p.N - p.k) / p.stride + 1 || p.T0 <= 0) return SMX_EINVAL;
    return SMX_OK;
}

extern "C" int smx_conv0_fwd(const SmxConv0Params* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxConv0Params p = *pp;
    int rc = conv0_check(p);
    if (rc) return rc;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    dim3 grid(ntiles, p.B);
    p.tiles_per_block = max(1, (ntiles * p.B + 1023) / 1024);
    dim3 rgrid((ntiles + p.tiles_per_block - 1) / p.tiles_per_block, p.B);
    if (p.group) {
        if (!p.stats || !p.gamma || !p.beta) return SMX_EINVAL;
        hipMemsetAsync(p.stats, 0, sizeof(double) * 2 * p.B * p.C, stream);
        hipLaunchKernelGGL(conv0_stats_kernel, rgrid, dim3(256), 0, stream, p);
    }
    if (dtype == SMX_F32) hipLaunchKernelGGL(conv0_apply_kernel<float>, grid, dim3(256), 0, stream, p);
    else if (dtype == SMX_BF16) hipLaunchKernelGGL(conv0_apply_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// group mode: needs `stats` from the forward; accumulates dw, dgamma, dbeta.  plain mode: dy is du.
extern "C" int smx_conv0_bwd(const SmxConv0Params* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxConv0Params p = *pp;
    int rc = conv0_check(p);
    if (rc) return rc;
    if (dtype != SMX_F32 && dtype != SMX_BF16) return SMX_EINVAL;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    p.tiles_per_block = max(1, (ntiles * p.B + 1023) / 1024);
    dim3 grid((ntiles + p.tiles_per_block - 1) / p.tiles_per_block, p.B);
    if (p.group) {
        if (!p.stats || !p.bstats) return SMX_EINVAL;
        hipMemsetAsync(p.bstats, 0, sizeof(double) * 2 * p.B * p.C, stream);
        if (dtype == SMX_F32) hipLaunchKernelGGL(conv0_bwd_stats_kernel<float>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv0_bwd_stats_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
        hipLaunchKernelGGL(conv0_bwd_affine_kernel, dim3((p.C + 255) / 256), dim3(256), 0, stream, p);
    }
    if (p.dw) {
        if (dtype == SMX_F32) hipLaunchKernelGGL(conv0_bwd_w_kernel<float>, grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(conv0_bwd_w_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
    }
    SMX_CHECK_LAUNCH();
}

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxConv0Params(void) { return (int)sizeof(SmxConv0Params); }
