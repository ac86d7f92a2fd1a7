// Layer 0 of the wav2vec2 / HuBERT feature extractor: Conv1d(1 -> C, k, stride) on the raw waveform,
// fused with GroupNorm(C groups: per-(clip,channel) statistics over time) + affine + GELU
// (TF:models/wav2vec2/modeling_wav2vec2.py:301-323), or emitted plain for the "layer" extractor variant
// (conv(+bias) -> LayerNorm over C -> GELU, :275-299; the LN+GELU is smx_norm_fwd with act=GELU).
//
// C_in = 1, so the 10-tap conv is recomputed from the waveform (0.64 MB / 10 s clip, LDS-staged, broadcast reads)
// instead of ever being stored; the only large traffic is the channels-last activation [B, T0, C] (32.8 MB / clip in
// bf16), written once (forward) and read twice (backward).  What is left is VALU work - 10 FMAs + GroupNorm + GELU per
// output element - so the kernels are built for the vector unit:
//   * thread <-> one PAIR of channels, every op a packed-fp32 instruction (v_pk_fma_f32: two channels per issue);
//     the whole block walks the same time steps, so the waveform window is an LDS broadcast;
//   * exact tap count (template K = 10; a zero-padded 16-tap instance covers other kernels);
//   * 20 weight + 20 gradient-accumulator registers per thread instead of 256: high occupancy, no spills;
//   * reductions over time are register-resident per block and leave the block as ONE partial row
//     (no atomics - device-scope atomics serialise in L2); a second tiny pass sums the partial rows;
//   * the GroupNorm backward reads dy ONCE: the weight gradient through the normalisation,
//       dW[c][t] = sum_b a_bc ( G_bc[t] - m1_bc X1_b[t] - m2_bc Q_bc[t] ),   G = sum dz x_t,  m1 = mean dz,  m2 = mean dz xhat,
//       Q_bc[t] = sum xhat x_t = rstd_bc ( cb_c X1_b[t] + sum_s w[c][s] R_b[s][t] ) - mean_bc rstd_bc X1_b[t],
//     needs from the pass over dy only G, sum dz and sum dz xhat per (clip, channel); X1_b[t] = sum x_t and the k x k
//     autocorrelation R_b[s][t] = sum x_s x_t (x_t = the waveform at stride * t' + t) do not depend on the channel and come
//     from a pass over the 0.64-MB waveform.
#include "smx_common.h"

#define C0_MAXK 16
#define C0_TT 64        // time steps per LDS stage
#define C0_NBMAX 64     // reduction kernels: at most this many blocks (partial rows) per clip
#define C0_CH 8         // backward: rows of dy requested ahead of their use

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
    float* dw;            // [C, k] fp32 (accumulated)
    float* dcbias;        // [C] or null (plain mode)
    float* dgamma;        // [C]
    float* dbeta;
    int B, N, C, k, stride, T0;
    int group;            // 1: GroupNorm+GELU fused, 0: plain conv
    float eps;
    int tiles_per_block;  // set by the launchers
    float* partials;      // workspace, >= smx_conv0_workspace_floats(B, C, k) floats (reduction kernels)
    int nb;               // set by the launchers: blocks per clip in the reduction kernels
};

extern "C" int smx_colsum(const void* x, float* out, int M, int N, long long ld, float alpha, int dtype, hipStream_t stream);

// workspace: partial rows of the reduction kernels | per-block waveform correlations | per-clip gradient contributions
extern "C" long long smx_conv0_workspace_floats(int B, int C, int k) {
    return (long long)B * C0_NBMAX * C * (k + 2) + (long long)B * C0_NBMAX * (k * k + k) + (long long)B * C * (k + 2);
}

template <int K>
struct C0Thread {
    smx_f2 w[K];
    smx_f2 cb;
    __device__ __forceinline__ void load(const SmxConv0Params& p, int c0) {
        cb = p.cbias ? (smx_f2){p.cbias[c0], p.cbias[c0 + 1]} : SMX_PK(0.f);
#pragma unroll
        for (int t = 0; t < K; ++t) w[t] = t < p.k ? (smx_f2){p.w[c0 * p.k + t], p.w[(c0 + 1) * p.k + t]} : SMX_PK(0.f);
    }
    __device__ __forceinline__ smx_f2 conv(const float* x) const {
        smx_f2 u = cb;
#pragma unroll
        for (int t = 0; t < K; ++t) u = __builtin_elementwise_fma(w[t], SMX_PK(x[t]), u);
        return u;
    }
};

__device__ __forceinline__ void stage_wave(const SmxConv0Params& p, float* sx, int b, int t0) {
    const int n0 = t0 * p.stride, cnt = C0_TT * p.stride + C0_MAXK;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int n = n0 + i;
        sx[i] = n < p.N ? p.wave[(long long)b * p.N + n] : 0.f;
    }
}
// GroupNorm of channel pair (c0, c0+1) of clip b as z = u * a + b0, xhat = u * rs + xo
struct C0Norm {
    smx_f2 a, b0, rs, xo;
    __device__ __forceinline__ void load(const SmxConv0Params& p, int b, int c0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double s = p.stats[((long long)b * p.C + c0 + j) * 2], q = p.stats[((long long)b * p.C + c0 + j) * 2 + 1];
            const double m = s / p.T0;
            double var = q / p.T0 - m * m;
            if (var < 0) var = 0;
            const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
            const float g = p.gamma[c0 + j];
            rs[j] = rstd;
            xo[j] = -mean * rstd;
            a[j] = rstd * g;
            b0[j] = p.beta[c0 + j] - mean * rstd * g;
        }
    }
};
__device__ __forceinline__ smx_f2 gelu2(smx_f2 z) {
    smx_f2 er, E;
    smx_erf_e2(z, er, E);
    const smx_f2 hz = z * SMX_PK(0.5f);
    return __builtin_elementwise_fma(hz, er, hz);
}
__device__ __forceinline__ smx_f2 gelu_grad2(smx_f2 z) {
    smx_f2 er, E;
    smx_erf_e2(z, er, E);
    const smx_f2 cdf = __builtin_elementwise_fma(er, SMX_PK(0.5f), SMX_PK(0.5f));
    return __builtin_elementwise_fma(z * SMX_PK(0.39894228040143267794f), E, cdf);
}
__device__ __forceinline__ smx_f2 load_pair(const float* p) { const float2 v = *reinterpret_cast<const float2*>(p); return (smx_f2){v.x, v.y}; }
__device__ __forceinline__ smx_f2 load_pair(const bf16_t* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return (smx_f2){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
__device__ __forceinline__ void store_pair(float* p, smx_f2 v) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
__device__ __forceinline__ void store_pair(bf16_t* p, smx_f2 v) { *reinterpret_cast<unsigned*>(p) = pack_bf2(v[0], v[1]); }

// pass 1 (group mode): per-(b,c) sum / sumsq of u = conv(x): one partial row [C][2] per block
template <int K>
__global__ __launch_bounds__(256) void conv0_stats_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, c0 = threadIdx.x * 2;
    const bool active = c0 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    C0Thread<K> th;
    if (active) th.load(p, c0);
    smx_f2 s = SMX_PK(0.f), q = SMX_PK(0.f);
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (active) {
            const int ntt = min(C0_TT, p.T0 - t0);
            smx_f2 st = SMX_PK(0.f), qt = SMX_PK(0.f);          // per-tile sums, then a second level (rounding)
            for (int tt = 0; tt < ntt; ++tt) {
                const smx_f2 u = th.conv(sx + tt * p.stride);
                st += u;
                qt = __builtin_elementwise_fma(u, u, qt);
            }
            s += st;
            q += qt;
        }
    }
    if (active)
        *reinterpret_cast<float4*>(p.partials + (((long long)b * p.nb + blockIdx.x) * p.C + c0) * 2) = make_float4(s[0], q[0], s[1], q[1]);
}
// dst[b][c][0..1] (fp64) = sum over the clip's partial rows
__global__ void conv0_stats_finalize_kernel(const float* __restrict__ partials, double* __restrict__ dst, int B, int C, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (b, c, j)
    if (i >= B * C * 2) return;
    const int b = i / (C * 2), r = i - b * C * 2;
    double a = 0;
    for (int k = 0; k < nb; ++k) a += (double)partials[((long long)b * nb + k) * C * 2 + r];
    dst[i] = a;
}

// pass 2: y = GELU(gamma * (u - mean) * rstd + beta)  (group)   or   y = u  (plain)
template <typename T, int K>
__global__ __launch_bounds__(256) void conv0_apply_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, c0 = threadIdx.x * 2;
    const bool active = c0 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    T* Y = reinterpret_cast<T*>(p.y) + (long long)b * p.T0 * p.C + c0;
    C0Thread<K> th;
    C0Norm nm;
    if (active) {
        th.load(p, c0);
        if (p.group) nm.load(p, b, c0);
    }
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (active) {
            const int ntt = min(C0_TT, p.T0 - t0);
#pragma unroll 2
            for (int tt = 0; tt < ntt; ++tt) {
                smx_f2 u = th.conv(sx + tt * p.stride);
                if (p.group) u = gelu2(__builtin_elementwise_fma(u, nm.a, nm.b0));
                store_pair(Y + (long long)(t0 + tt) * p.C, u);
            }
        }
    }
}

// backward (group), the one pass over dy: per (clip, channel) G[t] = sum dz x_t, S1 = sum dz, S2 = sum dz xhat with
// dz = dy * gelu'(z).  The block's sums stay in registers over all its time steps and leave as one partial row
// [C][k + 2] = G[0..k) | S1 | S2.
template <typename T, int K>
__global__ __launch_bounds__(256) void conv0_bwd_group_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, c0 = threadIdx.x * 2;
    const bool active = c0 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    const T* dY = reinterpret_cast<const T*>(p.dy) + (long long)b * p.T0 * p.C + c0;
    C0Thread<K> th;
    C0Norm nm;
    smx_f2 s1 = SMX_PK(0.f), s2 = SMX_PK(0.f), acc[K];
#pragma unroll
    for (int t = 0; t < K; ++t) acc[t] = SMX_PK(0.f);
    if (active) {
        th.load(p, c0);
        nm.load(p, b, c0);
    }
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (active) {
            const int ntt = min(C0_TT, p.T0 - t0);
            smx_f2 a1 = SMX_PK(0.f), a2 = SMX_PK(0.f);          // per-tile sums, then a second level (rounding)
            // C0_CH rows of dy are requested before the first is used: one 256-B row per wave and time step, so the
            // bytes in flight (not the arithmetic) set this kernel's speed.  Steps past the end read dy as zero (their x
            // window is staged, zero-filled), so they add nothing.
            for (int tc = 0; tc < ntt; tc += C0_CH) {
                smx_f2 d[C0_CH];
#pragma unroll
                for (int j = 0; j < C0_CH; ++j)
                    d[j] = tc + j < ntt ? load_pair(dY + (long long)(t0 + tc + j) * p.C) : SMX_PK(0.f);
#pragma unroll
                for (int j = 0; j < C0_CH; ++j) {
                    const float* x = sx + (tc + j) * p.stride;
                    const smx_f2 u = th.conv(x);
                    const smx_f2 dz = d[j] * gelu_grad2(__builtin_elementwise_fma(u, nm.a, nm.b0));
                    const smx_f2 xh = __builtin_elementwise_fma(u, nm.rs, nm.xo);
                    a1 += dz;
                    a2 = __builtin_elementwise_fma(dz, xh, a2);
#pragma unroll
                    for (int t = 0; t < K; ++t) acc[t] = __builtin_elementwise_fma(dz, SMX_PK(x[t]), acc[t]);
                }
            }
            s1 += a1;
            s2 += a2;
        }
    }
    if (active) {
        float* row = p.partials + ((long long)b * p.nb + blockIdx.x) * ((long long)p.C * (p.k + 2));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* r = row + (long long)(c0 + j) * (p.k + 2);
#pragma unroll
            for (int t = 0; t < K; ++t)
                if (t < p.k) r[t] = acc[t][j];
            r[p.k] = s1[j];
            r[p.k + 1] = s2[j];
        }
    }
}

// waveform correlations of the block's time steps: R[s][t] = sum x[stride t' + s] x[stride t' + t] (k x k), X1[t] = sum x[stride t' + t]
#define C0_XT 8         // K tiles of time steps staged per round (this kernel has almost no arithmetic: its time is the staging)
__global__ __launch_bounds__(256) void conv0_xcorr_kernel(SmxConv0Params p, float* __restrict__ xpart) {
    __shared__ float sx[C0_XT * C0_TT * 8 + C0_MAXK];
    __shared__ float red[256];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int xn = p.k * p.k + p.k;               // <= 128 entries, each summed by two threads (even / odd time steps)
    const int e = tid & 127, part = tid >> 7;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    const bool pairs = e < p.k * p.k;
    const int s = pairs ? e / p.k : 0, t = pairs ? e - s * p.k : min(e - p.k * p.k, p.k - 1);
    float acc = 0.f;
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; tile += C0_XT) {
        const int t0 = tile * C0_TT;
        const int span = min(C0_XT, tile_end - tile) * C0_TT, steps = min(span, p.T0 - t0);
        __syncthreads();
        for (int i = tid; i < span * p.stride + C0_MAXK; i += 256) {
            const int n = t0 * p.stride + i;
            sx[i] = n < p.N ? p.wave[(long long)b * p.N + n] : 0.f;
        }
        __syncthreads();
        if (e < xn) {
            float a = 0.f;
#pragma unroll 8
            for (int tt = part; tt < span; tt += 2) {       // (windows past the clip's last step are staged: read, not counted)
                const float* x = sx + tt * p.stride;
                const float v = pairs ? x[s] * x[t] : x[t];
                a += tt < steps ? v : 0.f;
            }
            acc += a;
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < xn) xpart[((long long)b * p.nb + blockIdx.x) * xn + tid] = red[tid] + red[tid + 128];
}

// per (clip, 16 channels): partial rows -> S1, S2, G (fp64), then this clip's contribution to dW through the GroupNorm
// (see the file header), written with S1 / S2 as one row [C k | C (dbeta) | C (dgamma)] of `contrib`; bstats gets S1, S2.
__global__ __launch_bounds__(256) void conv0_bwd_group_finalize_kernel(SmxConv0Params p, const float* __restrict__ xpart,
                                                                       float* __restrict__ contrib) {
    __shared__ double xs[C0_MAXK * C0_MAXK + C0_MAXK];
    __shared__ double sv[16][16];
    const int b = blockIdx.y, tid = threadIdx.x, cl = tid >> 4, slot = tid & 15;
    const int c = blockIdx.x * 16 + cl, k = p.k, xn = k * k + k;
    if (tid < xn) {
        double a = 0;
#pragma unroll 8
        for (int j = 0; j < p.nb; ++j) a += (double)xpart[((long long)b * p.nb + j) * xn + tid];
        xs[tid] = a;
    }
    double v = 0;
    if (c < p.C && slot < k + 2)
#pragma unroll 8
        for (int j = 0; j < p.nb; ++j) v += (double)p.partials[((long long)b * p.nb + j) * ((long long)p.C * (k + 2)) + (long long)c * (k + 2) + slot];
    sv[cl][slot] = v;
    __syncthreads();
    if (c >= p.C) return;
    float* row = contrib + (long long)b * p.C * (k + 2);
    if (slot < k) {
        const double S1 = sv[cl][k], S2 = sv[cl][k + 1];
        const double sum = p.stats[((long long)b * p.C + c) * 2], sq = p.stats[((long long)b * p.C + c) * 2 + 1];
        const double mean = sum / p.T0;
        double var = sq / p.T0 - mean * mean;
        if (var < 0) var = 0;
        const double rstd = 1.0 / sqrt(var + (double)p.eps);
        const double a = rstd * (double)p.gamma[c], m1 = S1 / p.T0, m2 = S2 / p.T0;
        const double x1 = xs[k * k + slot];
        double ux = p.cbias ? (double)p.cbias[c] * x1 : 0.0;
        for (int s = 0; s < k; ++s) ux += (double)p.w[c * k + s] * xs[s * k + slot];
        const double q = rstd * ux - mean * rstd * x1;
        row[c * k + slot] = (float)(a * (v - m1 * x1 - m2 * q));
    } else if (slot == k) {
        row[p.C * k + c] = (float)v;
        p.bstats[((long long)b * p.C + c) * 2] = v;
    } else if (slot == k + 1) {
        row[p.C * k + p.C + c] = (float)v;
        p.bstats[((long long)b * p.C + c) * 2 + 1] = v;
    }
}

// backward, plain mode (dy is du): dW[c][t] += sum du * x[stride*t' + t].  The block's sums stay in registers over all
// its time steps and leave as one partial row [C*k | C]; smx_colsum adds the rows into dw / dcbias.
template <typename T, int K>
__global__ __launch_bounds__(256) void conv0_bwd_w_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, c0 = threadIdx.x * 2;
    const bool active = c0 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    const T* dY = reinterpret_cast<const T*>(p.dy) + (long long)b * p.T0 * p.C + c0;
    smx_f2 accb = SMX_PK(0.f), acc[K];
#pragma unroll
    for (int t = 0; t < K; ++t) acc[t] = SMX_PK(0.f);
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (active) {
            const int ntt = min(C0_TT, p.T0 - t0);
            for (int tc = 0; tc < ntt; tc += C0_CH) {       // (loads batched as in conv0_bwd_group_kernel)
                smx_f2 d[C0_CH];
#pragma unroll
                for (int j = 0; j < C0_CH; ++j)
                    d[j] = tc + j < ntt ? load_pair(dY + (long long)(t0 + tc + j) * p.C) : SMX_PK(0.f);
#pragma unroll
                for (int j = 0; j < C0_CH; ++j) {
                    const float* x = sx + (tc + j) * p.stride;
                    accb += d[j];
#pragma unroll
                    for (int t = 0; t < K; ++t) acc[t] = __builtin_elementwise_fma(d[j], SMX_PK(x[t]), acc[t]);
                }
            }
        }
    }
    if (active) {
        float* row = p.partials + ((long long)b * p.nb + blockIdx.x) * ((long long)p.C * (p.k + 1));
#pragma unroll
        for (int t = 0; t < K; ++t)
            if (t < p.k) {
                row[c0 * p.k + t] = acc[t][0];
                row[(c0 + 1) * p.k + t] = acc[t][1];
            }
        store_pair(row + (long long)p.C * p.k + c0, accb);
    }
}

#include "conv0_mfma.h"

static bool conv0_use_mfma(const SmxConv0Params& p, int dtype) {
    static const bool off = getenv("SMX_CONV0_MFMA") && getenv("SMX_CONV0_MFMA")[0] == '0';      // A/B switch
    return !off && dtype == SMX_BF16 && C0M_OK(p);
}

static int conv0_check(const SmxConv0Params& p) {
    if (p.B <= 0 || p.C <= 0 || (p.C & 7) || p.C > 512 || p.k > C0_MAXK || p.k <= 0 || p.stride <= 0 || p.stride > 8) return SMX_EINVAL;
    if (p.T0 != (p.N - p.k) / p.stride + 1 || p.T0 <= 0) return SMX_EINVAL;
    return SMX_OK;
}
static void conv0_reduction_geometry(SmxConv0Params& p) {
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    int nb = 1024 / p.B;                       // ~1024 blocks in flight (4 per CU)
    if (nb < 1) nb = 1;
    if (nb > C0_NBMAX) nb = C0_NBMAX;
    if (nb > ntiles) nb = ntiles;
    p.tiles_per_block = (ntiles + nb - 1) / nb;
    p.nb = (ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
}
#define C0_LAUNCH_K(KERNEL, GRID, P, STREAM)                                                       \
    do {                                                                                           \
        if ((P).k == 10) hipLaunchKernelGGL((KERNEL<10>), GRID, dim3(256), 0, STREAM, P);          \
        else hipLaunchKernelGGL((KERNEL<C0_MAXK>), GRID, dim3(256), 0, STREAM, P);                 \
    } while (0)
#define C0_LAUNCH_TK(KERNEL, T, GRID, P, STREAM)                                                   \
    do {                                                                                           \
        if ((P).k == 10) hipLaunchKernelGGL((KERNEL<T, 10>), GRID, dim3(256), 0, STREAM, P);       \
        else hipLaunchKernelGGL((KERNEL<T, C0_MAXK>), GRID, dim3(256), 0, STREAM, P);              \
    } while (0)

extern "C" int smx_conv0_fwd(const SmxConv0Params* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxConv0Params p = *pp;
    int rc = conv0_check(p);
    if (rc) return rc;
    if (dtype != SMX_F32 && dtype != SMX_BF16) return SMX_EINVAL;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    if (p.group) {
        if (!p.stats || !p.gamma || !p.beta || !p.partials) return SMX_EINVAL;
        conv0_reduction_geometry(p);
        if (conv0_use_mfma(p, dtype)) hipLaunchKernelGGL(c0m_stats_kernel, dim3(p.nb, p.B), dim3(256), 0, stream, p);
        else C0_LAUNCH_K(conv0_stats_kernel, dim3(p.nb, p.B), p, stream);
        const int n = p.B * p.C * 2;
        hipLaunchKernelGGL(conv0_stats_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p.partials, p.stats, p.B, p.C, p.nb);
    }
    p.tiles_per_block = 4;
    const dim3 grid((ntiles + 3) / 4, p.B);
    if (p.group && conv0_use_mfma(p, dtype)) {
        hipLaunchKernelGGL(c0m_apply_kernel, grid, dim3(256), 0, stream, p);
        SMX_CHECK_LAUNCH();
    }
    if (dtype == SMX_F32) C0_LAUNCH_TK(conv0_apply_kernel, float, grid, p, stream);
    else C0_LAUNCH_TK(conv0_apply_kernel, bf16_t, grid, p, stream);
    SMX_CHECK_LAUNCH();
}

// group mode: needs `stats` from the forward; accumulates dw, dgamma, dbeta.  plain mode: dy is du.
extern "C" int smx_conv0_bwd(const SmxConv0Params* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxConv0Params p = *pp;
    int rc = conv0_check(p);
    if (rc) return rc;
    if (dtype != SMX_F32 && dtype != SMX_BF16) return SMX_EINVAL;
    if (!p.partials) return SMX_EINVAL;
    conv0_reduction_geometry(p);
    const dim3 grid(p.nb, p.B);
    if (p.group) {
        if (!p.stats || !p.bstats || !p.gamma) return SMX_EINVAL;
        float* xpart = p.partials + (long long)p.B * C0_NBMAX * p.C * (p.k + 2);
        float* contrib = xpart + (long long)p.B * C0_NBMAX * (p.k * p.k + p.k);
        hipLaunchKernelGGL(conv0_xcorr_kernel, grid, dim3(256), 0, stream, p, xpart);
        if (conv0_use_mfma(p, dtype)) hipLaunchKernelGGL(c0m_bwd_kernel, grid, dim3(256), 0, stream, p);
        else if (dtype == SMX_F32) C0_LAUNCH_TK(conv0_bwd_group_kernel, float, grid, p, stream);
        else C0_LAUNCH_TK(conv0_bwd_group_kernel, bf16_t, grid, p, stream);
        hipLaunchKernelGGL(conv0_bwd_group_finalize_kernel, dim3((p.C + 15) / 16, p.B), dim3(256), 0, stream, p, xpart, contrib);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        const long long ld = (long long)p.C * (p.k + 2);
        if (p.dw) {
            rc = smx_colsum(contrib, p.dw, p.B, p.C * p.k, ld, 1.0f, SMX_F32, stream);
            if (rc) return rc;
        }
        if (p.dbeta) {
            rc = smx_colsum(contrib + (long long)p.C * p.k, p.dbeta, p.B, p.C, ld, 1.0f, SMX_F32, stream);
            if (rc) return rc;
        }
        if (p.dgamma) {
            rc = smx_colsum(contrib + (long long)p.C * p.k + p.C, p.dgamma, p.B, p.C, ld, 1.0f, SMX_F32, stream);
            if (rc) return rc;
        }
        SMX_CHECK_LAUNCH();
    }
    if (p.dw) {
        if (dtype == SMX_F32) C0_LAUNCH_TK(conv0_bwd_w_kernel, float, grid, p, stream);
        else C0_LAUNCH_TK(conv0_bwd_w_kernel, bf16_t, grid, p, stream);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        const long long ld = (long long)p.C * (p.k + 1);
        rc = smx_colsum(p.partials, p.dw, p.B * p.nb, p.C * p.k, ld, 1.0f, SMX_F32, stream);
        if (rc) return rc;
        if (p.dcbias) {
            rc = smx_colsum(p.partials + (long long)p.C * p.k, p.dcbias, p.B * p.nb, p.C, ld, 1.0f, SMX_F32, stream);
            if (rc) return rc;
        }
    }
    SMX_CHECK_LAUNCH();
}

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxConv0Params(void) { return (int)sizeof(SmxConv0Params); }
