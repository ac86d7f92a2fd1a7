// MFMA forms of the three heavy kernels of conv0.hip (bf16 activations, GroupNorm mode, C a multiple of 128, 3 k <= 32) -
// included by conv0.hip.
//
// What bounded the vector-unit forms was never memory: per output element they issue the 10-tap convolution (10 FMAs),
// exact-erf GELU (~14 ops) and, in backward, another 10 FMAs for the weight gradient, and a wave64 VALU instruction costs
// 4 cycles on this chip - the backward ran at 1.4 TB/s of its 8 TB/s roofline.  Here the two contractions go to the matrix
// pipe and only the GroupNorm / GELU arithmetic stays on the vector unit:
//   u[t, c]   = sum_tap x[s t + tap] w[c, tap]       -> one v_mfma_f32_16x16x32_bf16 per 16 t x 16 c block.  K = 32 holds
//               three 10-wide groups  x_hi w_hi | x_lo w_hi | x_hi w_lo  (x = x_hi + x_lo, w = w_hi + w_lo in bf16), so u keeps
//               ~16 bits of each factor: the fp32 waveform is NOT rounded to bf16 at the first layer;
//   G[tap, c] = sum_t x[s t + tap] dz[t, c]          -> one MFMA per 32 t x 16 c (dz already sits in the accumulator layout
//               of u: with the "pair" K-slot mapping of the attention kernels it IS the B operand, no transposition).
// Layout: block = one clip x a run of 64-step time tiles (the vector kernels' geometry and partial-row formats, so their
// finalize kernels are reused); wave w owns channels [128 w, 128 w + 128): 8 MFMA column blocks with column i of block j =
// channel 128 w + 8 i + j, so a lane ends up with 8 CONSECUTIVE channels of a time step - one 16-byte load of dy / store of y.
#pragma once

#define C0M_OK(p) ((p).group && ((p).C % 128) == 0 && (p).C <= 512 && 3 * (p).k <= 32 && !(p).cbias)

struct C0MCtx {
    unsigned b1[8][4];        // W fragments (B operand of the u product), per channel block
    float a[8], b0[8], rs[8], xo[8];      // GroupNorm of my 8 channels: z = a u + b0, xhat = rs u + xo
    int tapE[8], kindE[8];    // per K slot of my k-block: tap, group (0 x_hi w_hi, 1 x_lo w_hi, 2 x_hi w_lo, 3 unused)
};

__device__ __forceinline__ bf16x8_t c0m_pack_pair(const f32x4_t& a, const f32x4_t& b) {
    union { bf16x8_t v; unsigned u[4]; } f;
    f.u[0] = pack_bf2(a[0], a[1]); f.u[1] = pack_bf2(a[2], a[3]);
    f.u[2] = pack_bf2(b[0], b[1]); f.u[3] = pack_bf2(b[2], b[3]);
    return f.v;
}

__device__ __forceinline__ float c0m_bf_hi(float v) { return __uint_as_float(pack_bf2(v, 0.f) << 16); }

// K slot e of k-block g is slot 8 g + e of the 32: group = slot / k, tap = slot % k (computed once per lane)
__device__ __forceinline__ void c0m_slots(const SmxConv0Params& p, int g, int (&tap)[8], int (&kind)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int s = 8 * g + e;
        const int grp = s / p.k;
        kind[e] = grp < 3 ? grp : 3;
        tap[e] = grp < 3 ? s - grp * p.k : 0;
    }
}

__device__ __forceinline__ void c0m_load_w(const SmxConv0Params& p, C0MCtx& cx, int wave, int lane) {
    const int i = lane & 15, g = lane >> 4;
    c0m_slots(p, g, cx.tapE, cx.kindE);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = wave * 128 + 8 * i + j;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kd = cx.kindE[2 * pr + e];
                float wv = (kd < 3 && c < p.C) ? p.w[c * p.k + cx.tapE[2 * pr + e]] : 0.f;
                if (kd == 2) wv -= c0m_bf_hi(wv);           // w_lo
                v[e] = wv;
            }
            cx.b1[j][pr] = pack_bf2(v[0], v[1]);
        }
    }
}

__device__ __forceinline__ void c0m_load_norm(const SmxConv0Params& p, C0MCtx& cx, int b, int wave, int lane) {
    const int i = lane & 15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = wave * 128 + 8 * i + j;
        const double s = p.stats[((long long)b * p.C + c) * 2], q = p.stats[((long long)b * p.C + c) * 2 + 1];
        const double m = s / p.T0;
        double var = q / p.T0 - m * m;
        if (var < 0) var = 0;
        const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
        const float gm = p.gamma[c];
        cx.rs[j] = rstd;
        cx.xo[j] = -mean * rstd;
        cx.a[j] = rstd * gm;
        cx.b0[j] = p.beta[c] - mean * rstd * gm;
    }
}

// A operand of the u product for time block tb of the staged tile: lane (row = t = 16 tb + i, k-block g)
__device__ __forceinline__ bf16x8_t c0m_xfrag(const SmxConv0Params& p, const C0MCtx& cx, const float* sx, int tb, int lane) {
    const int i = lane & 15;
    const float* x = sx + (16 * tb + i) * p.stride;
    union { bf16x8_t v; unsigned u[4]; } f;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int kd = cx.kindE[2 * pr + e];
            float xv = x[cx.tapE[2 * pr + e]];              // (unused slots meet zero weights: any finite value will do)
            if (kd == 1) xv -= c0m_bf_hi(xv);               // x_lo
            v[e] = xv;
        }
        f.u[pr] = pack_bf2(v[0], v[1]);
    }
    return f.v;
}

// A operand of the G product for the time-block pair (ta, tb): lane (row = tap i, k-block g), K slot e < 4 -> t = 16 ta + 4 g + e,
// e >= 4 -> t = 16 tb + 4 g + e - 4 (the slot order in which pack_pair lays out two accumulator blocks)
__device__ __forceinline__ bf16x8_t c0m_xtfrag(const SmxConv0Params& p, const float* sx, int ta, int tb, int lane) {
    const int i = lane & 15, g = lane >> 4;
    union { bf16x8_t v; unsigned u[4]; } f;
    const float* xa = sx + (16 * ta + 4 * g) * p.stride + i;
    const float* xb = sx + (16 * tb + 4 * g) * p.stride + i;
    f.u[0] = pack_bf2(xa[0], xa[p.stride]);
    f.u[1] = pack_bf2(xa[2 * p.stride], xa[3 * p.stride]);
    f.u[2] = pack_bf2(xb[0], xb[p.stride]);
    f.u[3] = pack_bf2(xb[2 * p.stride], xb[3 * p.stride]);
    return f.v;
}

__device__ __forceinline__ bf16x8_t c0m_wfrag(const C0MCtx& cx, int j) {
    union { bf16x8_t v; unsigned u[4]; } f;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) f.u[pr] = cx.b1[j][pr];
    return f.v;
}

// sum over the four lane groups (same lane & 15): permlane swaps, no LDS
__device__ __forceinline__ float c0m_gsum(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned ua = __float_as_uint(a);
    auto r2 = __builtin_amdgcn_permlane32_swap(ua, ua, false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

#define C0M_ZERO4 ((f32x4_t){0.f, 0.f, 0.f, 0.f})

// ---- pass 1: per-(clip, channel) sum / sum of squares of u; one partial row [C][2] per block ---------------------------
__global__ __launch_bounds__(256, 2) void c0m_stats_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const bool active = wave * 128 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    C0MCtx cx;
    if (active) c0m_load_w(p, cx, wave, lane);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (!active) continue;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const bf16x8_t xa = c0m_xfrag(p, cx, sx, tb, lane);
            bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) ok[r] = t0 + 16 * tb + 4 * g + r < p.T0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4_t u = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, c0m_wfrag(cx, j), C0M_ZERO4, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = ok[r] ? u[r] : 0.f;
                    s[j] += v;
                    q[j] = fmaf(v, v, q[j]);
                }
            }
        }
    }
    if (active) {
        const int i = lane & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float ss = c0m_gsum(s[j]), qq = c0m_gsum(q[j]);
            if (g == 0) {
                float* dst = p.partials + (((long long)b * p.nb + blockIdx.x) * p.C + wave * 128 + 8 * i + j) * 2;
                dst[0] = ss;
                dst[1] = qq;
            }
        }
    }
}

// ---- pass 2: y = GELU(gamma (u - mean) rstd + beta) ----------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void c0m_apply_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const bool active = wave * 128 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    bf16_t* Y = reinterpret_cast<bf16_t*>(p.y) + (long long)b * p.T0 * p.C + wave * 128 + 8 * i;
    C0MCtx cx;
    if (active) {
        c0m_load_w(p, cx, wave, lane);
        c0m_load_norm(p, cx, b, wave, lane);
    }
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (!active) continue;
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
            const bf16x8_t xa = c0m_xfrag(p, cx, sx, tb, lane);
            float y[4][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4_t u = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, c0m_wfrag(cx, j), C0M_ZERO4, 0, 0, 0);
                const smx_f2 z01 = {fmaf(u[0], cx.a[j], cx.b0[j]), fmaf(u[1], cx.a[j], cx.b0[j])};
                const smx_f2 z23 = {fmaf(u[2], cx.a[j], cx.b0[j]), fmaf(u[3], cx.a[j], cx.b0[j])};
                const smx_f2 y01 = gelu2(z01), y23 = gelu2(z23);
                y[0][j] = y01[0]; y[1][j] = y01[1]; y[2][j] = y23[0]; y[3][j] = y23[1];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = t0 + 16 * tb + 4 * g + r;
                if (t < p.T0)
                    *reinterpret_cast<uint4*>(Y + (long long)t * p.C) =
                        make_uint4(pack_bf2(y[r][0], y[r][1]), pack_bf2(y[r][2], y[r][3]), pack_bf2(y[r][4], y[r][5]), pack_bf2(y[r][6], y[r][7]));
            }
        }
    }
}

// ---- backward, the one pass over dy: per (clip, channel) G[tap] = sum dz x_tap, S1 = sum dz, S2 = sum dz xhat -------------
__global__ __launch_bounds__(256, 2) void c0m_bwd_kernel(SmxConv0Params p) {
    __shared__ float sx[C0_TT * 8 + C0_MAXK];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const bool active = wave * 128 < p.C;
    const int ntiles = (p.T0 + C0_TT - 1) / C0_TT;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * p.tiles_per_block);
    const bf16_t* dY = reinterpret_cast<const bf16_t*>(p.dy) + (long long)b * p.T0 * p.C + wave * 128 + 8 * i;
    C0MCtx cx;
    if (active) {
        c0m_load_w(p, cx, wave, lane);
        c0m_load_norm(p, cx, b, wave, lane);
    }
    float s1[8], s2[8];
    f32x4_t G[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = s2[j] = 0.f; G[j] = C0M_ZERO4; }
    for (int tile = blockIdx.x * p.tiles_per_block; tile < tile_end; ++tile) {
        const int t0 = tile * C0_TT;
        __syncthreads();
        stage_wave(p, sx, b, t0);
        __syncthreads();
        if (!active) continue;
#pragma unroll 1
        for (int pr = 0; pr < 2; ++pr) {          // time-block pairs (0, 1) and (2, 3)
            const int ta = 2 * pr, tb = 2 * pr + 1;
            uint4 d[2][4];                        // dy rows of my 8 time steps (8 channels each), requested before the MFMAs
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + 16 * (ta + h) + 4 * g + r;
                    d[h][r] = t < p.T0 ? *reinterpret_cast<const uint4*>(dY + (long long)t * p.C) : make_uint4(0, 0, 0, 0);
                }
            const bf16x8_t xa0 = c0m_xfrag(p, cx, sx, ta, lane), xa1 = c0m_xfrag(p, cx, sx, tb, lane);
            const bf16x8_t xt = c0m_xtfrag(p, sx, ta, tb, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8_t wf = c0m_wfrag(cx, j);
                f32x4_t u[2];
                u[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa0, wf, C0M_ZERO4, 0, 0, 0);
                u[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa1, wf, C0M_ZERO4, 0, 0, 0);
                f32x4_t dz[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const smx_f2 uu = {u[h][2 * r2], u[h][2 * r2 + 1]};
                        const smx_f2 z = __builtin_elementwise_fma(uu, SMX_PK(cx.a[j]), SMX_PK(cx.b0[j]));
                        const unsigned w0 = (&d[h][2 * r2].x)[j >> 1], w1 = (&d[h][2 * r2 + 1].x)[j >> 1];
                        const smx_f2 dyv = {(j & 1) ? __uint_as_float(w0 & 0xffff0000u) : __uint_as_float(w0 << 16),
                                            (j & 1) ? __uint_as_float(w1 & 0xffff0000u) : __uint_as_float(w1 << 16)};
                        const smx_f2 dzz = dyv * gelu_grad2(z);
                        const smx_f2 xh = __builtin_elementwise_fma(uu, SMX_PK(cx.rs[j]), SMX_PK(cx.xo[j]));
                        s1[j] += dzz[0] + dzz[1];
                        s2[j] = fmaf(dzz[0], xh[0], fmaf(dzz[1], xh[1], s2[j]));
                        dz[h][2 * r2] = dzz[0];
                        dz[h][2 * r2 + 1] = dzz[1];
                    }
                G[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xt, c0m_pack_pair(dz[0], dz[1]), G[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // one channel block at a time: interleaving all eight spills
            }
        }
    }
    if (active) {
        float* row = p.partials + ((long long)b * p.nb + blockIdx.x) * ((long long)p.C * (p.k + 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = wave * 128 + 8 * i + j;
            float* rr = row + (long long)c * (p.k + 2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < p.k) rr[4 * g + r] = G[j][r];          // accumulator rows = taps 4 g + r
            const float a1 = c0m_gsum(s1[j]), a2 = c0m_gsum(s2[j]);
            if (g == 0) {
                rr[p.k] = a1;
                rr[p.k + 1] = a2;
            }
        }
    }
}
