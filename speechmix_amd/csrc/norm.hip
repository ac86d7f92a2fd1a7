// LayerNorm / RMSNorm forward + backward (HBM-bound, one wave per row, fp32 statistics, 16-B accesses).
//   TF:models/wav2vec2/modeling_wav2vec2.py:429-434, 575-608 (LayerNorm eps=layer_norm_eps),
//   TF:models/wav2vec2/modeling_wav2vec2.py:275-299 (conv -> LayerNorm over C -> GELU, "layer" extractor),
//   TF:models/bart/modeling_bart.py:507-549 (x + positions -> layernorm_embedding),
//   TF:models/t5/modeling_t5.py:50-72 (RMS norm: no mean, no bias).
// Fusions: optional additive positional table (row r adds pos[(r % pos_period) + pos_offset]) with the
// summed input written back for backward, optional activation after the affine, optional residual
// gradient add in backward, gamma/beta gradients reduced per block and atomically added in fp32.
// Requires D % 8 == 0 and D <= 1024 (every width on the path: 512, 768, 1024; tiny test widths 32/64).
#include "smx_common.h"

#define LN_NCH 2  // 8-element chunks per lane -> D <= 64 * 8 * 2
#define LN_BLOCKS_PER_CU 3   // resident blocks per CU of the fused backward (its launch bound)
#define LN_FWD_BLOCKS_PER_CU 6   // ... of the forward (its launch bound).  15 968 x 768: 15.9 us with a block per four rows, 14.0 us with 1 536 persistent
                                 // blocks (768 / 1 024 / 2 048 / 3 072: 19.9 / 16.5 / 16.6 / 15.0); 511 968 x 512: 315 -> 257 us

struct SmxNormParams {
    const void* x;        // [M, D] input (dtype T)
    const void* pos;      // optional [*, D] table (dtype T) added to x
    void* xsum_out;       // optional [M, D]: x + pos (dtype T)
    void* y;              // [M, D]
    const float* gamma;   // [D]
    const float* beta;    // [D] or null (always null for rms)
    float* mean;          // [M] (unused for rms)
    float* rstd;          // [M]
    int M, D;
    int pos_period, pos_offset;
    int rms;              // 1: RMS norm
    int act;              // activation applied after affine
    float eps;
    float drop_p;         // dropout on the output (0: off); mask index = row * D + column
    unsigned drop_seed;
};

// One wave per row, blocks persistent over the rows (grid = one resident round of the chip, smx_norm_fwd): gamma / beta are
// loaded once per wave instead of once per row, and a wave's next row is requested while it finishes the current one.
template <typename T>
__global__ __launch_bounds__(256, LN_FWD_BLOCKS_PER_CU) void norm_fwd_kernel(SmxNormParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int lane = threadIdx.x & 63;
    float gm[LN_NCH][8], bt[LN_NCH][8];
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { gm[j][e] = 0.f; bt[j][e] = 0.f; }
        if (c < p.D) {
            load8(p.gamma + c, gm[j]);
            if (p.beta) load8(p.beta + c, bt[j]);
        }
    }
    const float invD = 1.0f / (float)p.D;
    const unsigned th = smx_thresh24(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < p.M; row += gridDim.x * 4) {
        const T* x = reinterpret_cast<const T*>(p.x) + (long long)row * p.D;
        const T* pos = p.pos ? reinterpret_cast<const T*>(p.pos) + (long long)((row % p.pos_period) + p.pos_offset) * p.D
                             : nullptr;
        T* xs = p.xsum_out ? reinterpret_cast<T*>(p.xsum_out) + (long long)row * p.D : nullptr;
        float v[LN_NCH][8];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j) {
            const int c = (lane + 64 * j) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
            if (c < p.D) {
                load8(x + c, v[j]);
                if (pos) {
                    float t[8];
                    load8(pos + c, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[j][e] = rt(v[j][e] + t[e], x);
                }
                if (xs) store8(xs + c, v[j]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[j][e];
            }
        }
        float mean = 0.f;
        if (!p.rms) mean = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j) {
            const int c = (lane + 64 * j) * 8;
            if (c < p.D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = v[j][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invD + p.eps);
        if (lane == 0) {
            if (p.mean) p.mean[row] = mean;
            if (p.rstd) p.rstd[row] = rstd;
        }
        T* y = reinterpret_cast<T*>(p.y) + (long long)row * p.D;
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j) {
            const int c = (lane + 64 * j) * 8;
            if (c < p.D) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = (v[j][e] - mean) * rstd * gm[j][e];
                    if (p.beta) t += bt[j][e];
                    t = act_fwd(t, p.act);
                    if (p.drop_p > 0.f) t *= smx_drop_mul(p.drop_seed, (unsigned)(row * p.D + c + e), th, inv_keep);
                    o[e] = t;
                }
                store8(y + c, o);
            }
        }
    }
}

struct SmxNormBwdParams {
    const void* dy;       // [M, D]
    const void* x;        // [M, D] the normalised input (x + pos if that was fused)
    const void* dres;     // optional [M, D]: added to dx (residual branch gradient)
    void* dx;             // [M, D]
    const float* gamma;
    const float* beta;    // needed only when act != none
    const float* mean;
    const float* rstd;
    float* dgamma;        // [D] fp32, accumulated atomically (or null)
    float* dbeta;         // [D] or null
    float* dpos;          // optional fp32 [*, D]: positional table gradient (atomic), same indexing as fwd
    float* partials;      // workspace [ceil(M/16)][2][D] fp32: per-block column sums, reduced by a 2nd kernel
    int M, D;
    int pos_period, pos_offset;
    int rms, act;
    float drop_p;         // the forward output was dropped: dy is multiplied by the same mask on load
    unsigned drop_seed;
    int defer_fold;       // 1: leave the gamma / beta partial rows in `partials` (smx_norm_bwd_partial_rows(M) rows of
                          // [2][D] floats); the caller reduces them into dgamma / dbeta later (smx_fold_many)
    // Optional third output (fused kernel, defer_fold == 1, no activation): dx_drop = dx * mask(drop2) - the gradient that
    // enters a Linear whose dropped output fed this norm's input (post-LN layers: out_proj / fc2) - and its column sums
    // (that Linear's bias gradient) as a THIRD partial row per block: `partials` is then [rows][3][D].
    void* dx_drop;
    float drop2_p;
    unsigned drop2_seed;
};

// Backward is two kernels: (1) dx, one wave per row at full occupancy (like the forward); (2) the gamma/beta
// gradients as per-block column partial sums over row chunks (no shuffles, no stores in the row loop),
// finished by a tiny column reduction.  Splitting costs one extra read of (x, dy) but removes the long-lived
// 32-register column accumulators from the latency-critical dx path.
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void norm_bwd_dx_kernel(SmxNormBwdParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const T* x = reinterpret_cast<const T*>(p.x) + (long long)row * p.D;
    const T* dy = reinterpret_cast<const T*>(p.dy) + (long long)row * p.D;
    const float mean = p.rms ? 0.f : p.mean[row];
    const float rstd = p.rstd[row];
    const float invD = 1.0f / (float)p.D;
    float xh[LN_NCH][8], g[LN_NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (c < p.D) {
            float xv[8], dv[8], gm[8], bt[8];
            load8(x + c, xv);
            load8(dy + c, dv);
            load8(p.gamma + c, gm);
            if (ACT) {
                if (p.beta) load8(p.beta + c, bt);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bt[e] = 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xhat = (xv[e] - mean) * rstd;
                float d = dv[e];
                if (p.drop_p > 0.f)
                    d *= smx_drop_mul(p.drop_seed, (unsigned)(row * p.D + c + e), smx_thresh24(p.drop_p), 1.0f / (1.0f - p.drop_p));
                if (ACT) d *= act_grad(xhat * gm[e] + bt[e], p.act);
                const float gg = d * gm[e];
                xh[j][e] = xhat;
                g[j][e] = gg;
                s1 += gg;
                s2 += gg * xhat;
            }
        }
    }
    s1 = p.rms ? 0.f : wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
    T* dx = reinterpret_cast<T*>(p.dx) + (long long)row * p.D;
    const T* dres = p.dres ? reinterpret_cast<const T*>(p.dres) + (long long)row * p.D : nullptr;
    float* dpos = p.dpos ? p.dpos + (long long)((row % p.pos_period) + p.pos_offset) * p.D : nullptr;
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
        if (c < p.D) {
            float o[8], r[8];
            if (dres) load8(dres + c, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (g[j][e] - s1 - xh[j][e] * s2) * rstd;
                if (dpos) atomicAdd(dpos + c + e, o[e]);
                if (dres) o[e] += r[e];
            }
            store8(dx + c, o);
        }
    }
}

#define LN_PR 4   // rows per wave in the parameter-gradient kernel (all their loads are issued up front)
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void norm_bwd_param_kernel(SmxNormBwdParams p) {
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + w) * LN_PR;
    float dg[LN_NCH][8], db[LN_NCH][8], gm[LN_NCH][8], bt[LN_NCH][8];
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[j][e] = db[j][e] = gm[j][e] = bt[j][e] = 0.f;
        if (ACT && c < p.D) {
            load8(p.gamma + c, gm[j]);
            if (p.beta) load8(p.beta + c, bt[j]);
        }
    }
    float xv[LN_PR][LN_NCH][8], dv[LN_PR][LN_NCH][8], mean[LN_PR], rstd[LN_PR];
#pragma unroll
    for (int r = 0; r < LN_PR; ++r) {
        const int row = row0 + r;
        mean[r] = 0.f; rstd[r] = 0.f;
        if (row < p.M) {
            mean[r] = p.rms ? 0.f : p.mean[row];
            rstd[r] = p.rstd[row];
        }
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j) {
            const int c = (lane + 64 * j) * 8;
            if (row < p.M && c < p.D) {
                load8(reinterpret_cast<const T*>(p.x) + (long long)row * p.D + c, xv[r][j]);
                load8(reinterpret_cast<const T*>(p.dy) + (long long)row * p.D + c, dv[r][j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[r][j][e] = dv[r][j][e] = 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < LN_PR; ++r)
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xhat = (xv[r][j][e] - mean[r]) * rstd[r];
                float d = dv[r][j][e];
                if (p.drop_p > 0.f)
                    d *= smx_drop_mul(p.drop_seed, (unsigned)((row0 + r) * p.D + (lane + 64 * j) * 8 + e), smx_thresh24(p.drop_p),
                                      1.0f / (1.0f - p.drop_p));
                if (ACT) d *= act_grad(xhat * gm[j][e] + bt[j][e], p.act);
                dg[j][e] += d * xhat;
                db[j][e] += d;
            }
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        if (64 * 8 * j >= p.D) break;
        const int c = (lane + 64 * j) * 8;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[w][lane][e] = pass == 0 ? dg[j][e] : db[j][e];
            __syncthreads();
            if (w == 0 && c < p.D) {
                float sum8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sum8[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
                store8(p.partials + ((long long)blockIdx.x * 2 + pass) * p.D + c, sum8);
            }
        }
    }
}

// dx AND the gamma / beta partial rows in ONE pass over dy and x (the two-kernel form read both tensors twice: the
// parameter pass alone was 1.8 % of the training step).  Same work split as the parameter kernel - a wave owns LN_PR rows,
// all of their loads issued up front - with the row's two reductions and its dx in between; partial rows are reduced over
// the block's four waves through LDS exactly as there (one partial row pair per block, folded later).
// PR = rows per wave (a block covers 4 PR rows and leaves one partial-row set): 4 for long inputs; 2 / 1 where 16 rows per
// block would leave the chip short of blocks (M = 7 968: 498 blocks on 256 CUs ran at 1.3 TB/s, M = 1 024: 64 blocks) and to
// halve the registers a wave holds (more waves per SIMD to overlap the row reductions with the loads).
template <typename T, bool ACT, int PR>
__global__ __launch_bounds__(256, PR == 1 ? LN_BLOCKS_PER_CU : 1) void norm_bwd_fused_kernel(SmxNormBwdParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    p.drop2_seed = smx_dseed(p.drop2_p, p.drop2_seed);
    __shared__ float red[4][64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float invD = 1.0f / (float)p.D;
    float dg[LN_NCH][8], db[LN_NCH][8], gm[LN_NCH][8], bt[LN_NCH][8];
    float dc[ACT ? 1 : LN_NCH][8];                 // column sums of the masked dx (third partial row)
    const bool third = !ACT && p.dx_drop != nullptr;
    const unsigned th2 = smx_thresh24(p.drop2_p);
    const float inv2 = 1.0f / (1.0f - p.drop2_p);
#pragma unroll
    for (int j = 0; j < (ACT ? 1 : LN_NCH); ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) dc[j][e] = 0.f;
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        const int c = (lane + 64 * j) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[j][e] = db[j][e] = gm[j][e] = bt[j][e] = 0.f;
        if (c < p.D) {
            load8(p.gamma + c, gm[j]);
            if (ACT && p.beta) load8(p.beta + c, bt[j]);
        }
    }
    // persistent over row groups when launched with fewer blocks than groups (ln_grid): the column accumulators live across
    // the groups and the block leaves ONE partial-row set - the LDS reduction below (six barrier pairs) was ~30 % of a
    // block's time when it followed every 8 rows
    const int ngroups = (p.M + 4 * PR - 1) / (4 * PR);
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int row0 = (grp * 4 + w) * PR;
    float xv[PR][LN_NCH][8], dv[PR][LN_NCH][8], mean[PR], rstd[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int row = row0 + r;
        mean[r] = 0.f; rstd[r] = 0.f;
        if (row < p.M) {
            mean[r] = p.rms ? 0.f : p.mean[row];
            rstd[r] = p.rstd[row];
        }
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j) {
            const int c = (lane + 64 * j) * 8;
            if (row < p.M && c < p.D) {
                load8(reinterpret_cast<const T*>(p.x) + (long long)row * p.D + c, xv[r][j]);
                load8(reinterpret_cast<const T*>(p.dy) + (long long)row * p.D + c, dv[r][j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[r][j][e] = dv[r][j][e] = 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int row = row0 + r;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_NCH; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xhat = (xv[r][j][e] - mean[r]) * rstd[r];
                float d = dv[r][j][e];
                if (p.drop_p > 0.f)
                    d *= smx_drop_mul(p.drop_seed, (unsigned)(row * p.D + (lane + 64 * j) * 8 + e), smx_thresh24(p.drop_p),
                                      1.0f / (1.0f - p.drop_p));
                if (ACT) d *= act_grad(xhat * gm[j][e] + bt[j][e], p.act);
                dg[j][e] += d * xhat;
                db[j][e] += d;
                const float gg = d * gm[j][e];
                xv[r][j][e] = xhat;               // (reuse the staging registers: xhat and the scaled gradient)
                dv[r][j][e] = gg;
                s1 += gg;
                s2 += gg * xhat;
            }
        s1 = p.rms ? 0.f : wave_sum(s1) * invD;
        s2 = wave_sum(s2) * invD;
        if (row < p.M) {
            T* dx = reinterpret_cast<T*>(p.dx) + (long long)row * p.D;
            const T* dres = p.dres ? reinterpret_cast<const T*>(p.dres) + (long long)row * p.D : nullptr;
            float* dpos = p.dpos ? p.dpos + (long long)((row % p.pos_period) + p.pos_offset) * p.D : nullptr;
#pragma unroll
            for (int j = 0; j < LN_NCH; ++j) {
                const int c = (lane + 64 * j) * 8;
                if (c < p.D) {
                    float o[8], rr[8];
                    if (dres) load8(dres + c, rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o[e] = (dv[r][j][e] - s1 - xv[r][j][e] * s2) * rstd[r];
                        if (dpos) atomicAdd(dpos + c + e, o[e]);
                        if (dres) o[e] += rr[e];
                    }
                    store8(dx + c, o);
                    if constexpr (!ACT) {
                        if (third) {                         // what smx_dropout_colsum would compute from the stored dx
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = rt(o[e], dx);
                            smx_drop_mul8(p.drop2_seed, (unsigned)((long long)row * p.D + c), th2, inv2, o);
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                o[e] = rt(o[e], dx);
                                dc[j][e] += o[e];
                            }
                            store8(reinterpret_cast<T*>(p.dx_drop) + (long long)row * p.D + c, o);
                        }
                    }
                }
            }
        }
    }
    }   // row groups
#pragma unroll
    for (int j = 0; j < LN_NCH; ++j) {
        if (64 * 8 * j >= p.D) break;
        const int c = (lane + 64 * j) * 8;
        const int npass = third ? 3 : 2;
        for (int pass = 0; pass < npass; ++pass) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[w][lane][e] = pass == 0 ? dg[j][e] : pass == 1 ? db[j][e] : dc[ACT ? 0 : j][e];
            __syncthreads();
            if (w == 0 && c < p.D) {
                float sum8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sum8[e] = red[0][lane][e] + red[1][lane][e] + red[2][lane][e] + red[3][lane][e];
                store8(p.partials + ((long long)blockIdx.x * npass + pass) * p.D + c, sum8);
            }
        }
    }
}

// second stage of the gamma/beta gradient: column sums over the per-block partials (grid.y-way split rows)
__global__ void norm_bwd_finalize_kernel(const float* __restrict__ partials, int nblocks, int D, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float g = 0.f, b = 0.f;
    for (int i = blockIdx.y; i < nblocks; i += gridDim.y) {
        g += partials[((long long)i * 2) * D + c];
        b += partials[((long long)i * 2 + 1) * D + c];
    }
    if (dgamma) atomicAdd(dgamma + c, g);
    if (dbeta) atomicAdd(dbeta + c, b);
}

static int ln_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    return cus;
}

extern "C" int smx_norm_fwd(const SmxNormParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxNormParams p = *pp;
    if (p.M <= 0 || p.D <= 0 || p.D > 64 * 8 * LN_NCH || (p.D & 7)) return SMX_EINVAL;
    if (p.pos && p.pos_period <= 0) return SMX_EINVAL;
    // one resident round: LN_FWD_BLOCKS_PER_CU blocks on every CU (SMX_NORMF_GRID=n overrides, 0: a block per four rows)
    static const int forced = getenv("SMX_NORMF_GRID") ? atoi(getenv("SMX_NORMF_GRID")) : -1;
    const int cap = forced >= 0 ? forced : LN_FWD_BLOCKS_PER_CU * ln_cus();
    const int blocks = (p.M + 3) / 4;
    dim3 grid(cap > 0 && blocks > cap ? cap : blocks);
    if (dtype == SMX_F32) hipLaunchKernelGGL(norm_fwd_kernel<float>, grid, dim3(256), 0, stream, p);
    else if (dtype == SMX_BF16) hipLaunchKernelGGL(norm_fwd_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
    else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// rows per wave of the fused backward (one partial-row set per 4 x this many rows); the two-kernel A/B form keeps 4
static int ln_rows_per_wave(int M) {
    static const bool fuse = !(getenv("SMX_NORM_FUSED") && getenv("SMX_NORM_FUSED")[0] == '0');
    static const int forced = getenv("SMX_NORM_PR") ? atoi(getenv("SMX_NORM_PR")) : 0;                 // A/B switch: 1 / 2 / 4
    if (!fuse) return LN_PR;
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // one row per wave: with the blocks persistent over row groups (ln_grid) the per-block reduction no longer argues for more rows
    // per block, and fewer registers keep three blocks per CU resident.  15 968 x 768: 26.1 us against 29.8 (2 rows, 512 blocks) and
    // 59.7 (4 rows); 7 968 rows 16.7 vs 18.1; 511 968 x 512 (the CNN's layer norms, configs 4 / 5) 357 us against 486 / 518 and 560 for the
    // raw-staged variant that round 2 used there
    (void)M;
    return 1;
}

// blocks of the fused backward = partial-row sets it leaves: one resident round of the chip (3 blocks per CU - the kernel's
// launch bound - on every CU), each block walking the row groups; fewer when there are fewer groups.  Measured on 15 968 x 768
// (tools/gpu_norm_bench.py, one row per wave): a block per group 41.1 us, 512 / 768 / 1024 / 1536 blocks 31.3 / 26.1 / 33.8 /
// 29.3 us - whole rounds of resident blocks win, and one round is best.  SMX_NORM_GRID=n overrides (0: a block per group).
static int ln_grid(int M, int pr) {
    static const bool fuse = !(getenv("SMX_NORM_FUSED") && getenv("SMX_NORM_FUSED")[0] == '0');
    static const int forced = getenv("SMX_NORM_GRID") ? atoi(getenv("SMX_NORM_GRID")) : -1;
    const int cap = forced >= 0 ? forced : LN_BLOCKS_PER_CU * ln_cus();
    const int groups = (M + 4 * pr - 1) / (4 * pr);
    return (fuse && cap > 0 && groups > cap) ? cap : groups;
}

extern "C" int smx_norm_bwd(const SmxNormBwdParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxNormBwdParams p = *pp;
    if (p.M <= 0 || p.D <= 0 || p.D > 64 * 8 * LN_NCH || (p.D & 7)) return SMX_EINVAL;
    if (p.dpos && p.pos_period <= 0) return SMX_EINVAL;
    if (dtype != SMX_F32 && dtype != SMX_BF16) return SMX_EINVAL;
    if ((p.dgamma || p.dbeta) && !p.partials) return SMX_EINVAL;
    if (p.dx_drop && (!p.defer_fold || !p.partials || !(p.dgamma || p.dbeta) || p.act != SMX_ACT_NONE || p.drop2_p < 0.f || p.drop2_p >= 1.f))
        return SMX_EINVAL;
    const bool act = p.act != SMX_ACT_NONE;
    dim3 grid((p.M + 3) / 4);
    static const bool fuse = !(getenv("SMX_NORM_FUSED") && getenv("SMX_NORM_FUSED")[0] == '0');      // A/B switch
    if (p.dx_drop && !fuse) return SMX_EINVAL;          // the third output exists in the fused kernel only
    if (fuse && (p.dgamma || p.dbeta)) {
        const int pr = ln_rows_per_wave(p.M);
        const int blocks = ln_grid(p.M, pr);                     // workspace: blocks * 2 (3) * D floats
#define LN_FUSED(T, A)                                                                                                  \
    do {                                                                                                                \
        if (pr == 4) hipLaunchKernelGGL((norm_bwd_fused_kernel<T, A, 4>), dim3(blocks), dim3(256), 0, stream, p);       \
        else if (pr == 2) hipLaunchKernelGGL((norm_bwd_fused_kernel<T, A, 2>), dim3(blocks), dim3(256), 0, stream, p);  \
        else hipLaunchKernelGGL((norm_bwd_fused_kernel<T, A, 1>), dim3(blocks), dim3(256), 0, stream, p);               \
    } while (0)
        if (dtype == SMX_F32) {
            if (act) LN_FUSED(float, true); else LN_FUSED(float, false);
        } else {
            if (act) LN_FUSED(bf16_t, true); else LN_FUSED(bf16_t, false);
        }
#undef LN_FUSED
        if (!p.defer_fold)
            hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((p.D + 63) / 64, blocks >= 64 ? 32 : 1), dim3(64), 0, stream,
                               p.partials, blocks, p.D, p.dgamma, p.dbeta);
        SMX_CHECK_LAUNCH();
    }
    if (dtype == SMX_F32) {
        if (act) hipLaunchKernelGGL((norm_bwd_dx_kernel<float, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((norm_bwd_dx_kernel<float, false>), grid, dim3(256), 0, stream, p);
    } else {
        if (act) hipLaunchKernelGGL((norm_bwd_dx_kernel<bf16_t, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((norm_bwd_dx_kernel<bf16_t, false>), grid, dim3(256), 0, stream, p);
    }
    if (p.dgamma || p.dbeta) {
        const int blocks = (p.M + 4 * LN_PR - 1) / (4 * LN_PR);   // workspace: blocks * 2 * D floats
        if (dtype == SMX_F32) {
            if (act) hipLaunchKernelGGL((norm_bwd_param_kernel<float, true>), dim3(blocks), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((norm_bwd_param_kernel<float, false>), dim3(blocks), dim3(256), 0, stream, p);
        } else {
            if (act) hipLaunchKernelGGL((norm_bwd_param_kernel<bf16_t, true>), dim3(blocks), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL((norm_bwd_param_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, stream, p);
        }
        if (!p.defer_fold)
            hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3((p.D + 63) / 64, blocks >= 64 ? 32 : 1), dim3(64), 0, stream,
                               p.partials, blocks, p.D, p.dgamma, p.dbeta);
    }
    SMX_CHECK_LAUNCH();
}

extern "C" int smx_norm_bwd_partial_rows(int M) { return ln_grid(M, ln_rows_per_wave(M)); }

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxNormParams(void) { return (int)sizeof(SmxNormParams); }
extern "C" int smx_sizeof_SmxNormBwdParams(void) { return (int)sizeof(SmxNormBwdParams); }

SMX_STEP_KEY_TU(norm)
