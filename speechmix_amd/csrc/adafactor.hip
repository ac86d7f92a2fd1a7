// Adafactor on the flat parameter buffer - the optimizer the reference trains with (ref:train.py:298 `optim="adafactor"`,
// which HF Trainer instantiates as Adafactor(lr, scale_parameter=False, relative_step=False); TF:optimization.py
// Adafactor.step, _approx_sq_grad, _rms).  Per tensor of shape [..., R, C] (>= 2-D: factored) or [n] (unfactored):
//     u    = g^2 + eps1
//     row  = beta2t row + (1 - beta2t) mean_C(u)        col = beta2t col + (1 - beta2t) mean_R(u)
//     upd  = g * rsqrt(row / mean_R(row)) * rsqrt(col)   (1-D:  v = beta2t v + (1 - beta2t) u ; upd = g rsqrt(v))
//     upd /= max(1, rms(upd) / clip_threshold) ;  p -= lr upd           beta2t = 1 - step^decay_rate
// The state is two vectors per matrix instead of AdamW's two full copies (2 MB instead of 1.9 GB at config 2).
//
// One step = six launches over host-built work lists (no per-tensor launches: 458 tensors at config 2):
//   stats   tiles of >= 64 rows x <= 2048 columns: row sums by wave reductions, column partials in registers; also each tile's
//           sum of squares - the global gradient norm for clipping comes out of this pass (gnorm: one block adds the tiles'
//           sums in tile order) instead of a pass of its own over the gradient (smx_sumsq: 147 us at config 2)
//   fold    one block per (tensor, leading index): the two moving averages, the row mean
//   rms     tiles again: sum upd^2 per tile     usq   per tensor     apply   tiles again: the update + the bf16 compute copy
// Gradient traffic 3 x 4 B + parameters 8 B + bf16 copy 2 B = 22 B / parameter (AdamW: 30).
//
// EVERY reduction has a fixed order (round 3).  Data-parallel replicas apply this update to identical all-reduced gradients and
// must end with identical parameters; with fp32 atomics summing the column partials of a matrix's row tiles and the per-tile
// sum upd^2 in arrival order, two ranks drifted apart bit-wise within four steps (tests/test_gpu_r3.py, two ranks on one GPU).
// Now a row tile stores its column partials into its own row of a scratch matrix [row tiles][C] that the fold sums in tile
// order, a tile stores its sum upd^2 into its own slot and one thread per tensor adds the slots in tile order; the row sums of a
// matrix wider than one column tile go the same way (every column tile stores its row sums into its own row of the scratch block
// [column tiles][R] at rp_off, the fold adds them in tile order).  NO atomics are left in this file, and racc / cacc are
// STORE-ONLY scratch: every element the fold reads was written by exactly one tile of the same step, so the host allocates them
// uninitialised and nothing zeroes them - a change that adds into them (atomicAdd) would read garbage.
#include "smx_common.h"

struct SmxAfTensor {
    long long off;        // element offset of the tensor in p / g / shadow
    int nb, R, C;         // leading (batch) count, rows, columns; 1-D tensors: nb = 1, R = 1, C = n, factored = 0
    int row_off, col_off; // offsets into row / col state (factored) ; vec_off = col_off for 1-D tensors (state v)
    int rm_off;           // offset into the per-(tensor, batch) row-mean array
    int factored;
    int tile0, ntile;     // this tensor's tiles in the tile list (contiguous)
    int _pad;
};
struct SmxAfTile {
    int tensor, b, r0, nr, c0, nc;
    int full_rows;        // tile spans every column: row sums are final (plain store)
    int full_cols;        // tile spans every row: column sums are final
    int cp_off;           // !full_cols: where this tile's nc column partials go in `cpart` (segment base + row tile * C + c0)
    int rp_off;           // !full_rows: where this tile's row sums go in `cpart` (segment base + column tile * R; index + row)
};
// cp_off / n_rt: the segment's [n_rt][C] block of column partials; rp_off / n_ct: its [n_ct][R] block of row partials
struct SmxAfSeg { int tensor, b, cp_off, n_rt, rp_off, n_ct; };
struct SmxAfParams {
    float* p;
    const float* g;
    void* shadow;                 // bf16 copy or null
    const SmxAfTensor* tensors;
    const SmxAfTile* tiles;
    const SmxAfSeg* segs;
    float* row;                   // factored state
    float* col;                   // factored column state; 1-D tensors keep their v here too
    float* racc;                  // scratch, same layout as row / col / per tensor (store-only: never zeroed, see the header)
    float* cacc;
    float* rmean;                 // [nsegs]
    float* usq;                   // [ntensors] sum upd^2
    float* usq_part;              // [ntiles] per-tile sum upd^2
    float* cpart;                 // column partials of the row tiles ([n_rt][C] per segment with more than one row tile)
    const float* beta2t;          // [ntensors] (tensors without a gradient this step: < 0 -> skipped, as HF does)
    float* gn2;                   // device scalar: sum (grad_scale g)^2 over every active tensor, written by the gnorm launch
    float* gsq_part;              // [ntiles] per-tile sum (grad_scale g)^2
    long long racc_n, cacc_n;
    int ntensors, ntiles, nsegs;
    float lr, eps1, clip_threshold, grad_scale, max_grad_norm;
};

#define AF_MAXC 2048
#define AF_ROWS 64

// gradient scale: 1 / world (grad_scale) x the global-norm clip coefficient (HF Trainer: clip_grad_norm_(max_grad_norm), then
// optimizer.step).  The second moments are linear in the squared scale, so the stats pass runs BEFORE the norm is known, on
// grad_scale alone, and the fold applies the coefficient's square.
__device__ __forceinline__ float af_clip(const SmxAfParams& o) {
    if (!(o.max_grad_norm > 0.f)) return 1.f;
    return fminf(1.f, o.max_grad_norm / (sqrtf(*o.gn2) + 1e-6f));
}
__device__ __forceinline__ float af_gscale(const SmxAfParams& o) { return o.grad_scale * af_clip(o); }

// ---- stats: u0 = (grad_scale g)^2 summed along rows and columns, and over the tile --------------------------------------
__global__ __launch_bounds__(256) void af_stats_kernel(SmxAfParams o, int tile0) {
    __shared__ float cpart[4][AF_MAXC];
    __shared__ float wsum[4];
    const int tix = blockIdx.x + tile0;          // (a ranged launch - smx_adafactor_phase - covers tiles tile0 .. tile0 + gridDim.x - 1)
    const SmxAfTile tl = o.tiles[tix];
    const SmxAfTensor T = o.tensors[tl.tensor];
    const float gs = o.grad_scale;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* g = o.g + T.off + (long long)tl.b * T.R * T.C;
    if (o.beta2t[tl.tensor] < 0.f) {
        // no update for this tensor in this step, but whatever its gradient range holds still counts in the global norm (the
        // norm is taken over the whole flat gradient; a frozen or layer-dropped tensor's range is zeros)
        float s = 0.f;
        if (o.max_grad_norm > 0.f) {
            const bool v4 = !(T.C & 3) && !(tl.c0 & 3) && !(tl.nc & 3);
            for (int r = tl.r0 + w; r < tl.r0 + tl.nr; r += 4) {
                const float* gr = g + (long long)r * T.C + tl.c0;
                if (v4) {
                    for (int c = 4 * lane; c < tl.nc; c += 256) {
                        const float4 q = *reinterpret_cast<const float4*>(gr + c);
                        s += (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
                    }
                } else {
                    for (int c = lane; c < tl.nc; c += 64) s += gr[c] * gr[c];
                }
            }
            s *= gs * gs;
        }
        s = block_sum(s, &cpart[0][0]);
        if (threadIdx.x == 0) o.gsq_part[tix] = s;
        return;
    }
    if (!T.factored) {               // 1-D: only the tile's sum of squares here (v and sum upd^2: the rms launch, once the norm is known)
        float s = 0.f;
        for (int c = tl.c0 + threadIdx.x; c < tl.c0 + tl.nc; c += 256) {
            const float gi = g[c] * gs;
            s += gi * gi;
        }
        s = block_sum(s, &cpart[0][0]);
        if (threadIdx.x == 0) o.gsq_part[tix] = s;
        return;
    }
    float* racc = o.racc + T.row_off + (long long)tl.b * T.R;
    float* cacc = o.cacc + T.col_off + (long long)tl.b * T.C + tl.c0;
    if (T.C < 64) {
        // narrow matrices (conv kernels [Co, Ci, k]: C = k): one thread per row; such a tile always covers its whole (R x C)
        // matrix.  Column sums: every thread's own rows first, then one fixed-order block reduction per column (LDS atomics
        // added in arrival order here until round 3)
        float tot = 0.f;
        for (int r = tl.r0 + threadIdx.x; r < tl.r0 + tl.nr; r += 256) {
            const float* gr = g + (long long)r * T.C;
            float rs = 0.f;
            for (int c = 0; c < T.C; ++c) {
                const float gi = gr[c] * gs;
                rs += gi * gi;
            }
            racc[r] = rs;
            tot += rs;
        }
        for (int c = 0; c < T.C; ++c) {
            float a = 0.f;
            for (int r = tl.r0 + threadIdx.x; r < tl.r0 + tl.nr; r += 256) {
                const float gi = g[(long long)r * T.C + c] * gs;
                a += gi * gi;
            }
            __syncthreads();
            a = block_sum(a, &cpart[0][0]);
            if (threadIdx.x == 0) cacc[c] = a;
        }
        __syncthreads();
        tot = block_sum(tot, &cpart[0][0]);
        if (threadIdx.x == 0) o.gsq_part[tix] = tot;
        return;
    }
    // wide: wave w takes rows r0 + w, + 4 ...; a lane takes 4 consecutive columns c0 + 4 (lane + 64 j) (16-B loads when
    // the row length allows it)
    constexpr int NJ = AF_MAXC / 256;
    const bool v4 = !(T.C & 3);
    float cs[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[j][e] = 0.f;
    float wtot = 0.f;                       // this wave's rows, in row order
    // two rows per trip with all their loads issued first: a wave walks 16+ rows of a tile, and one row's 16-B loads are
    // too few bytes in flight to stream at HBM speed
    for (int r = tl.r0 + w; r < tl.r0 + tl.nr; r += 8) {
        const bool two = r + 4 < tl.r0 + tl.nr;
        float x[2][NJ][4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float* gr = g + (long long)(r + 4 * k) * T.C + tl.c0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[k][j][e] = 0.f;
                if (c < tl.nc && (k == 0 || two)) {
                    if (v4) {
                        const float4 q = *reinterpret_cast<const float4*>(gr + c);
                        x[k][j][0] = q.x; x[k][j][1] = q.y; x[k][j][2] = q.z; x[k][j][3] = q.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[k][j][e] = c + e < tl.nc ? gr[c + e] : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !two) break;
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
                if (c < tl.nc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gi = x[k][j][e] * gs;
                        const float u = gi * gi;             // (columns beyond the tile were loaded as zeros)
                        rs += u;
                        cs[j][e] += u;
                    }
                }
            }
            rs = wave_sum(rs);
            wtot += rs;
            if (lane == 0) {
                if (tl.full_rows) racc[r + 4 * k] = rs;
                else o.cpart[tl.rp_off + r + 4 * k] = rs;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) cpart[w][4 * (lane + 64 * j) + e] = cs[j][e];
    __syncthreads();
    for (int c = threadIdx.x; c < tl.nc; c += 256) {
        const float sum = (cpart[0][c] + cpart[1][c]) + (cpart[2][c] + cpart[3][c]);
        if (tl.full_cols) cacc[c] = sum;
        else o.cpart[tl.cp_off + c] = sum;
    }
    if (lane == 0) wsum[w] = wtot;
    __syncthreads();
    if (threadIdx.x == 0) o.gsq_part[tix] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// ---- gnorm: sum (grad_scale g)^2 over all tiles, in tile order -----------------------------------
__global__ __launch_bounds__(1024) void af_gnorm_kernel(SmxAfParams o) {
    __shared__ float sh[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < o.ntiles; i += 1024) a += o.gsq_part[i];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) *o.gn2 = a;
}

// ---- fold: moving averages of one (tensor, leading index), and the mean of its row state ---------------------------
__global__ __launch_bounds__(256) void af_fold_kernel(SmxAfParams o) {
    __shared__ float sh[16];
    const SmxAfSeg sg = o.segs[blockIdx.x];
    const SmxAfTensor T = o.tensors[sg.tensor];
    const float b2 = o.beta2t[sg.tensor];
    if (b2 < 0.f || !T.factored) return;
    float* row = o.row + T.row_off + (long long)sg.b * T.R;
    const float* racc = o.racc + T.row_off + (long long)sg.b * T.R;
    float* col = o.col + T.col_off + (long long)sg.b * T.C;
    const float* cacc = o.cacc + T.col_off + (long long)sg.b * T.C;
    const float ic = 1.0f / (float)T.C, ir = 1.0f / (float)T.R;
    // u = (clip grad_scale g)^2 + eps1: the stats pass summed (grad_scale g)^2; the clip coefficient's square and the eps1 terms
    // (one per summed element) come in here
    const float c2 = af_clip(o) * af_clip(o), er = (float)T.C * o.eps1, ec = (float)T.R * o.eps1;
    float s = 0.f;
    const float* rp = o.cpart + sg.rp_off;
    auto rsum = [&](int r) -> float {            // row sum of u: one column tile, or the column tiles' partials in tile order
        float a = 0.f;
        if (sg.n_ct <= 1) a = racc[r];
        else
            for (int t = 0; t < sg.n_ct; ++t) a += rp[(long long)t * T.R + r];
        return fmaf(a, c2, er);
    };
    // (the embedding's 50 k rows are one segment, 196 trips per thread: eight trips' loads in flight instead of one)
    int r = threadIdx.x;
    for (; r + 7 * 256 < T.R; r += 8 * 256) {
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = row[r + i * 256]; b[i] = rsum(r + i * 256); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = b2 * a[i] + (1.f - b2) * (b[i] * ic);
            row[r + i * 256] = v;
            s += v;
        }
    }
    for (; r < T.R; r += 256) {
        const float v = b2 * row[r] + (1.f - b2) * (rsum(r) * ic);
        row[r] = v;
        s += v;
    }
    if (sg.n_rt <= 1) {
        for (int c = threadIdx.x; c < T.C; c += 256) col[c] = b2 * col[c] + (1.f - b2) * (fmaf(cacc[c], c2, ec) * ir);
    } else {
        // column sums = the row tiles' partials added in tile order (eight loads in flight per column)
        const float* cp = o.cpart + sg.cp_off;
        for (int c = threadIdx.x; c < T.C; c += 256) {
            float a = 0.f;
            int t = 0;
            for (; t + 8 <= sg.n_rt; t += 8) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = cp[(long long)(t + i) * T.C + c];
#pragma unroll
                for (int i = 0; i < 8; ++i) a += v[i];
            }
            for (; t < sg.n_rt; ++t) a += cp[(long long)t * T.C + c];
            col[c] = b2 * col[c] + (1.f - b2) * (fmaf(a, c2, ec) * ir);
        }
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) o.rmean[T.rm_off + sg.b] = s * ir;
}

// ---- rms / apply ---------------------------------------------------------------------------------------------------
template <bool APPLY>
__global__ __launch_bounds__(256) void af_update_kernel(SmxAfParams o, int tile0) {
    __shared__ float sh[16];
    const int tix = blockIdx.x + tile0;          // (a ranged launch - smx_adafactor_phase - covers tiles tile0 .. tile0 + gridDim.x - 1)
    const SmxAfTile tl = o.tiles[tix];
    const SmxAfTensor T = o.tensors[tl.tensor];
    if (o.beta2t[tl.tensor] < 0.f) return;
    const float gs = af_gscale(o);
    const long long base = T.off + (long long)tl.b * T.R * T.C;
    const float* g = o.g + base;
    float scale = 0.f;
    if (APPLY) {
        const float n = (float)T.nb * (float)T.R * (float)T.C;
        const float rms = sqrtf(o.usq[tl.tensor] / n);
        scale = o.lr / fmaxf(1.0f, rms / o.clip_threshold);
    }
    bf16_t* shd = reinterpret_cast<bf16_t*>(o.shadow);
    if (!T.factored) {
        if (!APPLY) {                        // 1-D: v updated in place, sum upd^2 on the fly
            const float b2 = o.beta2t[tl.tensor];
            float* v = o.col + T.col_off;
            float s1 = 0.f;
            for (int c = tl.c0 + threadIdx.x; c < tl.c0 + tl.nc; c += 256) {
                const float gi = g[c] * gs;
                const float vi = b2 * v[c] + (1.f - b2) * (gi * gi + o.eps1);
                v[c] = vi;
                const float u = gi * rsqrtf(vi);
                s1 += u * u;
            }
            s1 = block_sum(s1, sh);
            if (threadIdx.x == 0) o.usq_part[tix] = s1;
            return;
        }
        const float* v = o.col + T.col_off;
        for (int c = tl.c0 + threadIdx.x; c < tl.c0 + tl.nc; c += 256) {
            const float pi = o.p[base + c] - scale * (g[c] * gs) * rsqrtf(v[c]);
            o.p[base + c] = pi;
            if (shd) shd[base + c] = f2bf(pi);
        }
        return;
    }
    const float* row = o.row + T.row_off + (long long)tl.b * T.R;
    const float* col = o.col + T.col_off + (long long)tl.b * T.C + tl.c0;
    const float rmean = o.rmean[T.rm_off + tl.b];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = 0.f;
    if (T.C < 64) {                              // narrow: one thread per row
        for (int r = tl.r0 + threadIdx.x; r < tl.r0 + tl.nr; r += 256) {
            const float rf = rsqrtf(row[r] / rmean) * gs;
            const long long ro = (long long)r * T.C;
            for (int c = 0; c < T.C; ++c) {
                const float u = g[ro + c] * rf * rsqrtf(col[c]);
                if (APPLY) {
                    const float pi = o.p[base + ro + c] - scale * u;
                    o.p[base + ro + c] = pi;
                    if (shd) shd[base + ro + c] = f2bf(pi);
                } else {
                    s += u * u;
                }
            }
        }
    } else {
        constexpr int NJ = AF_MAXC / 256;
        const bool v4 = !(T.C & 3);
        float cf[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * (lane + 64 * j) + e;
                cf[j][e] = c < tl.nc ? rsqrtf(col[c]) : 0.f;
            }
        for (int r = tl.r0 + w; r < tl.r0 + tl.nr; r += 4) {
            const float rf = rsqrtf(row[r] / rmean) * gs;
            const long long ro = (long long)r * T.C + tl.c0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = 4 * (lane + 64 * j);
                if (c >= tl.nc) continue;
                if (v4) {
                    const float4 q = *reinterpret_cast<const float4*>(g + ro + c);
                    const float u0 = q.x * rf * cf[j][0], u1 = q.y * rf * cf[j][1], u2 = q.z * rf * cf[j][2], u3 = q.w * rf * cf[j][3];
                    if (APPLY) {
                        float4 pv = *reinterpret_cast<float4*>(o.p + base + ro + c);
                        pv.x -= scale * u0; pv.y -= scale * u1; pv.z -= scale * u2; pv.w -= scale * u3;
                        *reinterpret_cast<float4*>(o.p + base + ro + c) = pv;
                        if (shd) *reinterpret_cast<uint2*>(shd + base + ro + c) = make_uint2(pack_bf2(pv.x, pv.y), pack_bf2(pv.z, pv.w));
                    } else {
                        s += (u0 * u0 + u1 * u1) + (u2 * u2 + u3 * u3);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (c + e >= tl.nc) continue;
                        const float u = g[ro + c + e] * rf * cf[j][e];
                        if (APPLY) {
                            const float pi = o.p[base + ro + c + e] - scale * u;
                            o.p[base + ro + c + e] = pi;
                            if (shd) shd[base + ro + c + e] = f2bf(pi);
                        } else {
                            s += u * u;
                        }
                    }
                }
            }
        }
    }
    if (!APPLY) {
        s = block_sum(s, sh);
        if (threadIdx.x == 0) o.usq_part[tix] = s;
    }
}

// ---- usq: a tensor's sum upd^2 = its tiles' partials in a fixed order (one wave per tensor: lane l adds tiles l, l + 64, ...
// in order, then the fixed wave tree) -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void af_usq_kernel(SmxAfParams o) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= o.ntensors || o.beta2t[t] < 0.f) return;
    const SmxAfTensor T = o.tensors[t];
    float a = 0.f;
    for (int i = lane; i < T.ntile; i += 64) a += o.usq_part[T.tile0 + i];
    a = wave_sum(a);
    if (lane == 0) o.usq[t] = a;
}

extern "C" int smx_adafactor_step(const SmxAfParams* op, hipStream_t stream) {
    (void)hipGetLastError();
    SmxAfParams o = *op;
    if (o.ntiles <= 0 || o.ntensors <= 0) return SMX_OK;
    if (!o.p || !o.g || !o.tensors || !o.tiles || !o.segs || !o.row || !o.col || !o.racc || !o.cacc || !o.rmean || !o.usq ||
        !o.usq_part || !o.cpart || !o.beta2t || !o.gsq_part || !o.gn2) return SMX_EINVAL;
    hipLaunchKernelGGL(af_stats_kernel, dim3(o.ntiles), dim3(256), 0, stream, o, 0);
    if (o.max_grad_norm > 0.f) hipLaunchKernelGGL(af_gnorm_kernel, dim3(1), dim3(1024), 0, stream, o);
    if (o.nsegs > 0) hipLaunchKernelGGL(af_fold_kernel, dim3(o.nsegs), dim3(256), 0, stream, o);
    hipLaunchKernelGGL(af_update_kernel<false>, dim3(o.ntiles), dim3(256), 0, stream, o, 0);
    hipLaunchKernelGGL(af_usq_kernel, dim3((o.ntensors + 3) / 4), dim3(256), 0, stream, o);
    hipLaunchKernelGGL(af_update_kernel<true>, dim3(o.ntiles), dim3(256), 0, stream, o, 0);
    SMX_CHECK_LAUNCH();
}

// The same step in phases (round 6: the optimizer's tail beside the next step's front end, trainer.py):
//   phase 0: the statistics pass over EVERY tile, the global gradient norm and the fold of the column / row partials - nothing is updated
//            (= phase 2 over every tile + phase 3);
//   phase 1: the two update passes (per-tensor RMS of the update, then the parameters) over the tiles tile_first .. tile_first + tile_count - 1,
//            which must be whole tensors (a tensor's tiles are contiguous); the per-tensor sums between them run over all tensors - a tensor
//            outside the range gets a sum of stale partials that its own phase-1 call recomputes before it is used.
// phase 0 followed by phase-1 calls that cover every tile exactly once = smx_adafactor_step, bit for bit (same kernels, same order per tensor).
extern "C" int smx_adafactor_phase(const SmxAfParams* op, int phase, int tile_first, int tile_count, hipStream_t stream) {
    (void)hipGetLastError();
    SmxAfParams o = *op;
    if (o.ntiles <= 0 || o.ntensors <= 0) return SMX_OK;
    if (!o.p || !o.g || !o.tensors || !o.tiles || !o.segs || !o.row || !o.col || !o.racc || !o.cacc || !o.rmean || !o.usq ||
        !o.usq_part || !o.cpart || !o.beta2t || !o.gsq_part || !o.gn2) return SMX_EINVAL;
    if (phase == 0) {
        hipLaunchKernelGGL(af_stats_kernel, dim3(o.ntiles), dim3(256), 0, stream, o, 0);
        if (o.max_grad_norm > 0.f) hipLaunchKernelGGL(af_gnorm_kernel, dim3(1), dim3(1024), 0, stream, o);
        if (o.nsegs > 0) hipLaunchKernelGGL(af_fold_kernel, dim3(o.nsegs), dim3(256), 0, stream, o);
        SMX_CHECK_LAUNCH();
    }
    if (phase == 3) {          // the global norm and the partial folds: behind phase-2 calls that covered every tile
        if (o.max_grad_norm > 0.f) hipLaunchKernelGGL(af_gnorm_kernel, dim3(1), dim3(1024), 0, stream, o);
        if (o.nsegs > 0) hipLaunchKernelGGL(af_fold_kernel, dim3(o.nsegs), dim3(256), 0, stream, o);
        SMX_CHECK_LAUNCH();
    }
    if ((phase != 1 && phase != 2) || tile_first < 0 || tile_count < 0 || tile_first + tile_count > o.ntiles) return SMX_EINVAL;
    if (tile_count == 0) return SMX_OK;
    if (phase == 2) {          // the statistics pass over a tile range (whole tensors): gradients that are final early - the LM's and the
                               // encoder layers' while the front end is still in backward - are read beside it on another stream
        hipLaunchKernelGGL(af_stats_kernel, dim3(tile_count), dim3(256), 0, stream, o, tile_first);
        SMX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(af_update_kernel<false>, dim3(tile_count), dim3(256), 0, stream, o, tile_first);
    hipLaunchKernelGGL(af_usq_kernel, dim3((o.ntensors + 3) / 4), dim3(256), 0, stream, o);
    hipLaunchKernelGGL(af_update_kernel<true>, dim3(tile_count), dim3(256), 0, stream, o, tile_first);
    SMX_CHECK_LAUNCH();
}

extern "C" int smx_sizeof_SmxAfParams(void) { return (int)sizeof(SmxAfParams); }
extern "C" int smx_sizeof_SmxAfTensor(void) { return (int)sizeof(SmxAfTensor); }
extern "C" int smx_sizeof_SmxAfTile(void) { return (int)sizeof(SmxAfTile); }
extern "C" int smx_sizeof_SmxAfSeg(void) { return (int)sizeof(SmxAfSeg); }
