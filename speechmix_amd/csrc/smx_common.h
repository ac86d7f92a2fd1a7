// speechmix_amd — common device helpers for gfx950 (CDNA4) kernels.
// Wavefront = 64 lanes, MFMA 16x16x32 bf16, LDS-staged tiles.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMX_WAVE 64

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA 16x16x32 A/B operand (4 VGPR)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// ---- error convention: launchers return 0 on success, hipError_t (>0) or negative arg errors ----
#define SMX_OK 0
#define SMX_EINVAL (-22)
#define SMX_ENOSYS (-38)
#define SMX_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        return e__ == hipSuccess ? SMX_OK : (int)e__;        \
    } while (0)

// ---- dtype tags used across the C ABI ----
enum { SMX_F32 = 0, SMX_BF16 = 1 };

// ---- bf16 <-> f32 ----
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN): one instruction per pair
typedef __attribute__((ext_vector_type(2))) float smx_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 smx_bf16x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    union { smx_bf16x2_t b; unsigned u; } c;
    c.b = __builtin_convertvector((smx_f32x2_t){lo, hi}, smx_bf16x2_t);
    return c.u;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- 8-element vector access (16 B of bf16 / 32 B of fp32); pointers must be 16-B aligned ----
__device__ __forceinline__ void load8(const bf16_t* p, float o[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void load8(const float* p, float o[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]),
                                              pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// round-trip through the storage type (so statistics see exactly what is stored)
__device__ __forceinline__ float rt(float v, const bf16_t*) { return bf2f(f2bf(v)); }
__device__ __forceinline__ float rt(float v, const float*) { return v; }

// ---- activations (exact erf GELU as in HF ACT2FN["gelu"]) ----
enum { SMX_ACT_NONE = 0, SMX_ACT_GELU = 1, SMX_ACT_RELU = 2 };
// erf for the GELU epilogues: branch-free Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off
// class) - libm's erff is a divergent piecewise evaluation that made the FFN1 epilogue cost more than its GEMM.
__device__ __forceinline__ float smx_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float r = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == SMX_ACT_GELU) return 0.5f * x * (1.0f + smx_erf(x * 0.70710678118654752440f));
    if (act == SMX_ACT_RELU) return x > 0.f ? x : 0.f;
    return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {
    if (act == SMX_ACT_GELU) {       // erf(x/sqrt2) and the pdf share one exponential E = exp(-x^2/2)
        const float ax = fabsf(x);
        const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
        float poly = fmaf(1.061405429f, t, -1.453152027f);
        poly = fmaf(poly, t, 1.421413741f);
        poly = fmaf(poly, t, -0.284496736f);
        poly = fmaf(poly, t, 0.254829592f);
        const float E = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.44269504088896340736f));
        const float er = copysignf(1.0f - poly * t * E, x);
        return fmaf(er, 0.5f, 0.5f) + x * 0.39894228040143267794f * E;
    }
    if (act == SMX_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    return 1.f;
}

// ---- the same two functions on 8 values with packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 process two values per
// instruction at full rate; the activation epilogues are VALU-bound).  erf(x/sqrt2) and the Gaussian pdf share ONE
// exponential E = exp(-x^2/2): erf = 1 - poly(t) t E (A&S 7.1.26 at z = |x|/sqrt2), pdf = E / sqrt(2 pi).
typedef __attribute__((ext_vector_type(2))) float smx_f2;
#define SMX_PK(v) ((smx_f2){(v), (v)})
__device__ __forceinline__ void smx_erf_e2(smx_f2 x, smx_f2& erfv, smx_f2& E) {
    const smx_f2 ax = {fabsf(x[0]), fabsf(x[1])};
    const smx_f2 d = __builtin_elementwise_fma(SMX_PK(0.3275911f * 0.70710678118654752440f), ax, SMX_PK(1.0f));
    const smx_f2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    smx_f2 poly = __builtin_elementwise_fma(SMX_PK(1.061405429f), t, SMX_PK(-1.453152027f));
    poly = __builtin_elementwise_fma(poly, t, SMX_PK(1.421413741f));
    poly = __builtin_elementwise_fma(poly, t, SMX_PK(-0.284496736f));
    poly = __builtin_elementwise_fma(poly, t, SMX_PK(0.254829592f));
    const smx_f2 arg = x * x * SMX_PK(-0.5f * 1.44269504088896340736f);
    E = (smx_f2){__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
    const smx_f2 r = __builtin_elementwise_fma(-(poly * t), E, SMX_PK(1.0f));
    erfv = (smx_f2){copysignf(r[0], x[0]), copysignf(r[1], x[1])};
}
__device__ __forceinline__ void act_fwd8(float x[8], int act) {
    if (act == SMX_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const smx_f2 v = {x[e], x[e + 1]};
            smx_f2 er, E;
            smx_erf_e2(v, er, E);
            const smx_f2 hx = v * SMX_PK(0.5f);
            const smx_f2 y = __builtin_elementwise_fma(hx, er, hx);
            x[e] = y[0];
            x[e + 1] = y[1];
        }
    } else if (act == SMX_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
    }
}
// SmxGemmParams.act may carry this flag (bf16 GEMMs): the side tensor then holds the LOCAL DERIVATIVE of the epilogue,
// d out / d pre = act'(pre) * dropout multiplier, written by the forward launch (aux_out) and multiplied in by the backward
// launch (aux_in) - which then needs neither the erf / exp of the activation gradient nor the dropout hash.  Rounded to
// bf16 like the pre-activation it replaces (its error, <= 2^-9 of a value in [-0.13, 1.13], is of the size the rounding of
// the pre-activation already put on the recomputed derivative).
#define SMX_ACT_SAVE_GRAD 0x100
// x -> act(x), d = act'(x): one erf / exp evaluation serves both
__device__ __forceinline__ void act_fwd_grad8(float x[8], float d[8], int act) {
    if (act == SMX_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const smx_f2 v = {x[e], x[e + 1]};
            smx_f2 er, E;
            smx_erf_e2(v, er, E);
            const smx_f2 hx = v * SMX_PK(0.5f);
            const smx_f2 y = __builtin_elementwise_fma(hx, er, hx);
            const smx_f2 cdf = __builtin_elementwise_fma(er, SMX_PK(0.5f), SMX_PK(0.5f));
            const smx_f2 gr = __builtin_elementwise_fma(v * SMX_PK(0.39894228040143267794f), E, cdf);
            x[e] = y[0]; x[e + 1] = y[1];
            d[e] = gr[0]; d[e + 1] = gr[1];
        }
    } else if (act == SMX_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { d[e] = x[e] > 0.f ? 1.f : 0.f; x[e] = fmaxf(x[e], 0.f); }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = 1.f;
    }
}
__device__ __forceinline__ unsigned smx_hash32(unsigned seed, unsigned idx);      // (dropout section below)
// the forward epilogue of a Linear -> activation -> dropout with SMX_ACT_SAVE_GRAD, pair by pair (few live registers):
// x -> act(x) * m, returns the bf16-packed local derivative act'(x) * m (m = dropout multiplier, 1 when drop is false);
// idx: even mask index of x[0]
__device__ __forceinline__ uint4 act_fwd_grad_drop8(float x[8], int act, bool drop, unsigned seed, unsigned idx, unsigned thresh24,
                                                    float inv_keep) {
    unsigned pk[4];
    const unsigned th = thresh24 >> 8;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        smx_f2 y, gr;
        const smx_f2 v = {x[e], x[e + 1]};
        if (act == SMX_ACT_GELU) {
            smx_f2 er, E;
            smx_erf_e2(v, er, E);
            const smx_f2 hx = v * SMX_PK(0.5f);
            y = __builtin_elementwise_fma(hx, er, hx);
            gr = __builtin_elementwise_fma(v * SMX_PK(0.39894228040143267794f), E, __builtin_elementwise_fma(er, SMX_PK(0.5f), SMX_PK(0.5f)));
        } else if (act == SMX_ACT_RELU) {
            y = (smx_f2){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
            gr = (smx_f2){v[0] > 0.f ? 1.f : 0.f, v[1] > 0.f ? 1.f : 0.f};
        } else {
            y = v;
            gr = SMX_PK(1.f);
        }
        if (drop) {
            const unsigned h = smx_hash32(seed, (idx >> 1) + (e >> 1));
            const smx_f2 m = {(h & 0xffffu) >= th ? inv_keep : 0.f, (h >> 16) >= th ? inv_keep : 0.f};
            y *= m;
            gr *= m;
        }
        x[e] = y[0];
        x[e + 1] = y[1];
        pk[e >> 1] = pack_bf2(gr[0], gr[1]);
    }
    return make_uint4(pk[0], pk[1], pk[2], pk[3]);
}
// x[e] *= act'(pre[e])
__device__ __forceinline__ void act_grad_mul8(float x[8], const float pre[8], int act) {
    if (act == SMX_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const smx_f2 v = {pre[e], pre[e + 1]};
            smx_f2 er, E;
            smx_erf_e2(v, er, E);
            const smx_f2 cdf = __builtin_elementwise_fma(er, SMX_PK(0.5f), SMX_PK(0.5f));
            const smx_f2 gr = __builtin_elementwise_fma(v * SMX_PK(0.39894228040143267794f), E, cdf);
            x[e] *= gr[0];
            x[e + 1] *= gr[1];
        }
    } else if (act == SMX_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = pre[e] > 0.f ? x[e] : 0.f;
    }
}

// ---- per-step dropout key (round 5) ----
// A captured HIP graph bakes every kernel argument, dropout seeds included.  So that a replayed step still draws fresh masks, the
// seed a kernel hashes with is (its argument + the STEP KEY), and the key is one device word per translation unit that
// smx_set_step_key (misc.hip) rewrites once per training step - on the stream, ahead of the step's first kernel.  The word is 0
// until somebody sets it: launches outside the keyed engine mode (tests that pin a mask to its seed) behave as before.
// Kernels read it ONCE, on entry (smx_dseed), never inside a loop.
// (one word per translation unit = per code object: the build passes -DSMX_TU=<file stem>, which makes the symbol's name unique -
// a `static` device variable has no entry in the code object's symbol table, so hipGetSymbolAddress cannot find it)
#ifndef SMX_TU
#define SMX_TU anon
#endif
#define SMX_CAT2(a, b) a##b
#define SMX_CAT(a, b) SMX_CAT2(a, b)
#define SMX_KEYVAR SMX_CAT(smx_step_key_word_, SMX_TU)
__device__ unsigned SMX_KEYVAR = 0u;
__device__ __forceinline__ unsigned smx_dseed(float drop_p, unsigned seed) { return drop_p > 0.f ? seed + SMX_KEYVAR : seed; }
// every translation unit whose kernels hash exports the address of its copy (collected by smx_set_step_key)
#define SMX_STEP_KEY_TU(tag)                                                                                            \
    extern "C" int smx_step_key_addr_##tag(void** out) {                                                                \
        return hipGetSymbolAddress(out, HIP_SYMBOL(SMX_KEYVAR)) == hipSuccess ? SMX_OK : -5;                            \
    }

// ---- counter-based dropout: keep(idx) is a pure function of (seed, element index), so backward regenerates
// the forward mask instead of storing it (TF: nn.functional.dropout sites listed in engine.py) ----
// 32-bit integer mixes are quarter-rate on the vector unit (v_mul_lo_u32), and the GEMM epilogues hash every output element
// of a dropped Linear: two multiplies per hash (the "lowbias32" finaliser) and TWO elements per hash (element idx takes the
// low / high 16 bits of hash(idx >> 1); thresholds are compared at 16 bits: p is honoured to 2^-16).
__device__ __forceinline__ unsigned smx_hash32(unsigned seed, unsigned idx) {
    unsigned x = idx + seed * 0x9E3779B1u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// returns the multiplier: 0 (dropped) or 1/(1-p) (kept)
__device__ __forceinline__ float smx_drop_mul(unsigned seed, unsigned idx, unsigned thresh24, float inv_keep) {
    const unsigned h = smx_hash32(seed, idx >> 1);
    return ((idx & 1u) ? h >> 16 : h & 0xffffu) >= (thresh24 >> 8) ? inv_keep : 0.f;
}
// m[e] = multiplier(idx + e) for 8 consecutive elements starting at an EVEN idx: four hashes
__device__ __forceinline__ void smx_drop_mults8(unsigned seed, unsigned idx, unsigned thresh24, float inv_keep, float m[8]) {
    const unsigned th = thresh24 >> 8;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const unsigned h = smx_hash32(seed, (idx >> 1) + (e >> 1));
        m[e] = (h & 0xffffu) >= th ? inv_keep : 0.f;
        m[e + 1] = (h >> 16) >= th ? inv_keep : 0.f;
    }
}
// x[e] *= multiplier(idx + e) for 8 consecutive elements starting at an EVEN idx: four hashes
__device__ __forceinline__ void smx_drop_mul8(unsigned seed, unsigned idx, unsigned thresh24, float inv_keep, float x[8]) {
    const unsigned th = thresh24 >> 8;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const unsigned h = smx_hash32(seed, (idx >> 1) + (e >> 1));
        x[e] *= (h & 0xffffu) >= th ? inv_keep : 0.f;
        x[e + 1] *= (h >> 16) >= th ? inv_keep : 0.f;
    }
}
__host__ __device__ __forceinline__ unsigned smx_thresh24(float p) { return (unsigned)(p * 16777216.0f); }

// ---- wave / block reductions (64-wide) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `sh` needs 16 floats
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// ---- row view: maps a logical row index to an element offset.  Lets a GEMM operand be an
// overlapping / strided window over a [batch, time, channel] activation (conv as GEMM without im2col).
struct SmxRowView {
    long long batch_stride;  // elements between consecutive batches
    long long ld;            // elements between consecutive rows inside a batch
    long long off;           // constant element offset
    int rows_per_batch;      // logical rows per batch (<=0: single batch)
    int _pad;
};
__device__ __forceinline__ long long view_off(const SmxRowView& v, int r) {
    if (v.rows_per_batch > 0) {
        const int b = r / v.rows_per_batch;
        const int t = r - b * v.rows_per_batch;
        return v.off + (long long)b * v.batch_stride + (long long)t * v.ld;
    }
    return v.off + (long long)r * v.ld;
}
