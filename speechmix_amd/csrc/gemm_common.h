// Shared pieces of the bf16 GEMM kernels (gemm.hip: 128x128 family; gemm_pp.hip: 256-wide ping-pong kernel):
// the parameter block of the C ABI, the LDS tile images, fragment reads, LDS-DMA helpers and the row epilogue.
#pragma once
#include "smx_common.h"

struct SmxGemmParams {
    const void* A;
    const void* B;
    void* C;
    const float* bias;      // [N] fp32 or null
    const void* resid;      // same view as C (dtype = in dtype) or null: C += resid
    void* aux_out;          // pre-activation copy (same view as C) or null
    const void* aux_in;     // pre-activation of the consumer: C *= act'(aux_in) (same view as C) or null
    SmxRowView a, b, c;
    SmxRowView e;           // view of the epilogue side tensors (resid / aux_out / aux_in); usually == c
    long long batch_a, batch_b, batch_c, batch_bias, batch_e;  // element strides between grid.z batches
    int M, N, K;
    int a_rc, b_rc;
    int act;                // SMX_ACT_* applied after bias (fwd) or used for aux_in derivative
    int out_f32;            // C is fp32 regardless of input dtype
    int atomic;             // 0: C = .., 1: C += via fp32 atomics, 2: C += by plain read-modify-write (split_k == 1)
    int nbatch, split_k;    // split_k > 1 with atomic == 0: split s writes its partial to C + s * split_stride ("slabs")
    int tr_mode;            // 1: 128x128 LDS-DMA kernel, 9: its 64x128 form, 8: 256x256 ping-pong kernel (gemm_pp.hip), 2: register-staged + tr reads, 0: 16-bit LDS reads (debug)
    float alpha;
    long long split_stride; // elements between split-K slabs (atomic == 0)
    float drop_p;           // dropout applied after the activation and before the residual add (0: off); in the
    unsigned drop_seed;     // aux_in (backward-through-activation) mode it multiplies by the same forward mask
};

#define BM 128
#define BN 128
#define BK 64
#define GROUP_N 8
#define KC_TILE_BYTES (128 * 128)            // 128 rows x 64 bf16
#define RC_TILE_BYTES (64 * 256)             // 64 k-rows x 128 bf16
#define STAGE_BYTES (2 * 16384)

__device__ __forceinline__ int kc_addr(int row, int chunk) {  // chunk: 16-B unit 0..7
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
__device__ __forceinline__ int rc_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int rc_addr(int k, int col) {      // col: element 0..127
    return k * 256 + ((((col >> 4) ^ rc_swz(k)) << 5) | ((col & 15) << 1));
}

// gfx950 transposing LDS read: within each 16-lane group the 16 lanes x 4 b16 addressed by the lanes
// form a [4][16] block (lanes 4q..4q+3 supply row q); lane i receives column i (4 values, k = 0..3).
__device__ __forceinline__ uint2 lds_tr_b64(const char* p) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    union { s16x4_t v; uint2 u; } r;
    r.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p));
    return r.u;
}

template <bool RC>
struct TileLoader {
    // per-thread staging: 4 x 16 B
    uint4 r[4];
    long long off[4];   // KC: element offset of my 4 rows (chunk added); RC: unused
    bool ok[4];

    __device__ __forceinline__ void init(const SmxRowView& v, int row0, int nrows, int tid) {
        if (!RC) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = row0 + (tid >> 3) + 32 * p;
                ok[p] = row < nrows;
                off[p] = view_off(v, ok[p] ? row : 0) + (tid & 7) * 8;
            }
        }
    }
    // KC: rows fixed, k advances.  RC: k-rows advance (view applied per k-row), cols fixed.
    __device__ __forceinline__ void load(const bf16_t* base, const SmxRowView& v, int row0, int nrows, int k0, int K,
                                         int tid) {
        if (!RC) {
            const int kk = k0 + (tid & 7) * 8;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (ok[p] && kk < K) r[p] = *reinterpret_cast<const uint4*>(base + off[p] + k0);
                else r[p] = make_uint4(0, 0, 0, 0);
            }
        } else {
            const int col = row0 + (tid & 15) * 8;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int k = k0 + (tid >> 4) + 16 * p;
                if (k < K && col < nrows) r[p] = *reinterpret_cast<const uint4*>(base + view_off(v, k) + col);
                else r[p] = make_uint4(0, 0, 0, 0);
            }
        }
    }
    __device__ __forceinline__ void store(char* tile, int tid) const {
        if (!RC) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<uint4*>(tile + kc_addr((tid >> 3) + 32 * p, tid & 7)) = r[p];
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                *reinterpret_cast<uint4*>(tile + rc_addr((tid >> 4) + 16 * p, (tid & 15) * 8)) = r[p];
        }
    }
};

// fragment for the 16 rows starting at `r16` of the tile, K sub-step kk (0/1): 8 bf16 for k = 32kk+8g+e
template <bool RC>
__device__ __forceinline__ bf16x8_t load_frag(const char* tile, int r16, int kk, int lane, int tr_mode) {
    const int i = lane & 15, g = lane >> 4;
    union { bf16x8_t v; uint4 u; uint2 h[2]; bf16_t s[8]; } f;
    if (!RC) {
        f.u = *reinterpret_cast<const uint4*>(tile + kc_addr(r16 + i, kk * 4 + g));
    } else if (tr_mode) {
        // 16-lane group g reads a [4 k][16 col] block: lane supplies the 8-B address of (k = q, cols 4c..4c+3)
        const int q = i >> 2, c4 = (i & 3) * 4;
        const int kb = kk * 32 + 8 * g + q;
        f.h[0] = lds_tr_b64(tile + rc_addr(kb, r16 + c4));
        f.h[1] = lds_tr_b64(tile + rc_addr(kb + 4, r16 + c4));
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            f.s[e] = *reinterpret_cast<const bf16_t*>(tile + rc_addr(kk * 32 + 8 * g + e, r16 + i));
    }
    return f.v;
}

static __device__ uint4 smx_zero_page[4];

typedef __attribute__((address_space(3))) void* lds_vp_t;
typedef const __attribute__((address_space(1))) void* glb_vp_t;
__device__ __forceinline__ void glds16(const bf16_t* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_vp_t)src, (lds_vp_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void glds16_asm(const bf16_t* src, const char* lds_wave_base) {
    typedef __attribute__((address_space(3))) const char* lds_cp_t;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_cp_t)lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(dst)
                 : "memory");
}

// ------------------------------------------------------------------------------------------------
// Row-major epilogue for the MFMA kernels.  The accumulator layout (lane = one m, 4 consecutive n per 16x16 block)
// gives 8-byte accesses scattered over 16 rows and one address computation per block; measured on the FFN shapes
// that epilogue cost as much as the whole K loop.  Here each wave transposes its 64x64 fp32 sub-tile through a
// wave-private 8-KB LDS slice (two halves of 32 rows, XOR-swizzled 16-B chunks: conflict-free both ways), after
// which a lane owns 8 consecutive n of one row: one address per row visit, 16-B loads of resid / aux_in, 16-B
// stores of C / aux_out (128 contiguous bytes per 8 lanes), bias held in registers for the whole tile.
// ------------------------------------------------------------------------------------------------
// Stores issued from inline asm are invisible to hipcc's waitcnt pass.  The persistent ping-pong kernel uses them
// (ASM_ST) so that no compiler-tracked VMEM operation is pending when its K loop starts: with tracked stores in
// flight hipcc drains the whole queue - its LDS-DMA prefetches included - before the first LDS read of every K tile.
typedef __attribute__((ext_vector_type(4))) unsigned smx_u32x4_t;
// A store of more than 8 bytes reads its data registers for a few cycles after issue: a VALU write of one of them
// needs two wait states behind it (hipcc pads its own stores, it cannot see inside the asm) - without the s_nop a temporary
// written right behind the store showed up in 4 lanes of the output (found with the free-running schedule, round 3).
__device__ __forceinline__ void st_b128(void* p, uint4 v) {
    const smx_u32x4_t r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ void st_b32(void* p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_b16(void* p, unsigned v) {
    asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <bool ASM_ST>
__device__ __forceinline__ void st8(bf16_t* p, const float v[8]) {
    if (!ASM_ST) { store8(p, v); return; }
    st_b128(p, make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])));
}
template <bool ASM_ST>
__device__ __forceinline__ void st8(float* p, const float v[8]) {
    if (!ASM_ST) { store8(p, v); return; }
    st_b128(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
    st_b128(p + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
}

// SG: honour SMX_ACT_SAVE_GRAD (the split-K epilogue kernel; the GEMM kernels take flagged launches only through their
// class-specialised epilogues - the launcher checks - and keep this path's register footprint)
template <bool ASM_ST = false, bool SG = false>
__device__ __forceinline__ void epilogue_row8(const SmxGemmParams& p, long long zc, long long ze, int m, int n, float x[8],
                                              const float bs[8], unsigned th, float inv_keep, unsigned dseed) {
    // dseed: the seed the masks are hashed with = the launch's drop_seed + the step key (smx_dseed; the caller reads the key once)
    const long long base = zc + view_off(p.c, m) + n;
    const bool side = p.resid || p.aux_out || p.aux_in;
    const long long sb = side ? ze + view_off(p.e, m) + n : 0;
    const int nv = min(8, p.N - n);
    const bool fast = nv == 8 && !(base & 7) && !(sb & 7);
    const bf16_t* aux_in = reinterpret_cast<const bf16_t*>(p.aux_in);
    const bf16_t* resid = reinterpret_cast<const bf16_t*>(p.resid);
    bf16_t* aux_out = reinterpret_cast<bf16_t*>(p.aux_out);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], p.alpha, bs[e]);
    const int act = p.act & 0xff;
    const bool sg = SG && (p.act & SMX_ACT_SAVE_GRAD) != 0;  // side tensor = local derivative (see smx_common.h)
    if (fast) {
        if (aux_out && !sg) st8<ASM_ST>(aux_out + sb, x);
        const bool fused = sg && aux_out && !aux_in;       // forward with saved derivative: activation, mask and side store together
        if (aux_in) {
            float a[8];
            load8(aux_in + sb, a);
            if (sg) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] *= a[e];
            } else {
                act_grad_mul8(x, a, act);
            }
        } else if (fused) {
            const uint4 d = act_fwd_grad_drop8(x, act, p.drop_p > 0.f, dseed, (unsigned)((long long)m * p.N + n + zc), th, inv_keep);
            if (ASM_ST) st_b128(aux_out + sb, d); else *reinterpret_cast<uint4*>(aux_out + sb) = d;
        } else if (act) {
            act_fwd8(x, act);
        }
        if (p.drop_p > 0.f && !sg) {
            const unsigned idx = (unsigned)((long long)m * p.N + n + zc);
            if (!(idx & 1u)) smx_drop_mul8(dseed, idx, th, inv_keep, x);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] *= smx_drop_mul(dseed, idx + e, th, inv_keep);
        }
        if (resid) {
            float r[8];
            load8(resid + sb, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += r[e];
        }
        if (!p.out_f32) {
            st8<ASM_ST>(reinterpret_cast<bf16_t*>(p.C) + base, x);
        } else {
            float* c = reinterpret_cast<float*>(p.C) + base;
            if (!ASM_ST && p.atomic == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(c + e, x[e]);
            } else {
                if (p.atomic == 2) {
                    float o[8];
                    load8(c, o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] += o[e];
                }
                st8<ASM_ST>(c, x);
            }
        }
        return;
    }
    // ragged / unaligned tail (LM head with V % 8 != 0, odd views): element-wise
    for (int e = 0; e < nv; ++e) {
        float v = x[e];
        const float mk = (p.drop_p > 0.f && !(aux_in && sg)) ? smx_drop_mul(dseed, (unsigned)((long long)m * p.N + n + e + zc), th, inv_keep) : 1.f;
        if (aux_out) {
            const float s = sg ? act_grad(v, act) * mk : v;
            if (ASM_ST) st_b16(aux_out + sb + e, f2bf(s)); else aux_out[sb + e] = f2bf(s);
        }
        if (aux_in) v *= sg ? bf2f(aux_in[sb + e]) : act_grad(bf2f(aux_in[sb + e]), act);
        else v = act_fwd(v, act);
        v *= mk;
        if (resid) v += bf2f(resid[sb + e]);
        if (!p.out_f32) {
            if (ASM_ST) st_b16(reinterpret_cast<bf16_t*>(p.C) + base + e, f2bf(v));
            else reinterpret_cast<bf16_t*>(p.C)[base + e] = f2bf(v);
        } else {
            float* c = reinterpret_cast<float*>(p.C) + base + e;
            if (!ASM_ST && p.atomic == 1) atomicAdd(c, v);
            else {
                if (p.atomic == 2) v += *c;
                if (ASM_ST) st_b32(c, __float_as_uint(v)); else *c = v;
            }
        }
    }
}

// Epilogue classes with compile-time specialised code paths (everything else takes the generic row epilogue):
//   LINEAR: bias, dropout, residual -> bf16      ACT: bias, pre-activation copy, activation, dropout -> bf16
//   ACTGRAD: * act'(aux_in), forward dropout mask -> bf16      F32: alpha * acc + bias -> fp32 (plain or read-modify-write)
enum { PP_EPI_LINEAR = 0, PP_EPI_ACT = 1, PP_EPI_ACTGRAD = 2, PP_EPI_F32 = 3 };
// epilogue class of a parameter block, or -1 when only the generic epilogue applies
static inline int pp_epi_class(const SmxGemmParams& p) {
    if (p.out_f32) return (p.aux_in || p.aux_out || p.resid || p.act || p.drop_p > 0.f) ? -1 : PP_EPI_F32;
    if (p.atomic) return -1;
    if (p.aux_in) return (p.resid || p.aux_out) ? -1 : PP_EPI_ACTGRAD;
    if (p.act || p.aux_out) return p.resid ? -1 : PP_EPI_ACT;
    return PP_EPI_LINEAR;
}

// SMX_ACT_SAVE_GRAD launches the 256-wide kernels accept (round 3): the two classes that carry the saved local derivative, on
// aligned views only (their generic row epilogue does not know the flag) - forward ACT of a (KC, KC) launch, ACTGRAD of a
// bias-free (KC, RC) data gradient
static inline bool smx_epi_views_aligned(const SmxGemmParams& p);
static inline bool pp_saved_ok(const SmxGemmParams& p) {
    if (!(p.act & SMX_ACT_SAVE_GRAD)) return true;
    if (p.a_rc || !smx_epi_views_aligned(p) || p.atomic || p.out_f32) return false;
    const int epi = pp_epi_class(p);
    return (!p.b_rc && epi == PP_EPI_ACT) || (p.b_rc && epi == PP_EPI_ACTGRAD && !p.bias);
}

// the specialised epilogues use 16-B accesses on every view: all strides / offsets multiples of 8 elements, N % 8 == 0
static inline bool smx_epi_views_aligned(const SmxGemmParams& p) {
    const long long m = p.c.ld | p.c.off | p.c.batch_stride | p.e.ld | p.e.off | p.e.batch_stride | p.batch_c | p.batch_e |
                        p.split_stride | p.batch_bias;
    return !(m & 7) && !(p.N & 7);
}
