// Fused multi-head attention forward/backward for the SpeechMix hot path (gfx950).
//   softmax(Q K^T * scale + bias [+ causal]) V, no padding mask (the reference never passes one:
//   ref:speechmix/model.py:148, 135-136).  TF:models/wav2vec2/modeling_wav2vec2.py:466-548,
//   TF:models/bart/modeling_bart.py:133-257 (self / causal / cross), TF:models/t5/modeling_t5.py:176-369
//   (scale 1.0 + additive relative-position bias), TF:integrations/sdpa_attention.py:39-130.
//
// bf16 kernels (head_dim 64, every head on the path): flash-style, LDS-staged 64-row K/V (or Q/dO) tiles,
// v_mfma_f32_16x16x32_bf16.  The score tile is produced TRANSPOSED (S^T = K Q^T) so each lane owns one
// query column: softmax statistics are per-lane scalars and the exponentiated tile is already in the
// B-operand layout of the following P·V product; the matching A operand (V^T, K^T, Q^T, dO^T) comes from
// the row-major LDS tile through ds_read_b64_tr_b16.  No shuffles, no P round trip through LDS.
// Backward = two kernels (dK/dV owner-computes over key tiles, dQ owner-computes over query tiles), both
// recomputing S from the saved log-sum-exp: deterministic, no atomics.
// fp32 kernels: simple one-thread-per-row loops (parity path / on-device cross-check).
#include "smx_common.h"

struct SmxAttnParams {
    const void* Q; const void* K; const void* V;
    void* O;
    float* lse;           // [B, H, Tq]
    const float* bias;    // optional [H, Tq, Tk] fp32 additive bias
    const void* dO; void* dQ; void* dK; void* dV;   // backward only
    float* delta;         // [B, H, Tq] backward scratch: sum_d dO*O
    float* dbias;         // optional [H, Tq, Tk] fp32: += sum over the batch of dS (attn_dbias_kernel)
    long long q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld;       // element strides (batch, row)
    long long dq_bs, dq_ld, dk_bs, dk_ld, dv_bs, dv_ld, do_bs, do_ld;
    int B, H, Tq, Tk, D;
    int causal;
    float scale;
    float drop_p;         // dropout on the attention probabilities (0: off); mask index = ((b*H+h)*Tq+q)*Tkp+key, Tkp = Tk rounded up to a multiple of 4
    unsigned drop_seed;
    // MFMA path with dropout: the keep mask as bit matrices, written by smx_attn_dropout_mask (attention_v2.h):
    unsigned* mask_q;     // [B*H*Tq][2*ceil(Tk/64)] words, bit j of word w = key 32 w + j   (forward, dQ)
    unsigned* mask_k;     // [B*H*Tk][2*ceil(Tq/64)] words, bit j of word w = query 32 w + j (dK/dV)
    // Optional per-clip key length (a right-padded attention mask, TF:models/wav2vec2/modeling_wav2vec2.py:688-697,
    // TF:models/bart/modeling_bart.py:741-760; the reference's hook ref:speechmix/model.py:132-136): keys at positions >=
    // klen[b] get probability 0 in forward and backward.  null: every key up to Tk counts.  1 <= klen[b] <= Tk.
    const int* klen;
};
__device__ __forceinline__ int att_tk(const SmxAttnParams& p, int b) { return p.klen ? min(p.klen[b], p.Tk) : p.Tk; }
// Mask index of probability (b, h, q, key): ((b H + h) Tq + q) Tkp + key with Tkp = Tk rounded up to a multiple of 4, so
// that the 4 consecutive keys a lane holds per 16x16 score block (first key % 4 == 0) are 4 consecutive, 4-aligned
// indices: TWO hashes (the mask function yields two elements per hash, smx_common.h) instead of four, and one row-base
// computation per query instead of a 64-bit index per element.  Attention dropout hashes every score; in train mode
// that was as much vector work as the softmax itself.
__device__ __forceinline__ unsigned att_row_base(const SmxAttnParams& p, int b, int h, int q) {
    return (unsigned)((((long long)b * p.H + h) * p.Tq + q) * (long long)((p.Tk + 3) & ~3));
}
#define ATT_DROP(p, b, h, q, k) \
    smx_drop_mul((p).drop_seed, att_row_base(p, b, h, q) + (unsigned)(k), smx_thresh24((p).drop_p), 1.0f / (1.0f - (p).drop_p))
// multipliers of keys key0 .. key0 + 3 (key0 % 4 == 0) of the row with base rb
__device__ __forceinline__ void att_drop4(const SmxAttnParams& p, unsigned rb, int key0, float (&dm)[4]) {
    const unsigned th = smx_thresh24(p.drop_p) >> 8;
    const float inv = 1.0f / (1.0f - p.drop_p);
    const unsigned i2 = (rb + (unsigned)key0) >> 1;
    const unsigned h0 = smx_hash32(p.drop_seed, i2), h1 = smx_hash32(p.drop_seed, i2 + 1);
    dm[0] = (h0 & 0xffffu) >= th ? inv : 0.f;
    dm[1] = (h0 >> 16) >= th ? inv : 0.f;
    dm[2] = (h1 & 0xffffu) >= th ? inv : 0.f;
    dm[3] = (h1 >> 16) >= th ? inv : 0.f;
}

#define NEG_BIG (-1e30f)

// ============================== fp32 simple kernels ==============================================
#define SIMPLE_MAXD 128
template <typename T>
__global__ void attn_fwd_simple(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.B * p.H * p.Tq) return;
    const int q = idx % p.Tq, h = (idx / p.Tq) % p.H, b = idx / (p.Tq * p.H);
    const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + q * p.q_ld + h * p.D;
    const T* K = reinterpret_cast<const T*>(p.K) + b * p.k_bs + h * p.D;
    const T* V = reinterpret_cast<const T*>(p.V) + b * p.v_bs + h * p.D;
    float o[SIMPLE_MAXD];
    for (int d = 0; d < p.D; ++d) o[d] = 0.f;
    float m = NEG_BIG, l = 0.f;
    const int kmax = min(att_tk(p, b), p.causal ? min(p.Tk, q + (p.Tk - p.Tq) + 1) : p.Tk);
    for (int k = 0; k < kmax; ++k) {
        float s = 0.f;
        for (int d = 0; d < p.D; ++d) s = fmaf(Cvt<T>::ld(Q + d), Cvt<T>::ld(K + k * p.k_ld + d), s);
        s *= p.scale;
        if (p.bias) s += p.bias[((long long)h * p.Tq + q) * p.Tk + k];
        const float mn = fmaxf(m, s);
        const float a = expf(m - mn), e = expf(s - mn);
        l = l * a + e;
        const float ed = p.drop_p > 0.f ? e * ATT_DROP(p, b, h, q, k) : e;
        for (int d = 0; d < p.D; ++d) o[d] = o[d] * a + ed * Cvt<T>::ld(V + k * p.v_ld + d);
        m = mn;
    }
    T* O = reinterpret_cast<T*>(p.O) + b * p.o_bs + q * p.o_ld + h * p.D;
    const float inv = 1.f / l;
    for (int d = 0; d < p.D; ++d) Cvt<T>::st(O + d, o[d] * inv);
    p.lse[idx] = m + logf(l);
}

template <typename T>
__global__ void attn_delta_kernel(SmxAttnParams p) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.B * p.H * p.Tq) return;
    const int q = idx % p.Tq, h = (idx / p.Tq) % p.H, b = idx / (p.Tq * p.H);
    const T* O = reinterpret_cast<const T*>(p.O) + b * p.o_bs + q * p.o_ld + h * p.D;
    const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.do_bs + q * p.do_ld + h * p.D;
    float s = 0.f;
    for (int d = 0; d < p.D; d += 8) {
        float a[8], c[8];
        load8(O + d, a);
        load8(dO + d, c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(a[e], c[e], s);
    }
    p.delta[idx] = s;
}

template <typename T>
__global__ void attn_bwd_dq_simple(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.B * p.H * p.Tq) return;
    const int q = idx % p.Tq, h = (idx / p.Tq) % p.H, b = idx / (p.Tq * p.H);
    const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + q * p.q_ld + h * p.D;
    const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.do_bs + q * p.do_ld + h * p.D;
    const T* K = reinterpret_cast<const T*>(p.K) + b * p.k_bs + h * p.D;
    const T* V = reinterpret_cast<const T*>(p.V) + b * p.v_bs + h * p.D;
    const float lse = p.lse[idx], delta = p.delta[idx];
    float dq[SIMPLE_MAXD];
    for (int d = 0; d < p.D; ++d) dq[d] = 0.f;
    const int kmax = min(att_tk(p, b), p.causal ? min(p.Tk, q + (p.Tk - p.Tq) + 1) : p.Tk);
    for (int k = 0; k < kmax; ++k) {
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < p.D; ++d) {
            s = fmaf(Cvt<T>::ld(Q + d), Cvt<T>::ld(K + k * p.k_ld + d), s);
            dp = fmaf(Cvt<T>::ld(dO + d), Cvt<T>::ld(V + k * p.v_ld + d), dp);
        }
        s *= p.scale;
        if (p.bias) s += p.bias[((long long)h * p.Tq + q) * p.Tk + k];
        if (p.drop_p > 0.f) dp *= ATT_DROP(p, b, h, q, k);
        const float ds = expf(s - lse) * (dp - delta);
        for (int d = 0; d < p.D; ++d) dq[d] = fmaf(ds * p.scale, Cvt<T>::ld(K + k * p.k_ld + d), dq[d]);
    }
    T* dQ = reinterpret_cast<T*>(p.dQ) + b * p.dq_bs + q * p.dq_ld + h * p.D;
    for (int d = 0; d < p.D; ++d) Cvt<T>::st(dQ + d, dq[d]);
}

template <typename T>
__global__ void attn_bwd_dkv_simple(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.B * p.H * p.Tk) return;
    const int k = idx % p.Tk, h = (idx / p.Tk) % p.H, b = idx / (p.Tk * p.H);
    const T* Kp = reinterpret_cast<const T*>(p.K) + b * p.k_bs + k * p.k_ld + h * p.D;
    const T* Vp = reinterpret_cast<const T*>(p.V) + b * p.v_bs + k * p.v_ld + h * p.D;
    const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + h * p.D;
    const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.do_bs + h * p.D;
    float dk[SIMPLE_MAXD], dv[SIMPLE_MAXD];
    for (int d = 0; d < p.D; ++d) dk[d] = dv[d] = 0.f;
    const int q0 = k >= att_tk(p, b) ? p.Tq : (p.causal ? max(0, k - (p.Tk - p.Tq)) : 0);       // a padded key: no query sees it
    for (int q = q0; q < p.Tq; ++q) {
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < p.D; ++d) {
            s = fmaf(Cvt<T>::ld(Q + q * p.q_ld + d), Cvt<T>::ld(Kp + d), s);
            dp = fmaf(Cvt<T>::ld(dO + q * p.do_ld + d), Cvt<T>::ld(Vp + d), dp);
        }
        s *= p.scale;
        if (p.bias) s += p.bias[((long long)h * p.Tq + q) * p.Tk + k];
        const long long li = ((long long)b * p.H + h) * p.Tq + q;
        const float pr = expf(s - p.lse[li]);
        const float dm = p.drop_p > 0.f ? ATT_DROP(p, b, h, q, k) : 1.f;
        const float ds = pr * (dm * dp - p.delta[li]) * p.scale;
        for (int d = 0; d < p.D; ++d) {
            dv[d] = fmaf(pr * dm, Cvt<T>::ld(dO + q * p.do_ld + d), dv[d]);
            dk[d] = fmaf(ds, Cvt<T>::ld(Q + q * p.q_ld + d), dk[d]);
        }
    }
    T* dK = reinterpret_cast<T*>(p.dK) + b * p.dk_bs + k * p.dk_ld + h * p.D;
    T* dV = reinterpret_cast<T*>(p.dV) + b * p.dv_bs + k * p.dv_ld + h * p.D;
    for (int d = 0; d < p.D; ++d) { Cvt<T>::st(dK + d, dk[d]); Cvt<T>::st(dV + d, dv[d]); }
}

// Gradient of the additive score bias (T5's relative-position bias, TF:models/t5/modeling_t5.py:216-279, 329-345):
// dbias[h, q, k] += sum_b dS[b, h, q, k] with dS = P * (mask * dP - delta), S recomputed from the saved log-sum-exp.
// One thread owns one (h, q, k): no atomics, deterministic.  Used by the MFMA path (whose dQ / dK/dV kernels do not
// produce it); delta must have been written (the dQ kernel does).  The tables are tiny (T5 runs on S <= 249 keys).
template <typename T>
__global__ void attn_dbias_kernel(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)p.H * p.Tq * p.Tk) return;
    const int k = (int)(idx % p.Tk), q = (int)((idx / p.Tk) % p.Tq), h = (int)(idx / ((long long)p.Tk * p.Tq));
    if (p.causal && k > q + (p.Tk - p.Tq)) return;
    const float bias = p.bias ? p.bias[idx] : 0.f;
    float acc = 0.f;
    for (int b = 0; b < p.B; ++b) {
        if (k >= att_tk(p, b)) continue;
        const T* Q = reinterpret_cast<const T*>(p.Q) + b * p.q_bs + q * p.q_ld + h * p.D;
        const T* dO = reinterpret_cast<const T*>(p.dO) + b * p.do_bs + q * p.do_ld + h * p.D;
        const T* K = reinterpret_cast<const T*>(p.K) + b * p.k_bs + k * p.k_ld + h * p.D;
        const T* V = reinterpret_cast<const T*>(p.V) + b * p.v_bs + k * p.v_ld + h * p.D;
        float s = 0.f, dp = 0.f;
        for (int d = 0; d < p.D; d += 8) {
            float a[8], c[8], e[8], f[8];
            load8(Q + d, a); load8(K + d, c); load8(dO + d, e); load8(V + d, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s = fmaf(a[j], c[j], s); dp = fmaf(e[j], f[j], dp); }
        }
        s = s * p.scale + bias;
        if (p.drop_p > 0.f) dp *= ATT_DROP(p, b, h, q, k);
        const long long li = ((long long)b * p.H + h) * p.Tq + q;
        acc += expf(s - p.lse[li]) * (dp - p.delta[li]);
    }
    p.dbias[idx] += acc;
}

// dtable[bucket[q, k], h] += dbias[h, q, k]: one block per (bucket, head), fixed-order tree reduction (deterministic).
__global__ void attn_bias_scatter_kernel(const float* dbias, const int* bucket, float* dtable, int H, int QK) {
    const int bk = blockIdx.x, h = blockIdx.y;
    float acc = 0.f;
    for (int i = threadIdx.x; i < QK; i += blockDim.x)
        if (bucket[i] == bk) acc += dbias[(long long)h * QK + i];
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) dtable[bk * H + h] += red[0];
}

// ============================== bf16 MFMA kernels (D = 64) =======================================
// LDS tile: 64 rows x 64 bf16 (128-B rows), 16-B slots XOR-swizzled by (row>>1)&7 (conflict-free b128
// fragment reads; 8-B tr-read granules stay intact).
__device__ __forceinline__ int t_addr(int row, int d) {
    return row * 128 + ((((d >> 3) ^ ((row >> 1) & 7)) << 4) | ((d & 7) << 1));
}
__device__ __forceinline__ uint2 tr_b64(const char* p) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    union { s16x4_t v; uint2 u; } r;
    r.v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(p));
    return r.u;
}
// stage a [64][64] bf16 tile (rows row0.. of a [T, ld] matrix, zero-filled past nrows); 256 threads
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* base, long long ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        const int r = (tid >> 3) + 32 * pss, c = tid & 7;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < nrows) v = *reinterpret_cast<const uint4*>(base + (long long)(row0 + r) * ld + c * 8);
        *reinterpret_cast<uint4*>(tile + t_addr(r, c * 8)) = v;
    }
}
// KC fragment: lane (i = lane&15, g = lane>>4) gets row (r16+i), d = 32*kk + 8g .. +7
__device__ __forceinline__ bf16x8_t frag_kc(const char* tile, int r16, int kk, int lane) {
    union { bf16x8_t v; uint4 u; } f;
    f.u = *reinterpret_cast<const uint4*>(tile + t_addr(r16 + (lane & 15), kk * 32 + (lane >> 4) * 8));
    return f.v;
}
// transposed fragment for the "pair" reduction-slot mapping: k-slots e<4 -> row ra + 4g + e,
// e>=4 -> row rb + 4g + (e-4); lane i = lane&15 receives column (c16 + i).
__device__ __forceinline__ bf16x8_t frag_tr(const char* tile, int ra, int rb, int c16, int lane) {
    const int i = lane & 15, g = lane >> 4, qq = i >> 2, c4 = (i & 3) * 4;
    union { bf16x8_t v; uint2 h[2]; } f;
    f.h[0] = tr_b64(tile + t_addr(ra + 4 * g + qq, c16 + c4));
    f.h[1] = tr_b64(tile + t_addr(rb + 4 * g + qq, c16 + c4));
    return f.v;
}
__device__ __forceinline__ bf16x8_t pack_pair(const f32x4_t& a, const f32x4_t& b) {
    union { bf16x8_t v; unsigned u[4]; } f;
    f.u[0] = pack_bf2(a[0], a[1]); f.u[1] = pack_bf2(a[2], a[3]);
    f.u[2] = pack_bf2(b[0], b[1]); f.u[3] = pack_bf2(b[2], b[3]);
    return f.v;
}
__device__ __forceinline__ float group_max(float v) {   // across the 4 lane groups (same lane&15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
#define ZERO4 ((f32x4_t){0.f, 0.f, 0.f, 0.f})

#define SMX_LOG2E 1.44269504088896340736f
#define SMX_LN2 0.69314718055994530942f
// raw v_exp_f32 (2^x): the softmax runs in the log2 domain so that scale, shift and exponent are one v_fma + one v_exp
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// global -> registers -> LDS staging of a [64][64] bf16 tile, split so that the global loads of the NEXT tile are in
// flight while the current one is consumed (one barrier per tile, two LDS buffers)
__device__ __forceinline__ void tile_load(uint4 (&v)[2], const bf16_t* base, long long ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) {
        const int r = (tid >> 3) + 32 * pss, c = tid & 7;
        v[pss] = make_uint4(0, 0, 0, 0);
        if (row0 + r < nrows) v[pss] = *reinterpret_cast<const uint4*>(base + (long long)(row0 + r) * ld + c * 8);
    }
}
__device__ __forceinline__ void tile_store(char* tile, const uint4 (&v)[2], int tid) {
#pragma unroll
    for (int pss = 0; pss < 2; ++pss) *reinterpret_cast<uint4*>(tile + t_addr((tid >> 3) + 32 * pss, (tid & 7) * 8)) = v[pss];
}
__device__ __forceinline__ bf16x8_t load_row_frag(const bf16_t* base, long long ld, int row, int nrows, int kk, int g) {
    union { bf16x8_t v; uint4 u; } f;
    f.u = make_uint4(0, 0, 0, 0);
    if (row < nrows) f.u = *reinterpret_cast<const uint4*>(base + (long long)row * ld + kk * 32 + g * 8);
    return f.v;
}

// Forward: block = 64 U queries (wave: U 16-query column blocks), loops over 64-key tiles.  S^T = K Q^T so that the
// probabilities of a 16x16 block are directly the B operand of the P V product ("pair" k-slot mapping, see frag_tr).
template <int U>
__global__ __launch_bounds__(256, U == 1 ? 4 : 2) void attn_fwd_bf16(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    __shared__ __attribute__((aligned(16))) char sK[2][8192];
    __shared__ __attribute__((aligned(16))) char sV[2][8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, qw0 = blockIdx.x * (64 * U) + wave * (16 * U);
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    bf16x8_t qf[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[u][kk] = load_row_frag(Qp, p.q_ld, qw0 + u * 16 + i16, p.Tq, kk, g);
    f32x4_t o[U][4];
    float m[U], l[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[u][dt] = ZERO4;
#pragma unroll
    for (int u = 0; u < U; ++u) { m[u] = NEG_BIG; l[u] = 0.f; }
    const float sl2 = p.scale * SMX_LOG2E;
    const int coff = p.Tk - p.Tq;
    int kend = p.Tk;
    if (p.causal) kend = min(p.Tk, (int)blockIdx.x * (64 * U) + 64 * U + coff);   // keys beyond the block's last query are masked
    uint4 rk[2], rv[2];
    tile_load(rk, Kp, p.k_ld, 0, p.Tk, tid);
    tile_load(rv, Vp, p.v_ld, 0, p.Tk, tid);
    tile_store(sK[0], rk, tid);
    tile_store(sV[0], rv, tid);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < kend; k0 += 64) {
        const bool more = k0 + 64 < kend;
        if (more) {
            tile_load(rk, Kp, p.k_ld, k0 + 64, p.Tk, tid);
            tile_load(rv, Vp, p.v_ld, k0 + 64, p.Tk, tid);
        }
        const char* tK = sK[buf];
        const char* tV = sV[buf];
        f32x4_t s[U][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t k0f = frag_kc(tK, t * 16, 0, lane), k1f = frag_kc(tK, t * 16, 1, lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, qf[u][0], ZERO4, 0, 0, 0);
                s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, qf[u][1], s[u][t], 0, 0, 0);
            }
        }
        // masks only where they can bite: the ragged last key tile, and tiles crossing the causal diagonal
        const bool masked = (k0 + 64 > p.Tk) || (p.causal && k0 + 63 > (int)blockIdx.x * (64 * U) + coff);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = qw0 + u * 16 + i16;
            float mul = sl2;
            if (p.bias) {                                  // T5 relative-position bias: scores leave this block in log2 units
                mul = 1.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + t * 16 + 4 * g + r;
                        float v = s[u][t][r] * sl2;
                        if (q < p.Tq && key < p.Tk) v = fmaf(p.bias[((long long)h * p.Tq + q) * p.Tk + key], SMX_LOG2E, v);
                        s[u][t][r] = v;
                    }
            }
            if (masked) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + t * 16 + 4 * g + r;
                        if (key >= p.Tk || (p.causal && key > q + coff)) s[u][t][r] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[u][t][r]);
            mx = group_max(mx);
            const float mn = fmaxf(m[u], mx * mul);
            const float alpha = fast_exp2(m[u] - mn);
            float rs = 0.f;
            const bool drop = p.drop_p > 0.f;
            const unsigned rb = drop ? att_row_base(p, b, h, q) : 0u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (drop) att_drop4(p, rb, k0 + t * 16 + 4 * g, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(fmaf(s[u][t][r], mul, -mn));
                    rs += e;
                    s[u][t][r] = e * dm[r];
                }
            }
            l[u] = l[u] * alpha + group_sum(rs);
            m[u] = mn;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[u][dt] *= alpha;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {        // two 32-key reduction steps
            bf16x8_t vfr[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vfr[dt] = frag_tr(tV, st * 32, st * 32 + 16, dt * 16, lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bf16x8_t pf = pack_pair(s[u][2 * st], s[u][2 * st + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pf, o[u][dt], 0, 0, 0);
            }
        }
        if (more) {
            tile_store(sK[buf ^ 1], rk, tid);
            tile_store(sV[buf ^ 1], rv, tid);
        }
        __syncthreads();       // next tile published; everyone is done with this one before it is overwritten next round
        buf ^= 1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = qw0 + u * 16 + i16;
        if (q < p.Tq) {
            const float inv = 1.f / l[u];
            bf16_t* Op = reinterpret_cast<bf16_t*>(p.O) + b * p.o_bs + (long long)q * p.o_ld + h * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk = make_uint2(pack_bf2(o[u][dt][0] * inv, o[u][dt][1] * inv), pack_bf2(o[u][dt][2] * inv, o[u][dt][3] * inv));
                *reinterpret_cast<uint2*>(Op + dt * 16 + 4 * g) = pk;
            }
            if (g == 0) p.lse[((long long)b * p.H + h) * p.Tq + q] = m[u] * SMX_LN2 + __logf(l[u]);   // natural-log units
        }
    }
}

// dQ: block owns 64 U queries (wave: U x 16), loops over key tiles.  Also produces delta = rowsum(dO * O) for the
// dK/dV kernel that follows it on the stream.
template <int U>
__global__ __launch_bounds__(256, U == 1 ? 3 : 1) void attn_bwd_dq_bf16(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    __shared__ __attribute__((aligned(16))) char sK[2][8192];
    __shared__ __attribute__((aligned(16))) char sV[2][8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, qw0 = blockIdx.x * (64 * U) + wave * (16 * U);
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* dOp = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * 64;
    const bf16_t* Op = reinterpret_cast<const bf16_t*>(p.O) + b * p.o_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    bf16x8_t qf[U][2], dof[U][2];
    float nlse2[U], delta[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = qw0 + u * 16 + i16;
        float dsum = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[u][kk] = load_row_frag(Qp, p.q_ld, q, p.Tq, kk, g);
            dof[u][kk] = load_row_frag(dOp, p.do_ld, q, p.Tq, kk, g);
            if (q < p.Tq) {
                float ov[8], dv[8];
                load8(Op + (long long)q * p.o_ld + kk * 32 + g * 8, ov);
                load8(dOp + (long long)q * p.do_ld + kk * 32 + g * 8, dv);
#pragma unroll
                for (int e = 0; e < 8; ++e) dsum = fmaf(ov[e], dv[e], dsum);
            }
        }
        delta[u] = group_sum(dsum);
        nlse2[u] = 0.f;
        if (q < p.Tq) {
            const long long li = ((long long)b * p.H + h) * p.Tq + q;
            nlse2[u] = -p.lse[li] * SMX_LOG2E;
            if (g == 0) p.delta[li] = delta[u];
        }
    }
    f32x4_t dq[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[u][dt] = ZERO4;
    const float sl2 = p.scale * SMX_LOG2E;
    const int coff = p.Tk - p.Tq;
    int kend = p.Tk;
    if (p.causal) kend = min(p.Tk, (int)blockIdx.x * (64 * U) + 64 * U + coff);
    uint4 rk[2], rv[2];
    tile_load(rk, Kp, p.k_ld, 0, p.Tk, tid);
    tile_load(rv, Vp, p.v_ld, 0, p.Tk, tid);
    tile_store(sK[0], rk, tid);
    tile_store(sV[0], rv, tid);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < kend; k0 += 64) {
        const bool more = k0 + 64 < kend;
        if (more) {
            tile_load(rk, Kp, p.k_ld, k0 + 64, p.Tk, tid);
            tile_load(rv, Vp, p.v_ld, k0 + 64, p.Tk, tid);
        }
        const char* tK = sK[buf];
        const char* tV = sV[buf];
        const bool masked = (k0 + 64 > p.Tk) || ((int)blockIdx.x * (64 * U) + 64 * U > p.Tq) ||
                            (p.causal && k0 + 63 > (int)blockIdx.x * (64 * U) + coff);
        f32x4_t ds[U][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8_t k0f = frag_kc(tK, t * 16, 0, lane), k1f = frag_kc(tK, t * 16, 1, lane);
            const bf16x8_t v0f = frag_kc(tV, t * 16, 0, lane), v1f = frag_kc(tV, t * 16, 1, lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = qw0 + u * 16 + i16;
                f32x4_t sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0f, qf[u][0], ZERO4, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1f, qf[u][1], sc, 0, 0, 0);
                f32x4_t dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0f, dof[u][0], ZERO4, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1f, dof[u][1], dp, 0, 0, 0);
                float dm4[4] = {1.f, 1.f, 1.f, 1.f};
                if (p.drop_p > 0.f) att_drop4(p, att_row_base(p, b, h, q), k0 + t * 16 + 4 * g, dm4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + t * 16 + 4 * g + r;
                    float off = nlse2[u];
                    if (p.bias && q < p.Tq && key < p.Tk) off = fmaf(p.bias[((long long)h * p.Tq + q) * p.Tk + key], SMX_LOG2E, off);
                    float pr = fast_exp2(fmaf(sc[r], sl2, off));
                    if (masked && (key >= p.Tk || q >= p.Tq || (p.causal && key > q + coff))) pr = 0.f;
                    ds[u][t][r] = pr * (dm4[r] * dp[r] - delta[u]);
                }
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8_t kfr[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) kfr[dt] = frag_tr(tK, st * 32, st * 32 + 16, dt * 16, lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bf16x8_t pf = pack_pair(ds[u][2 * st], ds[u][2 * st + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[dt], pf, dq[u][dt], 0, 0, 0);
            }
        }
        if (more) {
            tile_store(sK[buf ^ 1], rk, tid);
            tile_store(sV[buf ^ 1], rv, tid);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int q = qw0 + u * 16 + i16;
        if (q < p.Tq) {
            bf16_t* dQp = reinterpret_cast<bf16_t*>(p.dQ) + b * p.dq_bs + (long long)q * p.dq_ld + h * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk = make_uint2(pack_bf2(dq[u][dt][0] * p.scale, dq[u][dt][1] * p.scale),
                                      pack_bf2(dq[u][dt][2] * p.scale, dq[u][dt][3] * p.scale));
                *reinterpret_cast<uint2*>(dQp + dt * 16 + 4 * g) = pk;
            }
        }
    }
}

// dK/dV: block owns 64 U keys (wave: U x 16), loops over 64-query tiles.  Scores are produced UN-transposed here
// (S = Q K^T: lane owns one key column) so that P and dS are again directly the B operand of the
// reductions over queries.
template <int U>
__global__ __launch_bounds__(256, U == 1 ? 2 : 1) void attn_bwd_dkv_bf16(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    __shared__ __attribute__((aligned(16))) char sQ[2][8192];
    __shared__ __attribute__((aligned(16))) char sDO[2][8192];
    __shared__ float sNlse2[2][64];
    __shared__ float sDelta[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, kw0 = blockIdx.x * (64 * U) + wave * (16 * U);
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* dOp = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    const long long rowbase = ((long long)b * p.H + h) * p.Tq;
    bf16x8_t kf[U][2], vf[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kf[u][kk] = load_row_frag(Kp, p.k_ld, kw0 + u * 16 + i16, p.Tk, kk, g);
            vf[u][kk] = load_row_frag(Vp, p.v_ld, kw0 + u * 16 + i16, p.Tk, kk, g);
        }
    f32x4_t dk[U][4], dv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[u][dt] = dv[u][dt] = ZERO4;
    const float sl2 = p.scale * SMX_LOG2E;
    const int coff = p.Tk - p.Tq;
    int qbeg = 0;
    if (p.causal) qbeg = max(0, (int)(blockIdx.x * (64 * U)) - coff) & ~63;   // queries before this see none of the block's keys
    uint4 rq[2], rd[2];
    float rl = 0.f, rdl = 0.f;
    tile_load(rq, Qp, p.q_ld, qbeg, p.Tq, tid);
    tile_load(rd, dOp, p.do_ld, qbeg, p.Tq, tid);
    tile_store(sQ[0], rq, tid);
    tile_store(sDO[0], rd, tid);
    if (tid < 64) {
        const int qq = qbeg + tid;
        sNlse2[0][tid] = qq < p.Tq ? -p.lse[rowbase + qq] * SMX_LOG2E : 0.f;
        sDelta[0][tid] = qq < p.Tq ? p.delta[rowbase + qq] : 0.f;
    }
    __syncthreads();
    int buf = 0;
    for (int q0 = qbeg; q0 < p.Tq; q0 += 64) {
        const bool more = q0 + 64 < p.Tq;
        if (more) {
            tile_load(rq, Qp, p.q_ld, q0 + 64, p.Tq, tid);
            tile_load(rd, dOp, p.do_ld, q0 + 64, p.Tq, tid);
            if (tid < 64) {
                const int qq = q0 + 64 + tid;
                rl = qq < p.Tq ? -p.lse[rowbase + qq] * SMX_LOG2E : 0.f;
                rdl = qq < p.Tq ? p.delta[rowbase + qq] : 0.f;
            }
        }
        const char* tQ = sQ[buf];
        const char* tDO = sDO[buf];
        const bool masked = (q0 + 64 > p.Tq) || ((int)blockIdx.x * (64 * U) + 64 * U > p.Tk) ||
                            (p.causal && (int)blockIdx.x * (64 * U) + 64 * U - 1 > q0 + coff);
        f32x4_t pt[U][4], ds[U][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {          // 16-query sub-tiles: D rows = queries 4g+r, cols = keys
            const bf16x8_t q0f = frag_kc(tQ, t * 16, 0, lane), q1f = frag_kc(tQ, t * 16, 1, lane);
            const bf16x8_t d0f = frag_kc(tDO, t * 16, 0, lane), d1f = frag_kc(tDO, t * 16, 1, lane);
            float nl[4], dl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nl[r] = sNlse2[buf][t * 16 + 4 * g + r];
                dl[r] = sDelta[buf][t * 16 + 4 * g + r];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int key = kw0 + u * 16 + i16;
                f32x4_t sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[u][0], ZERO4, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[u][1], sc, 0, 0, 0);
                f32x4_t dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[u][0], ZERO4, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[u][1], dp, 0, 0, 0);
                // dropout multipliers of my 4 queries x my key.  Keys 2j and 2j+1 of a row share a hash and sit in
                // neighbouring lanes (kw0 + 16 u is even): the even lane hashes queries r = 0, 1, the odd lane r = 2, 3, and
                // a DPP quad swap hands each its partner's pair - two hashes per lane instead of four.
                float dm4[4] = {1.f, 1.f, 1.f, 1.f};
                if (p.drop_p > 0.f) {
                    const unsigned odd = (unsigned)key & 1u, th16 = smx_thresh24(p.drop_p) >> 8;
                    const float inv = 1.0f / (1.0f - p.drop_p);
                    const int ra = q0 + t * 16 + 4 * g + 2 * (int)odd;
                    const unsigned kk = (unsigned)key >> 1;
                    const unsigned own0 = smx_hash32(p.drop_seed, (att_row_base(p, b, h, ra) >> 1) + kk);
                    const unsigned own1 = smx_hash32(p.drop_seed, (att_row_base(p, b, h, ra + 1) >> 1) + kk);
                    const unsigned oth0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own0, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                    const unsigned oth1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own1, 0xB1, 0xf, 0xf, true);
                    const unsigned h4[4] = {odd ? oth0 : own0, odd ? oth1 : own1, odd ? own0 : oth0, odd ? own1 : oth1};
#pragma unroll
                    for (int r = 0; r < 4; ++r) dm4[r] = (odd ? h4[r] >> 16 : h4[r] & 0xffffu) >= th16 ? inv : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q0 + t * 16 + 4 * g + r;
                    float off = nl[r];
                    if (p.bias && qq < p.Tq && key < p.Tk) off = fmaf(p.bias[((long long)h * p.Tq + qq) * p.Tk + key], SMX_LOG2E, off);
                    float pr = fast_exp2(fmaf(sc[r], sl2, off));
                    if (masked && (qq >= p.Tq || key >= p.Tk || (p.causal && key > qq + coff))) pr = 0.f;
                    const float dm = dm4[r];
                    pt[u][t][r] = pr * dm;
                    ds[u][t][r] = pr * (dm * dp[r] - dl[r]);
                }
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {       // two 32-query reduction steps
            bf16x8_t dfr[4], qfr[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dfr[dt] = frag_tr(tDO, st * 32, st * 32 + 16, dt * 16, lane);
                qfr[dt] = frag_tr(tQ, st * 32, st * 32 + 16, dt * 16, lane);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bf16x8_t pf = pack_pair(pt[u][2 * st], pt[u][2 * st + 1]);
                const bf16x8_t df = pack_pair(ds[u][2 * st], ds[u][2 * st + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[dt], pf, dv[u][dt], 0, 0, 0);
                    dk[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[dt], df, dk[u][dt], 0, 0, 0);
                }
            }
        }
        if (more) {
            tile_store(sQ[buf ^ 1], rq, tid);
            tile_store(sDO[buf ^ 1], rd, tid);
            if (tid < 64) {
                sNlse2[buf ^ 1][tid] = rl;
                sDelta[buf ^ 1][tid] = rdl;
            }
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int key = kw0 + u * 16 + i16;
        if (key < p.Tk) {
            bf16_t* dKp = reinterpret_cast<bf16_t*>(p.dK) + b * p.dk_bs + (long long)key * p.dk_ld + h * 64;
            bf16_t* dVp = reinterpret_cast<bf16_t*>(p.dV) + b * p.dv_bs + (long long)key * p.dv_ld + h * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<uint2*>(dKp + dt * 16 + 4 * g) =
                    make_uint2(pack_bf2(dk[u][dt][0] * p.scale, dk[u][dt][1] * p.scale),
                               pack_bf2(dk[u][dt][2] * p.scale, dk[u][dt][3] * p.scale));
                *reinterpret_cast<uint2*>(dVp + dt * 16 + 4 * g) =
                    make_uint2(pack_bf2(dv[u][dt][0], dv[u][dt][1]), pack_bf2(dv[u][dt][2], dv[u][dt][3]));
            }
        }
    }
}

#include "attention_v2.h"
#include "attention_v3.h"

static bool attn_use_v1() {
    static const bool v1 = getenv("SMX_ATTN_V1") && getenv("SMX_ATTN_V1")[0] == '1';      // A/B switch: first-generation kernels
    return v1;
}

// Words the two bit masks of a dropout call need (mask_q, mask_k); both 0 without dropout on the MFMA path.
extern "C" int smx_attn_mask_words(const SmxAttnParams* p, int dtype, long long* nq, long long* nk) {
    *nq = *nk = 0;
    if (dtype != SMX_BF16 || p->D != 64 || !(p->drop_p > 0.f) || attn_use_v1()) return SMX_OK;
    *nq = (long long)p->B * p->H * p->Tq * A2_QW(p->Tk);
    *nk = (long long)p->B * p->H * p->Tk * A2_QW(p->Tq);
    return SMX_OK;
}

// Fill p->mask_q / p->mask_k for (drop_p, drop_seed): call once per attention call, before forward; backward reuses them.
extern "C" int smx_attn_dropout_mask(const SmxAttnParams* pp, hipStream_t stream) {
    (void)hipGetLastError();
    SmxAttnParams p = *pp;
    if (!p.mask_q || !p.mask_k || !(p.drop_p > 0.f) || p.B <= 0 || p.H <= 0 || p.Tq <= 0 || p.Tk <= 0) return SMX_EINVAL;
    const int KW = A2_QW(p.Tk);
    hipLaunchKernelGGL(attn_mask_kernel, dim3((KW + 3) / 4, (p.Tq + 63) / 64, p.B * p.H), dim3(256), 0, stream, p);
    SMX_CHECK_LAUNCH();
}

static int attn_check(const SmxAttnParams& p, int dtype) {
    if (p.klen && dtype == SMX_BF16 && p.D == 64 && attn_use_v1()) return SMX_EINVAL;      // key lengths: v2 / simple kernels only
    if (p.B <= 0 || p.H <= 0 || p.Tq <= 0 || p.Tk <= 0) return SMX_EINVAL;
    if ((dtype == SMX_F32 || p.D != 64) && (p.D > SIMPLE_MAXD || (p.D & 7))) return SMX_EINVAL;
    if (p.causal && p.Tk < p.Tq) return SMX_EINVAL;
    return SMX_OK;
}

extern "C" int smx_attention_fwd(const SmxAttnParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxAttnParams p = *pp;
    int rc = attn_check(p, dtype);
    if (rc) return rc;
    if (dtype == SMX_F32) {
        const int n = p.B * p.H * p.Tq;
        hipLaunchKernelGGL(attn_fwd_simple<float>, dim3((n + 63) / 64), dim3(64), 0, stream, p);
    } else if (dtype == SMX_BF16 && p.D != 64) {   // head widths off the hot path (tiny test configs)
        const int n = p.B * p.H * p.Tq;
        hipLaunchKernelGGL(attn_fwd_simple<bf16_t>, dim3((n + 63) / 64), dim3(64), 0, stream, p);
    } else if (dtype == SMX_BF16 && !attn_use_v1()) {
        if (p.drop_p > 0.f && (!p.mask_q || !p.mask_k)) return SMX_EINVAL;       // smx_attn_dropout_mask first
        if (attn_v3_mode() == 2 && p.Tk <= 64 * A3_MAXT && p.Tq >= 128) {              // K / V of a head resident in LDS (attention_v3.h)
            const int nt = (p.Tk + 63) / 64;
            A3_DISPATCH(attn3_fwd, 8, p.Tq, nt, 0);          // (sixteen waves at 128 registers spill 84 - 200 bytes in the tile bodies)
            SMX_CHECK_LAUNCH();
        }
        const dim3 grid(((p.Tq + 63) / 64) * p.H * p.B);
        A2_DISPATCH(attn2_fwd, grid);
    } else if (dtype == SMX_BF16) {
        // U = 1 (16 queries per wave): measured faster than U = 2 at T = 499 (occupancy 4 vs 2 waves/SIMD; the
        // kernels are VALU-bound on the softmax, not LDS-bound)
        hipLaunchKernelGGL(attn_fwd_bf16<1>, dim3((p.Tq + 63) / 64, p.H, p.B), dim3(256), 0, stream, p);
    } else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

extern "C" int smx_attention_bwd(const SmxAttnParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxAttnParams p = *pp;
    int rc = attn_check(p, dtype);
    if (rc) return rc;
    if (!p.delta || !p.dO || !p.dQ || !p.dK || !p.dV) return SMX_EINVAL;
    const int n = p.B * p.H * p.Tq;
    if (dtype == SMX_F32) {
        hipLaunchKernelGGL(attn_delta_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(attn_bwd_dq_simple<float>, dim3((n + 63) / 64), dim3(64), 0, stream, p);
        const int nk = p.B * p.H * p.Tk;
        hipLaunchKernelGGL(attn_bwd_dkv_simple<float>, dim3((nk + 63) / 64), dim3(64), 0, stream, p);
        if (p.dbias) {
            const long long nb = (long long)p.H * p.Tq * p.Tk;
            hipLaunchKernelGGL(attn_dbias_kernel<float>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, p);
        }
    } else if (dtype == SMX_BF16 && p.D != 64) {
        hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(attn_bwd_dq_simple<bf16_t>, dim3((n + 63) / 64), dim3(64), 0, stream, p);
        const int nk = p.B * p.H * p.Tk;
        hipLaunchKernelGGL(attn_bwd_dkv_simple<bf16_t>, dim3((nk + 63) / 64), dim3(64), 0, stream, p);
        if (p.dbias) {
            const long long nb = (long long)p.H * p.Tq * p.Tk;
            hipLaunchKernelGGL(attn_dbias_kernel<bf16_t>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, p);
        }
    } else if (dtype == SMX_BF16) {
        if (!p.O) return SMX_EINVAL;
        // dQ also writes delta = rowsum(dO * O); the dK/dV kernel reads it (stream order)
        if (!attn_use_v1()) {
            if (p.drop_p > 0.f && (!p.mask_q || !p.mask_k)) return SMX_EINVAL;
            const dim3 gq(((p.Tq + 63) / 64) * p.H * p.B), gk(((p.Tk + 63) / 64) * p.H * p.B);
            // resident-operand forms (attention_v3.h): dQ with the head's K / V in LDS, dK/dV with its Q / dO (+ lse / delta rows)
            if ((attn_v3_mode() == 2 ? p.Tq >= 128 : attn_v3_mode() == 1 && p.Tq >= 384) && p.Tk <= 64 * A3_MAXT) A3_DISPATCH(attn3_dq, 8, p.Tq, (p.Tk + 63) / 64, 0);
            else A2_DISPATCH(attn2_dq, gq);
            if ((attn_v3_mode() == 2 ? p.Tk >= 128 : attn_v3_mode() == 1 && p.Tk >= 384) && p.Tq <= 64 * A3_MAXT) A3_DISPATCH(attn3_dkv, 8, p.Tk, (p.Tq + 63) / 64, (size_t)((p.Tq + 63) / 64) * 512);
            else A2_DISPATCH(attn2_dkv, gk);
        } else {
            hipLaunchKernelGGL(attn_bwd_dq_bf16<1>, dim3((p.Tq + 63) / 64, p.H, p.B), dim3(256), 0, stream, p);
            hipLaunchKernelGGL(attn_bwd_dkv_bf16<1>, dim3((p.Tk + 63) / 64, p.H, p.B), dim3(256), 0, stream, p);
        }
        if (p.dbias) {
            const long long nb = (long long)p.H * p.Tq * p.Tk;
            hipLaunchKernelGGL(attn_dbias_kernel<bf16_t>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, p);
        }
    } else return SMX_EINVAL;
    SMX_CHECK_LAUNCH();
}

// T5 relative-position bias: scatter-add the [H, Tq, Tk] score-bias gradient into the [buckets, H] table
// (TF:models/t5/modeling_t5.py:261-279: values = relative_attention_bias(bucket) -> permute).  bucket: int32 [Tq*Tk].
extern "C" int smx_attn_bias_scatter(const float* dbias, const int* bucket, float* dtable, int H, int Tq, int Tk, int nbuckets,
                                     hipStream_t stream) {
    (void)hipGetLastError();
    if (!dbias || !bucket || !dtable || H <= 0 || Tq <= 0 || Tk <= 0 || nbuckets <= 0) return SMX_EINVAL;
    hipLaunchKernelGGL(attn_bias_scatter_kernel, dim3(nbuckets, H), dim3(256), 0, stream, dbias, bucket, dtable, H, Tq * Tk);
    SMX_CHECK_LAUNCH();
}

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxAttnParams(void) { return (int)sizeof(SmxAttnParams); }

SMX_STEP_KEY_TU(attention)
