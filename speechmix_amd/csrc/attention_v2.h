// Second-generation MFMA attention kernels (head_dim 64, bf16) - included by attention.hip.
//
// Same tiling, LDS images and "transposed score" trick as the first generation (block = 4 waves x 16 queries or keys,
// 64-row K/V (Q/dO) tiles double-buffered through registers), rebuilt around what the ISA of the first generation showed:
// the kernels are bound by VALU issue, and most of what they issued was not softmax arithmetic -
//   * every feature (T5 bias, causal mask, ragged tails, dropout) was a RUN-TIME test inside the per-score loops:
//     ~28 VALU + ~12 SALU instructions per score where the arithmetic needs ~5.  Here bias / causal / dropout are
//     template parameters and the ragged / diagonal tiles take a separate MASKED instantiation of the tile body;
//   * attention-probability dropout hashed every score in all three kernels (two quarter-rate 32-bit multiplies per pair
//     of scores).  The mask is now generated ONCE per attention call as a bit matrix in both orientations
//     (smx_attn_dropout_mask: query-major words for the forward / dQ kernels, key-major words for the dK/dV kernel, the
//     transposition through wave ballots) - bit-identical to the hash-derived mask of every other dropout site - and
//     a score costs one v_bfe_i32 + one v_and_b32; the 1/(1-p) scale is folded into the output scales;
//   * the running row sum is kept per lane and reduced across the four lane groups once, after the last tile.
#pragma once
#include <utility>

#define A2_QW(T) ((((T) + 63) >> 6) << 1)          // 32-bit mask words per row: whole 64-wide tiles

// ---------------------------------------------------------------- dropout bit masks
// wave = 64 queries x one 32-key word; keep(q, k) <=> hash field of index rowbase(q) + k >= threshold - exactly
// smx_drop_mul's decision (rowbase % 4 == 0: the index parity is the key's).
template <int LANE>
__device__ __forceinline__ void a2_writelane(unsigned& v, unsigned s) {       // v[LANE] = s (wave-uniform s)
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(LANE));
}
// keys kw * 32 + 2 J, + 1 of my query: one hash, two compares; a compare's result IS the 64-query bit column of its key
template <int J>
__device__ __forceinline__ void a2_mask_pair(const SmxAttnParams& p, int kw, unsigned rb, unsigned th, bool qok, unsigned& w,
                                             unsigned& mine_lo, unsigned& mine_hi) {
    const int key = kw * 32 + 2 * J;
    const unsigned hsh = smx_hash32(p.drop_seed, (rb + (unsigned)key) >> 1);
    const bool k0 = qok && key < p.Tk && (hsh & 0xffffu) >= th;
    const bool k1 = qok && key + 1 < p.Tk && (hsh >> 16) >= th;
    if (k0) w |= 1u << (2 * J);
    if (k1) w |= 2u << (2 * J);
    const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);
    a2_writelane<2 * J>(mine_lo, (unsigned)b0);
    a2_writelane<2 * J>(mine_hi, (unsigned)(b0 >> 32));
    a2_writelane<2 * J + 1>(mine_lo, (unsigned)b1);
    a2_writelane<2 * J + 1>(mine_hi, (unsigned)(b1 >> 32));
}
template <int... Js>
__device__ __forceinline__ void a2_mask_pairs(std::integer_sequence<int, Js...>, const SmxAttnParams& p, int kw, unsigned rb, unsigned th,
                                              bool qok, unsigned& w, unsigned& mine_lo, unsigned& mine_hi) {
    (a2_mask_pair<Js>(p, kw, rb, th, qok, w, mine_lo, mine_hi), ...);
}
__global__ __launch_bounds__(256) void attn_mask_kernel(SmxAttnParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KW = A2_QW(p.Tk), QW = A2_QW(p.Tq);
    const int kw = blockIdx.x * 4 + wave;                 // key word
    const int qb = blockIdx.y;                            // 64-query block
    const int bh = blockIdx.z;
    if (kw >= KW) return;
    const int q = qb * 64 + lane;
    const unsigned th = smx_thresh24(p.drop_p) >> 8;
    const bool qok = q < p.Tq;
    const unsigned rb = (unsigned)(((long long)bh * p.Tq + (qok ? q : 0)) * (long long)((p.Tk + 3) & ~3));
    // The key-major words come straight from the compares' lane masks (two v_writelane_b32 per key into lane `key`); the round-2
    // form re-derived every column from the finished query-major word: 32 x (bit extract, compare, lane compare, two selects).
    unsigned w = 0, mine_lo = 0, mine_hi = 0;
    a2_mask_pairs(std::make_integer_sequence<int, 16>{}, p, kw, rb, th, qok, w, mine_lo, mine_hi);
    if (qok) p.mask_q[((long long)bh * p.Tq + q) * KW + kw] = w;
    const int key = kw * 32 + lane;
    if (lane < 32 && key < p.Tk) {
        unsigned* dst = p.mask_k + ((long long)bh * p.Tk + key) * QW + 2 * qb;
        dst[0] = mine_lo;
        dst[1] = mine_hi;
    }
}

// Block -> (batch, head, row block).  Workgroups go to the 8 XCDs round-robin in dispatch order, so with a (row block,
// head, batch) grid whose x extent is 8 (T = 499) the 8 row blocks of one head - which all read the head's whole K and V -
// landed on 8 DIFFERENT L2s: every XCD pulled every head's K/V through the fabric (PMC: 204 MB fetched per forward
// launch for 98 MB of operands; the launch ran at the speed of that traffic, not of its MFMAs or its softmax).  Here
// the grid is one-dimensional and XCD x owns the heads bh = x (mod 8): a head's row blocks share one L2.
__device__ __forceinline__ void a2_decode(const SmxAttnParams& p, int nx, int& b, int& h, int& xb) {
    const int nbh = p.B * p.H;
    int L = blockIdx.x, bh;
    if ((nbh & 7) == 0) {
        const int xcd = L & 7, idx = L >> 3;
        const int j = idx / nx;
        xb = idx - j * nx;
        bh = j * 8 + xcd;
    } else {
        bh = L / nx;
        xb = L - bh * nx;
    }
    b = bh / p.H;
    h = bh - b * p.H;
}

// max over the four 16-lane rows (same lane & 15) without the LDS crossbar: v_permlane16_swap / v_permlane32_swap exchange
// row pairs / wave halves at VALU rate (the ds_bpermute pair of __shfl_xor sat on the per-tile critical path)
// fmaxf() on values hipcc cannot prove canonical (MFMA results, permlane results) costs TWO instructions each - a self-max that
// quiets signalling NaNs, then the max: 28 v_max_f32 per 64-key tile for the row maximum of 16 scores.  Scores are never NaN
// here (masked ones are -inf), so the maxima go through the instructions directly: 8 x v_max3_f32 + 3 x v_max_f32.
__device__ __forceinline__ float a2_max(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float a2_max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float a2_group_max(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float a = a2_max(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const unsigned ua = __float_as_uint(a);
    auto r2 = __builtin_amdgcn_permlane32_swap(ua, ua, false, false);
    return a2_max(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}

// all-ones / all-zeros lane masks of the 4 scores a lane holds in 16-key block t of a 64-wide tile
// (bit position inside the tile's two words: 16 t + 4 g + r)
__device__ __forceinline__ void a2_bits4(const uint2& w, int t, int g4, unsigned (&m)[4]) {
    const unsigned wt = (t & 2 ? w.y : w.x) >> (16 * (t & 1));
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = (unsigned)__builtin_amdgcn_sbfe((int)wt, g4 + r, 1);
}
__device__ __forceinline__ float a2_and(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) & m); }

// SMX_ATTN_TRACE (lab builds, tools/gpu_attn_trace.py): shader-clock stamps at the phase boundaries of the forward tile body, summed per wave
// over its tiles and written by lane 0 of every wave of workgroups 0..63: [wg][wave][phase] cycles + tile count
#ifndef SMX_ATTN_TRACE
#define SMX_ATTN_TRACE 0
#endif
#if SMX_ATTN_TRACE
__device__ unsigned long long smx_attn_trace_buf[64 * 16 * 8];
extern "C" int smx_attn_trace_read(void* host, unsigned long long bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(smx_attn_trace_buf), bytes) == hipSuccess ? SMX_OK : -5;
}
struct A2Trace {
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    __device__ __forceinline__ void start() { t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t = __builtin_readcyclecounter(); ph[i] += t - t0; t0 = t; __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void flush(int wave, int ntiles) {
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 64 && wave < 16) {
            unsigned long long* d = smx_attn_trace_buf + (blockIdx.x * 16 + wave) * 8;
            for (int i = 0; i < 7; ++i) d[i] = ph[i];
            d[7] = (unsigned long long)ntiles;
        }
    }
};
#define A2_TR_START(tr) (tr).start()
#define A2_TR_MARK(tr, i) (tr).mark(i)
#else
struct A2Trace { __device__ __forceinline__ void flush(int, int) {} };
#define A2_TR_START(tr) do { } while (0)
#define A2_TR_MARK(tr, i) do { } while (0)
#endif

#ifndef A2_FWD_OCC          // waves per SIMD the forward kernel is compiled for
#define A2_FWD_OCC(BIAS, DROP) (((BIAS) || (DROP)) ? 2 : 3)          // (round 6: the dropout variants at 3 fit 168 registers with 0 / 12 bytes of scratch and run no faster: 65.5 vs 65.7 us)
#endif
// ---------------------------------------------------------------- forward
template <bool MASKED, bool BIAS, bool CAUSAL, bool DROP>
__device__ __forceinline__ void a2_fwd_tile(const SmxAttnParams& p, const char* tK, const char* tV, const bf16x8_t (&qf)[2],
                                            f32x4_t (&o)[4], float& m, float& l, int k0, int q, int h, int lane, float sl2,
                                            int coff, const uint2& mw, int tk, A2Trace* tr = nullptr) {
    const int g = lane >> 4;
    f32x4_t s[4];
    A2_TR_MARK(*tr, 6);
    // All fragment reads of the tile are ISSUED before their first consumer: read-then-wait per MFMA made one tile a chain of
    // ~24 LDS latencies (2 900 cycles per tile for a lone wave with the global traffic and the barrier compiled out).
    // K fragments first, the score MFMAs behind them; the V fragments right after, in flight under the softmax arithmetic.
    bf16x8_t kcur[2] = {frag_kc(tK, 0, 0, lane), frag_kc(tK, 0, 1, lane)}, knxt[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {              // fragment reads one 16-key block ahead of the MFMAs that consume them
        if (t < 3) { knxt[0] = frag_kc(tK, (t + 1) * 16, 0, lane); knxt[1] = frag_kc(tK, (t + 1) * 16, 1, lane); }
        __builtin_amdgcn_sched_barrier(0);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcur[0], qf[0], ZERO4, 0, 0, 0);
        s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcur[1], qf[1], s[t], 0, 0, 0);
        kcur[0] = knxt[0]; kcur[1] = knxt[1];
    }
    A2_TR_MARK(*tr, 0);
    bf16x8_t vfr0[4], vfr1[4];                 // V fragments of the first reduction step: in flight under the softmax arithmetic
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vfr0[dt] = frag_tr(tV, 0, 16, dt * 16, lane);
    __builtin_amdgcn_sched_barrier(0);
    A2_TR_MARK(*tr, 1);
    float mul = sl2;
    if constexpr (BIAS) {                       // T5 relative-position bias: scores leave this block in log2 units
        mul = 1.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + t * 16 + 4 * g + r;
                float v = s[t][r] * sl2;
                if (q < p.Tq && key < p.Tk) v = fmaf(p.bias[((long long)h * p.Tq + q) * p.Tk + key], SMX_LOG2E, v);
                s[t][r] = v;
            }
    }
    if constexpr (MASKED) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + t * 16 + 4 * g + r;
                if (key >= tk || (CAUSAL && key > q + coff)) s[t][r] = -INFINITY;
            }
    }
    float mx = a2_max3(s[0][0], s[0][1], s[0][2]);
    mx = a2_max3(mx, s[0][3], s[1][0]);
    mx = a2_max3(mx, s[1][1], s[1][2]);
    mx = a2_max3(mx, s[1][3], s[2][0]);
    mx = a2_max3(mx, s[2][1], s[2][2]);
    mx = a2_max3(mx, s[2][3], s[3][0]);
    mx = a2_max3(mx, s[3][1], s[3][2]);
    mx = a2_max(mx, s[3][3]);
    mx = a2_group_max(mx);
    const float mn = a2_max(m, mx * mul);
    const float alpha = fast_exp2(m - mn);
    float rs = 0.f;
#if SMX_ATTN_TRACE
    asm volatile("" : "+v"(rs) : "v"(alpha));
#endif
    A2_TR_MARK(*tr, 2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        unsigned bm[4];
        if constexpr (DROP) a2_bits4(mw, t, 4 * g, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = fast_exp2(fmaf(s[t][r], mul, -mn));
            rs += e;
            s[t][r] = DROP ? a2_and(e, bm[r]) : e;
        }
    }
    l = fmaf(l, alpha, rs);                      // per-lane partial: reduced across the lane groups after the last tile
    m = mn;
#if SMX_ATTN_TRACE
    asm volatile("" : "+v"(l));
#endif
    A2_TR_MARK(*tr, 3);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vfr1[dt] = frag_tr(tV, 32, 48, dt * 16, lane);       // second step's, under the first's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    A2_TR_MARK(*tr, 4);
    {                                            // two 32-key reduction steps
        const bf16x8_t pf0 = pack_pair(s[0], s[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr0[dt], pf0, o[dt], 0, 0, 0);
        const bf16x8_t pf1 = pack_pair(s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr1[dt], pf1, o[dt], 0, 0, 0);
    }
    A2_TR_MARK(*tr, 5);
}

template <bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, A2_FWD_OCC(BIAS, DROP)) void attn2_fwd(SmxAttnParams p) {
    __shared__ __attribute__((aligned(16))) char sK[2][8192];
    __shared__ __attribute__((aligned(16))) char sV[2][8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    int b, h, xb;
    a2_decode(p, (p.Tq + 63) >> 6, b, h, xb);
    const int qb0 = xb * 64, q = qb0 + wave * 16 + i16;
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    bf16x8_t qf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qf[kk] = load_row_frag(Qp, p.q_ld, q, p.Tq, kk, g);
    f32x4_t o[4] = {ZERO4, ZERO4, ZERO4, ZERO4};
    float m = NEG_BIG, l = 0.f;
    const float sl2 = p.scale * SMX_LOG2E;
    const int coff = p.Tk - p.Tq;
    const int tk = p.klen ? min(p.klen[b], p.Tk) : p.Tk;   // per-clip key length (right-padded attention mask): keys >= tk are masked
    int kend = tk;
    if (CAUSAL) kend = min(tk, qb0 + 64 + coff);          // keys beyond the block's last query are masked
    const uint2* mrow = nullptr;
    uint2 mw = make_uint2(0, 0), mnext = make_uint2(0, 0);
    if constexpr (DROP) {
        mrow = reinterpret_cast<const uint2*>(p.mask_q + (((long long)b * p.H + h) * p.Tq + min(q, p.Tq - 1)) * A2_QW(p.Tk));
        mnext = mrow[0];
    }
    uint4 rk[2], rv[2];
    tile_load(rk, Kp, p.k_ld, 0, p.Tk, tid);
    tile_load(rv, Vp, p.v_ld, 0, p.Tk, tid);
    tile_store(sK[0], rk, tid);
    tile_store(sV[0], rv, tid);
    __syncthreads();
    int buf = 0;
    A2Trace trc;
    int ntl = 0;
    A2_TR_START(trc);
    const bool lab_noload = !DROP && p.drop_seed == 0xdead0001u;      // LAB: timing without the tile traffic (wrong results)
    const bool lab_nosync = !DROP && p.drop_seed == 0xdead0002u;      // LAB: ... and without the per-tile barrier
    for (int k0 = 0; k0 < kend; k0 += 64) {
        const bool more = k0 + 64 < kend && !lab_noload && !lab_nosync;
        if constexpr (DROP) mw = mnext;
        if (more) {
            tile_load(rk, Kp, p.k_ld, k0 + 64, p.Tk, tid);
            tile_load(rv, Vp, p.v_ld, k0 + 64, p.Tk, tid);
            if constexpr (DROP) mnext = mrow[(k0 >> 6) + 1];
        }
        // masks only where they can bite: the ragged last key tile, and tiles crossing the causal diagonal
        const bool masked = (k0 + 64 > tk) || (CAUSAL && k0 + 63 > qb0 + coff);
        if (masked) a2_fwd_tile<true, BIAS, CAUSAL, DROP>(p, sK[buf], sV[buf], qf, o, m, l, k0, q, h, lane, sl2, coff, mw, tk, &trc);
        else a2_fwd_tile<false, BIAS, CAUSAL, DROP>(p, sK[buf], sV[buf], qf, o, m, l, k0, q, h, lane, sl2, coff, mw, tk, &trc);
        ++ntl;
        if (more) {
            tile_store(sK[buf ^ 1], rk, tid);
            tile_store(sV[buf ^ 1], rv, tid);
        }
        if (!lab_nosync) __syncthreads();       // next tile published; everyone is done with this one before it is overwritten next round
        if (!lab_noload && !lab_nosync) buf ^= 1;
    }
    trc.flush(wave, ntl);
    l = group_sum(l);
    if (q < p.Tq) {
        const float inv = (DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f) / l;
        bf16_t* Op = reinterpret_cast<bf16_t*>(p.O) + b * p.o_bs + (long long)q * p.o_ld + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 pk = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
            *reinterpret_cast<uint2*>(Op + dt * 16 + 4 * g) = pk;
        }
        if (g == 0) p.lse[((long long)b * p.H + h) * p.Tq + q] = m * SMX_LN2 + __logf(l);   // natural-log units
    }
}

// ---------------------------------------------------------------- dQ (+ delta = rowsum(dO * O))
template <bool MASKED, bool BIAS, bool CAUSAL, bool DROP>
__device__ __forceinline__ void a2_dq_tile(const SmxAttnParams& p, const char* tK, const char* tV, const bf16x8_t (&qf)[2],
                                           const bf16x8_t (&dof)[2], f32x4_t (&dq)[4], float nlse2, float delta, int k0, int q,
                                           int h, int lane, float sl2, int coff, const uint2& mw, float inv_keep, int tk) {
    const int g = lane >> 4;
    f32x4_t ds[4];
    // fragment reads one 16-key block ahead of their MFMAs (see a2_fwd_tile)
    bf16x8_t fcur[4] = {frag_kc(tK, 0, 0, lane), frag_kc(tK, 0, 1, lane), frag_kc(tV, 0, 0, lane), frag_kc(tV, 0, 1, lane)}, fnxt[4];
    f32x4_t sca[4], dpa[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < 3) {
            fnxt[0] = frag_kc(tK, (t + 1) * 16, 0, lane); fnxt[1] = frag_kc(tK, (t + 1) * 16, 1, lane);
            fnxt[2] = frag_kc(tV, (t + 1) * 16, 0, lane); fnxt[3] = frag_kc(tV, (t + 1) * 16, 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        sca[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[0], qf[0], ZERO4, 0, 0, 0);
        sca[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[1], qf[1], sca[t], 0, 0, 0);
        dpa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[2], dof[0], ZERO4, 0, 0, 0);
        dpa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[3], dof[1], dpa[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fcur[i] = fnxt[i];
    }
    bf16x8_t ktr0[4], ktr1[4];                // transposed K fragments of the dQ product: in flight under the arithmetic
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ktr0[dt] = frag_tr(tK, 0, 16, dt * 16, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const f32x4_t sc = sca[t], dp = dpa[t];
        unsigned bm[4];
        if constexpr (DROP) a2_bits4(mw, t, 4 * g, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = k0 + t * 16 + 4 * g + r;
            float off = nlse2;
            if constexpr (BIAS) {
                if (q < p.Tq && key < p.Tk) off = fmaf(p.bias[((long long)h * p.Tq + q) * p.Tk + key], SMX_LOG2E, off);
            }
            float pr = fast_exp2(fmaf(sc[r], sl2, off));
            if constexpr (MASKED) {
                if (key >= tk || q >= p.Tq || (CAUSAL && key > q + coff)) pr = 0.f;
            }
            const float gp = DROP ? fmaf(a2_and(dp[r], bm[r]), inv_keep, -delta) : dp[r] - delta;
            ds[t][r] = pr * gp;
        }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ktr1[dt] = frag_tr(tK, 32, 48, dt * 16, lane);
    __builtin_amdgcn_sched_barrier(0);
    {
        const bf16x8_t pf0 = pack_pair(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr0[dt], pf0, dq[dt], 0, 0, 0);
        const bf16x8_t pf1 = pack_pair(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr1[dt], pf1, dq[dt], 0, 0, 0);
    }
}

template <bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, (BIAS || DROP) ? 2 : 3) void attn2_dq(SmxAttnParams p) {
    __shared__ __attribute__((aligned(16))) char sK[2][8192];
    __shared__ __attribute__((aligned(16))) char sV[2][8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    int b, h, xb;
    a2_decode(p, (p.Tq + 63) >> 6, b, h, xb);
    const int qb0 = xb * 64, q = qb0 + wave * 16 + i16;
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* dOp = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * 64;
    const bf16_t* Op = reinterpret_cast<const bf16_t*>(p.O) + b * p.o_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    bf16x8_t qf[2], dof[2];
    float dsum = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        qf[kk] = load_row_frag(Qp, p.q_ld, q, p.Tq, kk, g);
        dof[kk] = load_row_frag(dOp, p.do_ld, q, p.Tq, kk, g);
        if (q < p.Tq) {
            float ov[8], dv[8];
            load8(Op + (long long)q * p.o_ld + kk * 32 + g * 8, ov);
            load8(dOp + (long long)q * p.do_ld + kk * 32 + g * 8, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) dsum = fmaf(ov[e], dv[e], dsum);
        }
    }
    const float delta = group_sum(dsum);
    float nlse2 = 0.f;
    if (q < p.Tq) {
        const long long li = ((long long)b * p.H + h) * p.Tq + q;
        nlse2 = -p.lse[li] * SMX_LOG2E;
        if (g == 0) p.delta[li] = delta;
    }
    f32x4_t dq[4] = {ZERO4, ZERO4, ZERO4, ZERO4};
    const float sl2 = p.scale * SMX_LOG2E;
    const float inv_keep = DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const int coff = p.Tk - p.Tq;
    const int tk = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    int kend = tk;
    if (CAUSAL) kend = min(tk, qb0 + 64 + coff);
    const uint2* mrow = nullptr;
    uint2 mw = make_uint2(0, 0), mnext = make_uint2(0, 0);
    if constexpr (DROP) {
        mrow = reinterpret_cast<const uint2*>(p.mask_q + (((long long)b * p.H + h) * p.Tq + min(q, p.Tq - 1)) * A2_QW(p.Tk));
        mnext = mrow[0];
    }
    uint4 rk[2], rv[2];
    tile_load(rk, Kp, p.k_ld, 0, p.Tk, tid);
    tile_load(rv, Vp, p.v_ld, 0, p.Tk, tid);
    tile_store(sK[0], rk, tid);
    tile_store(sV[0], rv, tid);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < kend; k0 += 64) {
        const bool more = k0 + 64 < kend;
        if constexpr (DROP) mw = mnext;
        if (more) {
            tile_load(rk, Kp, p.k_ld, k0 + 64, p.Tk, tid);
            tile_load(rv, Vp, p.v_ld, k0 + 64, p.Tk, tid);
            if constexpr (DROP) mnext = mrow[(k0 >> 6) + 1];
        }
        const bool masked = (k0 + 64 > tk) || (qb0 + 64 > p.Tq) || (CAUSAL && k0 + 63 > qb0 + coff);
        if (masked) a2_dq_tile<true, BIAS, CAUSAL, DROP>(p, sK[buf], sV[buf], qf, dof, dq, nlse2, delta, k0, q, h, lane, sl2, coff, mw, inv_keep, tk);
        else a2_dq_tile<false, BIAS, CAUSAL, DROP>(p, sK[buf], sV[buf], qf, dof, dq, nlse2, delta, k0, q, h, lane, sl2, coff, mw, inv_keep, tk);
        if (more) {
            tile_store(sK[buf ^ 1], rk, tid);
            tile_store(sV[buf ^ 1], rv, tid);
        }
        __syncthreads();
        buf ^= 1;
    }
    if (q < p.Tq) {
        bf16_t* dQp = reinterpret_cast<bf16_t*>(p.dQ) + b * p.dq_bs + (long long)q * p.dq_ld + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 pk = make_uint2(pack_bf2(dq[dt][0] * p.scale, dq[dt][1] * p.scale),
                                  pack_bf2(dq[dt][2] * p.scale, dq[dt][3] * p.scale));
            *reinterpret_cast<uint2*>(dQp + dt * 16 + 4 * g) = pk;
        }
    }
}

// ---------------------------------------------------------------- dK / dV
// block owns 64 keys (wave: 16), loops over 64-query tiles; scores UN-transposed (S = Q K^T: lane owns one key column)
template <bool MASKED, bool BIAS, bool CAUSAL, bool DROP>
__device__ __forceinline__ void a2_dkv_tile(const SmxAttnParams& p, const char* tQ, const char* tDO, const float* sNl,
                                            const float* sDl, const bf16x8_t (&kf)[2], const bf16x8_t (&vf)[2],
                                            f32x4_t (&dk)[4], f32x4_t (&dv)[4], int q0, int key, int h, int lane, float sl2,
                                            int coff, const uint2& mw, float inv_keep, int tk) {
    const int g = lane >> 4;
    f32x4_t pt[4], ds[4];
    // fragment reads one 16-query block ahead of their MFMAs (see a2_fwd_tile)
    bf16x8_t fcur[4] = {frag_kc(tQ, 0, 0, lane), frag_kc(tQ, 0, 1, lane), frag_kc(tDO, 0, 0, lane), frag_kc(tDO, 0, 1, lane)}, fnxt[4];
    f32x4_t sca[4], dpa[4];
    float4 nlv[4], dlv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {          // 16-query sub-tiles: D rows = queries 4g+r, cols = keys
        if (t < 3) {
            fnxt[0] = frag_kc(tQ, (t + 1) * 16, 0, lane); fnxt[1] = frag_kc(tQ, (t + 1) * 16, 1, lane);
            fnxt[2] = frag_kc(tDO, (t + 1) * 16, 0, lane); fnxt[3] = frag_kc(tDO, (t + 1) * 16, 1, lane);
        }
        nlv[t] = *reinterpret_cast<const float4*>(sNl + t * 16 + 4 * g);
        dlv[t] = *reinterpret_cast<const float4*>(sDl + t * 16 + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
        sca[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[0], kf[0], ZERO4, 0, 0, 0);
        sca[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[1], kf[1], sca[t], 0, 0, 0);
        dpa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[2], vf[0], ZERO4, 0, 0, 0);
        dpa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[3], vf[1], dpa[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) fcur[i] = fnxt[i];
    }
    bf16x8_t dtr[4], qtr[4];                  // transposed dO / Q fragments of the first reduction step, in flight under the arithmetic
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        dtr[dt] = frag_tr(tDO, 0, 16, dt * 16, lane);
        qtr[dt] = frag_tr(tQ, 0, 16, dt * 16, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float nla[4] = {nlv[t].x, nlv[t].y, nlv[t].z, nlv[t].w}, dla[4] = {dlv[t].x, dlv[t].y, dlv[t].z, dlv[t].w};
        const f32x4_t sc = sca[t], dp = dpa[t];
        unsigned bm[4];
        if constexpr (DROP) a2_bits4(mw, t, 4 * g, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q0 + t * 16 + 4 * g + r;
            float off = nla[r];
            if constexpr (BIAS) {
                if (qq < p.Tq && key < p.Tk) off = fmaf(p.bias[((long long)h * p.Tq + qq) * p.Tk + key], SMX_LOG2E, off);
            }
            float pr = fast_exp2(fmaf(sc[r], sl2, off));
            if constexpr (MASKED) {
                if (qq >= p.Tq || key >= tk || (CAUSAL && key > qq + coff)) pr = 0.f;
            }
            if constexpr (DROP) {
                pt[t][r] = a2_and(pr, bm[r]);                              // (x 1/(1-p): folded into dV's output scale)
                ds[t][r] = pr * fmaf(a2_and(dp[r], bm[r]), inv_keep, -dla[r]);
            } else {
                pt[t][r] = pr;
                ds[t][r] = pr * (dp[r] - dla[r]);
            }
        }
    }
    {                                      // two 32-query reduction steps
        const bf16x8_t pf0 = pack_pair(pt[0], pt[1]);
        const bf16x8_t df0 = pack_pair(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtr[dt], pf0, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], df0, dk[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dtr[dt] = frag_tr(tDO, 32, 48, dt * 16, lane);
            qtr[dt] = frag_tr(tQ, 32, 48, dt * 16, lane);
        }
        const bf16x8_t pf1 = pack_pair(pt[2], pt[3]);
        const bf16x8_t df1 = pack_pair(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtr[dt], pf1, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], df1, dk[dt], 0, 0, 0);
        }
    }
}

template <bool BIAS, bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn2_dkv(SmxAttnParams p) {
    __shared__ __attribute__((aligned(16))) char sQ[2][8192];
    __shared__ __attribute__((aligned(16))) char sDO[2][8192];
    __shared__ __attribute__((aligned(16))) float sNlse2[2][64];
    __shared__ __attribute__((aligned(16))) float sDelta[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    int b, h, xb;
    a2_decode(p, (p.Tk + 63) >> 6, b, h, xb);
    const int kb0 = xb * 64, key = kb0 + wave * 16 + i16;
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* dOp = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    const long long rowbase = ((long long)b * p.H + h) * p.Tq;
    bf16x8_t kf[2], vf[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        kf[kk] = load_row_frag(Kp, p.k_ld, key, p.Tk, kk, g);
        vf[kk] = load_row_frag(Vp, p.v_ld, key, p.Tk, kk, g);
    }
    f32x4_t dk[4] = {ZERO4, ZERO4, ZERO4, ZERO4}, dv[4] = {ZERO4, ZERO4, ZERO4, ZERO4};
    const float sl2 = p.scale * SMX_LOG2E;
    const float inv_keep = DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const int coff = p.Tk - p.Tq;
    const int tk = p.klen ? min(p.klen[b], p.Tk) : p.Tk;   // keys >= tk are padding: their probabilities are 0 -> dK = dV = 0
    int qbeg = 0;
    if (CAUSAL) qbeg = max(0, kb0 - coff) & ~63;          // queries before this see none of the block's keys
    const uint2* mrow = nullptr;
    uint2 mw = make_uint2(0, 0), mnext = make_uint2(0, 0);
    if constexpr (DROP) {
        mrow = reinterpret_cast<const uint2*>(p.mask_k + (((long long)b * p.H + h) * p.Tk + min(key, p.Tk - 1)) * A2_QW(p.Tq));
        mnext = mrow[qbeg >> 6];
    }
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
        if constexpr (DROP) mw = mnext;
        if (more) {
            tile_load(rq, Qp, p.q_ld, q0 + 64, p.Tq, tid);
            tile_load(rd, dOp, p.do_ld, q0 + 64, p.Tq, tid);
            if (tid < 64) {
                const int qq = q0 + 64 + tid;
                rl = qq < p.Tq ? -p.lse[rowbase + qq] * SMX_LOG2E : 0.f;
                rdl = qq < p.Tq ? p.delta[rowbase + qq] : 0.f;
            }
            if constexpr (DROP) mnext = mrow[(q0 >> 6) + 1];
        }
        const bool masked = (q0 + 64 > p.Tq) || (kb0 + 64 > tk) || (CAUSAL && kb0 + 63 > q0 + coff);
        if (masked) a2_dkv_tile<true, BIAS, CAUSAL, DROP>(p, sQ[buf], sDO[buf], sNlse2[buf], sDelta[buf], kf, vf, dk, dv, q0, key, h, lane, sl2, coff, mw, inv_keep, tk);
        else a2_dkv_tile<false, BIAS, CAUSAL, DROP>(p, sQ[buf], sDO[buf], sNlse2[buf], sDelta[buf], kf, vf, dk, dv, q0, key, h, lane, sl2, coff, mw, inv_keep, tk);
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
    if (key < p.Tk) {
        bf16_t* dKp = reinterpret_cast<bf16_t*>(p.dK) + b * p.dk_bs + (long long)key * p.dk_ld + h * 64;
        bf16_t* dVp = reinterpret_cast<bf16_t*>(p.dV) + b * p.dv_bs + (long long)key * p.dv_ld + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *reinterpret_cast<uint2*>(dKp + dt * 16 + 4 * g) =
                make_uint2(pack_bf2(dk[dt][0] * p.scale, dk[dt][1] * p.scale), pack_bf2(dk[dt][2] * p.scale, dk[dt][3] * p.scale));
            *reinterpret_cast<uint2*>(dVp + dt * 16 + 4 * g) =
                make_uint2(pack_bf2(dv[dt][0] * inv_keep, dv[dt][1] * inv_keep), pack_bf2(dv[dt][2] * inv_keep, dv[dt][3] * inv_keep));
        }
    }
}

// runtime -> compile-time dispatch over (bias, causal, dropout)
#define A2_DISPATCH(KERNEL, GRID)                                                                                   \
    do {                                                                                                            \
        const int v = (p.bias ? 4 : 0) | (p.causal ? 2 : 0) | (p.drop_p > 0.f ? 1 : 0);                              \
        switch (v) {                                                                                                \
            case 0: hipLaunchKernelGGL((KERNEL<false, false, false>), GRID, dim3(256), 0, stream, p); break;        \
            case 1: hipLaunchKernelGGL((KERNEL<false, false, true>), GRID, dim3(256), 0, stream, p); break;         \
            case 2: hipLaunchKernelGGL((KERNEL<false, true, false>), GRID, dim3(256), 0, stream, p); break;         \
            case 3: hipLaunchKernelGGL((KERNEL<false, true, true>), GRID, dim3(256), 0, stream, p); break;          \
            case 4: hipLaunchKernelGGL((KERNEL<true, false, false>), GRID, dim3(256), 0, stream, p); break;         \
            case 5: hipLaunchKernelGGL((KERNEL<true, false, true>), GRID, dim3(256), 0, stream, p); break;          \
            case 6: hipLaunchKernelGGL((KERNEL<true, true, false>), GRID, dim3(256), 0, stream, p); break;          \
            default: hipLaunchKernelGGL((KERNEL<true, true, true>), GRID, dim3(256), 0, stream, p); break;          \
        }                                                                                                           \
    } while (0)
