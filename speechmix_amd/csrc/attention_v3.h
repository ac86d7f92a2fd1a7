// Third-generation attention kernels (round 6): the second generation's tile bodies (attention_v2.h: a2_fwd_tile / a2_dq_tile /
// a2_dkv_tile - transposed scores, bit-mask dropout, masked / unmasked instantiations) under a different outer structure.
//
// Why: at T = 499 (config 2's speech encoder, TF:models/wav2vec2/modeling_wav2vec2.py:466-548) the second generation runs a
// 64-query workgroup over 8 key tiles, each staged global -> registers -> LDS behind a workgroup barrier: 98 304 wave-tiles per forward
// launch at ~1 250 cycles each on a SIMD, twice what their ~130 vector instructions need (profiles/r05: forward 0.13, backward 0.10
// of the MFMA peak - the time is the per-tile staging / barrier chain, not the arithmetic).  A head's K and V are only
// 2 x 64 KB at T <= 512, so here ONE workgroup keeps the whole head's operands RESIDENT in LDS:
//   * K / V (forward, dQ) or Q / dO (+ the log-sum-exp and delta rows; dK/dV) go global -> LDS once, by LDS-DMA in tile order;
//   * every wave then walks the key (query) tiles of its own 16 queries (keys) with NO barrier and no staging traffic - only in the
//     workgroup's first pass does tile t wait (counted vmcnt + one barrier) for its fills, which were issued up front;
//   * a workgroup covers 16 NW queries per pass and NP passes (chunk = 16 NW NP queries): 768 workgroups = 3 rounds at T = 499.
// Taken when the resident operands fit (T of the resident side <= 512) and the other side has at least 128 rows; everything else
// (the decoder's T = 32, 20-s clips) stays on the second generation.  Results are bit-identical to it: same tile bodies, same
// tile order per query (tests/test_gpu_r6.py).
#pragma once

#define A3_MAXT 8
// first pass: the workgroup meets once per A3_SYNC_EVERY tiles' fills (a barrier per tile cost ~900 cycles per tile: the two waves of a SIMD
// alternate, and every meeting waits for the partner's whole tile - tools/gpu_attn_trace.py)
#ifndef A3_SYNC_EVERY
#define A3_SYNC_EVERY 8
#endif
typedef __attribute__((ext_vector_type(4))) int a3_rsrc_t;
__device__ __forceinline__ a3_rsrc_t a3_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    a3_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));      // stride 0: raw buffer
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                    // offsets beyond it read zeros
    r[3] = 0x00020000;
    return r;
}
// 16 B per lane: LDS[lds_wave_base + lane * 16] = mem[rsrc.base + voff]  (zeros when voff + 16 > num_records)
__device__ __forceinline__ void a3_dma16(a3_rsrc_t rsrc, unsigned voff, unsigned lds_wave_base) {
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, 0 offen lds"
                 :: "s"(rsrc), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory");
}
// Fills of the two resident operands, tile by tile: tile t of A (B) = rows 64 t .. of a [rows, ld] matrix in the t_addr image at
// sA (sB) + 8192 t.  A tile is 8 pieces of 1 KB (8 rows); NW == 16: wave w issues piece w & 7 of A (w < 8) or B, one instruction per
// tile; NW == 8: wave w issues piece w of both.  Returns the number of instructions this wave issued per tile.
template <int NW>
__device__ __forceinline__ void a3_fill(unsigned sA, unsigned sB, const a3_rsrc_t& ra, const a3_rsrc_t& rb, long long lda, long long ldb,
                                        int nt, int wave, int lane) {
    const int j = wave & 7;
    const int r8 = j * 8 + (lane >> 3);                       // row inside the tile
    const int c = (lane & 7) ^ ((r8 >> 1) & 7);               // logical 16-B chunk that lives in my slot of the row
    for (int t = 0; t < nt; ++t) {
        const int row = t * 64 + r8;
        if (NW != 16 || wave < 8) a3_dma16(ra, (unsigned)(row * lda * 2 + c * 16), sA + t * 8192 + j * 1024);
        if (NW != 16 || wave >= 8) a3_dma16(rb, (unsigned)(row * ldb * 2 + c * 16), sB + t * 8192 + j * 1024);
    }
}
// wait until at most `left` of this wave's fills are outstanding (younger compiler-issued loads only make it stricter), then meet
__device__ __forceinline__ void a3_wait_tile(int left) {
    switch (left) {
#define A3_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        A3_W(0) A3_W(1) A3_W(2) A3_W(3) A3_W(4) A3_W(5) A3_W(6) A3_W(7) A3_W(8) A3_W(9) A3_W(10) A3_W(11) A3_W(12) A3_W(13) A3_W(14)
#undef A3_W
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Dropout mask words of a row's (up to) 8 tiles without a vector-memory wait inside the tile loop: memory operations retire in order, so a
// mask load issued behind the fills would hold its consumer until EVERY fill has landed.  The four lane groups of a row (same lane & 15)
// need the same words: group g loads the words of tiles g and g + 4 BEFORE the fills go out, tile t takes them from group t & 3 through
// the LDS crossbar.
struct A3Mask {
    uint2 lo, hi;
    __device__ __forceinline__ void load(const uint2* row, int nt, int g) {
        lo = g < nt ? row[g] : make_uint2(0u, 0u);
        hi = g + 4 < nt ? row[g + 4] : make_uint2(0u, 0u);
    }
    __device__ __forceinline__ uint2 tile(int t, int lane) const {
        const int src = ((lane & 15) | ((t & 3) << 4)) << 2;
        const uint2 w = t < 4 ? lo : hi;
        return make_uint2((unsigned)__builtin_amdgcn_ds_bpermute(src, (int)w.x), (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)w.y));
    }
};

// Block -> (batch, head, chunk): XCD x owns the heads bh = x (mod 8), as a2_decode
__device__ __forceinline__ void a3_decode(const SmxAttnParams& p, int nx, int& b, int& h, int& xb) { a2_decode(p, nx, b, h, xb); }

// ---------------------------------------------------------------- forward
template <bool BIAS, bool CAUSAL, bool DROP, int NW>
__global__ __launch_bounds__(NW * 64) void attn3_fwd(SmxAttnParams p, int np) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int nt = (p.Tk + 63) >> 6;
    char* const sK = smem;
    char* const sV = smem + nt * 8192;
    const int chunk = NW * 16 * np;
    int b, h, xb;
    a3_decode(p, (p.Tq + chunk - 1) / chunk, b, h, xb);
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    const float sl2 = p.scale * SMX_LOG2E;
    const int coff = p.Tk - p.Tq;
    const int tk = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    constexpr int IC = NW == 16 ? 1 : 2;           // fills per tile and wave
    A2Trace trc;
    int ntl = 0;
    for (int ps = 0; ps < np; ++ps) {
        const int q0w = xb * chunk + (ps * NW + wave) * 16;          // my wave's first query
        const int q = q0w + i16;
        bf16x8_t qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = load_row_frag(Qp, p.q_ld, q, p.Tq, kk, g);
        f32x4_t o[4] = {ZERO4, ZERO4, ZERO4, ZERO4};
        float m = NEG_BIG, l = 0.f;
        int kend = q0w < p.Tq ? tk : 0;             // (a wave past the last query still meets the first pass's barriers)
        if (CAUSAL) kend = min(kend, q0w + 16 + coff);
        A3Mask mk;
        uint2 mw = make_uint2(0, 0);
        if constexpr (DROP)
            mk.load(reinterpret_cast<const uint2*>(p.mask_q + (((long long)b * p.H + h) * p.Tq + min(q, p.Tq - 1)) * A2_QW(p.Tk)), nt, g);
        // (every load of this pass is consumed here, ahead of the fills: a load the compiler still tracks would be waited for behind them)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(qf[kk]));
        if constexpr (DROP) asm volatile("" : "+v"(mk.lo.x), "+v"(mk.lo.y), "+v"(mk.hi.x), "+v"(mk.hi.y));
        if (ps == 0) {          // the fills go out BEHIND this pass's own loads (in-order retirement: see A3Mask)
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
            const a3_rsrc_t rk = a3_make_rsrc(Kp, (unsigned)(((long long)(p.Tk - 1) * p.k_ld + 64) * 2));
            const a3_rsrc_t rv = a3_make_rsrc(Vp, (unsigned)(((long long)(p.Tk - 1) * p.v_ld + 64) * 2));
            a3_fill<NW>(lds0, lds0 + nt * 8192, rk, rv, p.k_ld, p.v_ld, nt, wave, lane);
        }
        const int tend = ps == 0 ? nt : (kend + 63) >> 6;
        if (ps == 0) A2_TR_START(trc);
        for (int t = 0; t < tend; ++t) {
            const int k0 = t * 64;
            if (ps == 0 && (t & (A3_SYNC_EVERY - 1)) == 0) a3_wait_tile(max(nt - A3_SYNC_EVERY - t, 0) * IC);
            if (k0 >= kend) continue;
            if constexpr (DROP) mw = mk.tile(t, lane);
            const bool masked = (k0 + 64 > tk) || (CAUSAL && k0 + 63 > q0w + coff);
            if (masked) a2_fwd_tile<true, BIAS, CAUSAL, DROP>(p, sK + t * 8192, sV + t * 8192, qf, o, m, l, k0, q, h, lane, sl2, coff, mw, tk, &trc);
            else a2_fwd_tile<false, BIAS, CAUSAL, DROP>(p, sK + t * 8192, sV + t * 8192, qf, o, m, l, k0, q, h, lane, sl2, coff, mw, tk, &trc);
            ++ntl;
        }
        if (ps == np - 1) trc.flush(wave, ntl);
        l = group_sum(l);
        if (q < p.Tq) {
            const float inv = (DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f) / l;
            bf16_t* Op = reinterpret_cast<bf16_t*>(p.O) + b * p.o_bs + (long long)q * p.o_ld + h * 64;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                uint2 pk = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
                *reinterpret_cast<uint2*>(Op + dt * 16 + 4 * g) = pk;
            }
            if (g == 0) p.lse[((long long)b * p.H + h) * p.Tq + q] = m * SMX_LN2 + __logf(l);
        }
    }
}

// ---------------------------------------------------------------- dQ (+ delta = rowsum(dO * O))
template <bool BIAS, bool CAUSAL, bool DROP, int NW>
__global__ __launch_bounds__(NW * 64) void attn3_dq(SmxAttnParams p, int np) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int nt = (p.Tk + 63) >> 6;
    char* const sK = smem;
    char* const sV = smem + nt * 8192;
    const int chunk = NW * 16 * np;
    int b, h, xb;
    a3_decode(p, (p.Tq + chunk - 1) / chunk, b, h, xb);
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* dOp = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * 64;
    const bf16_t* Op = reinterpret_cast<const bf16_t*>(p.O) + b * p.o_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    const float sl2 = p.scale * SMX_LOG2E;
    const float inv_keep = DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const int coff = p.Tk - p.Tq;
    const int tk = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    constexpr int IC = NW == 16 ? 1 : 2;
    for (int ps = 0; ps < np; ++ps) {
        const int q0w = xb * chunk + (ps * NW + wave) * 16;
        const int q = q0w + i16;
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
        float delta = group_sum(dsum);
        float nlse2 = 0.f;
        if (q < p.Tq) {
            const long long li = ((long long)b * p.H + h) * p.Tq + q;
            nlse2 = -p.lse[li] * SMX_LOG2E;
            if (g == 0) p.delta[li] = delta;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(dof[kk]));
        asm volatile("" : "+v"(nlse2), "+v"(delta));
        f32x4_t dq[4] = {ZERO4, ZERO4, ZERO4, ZERO4};
        int kend = q0w < p.Tq ? tk : 0;
        if (CAUSAL) kend = min(kend, q0w + 16 + coff);
        A3Mask mk;
        uint2 mw = make_uint2(0, 0);
        if constexpr (DROP)
            mk.load(reinterpret_cast<const uint2*>(p.mask_q + (((long long)b * p.H + h) * p.Tq + min(q, p.Tq - 1)) * A2_QW(p.Tk)), nt, g);
        // (every load of this pass is consumed here, ahead of the fills: a load the compiler still tracks would be waited for behind them)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(qf[kk]));
        if constexpr (DROP) asm volatile("" : "+v"(mk.lo.x), "+v"(mk.lo.y), "+v"(mk.hi.x), "+v"(mk.hi.y));
        if (ps == 0) {          // the fills go out BEHIND this pass's own loads (in-order retirement: see A3Mask)
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
            const a3_rsrc_t rk = a3_make_rsrc(Kp, (unsigned)(((long long)(p.Tk - 1) * p.k_ld + 64) * 2));
            const a3_rsrc_t rv = a3_make_rsrc(Vp, (unsigned)(((long long)(p.Tk - 1) * p.v_ld + 64) * 2));
            a3_fill<NW>(lds0, lds0 + nt * 8192, rk, rv, p.k_ld, p.v_ld, nt, wave, lane);
        }
        const int tend = ps == 0 ? nt : (kend + 63) >> 6;
        for (int t = 0; t < tend; ++t) {
            const int k0 = t * 64;
            if (ps == 0 && (t & (A3_SYNC_EVERY - 1)) == 0) a3_wait_tile(max(nt - A3_SYNC_EVERY - t, 0) * IC);
            if (k0 >= kend) continue;
            if constexpr (DROP) mw = mk.tile(t, lane);
            const bool masked = (k0 + 64 > tk) || (q0w + 16 > p.Tq) || (CAUSAL && k0 + 63 > q0w + coff);
            if (masked) a2_dq_tile<true, BIAS, CAUSAL, DROP>(p, sK + t * 8192, sV + t * 8192, qf, dof, dq, nlse2, delta, k0, q, h, lane, sl2, coff, mw, inv_keep, tk);
            else a2_dq_tile<false, BIAS, CAUSAL, DROP>(p, sK + t * 8192, sV + t * 8192, qf, dof, dq, nlse2, delta, k0, q, h, lane, sl2, coff, mw, inv_keep, tk);
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
}

// ---------------------------------------------------------------- dK / dV: Q, dO and the query rows' -lse / delta resident
template <bool BIAS, bool CAUSAL, bool DROP, int NW>
__global__ __launch_bounds__(NW * 64) void attn3_dkv(SmxAttnParams p, int np) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int nt = (p.Tq + 63) >> 6;
    char* const sQ = smem;
    char* const sDO = smem + nt * 8192;
    float* const sNl = reinterpret_cast<float*>(smem + 2 * nt * 8192);
    float* const sDl = sNl + nt * 64;
    const int chunk = NW * 16 * np;
    int b, h, xb;
    a3_decode(p, (p.Tk + chunk - 1) / chunk, b, h, xb);
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(p.Q) + b * p.q_bs + h * 64;
    const bf16_t* dOp = reinterpret_cast<const bf16_t*>(p.dO) + b * p.do_bs + h * 64;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.K) + b * p.k_bs + h * 64;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(p.V) + b * p.v_bs + h * 64;
    const long long rowbase = ((long long)b * p.H + h) * p.Tq;
    const float sl2 = p.scale * SMX_LOG2E;
    const float inv_keep = DROP ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    const int coff = p.Tk - p.Tq;
    const int tk = p.klen ? min(p.klen[b], p.Tk) : p.Tk;
    for (int i = tid; i < nt * 64; i += NW * 64) {          // (ahead of the fills: older in the memory queue, complete at the first tile's wait)
        sNl[i] = i < p.Tq ? -p.lse[rowbase + i] * SMX_LOG2E : 0.f;
        sDl[i] = i < p.Tq ? p.delta[rowbase + i] : 0.f;
    }
    constexpr int IC = NW == 16 ? 1 : 2;
    for (int ps = 0; ps < np; ++ps) {
        const int kb0w = xb * chunk + (ps * NW + wave) * 16;          // my wave's first key
        const int key = kb0w + i16;
        bf16x8_t kf[2], vf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kf[kk] = load_row_frag(Kp, p.k_ld, key, p.Tk, kk, g);
            vf[kk] = load_row_frag(Vp, p.v_ld, key, p.Tk, kk, g);
        }
        f32x4_t dk[4] = {ZERO4, ZERO4, ZERO4, ZERO4}, dv[4] = {ZERO4, ZERO4, ZERO4, ZERO4};
        const bool live = kb0w < p.Tk;
        int tbeg = 0;
        if (CAUSAL) tbeg = max(0, kb0w - coff) >> 6;          // query tiles before this see none of my keys
        A3Mask mk;
        uint2 mw = make_uint2(0, 0);
        if constexpr (DROP)
            mk.load(reinterpret_cast<const uint2*>(p.mask_k + (((long long)b * p.H + h) * p.Tk + min(key, p.Tk - 1)) * A2_QW(p.Tq)), nt, g);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) asm volatile("" : "+v"(kf[kk]), "+v"(vf[kk]));
        if constexpr (DROP) asm volatile("" : "+v"(mk.lo.x), "+v"(mk.lo.y), "+v"(mk.hi.x), "+v"(mk.hi.y));
        if (ps == 0) {
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
            const a3_rsrc_t rq = a3_make_rsrc(Qp, (unsigned)(((long long)(p.Tq - 1) * p.q_ld + 64) * 2));
            const a3_rsrc_t rd = a3_make_rsrc(dOp, (unsigned)(((long long)(p.Tq - 1) * p.do_ld + 64) * 2));
            a3_fill<NW>(lds0, lds0 + nt * 8192, rq, rd, p.q_ld, p.do_ld, nt, wave, lane);
        }
        for (int t = ps == 0 ? 0 : tbeg; t < nt; ++t) {
            const int q0 = t * 64;
            if (ps == 0 && (t & (A3_SYNC_EVERY - 1)) == 0) a3_wait_tile(max(nt - A3_SYNC_EVERY - t, 0) * IC);
            if (t < tbeg || !live) continue;
            if constexpr (DROP) mw = mk.tile(t, lane);
            const bool masked = (q0 + 64 > p.Tq) || (kb0w + 16 > tk) || (CAUSAL && kb0w + 15 > q0 + coff);
            if (masked) a2_dkv_tile<true, BIAS, CAUSAL, DROP>(p, sQ + t * 8192, sDO + t * 8192, sNl + q0, sDl + q0, kf, vf, dk, dv, q0, key, h, lane, sl2, coff, mw, inv_keep, tk);
            else a2_dkv_tile<false, BIAS, CAUSAL, DROP>(p, sQ + t * 8192, sDO + t * 8192, sNl + q0, sDl + q0, kf, vf, dk, dv, q0, key, h, lane, sl2, coff, mw, inv_keep, tk);
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
}

// Resident-operand kernels apply when the resident side fits 8 tiles and the walking side is long enough for two passes of a workgroup.
// Measured (tools/gpu_attn_bench.py, B = 32, H = 12, T = 499; profiles/r06_attention_v3.txt): backward 195 -> 188 us, 224 -> 206 us with
// dropout; forward 61 -> 67 us (67 -> 75 with dropout), and both slower at T = 249 - the tile bodies, not the staging, set the pace.
// So by default only the BACKWARD kernels of long sequences take this form.  SMX_ATTN_V3 = bwd (default) | 0 (off) | 1 (forward too).
static int attn_v3_mode() {          // 0 off, 1 backward only, 2 forward + backward   (read per call: the parity tests flip it inside one process)
    const char* e = getenv("SMX_ATTN_V3");
    if (!e) return 1;
    return e[0] == '0' ? 0 : e[0] == '1' ? 2 : 1;
}
static int a3_passes(int rows, int nw) { return rows > 16 * nw ? 2 : 1; }          // chunk = 16 NW NP rows of the walking side
template <class K>
static void a3_launch(K kernel, const SmxAttnParams& p, int rows, int nw, int resident_tiles, size_t extra, hipStream_t stream) {
    const int np = a3_passes(rows, nw), chunk = 16 * nw * np;
    const size_t lds = (size_t)2 * resident_tiles * 8192 + extra;
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid(((rows + chunk - 1) / chunk) * p.H * p.B);
    hipLaunchKernelGGL(kernel, grid, dim3(nw * 64), lds, stream, p, np);
}
#define A3_DISPATCH(KERNEL, NW, ROWS, TILES, EXTRA)                                                                          \
    do {                                                                                                                    \
        const int v = (p.bias ? 4 : 0) | (p.causal ? 2 : 0) | (p.drop_p > 0.f ? 1 : 0);                                      \
        switch (v) {                                                                                                        \
            case 0: a3_launch(KERNEL<false, false, false, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                   \
            case 1: a3_launch(KERNEL<false, false, true, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                    \
            case 2: a3_launch(KERNEL<false, true, false, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                    \
            case 3: a3_launch(KERNEL<false, true, true, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                     \
            case 4: a3_launch(KERNEL<true, false, false, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                    \
            case 5: a3_launch(KERNEL<true, false, true, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                     \
            case 6: a3_launch(KERNEL<true, true, false, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                     \
            default: a3_launch(KERNEL<true, true, true, NW>, p, ROWS, NW, TILES, EXTRA, stream); break;                     \
        }                                                                                                                   \
    } while (0)
