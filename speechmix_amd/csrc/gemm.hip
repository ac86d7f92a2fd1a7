// GEMM family for the SpeechMix hot path (gfx950).
//
//   C[m,n] (+)= epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// Each operand is either K-contiguous ("KC": logical [rows, K] row-major, rows addressed through a
// SmxRowView so conv windows need no im2col) or rows-contiguous ("RC": logical [K, rows], i.e. the
// reduction index is the slow one - dgrad's W[N,K] and both wgrad operands).  One kernel template
// therefore covers fwd (KC,KC), dgrad (KC,RC) and wgrad (RC,RC) of every Linear / Conv1d on the path
// (ref:speechmix/model.py:92-102, 148; TF:models/wav2vec2/modeling_wav2vec2.py:254-323, 466-572;
//  TF:models/bart/modeling_bart.py:260-390).
//
// bf16 kernel: 128x128x64 tile, 4 waves (2x2, 64x64 each), v_mfma_f32_16x16x32_bf16 with the two
// operands swapped so that each lane ends up owning 4 consecutive n of one m (8-byte epilogue
// accesses).  KC tiles are XOR-swizzled 128-B rows read with ds_read_b128; RC tiles are [k][128]
// images read with the gfx950 transposing read ds_read_b64_tr_b16.  Register-staged double-buffered
// LDS, one barrier per K step.  fp32 kernel: a deliberately simple, independent VALU tile kernel
// (parity path + on-device cross-check of the MFMA kernel).
#include <cstdlib>
#include "gemm_common.h"

template <typename TOUT>
__device__ __forceinline__ void epilogue4(const SmxGemmParams& p, long long zc, long long zbias, long long ze, int m,
                                          int n0, float v[4]) {
    // lane owns C[m, n0..n0+3]
    if (m >= p.M || n0 >= p.N) return;
    const long long base = zc + view_off(p.c, m) + n0;
    const long long sb = (p.resid || p.aux_out || p.aux_in) ? ze + view_off(p.e, m) + n0 : 0;
    const int nv = min(4, p.N - n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nv) {
            float x = v[j] * p.alpha;
            if (p.bias) x += p.bias[zbias + n0 + j];
            if (p.aux_out) reinterpret_cast<bf16_t*>(p.aux_out)[sb + j] = f2bf(x);
            if (!p.aux_in) x = act_fwd(x, p.act);
            else x *= act_grad(bf2f(reinterpret_cast<const bf16_t*>(p.aux_in)[sb + j]), p.act);
            if (p.drop_p > 0.f)
                x *= smx_drop_mul(p.drop_seed, (unsigned)((long long)m * p.N + n0 + j + zc), smx_thresh24(p.drop_p),
                                  1.0f / (1.0f - p.drop_p));
            if (p.resid) x += bf2f(reinterpret_cast<const bf16_t*>(p.resid)[sb + j]);
            v[j] = x;
        }
    }
    if (p.out_f32) {
        float* c = reinterpret_cast<float*>(p.C) + base;
        if (p.atomic == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) atomicAdd(c + j, v[j]);
        } else if (p.atomic == 2) {
            if (nv == 4 && ((base & 3) == 0)) {
                float4 o = *reinterpret_cast<float4*>(c);
                o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                *reinterpret_cast<float4*>(c) = o;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) c[j] += v[j];
            }
        } else if (nv == 4 && ((base & 3) == 0)) {
            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) c[j] = v[j];
        }
    } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + base;
        if (nv == 4 && ((base & 3) == 0)) {
            uint2 pk = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            *reinterpret_cast<uint2*>(c) = pk;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) c[j] = f2bf(v[j]);
        }
    }
}

template <bool A_RC, bool B_RC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(SmxGemmParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective remap of the linear tile id: blocks b, b+8, b+16.. (same XCD, same L2) get
    // consecutive tiles, which share the A row-panel.
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    // column-block rasterisation: consecutive tile ids sweep GROUP_N n-tiles of one m-panel, then the next
    // m-panel, inside one block of n-tiles.  With the XCD remap above, each XCD's L2 then holds a 1.5-MB block of
    // B while streaming A panels once (instead of cycling all of B through a 4-MB L2 for every m-panel).
    int tm, tn;
    {
        const int per_group = GROUP_N * ntm;
        const int grp = wg / per_group, rem = wg - grp * per_group;
        const int first = grp * GROUP_N;
        const int gsz = min(ntn - first, GROUP_N);
        tm = rem / gsz;
        tn = first + (rem - tm * gsz);
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int z = blockIdx.z;
    const int zb = z / p.split_k, zs = z - zb * p.split_k;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long long)zb * p.batch_a;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (long long)zb * p.batch_b;
    const long long zc = (long long)zb * p.batch_c + (long long)zs * p.split_stride, zbias = (long long)zb * p.batch_bias, ze = (long long)zb * p.batch_e;

    // K range of this split (multiple of BK)
    const int ksteps_total = (p.K + BK - 1) / BK;
    const int per = (ksteps_total + p.split_k - 1) / p.split_k;
    const int ks0 = zs * per, ks1 = min(ksteps_total, ks0 + per);
    if (ks0 >= ks1 && p.split_k > 1 && p.atomic == 1) return;

    TileLoader<A_RC> la;
    TileLoader<B_RC> lb;
    la.init(p.a, m0, p.M, tid);
    lb.init(p.b, n0, p.N, tid);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};


    la.load(A, p.a, m0, p.M, ks0 * BK, p.K, tid);
    lb.load(B, p.b, n0, p.N, ks0 * BK, p.K, tid);
    la.store(smem, tid);
    lb.store(smem + 16384, tid);
    __syncthreads();

    int cur = 0;
    for (int ks = ks0; ks < ks1; ++ks) {
        const bool more = ks + 1 < ks1;
        if (more) {
            la.load(A, p.a, m0, p.M, (ks + 1) * BK, p.K, tid);
            lb.load(B, p.b, n0, p.N, (ks + 1) * BK, p.K, tid);
        }
        const char* tA = smem + cur * STAGE_BYTES;
        const char* tB = tA + 16384;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = load_frag<A_RC>(tA, wm * 64 + i * 16, kk, lane, p.tr_mode);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_RC>(tB, wn * 64 + j * 16, kk, lane, p.tr_mode);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // swapped operands: D[row = n (4g+reg)][col = m (lane&15)]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            la.store(smem + (cur ^ 1) * STAGE_BYTES, tid);
            lb.store(smem + (cur ^ 1) * STAGE_BYTES + 16384, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    const int g = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue4<bf16_t>(p, zc, zbias, ze, m0 + wm * 64 + i * 16 + i16, n0 + wn * 64 + j * 16 + 4 * g, v);
        }
}


// ------------------------------------------------------------------------------------------------
// bf16, direct-to-LDS variant (the production path): the same 128x128x64 tiling and LDS images, but tiles are
// filled with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass).  An LDS-DMA writes
// wave-uniform-base + lane*16, so the LDS image stays lane-linear and the XOR swizzles are applied to the
// per-lane SOURCE address instead (same involution on the read side).  Out-of-range rows / k read a 64-B
// zero page through the per-lane source pointer.  One 32-KB LDS buffer and <=128 registers per lane:
// 4 workgroups per CU hide each other's fill latency (two barriers per K step).
// ------------------------------------------------------------------------------------------------
template <bool RC, int NPASS = 4>
struct DmaLoader {
    static constexpr int NP = RC ? 4 : NPASS;      // KC: passes of 32 rows (4: 128-row tile, 2: 64-row tile)
    // Minimal per-thread state (this kernel lives on a 128-register budget): KC keeps one 32-bit element offset per
    // pass (-1: row out of range -> zero page); RC recomputes its source address from k every step.
    int off[NP];
    int kc;                  // KC: my k offset inside a step (source chunk * 8)
    const bf16_t* base;
    const bf16_t* zero;

    __device__ __forceinline__ void init(const bf16_t* b, const SmxRowView& v, int row0, int nrows, int k0, int tid) {
        base = b;
        zero = reinterpret_cast<const bf16_t*>(smx_zero_page);
        const int lane = tid & 63, wave = tid >> 6;
        kc = 0;
        if (!RC) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int rl = p * 32 + wave * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((rl >> 1) & 7);
                if (p == 0) kc = c * 8;       // identical for all p: (rl>>1)&7 only depends on wave*8+(lane>>3)
                off[p] = row0 + rl < nrows ? (int)(view_off(v, row0 + rl) + c * 8) : -1;
            }
        }
    }
    // ASM: issue through inline asm so that hipcc does not drain the DMA queue (s_waitcnt vmcnt(0)) before the next
    // LDS read - the pipelined kernels count their own waits
    // rcp0 / RCNP: the 16-k-row passes of a rows-contiguous tile this caller issues (the eight-wave kernel gives each half of
    // the workgroup two of the four; rcp0 is wave-uniform)
    template <bool ASM = false, int RCNP = 4>
    __device__ __forceinline__ void issue(char* tile, const SmxRowView& v, int row0, int nrows, int k0, int K, int tid, int rcp0 = 0) {
        const int lane = tid & 63, wave = tid >> 6;
        if (!RC) {
            const bool kin = k0 + kc < K;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const bf16_t* src = (kin && off[p] >= 0) ? base + off[p] + k0 : zero;
                if (ASM) glds16_asm(src, tile + (p * 32 + wave * 8) * 128);
                else glds16(src, tile + (p * 32 + wave * 8) * 128);
            }
        } else {
#pragma unroll
            for (int pp = 0; pp < RCNP; ++pp) {
                const int p = rcp0 + pp;
                int kl = p * 16 + wave * 4 + (lane >> 4);
                // (eight-wave kernel, 128 registers with a second loader beside this one: keep the per-pass row products out
                // of the registers that live across the K loop - they are one 64-bit multiply-add to recompute)
                if constexpr (RCNP != 4) asm volatile("" : "+v"(kl));
                const int g16 = lane & 15;
                const int c = row0 + ((((g16 >> 1) ^ rc_swz(kl)) << 1) | (g16 & 1)) * 8;
                const bf16_t* src = zero;
                if (k0 + kl < K && c < nrows) src = base + view_off(v, k0 + kl) + c;
                if (ASM) glds16_asm(src, tile + (p * 16 + wave * 4) * 256);
                else glds16(src, tile + (p * 16 + wave * 4) * 256);
            }
        }
    }
};

// acc[i][j]: 16x16 block (rows mw0 + 16 i .., cols nw0 + 16 j ..) of this wave's 64x64 sub-tile; wbuf: 8 KB of LDS
// owned by this wave (no other wave touches it between the caller's barriers).
template <int NH = 2>       // NH: 32-row halves of the wave's sub-tile (2: 64 x 64, 1: 32 x 64)
__device__ __forceinline__ void epilogue_staged(const SmxGemmParams& p, f32x4_t (&acc)[2 * NH][4], char* wbuf, int mw0, int nw0,
                                                long long zc, long long zbias, long long ze, int lane) {
    const int i16 = lane & 15, g = lane >> 4;
    const int rr = lane >> 3, cc = lane & 7;
    const int n = nw0 + cc * 8;
    float bs[8];
    if (p.bias && n + 8 <= p.N && !((zbias + n) & 3)) {
        load8(p.bias + zbias + n, bs);                 // two 16-B loads, in flight while the first half is transposed
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = (p.bias && n + e < p.N) ? p.bias[zbias + n + e] : 0.f;
    }
    const unsigned th = smx_thresh24(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    const unsigned dseed = smx_dseed(p.drop_p, p.drop_seed);          // + the step key (the kernels hand in the kernarg copy of p)
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lr = i2 * 16 + i16, c = j * 4 + g;
                *reinterpret_cast<f32x4_t*>(wbuf + lr * 256 + ((c ^ (lr & 15)) << 4)) = acc[2 * h + i2][j];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lr = q * 8 + rr;
            const int m = mw0 + h * 32 + lr;
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(wbuf + lr * 256 + (((2 * cc) ^ (lr & 15)) << 4));
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(wbuf + lr * 256 + (((2 * cc + 1) ^ (lr & 15)) << 4));
            if (m < p.M && n < p.N) {
                float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                epilogue_row8(p, zc, ze, m, n, x, bs, th, inv_keep, dseed);
            }
        }
    }
}


// Specialised form of epilogue_staged for the epilogue classes of gemm_common.h on aligned views (launcher-checked): the
// same LDS transposition, but per 32-row half the row visits of a lane are batched in pairs - side loads (residual /
// pre-activation / accumulate target) issued first, all eight LDS reads together, then the arithmetic of the class only,
// then the stores - with no ragged-tail path and no per-visit flag tests.  The generic form spent ~3.5 us per tile here
// (a quarter of a K = 768 launch: the workgroups of a CU reach their epilogues together and are bound by VALU issue).
// EPIX: class of gemm_common.h, or 4 / 5 = ACT / ACTGRAD whose side tensor is the local derivative (SMX_ACT_SAVE_GRAD),
// or 6 = F32 without the accumulate form
template <int EPIX, int QBX = 0, int NH = 2>       // QBX: row visits per batch (0: by class); NH: 32-row halves of the wave's sub-tile
__device__ __forceinline__ void epilogue_staged_fast(const SmxGemmParams& p, f32x4_t (&acc)[2 * NH][4], char* wbuf, int mw0, int nw0,
                                                     long long zc, long long zbias, long long ze, int lane) {
    constexpr int EPI = EPIX == 4 ? PP_EPI_ACT : EPIX == 5 ? PP_EPI_ACTGRAD : EPIX == 6 ? PP_EPI_F32 : EPIX;
    constexpr bool sg = EPIX == 4 || EPIX == 5;
    constexpr bool F32_PLAIN = EPIX == 6;          // fp32 output that is only written (split-K slabs): no read-modify-write state
    const int i16 = lane & 15, g = lane >> 4;
    const int rr = lane >> 3, cc = lane & 7;
    const int n = nw0 + cc * 8;
    const bool nok = n < p.N;                          // N % 8 == 0: all 8 columns or none
    float bs[8];
    if (EPIX != 5 && p.bias && nok) load8(p.bias + zbias + n, bs);     // (class 5, a data gradient, is launched without a bias)
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = 0.f;
    }
    const unsigned th = smx_thresh24(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    const bool drop = EPI != PP_EPI_F32 && p.drop_p > 0.f;
    const unsigned dseed = smx_dseed(p.drop_p, p.drop_seed);          // + the step key (the kernels hand in the kernarg copy of p)
    const bool has_res = EPI == PP_EPI_LINEAR && p.resid;
    const bool has_acc = EPI == PP_EPI_F32 && !F32_PLAIN && p.atomic == 2;
    const bool has_aux = EPI == PP_EPI_ACT && p.aux_out;
    const int act = p.act & 0xff;
    // (the ACT class has no register to spare for the second addressing form: it keeps view_off)
    const bool c_plain = EPI != PP_EPI_ACT && p.c.rows_per_batch <= 0, e_plain = EPI != PP_EPI_ACT && p.e.rows_per_batch <= 0;
    const bool need_e = EPI == PP_EPI_ACT;          // (side rows compute their own offsets in side_load)
    const smx_f2 al2 = SMX_PK(p.alpha);
    constexpr int QB = QBX ? QBX : EPI == PP_EPI_F32 ? 2 : 1;         // row visits per batch: what fits beside the 64 accumulator registers
    // residual / pre-activation rows are requested one batch AHEAD of their use (across the two halves too), so their
    // latency sits behind the previous batch's arithmetic and the LDS round trip instead of in front of every row visit
    constexpr bool SIDE = EPI == PP_EPI_ACTGRAD || EPI == PP_EPI_LINEAR;
    auto side_load = [&](int hh, int q) -> uint4 {
        const int m = mw0 + hh * 32 + q * 8 + rr;
        const bool okk = nok && m < p.M && (EPI == PP_EPI_ACTGRAD || has_res);
        if (!okk) return make_uint4(0, 0, 0, 0);
        const long long e = ze + n + (e_plain ? p.e.off + (long long)m * p.e.ld : view_off(p.e, m));
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(EPI == PP_EPI_ACTGRAD ? p.aux_in : p.resid) + e);
    };
    uint4 side_nxt[QB];
    if constexpr (SIDE) {
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) side_nxt[qi] = side_load(0, qi);
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lr = i2 * 16 + i16, c = j * 4 + g;
                *reinterpret_cast<f32x4_t*>(wbuf + lr * 256 + ((c ^ (lr & 15)) << 4)) = acc[2 * h + i2][j];
            }
#pragma unroll
        for (int q0 = 0; q0 < 4; q0 += QB) {
            long long cb[QB], eb[QB];
            bool ok[QB];
            uint4 side[QB];
            float4 old[(EPI == PP_EPI_F32 && !F32_PLAIN) ? QB : 1][2];
            f32x4_t lo[QB], hi[QB];
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                const int lr = (q0 + qi) * 8 + rr, m = mw0 + h * 32 + lr;
                ok[qi] = nok && m < p.M;
                const int mm = ok[qi] ? m : 0;
                // plain row views (the usual case, wave-uniform test): one 64-bit multiply-add instead of the batched-view decode
                cb[qi] = zc + n + (c_plain ? p.c.off + (long long)mm * p.c.ld : view_off(p.c, mm));
                eb[qi] = !need_e ? 0 : ze + n + (e_plain ? p.e.off + (long long)mm * p.e.ld : view_off(p.e, mm));
                if constexpr (SIDE) {
                    side[qi] = side_nxt[qi];
                } else if (EPI == PP_EPI_F32 && !F32_PLAIN) {
                    if (has_acc && ok[qi]) {
                        const float* c = reinterpret_cast<const float*>(p.C) + cb[qi];
                        old[qi][0] = *reinterpret_cast<const float4*>(c);
                        old[qi][1] = *reinterpret_cast<const float4*>(c + 4);
                    } else {
                        old[qi][0] = old[qi][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                lo[qi] = *reinterpret_cast<const f32x4_t*>(wbuf + lr * 256 + (((2 * cc) ^ (lr & 15)) << 4));
                hi[qi] = *reinterpret_cast<const f32x4_t*>(wbuf + lr * 256 + (((2 * cc + 1) ^ (lr & 15)) << 4));
            }
            if constexpr (SIDE) {
                constexpr int NB = 4 / QB;                               // batches per half
                const int b1 = h * NB + q0 / QB + 1;                     // (compile-time after unrolling)
                if (b1 < NH * NB) {
#pragma unroll
                    for (int qi = 0; qi < QB; ++qi) side_nxt[qi] = side_load(b1 / NB, (b1 % NB) * QB + qi);
                }
            }
#pragma unroll
            for (int qi = 0; qi < QB; ++qi) {
                const int m = mw0 + h * 32 + (q0 + qi) * 8 + rr;
                float x[8] = {lo[qi][0], lo[qi][1], lo[qi][2], lo[qi][3], hi[qi][0], hi[qi][1], hi[qi][2], hi[qi][3]};
                if constexpr (EPI == PP_EPI_ACT) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], p.alpha, bs[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {          // packed fp32: two columns per instruction
                        const smx_f2 r = __builtin_elementwise_fma((smx_f2){x[e], x[e + 1]}, al2, (smx_f2){bs[e], bs[e + 1]});
                        x[e] = r[0];
                        x[e + 1] = r[1];
                    }
                }
                if (EPI == PP_EPI_F32 && F32_PLAIN) {
                    if (ok[qi]) {
                        float* c = reinterpret_cast<float*>(p.C) + cb[qi];
                        *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
                        *reinterpret_cast<float4*>(c + 4) = make_float4(x[4], x[5], x[6], x[7]);
                    }
                    continue;
                }
                if (EPI == PP_EPI_F32) {
                    x[0] += old[qi][0].x; x[1] += old[qi][0].y; x[2] += old[qi][0].z; x[3] += old[qi][0].w;
                    x[4] += old[qi][1].x; x[5] += old[qi][1].y; x[6] += old[qi][1].z; x[7] += old[qi][1].w;
                    if (ok[qi]) {
                        float* c = reinterpret_cast<float*>(p.C) + cb[qi];
                        *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
                        *reinterpret_cast<float4*>(c + 4) = make_float4(x[4], x[5], x[6], x[7]);
                    }
                    continue;
                }
                if (EPI == PP_EPI_ACT) {
                    const unsigned didx = (unsigned)((long long)m * p.N + n + zc);
                    if constexpr (sg) {       // side tensor = local derivative: activation, mask and derivative pair by pair
                        const uint4 d = act_fwd_grad_drop8(x, act, drop, dseed, didx, th, inv_keep);
                        if (has_aux && ok[qi]) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.aux_out) + eb[qi]) = d;
                    } else {
                        if (has_aux && ok[qi])
                            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.aux_out) + eb[qi]) =
                                make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
                        act_fwd8(x, act);
                        if (drop) smx_drop_mul8(dseed, didx, th, inv_keep, x);
                    }
                }
                if (EPI == PP_EPI_ACTGRAD) {
                    const uint4 u = side[qi];
                    if constexpr (sg) {
                        x[0] *= __uint_as_float(u.x << 16); x[1] *= __uint_as_float(u.x & 0xffff0000u);
                        x[2] *= __uint_as_float(u.y << 16); x[3] *= __uint_as_float(u.y & 0xffff0000u);
                        x[4] *= __uint_as_float(u.z << 16); x[5] *= __uint_as_float(u.z & 0xffff0000u);
                        x[6] *= __uint_as_float(u.w << 16); x[7] *= __uint_as_float(u.w & 0xffff0000u);
                    } else {
                        float s[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                      __uint_as_float(u.y & 0xffff0000u), __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                      __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
                        act_grad_mul8(x, s, act);
                        if (drop) smx_drop_mul8(dseed, (unsigned)((long long)m * p.N + n + zc), th, inv_keep, x);
                    }
                }
                if (EPI == PP_EPI_LINEAR && drop) smx_drop_mul8(dseed, (unsigned)((long long)m * p.N + n + zc), th, inv_keep, x);
                if (EPI == PP_EPI_LINEAR && has_res) {
                    const uint4 u = side[qi];
                    x[0] += __uint_as_float(u.x << 16); x[1] += __uint_as_float(u.x & 0xffff0000u);
                    x[2] += __uint_as_float(u.y << 16); x[3] += __uint_as_float(u.y & 0xffff0000u);
                    x[4] += __uint_as_float(u.z << 16); x[5] += __uint_as_float(u.z & 0xffff0000u);
                    x[6] += __uint_as_float(u.w << 16); x[7] += __uint_as_float(u.w & 0xffff0000u);
                }
                if (ok[qi])
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + cb[qi]) =
                        make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
            }
        }
    }
}

#ifndef TR1_MINWG
#define TR1_MINWG 4
#endif
// BMH: 64-row halves of the output tile.  2: 128 x 128.  1: 64 x 128 (wave sub-tile 32 x 64) for launches whose 128-row tiling
// leaves most of the 1024 resident workgroup slots empty (the LM's M = 7 968 and M = 1 024 GEMMs): twice the workgroups,
// each with half the A fill and half the MFMAs per K step.  K-contiguous A operands only.
template <bool A_RC, bool B_RC, int EPI = -1, int BMH = 2>       // EPI: epilogue class (gemm_common.h; 4 / 5: see epilogue_staged_fast) on aligned views, -1: generic
__global__ __launch_bounds__(256, TR1_MINWG) void gemm_bf16_dma_kernel(SmxGemmParams p) {
    static_assert(BMH == 2 || !A_RC, "64-row tiles: K-contiguous A only");
    constexpr int TBM = 64 * BMH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef SMX_PP_LAB
    if (p.drop_seed == 0xdead0001u) return;      // ablation builds (tools/gpu_small_gemm.py): launch cost only
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: scalar registers)
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + TBM - 1) / TBM;
    const int nwg = ntn * ntm;
    // Split-K launches (weight gradients: few output tiles, very long K) are bound by the operand fills.  All tiles of
    // one K slice read the same rows of A and B, so they should share an L2.  Workgroups go to XCDs round-robin in
    // dispatch order (dispatch id L -> XCD L % 8): give XCD c a CONTIGUOUS range of the slice-major work list
    // (slice, tile), i.e. whole slices or contiguous tile runs of at most two slices.
    const bool xcd_split = p.split_k > 1 && p.nbatch == 1 && (int)gridDim.x == nwg;
    int lin0 = blockIdx.x, z = blockIdx.z;
    if (xcd_split) {
        const int W = nwg * p.split_k, L = blockIdx.x + nwg * blockIdx.z;
        const int c = L & 7, j = L >> 3, base = W >> 3, rem = W & 7;
        const int w = c * base + min(c, rem) + j;
        z = w / nwg;
        lin0 = w - z * nwg;
    }
    // persistent over tiles when launched with fewer blocks than tiles: blocks then walk the tile list in step,
    // so the workgroups sharing an L2 read the same K slices of their shared panels at about the same time
    for (int lin = lin0; lin < nwg; lin += gridDim.x) {
    int wg = lin;
    if (!xcd_split) {
        const int q = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    // column-block rasterisation: consecutive tile ids sweep GROUP_N n-tiles of one m-panel, then the next
    // m-panel, inside one block of n-tiles.  With the XCD remap above, each XCD's L2 then holds a 1.5-MB block of
    // B while streaming A panels once (instead of cycling all of B through a 4-MB L2 for every m-panel).
    int tm, tn;
    {
        const int per_group = GROUP_N * ntm;
        const int grp = wg / per_group, rem = wg - grp * per_group;
        const int first = grp * GROUP_N;
        const int gsz = min(ntn - first, GROUP_N);
        tm = rem / gsz;
        tn = first + (rem - tm * gsz);
    }
    const int m0 = tm * TBM, n0 = tn * BN;
    const int zb = z / p.split_k, zs = z - zb * p.split_k;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long long)zb * p.batch_a;
    const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (long long)zb * p.batch_b;
    const long long zc = (long long)zb * p.batch_c + (long long)zs * p.split_stride, zbias = (long long)zb * p.batch_bias, ze = (long long)zb * p.batch_e;
    const int ksteps_total = (p.K + BK - 1) / BK;
    const int per = (ksteps_total + p.split_k - 1) / p.split_k;
    const int ks0 = zs * per, ks1 = min(ksteps_total, ks0 + per);
    if (ks0 >= ks1 && p.split_k > 1 && p.atomic == 1) continue;

    DmaLoader<A_RC, 2 * BMH> la;
    DmaLoader<B_RC> lb;
    la.init(A, p.a, m0, p.M, ks0 * BK, tid);
    lb.init(B, p.b, n0, p.N, ks0 * BK, tid);

    f32x4_t acc[2 * BMH][4];
#pragma unroll
    for (int i = 0; i < 2 * BMH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    char* tA = smem;
    char* tB = smem + 16384;
    // Waves whose whole sub-tile lies outside the output (edge tiles; the positional conv's 48-column groups use one of a
    // tile's two wave columns) skip their fragment reads and MFMAs - the other workgroups of the CU use the idle pipes.
    // (Finer, per-16-block predication made hipcc keep two copies of the accumulators: 220 spilled registers; skipping the
    // fills of rows nobody reads put eight scalar branches into every K step and cost more than the fills.)
    const bool wave_on = m0 + wm * 32 * BMH < p.M && n0 + wn * 64 < p.N;
    for (int ks = ks0; ks < ks1; ++ks) {
        la.issue(tA, p.a, m0, p.M, ks * BK, p.K, tid);
        lb.issue(tB, p.b, n0, p.N, ks * BK, p.K, tid);
        __syncthreads();                 // (the compiler drains the LDS-DMA queue before the barrier)
        if (wave_on) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[2 * BMH], fb[4];
#pragma unroll
            for (int i = 0; i < 2 * BMH; ++i) fa[i] = load_frag<A_RC>(tA, wm * 32 * BMH + i * 16, kk, lane, 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_RC>(tB, wn * 64 + j * 16, kk, lane, 1);
#pragma unroll
            for (int i = 0; i < 2 * BMH; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        }
        __syncthreads();                 // everyone is done reading before the next fill overwrites the tile
    }
#ifdef SMX_PP_LAB
    if (p.drop_seed == 0xdead0002u) { if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.C)[tid] = acc[1][1][1] + acc[2 * BMH - 1][2][2] + acc[2 * BMH - 1][3][3]; return; }   // ablation builds: no epilogue
#endif
    // the K loop ended on a barrier: the tile buffers are free, each wave transposes through its own 8-KB slice.
    // The lane id and the parameter block are re-read behind an opaque asm: everything the epilogue derives from them
    // (LDS addresses, row offsets, flags) is then computed HERE instead of being hoisted above the K loop, where it would
    // have to live across the loop on a 128-register budget (hipcc spilled a dozen VGPRs to scratch for it, and a kernel
    // that uses scratch at all pays ~1.3 us more per launch).
    {
        int lane_e = lane, wave_e = wave;
        asm volatile("" : "+v"(lane_e), "+v"(wave_e));
        wave_e = __builtin_amdgcn_readfirstlane(wave_e);
        auto ka = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const SmxGemmParams& pe = *(const SmxGemmParams*)ka;
        if (!wave_on) {
            // nothing of this wave's sub-tile lies inside the output
        } else if constexpr (EPI >= 0)
            epilogue_staged_fast<EPI, (EPI == 6 && !A_RC && B_RC) ? 1 : 0, BMH>(pe, acc, smem + wave_e * 8192, m0 + (wave_e >> 1) * 32 * BMH, n0 + (wave_e & 1) * 64, zc, zbias, ze, lane_e);
        else
            epilogue_staged<BMH>(pe, acc, smem + wave_e * 8192, m0 + (wave_e >> 1) * 32 * BMH, n0 + (wave_e & 1) * 64, zc, zbias, ze, lane_e);
    }
    if (lin + (int)gridDim.x < nwg) __syncthreads();   // slices are tile memory again for the next fill
    }   // tile loop
}

// ------------------------------------------------------------------------------------------------
// Eight waves on a 256 x 128 tile, two workgroups per CU (tr_mode 11).  The kernel above moves 128 KB into LDS per CU and K
// step (four 32-KB tiles), and its K loop runs at exactly the rate that volume is delivered (~42 B/clk/CU); this form keeps
// its structure - independent single-stage workgroups hiding each other's fills, 64 x 64 wave tiles, the same epilogue - on
// tiles that need 25 % fewer bytes per flop (48 KB per 256 x 128 x 64).  Measured against it (tools/lab/gemm16w_lab.hip, same
// loader / fragment / epilogue code, bit-identical results): +13 % on 16 k x 3072 x 768, +8 % on 16 k x 768 x 3072, +22 % on
// 16 k x 4096 x 1024, equal or worse where 256-row tiles quantise badly; the tuner picks per shape.  K-contiguous A, no split-K,
// classes with the specialised epilogue.
// ------------------------------------------------------------------------------------------------
template <bool B_RC, int EPI>
__global__ __launch_bounds__(512, 4) void gemm_bf16_dma8_kernel(SmxGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];            // 64 KB: 48 KB of tiles, 8 x 8 KB epilogue slices
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int q = wave >> 2, tq = tid & 255;          // loader role: half q fills A rows 128 q .. and its half of the B tile
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + 255) / 256;
    const int nwg = ntn * ntm;
    for (int lin = blockIdx.x; lin < nwg; lin += gridDim.x) {
        int wg = lin;
        {
            const int qq = nwg >> 3, r = nwg & 7, x = wg & 7, y = wg >> 3;
            wg = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + y;
        }
        int tm, tn;
        {
            const int per_group = GROUP_N * ntm;
            const int grp = wg / per_group, rem = wg - grp * per_group;
            const int first = grp * GROUP_N;
            const int gsz = min(ntn - first, GROUP_N);
            tm = rem / gsz;
            tn = first + (rem - tm * gsz);
        }
        const int m0 = tm * 256, n0 = tn * BN;
        const int z = blockIdx.z;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A) + (long long)z * p.batch_a;
        const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B) + (long long)z * p.batch_b;
        const long long zc = (long long)z * p.batch_c, zbias = (long long)z * p.batch_bias, ze = (long long)z * p.batch_e;
        const int ks1 = (p.K + BK - 1) / BK;
        DmaLoader<false> la;
        DmaLoader<B_RC, 2> lb;              // K-contiguous B: my 64 of its 128 rows; rows-contiguous B: two of its four k-row passes
        la.init(A, p.a, m0 + q * 128, p.M, 0, tq);
        lb.init(B, p.b, B_RC ? n0 : n0 + q * 64, p.N, 0, tq);
        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        char* tA = smem + (wm >> 1) * 16384;
        char* tB = smem + 32768;
        const bool wave_on = m0 + wm * 64 < p.M && n0 + wn * 64 < p.N;
        for (int ks = 0; ks < ks1; ++ks) {
            la.issue(smem + q * 16384, p.a, m0 + q * 128, p.M, ks * BK, p.K, tq);
            if constexpr (B_RC) {
                lb.template issue<false, 2>(tB, p.b, n0, p.N, ks * BK, p.K, tq, 2 * q);
            } else {
                lb.issue(tB + q * 8192, p.b, n0 + q * 64, p.N, ks * BK, p.K, tq);
            }
            __syncthreads();                 // (the compiler drains the LDS-DMA queue before the barrier)
            if (wave_on) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8_t fa[4], fb[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i] = load_frag<false>(tA, (wm & 1) * 64 + i * 16, kk, lane, 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_RC>(tB, wn * 64 + j * 16, kk, lane, 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            }
            __syncthreads();                 // everyone is done reading before the next fill overwrites the tile
        }
        {
            int lane_e = lane, wave_e = wave;
            asm volatile("" : "+v"(lane_e), "+v"(wave_e));
            wave_e = __builtin_amdgcn_readfirstlane(wave_e);
            auto ka = __builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            const SmxGemmParams& pe = *(const SmxGemmParams*)ka;
            if (wave_on)
                epilogue_staged_fast<EPI, 0>(pe, acc, smem + wave_e * 8192, m0 + (wave_e >> 1) * 64, n0 + (wave_e & 1) * 64, zc, zbias, ze, lane_e);
        }
        if (lin + (int)gridDim.x < nwg) __syncthreads();   // slices are tile memory again for the next fill
    }
}

template <bool B_RC, int EPI>
static void dma8_launch(const SmxGemmParams& p, dim3 grid, hipStream_t stream) {
    static bool attr_done = false;          // (once per instantiation: the attribute call costs tens of microseconds of host time)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_dma8_kernel<B_RC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_bf16_dma8_kernel<B_RC, EPI>), grid, dim3(512), 65536, stream, p);
}

int smx_gemm_pp(const SmxGemmParams& p, hipStream_t stream);   // gemm_pp.hip
int smx_gemm_fr(const SmxGemmParams& p, hipStream_t stream);   // gemm_fr.hip
int smx_gemm_ws(const SmxGemmParams& p, hipStream_t stream);   // gemm_ws.hip

// ------------------------------------------------------------------------------------------------
// fp32: simple 64x64x16 VALU tile kernel with fully generic operand addressing.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_f32_kernel(SmxGemmParams p) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const int tid = threadIdx.x;
    const int ntn = (p.N + 63) / 64;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * 64, n0 = tn * 64;
    const int z = blockIdx.z;
    const int zb = z / p.split_k, zs = z - zb * p.split_k;
    const float* A = reinterpret_cast<const float*>(p.A) + (long long)zb * p.batch_a;
    const float* B = reinterpret_cast<const float*>(p.B) + (long long)zb * p.batch_b;
    const long long zc = (long long)zb * p.batch_c + (long long)zs * p.split_stride, zbias = (long long)zb * p.batch_bias, ze = (long long)zb * p.batch_e;
    const int ksteps_total = (p.K + 15) / 16;
    const int per = (ksteps_total + p.split_k - 1) / p.split_k;
    const int ks0 = zs * per, ks1 = min(ksteps_total, ks0 + per);
    if (ks0 >= ks1 && p.split_k > 1 && p.atomic == 1) return;

    const int tx = tid & 15, ty = tid >> 4;  // each thread: 4 m (ty*4..) x 4 n (tx*4..)
    float acc[4][4] = {};
    for (int ks = ks0; ks < ks1; ++ks) {
        const int k0 = ks * 16;
        // cooperative loads: 1024 elements per operand, 4 per thread
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = tid + 256 * e;
            {   // A tile element (r, k)
                int r, k;
                if (p.a_rc) { r = idx & 63; k = idx >> 6; } else { k = idx & 15; r = idx >> 4; }
                float v = 0.f;
                if (m0 + r < p.M && k0 + k < p.K)
                    v = p.a_rc ? A[view_off(p.a, k0 + k) + m0 + r] : A[view_off(p.a, m0 + r) + k0 + k];
                As[k][r] = v;
            }
            {
                int r, k;
                if (p.b_rc) { r = idx & 63; k = idx >> 6; } else { k = idx & 15; r = idx >> 4; }
                float v = 0.f;
                if (n0 + r < p.N && k0 + k < p.K)
                    v = p.b_rc ? B[view_off(p.b, k0 + k) + n0 + r] : B[view_off(p.b, n0 + r) + k0 + k];
                Bs[k][r] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
        const long long rowb = zc + view_off(p.c, m);
        const long long rowe = ze + view_off(p.e, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float x = acc[i][j] * p.alpha;
            if (p.bias) x += p.bias[zbias + n];
            if (p.aux_out) reinterpret_cast<float*>(p.aux_out)[rowe + n] = x;
            if (!p.aux_in) x = act_fwd(x, p.act);
            else x *= act_grad(reinterpret_cast<const float*>(p.aux_in)[rowe + n], p.act);
            if (p.drop_p > 0.f)
                x *= smx_drop_mul(p.drop_seed, (unsigned)((long long)m * p.N + n + zc), smx_thresh24(p.drop_p),
                                  1.0f / (1.0f - p.drop_p));
            if (p.resid) x += reinterpret_cast<const float*>(p.resid)[rowe + n];
            float* c = reinterpret_cast<float*>(p.C) + rowb + n;
            if (p.atomic == 1) atomicAdd(c, x); else if (p.atomic == 2) *c += x; else *c = x;
        }
    }
}

// dst[i] (+)= sum_s slabs[s * stride + i]   (second stage of the split-K weight gradients)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nsplit, long long n, long long stride,
                                    float* __restrict__ dst, int accumulate) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 4 <= n; i += step) {
        float4 a = accumulate ? *reinterpret_cast<const float4*>(dst + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < nsplit; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + s * stride + i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + i) = a;
    }
}
extern "C" int smx_reduce_slabs(const float* slabs, int nsplit, long long n, long long stride, float* dst, int accumulate,
                                hipStream_t stream) {
    (void)hipGetLastError();
    if (n <= 0 || (n & 3) || (stride & 3) || nsplit < 1) return SMX_EINVAL;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, slabs, nsplit, n, stride, dst, accumulate);
    SMX_CHECK_LAUNCH();
}

// reduce_slabs for several (slabs, destination) pairs in one launch (the second stage of a grouped weight-gradient launch)
#define SMX_REDUCE_MAX 8
struct SmxReduceTable {
    const float* slabs[SMX_REDUCE_MAX];
    float* dst[SMX_REDUCE_MAX];
    long long n[SMX_REDUCE_MAX];             // elements per slab (= slab stride), % 4 == 0
    int nsplit[SMX_REDUCE_MAX];
    int count, accumulate;
};
__global__ void reduce_slabs_many_kernel(SmxReduceTable t) {
    const int e = blockIdx.y;
    const float* __restrict__ slabs = t.slabs[e];
    float* __restrict__ dst = t.dst[e];
    const long long n = t.n[e];
    const int nsplit = t.nsplit[e];
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 4 <= n; i += step) {
        float4 a = t.accumulate ? *reinterpret_cast<const float4*>(dst + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < nsplit; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(slabs + s * n + i);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(dst + i) = a;
    }
}
extern "C" int smx_reduce_slabs_many(const float* const* slabs, float* const* dst, const long long* n, const int* nsplit, int count,
                                     int accumulate, hipStream_t stream) {
    (void)hipGetLastError();
    if (!slabs || !dst || !n || !nsplit || count < 1 || count > SMX_REDUCE_MAX) return SMX_EINVAL;
    SmxReduceTable t = {};
    long long nmax = 0;
    for (int e = 0; e < count; ++e) {
        if (!slabs[e] || !dst[e] || n[e] <= 0 || (n[e] & 3) || nsplit[e] < 1) return SMX_EINVAL;
        t.slabs[e] = slabs[e]; t.dst[e] = dst[e]; t.n[e] = n[e]; t.nsplit[e] = nsplit[e];
        nmax = n[e] > nmax ? n[e] : nmax;
    }
    t.count = count; t.accumulate = accumulate;
    long long blocks = (nmax / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(reduce_slabs_many_kernel, dim3((unsigned)blocks, count), dim3(256), 0, stream, t);
    SMX_CHECK_LAUNCH();
}

// Second stage of a split-K forward / data-gradient GEMM (outputs with fewer tiles than CUs: the decoder, the LM head):
// C = epilogue(sum_s slabs[s]) with the full epilogue of SmxGemmParams (bias, activation or activation gradient,
// dropout, residual, aux_out, bf16 / fp32 / accumulate).  slabs: nsplit x [M, ldn] fp32, rows padded to ldn % 8 == 0.
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(SmxGemmParams p, const float* __restrict__ slabs, int nsplit,
                                                              long long stride, int ldn) {
    p.drop_seed = smx_dseed(p.drop_p, p.drop_seed);        // + the step key (smx_common.h), read once
    const int chunks = (p.N + 7) >> 3;
    const long long total = (long long)p.M * chunks;
    const unsigned th = smx_thresh24(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / chunks), n = (int)(i - (long long)m * chunks) * 8;
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* src = slabs + (long long)m * ldn + n;
        for (int s = 0; s < nsplit; ++s) {
            float v[8];
            load8(src + s * stride, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] += v[e];
        }
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = (p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
        epilogue_row8<false, true>(p, 0, 0, m, n, x, bs, th, inv_keep, p.drop_seed);
    }
}
extern "C" int smx_gemm_splitk_epilogue(const SmxGemmParams* pp, const float* slabs, int nsplit, long long stride, int ldn,
                                        hipStream_t stream) {
    (void)hipGetLastError();
    SmxGemmParams p = *pp;
    if (p.M <= 0 || p.N <= 0 || nsplit < 1 || (ldn & 7) || ldn < p.N || (stride & 7) || p.nbatch > 1) return SMX_EINVAL;
    const long long total = (long long)p.M * ((p.N + 7) >> 3);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, p, slabs, nsplit, stride, ldn);
    SMX_CHECK_LAUNCH();
}

extern "C" int smx_gemm(const SmxGemmParams* pp, int dtype, hipStream_t stream) {
    (void)hipGetLastError();  // drop stale errors left by other runtime users
    SmxGemmParams p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K < 0) return SMX_EINVAL;
    if (p.nbatch < 1) p.nbatch = 1;
    if (p.split_k < 1) p.split_k = 1;
    if (p.split_k > 1 && p.atomic != 1 && (p.atomic == 2 || p.split_stride <= 0)) return SMX_EINVAL;
    if (p.atomic && !(p.out_f32 || dtype == SMX_F32)) return SMX_EINVAL;
    if (dtype == SMX_F32) {
        if (p.act & SMX_ACT_SAVE_GRAD) return SMX_EINVAL;        // bf16 kernels only
        p.out_f32 = 1;
        dim3 grid(((p.M + 63) / 64) * ((p.N + 63) / 64), 1, p.nbatch * p.split_k);
        hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, p);
        SMX_CHECK_LAUNCH();
    }
    if (dtype != SMX_BF16) return SMX_EINVAL;
    if ((p.act & SMX_ACT_SAVE_GRAD) && ((p.tr_mode & 255) == 0 || (p.tr_mode & 255) == 2)) return SMX_EINVAL;   // production kernels only
    dim3 grid(((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN), 1, p.nbatch * p.split_k);
    if ((p.tr_mode & 255) == 8) return smx_gemm_pp(p, stream);        // 256 x 256, persistent ping-pong
    if ((p.tr_mode & 255) == 12 || (p.tr_mode & 255) == 13) return smx_gemm_fr(p, stream);       // 256 x 256 / 192 x 256, persistent free-running schedule
    if ((p.tr_mode & 255) == 14) return smx_gemm_ws(p, stream);          // 192 x 256, twelve compute + four loader waves (gemm_ws.hip)
    if (p.tr_mode == 1 || p.tr_mode == 7) { if (grid.x > 1024) grid.x = 1024; p.tr_mode = 1; }   // persistent tile walk (4 WG/CU resident)
    if (p.tr_mode == 11) {    // 256 x 128 tiles, eight waves, two workgroups per CU (gemm_bf16_dma8_kernel)
        const bool flagged = (p.act & SMX_ACT_SAVE_GRAD) != 0;
        const int epi = (p.atomic == 0 && smx_epi_views_aligned(p)) ? pp_epi_class(p) : -1;
        if (p.a_rc || p.split_k != 1 || epi < 0 || epi == PP_EPI_F32) return SMX_EINVAL;
        dim3 g11(((p.M + 255) / 256) * ((p.N + BN - 1) / BN), 1, p.nbatch);
        if (g11.x > 512) g11.x = 512;
        p.tr_mode = 1;
#define TR11_GO(BR, E) dma8_launch<BR, E>(p, g11, stream)
        if (!p.b_rc) {
            if (epi == PP_EPI_LINEAR) TR11_GO(false, PP_EPI_LINEAR);
            else if (epi == PP_EPI_ACT && flagged) TR11_GO(false, 4);
            else if (epi == PP_EPI_ACT) TR11_GO(false, PP_EPI_ACT);
            else return SMX_EINVAL;
        } else {
            if (epi == PP_EPI_LINEAR) TR11_GO(true, PP_EPI_LINEAR);
            else if (epi == PP_EPI_ACTGRAD && flagged && !p.bias) TR11_GO(true, 5);
            else if (epi == PP_EPI_ACTGRAD && !flagged) TR11_GO(true, PP_EPI_ACTGRAD);
            else return SMX_EINVAL;
        }
#undef TR11_GO
        SMX_CHECK_LAUNCH();
    }
    if (p.tr_mode == 9) {     // 64 x 128 tiles of the same kernel (gemm_bf16_dma_kernel<.., BMH = 1>): K-contiguous A, aligned classes
        const bool flagged = (p.act & SMX_ACT_SAVE_GRAD) != 0;
        const int epi = (p.atomic == 0 && smx_epi_views_aligned(p)) ? pp_epi_class(p) : -1;
        if (p.a_rc || p.split_k != 1 || epi < 0 || epi == PP_EPI_F32) return SMX_EINVAL;
        dim3 g9(((p.M + 63) / 64) * ((p.N + BN - 1) / BN), 1, p.nbatch);
        if (g9.x > 1024) g9.x = 1024;
        p.tr_mode = 1;
#define TR9_GO(BR, E) hipLaunchKernelGGL((gemm_bf16_dma_kernel<false, BR, E, 1>), g9, dim3(256), STAGE_BYTES, stream, p)
        if (!p.b_rc) {
            if (epi == PP_EPI_LINEAR) TR9_GO(false, PP_EPI_LINEAR);
            else if (epi == PP_EPI_ACT && flagged) TR9_GO(false, 4);
            else if (epi == PP_EPI_ACT) TR9_GO(false, PP_EPI_ACT);
            else return SMX_EINVAL;
        } else {
            if (epi == PP_EPI_LINEAR) TR9_GO(true, PP_EPI_LINEAR);
            else if (epi == PP_EPI_ACTGRAD && flagged && !p.bias) TR9_GO(true, 5);
            else if (epi == PP_EPI_ACTGRAD && !flagged) TR9_GO(true, PP_EPI_ACTGRAD);
            else return SMX_EINVAL;
        }
#undef TR9_GO
        SMX_CHECK_LAUNCH();
    }
    if (p.tr_mode == 1 || p.tr_mode == 4) {   // LDS-DMA fills, 128x128 tile, 4 workgroups / CU
        const size_t ldsz = STAGE_BYTES;
        // specialised epilogue when the launch belongs to a class and every view allows 16-B accesses (SMX_TR1_EPI=0: off)
        static const bool fast_ok = !(getenv("SMX_TR1_EPI") && getenv("SMX_TR1_EPI")[0] == '0');
        const bool flagged = (p.act & SMX_ACT_SAVE_GRAD) != 0;
        int epi = ((fast_ok || flagged) && p.atomic != 1 && smx_epi_views_aligned(p)) ? pp_epi_class(p) : -1;
        // saved-derivative side tensors exist in the forward ACT and the data-gradient ACTGRAD epilogues (and in the split-K
        // epilogue kernel, whose slab-writing GEMM carries no activation)
        if (flagged && !((!p.a_rc && !p.b_rc && epi == PP_EPI_ACT) || (!p.a_rc && p.b_rc && epi == PP_EPI_ACTGRAD))) return SMX_EINVAL;
#define TR1_GO(AR, BR, E) hipLaunchKernelGGL((gemm_bf16_dma_kernel<AR, BR, E>), grid, dim3(256), ldsz, stream, p)
        // instantiated for the (layout, class) pairs the model launches: forward LINEAR / ACT, data gradient LINEAR / ACTGRAD,
        // weight gradient F32
        if (!p.a_rc && !p.b_rc) {
            if (epi == PP_EPI_LINEAR) TR1_GO(false, false, PP_EPI_LINEAR);
            else if (epi == PP_EPI_ACT && flagged) TR1_GO(false, false, 4);
            else if (epi == PP_EPI_ACT) TR1_GO(false, false, PP_EPI_ACT);
            else if (epi == PP_EPI_F32 && p.atomic == 0) TR1_GO(false, false, 6);  // split-K slabs of the decoder-side GEMMs
            else if (epi == PP_EPI_F32) TR1_GO(false, false, PP_EPI_F32);
            else TR1_GO(false, false, -1);
        } else if (!p.a_rc && p.b_rc) {
            if (epi == PP_EPI_LINEAR) TR1_GO(false, true, PP_EPI_LINEAR);
            else if (epi == PP_EPI_ACTGRAD && flagged && !p.bias) TR1_GO(false, true, 5);
            else if (flagged) return SMX_EINVAL;
            else if (epi == PP_EPI_ACTGRAD) TR1_GO(false, true, PP_EPI_ACTGRAD);
            else if (epi == PP_EPI_F32 && p.atomic == 0) TR1_GO(false, true, 6);
            else TR1_GO(false, true, -1);        // (the accumulate form of F32 does not fit 128 registers in this layout)
        } else if (p.a_rc && !p.b_rc) {
            TR1_GO(true, false, -1);
        } else {
            if (epi == PP_EPI_F32 && p.atomic == 0) TR1_GO(true, true, 6);
            else if (epi == PP_EPI_F32) TR1_GO(true, true, PP_EPI_F32);
            else TR1_GO(true, true, -1);
        }
#undef TR1_GO
        SMX_CHECK_LAUNCH();
    }
    if (p.tr_mode == 2) p.tr_mode = 1;   // register-staged kernel with transposing reads
    const size_t lds = 2 * STAGE_BYTES;
    if (!p.a_rc && !p.b_rc)
        hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, dim3(256), lds, stream, p);
    else if (!p.a_rc && p.b_rc)
        hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, dim3(256), lds, stream, p);
    else if (p.a_rc && !p.b_rc)
        hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(256), lds, stream, p);
    SMX_CHECK_LAUNCH();
}

// ABI self-description (checked by the ctypes binding against its struct mirrors)
extern "C" int smx_sizeof_SmxGemmParams(void) { return (int)sizeof(SmxGemmParams); }

SMX_STEP_KEY_TU(gemm)
