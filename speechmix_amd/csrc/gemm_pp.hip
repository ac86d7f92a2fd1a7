// ------------------------------------------------------------------------------------------------
// bf16 "ping-pong" GEMM: 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 per wave), ONE persistent workgroup per CU
// (same SmxGemmParams contract, operand views and row epilogue as the 128x128 kernels of gemm.hip).
//
// Why: a 128x128 tile needs 64 B/clk/CU of L2->LDS fill to keep the matrix pipes busy - the whole L2 bandwidth of the
// chip.  256x256 halves the fill per flop, but only one such workgroup fits a CU, so nothing else hides its latencies:
// the schedule has to.
//
//  * Every K tile (64 deep) is cut into four 16-KB LDS units: AH0/AH1 = first/second 64 rows of each wave-row group,
//    BH0/BH1 = first/second 32 columns of each wave-column group.  A K tile is computed in four phases of 16 MFMAs -
//    one quadrant of the wave's output x K=64 - in the order (AH0,BH0) (AH0,BH1) (AH1,BH1) (AH1,BH0): a phase reads at
//    most one new A and one new B register sub-tile, and the last phase reads nothing.
//  * The two wave-row groups run ONE barrier apart (the second group executes an extra s_barrier up front): while one
//    group's waves issue their 16 MFMAs, the partner wave on the same SIMD does its LDS reads and issues the LDS-DMA
//    of a later unit.  s_setprio(1) around the MFMA block.
//  * The (work item, K tile) pairs of a workgroup form ONE flat stream of units.  Each phase issues exactly one unit,
//    six units ahead of the one it consumes, so the fill of the next output tile's first K tiles is in flight while the
//    current tile's epilogue runs; the only waits are counted (s_waitcnt vmcnt(8): four units may stay in flight),
//    placed one phase before the first read of the unit they retire.  A unit is re-filled no earlier than two phases
//    after its last read.
//  * LDS-DMA = `buffer_load_dwordx4 ... offen lds` from inline asm (hipcc would drain a DMA it knows about before every
//    barrier and LDS read): per-lane byte offsets are fixed per work item, the K position rides in the scalar offset,
//    and rows / k outside the operand carry an offset beyond num_records, which the hardware fills with zeros.
//  * No LDS in the epilogue: the B fragments are read with their columns permuted (fragment j, column 4g+r of a
//    32-column half <-> logical column 8g + 4j + r), so a lane ends up owning 8 consecutive columns of one row per
//    half - exactly what the shared row epilogue (16-B accesses, fused bias / activation / dropout / residual) takes.
//    Its stores are inline asm too: any VMEM operation hipcc still tracks when the next K loop starts makes it drain
//    the whole queue before the first LDS read of every K tile.
//  * Work list: (K slice | batch, output tile), slice-major, handed out so that each XCD owns a contiguous range and the
//    workgroups of an XCD walk it together; inside a slice tiles are rasterised in column blocks of PP_GROUP tiles.
// ------------------------------------------------------------------------------------------------
#include "gemm_pp.h"

template <int PH, bool A_RC, bool B_RC, bool BVIEW, int LAB, class ISSUE>
__device__ __forceinline__ void pp_phase(f32x4_t (&acc)[8][4], bf16x8_t (&fa)[4][2],
                                         bf16x8_t (&fb0)[2][2], bf16x8_t (&fb1)[2][2], ISSUE& is,
                                         const char* cur, int tid, int lane, int wr, int wc, int wmode) {
    // LAB (ablation builds only): 1 no DMA, 2 no LDS reads, 4 no MFMA, 16 no epilogue, 64 fills re-read one K tile (L2-resident)
    // ---- load segment: register sub-tile reads + one unit of LDS-DMA, then the counted wait for the NEXT phase's unit
    if constexpr (!(LAB & 2)) {
        if (PH == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fb0[j][kk] = pp_bfrag<B_RC>(cur + 2 * PP_UNIT, wc * 32, j, kk, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PH == 0 || PH == 2) {
            const char* ha = cur + (PH == 0 ? 0 : 1) * PP_UNIT;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fa[a][kk] = load_frag<A_RC>(ha, wr * 64 + a * 16, kk, lane, 1);
        }
        if (PH == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) fb1[j][kk] = pp_bfrag<B_RC>(cur + 3 * PP_UNIT, wc * 32, j, kk, lane);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(LAB & 1)) {
        // wmode 0: counted wait.  1: this item's last phase - drain, so that the epilogue's stores (same counter, in-order)
        // do not sit between the prefetched units and the waits that retire them.  2: first K tile after a drain - the
        // units these four phases would retire were covered by it.
        const bool issued = is.template issue<(PH + 2) & 3, (LAB & 64) != 0>(tid);
        if (wmode == 2) {
        } else if (issued && wmode == 0) {
            PP_WAITV(8);
        } else {
            PP_WAITV(0);
        }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- compute segment: one quadrant x K = 64
    constexpr int rh = PH >> 1, ch = (PH == 1 || PH == 2) ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (!(LAB & 4))
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x8_t bfrag;
                if constexpr (ch == 1) bfrag = fb1[j][kk]; else bfrag = fb0[j][kk];
                acc[rh * 4 + a][ch * 2 + j] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag, fa[a][kk], acc[rh * 4 + a][ch * 2 + j], 0, 0, 0);
            }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
}

template <bool A_RC, bool B_RC, int EPI, int LAB, bool BVIEW, bool GRP, class KARG>
__device__ __forceinline__ void pp_kernel_body(const KARG& karg) {
    static_assert(!GRP || !BVIEW, "grouped launches: plain operand views");
    const SmxGemmParams& p = pp_kernarg_g<GRP>(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int ntm = (p.M + PP_BM - 1) / PP_BM, ntn = (p.N + PP_BN - 1) / PP_BN;
    int W = ntm * ntn * p.nbatch * p.split_k;
    if constexpr (GRP) W = pp_group().W;

    PPIssue<A_RC, B_RC, BVIEW, GRP> is;
    is.dv.init(p, ntm, ntn);
    is.g = 0;
    is.q = blockIdx.x; is.qstep = gridDim.x;
    is.seq = 0; is.kt = 0; is.nk = 0; is.k0 = 0; is.ahead = 0;
    is.wave_u = __builtin_amdgcn_readfirstlane(wave);
    is.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    is.K = p.K;
    is.load_item(tid);

    // prologue: K tile 0 entirely, AH0 + BH0 of K tile 1
    if constexpr (!(LAB & 1)) {
        bool all = true;
        all &= is.template issue<0>(tid);
        all &= is.template issue<1>(tid);
        all &= is.template issue<2>(tid);
        all &= is.template issue<3>(tid);
        all &= is.template issue<0>(tid);
        all &= is.template issue<1>(tid);
        if (all) PP_WAITV(8);
        else PP_WAITV(0);
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();     // second wave-row group runs one barrier behind

    f32x4_t acc[8][4];
    bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
    if constexpr ((LAB & 2) != 0) {
        for (int a = 0; a < 4; ++a) for (int k = 0; k < 2; ++k) fa[a][k] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
        for (int a = 0; a < 2; ++a) for (int k = 0; k < 2; ++k) fb0[a][k] = fb1[a][k] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
    int seq = 0, items = 0;
    bool drained = false;
    // the launcher sets bit 7 of tr_mode when the parameters fit this instantiation's epilogue class
    bool fast_epi = (p.tr_mode & 128) && pp_views_aligned(p);
    int gc = 0;                 // GRP: problem of the item being computed
    for (int q = blockIdx.x; q < W; q += gridDim.x) {
        PPItem it;
        if constexpr (GRP) {
            while (q >= pp_group().wstart[gc + 1]) ++gc;
            const SmxGemmParams& pg = pp_kernarg_g<true>(gc);
            PPDiv d;
            d.init(pg, (pg.M + PP_BM - 1) / PP_BM, (pg.N + PP_BN - 1) / PP_BN);
            pp_decode(pg, d, q - pp_group().wstart[gc], it);
            fast_epi = (pg.tr_mode & 128) && pp_views_aligned(pg);
        } else {
            pp_decode(pp_kernarg(), is.dv, q, it);
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        ++items;
        if (fast_epi && wr == 0) {          // bias slice of this item -> LDS, behind everything already in flight (the last
            const SmxGemmParams& pk = pp_kernarg_g<GRP>(gc);      // phase's drain retires it; an extra older operation only makes
            if (pk.bias) {                                // the counted waits stricter)
                pp_rsrc_t br = pp_make_rsrc(pk.bias + it.zbias + it.n0);
                br[2] = max(pk.N - it.n0, 0) * 4;
                pp_dma4(br, (unsigned)(wc * 64 + lane) * 4u, is.lds0 + PP_BIAS_OFF + (items & 1) * 1024 + wc * 256);
            }
        }
        for (int t = 0; t < it.nk; ++t) {
            const char* cur = smem + (seq & 1) * PP_STAGE;
            const int w0 = (t == 0 && drained) ? 2 : 0;
            const int w3 = (t == it.nk - 1) ? 1 : w0;
            pp_phase<0, A_RC, B_RC, BVIEW, LAB>(acc, fa, fb0, fb1, is, cur, tid, lane, wr, wc, w0);
            pp_phase<1, A_RC, B_RC, BVIEW, LAB>(acc, fa, fb0, fb1, is, cur, tid, lane, wr, wc, w0);
            pp_phase<2, A_RC, B_RC, BVIEW, LAB>(acc, fa, fb0, fb1, is, cur, tid, lane, wr, wc, w0);
            pp_phase<3, A_RC, B_RC, BVIEW, LAB>(acc, fa, fb0, fb1, is, cur, tid, lane, wr, wc, w3);
            ++seq;
        }
        drained = it.nk > 0;
        if constexpr ((LAB & 16) != 0) {
            float sink = 0.f;
            for (int a = 0; a < 8; ++a) for (int j = 0; j < 4; ++j) sink += acc[a][j][0] + acc[a][j][1] + acc[a][j][2] + acc[a][j][3];
            if (sink == 1234.5f) reinterpret_cast<float*>(pp_kernarg_g<GRP>(0).C)[tid] = sink;
        } else if (fast_epi) {
            pp_epilogue_fast<EPI, GRP, 8, B_RC>(acc, it.m0 + wr * 128, it.n0 + wc * 64, it.n0, smem + PP_BIAS_OFF + (items & 1) * 1024, it.zc,
                                       it.ze, lane, gc);
        } else {
            pp_epilogue<GRP, 8, B_RC>(acc, it.m0 + wr * 128, it.n0 + wc * 64, it.zc, it.zbias, it.ze, lane, gc);
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
}

template <bool A_RC, bool B_RC, int EPI, int LAB = 0, bool BVIEW = false>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(SmxGemmParams p) {
    pp_kernel_body<A_RC, B_RC, EPI, LAB, BVIEW, false>(p);
}
// grouped form (smx_gemm_group): the same body over the concatenated work lists of up to PP_MAXG problems
template <bool A_RC, bool B_RC, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_pp_group_kernel(SmxGemmGroup grp) {
    pp_kernel_body<A_RC, B_RC, EPI, 0, false, true>(grp);
}

template <bool A_RC, bool B_RC, int EPI, bool BVIEW = false>
static void pp_launch(const SmxGemmParams& p, dim3 grid, hipStream_t stream) {
    static bool attr_done[16] = {};          // per device: the 160-KB LDS opt-in is a per-device function attribute
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<A_RC, B_RC, EPI, 0, BVIEW>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        attr_done[dev & 15] = true;
    }
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<A_RC, B_RC, EPI, 0, BVIEW>), grid, dim3(512), PP_LDS_BYTES, stream, p);
}

int smx_gemm_pp(const SmxGemmParams& pin, hipStream_t stream) {
    SmxGemmParams p = pin;
    if (!pp_saved_ok(p)) return SMX_EINVAL;                // saved-derivative side tensors: the two fast classes only
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;
    }
    const long long W = (long long)((p.M + PP_BM - 1) / PP_BM) * ((p.N + PP_BN - 1) / PP_BN) * p.nbatch * p.split_k;
    const int kst = (p.K + BK - 1) / BK, per = (kst + p.split_k - 1) / p.split_k;
    // fp32 atomics: 128x128 kernels only; every K slice must own at least one K tile; K in whole 16-B chunks
    // (sizes below 2^22: the in-kernel index arithmetic divides through fp32 reciprocals)
    if (W >= (1 << 22) || p.M >= (1 << 22) || p.N >= (1 << 22) || p.K >= (1 << 22) || p.atomic == 1 || (p.split_k - 1) * per >= kst ||
        ((p.K & 7) && !(p.a_rc && p.b_rc))) return SMX_EINVAL;
    // rows-contiguous operands through a batched view: instantiated for the (RC, RC) layout (conv weight gradients) and for
    // the B operand of the (KC, RC) layout (conv data gradients: the tap-major packed weight read tap by tap)
    if (p.a_rc && !p.b_rc && p.a.rows_per_batch > 0) return SMX_EINVAL;
    // bits 16.. of tr_mode: cap of the persistent grid (the caller keeps CUs free for a kernel running beside this one)
    const int cap = (p.tr_mode >> 16) & 0xfff;
    const int wgs = cap > 0 && cap < ncu ? cap : ncu;
    dim3 grid((unsigned)(W < wgs ? W : wgs));
    const int lab = (p.tr_mode >> 8) & 0xff;
    const int epi = pp_epi_class(p);
    p.tr_mode = 8;
#ifdef SMX_PP_LAB
    if (lab == 32) { p.c.ld = 0; p.e.ld = 0; }       // every output row on top of row 0: same instructions, no write volume
    else
    if (!p.a_rc && !p.b_rc && lab) {
        const size_t ldsz = PP_LDS_BYTES;
#define PP_LABV(L)                                                                                                          \
    case L:                                                                                                                 \
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_kernel<false, false, 0, L>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES); \
        hipLaunchKernelGGL((gemm_bf16_pp_kernel<false, false, 0, L>), grid, dim3(512), ldsz, stream, p);                   \
        break;
        switch (lab) {
            PP_LABV(16) PP_LABV(17) PP_LABV(18) PP_LABV(19) PP_LABV(80) PP_LABV(64)
            default: return SMX_EINVAL;
        }
        SMX_CHECK_LAUNCH();
    }
#endif
    (void)lab;
    // instantiated (layout, class) pairs; other combinations run a sibling with the generic epilogue
#define PP_GO(AR, BR, E) { if (epi == E) p.tr_mode |= 128; pp_launch<AR, BR, E>(p, grid, stream); SMX_CHECK_LAUNCH(); }
    if (!p.a_rc && !p.b_rc) {
        if (epi == PP_EPI_ACT) PP_GO(false, false, PP_EPI_ACT)
        if (epi == PP_EPI_F32) PP_GO(false, false, PP_EPI_F32)
        PP_GO(false, false, PP_EPI_LINEAR)
    }
    if (!p.a_rc && p.b_rc && p.b.rows_per_batch > 0) {
#define PP_GOV(E) { if (epi == E) p.tr_mode |= 128; pp_launch<false, true, E, true>(p, grid, stream); SMX_CHECK_LAUNCH(); }
        if (epi == PP_EPI_ACTGRAD) PP_GOV(PP_EPI_ACTGRAD)
        PP_GOV(PP_EPI_LINEAR)
#undef PP_GOV
    }
    if (!p.a_rc && p.b_rc) {
        if (epi == PP_EPI_ACTGRAD) PP_GO(false, true, PP_EPI_ACTGRAD)
        if (epi == PP_EPI_F32) PP_GO(false, true, PP_EPI_F32)
        PP_GO(false, true, PP_EPI_LINEAR)
    }
    if (p.a_rc && p.b_rc) {
        if (p.b.rows_per_batch > 0 || p.a.rows_per_batch > 0) { if (epi == PP_EPI_F32) p.tr_mode |= 128; pp_launch<true, true, PP_EPI_F32, true>(p, grid, stream); SMX_CHECK_LAUNCH(); }
        PP_GO(true, true, PP_EPI_F32)
    }
    PP_GO(true, false, PP_EPI_F32)
#undef PP_GO
}

// Up to PP_MAXG weight-gradient problems (rows-contiguous operands, fp32 slab or plain fp32 output, plain views) in ONE
// persistent launch: their (K slice, tile) work lists are concatenated.  Declared in include/speechmix_hip.h.
int smx_gemm_group_fr(const SmxGemmGroup& grp, dim3 grid, hipStream_t stream);     // gemm_fr.hip

extern "C" int smx_gemm_group(const SmxGemmParams* probs, int count, int dtype, hipStream_t stream) {
    (void)hipGetLastError();
    if (!probs || count < 1 || count > PP_MAXG || dtype != SMX_BF16) return SMX_EINVAL;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;
    }
    SmxGemmGroup grp = {};
    grp.count = count;
    long long W = 0;
    for (int g = 0; g < count; ++g) {
        SmxGemmParams p = probs[g];
        if (p.nbatch < 1) p.nbatch = 1;
        if (p.split_k < 1) p.split_k = 1;
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.nbatch != 1 || !p.a_rc || !p.b_rc || !p.out_f32 || p.atomic == 1 ||
            (p.split_k > 1 && (p.atomic != 0 || p.split_stride <= 0)) || p.a.rows_per_batch > 0 || p.b.rows_per_batch > 0 ||
            p.c.rows_per_batch > 0 || pp_epi_class(p) != PP_EPI_F32 || (p.act & SMX_ACT_SAVE_GRAD))
            return SMX_EINVAL;
        const int kst = (p.K + BK - 1) / BK, per = (kst + p.split_k - 1) / p.split_k;
        const long long w = (long long)((p.M + PP_BM - 1) / PP_BM) * ((p.N + PP_BN - 1) / PP_BN) * p.split_k;
        if (w >= (1 << 22) || p.M >= (1 << 22) || p.N >= (1 << 22) || p.K >= (1 << 22) || (p.split_k - 1) * per >= kst) return SMX_EINVAL;
        p.tr_mode = 8 | 128;                 // the class-specialised epilogue (taken when the views are 16-B aligned)
        grp.prob[g] = p;
        grp.wstart[g] = (int)W;
        W += w;
    }
    if (W >= (1 << 22)) return SMX_EINVAL;
    for (int g = count; g <= PP_MAXG; ++g) grp.wstart[g] = (int)W;
    grp.W = (int)W;
    const int cap = (probs[0].tr_mode >> 16) & 0xfff;          // as in smx_gemm_pp
    const int wgs = cap > 0 && cap < ncu ? cap : ncu;
    dim3 grid((unsigned)(W < wgs ? W : wgs));
    if ((probs[0].tr_mode & 255) == 12) return smx_gemm_group_fr(grp, grid, stream);       // free-running schedule
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_pp_group_kernel<true, true, PP_EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        attr_done[dev & 15] = true;
    }
    hipLaunchKernelGGL((gemm_bf16_pp_group_kernel<true, true, PP_EPI_F32>), grid, dim3(512), PP_LDS_BYTES, stream, grp);
    SMX_CHECK_LAUNCH();
}

SMX_STEP_KEY_TU(gemm_pp)
