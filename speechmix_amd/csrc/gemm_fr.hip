// ------------------------------------------------------------------------------------------------
// bf16 "free-running" GEMM (tr_mode 12): the ping-pong kernel's tile (256 x 256 x 64, 8 waves as 2 x 4, wave tile 128 x 64,
// one persistent workgroup per CU), LDS unit images, flat LDS-DMA unit stream, work list and register epilogues (gemm_pp.h),
// under a different schedule.
//
// Why (tools/lab/cu_lab.hip, round 3): one wave reads LDS at no more than ~33-40 B/clk (a CU: 224 B/clk from 8 waves) and
// issues at most one 1-KB LDS-DMA per ~50 clk, while LDS reads and LDS-DMA writes do not slow each other down (230 + 64
// B/clk/CU together).  The ping-pong schedule concentrates a wave's 24 fragment reads and 8 DMA issues per K tile in its
// four load segments (12 KB in the first one: ~370 clk at the per-wave rate against the partner's 256-clk MFMA segment)
// and pays 16 barriers per K tile.  Here every wave runs ONE stream in which the fragment reads of the next 32-deep half
// step and the LDS-DMA of the K tile after next are spread between the MFMAs of the current half step (12 B/clk per wave
// on average), the two waves of a SIMD fill each other's issue stalls, and the workgroup meets at ONE barrier per K tile:
//
//   tile t (stage t & 1):  H0  32 MFMAs (k 0..31)  | reads: tile t, k 32..63
//                          SYNC  s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier     -> tile t+1 landed for everyone, stage t & 1 free
//                          H1  32 MFMAs (k 32..63) | reads: tile t+1, k 0..31 | LDS-DMA: tile t+2 -> stage t & 1
//
// The unit stream runs across work items exactly as in the ping-pong kernel, so the first two K tiles of the next item
// are in flight / resident while the epilogue runs.
// ------------------------------------------------------------------------------------------------
#include "gemm_pp.h"

// Fragment registers: ten A slots of one 16-row block each (eight in use + two spares) and the two B column halves
// (two permuted 16-column blocks each).  Nothing is double-buffered wholesale: a slot is re-read for the next half step
// as soon as the last MFMA that uses it has been issued.
template <int NB>
struct FRFrags {
    bf16x8_t a[NB + 2];
    bf16x8_t b[2][2];
};

// A block blk = rh * 4 + a of my wave-row group (NB = 8: 128 rows, NB = 6: 96 rows - the second unit then holds 2 x 32 rows)
template <bool A_RC, int NB>
__device__ __forceinline__ bf16x8_t fr_afrag(const char* stage, int blk, int kk, int wr, int lane) {
    return load_frag<A_RC>(stage + (blk >> 2) * PP_UNIT, wr * ((blk >> 2) && NB == 6 ? 32 : 64) + (blk & 3) * 16, kk, lane, 1);
}
template <bool B_RC>
__device__ __forceinline__ bf16x8_t fr_bfrag(const char* stage, int ch, int j, int kk, int wc, int lane) {
    return pp_bfrag<B_RC>(stage + (2 + ch) * PP_UNIT, wc * 32, j, kk, lane);
}
// slot of A block `blk` in a half step of parity PAR (0: k 0..31, 1: k 32..63): the last two blocks alternate between their own
// slots and the two spare ones
template <int PAR, int NB>
__device__ __forceinline__ constexpr int fr_slot(int blk) { return (PAR && blk >= NB - 2) ? blk + 2 : blk; }

// One half step = two passes of 16 MFMAs: pass CH multiplies the eight A blocks with column half CH.
//   pass 0:  at its start the B fragments of column half 1 of THIS half step are read (their registers were in use until the
//            end of the previous pass); ISSUE0: two units of LDS-DMA.
//   pass 1:  the fragments of the NEXT half step (k sub-step 1 - PAR of stage `nstage`) are read: column half 0 and the two
//            spare-slot A blocks at its start, A blocks 0..5 each right after the last MFMA on its slot; ISSUE1: two units.
// sched_barrier(0) after every A block pins the interleave.
// SMX_FR_LAB (ablation builds, tools/lab/build_variant.sh; results are garbage, only the time means something): 1 = the K loop without
// its fragment reads (registers keep the item's first fragments), 2 = without its MFMAs, 3 = every LDS-DMA fill re-reads the item's first
// K tile (cache hits), 4 = no fills inside the K loop.
#ifndef SMX_FR_LAB
#define SMX_FR_LAB 0
#endif
template <bool A_RC, bool B_RC, int PAR, int CH, bool READ_, int ISSUE, int NB, class ISSUER>
__device__ __forceinline__ void fr_pass(f32x4_t (&acc)[NB][4], FRFrags<NB>& f, const char* stage, const char* nstage, ISSUER& is, int tid,
                                        int lane, int wr, int wc) {
    constexpr bool READ = READ_ && SMX_FR_LAB != 1;
    if constexpr (CH == 0 && SMX_FR_LAB != 1) {
        f.b[1][0] = fr_bfrag<B_RC>(stage, 1, 0, PAR, wc, lane);
        f.b[1][1] = fr_bfrag<B_RC>(stage, 1, 1, PAR, wc, lane);
    } else if constexpr (READ) {
        f.b[0][0] = fr_bfrag<B_RC>(nstage, 0, 0, 1 - PAR, wc, lane);
        f.b[0][1] = fr_bfrag<B_RC>(nstage, 0, 1, 1 - PAR, wc, lane);
        f.a[fr_slot<1 - PAR, NB>(NB - 2)] = fr_afrag<A_RC, NB>(nstage, NB - 2, 1 - PAR, wr, lane);
        f.a[fr_slot<1 - PAR, NB>(NB - 1)] = fr_afrag<A_RC, NB>(nstage, NB - 1, 1 - PAR, wr, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int sl = fr_slot<PAR, NB>(g);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (SMX_FR_LAB != 2) acc[g][CH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[CH][j], f.a[sl], acc[g][CH * 2 + j], 0, 0, 0);
            else asm volatile("" : "+v"(f.a[sl]), "+v"(f.b[CH][j]));            // (keeps the fragment reads alive)
        }
        if constexpr (CH == 1 && READ) {
            if (g < NB - 2) f.a[g] = fr_afrag<A_RC, NB>(nstage, g, 1 - PAR, wr, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ISSUE != 0 && SMX_FR_LAB != 4) {
            constexpr bool FZ = SMX_FR_LAB == 3;
            if (g == 1) { if (ISSUE == 1) is.template issue<0, FZ>(tid); else is.template issue<2, FZ>(tid); __builtin_amdgcn_sched_barrier(0); }
            if (g == NB - 3) { if (ISSUE == 1) is.template issue<1, FZ>(tid); else is.template issue<3, FZ>(tid); __builtin_amdgcn_sched_barrier(0); }
        }
    }
}

// first half step of an item: nothing was prefetched across the epilogue
template <bool A_RC, bool B_RC, int NB>
__device__ __forceinline__ void fr_read_first(FRFrags<NB>& f, const char* stage, int wr, int wc, int lane) {
    f.b[0][0] = fr_bfrag<B_RC>(stage, 0, 0, 0, wc, lane);
    f.b[0][1] = fr_bfrag<B_RC>(stage, 0, 1, 0, wc, lane);
#pragma unroll
    for (int g = 0; g < NB; ++g) f.a[g] = fr_afrag<A_RC, NB>(stage, g, 0, wr, lane);
}

// SMX_FR_TRACE (lab builds, tools/gpu_fr_timeline.py; PP_STAMP in gemm_pp.h): wave 0 of every workgroup stamps the 100-MHz clock at kernel
// entry, after the prologue and, per work item, at its start, after its K loop and after its epilogue (stores issued)
#define FR_STAMP(i) PP_STAMP(i)

#if SMX_FR_LAB == 5          // (timing only: the K tile's barrier without the wait for the fills)
#define FR_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FR_SYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

template <bool A_RC, bool B_RC, int EPI, bool BVIEW, bool GRP, int MT>
__device__ __forceinline__ void fr_kernel_body() {
    constexpr int NB = MT / 32;          // 16-row A blocks per wave (wave tile MT/2 x 64)
    static_assert(!GRP || !BVIEW, "grouped launches: plain operand views");
    const SmxGemmParams& p = pp_kernarg_g<GRP>(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int ntm = (p.M + MT - 1) / MT, ntn = (p.N + PP_BN - 1) / PP_BN;
    int W = ntm * ntn * p.nbatch * p.split_k;
    if constexpr (GRP) W = pp_group().W;

    FR_STAMP(0);
    PPIssue<A_RC, B_RC, BVIEW, GRP, MT> is;
    is.dv.init(p, ntm, ntn, MT);
    is.g = 0;
    is.q = blockIdx.x; is.qstep = gridDim.x;
    is.seq = 0; is.kt = 0; is.nk = 0; is.k0 = 0; is.ahead = 0;
    is.wave_u = __builtin_amdgcn_readfirstlane(wave);
    is.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    is.K = p.K;
    is.load_item(tid);

    // prologue: K tile 0 of the stream entirely and the first two units of tile 1 (EARLY: all of tile 1); tile 0 is resident after the barrier
    constexpr bool EARLY = MT == 192 && SMX_FR_LAB != 4;          // see below the K loop
    {
        bool all = true;
        all &= is.template issue<0>(tid);
        all &= is.template issue<1>(tid);
        all &= is.template issue<2>(tid);
        all &= is.template issue<3>(tid);
        all &= is.template issue<0>(tid);
        all &= is.template issue<1>(tid);
        if constexpr (EARLY) {
            all &= is.template issue<2>(tid);
            all &= is.template issue<3>(tid);
            is.ahead = 2;
            if (!all) PP_WAITV(0);
            else if constexpr (!A_RC && MT == 192) PP_WAITV(7);          // (the 64-row A unit is one instruction)
            else PP_WAITV(8);
        } else {
            if (all) PP_WAITV(4);
            else PP_WAITV(0);
        }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    FR_STAMP(1);

    f32x4_t acc[NB][4];
    FRFrags<NB> f;
    int seq = 0, items = 0;
    bool fast_epi = (p.tr_mode & 128) && pp_views_aligned(p);
    int gc = 0;                 // GRP: problem of the item being computed
    for (int q = blockIdx.x; q < W; q += gridDim.x) {
        PPItem it;
        if constexpr (GRP) {
            while (q >= pp_group().wstart[gc + 1]) ++gc;
            const SmxGemmParams& pg = pp_kernarg_g<true>(gc);
            PPDiv d;
            d.init(pg, (pg.M + MT - 1) / MT, (pg.N + PP_BN - 1) / PP_BN, MT);
            pp_decode(pg, d, q - pp_group().wstart[gc], it);
            fast_epi = (pg.tr_mode & 128) && pp_views_aligned(pg);
        } else {
            pp_decode(pp_kernarg(), is.dv, q, it);
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        ++items;
        FR_STAMP(3 * items - 1);
        if (fast_epi && wr == 0) {          // bias slice of this item -> LDS (retired by the item's first SYNC)
            const SmxGemmParams& pq = pp_kernarg_g<GRP>(gc);
            if (pq.bias) {
                pp_rsrc_t br = pp_make_rsrc(pq.bias + it.zbias + it.n0);
                br[2] = max(pq.N - it.n0, 0) * 4;
                pp_dma4(br, (unsigned)(wc * 64 + lane) * 4u, is.lds0 + PP_BIAS_OFF + (items & 1) * 1024 + wc * 256);
            }
        }
        // the item's K tile 0 is resident: prologue barrier, or the SYNC inside the previous item's last half step
        fr_read_first<A_RC, B_RC, NB>(f, smem + (seq & 1) * PP_STAGE, wr, wc, lane);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < it.nk; ++t) {
            const char* cur = smem + (seq & 1) * PP_STAGE;
            const char* oth = smem + ((seq & 1) ^ 1) * PP_STAGE;
            // H0 (k 0..31 of tile t): pass 0 issues the last two units of stream tile seq+1 (in an item's first tile they are on their way
            // already - is.ahead, see below the loop - and the two calls return at once), pass 1 reads k 32..63 of tile t
            fr_pass<A_RC, B_RC, 0, 0, true, 2, NB>(acc, f, cur, cur, is, tid, lane, wr, wc);
            fr_pass<A_RC, B_RC, 0, 1, true, 0, NB>(acc, f, cur, cur, is, tid, lane, wr, wc);
            // H1 (k 32..63): pass 0; SYNC: stream tile seq+1 resident for everyone, every read of tile t done -> its stage is free;
            // pass 1 reads k 0..31 of the item's next tile and issues the first two units of stream tile seq+2 into this stage
            fr_pass<A_RC, B_RC, 1, 0, true, 0, NB>(acc, f, cur, oth, is, tid, lane, wr, wc);
            FR_SYNC();
            // (in the item's last tile these reads fetch the next item's first fragments and are dropped: one copy of the pass)
            fr_pass<A_RC, B_RC, 1, 1, true, 1, NB>(acc, f, cur, oth, is, tid, lane, wr, wc);
#if SMX_FR_TRACE
            if (items == (gridDim.x * 2 <= W ? 2 : 1) && t < 16) FR_STAMP(45 + t);          // (per K tile, one item of every workgroup)
#endif
            ++seq;
        }
        // EARLY (192-row tiles; the 256-row forms spill with it): the last two units of the NEXT item's second K tile go out here, ahead of the
        // epilogue, instead of in that item's first half step: they are first touches of new operand rows (HBM latency) and the item's first
        // barrier used to wait for them ~0.5 us (round 5 timeline).  Their stage is free: nothing reads this item's last stage after the
        // barrier inside its last tile.
        if constexpr (EARLY) {
            is.template issue<2, SMX_FR_LAB == 3>(tid);
            is.template issue<3, SMX_FR_LAB == 3>(tid);
            is.ahead = 2;          // the next two calls for these kinds (the next item's first pass) are answered by these
            __builtin_amdgcn_sched_barrier(0);
        }
        FR_STAMP(3 * items);
        if (fast_epi) {
            pp_epilogue_fast<EPI, GRP, NB, B_RC>(acc, it.m0 + wr * (MT / 2), it.n0 + wc * 64, it.n0, smem + PP_BIAS_OFF + (items & 1) * 1024, it.zc,
                                       it.ze, lane, gc);
        } else {
            pp_epilogue<GRP, NB, B_RC>(acc, it.m0 + wr * (MT / 2), it.n0 + wc * 64, it.zc, it.zbias, it.ze, lane, gc);
        }
        __builtin_amdgcn_sched_barrier(0);
        FR_STAMP(3 * items + 1);
    }
}

template <bool A_RC, bool B_RC, int EPI, bool BVIEW, int MT>
__global__ __launch_bounds__(512) void gemm_bf16_fr_kernel(SmxGemmParams p) {
    fr_kernel_body<A_RC, B_RC, EPI, BVIEW, false, MT>();
}
// grouped form (smx_gemm_group): the same body over the concatenated work lists of up to PP_MAXG problems
template <bool A_RC, bool B_RC, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_fr_group_kernel(SmxGemmGroup grp) {
    fr_kernel_body<A_RC, B_RC, EPI, false, true, PP_BM>();
}

template <bool A_RC, bool B_RC, int EPI, int MT>
static void fr_launch(const SmxGemmParams& p, dim3 grid, hipStream_t stream) {
    constexpr bool BVIEW = false;
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_fr_kernel<A_RC, B_RC, EPI, BVIEW, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        attr_done[dev & 15] = true;
    }
    hipLaunchKernelGGL((gemm_bf16_fr_kernel<A_RC, B_RC, EPI, BVIEW, MT>), grid, dim3(512), PP_LDS_BYTES, stream, p);
}

int smx_gemm_pp(const SmxGemmParams& p, hipStream_t stream);   // gemm_pp.hip

// tr_mode 12: 256 x 256 tiles; 13: 192 x 256 tiles (N = 768 / 2304 at 16 k rows: 252 / 756 items on 256 CUs instead of 189 / 567)
int smx_gemm_fr(const SmxGemmParams& pin, hipStream_t stream) {
    SmxGemmParams p = pin;
    const int mt = (p.tr_mode & 255) == 13 ? 192 : PP_BM;
    if (!pp_saved_ok(p)) return SMX_EINVAL;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;
    }
    const long long W = (long long)((p.M + mt - 1) / mt) * ((p.N + PP_BN - 1) / PP_BN) * p.nbatch * p.split_k;
    const int kst = (p.K + BK - 1) / BK, per = (kst + p.split_k - 1) / p.split_k;
    if (W >= (1 << 22) || p.M >= (1 << 22) || p.N >= (1 << 22) || p.K >= (1 << 22) || p.atomic == 1 || (p.split_k - 1) * per >= kst ||
        ((p.K & 7) && !(p.a_rc && p.b_rc))) return SMX_EINVAL;
    // batched views of rows-contiguous operands and the (RC, KC) layout: instantiated for the ping-pong schedule only
    if ((p.a.rows_per_batch > 0 && p.a_rc) || (p.b.rows_per_batch > 0 && p.b_rc) || (p.a_rc && !p.b_rc)) return smx_gemm_pp(pin, stream);
    const int cap = (p.tr_mode >> 16) & 0xfff;
    const int wgs = cap > 0 && cap < ncu ? cap : ncu;
    dim3 grid((unsigned)(W < wgs ? W : wgs));
    const int epi = pp_epi_class(p);
    p.tr_mode = 8;
#define FR_GO(AR, BR, E) { if (epi == E) p.tr_mode |= 128; if (mt == 192) fr_launch<AR, BR, E, 192>(p, grid, stream); else fr_launch<AR, BR, E, PP_BM>(p, grid, stream); SMX_CHECK_LAUNCH(); }
    if (!p.a_rc && !p.b_rc) {
        if (epi == PP_EPI_ACT) FR_GO(false, false, PP_EPI_ACT)
        if (epi == PP_EPI_ACTGRAD) FR_GO(false, false, PP_EPI_ACTGRAD)      // (round 4: conv data gradients through transposed taps)
        if (epi == PP_EPI_F32) FR_GO(false, false, PP_EPI_F32)
        FR_GO(false, false, PP_EPI_LINEAR)
    }
    if (!p.a_rc && p.b_rc) {
        if (epi == PP_EPI_ACTGRAD) FR_GO(false, true, PP_EPI_ACTGRAD)
        if (epi == PP_EPI_F32) FR_GO(false, true, PP_EPI_F32)
        FR_GO(false, true, PP_EPI_LINEAR)
    }
    if (p.a_rc && p.b_rc) FR_GO(true, true, PP_EPI_F32)
    return SMX_EINVAL;
#undef FR_GO
}

// grouped weight gradients on the free-running schedule (called by smx_gemm_group, gemm_pp.hip, when the first problem asks for tr_mode 12)
int smx_gemm_group_fr(const SmxGemmGroup& grp, dim3 grid, hipStream_t stream) {
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_fr_group_kernel<true, true, PP_EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES);
        attr_done[dev & 15] = true;
    }
    hipLaunchKernelGGL((gemm_bf16_fr_group_kernel<true, true, PP_EPI_F32>), grid, dim3(512), PP_LDS_BYTES, stream, grp);
    SMX_CHECK_LAUNCH();
}

SMX_STEP_KEY_TU(gemm_fr)
