// ------------------------------------------------------------------------------------------------
// bf16 "wave-specialised" GEMM (tr_mode 14, round 6): the free-running kernel's 192 x 256 x 64 tile, LDS unit images, work
// list and register epilogues (gemm_pp.h) with the LDS-DMA issue taken OFF the MFMA waves.
//
// Why (profiles/r05_fr_timeline.txt, profiles/r05_probes_not_kept.txt): in the free-running kernel a K tile takes 1.18 us, 0.85 us
// without its LDS-DMA instructions (0.82 = the box's MFMA rate) - an in-order wave that issues a 1-KB `buffer_load ... lds`
// holds its own MFMA stream for 60 - 180 cycles, seven times per K tile, and the two waves of a SIMD reach their issue points
// together.  All waves of a workgroup share ONE register allocation, so loader waves beside 256-register compute waves do not
// exist; what fits is sixteen waves at 128 registers:
//
//   waves 0..11  compute: 3 x 4 wave tiles of 64 x 64 (64 accumulator registers, six A fragment slots + four B fragments),
//                three per SIMD; per K tile 32 MFMAs (16x16x32) and 16 fragment reads each, no VMEM instruction in the loop;
//   waves 12..15 loaders: one per SIMD, raised priority; each issues 14 (16 with a rows-contiguous A) of the K tile's 56 (64)
//                LDS-DMA instructions - two of the eight 1-KB pieces of every pass of every unit -, decodes the work list ONE ITEM AHEAD
//                (in two steps spread over two K tiles' slack), writes the item's record and fetches its bias slice into LDS
//                (four slots each): the compute waves read their item from LDS instead of decoding it themselves.
//
// Same flat unit stream as the other 256-wide kernels (two 64-KB stages, the stream runs across work items, so an item's first two
// K tiles are in flight / resident while the previous item's epilogue runs) and ONE workgroup barrier per K tile:
//
//   compute, tile t (stage t & 1):  k 0..31 | k 32..63, first column half | s_waitcnt lgkmcnt(0); s_barrier | second column half,
//                                   reading k 0..31 of tile t+1 from the other stage
//   loader:                         s_waitcnt vmcnt(0) (tile t+1 landed); s_barrier; issue tile t+2 into stage t & 1
//
// K order and epilogue arithmetic are those of the 128 x 128 kernel: bit-identical results (tests/test_gpu_r6.py).
// A unit images differ from the 8-wave forms: unit AH0 holds tile rows 0..127, AH1 rows 128..191 (one 64-row pass).
// ------------------------------------------------------------------------------------------------
#include "gemm_pp.h"

#define WS_MT 192
// SMX_WS_LAB (ablation builds, tools/lab/build_variant.sh; results are garbage, only the time means something): 4 = no LDS-DMA issued inside
// the K loop (the loader keeps its bookkeeping), 5 = the loader's barrier without the wait for its fills, 6 = 4 without the K tile's barrier,
// 7 = 4 without the fragment reads
#ifndef SMX_WS_LAB
#define SMX_WS_LAB 0
#endif
#define WS_NB 4
#define WS_CWAVES 12

// One operand's share of a loader wave: loader lw (0..3) stands in for the issue lanes of "virtual waves" 2 lw and 2 lw + 1 of the
// eight-wave unit layout (PPOperand, gemm_pp.h), i.e. pieces {2 lw, 2 lw + 1} of each pass.
template <bool RC, bool IS_A>
struct WSOperand {
    pp_rsrc_t rsrc;
    unsigned soff, sstep;
    unsigned vo[2][2][2];       // KC: [v][h][ps] byte offset of my (row, chunk);  RC: [v][0][ps] my k-row, [0][1][h] my columns

    static __device__ __forceinline__ int grow(int h, int hr) {          // unit-local row -> tile row (-1: not part of the unit)
        if (IS_A) { const int r = h * 128 + hr; return r < WS_MT ? r : -1; }
        return (hr >> 5) * 64 + h * 32 + (hr & 31);
    }
    __device__ __forceinline__ void init(const bf16_t* base, const SmxRowView& v, int row0, int nrows, int k0, int lane, int lw) {
        rsrc = pp_make_rsrc(base);
        // (byte offsets are formed modulo 2^32, as PPOperand's casts do: every operand is smaller than 2^31 bytes)
        const unsigned ld2 = (unsigned)v.ld * 2u;
        if (!RC) {
            soff = __builtin_amdgcn_readfirstlane((unsigned)k0 * 2u);
            sstep = BK * 2u;
            if (v.rows_per_batch <= 0) {
                // plain rows: a unit's rows are affine in (unit, pass) - ONE row offset per piece and lane, the rest are uniform strides
                // (this runs once per work item on the loader waves, inside a K tile's time)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int hr0 = (lw * 2 + w) * 8 + (lane >> 3);          // my row inside a pass; the swizzles do not depend on the pass
                    const int c = (lane & 7) ^ (IS_A ? ((hr0 >> 1) & 7) : pp_bswz(hr0));
                    const int g0 = IS_A ? hr0 : (hr0 >> 5) * 64 + (hr0 & 31);
                    const unsigned b0 = ((unsigned)v.off + (unsigned)(row0 + g0) * (unsigned)v.ld + (unsigned)c * 8u) * 2u;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const int dg = IS_A ? h * 128 + ps * 64 : ps * 128 + h * 32;
                            const bool in_unit = !IS_A || h * 128 + ps * 64 < WS_MT;
                            vo[w][h][ps] = (in_unit && row0 + g0 + dg < nrows) ? b0 + (unsigned)dg * ld2 : PP_OOB;
                        }
                }
                return;
            }
            const float rrpb = __builtin_amdgcn_rcpf((float)max(v.rows_per_batch, 1));
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int hr = ps * 64 + (lw * 2 + w) * 8 + (lane >> 3);
                        const int c = (lane & 7) ^ (IS_A ? ((hr >> 1) & 7) : pp_bswz(hr));
                        const int gr = grow(h, hr);
                        const int r = row0 + gr;
                        vo[w][h][ps] = (gr >= 0 && r < nrows) ? (unsigned)(pp_view_off(v, r, rrpb) + c * 8) * 2u : PP_OOB;
                    }
        } else {
            // (the chunk swizzles depend on the k-row through its bits 0-1 (= lane >> 4) and 3 (= lw & 1): one column part for both pieces)
            const int kl0 = lw * 8 + (lane >> 4), g16 = lane & 15;
            const int hc = IS_A ? ((((g16 >> 1) ^ rc_swz(kl0)) << 1) | (g16 & 1)) * 8 : (g16 ^ pp_rcb_swz(kl0)) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int gc = grow(h, hc);
                const int c = row0 + gc;
                vo[0][1][h] = (gc >= 0 && c < nrows) ? (unsigned)c * 2u : PP_OOB;
            }
            soff = __builtin_amdgcn_readfirstlane((unsigned)k0 * ld2);
            sstep = ld2 * BK;
            const unsigned b0 = (unsigned)v.off * 2u + (unsigned)kl0 * ld2;
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) vo[w][0][ps] = b0 + (unsigned)(ps * 32 + w * 4) * ld2;
        }
    }
    // unit H of the K tile whose first k is k0 -> LDS at byte address lds
    template <int H, bool DRY = false>
    __device__ __forceinline__ void issue(unsigned lds, int k0, int K, int lw, int lane) const {
        constexpr int NPS = (!RC && IS_A && H == 1) ? 1 : 2;          // the 64-row unit: one pass
        if (__builtin_expect(k0 + BK > K, 0)) {          // the K tile that crosses K (K % 64 != 0): a branch of its own, as in PPOperand
            asm volatile("" ::: "memory");
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int wu = lw * 2 + w;
                const int krow = wu * 4 + (lane >> 4);
                const int hr0 = wu * 8 + (lane >> 3);
                const int kc = ((lane & 7) ^ (IS_A ? ((hr0 >> 1) & 7) : pp_bswz(hr0))) * 8;
#pragma unroll
                for (int ps = 0; ps < NPS; ++ps) {
                    unsigned o = RC ? vo[w][0][ps] + vo[0][1][H] : vo[w][H][ps];
                    if (RC ? (k0 + ps * 32 + krow >= K) : (k0 + kc >= K)) o = PP_OOB;
                    if (!DRY) pp_dma16(rsrc, o, soff, lds + (unsigned)wu * 1024u + ps * 8192);
                }
            }
            return;
        }
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
            for (int w = 0; w < 2; ++w)
                if (!DRY) pp_dma16(rsrc, RC ? vo[w][0][ps] + vo[0][1][H] : vo[w][H][ps], soff, lds + (unsigned)(lw * 2 + w) * 1024u + ps * 8192);
    }
    __device__ __forceinline__ void advance() { soff += sstep; }
};

// LDS behind the two stages: four bias slices (256 floats each) and four item records, slot = (item count of the workgroup) & 3 -
// the loader prepares an item while the compute waves are at most two items behind it
#define WS_BIAS_OFF (2 * PP_STAGE)
#define WS_REC_OFF (WS_BIAS_OFF + 4 * 1024)
#define WS_LDS_BYTES (WS_REC_OFF + 4 * 64)
// item record (ints): 0 m0, 1 n0, 2 nk (0: the work list has ended), 4-5 zc, 6-7 ze, 8-9 zbias
struct WSRec { int m0, n0, nk, _pad; long long zc, ze, zbias; };

// The loader's side of the work list.  An item is PREPARED ahead of its first K tile in two steps that run in different K tiles' slack
// (the loader issues a tile in well under a K tile's time; decode + descriptor set-up in ONE tile stalled the barrier by ~0.9 us per item):
//   prep1 (right after the switch to the previous item): decode, item record + bias slice -> LDS;
//   prep2 (after that item's first K tile has been issued): the operand descriptors into the shadow copies (na, nb);
// `swap` makes the shadow current when the previous item's last K tile has been issued.
template <bool A_RC, bool B_RC>
struct WSLoader {
    WSOperand<A_RC, true> a, na;
    WSOperand<B_RC, false> b, nb;
    PPDiv dv;
    int q, qstep, kt, nk, k0, seq, K, live, pitems;
    int n_live, n_nk, n_k0, n_m0, n_n0, n_done;
    long long n_za, n_zb;
    unsigned lds0;

    __device__ __forceinline__ void prep1(int lane, int lw, char* smem) {
        n_live = q < dv.W ? 1 : 0;
        n_done = 0;
        ++pitems;
        WSRec* rec = reinterpret_cast<WSRec*>(smem + WS_REC_OFF + (pitems & 3) * 64);
        if (!n_live) {
            if (lw == 0 && lane == 0) rec->nk = 0;
            return;
        }
        const SmxGemmParams& p = pp_kernarg();
        PPItem it;
        pp_decode(p, dv, q, it);
        n_k0 = it.ks0 * BK;
        n_nk = it.nk;
        n_m0 = it.m0; n_n0 = it.n0; n_za = it.za; n_zb = it.zb;
        if (lw == 0 && lane == 0) {
            rec->m0 = it.m0; rec->n0 = it.n0; rec->nk = it.nk;
            rec->zc = it.zc; rec->ze = it.ze; rec->zbias = it.zbias;
        }
        const float* const pbias = p.bias;
        const int pN = p.N;
        if (pbias) {
            pp_rsrc_t br = pp_make_rsrc(pbias + it.zbias + it.n0);
            br[2] = max(pN - it.n0, 0) * 4;
            pp_dma4(br, (unsigned)(lw * 64 + lane) * 4u, lds0 + WS_BIAS_OFF + (pitems & 3) * 1024 + lw * 256);
        }
    }
    __device__ __forceinline__ void prep2(int lane, int lw) {
        n_done = 1;
        if (!n_live) return;
        const SmxGemmParams& p = pp_kernarg();
        const SmxRowView va = p.a, vb = p.b;
        const bf16_t* const pa = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* const pb = reinterpret_cast<const bf16_t*>(p.B);
        const int pM = p.M, pN = p.N;
        na.init(pa + n_za, va, n_m0, pM, n_k0, lane, lw);
        nb.init(pb + n_zb, vb, n_n0, pN, n_k0, lane, lw);
    }
    __device__ __forceinline__ void swap(int lane, int lw) {
        if (!n_done) prep2(lane, lw);
        live = n_live;
        if (!live) return;
        a = na; b = nb;
        k0 = n_k0; nk = n_nk; kt = 0;
        q += qstep;
    }
    template <bool DRY = false>
    __device__ __forceinline__ void issue_tile(int lane, int lw, char* smem) {
        const unsigned st = lds0 + (unsigned)(seq & 1) * PP_STAGE;
        a.template issue<0, DRY>(st + 0 * PP_UNIT, k0, K, lw, lane);
        b.template issue<0, DRY>(st + 2 * PP_UNIT, k0, K, lw, lane);
        b.template issue<1, DRY>(st + 3 * PP_UNIT, k0, K, lw, lane);
        a.template issue<1, DRY>(st + 1 * PP_UNIT, k0, K, lw, lane);
        ++seq;
        if (++kt == nk) {
            swap(lane, lw);
            prep1(lane, lw, smem);
        } else {
            k0 += BK;
            a.advance();
            b.advance();
            if (!n_done) prep2(lane, lw);
        }
    }
};

// Fragment registers of a compute wave: six A slots of one 16-row block each (four in use + two spares) and the two B column
// halves (two permuted 16-column blocks each); a slot is re-read for the next half step as soon as the last MFMA that uses it
// has been issued (the free-running kernel's scheme, gemm_fr.hip).
struct WSFrags {
    bf16x8_t a[WS_NB + 2];
    bf16x8_t b[2][2];
};
template <bool A_RC>
__device__ __forceinline__ bf16x8_t ws_afrag(const char* stage, int blk, int kk, int wr, int lane) {
    return load_frag<A_RC>(stage + (wr >> 1) * PP_UNIT, (wr & 1) * 64 + blk * 16, kk, lane, 1);
}
template <bool B_RC>
__device__ __forceinline__ bf16x8_t ws_bfrag(const char* stage, int ch, int j, int kk, int wc, int lane) {
    return pp_bfrag<B_RC>(stage + (2 + ch) * PP_UNIT, wc * 32, j, kk, lane);
}
template <int PAR>
__device__ __forceinline__ constexpr int ws_slot(int blk) { return (PAR && blk >= WS_NB - 2) ? blk + 2 : blk; }

// One half step (32 of k) = two passes of 8 MFMAs: pass CH multiplies the four A blocks with column half CH.
//   pass 0: at its start the B fragments of column half 1 of THIS half step are read;
//   pass 1: the fragments of the NEXT half step (k sub-step 1 - PAR of `nstage`): column half 0 and the two spare-slot A blocks at its
//           start, A blocks 0 and 1 each right after the last MFMA on its slot.
template <bool A_RC, bool B_RC, int PAR, int CH>
__device__ __forceinline__ void ws_pass(f32x4_t (&acc)[WS_NB][4], WSFrags& f, const char* stage, const char* nstage, int lane, int wr, int wc) {
    constexpr bool RD = SMX_WS_LAB != 7;
    if constexpr (!RD) {
    } else if constexpr (CH == 0) {
        f.b[1][0] = ws_bfrag<B_RC>(stage, 1, 0, PAR, wc, lane);
        f.b[1][1] = ws_bfrag<B_RC>(stage, 1, 1, PAR, wc, lane);
    } else {
        f.b[0][0] = ws_bfrag<B_RC>(nstage, 0, 0, 1 - PAR, wc, lane);
        f.b[0][1] = ws_bfrag<B_RC>(nstage, 0, 1, 1 - PAR, wc, lane);
        f.a[ws_slot<1 - PAR>(WS_NB - 2)] = ws_afrag<A_RC>(nstage, WS_NB - 2, 1 - PAR, wr, lane);
        f.a[ws_slot<1 - PAR>(WS_NB - 1)] = ws_afrag<A_RC>(nstage, WS_NB - 1, 1 - PAR, wr, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < WS_NB; ++g) {
        const int sl = ws_slot<PAR>(g);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            acc[g][CH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[CH][j], f.a[sl], acc[g][CH * 2 + j], 0, 0, 0);
        if constexpr (CH == 1 && RD) {
            if (g < WS_NB - 2) f.a[g] = ws_afrag<A_RC>(nstage, g, 1 - PAR, wr, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool A_RC, bool B_RC>
__device__ __forceinline__ void ws_read_first(WSFrags& f, const char* stage, int wr, int wc, int lane) {
    f.b[0][0] = ws_bfrag<B_RC>(stage, 0, 0, 0, wc, lane);
    f.b[0][1] = ws_bfrag<B_RC>(stage, 0, 1, 0, wc, lane);
#pragma unroll
    for (int g = 0; g < WS_NB; ++g) f.a[g] = ws_afrag<A_RC>(stage, g, 0, wr, lane);
}

#if SMX_WS_LAB == 6
#define WS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WS_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

template <bool A_RC, bool B_RC, int EPI>
__global__ __launch_bounds__(1024) void gemm_bf16_ws_kernel(SmxGemmParams p_) {
    const SmxGemmParams& p = pp_kernarg();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntm = (p.M + WS_MT - 1) / WS_MT, ntn = (p.N + PP_BN - 1) / PP_BN;
    PP_STAMP(0);

    if (wave >= WS_CWAVES) {
        // ---- loader waves ----
#ifndef SMX_WS_PRIO
#define SMX_WS_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(SMX_WS_PRIO);
        const int lw = wave - WS_CWAVES;
        WSLoader<A_RC, B_RC> ld;
        ld.dv.init(p, ntm, ntn, WS_MT);
        ld.q = blockIdx.x; ld.qstep = gridDim.x;
        ld.seq = 0; ld.kt = 0; ld.nk = 0; ld.k0 = 0; ld.pitems = 0;
        ld.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
        ld.K = p.K;
        ld.prep1(lane, lw, smem);          // the workgroup's first item (the grid never exceeds the work list)
        ld.swap(lane, lw);
        // its successor's record / bias slice go out behind the first K tile's fills (a one-tile item: ahead of them, the switch follows at once)
        const bool late = ld.nk > 1;
        if (late) { ld.n_done = 1; ld.n_live = 0; }
        else ld.prep1(lane, lw, smem);
        int issued = 0;
        if (ld.live) { ld.issue_tile(lane, lw, smem); ++issued; }
        if (late) ld.prep1(lane, lw, smem);
        if (ld.live) {
            ld.issue_tile(lane, lw, smem); ++issued;
            // K tile 0 (and the bias slices) landed, tile 1 on its way; the item records are written
            if constexpr (A_RC) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < issued; ++s) {
            // stream tile s + 1 landed (it had a K tile's time); behind the barrier every read of tile s is done: its stage takes tile s + 2
#if SMX_WS_LAB == 6
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif SMX_WS_LAB == 5
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            if (ld.live) { ld.template issue_tile<SMX_WS_LAB == 4 || SMX_WS_LAB == 6 || SMX_WS_LAB == 7>(lane, lw, smem); ++issued; }
        }
        return;
    }

    // ---- compute waves ----
    const int wr = wave >> 2, wc = wave & 3;
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(1);

    f32x4_t acc[WS_NB][4];
    WSFrags f;
    int seq = 0, items = 0;
    const bool fast_epi = (p.tr_mode & 128) && pp_views_aligned(p);
    for (;;) {
        // the item as the loader decoded it (record slot items & 3; written at least one barrier ago)
        ++items;
        PPItem it;
        {
            const WSRec* rec = reinterpret_cast<const WSRec*>(smem + WS_REC_OFF + (items & 3) * 64);
            const int4 r0 = *reinterpret_cast<const int4*>(rec);
            const int4 r1 = *reinterpret_cast<const int4*>(reinterpret_cast<const char*>(rec) + 16);
            const int2 r2 = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(rec) + 32);
            it.m0 = __builtin_amdgcn_readfirstlane(r0.x); it.n0 = __builtin_amdgcn_readfirstlane(r0.y); it.nk = __builtin_amdgcn_readfirstlane(r0.z);
            it.zc = ((long long)__builtin_amdgcn_readfirstlane(r1.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r1.x);
            it.ze = ((long long)__builtin_amdgcn_readfirstlane(r1.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r1.z);
            it.zbias = ((long long)__builtin_amdgcn_readfirstlane(r2.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(r2.x);
        }
        if (it.nk == 0) break;
#pragma unroll
        for (int a = 0; a < WS_NB; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        PP_STAMP(3 * items - 1);
        // the item's K tile 0 is resident: prologue barrier, or the barrier inside the previous item's last K tile
        ws_read_first<A_RC, B_RC>(f, smem + (seq & 1) * PP_STAGE, wr, wc, lane);
        __builtin_amdgcn_sched_barrier(0);
        for (int t = 0; t < it.nk; ++t) {
            const char* cur = smem + (seq & 1) * PP_STAGE;
            const char* oth = smem + ((seq & 1) ^ 1) * PP_STAGE;
            ws_pass<A_RC, B_RC, 0, 0>(acc, f, cur, cur, lane, wr, wc);
            ws_pass<A_RC, B_RC, 0, 1>(acc, f, cur, cur, lane, wr, wc);
            ws_pass<A_RC, B_RC, 1, 0>(acc, f, cur, oth, lane, wr, wc);
            WS_SYNC();
            // (in the item's last tile these reads fetch the next item's first fragments and are dropped: one copy of the pass)
            ws_pass<A_RC, B_RC, 1, 1>(acc, f, cur, oth, lane, wr, wc);
#if SMX_FR_TRACE
            if (items == (gridDim.x * 2 <= ntm * ntn * p.nbatch * p.split_k ? 2 : 1) && t < 16) PP_STAMP(45 + t);
#endif
            ++seq;
        }
        PP_STAMP(3 * items);
        if (fast_epi) {
            pp_epilogue_fast<EPI, false, WS_NB, B_RC>(acc, it.m0 + wr * 64, it.n0 + wc * 64, it.n0, smem + WS_BIAS_OFF + (items & 3) * 1024, it.zc,
                                                      it.ze, lane, 0);
        } else {
            pp_epilogue<false, WS_NB, B_RC>(acc, it.m0 + wr * 64, it.n0 + wc * 64, it.zc, it.zbias, it.ze, lane, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP(3 * items + 1);
    }
}

template <bool A_RC, bool B_RC, int EPI>
static void ws_launch(const SmxGemmParams& p, dim3 grid, hipStream_t stream) {
    static bool attr_done[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 15]) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_ws_kernel<A_RC, B_RC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
        attr_done[dev & 15] = true;
    }
    hipLaunchKernelGGL((gemm_bf16_ws_kernel<A_RC, B_RC, EPI>), grid, dim3(1024), WS_LDS_BYTES, stream, p);
}

// tr_mode 14: 192 x 256 tiles, twelve compute + four loader waves.  Instantiated for the layouts / classes of the free-running kernel;
// one-tile items included (four bias / record slots: the loader is at most two items ahead); batched views of rows-contiguous operands are refused
// (the tuner never offers them).
int smx_gemm_ws(const SmxGemmParams& pin, hipStream_t stream) {
    SmxGemmParams p = pin;
    if (!pp_saved_ok(p)) return SMX_EINVAL;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;
    }
    const long long W = (long long)((p.M + WS_MT - 1) / WS_MT) * ((p.N + PP_BN - 1) / PP_BN) * p.nbatch * p.split_k;
    const int kst = (p.K + BK - 1) / BK, per = (kst + p.split_k - 1) / p.split_k;
    if (W >= (1 << 22) || p.M >= (1 << 22) || p.N >= (1 << 22) || p.K >= (1 << 22) || p.atomic == 1 || (p.split_k - 1) * per >= kst ||
        ((p.K & 7) && !(p.a_rc && p.b_rc))) return SMX_EINVAL;
    if ((p.a.rows_per_batch > 0 && p.a_rc) || (p.b.rows_per_batch > 0 && p.b_rc) || (p.a_rc && !p.b_rc)) return SMX_EINVAL;
    const int cap = (p.tr_mode >> 16) & 0xfff;
    const int wgs = cap > 0 && cap < ncu ? cap : ncu;
    dim3 grid((unsigned)(W < wgs ? W : wgs));
    const int epi = pp_epi_class(p);
    p.tr_mode = 8;
#define WS_GO(AR, BR, E) { if (epi == E) p.tr_mode |= 128; ws_launch<AR, BR, E>(p, grid, stream); SMX_CHECK_LAUNCH(); }
    if (!p.a_rc && !p.b_rc) {
        if (epi == PP_EPI_ACT) WS_GO(false, false, PP_EPI_ACT)
        if (epi == PP_EPI_ACTGRAD) WS_GO(false, false, PP_EPI_ACTGRAD)
        if (epi == PP_EPI_F32) WS_GO(false, false, PP_EPI_F32)
        WS_GO(false, false, PP_EPI_LINEAR)
    }
    if (!p.a_rc && p.b_rc) {
        if (epi == PP_EPI_ACTGRAD) WS_GO(false, true, PP_EPI_ACTGRAD)
        if (epi == PP_EPI_F32) WS_GO(false, true, PP_EPI_F32)
        WS_GO(false, true, PP_EPI_LINEAR)
    }
    if (p.a_rc && p.b_rc) WS_GO(true, true, PP_EPI_F32)
    return SMX_EINVAL;
#undef WS_GO
}

SMX_STEP_KEY_TU(gemm_ws)
