// Shared pieces of the 256-wide persistent GEMM kernels (gemm_pp.hip: ping-pong schedule; gemm_fr.hip: free-running
// schedule): LDS unit images, the flat LDS-DMA unit stream, work-list decode, fragment reads and the register epilogues.
#pragma once
#include "gemm_common.h"

#define PP_BM 256
#define PP_BN 256
#define PP_UNIT 16384
#define PP_STAGE (4 * PP_UNIT)               // AH0 AH1 BH0 BH1
#define PP_BIAS_OFF (2 * PP_STAGE)          // 2 x 256 floats: the bias slice of the current / next work item
#define PP_LDS_BYTES (2 * PP_STAGE + 2048)
#define PP_GROUP 4
#define PP_OOB 0x80000000u                   // = num_records of the operand descriptors: any offset >= it reads zeros

#define PP_WAITV(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

typedef __attribute__((ext_vector_type(4))) int pp_rsrc_t;

// SMX_FR_TRACE (lab builds of gemm_fr.hip, tools/gpu_fr_timeline.py): thread 0 of every workgroup stamps the 100-MHz clock into slot i
#ifndef SMX_FR_TRACE
#define SMX_FR_TRACE 0
#endif
#if SMX_FR_TRACE
__device__ unsigned long long smx_fr_trace_buf[256 * 64];
extern "C" int smx_fr_trace_read(void* host, unsigned long long bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(smx_fr_trace_buf), bytes) == hipSuccess ? SMX_OK : -5;
}
#define PP_STAMP(i) do { if (threadIdx.x == 0 && (i) < 64 && blockIdx.x < 256) smx_fr_trace_buf[blockIdx.x * 64 + (i)] = wall_clock64(); } while (0)
#else
#define PP_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ pp_rsrc_t pp_make_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    pp_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));      // stride 0: raw buffer
    r[2] = (int)PP_OOB;
    r[3] = 0x00020000;
    return r;
}
// 16 B per lane: LDS[m0 + lane * 16] = mem[rsrc.base + soff + voff]  (zeros when voff >= num_records)
__device__ __forceinline__ void pp_dma16(pp_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
    // (the descriptor is uniform by construction; said again here because a descriptor the register allocator parked in vector registers
    // reaches the "s" operand as a vector register - an assembler error in the busiest instantiations)
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %0, %2 offen lds"
                 :: "s"(rsrc), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(soff)),
                    "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory");
}

// The parameter block re-read from the kernarg segment behind an opaque asm: values loaded through it cannot be kept
// live across the K loop, which keeps the loop's scalar registers for the loop (hipcc otherwise parks dozens of epilogue
// / work-list scalars in VGPR lanes and reads them back inside every phase).
__device__ __forceinline__ const SmxGemmParams& pp_kernarg() {
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const SmxGemmParams*)k;
}
// Grouped launches (smx_gemm_group): up to PP_MAXG independent problems of one (layout, epilogue class) share ONE launch;
// their work items are concatenated (problem g owns items wstart[g] .. wstart[g+1]-1 of the launch).  A layer's four weight
// gradients then fill the chip with TWO K slices per output tile instead of seven each, which cuts the slab traffic of the
// split-K reduction 3.5x and four launches (+ their tails) to one.
#define PP_MAXG 8            // (round 4: a decoder layer has seven weight gradients; 8 x 288 B of parameter blocks fit the 4-KB kernarg segment)
struct SmxGemmGroup {
    int count, W;
    int wstart[PP_MAXG + 1];
    int _pad;
    SmxGemmParams prob[PP_MAXG];
};
template <bool GRP>
__device__ __forceinline__ const SmxGemmParams& pp_kernarg_g(int g) {
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    if constexpr (GRP) return ((const SmxGemmGroup*)k)->prob[g];
    else return *(const SmxGemmParams*)k;
}
__device__ __forceinline__ const SmxGemmGroup& pp_group() {
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const SmxGemmGroup*)k;
}

// 4 B per lane: LDS[m0 + lane * 4] = mem[rsrc.base + voff]  (zeros beyond num_records)
__device__ __forceinline__ void pp_dma4(pp_rsrc_t rsrc, unsigned voff, unsigned lds_wave_base) {
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);          // (as in pp_dma16)
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %0, 0 offen lds"
                 :: "s"(rsrc), "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory");
}

struct PPItem {
    int m0, n0, ks0, nk;
    long long za, zb, zc, zbias, ze;
};

// a / b for 0 <= a < 2^22, b >= 1, rb = 1.0f / b: one multiply and a one-step fix-up instead of the ~40-instruction
// integer division (the work-list and row-view decodes run inside a load segment, with the partner group waiting)
__device__ __forceinline__ int pp_fdiv(int a, int b, float rb) {
    int q = (int)((float)a * rb);
    const int r = a - q * b;
    q += (r >= b) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
__device__ __forceinline__ long long pp_view_off(const SmxRowView& v, int r, float rrpb) {
    if (v.rows_per_batch > 0) {
        const int b = pp_fdiv(r, v.rows_per_batch, rrpb);
        return v.off + (long long)b * v.batch_stride + (long long)(r - b * v.rows_per_batch) * v.ld;
    }
    return v.off + (long long)r * v.ld;
}

// reciprocals of the work-list divisors, computed once per workgroup
struct PPDiv {
    float r_nwg, r_pg, r_split;
    int nwg, per_group, ntm, ntn, W, kst, per, bm;
    __device__ __forceinline__ void init(const SmxGemmParams& p, int ntm_, int ntn_, int bm_ = PP_BM) {
        ntm = ntm_; ntn = ntn_; bm = bm_;
        nwg = ntm * ntn;
        per_group = PP_GROUP * ntm;
        W = nwg * p.nbatch * p.split_k;
        r_nwg = 1.0f / (float)nwg;
        r_pg = 1.0f / (float)per_group;
        r_split = 1.0f / (float)p.split_k;
        kst = (p.K + BK - 1) / BK;
        per = (kst + p.split_k - 1) / p.split_k;
    }
};

__device__ __forceinline__ void pp_decode(const SmxGemmParams& p, const PPDiv& d, int q, PPItem& it) {
    // XCD x (= q & 7 in dispatch order) owns a contiguous range of the work list (bijective for any W)
    const int qq = d.W >> 3, r = d.W & 7, x = q & 7, y = q >> 3;
    const int w = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + y;
    const int z = pp_fdiv(w, d.nwg, d.r_nwg), lin = w - z * d.nwg;
    const int grp = pp_fdiv(lin, d.per_group, d.r_pg), rem = lin - grp * d.per_group;
    const int first = grp * PP_GROUP;
    const int gsz = min(d.ntn - first, PP_GROUP);                     // 1..4
    const int tm = gsz == 4 ? rem >> 2 : gsz == 2 ? rem >> 1 : gsz == 1 ? rem : pp_fdiv(rem, 3, 1.0f / 3.0f);
    const int tn = first + (rem - tm * gsz);
    it.m0 = tm * d.bm;
    it.n0 = tn * PP_BN;
    const int zb = pp_fdiv(z, p.split_k, d.r_split), zs = z - zb * p.split_k;
    it.za = (long long)zb * p.batch_a;
    it.zb = (long long)zb * p.batch_b;
    it.zc = (long long)zb * p.batch_c + (long long)zs * p.split_stride;
    it.zbias = (long long)zb * p.batch_bias;
    it.ze = (long long)zb * p.batch_e;
    it.ks0 = zs * d.per;
    it.nk = max(min(d.kst, it.ks0 + d.per) - it.ks0, 0);
}

// LDS images.  A units use the 128x128 kernels' images (kc_addr / rc_addr).  KC B units swizzle their 16-B chunks with pp_bswz
// so that the permuted fragment rows {8 (i>>2) + 4 j + (i & 3)} stay conflict-free.
__device__ __forceinline__ int pp_bswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }
// RC B units (round 3): k-row k is 256 B = 16 chunks of 8 columns, chunk c stored at slot c ^ pp_rcb_swz(k).  The permuted
// fragment read takes 8 B (4 columns) of each of 4 consecutive chunks per k-row - HALF of every chunk it touches - so with the
// rc_addr image (what rounds 1-2 used here) the 16 k-rows of one ds_read_b64_tr_b16 met on 32 of the 64 banks: a 4-way
// conflict where 512 B need 2 cycles (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25 on every kernel with a
// rows-contiguous B, 0.00 without).  Here (a) the four k-rows of a quad go to four different 64-B blocks (slot bits 2-3 <- k & 3),
// (b) k-rows 8 apart in the same block swap chunk parity (slot bit 0 <- bit 3 of k) and (c) odd lane quads read the OTHER half
// of their chunk (fragment j of lane quad c holds columns 8 c + 4 (j ^ (c & 1)) ..: still 8 consecutive columns per lane
// after both fragments; the epilogues swap the two halves back, template flag BSW), so one read covers all 64 banks twice.
__device__ __forceinline__ int pp_rcb_swz(int k) { return ((k & 3) << 2) | ((k >> 3) & 1); }

// MT: rows of the output tile (256, or 192 = two wave-row groups of 96: unit AH1 then holds 2 x 32 rows)
template <bool RC, bool IS_A, bool VIEW, int MT = PP_BM>
struct PPOperand {
    pp_rsrc_t rsrc;
    unsigned soff;          // scalar byte offset of the current K tile
    unsigned sstep;         // its increment per K tile
    unsigned voff[2][2];    // KC: [half][pass] byte offset of my (row, chunk);  RC: [0][pass] = my k-row, [1][half] = my columns
    int rt[1], rb[1];       // RC through a batched view (VIEW): my pass-0 k-row as (row inside batch, batch); pass 1 = +32 rows
    int rpb;                // RC + VIEW: rows per batch of the view

    static __device__ __forceinline__ int grow(int h, int hr) {      // unit-local row -> tile row (-1: not part of the unit)
        if (IS_A) {
            if (MT == 192 && h == 1) return hr < 64 ? (hr >> 5) * 96 + 64 + (hr & 31) : -1;
            return (hr >> 6) * (MT / 2) + h * 64 + (hr & 63);
        }
        return (hr >> 5) * 64 + h * 32 + (hr & 31);
    }
    __device__ __forceinline__ void init(const bf16_t* b, const SmxRowView& v, int row0, int nrows, int k0, int tid) {
        rsrc = pp_make_rsrc(b);
        const float rrpb = __builtin_amdgcn_rcpf((float)max(v.rows_per_batch, 1));          // (pp_fdiv repairs the last place)
        const int lane = tid & 63, wave = tid >> 6;
        if (!RC) {
            soff = __builtin_amdgcn_readfirstlane((unsigned)k0 * 2u);          // (uniform: kept in a scalar register)
            sstep = BK * 2u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int hr = ps * 64 + wave * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ (IS_A ? ((hr >> 1) & 7) : pp_bswz(hr));
                    const int gr = grow(h, hr);
                    const int r = row0 + gr;
                    voff[h][ps] = (gr >= 0 && r < nrows) ? (unsigned)(pp_view_off(v, r, rrpb) + c * 8) * 2u : PP_OOB;
                }
        } else {
            const int kl = wave * 4 + (lane >> 4), g16 = lane & 15;      // rc_swz(kl) is the same for both passes
            const int hc = IS_A ? ((((g16 >> 1) ^ rc_swz(kl)) << 1) | (g16 & 1)) * 8 : (g16 ^ pp_rcb_swz(kl)) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int gc = grow(h, hc);
                const int c = row0 + gc;
                voff[1][h] = (gc >= 0 && c < nrows) ? (unsigned)c * 2u : PP_OOB;
            }
            rpb = VIEW ? (v.rows_per_batch > 0 ? v.rows_per_batch : 0x40000000) : 0;     // plain rows: one endless batch
            if constexpr (VIEW) {
                soff = 0; sstep = 0;
                const int k = k0 + kl;
                rb[0] = v.rows_per_batch > 0 ? pp_fdiv(k, v.rows_per_batch, rrpb) : 0;
                rt[0] = k - rb[0] * (v.rows_per_batch > 0 ? v.rows_per_batch : 0);
                view_rows(v);
            } else {
                soff = __builtin_amdgcn_readfirstlane((unsigned)((long long)k0 * v.ld * 2));
                sstep = (unsigned)(v.ld * BK * 2);
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) voff[0][ps] = (unsigned)((v.off + (long long)(ps * 32 + kl) * v.ld) * 2);
            }
        }
    }
    __device__ __forceinline__ void view_rows(const SmxRowView& v) {
        int t1 = rt[0] + 32, b1 = rb[0];                       // pass 1: 32 k-rows further
        while (t1 >= rpb) { t1 -= rpb; b1 += 1; }
        voff[0][0] = (unsigned)((v.off + (long long)rb[0] * v.batch_stride + (long long)rt[0] * v.ld) * 2);
        voff[0][1] = (unsigned)((v.off + (long long)b1 * v.batch_stride + (long long)t1 * v.ld) * 2);
    }
    // unit H of the K tile whose first k is k0 -> LDS at byte address lds (wave-uniform part added here)
    // The K tile that crosses K (only when K % 64 != 0) takes a branch of its own: on every other tile an instruction is its m0 write and
    // the load - the selects of the masked form, executed for every tile, cost the K loop more than the loads themselves (round 5).
    template <int H>
    __device__ __forceinline__ void issue(unsigned lds, int k0, int K, int wave_u) const {
        const bool tail = k0 + BK > K;                // uniform
        const unsigned ldsw = lds + (unsigned)wave_u * 1024u;          // (ps * 64 + wave * 8) rows x 128 B = (ps * 32 + wave * 4) k-rows x 256 B
        constexpr int NPS = (!RC && IS_A && MT == 192 && H == 1) ? 1 : 2;      // the 64-row unit: one pass
        if (__builtin_expect(tail, 0)) {
            asm volatile("" ::: "memory");            // (keeps this a branch: no if-conversion into the common path)
            const int ln = (int)(threadIdx.x & 63);
            const int krow = wave_u * 4 + (ln >> 4);       // RC: my k-row inside a pass
            // KC: first k of my 16-B chunk inside a K tile (recomputed here, on the rare path, rather than held in a register across the K loop)
            const int hr0 = wave_u * 8 + (ln >> 3);
            const int kc = ((ln & 7) ^ (IS_A ? ((hr0 >> 1) & 7) : pp_bswz(hr0))) * 8;
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                unsigned vo = RC ? voff[0][ps] + voff[1][H] : voff[H][ps];       // RC: the column part is PP_OOB when out of range: the sum stays >= 2^31
                if (RC ? (k0 + ps * 32 + krow >= K) : (k0 + kc >= K)) vo = PP_OOB;
                pp_dma16(rsrc, vo, soff, ldsw + ps * 8192);
            }
            return;
        }
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            pp_dma16(rsrc, RC ? voff[0][ps] + voff[1][H] : voff[H][ps], soff, ldsw + ps * 8192);
        }
    }
    __device__ __forceinline__ void advance() {
        soff += sstep;
        if constexpr (RC && VIEW) {
            rt[0] += BK;
            while (rt[0] >= rpb) { rt[0] -= rpb; rb[0] += 1; }
            view_rows(IS_A ? pp_kernarg().a : pp_kernarg().b);
        }
    }
};

// Issue side of the flat unit stream: runs six units ahead of the compute side over the same (item, K tile) sequence.
template <bool A_RC, bool B_RC, bool BVIEW, bool GRP = false, int MT = PP_BM>
struct PPIssue {
    PPOperand<A_RC, true, BVIEW && A_RC, MT> a;    // BVIEW: the (RC, RC) instantiation whose operands go through batched views
    PPOperand<B_RC, false, BVIEW> b;
    PPDiv dv;
    int q, qstep;
    int kt, nk, k0, seq, wave_u, K;
    int ahead;              // free-running schedule: calls for kinds 2 / 3 that were served early (gemm_fr.hip); 0 elsewhere
    int g;                  // GRP: problem of the item being issued
    unsigned lds0;          // LDS byte address of the stage buffers
    int live;               // (an int: a bool carried around the K loop becomes a lane mask that every issue point converts back)

    __device__ __forceinline__ void load_item(int tid) {
        if constexpr (GRP) {
            live = q < pp_group().W ? 1 : 0;
            if (!live) return;
            while (q >= pp_group().wstart[g + 1]) ++g;          // items are handed out in increasing order
        } else {
            live = q < dv.W ? 1 : 0;
            if (!live) return;
        }
        const SmxGemmParams& p = pp_kernarg_g<GRP>(g);
        PPItem it;
        if constexpr (GRP) {
            PPDiv d;
            d.init(p, (p.M + MT - 1) / MT, (p.N + PP_BN - 1) / PP_BN, MT);
            pp_decode(p, d, q - pp_group().wstart[g], it);
            K = p.K;
        } else {
            pp_decode(p, dv, q, it);
        }
        k0 = it.ks0 * BK;
        nk = it.nk;                       // >= 1: the launcher rejects split counts that leave a slice empty
        kt = 0;
        // (both views and the sizes are read from the parameter block in one go: taken field by field where they are used, every group of
        // scalar loads costs its own wait - and this runs inside a K tile, with all eight waves of the workgroup at the same point)
        const SmxRowView va = p.a, vb = p.b;
        const bf16_t* const pa = reinterpret_cast<const bf16_t*>(p.A);
        const bf16_t* const pb = reinterpret_cast<const bf16_t*>(p.B);
        const int pM = p.M, pN = p.N;
        a.init(pa + it.za, va, it.m0, pM, k0, tid);
        b.init(pb + it.zb, vb, it.n0, pN, k0, tid);
    }
    // KIND: 0 AH0, 1 BH0, 2 BH1, 3 AH1 (then move to the next K tile).  Returns false when the stream has ended.
    // FREEZE (ablation builds): every K tile re-reads the item's first one.
    template <int KIND, bool FREEZE = false>
    __device__ __forceinline__ bool issue(int tid) {
        if (MT == 192 && KIND >= 2 && ahead) { --ahead; return true; }          // (the 256-row forms have no register to spare for it)
        if (!live) return false;
        const unsigned st = lds0 + (unsigned)(seq & 1) * PP_STAGE;
        if (KIND == 0) a.template issue<0>(st + 0 * PP_UNIT, k0, K, wave_u);
        else if (KIND == 1) b.template issue<0>(st + 2 * PP_UNIT, k0, K, wave_u);
        else if (KIND == 2) b.template issue<1>(st + 3 * PP_UNIT, k0, K, wave_u);
        else {
            a.template issue<1>(st + 1 * PP_UNIT, k0, K, wave_u);
            ++seq;
            if (++kt == nk) {
                q += qstep;
                load_item(tid);
            } else if (!FREEZE) {
                k0 += BK;
                a.advance();
                b.advance();
            }
        }
        return true;
    }
};

// B fragment j (0/1) of a 32-column half for the wave-column block starting at unit row r32: fragment column i of lane
// i <-> logical column 8 (i >> 2) + 4 j + (i & 3)   (RC: 4 (j ^ ((i >> 2) & 1)), see pp_rcb_swz)
template <bool RC>
__device__ __forceinline__ bf16x8_t pp_bfrag(const char* unit, int r32, int j, int kk, int lane) {
    const int i = lane & 15, g = lane >> 4;
    union { bf16x8_t v; uint4 u; uint2 h[2]; } f;
    if (!RC) {
        const int row = r32 + 8 * (i >> 2) + 4 * j + (i & 3);
        f.u = *reinterpret_cast<const uint4*>(unit + row * 128 + (((kk * 4 + g) ^ pp_bswz(row)) << 4));
    } else {
        const int q = i >> 2, c = i & 3;
        const int kb = kk * 32 + 8 * g + q;                          // pp_rcb_swz(kb) == pp_rcb_swz(kb + 4) == (q << 2) | (g & 1)
        const int a = kb * 256 + ((((r32 >> 3) + c) ^ ((q << 2) | (g & 1))) << 4) + ((j ^ (c & 1)) << 3);
        f.h[0] = lds_tr_b64(unit + a);
        f.h[1] = lds_tr_b64(unit + a + 4 * 256);
    }
    return f.v;
}

// acc[rh*4+a][2 ch + j][r]: row mw0 + rh*64 + a*16 + (lane & 15), column nw0 + ch*32 + 8 (lane >> 4) + 4 j + r
// BSW (rows-contiguous B): lanes of odd 16-lane groups hold the two 4-column halves in the other order (pp_rcb_swz)
template <bool GRP = false, int NB = 8, bool BSW = false>
__device__ __forceinline__ void pp_epilogue(f32x4_t (&acc)[NB][4], int mw0, int nw0, long long zc, long long zbias,
                                            long long ze, int lane, int gi = 0) {
    const SmxGemmParams& p = pp_kernarg_g<GRP>(gi);
    const int i16 = lane & 15, g = lane >> 4;
    const bool sw = BSW && (g & 1);
    float bs[2][8];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int n = nw0 + ch * 32 + g * 8;
        if (p.bias && n + 8 <= p.N && !((zbias + n) & 3)) {
            load8(p.bias + zbias + n, bs[ch]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[ch][e] = (p.bias && n + e < p.N) ? p.bias[zbias + n + e] : 0.f;
        }
        // Consume the bias right here: a load hipcc still tracks as pending when the next K loop starts makes it drain
        // the whole VMEM queue (LDS-DMA prefetches included) before the first LDS read of every K tile.
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bs[ch][e]));
    }
    const unsigned th = smx_thresh24(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    const unsigned dseed = smx_dseed(p.drop_p, p.drop_seed);          // + the step key (smx_common.h), read once per item
    // rolled over the 16 (row block, column half) pieces - ONE copy of the row epilogue in the binary; the accumulators
    // are picked by a wave-uniform switch so that every register index stays static
#pragma clang loop unroll(disable)
    for (int it = 0; it < 2 * NB; ++it) {
        float x[8], b8[8];
#define PP_GET(A_, C)                                                                                    \
    constexpr int A = (A_) < NB ? (A_) : 0;                                                              \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) b8[e] = bs[C][e];                                      \
    x[0] = acc[A][2 * C][0]; x[1] = acc[A][2 * C][1]; x[2] = acc[A][2 * C][2]; x[3] = acc[A][2 * C][3];  \
    x[4] = acc[A][2 * C + 1][0]; x[5] = acc[A][2 * C + 1][1]; x[6] = acc[A][2 * C + 1][2]; x[7] = acc[A][2 * C + 1][3];
        switch (it) {
            case 0: { PP_GET(0, 0) } break;  case 1: { PP_GET(0, 1) } break;  case 2: { PP_GET(1, 0) } break;  case 3: { PP_GET(1, 1) } break;
            case 4: { PP_GET(2, 0) } break;  case 5: { PP_GET(2, 1) } break;  case 6: { PP_GET(3, 0) } break;  case 7: { PP_GET(3, 1) } break;
            case 8: { PP_GET(4, 0) } break;  case 9: { PP_GET(4, 1) } break;  case 10: { PP_GET(5, 0) } break; case 11: { PP_GET(5, 1) } break;
            case 12: { PP_GET(6, 0) } break; case 13: { PP_GET(6, 1) } break; case 14: { PP_GET(7, 0) } break; default: { PP_GET(7, 1) } break;
        }
#undef PP_GET
        if (BSW) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = x[e], hi = x[e + 4];
                x[e] = sw ? hi : lo;
                x[e + 4] = sw ? lo : hi;
            }
        }
        const int a8 = it >> 1, ch = it & 1;
        const int m = mw0 + (a8 >> 2) * 64 + (a8 & 3) * 16 + i16;
        const int n = nw0 + ch * 32 + g * 8;
        if (m < p.M && n < p.N) epilogue_row8<true>(p, zc, ze, m, n, x, b8, th, inv_keep, dseed);
    }
}

// ---- fast epilogues: fully unrolled, specialised at compile time by class, taken when every 16-B access is aligned
// (checked per launch, wave-uniform); anything else goes through the rolled generic epilogue above.
//   EPI 0 "linear":  C = bf16(alpha acc + bias) [x dropout] [+ resid]
//   EPI 1 "act":     aux_out = bf16(pre) ; C = bf16(act(pre)) [x dropout]          (pre = alpha acc + bias)
//   EPI 2 "actgrad": C = bf16(pre x act'(aux_in)) [x dropout]
//   EPI 3 "f32":     C(fp32) = pre [+ C when atomic == 2]                            (weight gradients, split-K slabs)
// Side inputs of 8 row pieces are loaded together BEFORE the first store of the group: hipcc waits for its own loads with
// counts that do not know about the inline-asm stores, so a load issued behind a store would wait for that store too.

__device__ __forceinline__ bool pp_views_aligned(const SmxGemmParams& p) {
    const long long m = p.c.ld | p.c.off | p.c.batch_stride | p.e.ld | p.e.off | p.e.batch_stride | p.batch_c | p.batch_e |
                        p.split_stride | p.batch_bias;
    return !(m & 7) && !(p.N & 7);
}

// Memory-side lane layout of the bf16 classes (round 5, tools/lab/st_lab.hip).  The accumulators give lane = row + 16 * chunk (a quad of
// lanes = four ROWS, 16 bytes each): a CU issues such a 16-B-per-lane access at ~17 B/clk whether it is a store or a load, against 64 B/clk
// when every quad of lanes covers 64 contiguous bytes.  So the packed outputs (and the side inputs) cross the lanes once through
// ds_bpermute_b32 - no LDS memory - between that layout and lane = 4 * row + chunk, in which all 16-B global accesses are made.
#ifndef SMX_EPI_PERM
#define SMX_EPI_PERM 1
#endif
__device__ __forceinline__ uint4 pp_lane_perm(uint4 v, int src4) {          // every lane takes the 16 bytes of lane src4 / 4
    return make_uint4((unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)v.x), (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)v.y),
                      (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)v.z), (unsigned)__builtin_amdgcn_ds_bpermute(src4, (int)v.w));
}

template <int EPI, bool GRP = false, int NB = 8, bool BSW = false>
__device__ __forceinline__ void pp_epilogue_fast(f32x4_t (&acc)[NB][4], int mw0, int nw0, int n0, const char* bias_lds,
                                                 long long zc, long long ze, int lane, int gi = 0) {
    const SmxGemmParams& p = pp_kernarg_g<GRP>(gi);
    const int i16 = lane & 15, g = lane >> 4;
    const bool sw = BSW && (g & 1);
    const int nl = nw0 + g * 8;                       // my first column (half 0); half 1 = + 32
    // (where it pays: the linear class; the 192-row form of the data-gradient class - the activation class is VALU-bound, the 256-row forms of
    // the other two have no registers left for the exchange)
    constexpr bool PERM = SMX_EPI_PERM && (EPI == PP_EPI_LINEAR || (EPI == PP_EPI_ACTGRAD && NB <= 6));
    const int rm = PERM ? (lane >> 2) : i16;          // memory side: my row inside a 16-row block ...
    const int nm = nw0 + (PERM ? (lane & 3) : g) * 8; // ... and my first column
    const int to_mem = ((lane >> 2) + 16 * (lane & 3)) << 2, from_mem = (4 * i16 + g) << 2;      // ds_bpermute source lanes (x 4)
    // bias: the item's 256-column slice was put into LDS by an LDS-DMA issued when the item started (zeros beyond N)
    float bs[2][8];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        if (p.bias) {
            const float* b = reinterpret_cast<const float*>(bias_lds) + (nl - n0) + ch * 32;
            const float4 lo = *reinterpret_cast<const float4*>(b), hi = *reinterpret_cast<const float4*>(b + 4);
            bs[ch][0] = lo.x; bs[ch][1] = lo.y; bs[ch][2] = lo.z; bs[ch][3] = lo.w;
            bs[ch][4] = hi.x; bs[ch][5] = hi.y; bs[ch][6] = hi.z; bs[ch][7] = hi.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[ch][e] = 0.f;
        }
    }
    const unsigned th = smx_thresh24(p.drop_p);
    const float inv_keep = 1.0f / (1.0f - p.drop_p);
    const bool drop = p.drop_p > 0.f;
    // The views, sizes and pointers are taken out of the parameter block ONCE: behind every inline-asm store (a memory clobber) the compiler
    // re-reads whatever it still addresses through `p` - a scalar load and its wait per row group (round 5 timeline: 0.5 us of 1.2 per group).
    const SmxRowView vc = p.c, ve = p.e;
    const int pM = p.M, pN = p.N;
    const float alpha = p.alpha;
    char* const pC = reinterpret_cast<char*>(p.C);
    const bf16_t* const p_side = reinterpret_cast<const bf16_t*>(EPI == PP_EPI_ACTGRAD ? p.aux_in : p.resid);
    bf16_t* const p_aux = reinterpret_cast<bf16_t*>(p.aux_out);
    const float rrc = __builtin_amdgcn_rcpf((float)max(vc.rows_per_batch, 1)), rre = __builtin_amdgcn_rcpf((float)max(ve.rows_per_batch, 1));
    const bool has_res = EPI == PP_EPI_LINEAR && p.resid;
    const bool has_acc = EPI == PP_EPI_F32 && p.atomic == 2;
    const bool need_e = EPI == PP_EPI_ACT || EPI == PP_EPI_ACTGRAD || has_res;
    // the seed + the step key (one scalar load, dropout launches only); the ACTGRAD class reads it where it hashes - which it does only without
    // the saved derivative - because one more value held across its groups spills in the 256-row form
    const unsigned dseed_h = (EPI != PP_EPI_ACTGRAD && drop) ? smx_dseed(p.drop_p, p.drop_seed) : 0u;
#define PP_DSEED() (EPI == PP_EPI_ACTGRAD ? smx_dseed(p.drop_p, p.drop_seed) : dseed_h)
    // SMX_ACT_SAVE_GRAD (round 3): the side tensor holds the epilogue's local derivative act'(pre) x dropout multiplier - written by
    // the forward ACT class in place of the pre-activation copy, multiplied in by the data gradient's ACTGRAD class (no
    // activation derivative, no mask regeneration there); same arithmetic as epilogue_staged_fast<4 / 5> of the 128-family kernels
    const bool saved = (EPI == PP_EPI_ACT || EPI == PP_EPI_ACTGRAD) && (p.act & SMX_ACT_SAVE_GRAD);
    const int act = p.act & 0xff;
#ifndef SMX_EPI_GA
#define SMX_EPI_GA 0
#endif
#ifndef SMX_EPI_AG8_GA          // row blocks per group of the 256-row data-gradient class: 4 spills 8 - 12 bytes there (lab switch)
#define SMX_EPI_AG8_GA 2
#endif
    // row blocks per group (x 2 halves = pieces in registers at once)
    PP_STAMP(40);
#ifndef SMX_WS_GA          // row blocks per group of the 128-register waves of gemm_ws.hip (NB == 4)
#define SMX_WS_GA 1
#endif
    constexpr int GA = NB == 4 ? SMX_WS_GA : (SMX_EPI_GA && (EPI == PP_EPI_LINEAR || EPI == PP_EPI_ACTGRAD) && NB % (SMX_EPI_GA ? SMX_EPI_GA : 1) == 0) ? SMX_EPI_GA
                       : (EPI == PP_EPI_F32 || EPI == PP_EPI_ACT || (EPI == PP_EPI_ACTGRAD && NB == 8 && SMX_EPI_AG8_GA == 2) || NB % 4) ? 2 : 4;
#pragma unroll
    for (int grp = 0; grp < NB / GA; ++grp) {
        long long cb[GA], eb[GA];
        bool rok[GA];
        uint4 side[GA][2];
        float4 accum[GA][2][2];
#pragma unroll
        for (int a = 0; a < GA; ++a) {
            const int a8 = grp * GA + a;
            const int m = mw0 + (a8 >> 2) * 64 + (a8 & 3) * 16 + rm;
            rok[a] = m < pM;
            const int mm = rok[a] ? m : 0;
            cb[a] = zc + pp_view_off(vc, mm, rrc) + nm;
            eb[a] = need_e ? ze + pp_view_off(ve, mm, rre) + nm : 0;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const bool ok = rok[a] && nm + ch * 32 < pN;
                if (PERM) side[a][ch] = make_uint4(0u, 0u, 0u, 0u);          // (lanes outside the output still take part in the lane exchange)
                if (EPI == PP_EPI_ACTGRAD) {
                    if (ok) side[a][ch] = *reinterpret_cast<const uint4*>(p_side + eb[a] + ch * 32);
                } else if (EPI == PP_EPI_LINEAR) {
                    if (has_res && ok) side[a][ch] = *reinterpret_cast<const uint4*>(p_side + eb[a] + ch * 32);
                } else if (EPI == PP_EPI_F32) {
                    if (has_acc && ok) {
                        const float* c = reinterpret_cast<const float*>(pC) + cb[a] + ch * 32;
                        accum[a][ch][0] = *reinterpret_cast<const float4*>(c);
                        accum[a][ch][1] = *reinterpret_cast<const float4*>(c + 4);
                    }
                }
            }
        }
        // every side input is consumed HERE on every path: a load hipcc still tracks as pending when the next K loop
        // starts makes it drain the whole VMEM queue before the first LDS read of every K tile
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                if (EPI == PP_EPI_ACTGRAD || EPI == PP_EPI_LINEAR)
                    asm volatile("" : "+v"(side[a][ch].x), "+v"(side[a][ch].y), "+v"(side[a][ch].z), "+v"(side[a][ch].w));
                if (EPI == PP_EPI_F32)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        asm volatile("" : "+v"(accum[a][ch][h].x), "+v"(accum[a][ch][h].y), "+v"(accum[a][ch][h].z),
                                     "+v"(accum[a][ch][h].w));
            }
        if (grp == 0) PP_STAMP(41);
        if (PERM && (EPI == PP_EPI_ACTGRAD || (EPI == PP_EPI_LINEAR && has_res))) {
#pragma unroll
            for (int a = 0; a < GA; ++a)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) side[a][ch] = pp_lane_perm(side[a][ch], from_mem);
        }
        // all the arithmetic of the group first (independent chains the compiler can interleave: the inline-asm stores
        // are ordering points), results packed in registers, then the stores back to back
        uint4 outv[GA][2], auxv[EPI == PP_EPI_ACT ? GA : 1][2];
        float4 outf[EPI == PP_EPI_F32 ? GA : 1][2][2];
#pragma unroll
        for (int a = 0; a < GA; ++a) {
            const int a8 = grp * GA + a;
            const int m = mw0 + (a8 >> 2) * 64 + (a8 & 3) * 16 + i16;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int n = nl + ch * 32;
                float x[8] = {acc[a8][2 * ch][0], acc[a8][2 * ch][1], acc[a8][2 * ch][2], acc[a8][2 * ch][3],
                              acc[a8][2 * ch + 1][0], acc[a8][2 * ch + 1][1], acc[a8][2 * ch + 1][2], acc[a8][2 * ch + 1][3]};
                if (BSW) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = x[e], hi = x[e + 4];
                        x[e] = sw ? hi : lo;
                        x[e + 4] = sw ? lo : hi;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], alpha, bs[ch][e]);
                if (EPI == PP_EPI_F32) {
                    if (has_acc) {
                        x[0] += accum[a][ch][0].x; x[1] += accum[a][ch][0].y; x[2] += accum[a][ch][0].z; x[3] += accum[a][ch][0].w;
                        x[4] += accum[a][ch][1].x; x[5] += accum[a][ch][1].y; x[6] += accum[a][ch][1].z; x[7] += accum[a][ch][1].w;
                    }
                    outf[a][ch][0] = make_float4(x[0], x[1], x[2], x[3]);
                    outf[a][ch][1] = make_float4(x[4], x[5], x[6], x[7]);
                    continue;
                }
                if (EPI == PP_EPI_ACT) {
                    if (saved) {
                        auxv[a][ch] = act_fwd_grad_drop8(x, act, drop, PP_DSEED(), (unsigned)((long long)m * pN + n + zc), th, inv_keep);
                    } else {
                        auxv[a][ch] = make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
                        act_fwd8(x, act);
                    }
                }
                if (EPI == PP_EPI_ACTGRAD) {
                    const uint4 u = side[a][ch];
                    float s[8] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                  __uint_as_float(u.y & 0xffff0000u), __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                  __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)};
                    if (saved) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) x[e] *= s[e];
                    } else {
                        act_grad_mul8(x, s, act);
                    }
                }
                if (drop && !saved) {
                    const unsigned idx = (unsigned)((long long)m * pN + n + zc);
                    smx_drop_mul8(PP_DSEED(), idx, th, inv_keep, x);
                }
                if (EPI == PP_EPI_LINEAR && has_res) {
                    const uint4 u = side[a][ch];
                    x[0] += __uint_as_float(u.x << 16); x[1] += __uint_as_float(u.x & 0xffff0000u);
                    x[2] += __uint_as_float(u.y << 16); x[3] += __uint_as_float(u.y & 0xffff0000u);
                    x[4] += __uint_as_float(u.z << 16); x[5] += __uint_as_float(u.z & 0xffff0000u);
                    x[6] += __uint_as_float(u.w << 16); x[7] += __uint_as_float(u.w & 0xffff0000u);
                }
                outv[a][ch] = make_uint4(pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(x[4], x[5]), pack_bf2(x[6], x[7]));
            }
        }
        if (grp == 0) PP_STAMP(42);
        if (PERM) {
#pragma unroll
            for (int a = 0; a < GA; ++a)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    outv[a][ch] = pp_lane_perm(outv[a][ch], to_mem);
                    if (EPI == PP_EPI_ACT) auxv[EPI == PP_EPI_ACT ? a : 0][ch] = pp_lane_perm(auxv[EPI == PP_EPI_ACT ? a : 0][ch], to_mem);
                }
        }
        if (grp == 0) PP_STAMP(43);
#pragma unroll
        for (int a = 0; a < GA; ++a)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                if (!(rok[a] && nm + ch * 32 < pN)) continue;
                if (EPI == PP_EPI_F32) {
                    float* c = reinterpret_cast<float*>(pC) + cb[a] + ch * 32;
                    st_b128(c, make_uint4(__float_as_uint(outf[a][ch][0].x), __float_as_uint(outf[a][ch][0].y),
                                          __float_as_uint(outf[a][ch][0].z), __float_as_uint(outf[a][ch][0].w)));
                    st_b128(c + 4, make_uint4(__float_as_uint(outf[a][ch][1].x), __float_as_uint(outf[a][ch][1].y),
                                              __float_as_uint(outf[a][ch][1].z), __float_as_uint(outf[a][ch][1].w)));
                    continue;
                }
                if (EPI == PP_EPI_ACT && p_aux) st_b128(p_aux + eb[a] + ch * 32, auxv[a][ch]);
                st_b128(reinterpret_cast<bf16_t*>(pC) + cb[a] + ch * 32, outv[a][ch]);
            }
        if (grp == 0) PP_STAMP(44);
    }
}
#undef PP_DSEED
