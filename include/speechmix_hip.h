/* speechmix_hip.h - C ABI of libspeechmix_hip.so, the MI355X (gfx950) kernel library under the SpeechMix
 * drop-in classes.
 *
 * The reference (voidful/SpeechMix) is 100 % Python and has NO native interface of its own: its hot path calls
 * PyTorch modules (ref:speechmix/model.py:148 `self.encoder_model(input_values)`, :135-136
 * `self.decoder_model(inputs_embeds=..., decoder_input_ids=..., labels=...)`, :162 `self.length_adapters`, :165
 * `self.enc_to_dec_proj`).  The entry points below are therefore what a Python binding (ctypes, see
 * speechmix_amd/_lib.py, or any other FFI) binds INSTEAD of those module calls; each one cites the reference /
 * transformers (TF:) code whose arithmetic it executes.  Conventions:
 *   - plain C: raw device pointers + sizes, no torch types; the caller owns every buffer (PyTorch's caching
 *     allocator in this repo) - kernels never allocate or free;
 *   - every launcher enqueues on the given hipStream_t and returns 0 on success, a hipError_t (>0) or a negative
 *     argument error (-22); nothing throws across the boundary;
 *   - dtype: SMX_F32 (parity path) or SMX_BF16 (bf16 storage, fp32 accumulate / statistics);
 *   - struct layouts are mirrored in speechmix_amd/_lib.py and checked at load time via smx_sizeof_*().
 */
#ifndef SPEECHMIX_HIP_H
#define SPEECHMIX_HIP_H
#include <stddef.h>
#include <hip/hip_runtime_api.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { SMX_F32 = 0, SMX_BF16 = 1 };
enum { SMX_ACT_NONE = 0, SMX_ACT_GELU = 1, SMX_ACT_RELU = 2 };
/* OR-ed into SmxGemmParams.act (bf16 GEMMs, forward with aux_out / data gradient with aux_in, and the split-K epilogue):
 * the side tensor holds the epilogue's LOCAL DERIVATIVE act'(pre) * dropout multiplier instead of the pre-activation; the
 * backward launch multiplies it in (no activation gradient, no dropout hash there).  Launches outside those forms: EINVAL. */
#define SMX_ACT_SAVE_GRAD 0x100

/* logical row r of an operand -> element offset: off + (r / rows_per_batch) * batch_stride + (r % rows_per_batch) * ld
 * (rows_per_batch <= 0: off + r * ld).  Lets a Conv1d be a GEMM over overlapping windows of a [B,T,C] activation. */
typedef struct SmxRowView { long long batch_stride, ld, off; int rows_per_batch, _pad; } SmxRowView;

/* C[m,n] (+)= epilogue(alpha * sum_k A(m,k) B(n,k)).  Replaces every nn.Linear / nn.Conv1d forward, data-gradient
 * and weight-gradient on the path: ref:speechmix/model.py:92-102,162,165; TF:models/wav2vec2/modeling_wav2vec2.py:254-323
 * (conv layers 1-6), :326-379 (positional conv, batched per group), :429-434, :466-572; TF:models/bart/modeling_bart.py:260-390,
 * :939-940 (LM head).  a_rc / b_rc: operand stored [K, rows] (reduction index slow) - dgrad's W, both wgrad operands. */
typedef struct SmxGemmParams {
    const void *A, *B; void* C; const float* bias; const void* resid; void* aux_out; const void* aux_in;
    SmxRowView a, b, c, e;
    long long batch_a, batch_b, batch_c, batch_bias, batch_e;
    int M, N, K, a_rc, b_rc, act, out_f32, atomic, nbatch, split_k, tr_mode; float alpha; long long split_stride;
    float drop_p; unsigned drop_seed;   /* dropout after the activation, before the residual (see "dropout" below) */
} SmxGemmParams;
int smx_gemm(const SmxGemmParams* p, int dtype, hipStream_t stream);
/* second stage of a split-K forward / data-gradient GEMM: C = epilogue(sum_s slabs[s]); slabs: nsplit x [M, ldn] fp32 */
int smx_gemm_splitk_epilogue(const SmxGemmParams* p, const float* slabs, int nsplit, long long stride, int ldn, hipStream_t stream);
int smx_reduce_slabs(const float* slabs, int nsplit, long long n, long long stride, float* dst, int accumulate, hipStream_t stream);
/* reduce_slabs for up to 8 (slabs, destination) pairs in one launch; arrays in host memory, slab stride = n[e] */
int smx_reduce_slabs_many(const float* const* slabs, float* const* dst, const long long* n, const int* nsplit, int count,
                          int accumulate, hipStream_t stream);
/* Up to 4 weight-gradient GEMMs (a_rc = b_rc = 1, fp32 output: plain or split-K slabs, plain row views) in ONE persistent
 * launch of the 256x256 kernel: the (K slice, tile) work lists of the problems are concatenated, so the weight gradients of
 * one transformer layer (the autograd of the nn.Linear calls at TF:models/wav2vec2/modeling_wav2vec2.py:466-572) fill the
 * chip with two K slices per output tile instead of seven each.  probs: `count` parameter blocks in host memory. */
int smx_gemm_group(const SmxGemmParams* probs, int count, int dtype, hipStream_t stream);

/* LayerNorm / RMSNorm (+ fused positional-table add, + fused activation) forward and backward.
 * TF:models/wav2vec2/modeling_wav2vec2.py:275-299, 429-434, 575-654; TF:models/bart/modeling_bart.py:507-549;
 * TF:models/t5/modeling_t5.py:50-72. */
typedef struct SmxNormParams {
    const void *x, *pos; void *xsum_out, *y; const float *gamma, *beta; float *mean, *rstd;
    int M, D, pos_period, pos_offset, rms, act; float eps; float drop_p; unsigned drop_seed;
} SmxNormParams;
typedef struct SmxNormBwdParams {
    const void *dy, *x, *dres; void* dx; const float *gamma, *beta, *mean, *rstd; float *dgamma, *dbeta, *dpos, *partials;
    int M, D, pos_period, pos_offset, rms, act; float drop_p; unsigned drop_seed;
    int defer_fold;   /* 1: leave the gamma/beta partial rows (smx_norm_bwd_partial_rows(M) x [2][D] floats) in `partials` for smx_fold_many */
    /* optional (defer_fold == 1, no activation): dx_drop = dx * dropout mask (drop2_p, drop2_seed) of the Linear whose dropped output
     * fed this norm (post-LN out_proj / fc2, TF:models/bart/modeling_bart.py:305-322) and its column sums - that Linear's bias
     * gradient - as a third partial row per block: partials is then rows x [3][D] */
    void* dx_drop; float drop2_p; unsigned drop2_seed;
} SmxNormBwdParams;
int smx_norm_fwd(const SmxNormParams* p, int dtype, hipStream_t stream);
int smx_norm_bwd(const SmxNormBwdParams* p, int dtype, hipStream_t stream);
int smx_norm_bwd_partial_rows(int M);

/* softmax(Q K^T * scale + bias [+causal]) V and its backward (dQ, dK, dV), Q/K/V/O addressed inside fused
 * projection buffers.  TF:models/wav2vec2/modeling_wav2vec2.py:466-548; TF:models/bart/modeling_bart.py:133-257;
 * TF:models/t5/modeling_t5.py:176-369; TF:integrations/sdpa_attention.py:39-130. */
typedef struct SmxAttnParams {
    const void *Q, *K, *V; void* O; float* lse; const float* bias; const void* dO; void *dQ, *dK, *dV; float *delta, *dbias;
    long long q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, dq_bs, dq_ld, dk_bs, dk_ld, dv_bs, dv_ld, do_bs, do_ld;
    int B, H, Tq, Tk, D, causal; float scale; float drop_p; unsigned drop_seed;
    unsigned *mask_q, *mask_k;   /* dropout keep bits, both orientations (bf16 / head_dim 64 path; smx_attn_dropout_mask) */
    /* optional [B] int32 on the device: keys at positions >= klen[b] are padding (a right-padded attention_mask) and get
     * probability 0 in forward and backward; null = all Tk keys.  TF:models/wav2vec2/modeling_wav2vec2.py:688-697 (speech
     * encoder), TF:models/bart/modeling_bart.py:741-760, 1010-1016 (text encoder self-attention, decoder cross-attention);
     * the reference's hook ref:speechmix/model.py:132-136 forwards such a mask to the LM. */
    const int* klen;
} SmxAttnParams;
/* Attention-probability dropout on the MFMA path (TF:models/wav2vec2/modeling_wav2vec2.py:533-536 nn.functional.dropout on the
 * probabilities): smx_attn_mask_words gives the sizes (32-bit words, 0/0 when no mask is needed), smx_attn_dropout_mask
 * fills both bit matrices for (drop_p, drop_seed) - the same keep decisions as every other dropout site of the library. */
int smx_attn_mask_words(const SmxAttnParams* p, int dtype, long long* nq, long long* nk);
int smx_attn_dropout_mask(const SmxAttnParams* p, hipStream_t stream);
int smx_attention_fwd(const SmxAttnParams* p, int dtype, hipStream_t stream);
int smx_attention_bwd(const SmxAttnParams* p, int dtype, hipStream_t stream);
/* T5 relative-position bias gradient: smx_attention_bwd with p->dbias set accumulates dbias[H,Tq,Tk] += sum_b dS; this
 * scatter-adds it into the [buckets, H] embedding table through the int32 bucket map [Tq*Tk]
 * (TF:models/t5/modeling_t5.py:216-279, the gather `relative_attention_bias(bucket)` run backwards). */
int smx_attn_bias_scatter(const float* dbias, const int* bucket, float* dtable, int H, int Tq, int Tk, int nbuckets,
                          hipStream_t stream);

/* feature-extractor layer 0: Conv1d(1->C,k,stride) on the waveform, fused with GroupNorm+GELU ("group") or plain
 * ("layer" extractor).  TF:models/wav2vec2/modeling_wav2vec2.py:301-323 / 275-299. */
typedef struct SmxConv0Params {
    const float *wave, *w, *cbias, *gamma, *beta; double* stats; void* y; const void* dy; double* bstats;
    float *dw, *dcbias, *dgamma, *dbeta; int B, N, C, k, stride, T0, group; float eps; int tiles_per_block;
    float* partials; int nb;   /* partials: caller workspace of smx_conv0_workspace_floats(B, C, k) floats; nb: set by the launchers */
} SmxConv0Params;
long long smx_conv0_workspace_floats(int B, int C, int k);
int smx_conv0_fwd(const SmxConv0Params* p, int dtype, hipStream_t stream);
int smx_conv0_bwd(const SmxConv0Params* p, int dtype, hipStream_t stream);

/* positional conv helpers (group-major pack, weight_norm forward/backward). TF:...wav2vec2.py:326-379 */
int smx_group_pack(const void* x, void* xg, int B, int T, int C, int G, int K, int pad_front, int dtype, hipStream_t stream);
/* Time-blocked form of the same convolution (round 4; csrc/posconv.hip): J consecutive frames per GEMM row give N = J * Cg outputs.
 * smx_posconv_pack_w: the J-times shifted taps [G][J * Cg][(K + J - 1) * Cg] from smx_wn_fwd's forward pack (flip = 1: the
 * data-gradient operand); smx_posconv_unpack: the GEMM's group-major fp32 output [G][B][Tq][Cg] -> token-major [B * T][C] with
 * bias / pre-activation copy / activation / residual; smx_posconv_fold_dw: the blocked weight gradient's J shifted diagonals
 * summed into the forward-pack layout [G][Cg][K * Cg] that smx_wn_bwd takes. */
int smx_posconv_pack_w(const void* wp, void* out, int G, int Cg, int K, int J, int flip, int dtype, hipStream_t stream);
int smx_posconv_unpack(const float* tmp, const float* bias, const void* resid, void* pre, void* y, int B, int T, int C, int G,
                       int Tq, int act, int dtype, hipStream_t stream);
int smx_posconv_fold_dw(const float* dwJ, float* dwp, int G, int Cg, int K, int J, hipStream_t stream);
/* norm and scratch_s: K * (1 + smx_wn_partial_blocks(C, Cg)) floats each (results in the first K; K <= 256) */
int smx_wn_partial_blocks(int C, int Cg);
int smx_wn_fwd(const float* v, const float* g, void* wp, void* wf, float* norm, int C, int Cg, int K, int dtype, hipStream_t stream);
int smx_wn_bwd(const float* dwp, const float* v, const float* g, const float* norm, float* scratch_s, float* dg, float* dv,
               int C, int Cg, int K, hipStream_t stream);

/* CrossEntropyLoss(ignore_index=-100, mean) + argmax + dlogits. TF:models/bart/modeling_bart.py:942-946; ref:speechmix/model.py:174 */
typedef struct SmxCEParams {
    const float* logits; const long long* labels; float* loss; long long* argmax; void* dlogits; float* lse;
    int M, V; long long ldl, ldd; float gscale;
    const float* logits_t; float* kld; float kld_scale;   /* SpeechMixSelf KLD term, ref:speechmix/model.py:257-259 */
    /* row-chunked calls (the LM head streamed over row chunks, ref:train.py:312-313 on logits memory): the mean's denominator is
     * the count of valid labels in count_labels[0 .. count_M) - the whole batch - instead of this call's own rows */
    const long long* count_labels; int count_M;
} SmxCEParams;
int smx_cross_entropy(const SmxCEParams* p, int dtype, hipStream_t stream);

/* token embedding gather / scatter-add (TF:models/bart/modeling_bart.py:101-113), bias-gradient column sums,
 * dtype casts, conv-weight re-layouts, activation backward, SpecAugment row masking
 * (TF:models/wav2vec2/modeling_wav2vec2.py:1272-1316), element-wise add. */
int smx_embed_fwd(const long long* ids, const void* table, void* out, int M, int D, float scale, int dtype, hipStream_t stream);
int smx_embed_bwd(const long long* ids, const void* dy, float* dtable, int M, int D, float scale, int dtype, hipStream_t stream);
int smx_colsum(const void* x, float* out, int M, int N, long long ld, float alpha, int dtype, hipStream_t stream);
/* same sum in two stages through a caller-owned scratch of smx_colsum_ws_floats(M, N) floats (no atomics; tall inputs) */
long long smx_colsum_ws_floats(int M, int N);
/* dropout (same mask as smx_dropout) fused with the column sums of its output: masked gradient + bias gradient in one pass */
int smx_dropout_colsum(const void* x, void* out, int M, int N, float p, unsigned seed, float* colsum, float alpha, float* ws,
                       int dtype, hipStream_t stream);
int smx_colsum_ws(const void* x, float* out, int M, int N, long long ld, float alpha, int dtype, float* ws, hipStream_t stream);
/* Deferred second stages: with out / colsum == NULL the two launchers above leave smx_colsum_slices(M, N) (resp.
 * smx_dropout_colsum_slices) partial rows of ceil8(N) (resp. N) floats in ws; smx_fold_many then performs up to
 * SMX_FOLD_MAX such reductions, dst[c] += alpha * sum_r ws[r * ld + c], in one launch (the engine flushes its queue at
 * the end of every backward stage).  smx_colsum_ws takes the two-stage path only for M >= smx_colsum_min_rows(), N % 8 == 0. */
#define SMX_FOLD_MAX 48
typedef struct SmxFoldEntry { const float* ws; float* dst; int nrows, ncols; long long ld; float alpha; int pad; } SmxFoldEntry;
typedef struct SmxFoldTable { int n, pad; SmxFoldEntry e[SMX_FOLD_MAX]; } SmxFoldTable;
int smx_fold_many(const SmxFoldTable* t, hipStream_t stream);
int smx_colsum_slices(int M, int N);
int smx_dropout_colsum_slices(int M, int N);
int smx_colsum_min_rows(void);
int smx_fold_max(void);
int smx_sizeof_SmxFoldTable(void);
int smx_cast_from_f32(const float* src, void* dst, long long n, int dtype, hipStream_t stream);
int smx_cast_to_f32(const void* src, float* dst, long long n, int dtype, hipStream_t stream);
int smx_pack_conv_w(const float* w, void* out, int Co, int Ci, int k, int dtype, hipStream_t stream);
/* Conv1d weight [Co, Ci, k] -> the K-contiguous data-gradient operands of its `stride` input residues, back to back:
 * Wd_r[ci, c * Co + co] = w[co, ci, r + (nj_r - 1 - c) * stride] (the backward of TF:models/wav2vec2/modeling_wav2vec2.py:254-323's
 * strided convolutions as forward-layout GEMMs over runs of the output gradient) */
int smx_pack_conv_w_dgrad(const float* w, void* out, int Co, int Ci, int k, int stride, int dtype, hipStream_t stream);
int smx_unpack_conv_dw(const float* dwp, float* dw, int Co, int Ci, int k, hipStream_t stream);
int smx_act_bwd(const void* dy, const void* pre, void* dx, int M, int N, const SmxRowView* out_view, int act, int dtype, hipStream_t stream);
int smx_mask_rows(void* x, const int* rows, int nrows, const float* emb, int D, int dtype, hipStream_t stream);
int smx_mask_rows_bwd(void* dx, const int* rows, int nrows, float* demb, int D, int dtype, hipStream_t stream);
int smx_add(const void* a, const void* b, void* out, long long n, int dtype, hipStream_t stream);
/* base[table[2 i] + j] = 0 for j < table[2 i + 1], i < n (device table of element offset / count pairs, counts <= 65536): the
 * step's gradient zeroing restricted to the ranges no first-writer stores (what `optimizer.zero_grad()` / DDP's bucket reset do
 * around ref:train.py:291-330 for the whole 942-MB buffer; round 4: the weight-gradient GEMMs' first write of a step stores). */
int smx_zero_ranges(float* base, const long long* table, int n, hipStream_t stream);
/* SpeechMixSelf hidden-state matching (ref:speechmix/model.py:247-255): row softmax fwd/bwd, MSE (+gradient), fp32 -> T add */
int smx_softmax_rows(float* x, int R, int Cn, hipStream_t stream);
int smx_softmax_rows_bwd(const float* p, const float* dp, float* dx, int R, int Cn, float scale, hipStream_t stream);
int smx_mse(const float* a, const float* b, float* loss, float* da, long long n, float gscale, hipStream_t stream);
int smx_add_f32_into(const float* src, void* dst, long long n, int dtype, hipStream_t stream);

/* dropout (nn.functional.dropout at TF:models/wav2vec2/modeling_wav2vec2.py:433,545,569,634,703; TF:models/bart/
 * modeling_bart.py:233,331,367-369,829,1064; TF:models/t5/modeling_t5.py:112,147,333,1020,1102).  Counter based: the
 * keep decision of element i at a site is hash32(seed, i) >> 8 >= p * 2^24, kept values are scaled by 1/(1-p); the
 * fused sites (GEMM epilogue, norm output, attention probabilities) and this stand-alone kernel use the same function,
 * so a backward pass regenerates the forward mask from (p, seed) instead of storing it. */
int smx_dropout(const void* x, void* out, long long n, float p, unsigned seed, int dtype, hipStream_t stream);
/* The STEP KEY (round 5): every kernel that hashes a dropout mask uses (its seed argument + the key), one device word the
 * library owns and this call rewrites on `stream` (0 until first set: a seed then means what it always meant).  A training
 * step captured into a HIP graph bakes its seed arguments; setting a fresh key ahead of every replay is what makes the
 * replayed step draw fresh masks (the reference draws from torch's Philox stream per call: TF sites above).  Rows of
 * smx_mask_rows / smx_mask_rows_bwd lists that are negative are skipped (fixed-capacity SpecAugment row lists of such steps). */
int smx_set_step_key(unsigned key, hipStream_t stream);   /* (the library's smx_step_key_addr_<unit> exports are its own plumbing: the key word of each translation unit) */
/* dst[0 .. bytes) = src[0 .. bytes) as one kernel on `stream` (16-byte aligned; src may be pinned host memory): the copies of a
 * replayed step - inputs, SpecAugment rows, a LayerDrop-dropped layer's pass-through (TF:models/wav2vec2/modeling_wav2vec2.py:
 * 709-723 skips the layer; the captured graphs of the neighbours read fixed buffers) - without the runtime's copy path */
int smx_copy_bytes(const void* src, void* dst, long long bytes, hipStream_t stream);

/* layer-weighted sum of the L+1 encoder hidden states (ref:speechmix/hf_model.py:411-423; ref:speechmix/model.py:150-157) */
typedef struct SmxWsumParams {
    const void* h[40]; const float* w; void* out; const void* dy; float *dots, *dw, *sw; long long n; int L1;
} SmxWsumParams;
int smx_weighted_sum_fwd(const SmxWsumParams* p, int dtype, hipStream_t stream);
int smx_weighted_sum_bwd(const SmxWsumParams* p, int dtype, hipStream_t stream);
int smx_axpy_dev(void* y, const void* x, const float* a, int idx, long long n, int init, int dtype, hipStream_t stream);

/* flat-buffer optimizer step (what HF Trainer's clip + optimizer.step do per tensor, ref:train.py:291-330) */
typedef struct SmxOptParams {
    float* p; const float* g; float *m, *v; void* shadow; const float* gnorm_sq; long long n;
    float lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2, grad_scale, max_grad_norm; int kind;
} SmxOptParams;
int smx_sumsq(const float* g, long long n, float* out, hipStream_t stream);
int smx_optimizer_step(const SmxOptParams* p, hipStream_t stream);

/* Adafactor over the flat buffer: the optimizer the reference trains with (ref:train.py:298 optim="adafactor" -> HF
 * Trainer: transformers.optimization.Adafactor(lr, scale_parameter=False, relative_step=False)); replaces the per-tensor
 * loop of TF:optimization.py Adafactor.step.  Work lists are built by the caller (speechmix_amd/ops.py AdafactorPlan). */
/* Every reduction of the step runs in a fixed order (no order-dependent fp32 atomics): data-parallel replicas that apply it to the
 * same all-reduced gradients keep bit-identical parameters.  tile0 / ntile: a tensor's (contiguous) tiles; cp_off / rp_off: where a
 * The global gradient norm for clipping (max_grad_norm > 0) is computed by the step itself from its statistics pass (gsq_part -> gn2).
 * tile's column partials / row sums go in `cpart` ([row tiles][C] and [column tiles][R] blocks per (tensor, leading index)). */
typedef struct SmxAfTensor { long long off; int nb, R, C, row_off, col_off, rm_off, factored, tile0, ntile, _pad; } SmxAfTensor;
typedef struct SmxAfTile { int tensor, b, r0, nr, c0, nc, full_rows, full_cols, cp_off, rp_off; } SmxAfTile;
typedef struct SmxAfSeg { int tensor, b, cp_off, n_rt, rp_off, n_ct; } SmxAfSeg;
typedef struct SmxAfParams {
    float* p; const float* g; void* shadow; const SmxAfTensor* tensors; const SmxAfTile* tiles; const SmxAfSeg* segs;
    float *row, *col, *racc, *cacc, *rmean, *usq, *usq_part, *cpart; const float* beta2t; float *gn2, *gsq_part;
    long long racc_n, cacc_n; int ntensors, ntiles, nsegs;
    float lr, eps1, clip_threshold, grad_scale, max_grad_norm;
} SmxAfParams;
/* racc / cacc / cpart / usq_part / gsq_part are STORE-ONLY scratch (every element read was stored by exactly one tile of the same
 * step): the caller need not zero them; no atomics anywhere in the step. */
int smx_adafactor_step(const SmxAfParams* p, hipStream_t stream);
/* The same step in phases (round 6): phase 0 = statistics pass over every tile + global norm + partial folds (no update); phase 1 = the two
 * update passes over tiles tile_first .. tile_first + tile_count - 1 (whole tensors); phase 2 = the statistics pass over a tile range, phase 3 = global
 * norm + partial folds behind phase-2 calls that covered every tile (0 = 2 over everything + 3).  Phase 0 followed by phase-1 calls covering every tile
 * once equals smx_adafactor_step bit for bit; the host may put the later ranges on another stream (the optimizer's tail beside the next
 * step's front end - what HF Trainer's optimizer.step() of TF:trainer.py cannot do). */
int smx_adafactor_phase(const SmxAfParams* p, int phase, int tile_first, int tile_count, hipStream_t stream);

/* Data-parallel gradient reduction: in-place sum-all-reduce of one contiguous bucket of the flat
 * gradient buffer over RCCL, on the caller's (side) stream - what HF Trainer -> accelerate -> DistributedDataParallel's
 * bucketed NCCL all-reduce does for the reference (TF:trainer.py:720-737).  comm: the caller's ncclComm_t; dtype SMX_F32 or
 * SMX_BF16; n in elements.  RCCL is resolved at run time (no link-time dependency): -38 (ENOSYS) when none can be loaded,
 * otherwise 0 or RCCL's ncclResult_t.  The Python host issues the same collective through torch.distributed (its process
 * group owns the communicator); this entry is for hosts that create their own. */
int smx_allreduce_bucket(void* comm, void* buf, size_t n, int dtype, hipStream_t stream);

/* On-box peak probes for the measurement contract (the datasheet peaks the roofline fractions use AND
 * what this box delivers): a pure v_mfma_f32_32x32x16_bf16 loop (returns the flops one launch issues; time it with events)
 * and a 16-B-per-lane streaming copy (read + write). */
double smx_probe_mfma(float* out, int blocks, int iters, hipStream_t stream);
/* the same loop on zero (zero = 1) or non-zero operands, with block 0's shader cycles / 100-MHz ticks over the loop in clk[0] / clk[1]
 * (device memory, may be null): the clock the chip sustains under the load is clk[0] / (10 clk[1]) GHz */
double smx_probe_mfma_clk(float* out, int blocks, int iters, int zero, unsigned long long* clk, hipStream_t stream);
int smx_probe_copy(const void* src, void* dst, long long bytes, hipStream_t stream);

/* Batched 2-D transposes of 16-bit matrices (round 6): dst[cols][rows] = src[rows][cols] for up to SMX_TR_MAX matrices per launch
 * (rows, cols multiples of 8, 16-byte aligned bases); tile0 / tcols are filled by the library.  Used for the K-contiguous copies of
 * the Linear weights the data gradients may read (speechmix_amd/engine.py Engine._wt; weights of ref:speechmix/model.py:148's encoder,
 * TF:models/wav2vec2/modeling_wav2vec2.py:466-572). */
#define SMX_TR_MAX 64
typedef struct SmxTrEntry { const void* src; void* dst; int rows, cols; int tile0, tcols; } SmxTrEntry;
typedef struct SmxTrTable { int n, tiles; SmxTrEntry e[SMX_TR_MAX]; } SmxTrTable;
int smx_transpose_many(const SmxTrTable* t, hipStream_t stream);
int smx_sizeof_SmxTrTable(void);
int smx_tr_max(void);

/* x[i] = T(float(x[i]) * *scale) in place, `scale` one fp32 word in device memory: the seed of backward scaled by the device scalar
 * autograd hands to loss.backward() (1 / k under the reference's gradient accumulation, ref:train.py:291-330) without a host read. */
int smx_scale_dev(void* x, long long n, const float* scale, int dtype, hipStream_t stream);

/* Per-step dropout key (round 5): every translation unit whose kernels hash dropout masks keeps one device word that
 * smx_set_step_key rewrites on the stream; these return the word's device address (diagnostics; smx_set_step_key collects them). */
int smx_step_key_addr_misc(void** out);
int smx_step_key_addr_gemm(void** out);
int smx_step_key_addr_gemm_pp(void** out);
int smx_step_key_addr_gemm_fr(void** out);
int smx_step_key_addr_gemm_ws(void** out);
int smx_step_key_addr_norm(void** out);
int smx_step_key_addr_attention(void** out);

/* ABI self-description */
int smx_sizeof_SmxGemmParams(void);
int smx_sizeof_SmxNormParams(void);
int smx_sizeof_SmxNormBwdParams(void);
int smx_sizeof_SmxAttnParams(void);
int smx_sizeof_SmxConv0Params(void);
int smx_sizeof_SmxCEParams(void);
int smx_sizeof_SmxOptParams(void);
int smx_sizeof_SmxWsumParams(void);
int smx_sizeof_SmxAfParams(void);
int smx_sizeof_SmxAfTensor(void);
int smx_sizeof_SmxAfTile(void);
int smx_sizeof_SmxAfSeg(void);

#ifdef __cplusplus
}
#endif
#endif
