/*
 * sealnn.h -- fused decoder-step kernels for the BART step decoder (gfx950), part of libsealfm.so: fp32 storage (the
 * arithmetic the reference runs BART in) and, with the suffix _bf16, bf16 storage (BASELINE.json configs[4]) -- the same
 * kernels instantiated for 2-byte elements: loads widen to fp32, every product / sum / softmax is fp32, stores round.
 * They replace runs of small PyTorch kernels inside seal_amd/bart_decoder.py's hipGraph-captured step
 * (the reference drives HF's BartDecoderLayer through generate(); reference seal/beam_search.py:231-238).
 * Device pointers + hipStream_t (as void*), no allocation, no synchronisation: capture-safe.
 */
#ifndef SEALNN_H
#define SEALNN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* single-position self-attention with KV-cache append.
 *   qkv    [rows, 3, heads, 64] fp32 (q, k, v of the new position; q already scaled or scale passed)
 *   kcache, vcache [rows, heads, T, 64]; position *d_t is written, positions 0..*d_t attended
 *   out    [rows, heads*64]
 *   anc    NULL, or [T, rows] int32: the cache is addressed through it -- position p of `row`'s history
 *          is slot p of row anc[p][row] -- and anc[*d_t][row] = row is written.  Re-ranking beams
 *          (HF _reorder_cache) then permutes the columns of this table instead of the cache. */
int sealnn_self_attn_step(void *stream, const float *qkv, float *kcache, float *vcache, const int64_t *d_t,
                          uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out, int32_t *anc);

/* single-position cross-attention, encoder K/V shared by the `beams` rows of a query.
 *   q [batch*beams, heads, 64]; ck [batch, heads, 64, S]; cv [batch, heads, S, 64]; bias [batch, S] (0 / -big)
 *   out [batch*beams, heads*64];  S <= 64 */
int sealnn_cross_attn_step(void *stream, const float *q, const float *ck, const float *cv, const float *bias,
                           uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S, float scale, float *out);

/* out = LayerNorm(x + y) * gamma + beta over the last dimension d (multiple of 4, <= 4096); eps as torch */
int sealnn_add_layernorm(void *stream, const float *x, const float *y, const float *gamma, const float *beta,
                         uint32_t rows, uint32_t d, float eps, float *out);

/* teacher-forced causal self-attention: qkv [n_seq, T, 3, heads, 64] -> out [n_seq, T, heads*64]; T <= 17 */
int sealnn_causal_self_attn(void *stream, const float *qkv, uint32_t n_seq, uint32_t T, uint32_t heads, float scale, float *out);

/* The same attention over a prefix TREE (rescoring, reference keys.py:64-141: the keys of a query share prefixes and the
 * decoder is causal, so every distinct prefix is ONE decoder position): node i attends the nodes anc[i][0 .. depth_i]
 * (its ancestors from the root down, then itself; -1 beyond its depth; max_depth1 <= 17 columns).  qkv [n_nodes, 3 * heads * 64],
 * out [n_nodes, heads * 64].  Same arithmetic, in the same order, as sealnn_causal_self_attn for a row holding that prefix. */
int sealnn_tree_self_attn(void *stream, const float *qkv, const int32_t *anc, uint32_t n_nodes, uint32_t max_depth1, uint32_t heads,
                          float scale, float *out);

/* cross-attention of arbitrary rows: row_batch[row] = query whose encoder K/V (layouts as above) the row attends */
int sealnn_cross_attn_rows(void *stream, const float *q, const float *ck, const float *cv, const float *bias,
                           const int32_t *row_batch, uint32_t rows, uint32_t heads, uint32_t S, float scale, float *out);

/* the same, one workgroup per (`group` consecutive rows, head): the K/V of the run's FIRST row's query are staged in LDS once and every
 * row of that query reads them there; a row of another query reads its own from memory (so any row_batch is served; rows that come in
 * runs per query -- teacher forcing, the nodes of a query's prefix tree -- are served fast); the last run may be short.  Bit-identical
 * results. */
int sealnn_cross_attn_runs(void *stream, const float *q, const float *ck, const float *cv, const float *bias,
                           const int32_t *row_batch, uint32_t rows, uint32_t group, uint32_t heads, uint32_t S, float scale, float *out);

/* ---- bf16 storage: same shapes and meaning, every tensor argument (incl. gamma / beta / bias) bf16 ---- */
int sealnn_self_attn_step_bf16(void *stream, const void *qkv, void *kcache, void *vcache, const int64_t *d_t,
                               uint32_t rows, uint32_t heads, uint32_t T, float scale, void *out, int32_t *anc);
int sealnn_cross_attn_step_bf16(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                                uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S, float scale, void *out);
int sealnn_add_layernorm_bf16(void *stream, const void *x, const void *y, const void *gamma, const void *beta,
                              uint32_t rows, uint32_t d, float eps, void *out);
int sealnn_causal_self_attn_bf16(void *stream, const void *qkv, uint32_t n_seq, uint32_t T, uint32_t heads, float scale, void *out);
int sealnn_tree_self_attn_bf16(void *stream, const void *qkv, const int32_t *anc, uint32_t n_nodes, uint32_t max_depth1, uint32_t heads,
                               float scale, void *out);
int sealnn_cross_attn_rows_bf16(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                                const int32_t *row_batch, uint32_t rows, uint32_t heads, uint32_t S, float scale, void *out);
int sealnn_cross_attn_runs_bf16(void *stream, const void *q, const void *ck, const void *cv, const void *bias,
                                const int32_t *row_batch, uint32_t rows, uint32_t group, uint32_t heads, uint32_t S, float scale, void *out);

/* ---- operands of the split GEMM (fp32-accurate linear layers on the fp16 matrix cores; seal_amd/split_gemm.py) ----
 *   x [rows, K] fp32 -> out [rows, 3K] fp16 = [hi | hi | lo'],  hi = fp16(x), lo' = fp16((x - hi) * 2^11);  K % 4 == 0
 *   *d_flag (may be NULL) += the number of 4-element groups holding a finite |x| > 65504 (not representable: the caller must check) */
int sealnn_split_planes(void *stream, const float *x, uint32_t rows, uint32_t K, void *out, uint32_t *d_flag);
/* the same planes written by the kernel that PRODUCES the activation (no separate pass): LayerNorm(x + y) as fp32 `out` (the residual
 * stream) and as `planes` [rows, 3d] (the next projection's operand);  gelu(x) (erf form, torch's arithmetic) as planes only (fc2's operand) */
int sealnn_add_layernorm_planes(void *stream, const float *x, const float *y, const float *gamma, const float *beta, uint32_t rows,
                                uint32_t d, float eps, float *out, void *planes, uint32_t *d_flag);
int sealnn_gelu_planes(void *stream, const float *x, uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag);
/* The same kernels reading the RAW fp32 accumulators of a split GEMM (torch.mm(planes, W^T, out_dtype=float32)): the projection's output is
 * alpha * acc + bias -- the GEMM's epilogue, applied on read by the kernel that consumes it instead of as a pass over the output (which is
 * what torch.addmm(out_dtype=float32) costs: a copy of the broadcast bias in front of every product).  bias: [3 * heads * 64] for the qkv
 * projections, [d] otherwise; alpha a power of two (split_gemm.split_weight), so alpha * acc is exact and the result is bit for bit the
 * epilogue's.  `planes` of sealnn_add_layernorm_acc may be NULL. */
int sealnn_self_attn_step_acc(void *stream, const float *qkv_acc, const float *qkv_bias, float alpha, float *kcache, float *vcache,
                              const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out, int32_t *anc);
int sealnn_tree_self_attn_acc(void *stream, const float *qkv_acc, const float *qkv_bias, float alpha, const int32_t *anc, uint32_t n_nodes,
                              uint32_t max_depth1, uint32_t heads, float scale, float *out);
int sealnn_add_layernorm_acc(void *stream, const float *x, const float *y_acc, const float *y_bias, float alpha, const float *gamma,
                             const float *beta, uint32_t rows, uint32_t d, float eps, float *out, void *planes, uint32_t *d_flag);
int sealnn_gelu_planes_acc(void *stream, const float *x_acc, const float *x_bias, float alpha, uint32_t rows, uint32_t d, void *planes,
                           uint32_t *d_flag);
/* sealnn_self_attn_step / sealnn_cross_attn_step BETWEEN two hand-written products (sealnn_hgemm_nt): the projection in front arrives as raw
 * accumulators -- n_slabs split-K slabs, slab s at acc + s * slab_stride floats, the projection = alpha * (slab 0 + slab 1 + ...) + bias
 * (q_bias of the cross form may be NULL: q_acc is then the finished projection) -- and the result leaves as the split planes of the projection
 * that follows ([rows][3 * heads * 64] fp16 = [hi | hi | lo * 2^11], as sealnn_split_planes writes them; d_flag counts rows beyond fp16's range)
 * and / or as fp32 `out` (either may be NULL, not both).  Same arithmetic, in the same order, as the plain kernels on the finished operand. */
int sealnn_self_attn_step_x(void *stream, const float *qkv_acc, uint32_t n_slabs, uint64_t slab_stride, const float *qkv_bias, float alpha,
                            float *kcache, float *vcache, const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out,
                            void *out_planes, uint32_t *d_flag, int32_t *anc);
int sealnn_cross_attn_step_x(void *stream, const float *q_acc, uint32_t n_slabs, uint64_t slab_stride, const float *q_bias, float alpha,
                             const float *ck, const float *cv, const float *bias, uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S,
                             float scale, float *out, void *out_planes, uint32_t *d_flag);
/* sealnn_gelu_planes_acc over the n_slabs slabs of fc1 as a split-K product: x = alpha * (slab 0 + slab 1 + ...) + bias, added in slab order. */
int sealnn_gelu_planes_acc_slabs(void *stream, const float *x_acc, uint32_t n_slabs, uint64_t slab_stride, const float *x_bias, float alpha,
                                 uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag);
/* The PAIRS operand (round 6): the planes of an activation as [hi of 32 columns | lo * 2^11 of the same 32 columns] per 128-byte line, rows of 2 d halves --
 * two thirds of the three-block operand, for sealnn_hgemm_nt's PAIRS products (config bit 29: three products per K step from four tiles instead of six, of
 * which two were copies).  The same kernels as their namesakes, only the layout of `planes` / `out_planes` differs (d a multiple of 32). */
int sealnn_split_planes_pairs(void *stream, const float *x, uint32_t rows, uint32_t K, void *out, uint32_t *d_flag);
int sealnn_add_layernorm_acc_slabs_pairs(void *stream, const float *x, const float *y_acc, uint32_t n_slabs, uint64_t slab_stride, const float *y_bias,
                                         float alpha, const float *gamma, const float *beta, uint32_t rows, uint32_t d, float eps, float *out,
                                         void *planes, uint32_t *d_flag);
int sealnn_gelu_planes_acc_slabs_pairs(void *stream, const float *x_acc, uint32_t n_slabs, uint64_t slab_stride, const float *x_bias, float alpha,
                                       uint32_t rows, uint32_t d, void *planes, uint32_t *d_flag);
int sealnn_self_attn_step_x_pairs(void *stream, const float *qkv_acc, uint32_t n_slabs, uint64_t slab_stride, const float *qkv_bias, float alpha,
                                  float *kcache, float *vcache, const int64_t *d_t, uint32_t rows, uint32_t heads, uint32_t T, float scale, float *out,
                                  void *out_planes, uint32_t *d_flag, int32_t *anc);
int sealnn_cross_attn_step_x_pairs(void *stream, const float *q_acc, uint32_t n_slabs, uint64_t slab_stride, const float *q_bias, float alpha,
                                   const float *ck, const float *cv, const float *bias, uint32_t batch, uint32_t beams, uint32_t heads, uint32_t S,
                                   float scale, float *out, void *out_planes, uint32_t *d_flag);
/* out[rows][n] = alpha * (slab 0 + slab 1 + ...) + bias: a product of sealnn_hgemm_nt finished for a consumer that is not one of these kernels
 * (torch's fused attention in the encoder; reference: the bias add of every nn.Linear of modeling_bart). */
int sealnn_finish_product(void *stream, const float *acc, uint32_t n_slabs, uint64_t slab_stride, const float *bias, float alpha, uint32_t rows,
                          uint32_t n, float *out);
/* sealnn_add_layernorm_acc whose addend arrives as the n_slabs slabs of a split-K product (sealnn_hgemm_nt with slices > 1: slab s at
 * y_acc + s * slab_stride floats): y = alpha * (slab 0 + slab 1 + ...) + bias, the slabs added in slab order as they are read. */
int sealnn_add_layernorm_acc_slabs(void *stream, const float *x, const float *y_acc, uint32_t n_slabs, uint64_t slab_stride, const float *y_bias,
                                   float alpha, const float *gamma, const float *beta, uint32_t rows, uint32_t d, float eps, float *out,
                                   void *planes, uint32_t *d_flag);

/* C[M][N] (fp32, row stride ldc) = A[M][K] (fp16, K contiguous) x W[N][K]^T (fp16, K contiguous), fp32 accumulation on the fp16 matrix cores
 * of gfx950: the linear layers of a decode step (reference seal/beam_search.py:231-253 runs them through torch.nn.Linear) as ONE product over
 * the three split planes of an fp32 operand (K = 3 x in_features; sealnn_*_planes write A, seal_amd/split_gemm.py W).  A hand-written kernel
 * for the decode's heights (M = 300 .. 640 rows, a few hundred workgroups): LDS-DMA staging, no stream-K hand-off between workgroups.
 * K % 64 == 0, operands 16-byte aligned.  config: 0 = tile picked by shape; probes / tests: tile (1: 128 x 128, 2: 64 x 64, 3: 128 x 64,
 * 4: 64 x 128; 5: 320 x 128, 6: 320 x 64 -- the tall tiles of 8 waves, for 300 / 600 rows) | stages << 8 (LDS stages 1..3, 0: two) | 1 << 29 (the operands are PAIRS planes, K = 2 x in_features) | slices << 16 (split-K: slab s at C + s * M * ldc, the caller sums the slabs). */
int sealnn_hgemm_nt(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config);
/* The same product FINISHED in the kernel's store: c[row][col] = alpha * acc + bias[row / rows_per_bias_row][col] (bias: [ceil(M / rows_per_bias_row)][N] fp32).
 * One slab, the PAIRS form of the tall tiles (5..7) only -- the output projection of a decode step: bias = final_logits_bias + the per-query logit bias,
 * rows_per_bias_row = beams (reference beam_search.py:246-253: lm_head, then the logits processors' additive terms). */
int sealnn_hgemm_nt_ep(void *stream, const void *a, const void *w, float *c, uint32_t M, uint32_t N, uint32_t K, uint64_t ldc, uint32_t config,
                       const float *bias, uint32_t rows_per_bias_row, float alpha);

#ifdef __cplusplus
}
#endif
#endif
