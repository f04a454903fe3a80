/*
 * dta.h — C ABI of libdta_mi355x.so: the MI355X (gfx950) tree-attention hot path of DynamicTreeAttn.
 *
 * The reference (/root/reference, 17 Python files) has NO native code and therefore no FFI; these
 * entry points are what a binding for its hot path would call.  Each one names the reference
 * interface it replaces (file:line into /root/reference).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions: raw device pointers + sizes + a hipStream_t passed as void*;
 * returns 0 on success, a negative DTA_E* code on invalid arguments (nothing is launched then);
 * no allocation, no ownership transfer, no global state, re-entrant across streams.  All
 * launches are asynchronous on `stream`.  Device code exists for gfx950 only.
 *
 * Packed-trie vocabulary: the T tokens of a trie are laid out in DFS pre-order of its leaves
 * ("packed order"): leaf i contributes the segment of its tokens at depths [lcp[i-1], len[i]).
 * For packed token s, subtree_end[s] is one past its last descendant, so
 *     s is an ancestor-or-self of t   <=>   s <= t < subtree_end[s].
 */
#ifndef DTA_H
#define DTA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTA_OK 0
#define DTA_EINVAL (-1)      /* null pointer / negative size / inconsistent sizes */
#define DTA_EUNSUPPORTED (-2)/* head_dim != 128, dtype not bf16/f16, Hq % Hkv != 0 ...  */
#define DTA_EALIGN (-3)      /* pointer or stride not 16-byte aligned */
#define DTA_ELAUNCH (-4)     /* hipGetLastError() after the launch was not hipSuccess */
#define DTA_EPRIOR (-5)      /* a HIP error was ALREADY pending on this thread when the entry point was called (an earlier
                              * asynchronous kernel fault or an unchecked runtime call): nothing was launched and the error
                              * is left in place; dta_take_pending_error() names and clears it */

#define DTA_BF16 0
#define DTA_F16 1
#define DTA_F32 2            /* fp32 models (reference --dtype fp32, run.py:122-132): plain-FMA attention kernels (a gradient-check path, not a
                              * performance path), fp32 row kernels; and fp32 logits of the log-prob / entropy kernels (vocab_parallel.py:16,24) */

#define DTA_QTILE 128        /* query rows per workgroup (fwd / dQ kernels)  */
#define DTA_KTILE 128        /* key rows per workgroup (dK/dV kernel): 8 waves = 2 groups x (4 waves x 32 keys) sharing the keys */

int dta_version(void);

/* Name (into msg[cap], NUL-terminated) and CLEAR the HIP error pending on the calling thread; returns its hipError_t value
 * (0 = none).  After an asynchronous kernel fault the context stays unusable - clearing only lets the caller report it. */
int dta_take_pending_error(char* msg, int32_t cap);

/* ---------------------------------------------------------------------------------------------
 * Trie build kernels (integer, bit-exact, HBM-bound)
 * ------------------------------------------------------------------------------------------- */

/* Adjacent longest-common-prefix of S sequences held in one int64 `tokens` buffer: sequence i (in
 * the order to be compared) is tokens[starts[i] .. starts[i]+lens[i]).  out_lcp[i] = lcp(seq i,
 * seq i+1); *out_unsorted += number of adjacent pairs that violate lexicographic order
 * (seq_i[lcp] > seq_{i+1}[lcp]).  The caller zeroes out_unsorted.  S = 0 or 1: nothing to compare, DTA_OK.
 * Replaces token_trie.py:6-10 (_lcp_torch), the order check token_trie.py:24-30 and the
 * recomputation after a permutation, token_trie.py:94.  */
int dta_lcp_adjacent(const int64_t* tokens, const int64_t* starts, const int32_t* lens, int32_t S,
                     int32_t* out_lcp, int32_t* out_unsorted, void* stream);

/* Leafization as a stream compaction over the S sorted sequences: keep[i] = (i == S-1) ||
 * lcp[i] < min(len[i], len[i+1]).  Writes the kept positions (ascending) to out_leaf_pos, the
 * leaf's LCP with the next leaf to out_leaf_lcp, for every sequence the leaf it folds onto to
 * out_seq_leaf[S] (lens[S] = sequence lengths in sorted order), and the leaf count to *out_M.  One workgroup (S <= 2^20).
 * Replaces token_trie.py:32-49 (_leafization, second half).  */
int dta_leafize(const int32_t* lens, const int32_t* lcp, int32_t S,
                int32_t* out_leaf_pos, int32_t* out_leaf_lcp, int32_t* out_seq_leaf, int32_t* out_M,
                void* stream);

/* Packed pre-order metadata for M leaves visited in the given DFS order.  Inputs per leaf i:
 * seg_off[i] (packed offset of its segment; seg_off[M] = T), seg_depth0[i] = lcp[i-1] (0 for i=0),
 * leaf_tok_off[i] = offset of the leaf's tokens in `tokens`; and the per-segment "closing" table
 * brk_ptr[M+1], brk_depth[], brk_end[]: tokens of segment i at depth d in
 * [brk_depth[j], brk_depth[j+1]) have subtree_end = brk_end[j].  parent_of_seg[i] = packed index of
 * the token at depth seg_depth0[i]-1 on leaf i's path (-1 when seg_depth0[i] == 0).
 * Outputs per packed token: token id, depth (= RoPE position, the stack position of
 * tree_training_engine.py:166,293), parent index, subtree_end.
 * Replaces the per-leaf H2D copy + stack bookkeeping of tree_training_engine.py:536-548, 582-611. */
int dta_preorder_meta(const int64_t* tokens, const int64_t* leaf_tok_off,
                      const int32_t* seg_off, const int32_t* seg_depth0, const int32_t* parent_of_seg,
                      const int32_t* brk_ptr, const int32_t* brk_depth, const int32_t* brk_end,
                      int32_t M, int32_t T,
                      int64_t* out_token, int32_t* out_depth, int32_t* out_parent, int32_t* out_subtree_end,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Tree attention (MFMA-bound).  head_dim must be 128; dtype DTA_BF16 or DTA_F16 (MFMA kernels), or DTA_F32 (every buffer fp32;
 * plain fp32 FMAs, one workgroup per 64 rows, split-Q work units ignored - the correctness path of fp32 models).
 *
 * q/out/dout/dq: [Tq, Hq, 128] with element strides (q_stride_t, 128); k/v/dk/dv: [Tk, Hkv, 128]
 * with (kv_stride_t, 128).  Query row i has packed index q_offset + i.  It attends key s iff
 * s <= q_offset+i  &&  q_offset+i < subtree_end[s]   (subtree_end == NULL: no upper bound, i.e. the
 * rectangular-causal stack form of tree_training_engine.py:171-186 with q_offset = start).
 *
 * Query tiles are DTA_QTILE rows.  Tile j visits the key runs runs[run_ptr[j] .. run_ptr[j+1]),
 * each run = 4 int32 {key_begin, key_end, needs_mask, 0}; runs == NULL: one run [0, last row + 1).
 * A run with needs_mask == 0 promises that every key in it is visible to every row of the tile.
 * lse: [Hq, Tq] float (head-major), log2-domain log-sum-exp of the scaled scores (natural lse = lse * ln 2).
 * Replaces the attention the reference reaches through the model call,
 * tree_training_engine.py:182-186, 248-252, 351-353 (third-party attention backend).  */
int dta_tree_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                      const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                      int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                      int64_t q_stride_t, int64_t kv_stride_t, int64_t o_stride_t,
                      float scale, int32_t dtype, void* stream);

/* General-stride form of dta_tree_attn_fwd: explicit head strides (elements) so that head-major
 * layouts such as the reference's [1, H, S, D] KV stack (tree_training_engine.py:108-131) work in place.
 * Token strides of k and v (forward) and of q and dout (backward) must be in [0, 2^24] elements: a 64-row tile
 * is addressed as scalar base + 32-bit lane offset by the tile DMA (DTA_EUNSUPPORTED otherwise). */
int dta_tree_attn_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse,
                         const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                         int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                         int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h,
                         int64_t v_stride_t, int64_t v_stride_h,
                         int64_t o_stride_t, int64_t o_stride_h, float scale, int32_t dtype, void* stream);

/* Backward.  Two launches on `stream`: (1) per query tile: delta = rowsum(dout*out), dq;
 * (2) per key tile of DTA_KTILE keys: dk, dv summed over the query range
 * [max(key0, q_offset), ktile_qend[tile]) and over the Hq/Hkv query heads of the group — no
 * cross-workgroup reduction, no atomics, bitwise reproducible.  ktile_qend[j] = max subtree_end
 * over the tile's keys (NULL: q_offset + Tq).  `accumulate`: 0 overwrites dk/dv; 1 adds into them (the grad-KV
 * stack of tree_training_engine.py:447-451; model dtype, rounded after every add as the reference's `+=`); 2 adds into
 * FP32 buffers (dk/dv are float*, strides in floats) so that the hundreds of adds a root-side row receives in the
 * block-wise engine are not rounded to 16 bits each time.  delta: [Hq, Tq] float workspace.
 * Replaces torch.autograd.backward through the attention backend, tree_training_engine.py:440.  */
int dta_tree_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                      const float* lse, float* delta, void* dq, void* dk, void* dv,
                      const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                      const int32_t* ktile_qend,
                      int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                      int64_t q_stride_t, int64_t kv_stride_t, int64_t o_stride_t,
                      int64_t dq_stride_t, int64_t dkv_stride_t,
                      float scale, int32_t dtype, int32_t accumulate, void* stream);

int dta_tree_attn_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout,
                         const float* lse, float* delta, void* dq, void* dk, void* dv,
                         const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs,
                         const int32_t* ktile_qend,
                         int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv, int32_t head_dim,
                         int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h,
                         int64_t v_stride_t, int64_t v_stride_h,
                         int64_t o_stride_t, int64_t o_stride_h, int64_t dq_stride_t, int64_t dq_stride_h,
                         int64_t dkv_stride_t, int64_t dkv_stride_h,
                         float scale, int32_t dtype, int32_t accumulate,
                         int32_t which /* bit0: delta+dq launch, bit1: dk/dv launch (needs delta) followed by the slab finalize
                                          unless bit3; bit2: slab finalize alone (lets a profiler bracket each launch) */,
                         /* optional split of the dK/dV sweep into balanced work units (NULL: one per key tile):
                          * dkv_units[u] = {key tile, q_begin, q_end (packed), slab or -1}; units of a split key tile
                          * write fp32 slabs [2][DTA_KTILE][128] into dkv_ws (slab-major, then kv head) which a finalize
                          * launch sums in order: dkv_splits[s] = {key tile, first slab, n slabs, 0}. */
                         const int32_t* dkv_units, int32_t n_units, const int32_t* dkv_splits, int32_t n_splits, float* dkv_ws,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Log-prob / entropy over vocabulary rows (HBM-bound).  logits: [R, V] bf16 / f16 / f32 with row_stride elements
 * between rows (multiple of 8; base 16-byte aligned, 32-byte for f32).  All statistics are fp32.
 * fwd writes lse[r] = ln sum_j exp(x_j/T), entropy[r] (may be NULL) and logprob[r] = x[labels[r]]/T - lse[r] (may be
 * NULL; a label outside [0, V) yields 0).  EXTRA picks: a trie node with several children predicts one token per child
 * (the fork-position logits of tree_training_engine.py:205-209, 217-220, 369-372).  They come as a CSR over the rows:
 * extra_ptr[R+1] (int32, absolute indices into the extra arrays; NULL = none), extra_labels[F]; the forward writes
 * extra_logprob[f] = x[extra_labels[f]]/T - lse[row of f].
 * bwd writes dLoss/dlogits to `dlogits` (== logits: in place) given g_logprob[r], g_extra_logprob[F] and g_entropy[r]
 * (each may be NULL), including the one-hot terms of every pick.
 * Replaces vocab_parallel.py:13-27 (_gather_logprobs[_entropy]) with its autograd backward, and the torch indexing of
 * the fork rows.  */
int dta_logprob_entropy_fwd(const void* logits, const int64_t* labels, const int32_t* extra_ptr, const int64_t* extra_labels,
                            float* lse, float* entropy, float* logprob, float* extra_logprob,
                            int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream);
/* Vocab-sharded forward: raw per-shard statistics stats[R][4] = {m, s, t, picked} (log2 domain of x*log2(e)/T;
 * labels shard-local, -1 = owned by another rank; extra_picked[F] likewise raw x/T or 0) for the cross-rank combine of
 * vocab_parallel.py:125-160, 258-300. */
int dta_logprob_entropy_shard_stats(const void* logits, const int64_t* labels, const int32_t* extra_ptr, const int64_t* extra_labels,
                                    float* stats, float* extra_picked,
                                    int32_t R, int32_t V, int64_t row_stride, float temperature, int32_t dtype, void* stream);
int dta_logprob_entropy_bwd(const void* logits, void* dlogits, const int64_t* labels, const int32_t* extra_ptr, const int64_t* extra_labels,
                            const float* lse, const float* entropy,
                            const float* g_logprob, const float* g_extra_logprob, const float* g_entropy,
                            int32_t R, int32_t V, int64_t row_stride, int64_t out_row_stride, float temperature, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused row kernels of the decoder layer (HBM-bound; bf16/f16 storage, fp32 math).  They restate the
 * arithmetic of the third-party Qwen3 layers the reference calls (tree_training_engine.py:182-186,
 * 248-252, 351-353): RMSNorm = w * cast(x * rsqrt(mean(x^2)+eps)); rotate-half RoPE at position =
 * trie depth; SwiGLU = cast(silu(g)) * u.
 * ------------------------------------------------------------------------------------------- */
/* delta/x_out both NULL: y = norm(x).  Both given: x_out = x + delta (the residual-stream update of the decoder
 * layer, rounded to the storage dtype) and y = norm(x_out), in one pass.  bwd: `x` is the normalised input
 * (x_out of the forward), `dres` (may be NULL) the gradient arriving on the residual stream, added to dx. */
int dta_rmsnorm_fwd(const void* x, const void* delta, const void* w, void* x_out, void* y, float* rstd,
                    int32_t R, int32_t H, float eps, int32_t dtype, void* stream);
int dta_rmsnorm_bwd_blocks(int32_t R);   /* rows of the dw_partial workspace [blocks, H] (float); caller sums dim 0.  H % 8 == 0; bwd: H <= 8192 */
int dta_rmsnorm_bwd(const void* x, const void* w, const void* dy, const void* dres, const float* rstd, void* dx, float* dw_partial,
                    int32_t R, int32_t H, int32_t dtype, void* stream);
/* x: [T, NH, 128] with token stride x_stride_t; cos_sin: float [T, 128] = {cos[64], sin[64]} of the token's
 * depth; y: [T, NH, 128] contiguous; w (head-norm weight [128]) may be NULL = RoPE only. */
int dta_qk_norm_rope_fwd(const void* x, const void* w, const float* cos_sin, void* y, float* rstd,
                         int32_t T, int32_t NH, int32_t head_dim, int64_t x_stride_t, float eps, int32_t dtype, void* stream);
int dta_qk_norm_rope_bwd_blocks(int64_t n_heads_total);   /* rows of dw_partial [blocks, 128] */
int dta_qk_norm_rope_bwd(const void* x, const void* w, const float* cos_sin, const void* dy, const float* rstd,
                         void* dx, float* dw_partial, int32_t T, int32_t NH, int32_t head_dim,
                         int64_t x_stride_t, int64_t dy_stride_t, int64_t dy_stride_h, int64_t dx_stride_t,
                         int32_t dtype, void* stream);   /* dx: [T, NH, 128] with dx_stride_t elements between tokens (>= NH*128); dx == dy (same strides) is allowed: in place */
/* gate/up: [rows, cols] with `ld` elements between rows (they may be the two halves of one fused [rows, 2*cols]
 * projection output); y/dy: [rows, cols] contiguous; dgate/dup: `ld_grad` between rows. */
int dta_swiglu_fwd(const void* gate, const void* up, void* y, int64_t rows, int32_t cols, int64_t ld, int32_t dtype, void* stream);
int dta_swiglu_bwd(const void* gate, const void* up, const void* dy, void* dgate, void* dup,
                   int64_t rows, int32_t cols, int64_t ld, int64_t ld_grad, int32_t dtype, void* stream);

/* out[c][r] = in[r][c]: `rows` x `cols` elements of `elem_size` bytes (2: bf16 / f16, 4: f32), `ld_in` / `ld_out` elements between rows (both
 * multiples of 8 elements, pointers 16-byte aligned).  HBM-bound (one read + one write).  Used for transposed copies of the projection and
 * LM-head weights, made once per weight version: the input-gradient GEMMs dx = dy . W of the model calls (tree_training_engine.py:440,
 * torch.autograd.backward) run 12-25 % faster with the contraction index contiguous in both operands. */
int dta_transpose(const void* in, void* out, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int32_t elem_size, void* stream);

/* out[i] = round_to(out_dtype)( sum_{s < slabs} part[s * slab_stride + i]  + (extra ? extra[i] : 0) ),  i < n; all sums in fp32.
 * The reduction of weight-gradient partials fused with the rounding to the parameter dtype: the per-workgroup dw partials of
 * dta_rmsnorm_bwd / dta_qk_norm_rope_bwd (slabs = workgroups, n = H or 128) and the slices of the split-K weight-gradient GEMM
 * (slabs = the split, n = out*in, extra = the product of the rows the equal slices leave over).  Stands where the reference's
 * autograd sums a weight's gradient over all rows of a model call in one GEMM (tree_training_engine.py:440).  `out` may not alias
 * `part` or `extra`.  Any n; the vector form is taken when slabs <= 16 and n, slab_stride are multiples of 4 with 16-byte aligned pointers. */
int dta_sum_slabs(const float* part, int64_t slabs, int64_t n, int64_t slab_stride, const float* extra, void* out, int32_t out_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DTA_H */
