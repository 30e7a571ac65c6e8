/*
 * hstu_hip.h -- C ABI of libhstu_hip.so, the MI355X (gfx950) HSTU hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch types.  Every
 * entry point cites the reference interface it replaces (paths relative to
 * /root/reference/generative_recommenders).  All functions are stream-ordered
 * (launch on `stream`, a hipStream_t passed as void*; NULL = default stream),
 * never synchronise the host, never allocate, borrow their inputs and write only
 * to the output / workspace pointers they are given.  Return 0 on success, a
 * negative HSTU_E* code otherwise; hstu_last_error() holds the message (the
 * analogue of the reference's TORCH_CHECK text, ops/cpp/hstu_attention/
 * flash_common.cpp:339-456).
 *
 * Tensors are jagged: row r of user b lives at row (seq_offsets[b] + r) of a
 * (total_rows, heads, head_dim) array whose last dimension is contiguous; row
 * and head strides are given in ELEMENTS and are arbitrary (q/k/v may be views
 * of one fused uvqk buffer, triton_hstu_preprocess_and_attention.py:200-252),
 * but every (row, head) vector must start 16-byte aligned.
 */
#ifndef HSTU_HIP_H_
#define HSTU_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HSTU_ABI_VERSION 13

enum {
  HSTU_OK = 0,
  HSTU_EINVAL = -1,      /* bad argument (shape, alignment, dtype) */
  HSTU_EUNSUPPORTED = -2,/* head dim / dtype combination not instantiated */
  HSTU_ELAUNCH = -3,     /* HIP launch error */
};

enum { HSTU_DTYPE_BF16 = 0, HSTU_DTYPE_F16 = 1, HSTU_DTYPE_F32 = 2 };
enum { HSTU_INDEX_I32 = 0, HSTU_INDEX_I64 = 1 };

/*
 * Attention problem description, shared by forward and backward.  Models the
 * argument list of hstu_mha (ops/hstu_attention.py:44-61) /
 * hstu::hstu_mha_fwd|bwd (ops/cpp/hstu_attention/flash_api.cpp:275-352) and the
 * POD role of Flash_fwd_params (ops/cpp/hstu_attention/flash.h:23-141).
 *
 *   P[i,j] = silu(alpha * <q_i, k_j>) * scale * M[i,j],   O = P V
 *
 * scale = 1/max_seq_len in the reference (pt_hstu_attention.py:151); M is the
 * causal/target/window/contextual mask of pt_hstu_attention.py:32-84.
 */
typedef struct HstuAttnParams {
  /* data */
  const void* q;          /* (total_q_rows, H, dqk) */
  const void* k;          /* (total_rows,   H, dqk) */
  const void* v;          /* (total_rows,   H, dv)  */
  void* out;              /* fwd: (total_q_rows, H, dv), written */
  const void* seq_offsets;/* (B+1) int32|int64, device */
  const void* num_targets;/* (B) int32|int64 or NULL */
  int64_t q_row_stride, q_head_stride;
  int64_t k_row_stride, k_head_stride;
  int64_t v_row_stride, v_head_stride;
  int64_t o_row_stride, o_head_stride;
  /* shape */
  int32_t batch;          /* B */
  int32_t heads;          /* H */
  int32_t dqk, dv;        /* head dims; multiples of 8 (16-bit) / 4 (fp32), <= 128 */
  int32_t max_seq_len;    /* N: launch bound on per-user length (grid size) */
  int32_t delta_q;        /* 0: q is jagged like k/v.  >0: q holds the last
                             delta_q rows of every user, densely (B*delta_q rows):
                             delta_hstu_mha, ops/hstu_attention.py:131-203 */
  /* semantics */
  float alpha;
  float scale;            /* 1/max_seq_len of the CALLER (not necessarily 1/N above) */
  int32_t max_attn_len;
  int32_t contextual_seq_len;
  int32_t min_full_attn_seq_len;
  /* types */
  int32_t dtype;          /* HSTU_DTYPE_* of q,k,v,out,dout,dq,dk,dv */
  int32_t offsets_dtype;  /* HSTU_INDEX_* of seq_offsets */
  int32_t targets_dtype;  /* HSTU_INDEX_* of num_targets */
  /*
   * Optional additive pre-activation bias of the research path
   * (RelativeBucketedTimeAndPositionBasedBias, research/modeling/sequential/hstu.py:87-144):
   *   S[i,j] += pos_w[(N-1) + j - i] + ts_w[clamp(floor(log(max(|ts[i+1]-ts[j]|,1)) / bucket_div), 0, num_buckets)]
   * with ts[N] := ts[N-1], shared by all heads, N = max_seq_len.  pos_w == NULL disables it.
   * ts_w / timestamps may be NULL (position-only bias, RelativePositionalBias :66-84).
   */
  const float* pos_w;          /* (2N-1) fp32 */
  const float* ts_w;           /* (num_buckets+1) fp32 or NULL */
  const int64_t* timestamps;   /* (B, ts_row_stride) int64 or NULL */
  int64_t ts_row_stride;
  int32_t num_buckets;         /* 128 in every shipped config */
  float bucket_div;            /* 0.301 */
  /* optional: device pointer to ONE float that replaces `scale` (read by the kernels at launch time, no host
   * sync) -- the reference's `attn_scale` tensor, of which its kernels use element 0 (flash_api.cpp:283,
   * mainloop_fwd_sm80.h:790-793, mainloop_bwd_sm80.h:892-894).  NULL: `scale` is used. */
  const float* attn_scale;
  /* optional: a permutation of 0..batch-1 (int32, device).  Workgroup slot i then takes user user_order[i]: the
   * reference's `sort_by_length` (ops/triton/triton_hstu_attention.py:1968-1973 sorts the lengths in descending order
   * and walks the users in that order; the CUDA kernels' dynamic tile scheduler, flash_common.cpp:496-507, serves the
   * same purpose): with long-tailed length distributions the heavy users start first and the launch does not end on
   * them.  Results never depend on it.  NULL: users in index order. */
  const int32_t* user_order;
} HstuAttnParams;

/*
 * Backward-only additions (hstu::hstu_mha_bwd, flash_api.cpp:323-346: dq/dk/dv
 * are PRE-ALLOCATED by the caller and may be strided views).
 */
typedef struct HstuAttnBwdParams {
  HstuAttnParams fwd;     /* q,k,v,offsets,... as in forward; fwd.out unused */
  const void* dout;       /* (total_rows, H, dv) */
  void* dq; void* dk; void* dv;
  int64_t do_row_stride, do_head_stride;
  int64_t dq_row_stride, dq_head_stride;
  int64_t dk_row_stride, dk_head_stride;
  int64_t dv_row_stride, dv_head_stride;
  void* workspace;        /* hstu_attn_bwd_workspace_bytes() bytes, or NULL if 0 */
  int64_t total_rows;     /* rows of q/k/v (= seq_offsets[B]), needed to size/zero the workspace */
  float* dpos_w;          /* out (2N-1) fp32, required iff fwd.pos_w != NULL */
  float* dts_w;           /* out (num_buckets+1) fp32, required iff fwd.ts_w != NULL */
  int deterministic;      /* ABI v8: != 0 -> every sum in a fixed order (hstu::hstu_mha_bwd's `deterministic`, flash_api.cpp:291;
                           * the CUDA reference serialises its dQ adds with a semaphore, flash_common.cpp:806-858).  One key block
                           * (the folded / 4-wave / one-wave kernels): always the case.  Several key blocks: each block's fp32 dq
                           * partial goes to a slab of its own and the slabs are added in block order (workspace = number of key
                           * blocks x total_rows x heads x dqk x 4 bytes, all of it zeroed and read back: at max_seq_len 8192 that
                           * is dozens of slabs -- hstu_attn_bwd_workspace_bytes reports it, size the batch accordingly).  With the research-path bias
                           * (table gradients are histograms of float atomics) the call is refused: HSTU_EUNSUPPORTED. */
} HstuAttnBwdParams;

/* library identity / errors */
int hstu_abi_version(void);
const char* hstu_last_error(void);

/* HSTU attention forward.  Replaces triton_hstu_attention_fwd
 * (ops/triton/triton_hstu_attention.py:1767-1846), triton_cached_hstu_mha
 * (:2095-2170, when delta_q > 0) and hstu::hstu_mha_fwd (flash_api.cpp:34-110). */
int hstu_attn_fwd(const HstuAttnParams* p, void* stream);

/* HSTU attention backward: dq, dk, dv from dout.  Replaces
 * triton_hstu_attention_bwd (triton_hstu_attention.py:1849-1948) and
 * hstu::hstu_mha_bwd (flash_api.cpp:111-141).  fp32 scratch is needed only when
 * one user's keys do not fit one workgroup. */
size_t hstu_attn_bwd_workspace_bytes(const HstuAttnBwdParams* p);
int hstu_attn_bwd(const HstuAttnBwdParams* p, void* stream);

/* Which kernel instantiation hstu_attn_fwd / hstu_attn_bwd dispatches for these shapes, dtype and mask / bias
 * options (pointers are not dereferenced; only pos_w == NULL or not matters), as the name a profiler shows, e.g.
 * "hstu_attn_bwd_fold_kernel<bf16,128,128>".  Written NUL-terminated into buf (truncated to len); returns HSTU_OK,
 * or the code hstu_attn_* would refuse the call with.  The reference exposes the same information only through its
 * dispatcher branches (ops/hstu_attention.py:87-128, flash_common.cpp:496-507). */
int hstu_attn_fwd_kernel_name(const HstuAttnParams* p, char* buf, size_t len);
int hstu_attn_bwd_kernel_name(const HstuAttnBwdParams* p, char* buf, size_t len);

/* out[0] = 0, out[i+1] = sum(in[0..i]); n elements in, n+1 out; dtype preserved.
 * Replaces hstu::complete_cumsum (ops/cpp/complete_cumsum.cu:7-47) and
 * fbgemm::asynchronous_complete_cumsum (call site modules/stu.py:97). */
int hstu_complete_cumsum(const void* in, void* out, int64_t n, int index_dtype, void* stream);

/*
 * Row copies between jagged 2-D tensors (bit-exact, elem_bytes in {1,2,4,8};
 * rows are `dim` elements, contiguous).  Either side may be dense: pass
 * offsets == NULL and its fixed per-user length in max_len_*.
 *
 * concat: out rows of user b = [right[:n_prefix] ; left ; right[n_prefix:]]
 * split : inverse.  n_prefix = 0 gives plain concat_2D_jagged / split_2D_jagged
 * (ops/jagged_tensors.py:55-144, ops/triton/triton_jagged_tensors.py:31-142);
 * n_prefix > 0 gives hstu_concat_l2_embeddings / hstu_split_l2_embeddings
 * (ops/jagged_tensors.py:147-207).
 */
int hstu_concat_2d_jagged(const void* left, const void* right, void* out,
                          const void* offsets_left, const void* offsets_right,
                          int32_t max_len_left, int32_t max_len_right, int32_t max_seq_len,
                          int32_t batch, int32_t dim, int32_t elem_bytes, int32_t n_prefix,
                          int index_dtype, void* stream);
int hstu_split_2d_jagged(const void* in, void* left, void* right,
                         const void* offsets_left, const void* offsets_right,
                         int32_t max_len_left, int32_t max_len_right, int32_t max_seq_len,
                         int32_t batch, int32_t dim, int32_t elem_bytes, int32_t n_prefix,
                         int index_dtype, void* stream);

/* jagged (total, dim) <-> padded dense (batch, max_len, dim); rows >= max_len are
 * dropped, missing rows are filled with zero bytes.  Replaces
 * fbgemm::jagged_to_padded_dense / fbgemm::dense_to_jagged as used at
 * ops/pytorch/pt_hstu_attention.py:97-125,167-171 and
 * research/modeling/sequential/hstu.py:523,534. */
int hstu_jagged_to_padded_dense(const void* values, void* dense, const void* offsets,
                                int32_t batch, int32_t max_len, int32_t dim, int32_t elem_bytes,
                                int index_dtype, void* stream);
int hstu_dense_to_jagged(const void* dense, void* values, const void* offsets,
                         int32_t batch, int32_t max_len, int32_t dim, int32_t elem_bytes,
                         int index_dtype, void* stream);

/* In-place KV-cache append: for every user b the LAST `tail` rows of its region
 * [offsets[b], offsets[b+1]) of the jagged buffer `values` (rows of dim * elem_bytes bytes) are overwritten with
 * dense[b*tail .. (b+1)*tail).  With it STULayer.cached_forward (modules/stu.py:354-418) writes only the new
 * microbatch's K/V rows into a persistent [cache ; delta] buffer instead of rebuilding that buffer with
 * concat_2D_jagged on every call (_construct_full_kv, modules/stu.py:134-172: O(history) bytes per microbatch). */
int hstu_jagged_write_tail(const void* dense, void* values, const void* offsets, int32_t batch, int32_t tail, int32_t dim,
                           int32_t elem_bytes, int index_dtype, void* stream);

/* 1-D jagged helpers: hstu::expand_1d_jagged_to_dense
 * (ops/cpp/expand_1d_jagged_to_dense.cu:31-103; pads with the user's LAST value,
 * zeros if empty) and hstu::concat_1d_jagged_jagged
 * (ops/cpp/concat_1d_jagged_jagged.cu:33-127).  elem_bytes in {4, 8}. */
int hstu_expand_1d_jagged_to_dense(const void* values, const void* offsets, void* dense,
                                   int32_t batch, int32_t max_len, int32_t elem_bytes,
                                   int index_dtype, void* stream);
int hstu_concat_1d_jagged_jagged(const void* values_left, const void* offsets_left,
                                 const void* values_right, const void* offsets_right,
                                 void* out, int32_t batch, int32_t elem_bytes,
                                 int index_dtype, void* stream);

/*
 * Row-wise layer norm with affine, fp32 math (ops/layer_norm.py:46-76,
 * ops/triton/triton_layer_norm.py:312-470).  fwd writes y and (optionally)
 * mean / rstd (fp32, per row); bwd returns dx and fp32 dweight / dbias.
 */
int hstu_layer_norm_fwd(const void* x, const void* weight, const void* bias, void* y,
                        float* mean, float* rstd, int64_t rows, int32_t dim, float eps,
                        int dtype, void* stream);
/* ABI v9: y = LayerNorm(x) . W^T + bias as ONE kernel -- the UVQK projection of hstu_compute_uqvk
 * (ops/hstu_compute.py:62-89: layer_norm + addmm; Triton: ops/triton/triton_layer_norm.py:77-309 followed by
 * ops/triton/triton_addmm.py:185-340).  x (rows, k) with leading dimension ldx, w_nk the (n, k) K-contiguous weight
 * (= the reference's (k, n) parameter transposed), y (rows, n) with leading dimension ldy; normed (rows, k; the
 * normalised rows, as hstu_layer_norm_fwd would write them), mean and rstd (fp32, per row) are optional outputs.
 * bf16 / fp16, k == 512, n a multiple of 32 up to 4096, 16-byte aligned pointers, leading dimensions multiples of 8:
 * hstu_ln_linear_fwd_supported says whether a shape qualifies (callers otherwise run hstu_layer_norm_fwd + a GEMM). */
int hstu_ln_linear_fwd_supported(int64_t rows, int32_t k, int32_t n, int dtype);
int hstu_ln_linear_fwd(const void* x, int64_t ldx, const void* ln_weight, const void* ln_bias, float eps,
                       const void* w_nk, const void* bias, void* y, int64_t ldy, void* normed, int64_t ldn,
                       float* mean, float* rstd, int64_t rows, int32_t k, int32_t n, int dtype, void* stream);
/* ABI v12: y = x * sigmoid(LayerNorm(x)) -- swish_layer_norm (ops/layer_norm.py:79-112; math ops/pytorch/pt_layer_norm.py:41-62;
 * Triton: ops/triton/triton_layer_norm.py), the gate SwishLayerNorm puts in front of the MLPs of the input preprocessors
 * (modules/preprocessors.py:160,181), the contextual MLPs (modules/contextualize_mlps.py:63,116) and DlrmHSTU
 * (modules/dlrm_hstu.py:144,240).  Same contract as hstu_layer_norm_fwd / _bwd: fp32 math, mean / rstd (fp32, per row) are
 * outputs of the forward (may be NULL) and inputs of the backward, dweight / dbias fp32, partial_ws of
 * hstu_norm_bwd_workspace_bytes(rows, dim) bytes; rows == 0 zeroes dweight / dbias. */
int hstu_swish_layer_norm_fwd(const void* x, const void* weight, const void* bias, void* y, float* mean, float* rstd,
                              int64_t rows, int32_t dim, float eps, int dtype, void* stream);
int hstu_swish_layer_norm_bwd(const void* dy, const void* x, const void* weight, const void* bias, const float* mean,
                              const float* rstd, void* dx, float* dweight, float* dbias, float* partial_ws, int64_t rows,
                              int32_t dim, int dtype, void* stream);
/* ABI v11: y = x . W^T (+ bias) for a contraction length of 512 -- the same kernel without the LayerNorm (rows of x in
 * registers, W streamed through LDS).  Used for d y = d out . W_out^T in the output stage's backward, where k is the
 * embedding dim and the (n, k) K-contiguous operand is the reference's (3 H d, D) `_output_weight` as stored
 * (dx = torch.mm(dz, w.t()): ops/triton/triton_addmm.py:302-315; caller ops/hstu_compute.py:92-136).  Same shape rule
 * and alignment as hstu_ln_linear_fwd; bias may be NULL. */
int hstu_linear_k512_supported(int64_t rows, int32_t k, int32_t n, int dtype);
int hstu_linear_k512(const void* x, int64_t ldx, const void* w_nk, const void* bias, void* y, int64_t ldy,
                     int64_t rows, int32_t k, int32_t n, int dtype, void* stream);
/* ABI v13: d = a . b + c, row-major, fp32 accumulation, C and D DIFFERENT buffers -- the output stage out = x + y . W_o
 * (`torch.addmm(x, y, output_weight)`, ops/hstu_compute.py:92-136; the reference's own kernel for it: ops/triton/triton_addmm.py:185-340)
 * as ONE hipBLASLt launch.  Through torch.addmm the same product is a copy of x into the result followed by an in-place GEMM with
 * beta = 1 (40 us per layer at 204,800 x 512).  A plain library GEMM: hipBLASLt is looked up at run time (dlopen), nothing links
 * against it; _supported() says whether it was found (else the caller keeps torch.addmm -- the same library behind it).
 * a (m, k) lda, b (k, n) ldb, c (m, n) ldc, d (m, n) ldd: leading dimensions in elements, operands 16-byte aligned, bf16 / fp16;
 * workspace: any size (0 / NULL allowed), handed to hipBLASLt's heuristic as the upper bound. */
int hstu_addmm_residual_supported(void);
int hstu_addmm_residual(const void* c, int64_t ldc, const void* a, int64_t lda, const void* b, int64_t ldb, void* d, int64_t ldd,
                        int64_t m, int32_t n, int32_t k, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int hstu_layer_norm_bwd(const void* dy, const void* x, const void* weight,
                        const float* mean, const float* rstd, void* dx,
                        float* dweight, float* dbias, float* partial_ws,
                        int64_t rows, int32_t dim, int dtype, void* stream);
/* ABI v7: dx = LayerNorm'(dy) + dresidual -- the gradient that reaches x around the norm (the STU layer's residual,
 * stu.py:340-351) added inside the kernel instead of by a separate pass; dresidual (rows, dim) in `dtype`, may be NULL. */
int hstu_layer_norm_bwd_residual(const void* dy, const void* x, const void* weight,
                                 const float* mean, const float* rstd, const void* dresidual,
                                 void* dx, float* dweight, float* dbias, float* partial_ws,
                                 int64_t rows, int32_t dim, int dtype, void* stream);
size_t hstu_norm_bwd_workspace_bytes(int64_t rows, int32_t dim);

/*
 * y = u * Norm(attn) with Norm = LayerNorm over the row or per-head GroupNorm,
 * optionally written as the concatenation [u, attn, y]  (rows, 3*dim).
 * Replaces _ln_mul_dropout_fwd/_bwd and _group_norm_mul_dropout_fwd/_bwd
 * (ops/triton/triton_hstu_linear.py:48-337,570-1036).
 *
 * The _dropout_ entry points (ABI v6) fuse the reference's in-kernel dropout
 * (triton_hstu_linear.py:101-120, 196-215; pt_hstu_linear.py:60-64 drops the concatenated
 * tensor): every element of the (rows, dim | 3 dim) output is zeroed with probability
 * p = round(dropout_ratio * 65536) / 65536 and the survivors are scaled by 1 / (1 - p).
 * The mask is a pure function of (seed, element index): bwd -- and a forward recompute of y --
 * regenerate it from the same seed, nothing is stored.  dropout_ratio == 0 is the plain op.
 * bwd takes dy with respect to the DROPPED output.
 */
int hstu_norm_mul_fwd(const void* attn, const void* u, const void* weight, const void* bias,
                      void* y, float* mean, float* rstd, int64_t rows, int32_t heads,
                      int32_t head_dim, float eps, int group_norm, int concat_ux,
                      int dtype, void* stream);
int hstu_norm_mul_bwd(const void* dy, const void* attn, const void* u, const void* weight,
                      const void* bias, const float* mean, const float* rstd,
                      void* dattn, void* du, float* dweight, float* dbias, float* partial_ws,
                      int64_t rows, int32_t heads, int32_t head_dim, int group_norm,
                      int concat_ux, int dtype, void* stream);
int hstu_norm_mul_dropout_fwd(const void* attn, const void* u, const void* weight,
                              const void* bias, void* y, float* mean, float* rstd, int64_t rows,
                              int32_t heads, int32_t head_dim, float eps, int group_norm,
                              int concat_ux, float dropout_ratio, uint64_t seed, int dtype,
                              void* stream);
int hstu_norm_mul_dropout_bwd(const void* dy, const void* attn, const void* u, const void* weight,
                              const void* bias, const float* mean, const float* rstd,
                              void* dattn, void* du, float* dweight, float* dbias,
                              float* partial_ws, int64_t rows, int32_t heads, int32_t head_dim,
                              int group_norm, int concat_ux, float dropout_ratio, uint64_t seed,
                              int dtype, void* stream);

/* ABI v7: the same with u given by rows `u_row_stride` elements apart and, when u_is_preactivation != 0, as the
 * PRE-activation of SiLU (the u slice of the uvqk buffer): the kernels apply SiLU on the fly, rounded to `dtype` where
 * hstu_silu_fwd would have stored it, and the backward multiplies d u by SiLU' and writes rows `du_row_stride` apart
 * (the u slice of the d uvqk buffer) -- hstu_silu_fwd / hstu_silu_bwd fused away, results bit-identical to the
 * separate calls.  Replaces the F.silu(u) of hstu_compute_uqvk (ops/hstu_compute.py:85) next to
 * _group_norm_mul_dropout_fwd/_bwd inside one STU layer. */
int hstu_norm_mul_silu_fwd(const void* attn, const void* u, int64_t u_row_stride, int u_is_preactivation,
                           const void* weight, const void* bias, void* y, float* mean, float* rstd,
                           int64_t rows, int32_t heads, int32_t head_dim, float eps, int group_norm,
                           int concat_ux, float dropout_ratio, uint64_t seed, int dtype, void* stream);
int hstu_norm_mul_silu_bwd(const void* dy, const void* attn, const void* u, int64_t u_row_stride,
                           int u_is_preactivation, const void* weight, const void* bias,
                           const float* mean, const float* rstd, void* dattn, void* du,
                           int64_t du_row_stride, float* dweight, float* dbias, float* partial_ws,
                           int64_t rows, int32_t heads, int32_t head_dim, int group_norm, int concat_ux,
                           float dropout_ratio, uint64_t seed, int dtype, void* stream);

/* u = silu(u) in place on the leading `u_cols` columns of each row of a
 * (rows, row_stride) matrix, and its backward (du *= silu'(u_pre)); the SiLU-on-u
 * epilogue of hstu_compute_uqvk (ops/hstu_compute.py:85). */
int hstu_silu_fwd(const void* in, void* out, int64_t rows, int32_t cols,
                  int64_t in_row_stride, int64_t out_row_stride, int dtype, void* stream);
int hstu_silu_bwd(const void* dout, const void* in, void* din, int64_t rows, int32_t cols,
                  int64_t dout_row_stride, int64_t in_row_stride, int64_t din_row_stride,
                  int dtype, void* stream);

/* ---- timestamp / position additive encoder (the step before the STU stack) ------------------------------
 * out[row] = alpha * x[row] + pos_w[pos_idx(row)] + ts_w[ts_idx(row)], x / out (sum L, dim) in `dtype`, tables fp32
 * (max_pos_ind, dim) and (>= max_time_bucket + 1, dim); also writes the two int32 table indices of every row (the
 * backward needs them).  Index arithmetic of ops/pytorch/pt_position.py:40-122 (position: bit-exact integers;
 * time: fp32 bucket of query_time - timestamp, time_bucket_fn 0 = sqrt, 1 = log; NB the reference clamps the
 * bucket to ts_embeddings.size(1) - 1, pass that as max_time_bucket).  timestamps (sum L) int64; num_targets may be
 * NULL; seq_offsets / num_targets share `index_dtype`.  Replaces triton_add_timestamp_positional_embeddings
 * (ops/triton/triton_position.py:62-158, 241-337), selected by ops/position.py:38-96. */
int hstu_add_ts_pos_emb_fwd(const void* x, void* out, const void* seq_offsets, const int64_t* timestamps,
                            const void* num_targets, const float* pos_w, const float* ts_w, int32_t* pos_idx,
                            int32_t* ts_idx, int32_t batch, int32_t dim, int32_t max_contextual_seq_len,
                            int32_t max_pos_ind, int32_t max_time_bucket, int32_t interleave_targets,
                            int32_t time_bucket_fn, float alpha, int dtype, int index_dtype, void* stream);
/* table_grad[i, :] = sum over the rows e with idx[e] == i of dout[e, :]  (fp32, (table_rows, dim), zeroed here): the
 * index_select backward of one embedding table.  Rows are grouped by table index inside (a radix sort over the
 * ceil(log2(table_rows)) significant bits, stable: each table row's sum runs in row order), then summed by
 * segment.  `workspace`: device memory, 256-byte aligned, at least hstu_embedding_grad_workspace_bytes(n, table_rows)
 * bytes.  Rows of dout: 16-byte multiples, <= 4 KiB.  0 <= idx[e] < table_rows (not checked).  Replaces
 * _add_embeddings_bwd_kernel and the sort on its host side (ops/triton/triton_position.py:188-238, :339-407). */
int hstu_embedding_grad_workspace_bytes(int64_t n, int32_t table_rows, int64_t* bytes);
int hstu_embedding_grad(const void* dout, const int32_t* idx, int64_t n, int32_t dim, int32_t table_rows, float* table_grad,
                        void* workspace, int64_t workspace_bytes, int dtype, void* stream);
/* The second half alone, for callers that hold the rows sorted already: table_grad[i, :] = sum over e with
 * sorted_idx[e] == i of dout[sorted_rows[e], :]. */
int hstu_embedding_grad_segment_sum(const void* dout, const int64_t* sorted_rows, const int32_t* sorted_idx, int64_t n,
                                    int32_t dim, int32_t table_rows, float* table_grad, int dtype, void* stream);

/* ---- output postprocessor: row L2 normalisation ---------------------------------------------------------
 * y = x / max(||x||_2, eps) per row of (rows, dim), and its backward.  Replaces L2NormPostprocessor.forward
 * (modules/postprocessors.py:55-69: seq / linalg.norm(seq, dim=-1).clamp(min=1e-6)) and its autograd. */
int hstu_l2_norm_fwd(const void* x, void* y, int64_t rows, int32_t dim, float eps, int dtype, void* stream);
int hstu_l2_norm_bwd(const void* dy, const void* x, void* dx, int64_t rows, int32_t dim, float eps, int dtype,
                     void* stream);

/* ---- sampled-softmax loss, dot-product similarity --------------------------------------
 * Per supervision row i (n_rows of them):
 *   l_i0 = <q_i, P(pos_emb_i)> / T,  l_ik = <q_i, N(table[neg_rows[i,k]])> / T  for k < num_negatives,
 *   l_ik = -5e4 where neg_ids[i,k] == pos_ids[i];  row_loss_i = logsumexp_k(l_i0, l_i1 ..) - l_i0,  lse_i = that logsumexp;
 *   P / N = x / max(||x||_2, eps) when pos_l2_norm / table_l2_norm, identity otherwise.
 * Replaces SampledSoftmaxLoss.jagged_forward (research/modeling/sequential/losses/sampled_softmax.py:44-95) with the
 * DotProductSimilarity (research/rails/similarities/dot_product_similarity_fn.py:38-62): the caller reduces
 * sum_i w_i row_loss_i / sum_i w_i.  neg_rows index `table`; neg_ids are the item ids of those rows: the same matrix
 * for LocalNegativesSampler (autoregressive_losses.py:112-131: table = the item embedding weight, pos/table l2 norm =
 * the sampler's l2_norm), `cached_ids[offsets]` for InBatchNegativesSampler (:186-204: table = its cached, already
 * normalised embeddings, table_l2_norm = 0).  q, pos_emb, table: (., dim) in `dtype`, rows 16-byte aligned
 * (dim a multiple of 4 for fp32 / 8 for 16-bit, dim * elt <= 1024 bytes); ids int64; row_loss, lse fp32 (n_rows).
 * Indices outside [0, table_rows) are clamped (memory safety only). */
int hstu_sampled_softmax_fwd(const void* q, int64_t q_row_stride, const void* pos_emb, int64_t pos_row_stride,
                             const int64_t* pos_ids, const int64_t* neg_rows, const int64_t* neg_ids, const void* table,
                             int64_t table_row_stride, int64_t table_rows, int64_t n_rows, int32_t num_negatives,
                             int32_t dim, float temperature, int32_t pos_l2_norm, int32_t table_l2_norm, float eps,
                             float* row_loss, float* lse, int dtype, void* stream);
/* Gradients of sum_i grad_row_loss[i] * row_loss_i: dq, dpos_emb (n_rows, dim) in `dtype` (every row written; rows
 * with grad_row_loss == 0 get zeros and gather nothing), dtable (table_rows, dim) fp32 contiguous, ACCUMULATED into
 * (zero it first) with atomics -- the one sum of this op whose order is not fixed.  A masked negative passes no
 * gradient (torch.where picks the constant, sampled_softmax.py:78-82). */
int hstu_sampled_softmax_bwd(const void* q, int64_t q_row_stride, const void* pos_emb, int64_t pos_row_stride,
                             const int64_t* pos_ids, const int64_t* neg_rows, const int64_t* neg_ids, const void* table,
                             int64_t table_row_stride, int64_t table_rows, int64_t n_rows, int32_t num_negatives,
                             int32_t dim, float temperature, int32_t pos_l2_norm, int32_t table_l2_norm, float eps,
                             const float* lse, const float* grad_row_loss, void* dq, int64_t dq_row_stride, void* dpos_emb,
                             int64_t dpos_row_stride, float* dtable, int dtype, void* stream);

/* ---- ABI v10: the glue around the projections of an STU layer ---------------------------------------------
 * hstu_cast_params: up to HSTU_CAST_MAX_ITEMS fp32 tensors -> bf16 / fp16 in ONE launch; an item with transpose != 0 is a
 * (rows, cols) row-major matrix written as its (cols, rows) transpose (the UVQK weight's K-contiguous copy).  Replaces the
 * per-parameter ``.to(x.dtype)`` casts the reference issues in front of every projection (ops/hstu_compute.py:62-72,
 * ops/triton/triton_hstu_linear.py:1160-1170; ops/triton/triton_hstu_preprocess_and_attention.py:60-75). */
#define HSTU_CAST_MAX_ITEMS 8
typedef struct HstuCastItem {
  const float* src;
  void* dst;
  int64_t numel;
  int32_t rows, cols;     /* only read when transpose != 0: rows * cols == numel */
  int32_t transpose;
} HstuCastItem;
int hstu_cast_params(const HstuCastItem* items, int32_t n_items, int dst_dtype, void* stream);
/* out[c] = sum over r < rows of x[r * ldx + c] (fp32, fixed summation order: bit-identical run to run) for a bf16 / fp16
 * matrix -- the bias gradient of a projection (ops/triton/triton_addmm.py:309 ``torch.sum(dz, dim=0)``; d uvqk_beta in
 * ops/triton/triton_hstu_preprocess_and_attention.py:122-293).  cols and ldx multiples of 8, x 16-byte aligned;
 * `workspace`: hstu_column_sum_workspace_bytes(rows, cols) bytes of device memory, 16-byte aligned. */
size_t hstu_column_sum_workspace_bytes(int64_t rows, int32_t cols);
int hstu_column_sum(const void* x, int64_t ldx, int64_t rows, int32_t cols, float* out, void* workspace, int dtype,
                    void* stream);
/* Calibration streams for bench.py (no reference counterpart; measurement only): `iters` x 16 independent 32x32x16 bf16
 * MFMAs per wave, two waves per SIMD on every CU, no memory traffic (*flops = the FLOP the launch performs), and a
 * non-temporal 16-byte-per-lane read of `bytes` bytes without arithmetic.  What this box sustains for the two resources the
 * product kernels are priced against, measured in the same process as they are. */
int hstu_calib_mfma_stream(int32_t iters, float* sink, double* flops, void* stream);
int hstu_calib_read_stream(const void* src, size_t bytes, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HSTU_HIP_H_ */
