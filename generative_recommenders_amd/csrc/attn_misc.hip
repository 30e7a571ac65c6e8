// dtype-independent helpers of the attention ABI.
#include <stdio.h>
#include <stdlib.h>

#include "capi_internal.h"
namespace hstu {
int attn_bwd_tiles_bf16(int, int, int, int);
int attn_bwd_tiles_f16(int, int, int, int);
int attn_bwd_tiles_f32(int, int, int, int);

int attn_bwd_tiles_per_block(int dtype, int dqk, int dv, int max_seq_len, int extra_lds) {
  switch (dtype) {
    case HSTU_DTYPE_BF16: return attn_bwd_tiles_bf16(dqk, dv, max_seq_len, extra_lds);
    case HSTU_DTYPE_F16: return attn_bwd_tiles_f16(dqk, dv, max_seq_len, extra_lds);
    default: return attn_bwd_tiles_f32(dqk, dv, max_seq_len, extra_lds);
  }
}

// The folded schedule needs: 16-bit I/O, no bias, no contextual rows, one key block of <= 7 tiles, and head
// dims equal to an instantiated size (its LDS-DMA staging cannot zero-pad).  HSTU_BWD_FOLD=0 forces the
// general kernel (A/B measurements).
bool attn_bwd_fold_applicable(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  static const bool enabled = [] { const char* e = getenv("HSTU_BWD_FOLD"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (p.dtype == HSTU_DTYPE_F32 || p.pos_w || p.contextual_seq_len > 0) return false;
  if (p.dqk != p.dv || (p.dqk != 128 && p.dqk != 64)) return false;
  // masked elements start their S accumulator at -1e30 (hstu_attn_bwd_fold.cuh, fold_pair): alpha * 1e30 must stay
  // finite and alpha * 1.44e30 must overflow exp2; alpha == 0 is fine (every product with a masked element is then
  // multiplied by alpha = 0 or by silu(0) = 0 anyway)
  const float aa = p.alpha < 0.f ? -p.alpha : p.alpha;
  if (!(aa == 0.f || (aa > 1e-20f && aa < 1e6f))) return false;
  return (p.max_seq_len + 31) / 32 <= 7;
}

// The two-kernel backward for long sequences (hstu_attn_bwd_long.cuh): what the folded schedule does not take for length only.
bool attn_bwd_long_applicable(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  static const bool enabled = [] { const char* e = getenv("HSTU_BWD_LONG"); return !(e && e[0] == '0'); }();
  if (!enabled) return false;
  if (p.dtype == HSTU_DTYPE_F32 || p.pos_w || p.delta_q != 0) return false;
  if (p.dqk != p.dv || (p.dqk != 128 && p.dqk != 64)) return false;
  const float aa = p.alpha < 0.f ? -p.alpha : p.alpha;          // masks ride on the S accumulator's start value (-1e30)
  if (!(aa == 0.f || (aa > 1e-20f && aa < 1e6f))) return false;
  // longer than the folded schedule takes -- or contextual rows, which it does not take at any length
  return (p.max_seq_len + 31) / 32 > 7 || (p.contextual_seq_len > 0 && p.max_seq_len > 64);
}

bool attn_solo_applicable(const HstuAttnParams& p, bool backward) {
  static const bool enabled = [] { const char* e = getenv("HSTU_SOLO"); return !(e && e[0] == '0'); }();
  if (!enabled || p.dtype == HSTU_DTYPE_F32 || p.pos_w || p.delta_q != 0) return false;
  if (p.max_seq_len > 64 || p.dqk > 32 || p.dv > 32) return false;
  if (backward && p.contextual_seq_len > 0) return false;
  const float aa = p.alpha < 0.f ? -p.alpha : p.alpha;          // masks ride on the S accumulator's start value (-1e30)
  return aa == 0.f || (aa > 1e-20f && aa < 1e6f);
}

// LDS of the short-sequence research backward behind its four slices (`base`): histograms with as many time-bucket copies
// as fit (32, 8 or 1), tables, the waves' private bucket bytes (`cache`)
bool attn_solo_bias_lds(const HstuAttnParams& p, int base, int cache, int* ts_copies, int* hist_bytes, int* smem) {
  const int tables = 2 * bias_table_bytes(p.max_seq_len, p.num_buckets);      // (double-buffered: hstu_attn_solo.cuh)
  for (int c : {32, 8, 1}) {
    const int hist = ((2 * p.max_seq_len + (p.num_buckets + 1) * c) * 4 + 15) / 16 * 16;
    if (base + hist + tables + cache <= kLdsBudget) {
      if (ts_copies) *ts_copies = c;
      if (hist_bytes) *hist_bytes = hist;
      if (smem) *smem = base + hist + tables + cache;
      return true;
    }
  }
  return false;
}

// research path at the short-sequence shapes (Amazon-Books: N = 61, 4 heads of 16): relative bias inside the one-wave-per-
// (user, head) kernels.  16-bit I/O, max_seq_len <= 64, head dims <= 32, <= 255 time buckets (bucket bytes), no contextual
// rows, no delta; HSTU_SOLO_BIAS=0: the general bias kernels (A/B measurements)
bool attn_solo_bias_applicable(const HstuAttnParams& p, bool backward) {
  static const bool enabled = [] { const char* e = getenv("HSTU_SOLO_BIAS"); return !(e && e[0] == '0'); }();
  if (!enabled || !p.pos_w || p.dtype == HSTU_DTYPE_F32 || p.delta_q != 0 || p.contextual_seq_len > 0) return false;
  if (p.max_seq_len > 64 || p.dqk > 32 || p.dv > 32 || p.num_buckets > 255) return false;
  const float aa = p.alpha < 0.f ? -p.alpha : p.alpha;          // masks ride on the S accumulator's start value (-1e30)
  if (!(aa == 0.f || (aa > 1e-20f && aa < 1e6f))) return false;
  if (p.max_seq_len + 32 + 3 > 256) return false;      // (a thread per timestamp entry)
  if (!backward) return 4 * (6 * 2048 + bias_table_bytes(p.max_seq_len, p.num_buckets) + 3 * 1024) <= kLdsBudget;
  return attn_solo_bias_lds(p, 4 * (8 * 2048 + 2 * 32 * 64), 2 * 3 * 1024, nullptr, nullptr, nullptr);
}

bool attn_bwd_quad_applicable(const HstuAttnBwdParams& bp) {
  static const bool enabled = [] { const char* e = getenv("HSTU_BWD_QUAD"); return !(e && e[0] == '0'); }();
  return enabled && attn_bwd_fold_applicable(bp) && bp.fwd.dqk == 64;
}

// K/V ring of the forward kernel (FwdCfg in hstu_attn_fwd.cuh): three stages when every wave issues whole DMA instructions
// per tile and a stage is at most 16 KiB, else two
int attn_fwd_ring_bytes(int eb, int a, int v) {
  const int stage = 32 * (a + v) * eb;
  const int nch_k = 32 * (a * eb / 16) / 64, nch_v = 32 * (v * eb / 16) / 64;
  const bool counted = nch_k % 4 == 0 && nch_v % 4 == 0;
  return ((counted && stage <= 16384) ? 3 : 2) * stage;
}

bool attn_fwd_head_loop_applicable(const HstuAttnParams& p, int ring_bytes, int* tables_out, int* cache_out) {
  const int q_rows = p.delta_q > 0 ? p.delta_q : p.max_seq_len;
  const int nqb = (q_rows + 127) / 128, tmax = (p.max_seq_len + 31) / 32;
  const int tables = p.pos_w ? bias_table_bytes(p.max_seq_len, p.num_buckets) : 0;
  // 1 KiB per key tile and wave: wave w of query block qb keeps 4 qb + w + 1 tiles.  The LARGEST block counts -- not
  // always the last one: at 129..160 rows the last block has one wave with rows (5 tiles) and the first four (1 + 2 + 3 + 4)
  int cache = 0;
  for (int qb = 0; qb < nqb; ++qb) {
    int c = 0;
    for (int w = 0; w < 4; ++w)
      if (4 * qb + w < tmax) c += (4 * qb + w + 1) * 1024;
    if (c > cache) cache = c;
  }
  if (tables_out) *tables_out = tables;
  if (cache_out) *cache_out = cache;
  return p.pos_w && p.dtype != HSTU_DTYPE_F32 && p.heads > 1 && p.delta_q == 0 && tmax <= 7 && p.ts_w && p.timestamps &&
         p.num_buckets <= 255 && p.contextual_seq_len == 0 && ring_bytes + tables + cache <= 52 * 1024 && attn_bias_head_loop_enabled();
}

// Name of the instantiation attn_launch.cuh / attn_fold.cuh dispatch (the same decisions, restated once here; the
// launch tests compare it with the kernel names rocprofv3 reports).
int attn_kernel_name(const HstuAttnParams& p, const HstuAttnBwdParams* bwd, char* buf, size_t len) {
  if (!buf || len == 0) return HSTU_EINVAL;
  const char* dt = p.dtype == HSTU_DTYPE_BF16 ? "bf16" : (p.dtype == HSTU_DTYPE_F16 ? "f16" : "f32");
  const int a = pad_head_dim(p.dqk), v = pad_head_dim(p.dv);
  if (a == 0 || v == 0) { buf[0] = 0; return set_error(HSTU_EUNSUPPORTED, "head dims (%d, %d) not instantiated", p.dqk, p.dv); }
  if (p.pos_w && a != v) { buf[0] = 0; return set_error(HSTU_EUNSUPPORTED, "relative-bias attention is instantiated for dqk == dv"); }
  if (attn_solo_applicable(p, bwd != nullptr)) snprintf(buf, len, "hstu_attn_%s_solo_kernel<%s>", bwd ? "bwd" : "fwd", dt);
  else if (attn_solo_bias_applicable(p, bwd != nullptr)) snprintf(buf, len, "hstu_attn_%s_solo_bias_kernel<%s>", bwd ? "bwd" : "fwd", dt);
  else if (bwd && attn_bwd_quad_applicable(*bwd)) snprintf(buf, len, "hstu_attn_bwd_quad_kernel<%s,%d>", dt, a);
  else if (bwd && attn_bwd_fold_applicable(*bwd)) snprintf(buf, len, "hstu_attn_bwd_fold_kernel<%s,%d,%d>", dt, a, v);
  else if (bwd && attn_bwd_long_applicable(*bwd)) snprintf(buf, len, "hstu_attn_bwd_dkv_kernel<%s,%d>+hstu_attn_bwd_dq_kernel<%s,%d>", dt, a, dt, a);
  else if (bwd && attn_bwd_fold_bias_applicable(*bwd)) snprintf(buf, len, "hstu_attn_bwd_fold_bias_kernel<%s,64>", dt);
  else if (!bwd && attn_fwd_head_loop_applicable(p, attn_fwd_ring_bytes(p.dtype == HSTU_DTYPE_F32 ? 4 : 2, a, v), nullptr, nullptr))
    snprintf(buf, len, "hstu_attn_fwd_kernel<%s,%d,%d,bias,heads>", dt, a, v);
  else if (!bwd && !p.pos_w && p.dtype != HSTU_DTYPE_F32 && a == v && (a == 64 || a == 128) && attn_fwd_precise_enabled())
    snprintf(buf, len, "hstu_attn_fwd_kernel<%s,%d,%d,precise>", dt, a, v);
  else snprintf(buf, len, "hstu_attn_%s_kernel<%s,%d,%d%s>", bwd ? "bwd" : "fwd", dt, a, v, p.pos_w ? ",bias" : "");
  return HSTU_OK;
}

// Time buckets are few and lopsided (most pairs of a user fall in the top three or four): the lanes of a wave mostly
// add to the SAME histogram entry and the LDS atomics serialise.  Every lane group therefore gets its own copy of
// the time-bucket histogram (entry b of copy c at b * copies + c: the lanes of a wave that share a bucket hit
// consecutive words), as many copies as fit without costing a key tile; the copies are summed at the flush.
// LDS of the folded research-path backward behind the folded kernel's own `base` bytes: histograms (as many time-bucket
// copies as fit: 32, 8 or 1), tables, one bucket byte per element of the causal triangle of 7 tiles
bool attn_bwd_fold_bias_lds(const HstuAttnParams& p, int base, int* ts_copies, int* hist_bytes, int* smem) {
  const int tables = bias_table_bytes(p.max_seq_len, p.num_buckets), cache = 28 * 1024;
  for (int c : {32, 8, 1}) {
    const int hist = ((2 * p.max_seq_len + (p.num_buckets + 1) * c) * 4 + 15) / 16 * 16;
    if (base + hist + tables + cache <= kLdsBudget) {
      if (ts_copies) *ts_copies = c;
      if (hist_bytes) *hist_bytes = hist;
      if (smem) *smem = base + hist + tables + cache;
      return true;
    }
  }
  return false;
}

// research-path backward on the folded schedule: head dim 64, 16-bit I/O, the whole sequence in 7 tiles, position AND
// time tables (HSTU_BIAS_FOLD=0: the general kernel, A/B measurements)
bool attn_bwd_fold_bias_applicable(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  static const bool enabled = [] { const char* e = getenv("HSTU_BIAS_FOLD"); return !(e && e[0] == '0'); }();
  if (!enabled || !p.pos_w || !p.ts_w || !p.timestamps) return false;
  if (p.dtype == HSTU_DTYPE_F32 || p.contextual_seq_len > 0 || p.dqk != 64 || p.dv != 64) return false;
  if (p.num_buckets > 255 || (p.max_seq_len + 31) / 32 > 7) return false;
  const float aa = p.alpha < 0.f ? -p.alpha : p.alpha;
  if (!(aa == 0.f || (aa > 1e-20f && aa < 1e6f))) return false;
  return attn_bwd_fold_bias_lds(p, (7 + 2) * 2 * 32 * 64 * 2 + 8 * 32 * 64, nullptr, nullptr, nullptr);
}

bool attn_fwd_precise_enabled() {
  static const bool on = [] { const char* e = getenv("HSTU_ATTN_PRECISE"); return e && e[0] == '1'; }();
  return on;
}

// HSTU_BIAS_HEAD_LOOP=0: one workgroup per (user, head) for the research-path backward, as before (A/B measurements)
bool attn_bias_head_loop_enabled() {
  static const bool on = [] {
    const char* e = getenv("HSTU_BIAS_HEAD_LOOP");
    return !(e && e[0] == '0');
  }();
  return on;
}

int attn_bwd_bias_lds(const HstuAttnParams& p, int* ts_copies) {
  if (ts_copies) *ts_copies = 1;
  if (!p.pos_w) return 0;
  auto bytes = [&](int nc) {
    return ((2 * p.max_seq_len + (p.num_buckets + 1) * nc) * 4 + 15) / 16 * 16 + bias_table_bytes(p.max_seq_len, p.num_buckets);
  };
  const int base = attn_bwd_tiles_per_block(p.dtype, p.dqk, p.dv, p.max_seq_len, bytes(1));
  int nc = 1;
  for (int c : {32, 8})
    if (attn_bwd_tiles_per_block(p.dtype, p.dqk, p.dv, p.max_seq_len, bytes(c)) == base) { nc = c; break; }
  if (ts_copies) *ts_copies = nc;
  return bytes(nc);
}

size_t attn_bwd_workspace_bytes(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  // the short-sequence schedules and the two-kernel long backward add nothing in memory (the research-path kernels keep their
  // histogram rows: p.pos_w below)
  if (!p.pos_w && (attn_solo_applicable(p, true) || attn_bwd_fold_applicable(bp) || attn_bwd_long_applicable(bp))) return 0;
  const int hist = attn_bwd_bias_lds(p, nullptr);
  const int nw = attn_bwd_tiles_per_block(p.dtype, p.dqk, p.dv, p.max_seq_len, hist);
  if (nw <= 0) return 0;
  const int nkb = (p.max_seq_len + 32 * nw - 1) / (32 * nw);
  size_t bytes = 0;
  // fp32 dq accumulator: one for the atomic adds of all key blocks, or (deterministic) one slab per key block
  if (nkb > 1) bytes += (((size_t)bp.total_rows * p.heads * p.dqk * sizeof(float) + 255) / 256 * 256) * ((bp.deterministic && !p.pos_w) ? nkb : 1);   // (deterministic with the bias is refused: hstu_attn_bwd)
  if (p.pos_w) {
    const size_t nblocks = (size_t)((p.batch * p.heads + 7) / 8) * 8 * nkb;
    bytes += (nblocks + 128) * (2 * p.max_seq_len + p.num_buckets) * sizeof(float);   // + the reduce's 128 chunk-sum rows
  }
  return bytes;
}

// column sums of the (rows, width) partial matrix in two fixed-order stages (deterministic across launches):
// stage 1, grid (column blocks, kRedChunks): every workgroup sums its contiguous chunk of rows for 64 columns into
// row `chunk` of the chunk-sum rows that follow the partial rows in the workspace; stage 2: one thread per column
// sums the kRedChunks chunk rows.  (One thread per column over ALL rows -- the first version -- took 3.6 ms for the
// 32768 partial rows of an 8192-user batch: 9 waves on the whole chip.)
constexpr int kRedChunks = 128;

__global__ void bias_grad_reduce_stage1(const float* partial, int rows, int width, int chunks, float* chunk_sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  const int per = (rows + chunks - 1) / chunks;
  const int r0 = blockIdx.y * per, r1 = min(r0 + per, rows);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += partial[(int64_t)r * width + c];
    s1 += partial[(int64_t)(r + 1) * width + c];
    s2 += partial[(int64_t)(r + 2) * width + c];
    s3 += partial[(int64_t)(r + 3) * width + c];
  }
  for (; r < r1; ++r) s0 += partial[(int64_t)r * width + c];
  chunk_sums[(int64_t)blockIdx.y * width + c] = (s0 + s1) + (s2 + s3);
}

__global__ void bias_grad_reduce_stage2(const float* chunk_sums, int chunks, int width, int npos, float* dpos_w, float* dts_w) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  // (four independent partial sums: the loads of a column are 4 deep in flight instead of one after the other)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int r = 0;
  for (; r + 3 < chunks; r += 4) {
    s0 += chunk_sums[(int64_t)r * width + c];
    s1 += chunk_sums[(int64_t)(r + 1) * width + c];
    s2 += chunk_sums[(int64_t)(r + 2) * width + c];
    s3 += chunk_sums[(int64_t)(r + 3) * width + c];
  }
  for (; r < chunks; ++r) s0 += chunk_sums[(int64_t)r * width + c];
  const float s = (s0 + s1) + (s2 + s3);
  if (c < npos) dpos_w[c] = s;
  else if (dts_w) dts_w[c - npos] = s;
}

int launch_bias_grad_reduce(const float* partial, int rows, int width, int npos, float* dpos_w, float* dts_w,
                            hipStream_t st) {
  // the chunk sums live right behind the partial rows (attn_bwd_workspace_bytes reserves kRedChunks extra rows).  Chunks ~ sqrt(rows):
  // both stages are one thread per column walking its rows one after the other, and with a fixed 128 chunks the second stage was 128
  // dependent loads for the 256 .. 768 rows of the persistent kernels: 31 us per backward call (12 % of the Amazon-Books research-path
  // backward: profiles/r06_c3bias_rocprofv3_pmc.md)
  int chunks = 1;
  while (chunks * chunks < rows && chunks < kRedChunks) ++chunks;
  float* chunk_sums = const_cast<float*>(partial) + (size_t)rows * width;
  hipLaunchKernelGGL(bias_grad_reduce_stage1, dim3((width + 63) / 64, chunks), dim3(64), 0, st, partial, rows, width, chunks, chunk_sums);
  if (int e = check_launch("hstu_attn_bwd(bias gradient reduce 1)")) return e;
  hipLaunchKernelGGL(bias_grad_reduce_stage2, dim3((width + 63) / 64), dim3(64), 0, st, chunk_sums, chunks, width, npos, dpos_w, dts_w);
  return check_launch("hstu_attn_bwd(bias gradient reduce 2)");
}

}  // namespace hstu
