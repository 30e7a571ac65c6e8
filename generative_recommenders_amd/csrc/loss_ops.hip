// Sampled-softmax loss with a dot-product similarity (HBM-bound random gather).
//   l_i0 = <q_i, norm(pos_i)> / T,   l_ik = <q_i, norm(table[rows_ik])> / T   (= -5e4 where ids_ik == pos_id_i)
//   loss_i = logsumexp(l_i0 .. l_iR) - l_i0
// Reference semantics: research/modeling/sequential/losses/sampled_softmax.py:44-95 (jagged_forward),
// autoregressive_losses.py:37-45 (l2 norm with clamp), :112-131 / :186-204 (what the samplers hand over),
// rails/similarities/dot_product_similarity_fn.py:38-62.  The reference materialises the (N', R, D) tensor of
// negative embeddings (R = 512 on Amazon-Books: 128 KB per row in fp32), normalises it, runs a bmm, builds the
// (N', R+1) logits and a log_softmax: five passes over the gathered rows forward, more backward.  Here the rows are
// gathered ONCE per pass straight from the table, never written anywhere.
//
// Mapping: one workgroup of 4 waves per supervision row.  An embedding occupies LPE = dim * elt / 16 lanes (a power of
// two, 16 bytes per lane), so a wave handles 64 / LPE negatives per step; q_i's slice lives in registers.  Dot product
// and squared norm are reduced over the LPE lanes with xor-shuffles; the softmax is online (running max / sum per lane
// group), merged across groups by shuffles and across waves through LDS.  Algorithmic bytes per row: R * dim * elt
// (the gather) + 2 * dim * elt + 16 * R (two int64 index matrices).
// Backward: the same gather again; p_ik = exp(l_ik - lse_i) recomputed; dq accumulated in registers; the gradient of
// every gathered table row is added to a (table_rows, dim) fp32 buffer with atomics (the summation order over rows that
// sampled the same item is not fixed: the one non-deterministic sum of this op, as index_put(accumulate) is in the
// reference's backward).  Rows with zero upstream gradient (masked positions: supervision weight 0) are skipped.
#include "hstu_common.cuh"
#include "capi_internal.h"

namespace hstu {

constexpr int kLossThreads = 256;
constexpr int kLossWaves = kLossThreads / 64;
constexpr float kMaskedLogit = -5e4f;   // sampled_softmax.py:80

template <typename T> struct LossVec;
template <> struct LossVec<float> { static constexpr int N = 4; };
template <> struct LossVec<bf16_t> { static constexpr int N = 8; };
template <> struct LossVec<f16_t> { static constexpr int N = 8; };

// 16 bytes of a row as floats (zeros when the lane's slice is past the row)
template <typename T>
HSTU_DEV void load16f(float (&v)[LossVec<T>::N], const T* p, bool ok) {
  constexpr int N = LossVec<T>::N;
  if (!ok) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0.f;
    return;
  }
  if constexpr (N == 8) {
    const u32x4 x = *reinterpret_cast<const u32x4*>(p);
    typedef T t8 __attribute__((ext_vector_type(8)));
    const t8 t = __builtin_bit_cast(t8, x);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  } else {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = t[i];
  }
}
template <typename T>
HSTU_DEV void store16f(const float (&v)[LossVec<T>::N], T* p) {
  constexpr int N = LossVec<T>::N;
  if constexpr (N == 8) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (T)v[i];
    *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, t);
  } else {
    const f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
}

// sum over the LPE lanes of a lane group (LPE a power of two, groups aligned)
template <int LPE>
HSTU_DEV float group_sum(float x) {
#pragma unroll
  for (int m = 1; m < LPE; m <<= 1) x += __shfl_xor(x, m, 64);
  return x;
}

struct LossArgs {
  const void* q; const void* pos; const void* table;
  const int64_t* pos_ids; const int64_t* neg_rows; const int64_t* neg_ids;
  int64_t q_stride, pos_stride, table_stride, table_rows, n_rows;
  int num_neg, dim;
  float inv_t, eps;
  int pos_l2, table_l2;
};

// (running max, running sum) merge
HSTU_DEV void lse_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) return;                       // both sides empty (exp(-inf - -inf) would be NaN)
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

// logit of one gathered row from its reduced dot product / squared norm
HSTU_DEV float neg_logit(float dot, float nn, const LossArgs& a, bool masked, float& inv_c) {
  inv_c = a.table_l2 ? 1.0f / fmaxf(sqrtf(nn), a.eps) : 1.0f;
  return masked ? kMaskedLogit : dot * inv_c * a.inv_t;
}

template <typename T, int LPE>
__global__ __launch_bounds__(kLossThreads) void sampled_softmax_fwd_kernel(const LossArgs a, const float* g_row,
                                                                           float* row_loss, float* lse_out) {
  constexpr int N = LossVec<T>::N;
  constexpr int G = 64 / LPE;                       // negatives per wave per step
  __shared__ float sm[kLossWaves], ss[kLossWaves];
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPE - 1), grp = lane / LPE;
  const bool ok = sub * N < a.dim;
  float qv[N], pv[N];
  load16f<T>(qv, (const T*)a.q + i * a.q_stride + sub * N, ok);
  load16f<T>(pv, (const T*)a.pos + i * a.pos_stride + sub * N, ok);
  const int64_t pid = a.pos_ids[i];
  float dp = 0.f, pp = 0.f;
#pragma unroll
  for (int f = 0; f < N; ++f) { dp += qv[f] * pv[f]; pp += pv[f] * pv[f]; }
  dp = group_sum<LPE>(dp);
  pp = group_sum<LPE>(pp);
  const float l0 = dp * (a.pos_l2 ? 1.0f / fmaxf(sqrtf(pp), a.eps) : 1.0f) * a.inv_t;

  float m = -INFINITY, s = 0.f;
  const int64_t* rows = a.neg_rows + i * a.num_neg;
  const int64_t* ids = a.neg_ids + i * a.num_neg;
  for (int k0 = wave * G; k0 < a.num_neg; k0 += kLossWaves * G) {
    const int k = k0 + grp;
    const bool live = k < a.num_neg;
    int64_t r = live ? rows[k] : 0;
    r = r < 0 ? 0 : (r >= a.table_rows ? a.table_rows - 1 : r);     // memory safety; a bad index is the caller's bug
    const bool masked = live && ids[k] == pid;
    float nv[N];
    load16f<T>(nv, (const T*)a.table + r * a.table_stride + sub * N, ok && live);
    float d = 0.f, nn = 0.f;
#pragma unroll
    for (int f = 0; f < N; ++f) { d += qv[f] * nv[f]; nn += nv[f] * nv[f]; }
    d = group_sum<LPE>(d);
    nn = group_sum<LPE>(nn);
    float inv_c;
    const float l = neg_logit(d, nn, a, masked, inv_c);
    if (live) lse_merge(m, s, l, 1.0f);
  }
  // merge the G lane groups of the wave (every lane of a group holds the same (m, s)), then the waves
#pragma unroll
  for (int sh = LPE; sh < 64; sh <<= 1) {
    const float m2 = __shfl_xor(m, sh, 64), s2 = __shfl_xor(s, sh, 64);
    lse_merge(m, s, m2, s2);
  }
  if (lane == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mt = l0, st = 1.0f;
#pragma unroll
    for (int w = 0; w < kLossWaves; ++w)
      if (ss[w] > 0.f) lse_merge(mt, st, sm[w], ss[w]);
    const float lse = mt + __logf(st);
    lse_out[i] = lse;
    row_loss[i] = lse - l0;
  }
  (void)g_row;
}

template <typename T, int LPE>
__global__ __launch_bounds__(kLossThreads) void sampled_softmax_bwd_kernel(const LossArgs a, const float* lse_in,
                                                                           const float* g_row, T* dq, int64_t dq_stride,
                                                                           T* dpos, int64_t dpos_stride, float* dtable) {
  constexpr int N = LossVec<T>::N;
  constexpr int G = 64 / LPE;
  __shared__ float red[kLossWaves][LPE * N];
  __shared__ __attribute__((aligned(16))) float tr[kLossWaves][64 * N];   // per-wave transpose tile
  const int64_t i = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & (LPE - 1), grp = lane / LPE;
  const bool ok = sub * N < a.dim;
  const float g = g_row[i];
  float zero[N];
#pragma unroll
  for (int f = 0; f < N; ++f) zero[f] = 0.f;
  if (g == 0.f) {   // masked position: exact zeros, nothing gathered
    if (wave == 0 && grp == 0 && ok) {
      store16f<T>(zero, dq + i * dq_stride + sub * N);
      store16f<T>(zero, dpos + i * dpos_stride + sub * N);
    }
    return;
  }
  float qv[N], pv[N];
  load16f<T>(qv, (const T*)a.q + i * a.q_stride + sub * N, ok);
  load16f<T>(pv, (const T*)a.pos + i * a.pos_stride + sub * N, ok);
  const int64_t pid = a.pos_ids[i];
  const float lse = lse_in[i];
  float dp = 0.f, pp = 0.f;
#pragma unroll
  for (int f = 0; f < N; ++f) { dp += qv[f] * pv[f]; pp += pv[f] * pv[f]; }
  dp = group_sum<LPE>(dp);
  pp = group_sum<LPE>(pp);
  const float pn = sqrtf(pp);
  const float inv_pc = a.pos_l2 ? 1.0f / fmaxf(pn, a.eps) : 1.0f;
  const float l0 = dp * inv_pc * a.inv_t;
  const float c0 = g * (__expf(l0 - lse) - 1.0f) * a.inv_t;      // d loss / d <q, pos_hat>
  float dqv[N];
#pragma unroll
  for (int f = 0; f < N; ++f) dqv[f] = 0.f;

  const int64_t* rows = a.neg_rows + i * a.num_neg;
  const int64_t* ids = a.neg_ids + i * a.num_neg;
  for (int k0 = wave * G; k0 < a.num_neg; k0 += kLossWaves * G) {
    const int k = k0 + grp;
    const bool live = k < a.num_neg;
    int64_t r = live ? rows[k] : 0;
    r = r < 0 ? 0 : (r >= a.table_rows ? a.table_rows - 1 : r);
    const bool masked = live && ids[k] == pid;
    float nv[N];
    load16f<T>(nv, (const T*)a.table + r * a.table_stride + sub * N, ok && live);
    float d = 0.f, nn = 0.f;
#pragma unroll
    for (int f = 0; f < N; ++f) { d += qv[f] * nv[f]; nn += nv[f] * nv[f]; }
    d = group_sum<LPE>(d);
    nn = group_sum<LPE>(nn);
    float inv_c;
    const float l = neg_logit(d, nn, a, masked, inv_c);
    // torch.where picks the constant for a masked negative: no gradient reaches the similarity
    const float c = (live && !masked) ? g * __expf(l - lse) * a.inv_t : 0.f;
    // n_hat = nv * inv_c;  dq += c * n_hat;  d n_hat = c * q;  d nv = (d n_hat - n_hat <n_hat, d n_hat>) * inv_c, or
    // d n_hat * inv_c when the clamp is active (zero gradient through the norm)
    const bool clamp_on = a.table_l2 && sqrtf(nn) < a.eps;
    const float proj = (a.table_l2 && !clamp_on) ? c * d * inv_c * inv_c * inv_c : 0.f;   // c <n_hat, q> / |n| applied to nv / |n|
#pragma unroll
    for (int f = 0; f < N; ++f) dqv[f] += c * inv_c * nv[f];       // c == 0 for dead / masked negatives
    // table gradient of this negative: a lane holds N consecutive features, but an atomic instruction is served
    // one cache line at a time (measured: 16-byte-strided lanes 1624 us, LPE consecutive floats per negative 421 us
    // at the Books shape), so the wave's G x (LPE * N) tile goes through LDS and is added 64 CONSECUTIVE floats per
    // instruction.
    float* tw = tr[wave] + grp * (LPE * N);
    if constexpr (N == 4) {
      *reinterpret_cast<f32x4*>(tw + sub * N) = f32x4{c * inv_c * qv[0] - proj * nv[0], c * inv_c * qv[1] - proj * nv[1],
                                                      c * inv_c * qv[2] - proj * nv[2], c * inv_c * qv[3] - proj * nv[3]};
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        *reinterpret_cast<f32x4*>(tw + sub * N + 4 * h) =
            f32x4{c * inv_c * qv[4 * h] - proj * nv[4 * h], c * inv_c * qv[4 * h + 1] - proj * nv[4 * h + 1],
                  c * inv_c * qv[4 * h + 2] - proj * nv[4 * h + 2], c * inv_c * qv[4 * h + 3] - proj * nv[4 * h + 3]};
    }
    // 64 consecutive floats of the tile per instruction: [negative g][feature e], rows of LPE * N floats
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int idx = j * 64 + lane;
      const int gsrc = idx / (LPE * N), e = idx & (LPE * N - 1);
      const int64_t rr = __shfl(r, gsrc * LPE, 64);
      const float cc = __shfl(c, gsrc * LPE, 64);
      if (cc != 0.f && e < a.dim) atomicAdd(dtable + rr * (int64_t)a.dim + e, tr[wave][idx]);
    }
  }
  // dq: sum over the G lane groups (shuffles), then over the waves (LDS), plus the positive's term
#pragma unroll
  for (int f = 0; f < N; ++f)
#pragma unroll
    for (int sh = LPE; sh < 64; sh <<= 1) dqv[f] += __shfl_xor(dqv[f], sh, 64);
  if (grp == 0)
#pragma unroll
    for (int f = 0; f < N; ++f) red[wave][sub * N + f] = dqv[f];
  __syncthreads();
  if (wave == 0 && grp == 0 && ok) {
    float o[N], dpv[N];
    const bool pclamp = a.pos_l2 && pn < a.eps;
    const float pproj = (a.pos_l2 && !pclamp) ? c0 * dp * inv_pc * inv_pc * inv_pc : 0.f;
#pragma unroll
    for (int f = 0; f < N; ++f) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < kLossWaves; ++w) acc += red[w][sub * N + f];
      o[f] = acc + c0 * inv_pc * pv[f];
      dpv[f] = c0 * inv_pc * qv[f] - pproj * pv[f];
    }
    store16f<T>(o, dq + i * dq_stride + sub * N);
    store16f<T>(dpv, dpos + i * dpos_stride + sub * N);
  }
}

template <typename T, int LPE>
static int loss_launch_lpe(const LossArgs& a, bool bwd, const float* lse_in, const float* g_row, float* row_loss,
                           float* lse_out, void* dq, int64_t dq_stride, void* dpos, int64_t dpos_stride, float* dtable,
                           hipStream_t st) {
  const dim3 grid((unsigned)a.n_rows), block(kLossThreads);
  if (!bwd) {
    hipLaunchKernelGGL((sampled_softmax_fwd_kernel<T, LPE>), grid, block, 0, st, a, g_row, row_loss, lse_out);
    return check_launch("hstu_sampled_softmax_fwd");
  }
  hipLaunchKernelGGL((sampled_softmax_bwd_kernel<T, LPE>), grid, block, 0, st, a, lse_in, g_row, (T*)dq, dq_stride, (T*)dpos,
                     dpos_stride, dtable);
  return check_launch("hstu_sampled_softmax_bwd");
}

template <typename T>
static int loss_launch(const LossArgs& a, bool bwd, const float* lse_in, const float* g_row, float* row_loss, float* lse_out,
                       void* dq, int64_t dq_stride, void* dpos, int64_t dpos_stride, float* dtable, hipStream_t st) {
  const int units = a.dim / LossVec<T>::N;   // 16-byte units per embedding
#define LPE_CASE(L) \
  if (units <= L) return loss_launch_lpe<T, L>(a, bwd, lse_in, g_row, row_loss, lse_out, dq, dq_stride, dpos, dpos_stride, dtable, st);
  LPE_CASE(1) LPE_CASE(2) LPE_CASE(4) LPE_CASE(8) LPE_CASE(16) LPE_CASE(32) LPE_CASE(64)
#undef LPE_CASE
  return set_error(HSTU_EUNSUPPORTED, "sampled_softmax: embedding dim %d is above 64 x 16 bytes", a.dim);
}

static int loss_check(const char* who, const LossArgs& a, int dtype) {
  if (!a.q || !a.pos || !a.table || !a.pos_ids || !a.neg_rows || !a.neg_ids) return set_error(HSTU_EINVAL, "%s: NULL tensor", who);
  const int vec = dtype == HSTU_DTYPE_F32 ? 4 : 8;
  if (a.dim <= 0 || a.dim % vec) return set_error(HSTU_EINVAL, "%s: embedding dim %d must be a multiple of %d (16 bytes)", who, a.dim, vec);
  if (a.q_stride % vec || a.pos_stride % vec || a.table_stride % vec)
    return set_error(HSTU_EINVAL, "%s: row strides must keep rows 16-byte aligned", who);
  if (((uintptr_t)a.q | (uintptr_t)a.pos | (uintptr_t)a.table) & 15) return set_error(HSTU_EINVAL, "%s: base pointers must be 16-byte aligned", who);
  if (a.num_neg <= 0 || a.table_rows <= 0) return set_error(HSTU_EINVAL, "%s: num_negatives and table_rows must be positive", who);
  if (!(a.inv_t > 0.f) || !(a.inv_t < INFINITY)) return set_error(HSTU_EINVAL, "%s: temperature must be positive and finite", who);
  if (a.n_rows > 0x7fffffffLL) return set_error(HSTU_EINVAL, "%s: too many rows", who);
  return HSTU_OK;
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_sampled_softmax_fwd(const void* q, int64_t q_row_stride, const void* pos_emb, int64_t pos_row_stride,
                             const int64_t* pos_ids, const int64_t* neg_rows, const int64_t* neg_ids, const void* table,
                             int64_t table_row_stride, int64_t table_rows, int64_t n_rows, int32_t num_negatives,
                             int32_t dim, float temperature, int32_t pos_l2_norm, int32_t table_l2_norm, float eps,
                             float* row_loss, float* lse, int dtype, void* stream) {
  if (n_rows == 0) return HSTU_OK;
  LossArgs a{q, pos_emb, table, pos_ids, neg_rows, neg_ids, q_row_stride, pos_row_stride, table_row_stride, table_rows,
             n_rows, num_negatives, dim, 1.0f / temperature, eps, pos_l2_norm, table_l2_norm};
  if (int e = loss_check("hstu_sampled_softmax_fwd", a, dtype)) return e;
  if (!row_loss || !lse) return set_error(HSTU_EINVAL, "hstu_sampled_softmax_fwd: row_loss and lse are required");
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HSTU_DTYPE_BF16: return loss_launch<bf16_t>(a, false, nullptr, nullptr, row_loss, lse, nullptr, 0, nullptr, 0, nullptr, st);
    case HSTU_DTYPE_F16: return loss_launch<f16_t>(a, false, nullptr, nullptr, row_loss, lse, nullptr, 0, nullptr, 0, nullptr, st);
    case HSTU_DTYPE_F32: return loss_launch<float>(a, false, nullptr, nullptr, row_loss, lse, nullptr, 0, nullptr, 0, nullptr, st);
    default: return set_error(HSTU_EINVAL, "dtype must be bf16, fp16 or fp32");
  }
}

int hstu_sampled_softmax_bwd(const void* q, int64_t q_row_stride, const void* pos_emb, int64_t pos_row_stride,
                             const int64_t* pos_ids, const int64_t* neg_rows, const int64_t* neg_ids, const void* table,
                             int64_t table_row_stride, int64_t table_rows, int64_t n_rows, int32_t num_negatives,
                             int32_t dim, float temperature, int32_t pos_l2_norm, int32_t table_l2_norm, float eps,
                             const float* lse, const float* grad_row_loss, void* dq, int64_t dq_row_stride, void* dpos_emb,
                             int64_t dpos_row_stride, float* dtable, int dtype, void* stream) {
  if (n_rows == 0) return HSTU_OK;
  LossArgs a{q, pos_emb, table, pos_ids, neg_rows, neg_ids, q_row_stride, pos_row_stride, table_row_stride, table_rows,
             n_rows, num_negatives, dim, 1.0f / temperature, eps, pos_l2_norm, table_l2_norm};
  if (int e = loss_check("hstu_sampled_softmax_bwd", a, dtype)) return e;
  if (!lse || !grad_row_loss || !dq || !dpos_emb || !dtable) return set_error(HSTU_EINVAL, "hstu_sampled_softmax_bwd: NULL tensor");
  const int vec = dtype == HSTU_DTYPE_F32 ? 4 : 8;
  if (dq_row_stride % vec || dpos_row_stride % vec || (((uintptr_t)dq | (uintptr_t)dpos_emb) & 15))
    return set_error(HSTU_EINVAL, "hstu_sampled_softmax_bwd: gradient rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HSTU_DTYPE_BF16: return loss_launch<bf16_t>(a, true, lse, grad_row_loss, nullptr, nullptr, dq, dq_row_stride, dpos_emb, dpos_row_stride, dtable, st);
    case HSTU_DTYPE_F16: return loss_launch<f16_t>(a, true, lse, grad_row_loss, nullptr, nullptr, dq, dq_row_stride, dpos_emb, dpos_row_stride, dtable, st);
    case HSTU_DTYPE_F32: return loss_launch<float>(a, true, lse, grad_row_loss, nullptr, nullptr, dq, dq_row_stride, dpos_emb, dpos_row_stride, dtable, st);
    default: return set_error(HSTU_EINVAL, "dtype must be bf16, fp16 or fp32");
  }
}

}  // extern "C"
