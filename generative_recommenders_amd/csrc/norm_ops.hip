// Row-wise normalisation kernels around the HSTU projections (gfx950, HBM-bound).
//
// One wavefront owns one row at a time (grid-stride over rows): each lane loads its
// 16-byte pieces of the row once, keeps them in registers, reduces with wave shuffles,
// and writes once.  Weight gradients are accumulated per lane across all the rows a
// wave visits (lanes own fixed columns), reduced across the workgroup's waves through
// LDS and written as one fp32 partial per workgroup; a second small kernel sums the
// partials.  All math is fp32, outputs are cast to the I/O dtype, like the reference:
//   ops/pytorch/pt_layer_norm.py:24-38, ops/pytorch/pt_hstu_linear.py:23-65;
// kernels replaced: ops/triton/triton_layer_norm.py:77-309,
// ops/triton/triton_hstu_linear.py:48-337 (LN * u), :570-1036 (GroupNorm * u).
#include "hstu_common.cuh"
#include "capi_internal.h"

namespace hstu {

constexpr int kNormThreads = 256;
constexpr int kNormWaves = kNormThreads / 64;
// register-resident pieces per lane: 16-bit x8 -> dim <= 1024, fp32 x4 -> dim <= 1024, scalar -> dim <= 512
template <int VEC> constexpr int max_chunks() { return VEC == 8 ? 2 : (VEC == 4 ? 4 : 8); }
constexpr int kMaxNormBlocks = 2048;

template <typename T, int VEC> struct RowVec {
  float v[VEC];
};

template <typename T, int VEC>
HSTU_DEV void load_vec(RowVec<T, VEC>& r, const T* p, bool ok) {
  if (!ok) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) r.v[i] = 0.f;
    return;
  }
  if constexpr (VEC == 1) {
    r.v[0] = (float)p[0];
  } else if constexpr (sizeof(T) == 2) {  // VEC == 8
    u32x4 x = *reinterpret_cast<const u32x4*>(p);
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 t = __builtin_bit_cast(t8, x);
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (float)t[i];
  } else {  // fp32, VEC == 4
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = t[i];
  }
}

template <typename T, int VEC>
HSTU_DEV void store_vec(const RowVec<T, VEC>& r, T* p) {
  if constexpr (VEC == 1) {
    p[0] = (T)r.v[0];
  } else if constexpr (sizeof(T) == 2) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (T)r.v[i];
    *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, t);
  } else {
    f32x4 t = {r.v[0], r.v[1], r.v[2], r.v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
}

HSTU_DEV float wave_sum(float x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  return x;
}

// ------------------------------------------------------------------ fused dropout of the output stage
// Replaces the Philox dropout inside _ln_mul_dropout_fwd / _group_norm_mul_dropout_fwd (triton_hstu_linear.py:101-120,
// 631-652; backward :196-215, :718-738): the mask of an output element is a pure function of (seed, element index in the
// (rows, out_stride) output), so the backward -- and the forward recompute of y -- regenerate it instead of storing it.
// One hash gives 32 bits = two 16-bit uniforms for the elements 2j and 2j+1: keep iff r16 >= thr, thr = round(p 65536),
// survivors scaled by 65536 / (65536 - thr) (the exact inverse of the keep probability).  The hash is two rounds of
// 32-bit multiply-xorshift finalisers (murmur3 fmix32, then lowbias32) with one seed word folded in before each;
// the CPU checker of the tests restates it bit for bit (dropout_keep_mask).
struct DropCtx {
  uint32_t thr;        // 0 = no dropout
  float scale;
  uint32_t s0, s1;
};
HSTU_DEV uint32_t drop_hash(uint32_t lo, uint32_t hi, uint32_t s0, uint32_t s1) {
  uint32_t h = lo ^ s0;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  h ^= s1 + hi * 0x9e3779b9u;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
// v[i] *= keep(e0 + i) ? scale : 0 for i < VEC; e0 = index of v[0] in the output tensor (even when VEC > 1)
template <typename T, int VEC>
HSTU_DEV void drop_apply(RowVec<T, VEC>& r, int64_t e0, const DropCtx& dc) {
  if constexpr (VEC == 1) {
    const uint64_t pe = (uint64_t)e0 >> 1;
    const uint32_t h = drop_hash((uint32_t)pe, (uint32_t)(pe >> 32), dc.s0, dc.s1);
    const uint32_t r16 = (e0 & 1) ? (h >> 16) : (h & 0xffffu);
    r.v[0] = r16 >= dc.thr ? r.v[0] * dc.scale : 0.f;
  } else {
    const uint64_t pe0 = (uint64_t)e0 >> 1;
#pragma unroll
    for (int j = 0; j < VEC / 2; ++j) {
      const uint64_t pe = pe0 + j;
      const uint32_t h = drop_hash((uint32_t)pe, (uint32_t)(pe >> 32), dc.s0, dc.s1);
      r.v[2 * j] = (h & 0xffffu) >= dc.thr ? r.v[2 * j] * dc.scale : 0.f;
      r.v[2 * j + 1] = (h >> 16) >= dc.thr ? r.v[2 * j + 1] * dc.scale : 0.f;
    }
  }
}
static DropCtx make_drop_ctx(float ratio, uint64_t seed) {
  DropCtx dc;
  double t = (double)ratio * 65536.0 + 0.5;
  dc.thr = ratio > 0.f ? (uint32_t)(t < 1.0 ? 1.0 : (t > 65535.0 ? 65535.0 : t)) : 0u;
  dc.scale = 65536.0f / (float)(65536u - dc.thr);
  dc.s0 = (uint32_t)seed;
  dc.s1 = (uint32_t)(seed >> 32);
  return dc;
}

// ------------------------------------------------------------------ layer norm
template <typename T, int VEC>
__global__ __launch_bounds__(kNormThreads) void layer_norm_fwd_kernel(const T* x, const T* w, const T* b, T* y,
                                                                      float* mean_out, float* rstd_out, int64_t rows,
                                                                      int dim, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  RowVec<T, VEC> wv[max_chunks<VEC>()], bv[max_chunks<VEC>()];
#pragma unroll
  for (int k = 0; k < max_chunks<VEC>(); ++k) {
    const int c = (k * 64 + lane) * VEC;
    load_vec<T, VEC>(wv[k], w + c, k < nch && c < dim);
    load_vec<T, VEC>(bv[k], b + c, k < nch && c < dim);
  }
  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    RowVec<T, VEC> xv[max_chunks<VEC>()];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      load_vec<T, VEC>(xv[k], x + row * dim + c, k < nch && c < dim);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s += xv[k].v[i];
    }
    const float mean = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (k < nch && c < dim) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const float d = xv[k].v[i] - mean; q += d * d; }
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / dim + eps);
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (k < nch && c < dim) {
        RowVec<T, VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = (xv[k].v[i] - mean) * rstd * wv[k].v[i] + bv[k].v[i];
        store_vec<T, VEC>(o, y + row * dim + c);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

// block-level reduction of per-lane column partials -> one fp32 partial row per workgroup
template <int VEC>
HSTU_DEV void block_reduce_cols(float (&acc)[max_chunks<VEC>()][VEC], float* lds, float* out_row, int dim, int nch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lds: [kNormWaves][dim_padded]
  const int dpad = nch * 64 * VEC;
#pragma unroll
  for (int k = 0; k < max_chunks<VEC>(); ++k)
    if (k < nch)
#pragma unroll
      for (int i = 0; i < VEC; ++i) lds[wave * dpad + (k * 64 + lane) * VEC + i] = acc[k][i];
  __syncthreads();
  for (int c = threadIdx.x; c < dim; c += kNormThreads) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kNormWaves; ++w2) s += lds[w2 * dpad + c];
    out_row[c] = s;
  }
  __syncthreads();
}

template <typename T, int VEC>
__global__ __launch_bounds__(kNormThreads) void layer_norm_bwd_kernel(const T* dy, const T* x, const T* w,
                                                                      const float* mean_in, const float* rstd_in, T* dx,
                                                                      float* partial, int64_t rows, int dim) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  RowVec<T, VEC> wv[max_chunks<VEC>()];
  float dw[max_chunks<VEC>()][VEC], db[max_chunks<VEC>()][VEC];
#pragma unroll
  for (int k = 0; k < max_chunks<VEC>(); ++k) {
    const int c = (k * 64 + lane) * VEC;
    load_vec<T, VEC>(wv[k], w + c, k < nch && c < dim);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { dw[k][i] = 0.f; db[k][i] = 0.f; }
  }
  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    RowVec<T, VEC> xh[max_chunks<VEC>()], g[max_chunks<VEC>()];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      const bool ok = k < nch && c < dim;
      RowVec<T, VEC> dyv;
      load_vec<T, VEC>(xh[k], x + row * dim + c, ok);
      load_vec<T, VEC>(dyv, dy + row * dim + c, ok);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xhat = ok ? (xh[k].v[i] - mean) * rstd : 0.f;
        xh[k].v[i] = xhat;
        g[k].v[i] = dyv.v[i] * wv[k].v[i];
        s1 += g[k].v[i] * xhat;
        s2 += g[k].v[i];
        dw[k][i] += dyv.v[i] * xhat;
        db[k][i] += dyv.v[i];
      }
    }
    const float c1 = wave_sum(s1) / dim, c2 = wave_sum(s2) / dim;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (k < nch && c < dim) {
        RowVec<T, VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = rstd * (g[k].v[i] - c2 - xh[k].v[i] * c1);
        store_vec<T, VEC>(o, dx + row * dim + c);
      }
    }
  }
  block_reduce_cols<VEC>(dw, lds, partial + (int64_t)blockIdx.x * 2 * dim, dim, nch);
  block_reduce_cols<VEC>(db, lds, partial + (int64_t)blockIdx.x * 2 * dim + dim, dim, nch);
}

// Column sums of the (nparts, 2 * width) partial matrix: columns [0, width) -> out_w, [width, 2 width) -> out_b.
// One 256-thread block per column: thread t adds rows t, t + 256, ... in a fixed order, then the block combines
// the 256 sums in a fixed tree (deterministic).  (One thread per column walking all the rows is latency bound:
// with group norm the width is the number of heads, i.e. 4 threads for ~1000 dependent loads.)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partial, int nparts, int width, float* out_w,
                                                              float* out_b) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[(int64_t)i * (2 * width) + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (c < width) out_w[c] = red[0];
    else out_b[c - width] = red[0];
  }
}

// ------------------------------------------------------------------ y = u * Norm(attn) [, concat]
// LN: statistics over the whole row, weight/bias per column.
// GN: statistics per head (head_dim columns), weight/bias per head.
template <typename T, int VEC, bool GN>
__global__ __launch_bounds__(kNormThreads) void norm_mul_fwd_kernel(const T* attn, const T* u, const T* w, const T* b,
                                                                    T* y, float* mean_out, float* rstd_out,
                                                                    int64_t rows, int heads, int hdim, float eps,
                                                                    int concat, DropCtx dc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = heads * hdim;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  const int ostride = concat ? 3 * dim : dim;
  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    RowVec<T, VEC> xv[max_chunks<VEC>()], uv[max_chunks<VEC>()];
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      const bool ok = k < nch && c < dim;
      load_vec<T, VEC>(xv[k], attn + row * dim + c, ok);
      load_vec<T, VEC>(uv[k], u + row * dim + c, ok);
    }
    const int ngroups = GN ? heads : 1;
    const int gdim = GN ? hdim : dim;
    for (int gi = 0; gi < ngroups; ++gi) {
      const int lo = gi * gdim, hi = lo + gdim;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < max_chunks<VEC>(); ++k) {
        const int c = (k * 64 + lane) * VEC;
        if (k < nch && c >= lo && c < hi)
#pragma unroll
          for (int i = 0; i < VEC; ++i) s += xv[k].v[i];
      }
      const float mean = wave_sum(s) / gdim;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < max_chunks<VEC>(); ++k) {
        const int c = (k * 64 + lane) * VEC;
        if (k < nch && c >= lo && c < hi)
#pragma unroll
          for (int i = 0; i < VEC; ++i) { const float d = xv[k].v[i] - mean; q += d * d; }
      }
      const float rstd = 1.0f / sqrtf(wave_sum(q) / gdim + eps);
      float gw = 0.f, gb = 0.f;
      if (GN) { gw = (float)w[gi]; gb = (float)b[gi]; }
#pragma unroll
      for (int k = 0; k < max_chunks<VEC>(); ++k) {
        const int c = (k * 64 + lane) * VEC;
        if (k < nch && c >= lo && c < hi) {
          RowVec<T, VEC> wv, bv, o;
          if (!GN) { load_vec<T, VEC>(wv, w + c, true); load_vec<T, VEC>(bv, b + c, true); }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float n = (xv[k].v[i] - mean) * rstd * (GN ? gw : wv.v[i]) + (GN ? gb : bv.v[i]);
            o.v[i] = uv[k].v[i] * n;
          }
          T* yrow = y + row * ostride;
          if (concat) {
            RowVec<T, VEC> ud = uv[k], xd = xv[k];
            if (dc.thr) {   // kernel-uniform
              drop_apply<T, VEC>(ud, row * ostride + c, dc);
              drop_apply<T, VEC>(xd, row * ostride + dim + c, dc);
              drop_apply<T, VEC>(o, row * ostride + 2 * dim + c, dc);
            }
            store_vec<T, VEC>(ud, yrow + c);
            store_vec<T, VEC>(xd, yrow + dim + c);
            store_vec<T, VEC>(o, yrow + 2 * dim + c);
          } else {
            if (dc.thr) drop_apply<T, VEC>(o, row * ostride + c, dc);
            store_vec<T, VEC>(o, yrow + c);
          }
        }
      }
      if (lane == 0) {
        if (mean_out) mean_out[row * ngroups + gi] = mean;
        if (rstd_out) rstd_out[row * ngroups + gi] = rstd;
      }
    }
  }
}

// backward of the above.  partial row layout per workgroup: [dweight(width) | dbias(width)],
// width = dim (LN) or heads (GN).
template <typename T, int VEC, bool GN>
__global__ __launch_bounds__(kNormThreads) void norm_mul_bwd_kernel(const T* dy, const T* attn, const T* u, const T* w,
                                                                    const T* b, const float* mean_in,
                                                                    const float* rstd_in, T* dattn, T* du,
                                                                    float* partial, int64_t rows, int heads, int hdim,
                                                                    int concat, DropCtx dc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = heads * hdim;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  const int istride = concat ? 3 * dim : dim;
  const int ngroups = GN ? heads : 1;
  const int gdim = GN ? hdim : dim;
  float dw[max_chunks<VEC>()][VEC], db[max_chunks<VEC>()][VEC];   // LN: per column.  GN: slot [0][0..] unused, see ghw/ghb
#pragma unroll
  for (int k = 0; k < max_chunks<VEC>(); ++k)
#pragma unroll
    for (int i = 0; i < VEC; ++i) { dw[k][i] = 0.f; db[k][i] = 0.f; }
  float ghw[16], ghb[16];                             // GN: per-head partials (heads <= 16), lane-local
#pragma unroll
  for (int h = 0; h < 16; ++h) { ghw[h] = 0.f; ghb[h] = 0.f; }

  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    RowVec<T, VEC> xv[max_chunks<VEC>()], uv[max_chunks<VEC>()], gy[max_chunks<VEC>()];
    const T* dyrow = dy + row * istride;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      const bool ok = k < nch && c < dim;
      load_vec<T, VEC>(xv[k], attn + row * dim + c, ok);
      load_vec<T, VEC>(uv[k], u + row * dim + c, ok);
      load_vec<T, VEC>(gy[k], dyrow + (concat ? 2 * dim : 0) + c, ok);
      if (dc.thr && ok) drop_apply<T, VEC>(gy[k], row * istride + (concat ? 2 * dim : 0) + c, dc);   // d y3 -> d y: same mask, same scale
    }
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) {
      if (gi < ngroups) {
        const int lo = gi * gdim, hi = lo + gdim;
        const float mean = mean_in[row * ngroups + gi], rstd = rstd_in[row * ngroups + gi];
        float gw = 0.f, gb = 0.f;
        if (GN) { gw = (float)w[gi]; gb = (float)b[gi]; }
        float s1 = 0.f, s2 = 0.f, hw = 0.f, hb = 0.f;
        RowVec<T, VEC> gg[max_chunks<VEC>()], xh[max_chunks<VEC>()];
#pragma unroll
        for (int k = 0; k < max_chunks<VEC>(); ++k) {
          const int c = (k * 64 + lane) * VEC;
          const bool in = k < nch && c >= lo && c < hi;
          RowVec<T, VEC> wv, bv;
          if (!GN) { load_vec<T, VEC>(wv, w + c, in); load_vec<T, VEC>(bv, b + c, in); }
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const float xhat = in ? (xv[k].v[i] - mean) * rstd : 0.f;
            const float gam = GN ? gw : wv.v[i];
            const float bet = GN ? gb : bv.v[i];
            const float dyu = in ? gy[k].v[i] * uv[k].v[i] : 0.f;   // d(norm out)
            xh[k].v[i] = xhat;
            gg[k].v[i] = dyu * gam;
            s1 += gg[k].v[i] * xhat;
            s2 += gg[k].v[i];
            if (GN) { hw += dyu * xhat; hb += dyu; }
            else if (in) { dw[k][i] += dyu * xhat; db[k][i] += dyu; }
            if (in) uv[k].v[i] = gy[k].v[i] * (gam * xhat + bet);      // du contribution dy * n_hat (reuses uv)
          }
        }
        if (GN) { ghw[gi] += hw; ghb[gi] += hb; }
        const float c1 = wave_sum(s1) / gdim, c2 = wave_sum(s2) / gdim;
#pragma unroll
        for (int k = 0; k < max_chunks<VEC>(); ++k) {
          const int c = (k * 64 + lane) * VEC;
          if (k < nch && c >= lo && c < hi) {
            RowVec<T, VEC> o, o2, e1, e2;
            if (concat) {
              load_vec<T, VEC>(e1, dyrow + c, true);          // d u   from the concat slot
              load_vec<T, VEC>(e2, dyrow + dim + c, true);    // d attn from the concat slot
              if (dc.thr) {
                drop_apply<T, VEC>(e1, row * istride + c, dc);
                drop_apply<T, VEC>(e2, row * istride + dim + c, dc);
              }
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              o.v[i] = rstd * (gg[k].v[i] - c2 - xh[k].v[i] * c1) + (concat ? e2.v[i] : 0.f);
              o2.v[i] = uv[k].v[i] + (concat ? e1.v[i] : 0.f);
            }
            store_vec<T, VEC>(o, dattn + row * dim + c);
            store_vec<T, VEC>(o2, du + row * dim + c);
          }
        }
      }
    }
  }
  if (GN) {
    // per-head partials: reduce over the wave, then over the workgroup's waves
    float* out_row = partial + (int64_t)blockIdx.x * 2 * heads;
#pragma unroll
    for (int h = 0; h < 16; ++h) {
      if (h < heads) {
        const float a = wave_sum(ghw[h]), c = wave_sum(ghb[h]);
        if (lane == 0) { lds[wave * 32 + h] = a; lds[wave * 32 + 16 + h] = c; }
      }
    }
    __syncthreads();
    if (threadIdx.x < heads) {
      float a = 0.f, c = 0.f;
      for (int w2 = 0; w2 < kNormWaves; ++w2) { a += lds[w2 * 32 + threadIdx.x]; c += lds[w2 * 32 + 16 + threadIdx.x]; }
      out_row[threadIdx.x] = a;
      out_row[heads + threadIdx.x] = c;
    }
  } else {
    block_reduce_cols<VEC>(dw, lds, partial + (int64_t)blockIdx.x * 2 * dim, dim, nch);
    block_reduce_cols<VEC>(db, lds, partial + (int64_t)blockIdx.x * 2 * dim + dim, dim, nch);
  }
}

// ------------------------------------------------------------------ group norm * u, fast path
// Group norm with head_dim / VEC lanes per head (a power of two <= 64, e.g. 128 / 8 = 16): every lane's 16-byte
// piece lies inside ONE head, so all heads of a row are normalised at once with segmented (xor-shuffle) sums over
// the lanes of a head.  The general kernels above walk the heads one after the other with the other lanes
// masked (heads x the arithmetic) and fetch the concat slots in a second round trip per row.
HSTU_DEV float seg_sum(float x, int lanes_per_head) {
  for (int d = 1; d < lanes_per_head; d <<= 1) x += __shfl_xor(x, d, 64);
  return x;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kNormThreads) void norm_mul_fwd_gn_kernel(const T* attn, const T* u, const T* w, const T* b,
                                                                       T* y, float* mean_out, float* rstd_out,
                                                                       int64_t rows, int heads, int hdim, float eps,
                                                                       int concat, DropCtx dc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = heads * hdim;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  const int lph = hdim / VEC;
  const int ostride = concat ? 3 * dim : dim;
  const float inv = 1.0f / hdim;
  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    RowVec<T, VEC> xv[max_chunks<VEC>()], uv[max_chunks<VEC>()];
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      const bool ok = k < nch && c < dim;
      load_vec<T, VEC>(xv[k], attn + row * dim + c, ok);
      load_vec<T, VEC>(uv[k], u + row * dim + c, ok);
    }
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (k < nch) {               // wave-uniform; lanes past dim take part in the shuffles with zeros
        const bool ok = c < dim;
        const int h = ok ? c / hdim : 0;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += xv[k].v[i];
        const float mean = seg_sum(s, lph) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) { const float d = xv[k].v[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(seg_sum(q, lph) * inv + eps);
        if (ok) {
          const float gw = (float)w[h], gb = (float)b[h];
          RowVec<T, VEC> o;
#pragma unroll
          for (int i = 0; i < VEC; ++i) o.v[i] = uv[k].v[i] * ((xv[k].v[i] - mean) * rstd * gw + gb);
          T* yrow = y + row * ostride;
          if (concat) {
            if (dc.thr) {   // kernel-uniform
              drop_apply<T, VEC>(uv[k], row * ostride + c, dc);
              drop_apply<T, VEC>(xv[k], row * ostride + dim + c, dc);
              drop_apply<T, VEC>(o, row * ostride + 2 * dim + c, dc);
            }
            store_vec<T, VEC>(uv[k], yrow + c);
            store_vec<T, VEC>(xv[k], yrow + dim + c);
            store_vec<T, VEC>(o, yrow + 2 * dim + c);
          } else {
            if (dc.thr) drop_apply<T, VEC>(o, row * ostride + c, dc);
            store_vec<T, VEC>(o, yrow + c);
          }
          if (c % hdim == 0) {
            if (mean_out) mean_out[row * heads + h] = mean;
            if (rstd_out) rstd_out[row * heads + h] = rstd;
          }
        }
      }
    }
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kNormThreads) void norm_mul_bwd_gn_kernel(const T* dy, const T* attn, const T* u, const T* w,
                                                                       const T* b, const float* mean_in,
                                                                       const float* rstd_in, T* dattn, T* du,
                                                                       float* partial, int64_t rows, int heads, int hdim,
                                                                       int concat, DropCtx dc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = heads * hdim;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  const int lph = hdim / VEC;
  const int istride = concat ? 3 * dim : dim;
  const float inv = 1.0f / hdim;
  // this lane's head per chunk never changes: its dweight / dbias partial sums stay in two registers per chunk
  float hw[max_chunks<VEC>()], hb[max_chunks<VEC>()], gw[max_chunks<VEC>()], gb[max_chunks<VEC>()];
  int head[max_chunks<VEC>()];
#pragma unroll
  for (int k = 0; k < max_chunks<VEC>(); ++k) {
    const int c = (k * 64 + lane) * VEC;
    const bool ok = k < nch && c < dim;
    head[k] = ok ? c / hdim : 0;
    gw[k] = ok ? (float)w[head[k]] : 0.f;
    gb[k] = ok ? (float)b[head[k]] : 0.f;
    hw[k] = 0.f;
    hb[k] = 0.f;
  }
  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    RowVec<T, VEC> xv[max_chunks<VEC>()], uv[max_chunks<VEC>()], gy[max_chunks<VEC>()], e1[max_chunks<VEC>()], e2[max_chunks<VEC>()];
    float mean[max_chunks<VEC>()], rstd[max_chunks<VEC>()];
    const T* dyrow = dy + row * istride;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {   // every load of the row is issued before anything is used
      const int c = (k * 64 + lane) * VEC;
      const bool ok = k < nch && c < dim;
      load_vec<T, VEC>(xv[k], attn + row * dim + c, ok);
      load_vec<T, VEC>(uv[k], u + row * dim + c, ok);
      load_vec<T, VEC>(gy[k], dyrow + (concat ? 2 * dim : 0) + c, ok);
      load_vec<T, VEC>(e1[k], dyrow + c, ok && concat);           // d u    from the concat slot
      load_vec<T, VEC>(e2[k], dyrow + dim + c, ok && concat);     // d attn from the concat slot
      mean[k] = ok ? mean_in[row * heads + head[k]] : 0.f;
      rstd[k] = ok ? rstd_in[row * heads + head[k]] : 0.f;
    }
    if (dc.thr) {   // kernel-uniform: d y3 -> the gradients of the three slices before dropout (same masks, same scale)
#pragma unroll
      for (int k = 0; k < max_chunks<VEC>(); ++k) {
        const int c = (k * 64 + lane) * VEC;
        if (k < nch && c < dim) {
          drop_apply<T, VEC>(gy[k], row * istride + (concat ? 2 * dim : 0) + c, dc);
          if (concat) {
            drop_apply<T, VEC>(e1[k], row * istride + c, dc);
            drop_apply<T, VEC>(e2[k], row * istride + dim + c, dc);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (k < nch) {
        const bool ok = c < dim;
        float s1 = 0.f, s2 = 0.f;
        RowVec<T, VEC> gg, xh, o, o2;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float xhat = (xv[k].v[i] - mean[k]) * rstd[k];
          const float dyu = gy[k].v[i] * uv[k].v[i];               // d(norm out)
          xh.v[i] = xhat;
          gg.v[i] = dyu * gw[k];
          s1 += gg.v[i] * xhat;
          s2 += gg.v[i];
          hw[k] += dyu * xhat;
          hb[k] += dyu;
          o2.v[i] = gy[k].v[i] * (gw[k] * xhat + gb[k]) + e1[k].v[i];
        }
        const float c1 = seg_sum(s1, lph) * inv, c2 = seg_sum(s2, lph) * inv;
        if (ok) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) o.v[i] = rstd[k] * (gg.v[i] - c2 - xh.v[i] * c1) + e2[k].v[i];
          store_vec<T, VEC>(o, dattn + row * dim + c);
          store_vec<T, VEC>(o2, du + row * dim + c);
        }
      }
    }
  }
  // per-head partials: lanes of a head -> its first lane -> LDS[wave][head] (a head lives in exactly one chunk) ->
  // fixed-order sum over the waves
  for (int i = threadIdx.x; i < kNormWaves * 32; i += kNormThreads) lds[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < max_chunks<VEC>(); ++k) {
    const int c = (k * 64 + lane) * VEC;
    if (k < nch) {
      const float a = seg_sum(hw[k], lph), cc = seg_sum(hb[k], lph);
      if (c < dim && c % hdim == 0) { lds[wave * 32 + head[k]] = a; lds[wave * 32 + 16 + head[k]] = cc; }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < heads) {
    float a = 0.f, cc = 0.f;
    for (int w2 = 0; w2 < kNormWaves; ++w2) { a += lds[w2 * 32 + threadIdx.x]; cc += lds[w2 * 32 + 16 + threadIdx.x]; }
    float* out_row = partial + (int64_t)blockIdx.x * 2 * heads;
    out_row[threadIdx.x] = a;
    out_row[heads + threadIdx.x] = cc;
  }
}

static bool gn_fast_ok(int hdim, int v) {
  if (v <= 1 || hdim % v) return false;
  const int lph = hdim / v;
  return lph <= 64 && (lph & (lph - 1)) == 0;
}

// ------------------------------------------------------------------ SiLU on a column slice
template <typename T, bool BWD>
__global__ void silu_kernel(const T* dout, const T* in, T* out, int64_t rows, int cols, int64_t s_dout, int64_t s_in,
                            int64_t s_out) {
  const int64_t n = rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i % cols);
    const float x = (float)in[r * s_in + c];
    const float sg = 1.0f / (1.0f + __expf(-x));
    if (BWD) out[r * s_out + c] = (T)((float)dout[r * s_dout + c] * sg * (1.f + x * (1.f - sg)));
    else out[r * s_out + c] = (T)(x * sg);
  }
}

// ------------------------------------------------------------------ host-side dispatch
static int norm_blocks(int64_t rows) {
  int64_t b = (rows + kNormWaves - 1) / kNormWaves;
  if (b > kMaxNormBlocks) b = kMaxNormBlocks;
  return b < 1 ? 1 : (int)b;
}

template <typename T> static int vec_for(int dim, const void* a, const void* b2, const void* c) {
  const int v = sizeof(T) == 2 ? 8 : 4;
  const uintptr_t bits = (uintptr_t)a | (uintptr_t)b2 | (uintptr_t)c;
  return (dim % v == 0 && (bits & 15) == 0) ? v : 1;
}

static int check_dim(int dim, int vec, const char* who) {
  if (dim <= 0) return set_error(HSTU_EINVAL, "%s: dim must be positive", who);
  if (dim > 64 * vec * (vec == 8 ? 2 : (vec == 4 ? 4 : 8))) return set_error(HSTU_EUNSUPPORTED, "%s: dim %d exceeds the %d supported with this alignment", who, dim, 64 * vec * (vec == 8 ? 2 : (vec == 4 ? 4 : 8)));
  return HSTU_OK;
}

template <typename T>
static int ln_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows, int dim,
                  float eps, hipStream_t st) {
  const int v = vec_for<T>(dim, x, y, w) == 1 ? 1 : vec_for<T>(dim, b, nullptr, nullptr);
  if (int e = check_dim(dim, v, "layer_norm_fwd")) return e;
  const int nb = norm_blocks(rows);
  if (v == 1)
    hipLaunchKernelGGL((layer_norm_fwd_kernel<T, 1>), dim3(nb), dim3(kNormThreads), 0, st, (const T*)x, (const T*)w, (const T*)b, (T*)y, mean, rstd, rows, dim, eps);
  else
    hipLaunchKernelGGL((layer_norm_fwd_kernel<T, (sizeof(T) == 2 ? 8 : 4)>), dim3(nb), dim3(kNormThreads), 0, st, (const T*)x, (const T*)w, (const T*)b, (T*)y, mean, rstd, rows, dim, eps);
  return check_launch("layer_norm_fwd");
}

template <typename T>
static int ln_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                  float* dweight, float* dbias, float* partial, int64_t rows, int dim, hipStream_t st) {
  int v = vec_for<T>(dim, dy, x, dx);
  if (v != 1) v = vec_for<T>(dim, w, nullptr, nullptr);
  if (int e = check_dim(dim, v, "layer_norm_bwd")) return e;
  const int nb = norm_blocks(rows);
  const int nch = (dim + 64 * v - 1) / (64 * v);
  const size_t lds = (size_t)kNormWaves * nch * 64 * v * sizeof(float);
  if (v == 1)
    hipLaunchKernelGGL((layer_norm_bwd_kernel<T, 1>), dim3(nb), dim3(kNormThreads), lds, st, (const T*)dy, (const T*)x, (const T*)w, mean, rstd, (T*)dx, partial, rows, dim);
  else
    hipLaunchKernelGGL((layer_norm_bwd_kernel<T, (sizeof(T) == 2 ? 8 : 4)>), dim3(nb), dim3(kNormThreads), lds, st, (const T*)dy, (const T*)x, (const T*)w, mean, rstd, (T*)dx, partial, rows, dim);
  if (int e = check_launch("layer_norm_bwd")) return e;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 * dim), dim3(256), 0, st, partial, nb, dim, dweight, dbias);
  return check_launch("layer_norm_bwd(reduce)");
}

template <typename T>
static int nm_fwd(const void* attn, const void* u, const void* w, const void* b, void* y, float* mean, float* rstd,
                  int64_t rows, int heads, int hdim, float eps, int gn, int concat, DropCtx dc, hipStream_t st) {
  const int dim = heads * hdim;
  int v = vec_for<T>(dim, attn, u, y);
  if (v != 1 && !gn) v = vec_for<T>(dim, w, b, nullptr);
  if (v != 1 && gn && hdim % v) v = 1;
  if (int e = check_dim(dim, v, "norm_mul_fwd")) return e;
  if (gn && heads > 16) return set_error(HSTU_EUNSUPPORTED, "norm_mul: group norm supports at most 16 heads");
  const int nb = norm_blocks(rows);
#define NM_LAUNCH(V, G) hipLaunchKernelGGL((norm_mul_fwd_kernel<T, V, G>), dim3(nb), dim3(kNormThreads), 0, st, (const T*)attn, (const T*)u, (const T*)w, (const T*)b, (T*)y, mean, rstd, rows, heads, hdim, eps, concat, dc)
  constexpr int VV = sizeof(T) == 2 ? 8 : 4;
  if (gn && gn_fast_ok(hdim, v))
    hipLaunchKernelGGL((norm_mul_fwd_gn_kernel<T, VV>), dim3(nb), dim3(kNormThreads), 0, st, (const T*)attn, (const T*)u, (const T*)w, (const T*)b, (T*)y, mean, rstd, rows, heads, hdim, eps, concat, dc);
  else if (v == 1) { if (gn) NM_LAUNCH(1, true); else NM_LAUNCH(1, false); }
  else { if (gn) NM_LAUNCH(VV, true); else NM_LAUNCH(VV, false); }
#undef NM_LAUNCH
  return check_launch("norm_mul_fwd");
}

template <typename T>
static int nm_bwd(const void* dy, const void* attn, const void* u, const void* w, const void* b, const float* mean,
                  const float* rstd, void* dattn, void* du, float* dweight, float* dbias, float* partial, int64_t rows,
                  int heads, int hdim, int gn, int concat, DropCtx dc, hipStream_t st) {
  const int dim = heads * hdim;
  int v = vec_for<T>(dim, attn, u, dy);
  if (v != 1) v = vec_for<T>(dim, dattn, du, nullptr);
  if (v != 1 && !gn) v = vec_for<T>(dim, w, b, nullptr);
  if (v != 1 && gn && hdim % v) v = 1;
  if (int e = check_dim(dim, v, "norm_mul_bwd")) return e;
  if (gn && heads > 16) return set_error(HSTU_EUNSUPPORTED, "norm_mul: group norm supports at most 16 heads");
  const int nb = norm_blocks(rows);
  const int nch = (dim + 64 * v - 1) / (64 * v);
  const size_t lds = gn ? kNormWaves * 32 * sizeof(float) : (size_t)kNormWaves * nch * 64 * v * sizeof(float);
#define NM_LAUNCH(V, G) hipLaunchKernelGGL((norm_mul_bwd_kernel<T, V, G>), dim3(nb), dim3(kNormThreads), lds, st, (const T*)dy, (const T*)attn, (const T*)u, (const T*)w, (const T*)b, mean, rstd, (T*)dattn, (T*)du, partial, rows, heads, hdim, concat, dc)
  constexpr int VV = sizeof(T) == 2 ? 8 : 4;
  if (gn && gn_fast_ok(hdim, v))
    hipLaunchKernelGGL((norm_mul_bwd_gn_kernel<T, VV>), dim3(nb), dim3(kNormThreads), lds, st, (const T*)dy, (const T*)attn, (const T*)u, (const T*)w, (const T*)b, mean, rstd, (T*)dattn, (T*)du, partial, rows, heads, hdim, concat, dc);
  else if (v == 1) { if (gn) NM_LAUNCH(1, true); else NM_LAUNCH(1, false); }
  else { if (gn) NM_LAUNCH(VV, true); else NM_LAUNCH(VV, false); }
#undef NM_LAUNCH
  if (int e = check_launch("norm_mul_bwd")) return e;
  const int width = gn ? heads : dim;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2 * width), dim3(256), 0, st, partial, nb, width, dweight, dbias);
  return check_launch("norm_mul_bwd(reduce)");
}

// 16 bytes per lane: a row of the column slice is `cols / VEC` pieces; used when the slice start, the row strides
// and the width are all 16-byte multiples (the u slice of the fused uvqk buffer is)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void silu_vec_kernel(const T* dout, const T* in, T* out, int64_t rows, int cols,
                                                       int64_t s_dout, int64_t s_in, int64_t s_out) {
  constexpr int VEC = 16 / sizeof(T);
  const int ppr = cols / VEC;                               // pieces per row
  const int64_t n = rows * ppr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ppr;
    const int c = (int)(i - r * ppr) * VEC;
    RowVec<T, VEC> x, g, o;
    load_vec<T, VEC>(x, in + r * s_in + c, true);
    if (BWD) load_vec<T, VEC>(g, dout + r * s_dout + c, true);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float sg = 1.0f / (1.0f + __expf(-x.v[k]));
      o.v[k] = BWD ? g.v[k] * sg * (1.f + x.v[k] * (1.f - sg)) : x.v[k] * sg;
    }
    store_vec<T, VEC>(o, out + r * s_out + c);
  }
}

template <typename T, bool BWD>
static int silu_launch(const void* dout, const void* in, void* out, int64_t rows, int cols, int64_t s0, int64_t s1,
                       int64_t s2, hipStream_t st) {
  const int64_t n = rows * cols;
  if (n == 0) return HSTU_OK;
  constexpr int VEC = 16 / sizeof(T);
  const bool vec_ok = cols % VEC == 0 && s1 % VEC == 0 && s2 % VEC == 0 && (!BWD || s0 % VEC == 0) &&
                      (((uintptr_t)in | (uintptr_t)out | (BWD ? (uintptr_t)dout : 0)) & 15) == 0;
  if (vec_ok) {
    const int64_t pieces = n / VEC;
    int blocks = (int)((pieces + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL((silu_vec_kernel<T, BWD>), dim3(blocks), dim3(256), 0, st, (const T*)dout, (const T*)in, (T*)out, rows, cols, s0, s1, s2);
    return check_launch("silu");
  }
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((silu_kernel<T, BWD>), dim3(blocks), dim3(256), 0, st, (const T*)dout, (const T*)in, (T*)out, rows, cols, s0, s1, s2);
  return check_launch("silu");
}

// ------------------------------------------------------------------ row L2 normalisation (output postprocessor)
// y = x / max(||x||_2, eps)   (modules/postprocessors.py:55-69: seq / linalg.norm(seq).clamp(min=1e-6)), one wave per
// row, fp32 math.  Backward: with n = ||x||: n > eps -> dx = (g - y <y, g>) / n;  n <= eps (the clamp is active and
// has zero gradient) -> dx = g / eps.
template <typename T, int VEC, bool BWD>
__global__ __launch_bounds__(kNormThreads) void l2_norm_kernel(const T* x, const T* g, T* out, int64_t rows, int dim, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = (dim + 64 * VEC - 1) / (64 * VEC);
  for (int64_t row = (int64_t)blockIdx.x * kNormWaves + wave; row < rows; row += (int64_t)gridDim.x * kNormWaves) {
    RowVec<T, VEC> xv[max_chunks<VEC>()], gv[max_chunks<VEC>()];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      const bool ok = k < nch && c < dim;
      load_vec<T, VEC>(xv[k], x + row * dim + c, ok);
      if (BWD) load_vec<T, VEC>(gv[k], g + row * dim + c, ok);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { ss += xv[k].v[i] * xv[k].v[i]; if (BWD) dot += xv[k].v[i] * gv[k].v[i]; }
    }
    const float n = sqrtf(wave_sum(ss));
    const bool clamped = n <= eps;
    const float inv = 1.0f / fmaxf(n, eps);
    float coef = 0.f;
    if (BWD) coef = clamped ? 0.f : wave_sum(dot) * inv * inv * inv;   // <y, g> / n * (1/n) applied to x
#pragma unroll
    for (int k = 0; k < max_chunks<VEC>(); ++k) {
      const int c = (k * 64 + lane) * VEC;
      if (k < nch && c < dim) {
        RowVec<T, VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o.v[i] = BWD ? gv[k].v[i] * inv - xv[k].v[i] * coef : xv[k].v[i] * inv;
        store_vec<T, VEC>(o, out + row * dim + c);
      }
    }
  }
}

template <typename T, bool BWD>
static int l2_launch(const void* x, const void* g, void* out, int64_t rows, int dim, float eps, hipStream_t st) {
  if (rows == 0) return HSTU_OK;
  const int v = vec_for<T>(dim, x, BWD ? g : x, out);
  if (int e = check_dim(dim, v, "l2_norm")) return e;
  const int nb = norm_blocks(rows);
  constexpr int VV = sizeof(T) == 2 ? 8 : 4;
  if (v == 1) hipLaunchKernelGGL((l2_norm_kernel<T, 1, BWD>), dim3(nb), dim3(kNormThreads), 0, st, (const T*)x, (const T*)g, (T*)out, rows, dim, eps);
  else hipLaunchKernelGGL((l2_norm_kernel<T, VV, BWD>), dim3(nb), dim3(kNormThreads), 0, st, (const T*)x, (const T*)g, (T*)out, rows, dim, eps);
  return check_launch("l2_norm");
}

}  // namespace hstu

using namespace hstu;

#define DISPATCH_DTYPE(dtype, CALL_BF16, CALL_F16, CALL_F32)                          \
  switch (dtype) {                                                                    \
    case HSTU_DTYPE_BF16: return CALL_BF16;                                           \
    case HSTU_DTYPE_F16: return CALL_F16;                                             \
    case HSTU_DTYPE_F32: return CALL_F32;                                             \
    default: return set_error(HSTU_EINVAL, "dtype must be bf16, fp16 or fp32");       \
  }

extern "C" {

size_t hstu_norm_bwd_workspace_bytes(int64_t rows, int32_t dim) {
  (void)rows;
  return (size_t)kMaxNormBlocks * 2 * (size_t)dim * sizeof(float);
}

int hstu_layer_norm_fwd(const void* x, const void* weight, const void* bias, void* y, float* mean, float* rstd,
                        int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows == 0) return HSTU_OK;
  if (!x || !weight || !bias || !y) return set_error(HSTU_EINVAL, "layer_norm_fwd: NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, ln_fwd<bf16_t>(x, weight, bias, y, mean, rstd, rows, dim, eps, st),
                 ln_fwd<f16_t>(x, weight, bias, y, mean, rstd, rows, dim, eps, st),
                 ln_fwd<float>(x, weight, bias, y, mean, rstd, rows, dim, eps, st));
}

int hstu_layer_norm_bwd(const void* dy, const void* x, const void* weight, const float* mean, const float* rstd,
                        void* dx, float* dweight, float* dbias, float* partial_ws, int64_t rows, int32_t dim, int dtype,
                        void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!dweight || !dbias) return set_error(HSTU_EINVAL, "layer_norm_bwd: dweight/dbias are required");
  if (rows == 0) {
    (void)hipMemsetAsync(dweight, 0, dim * sizeof(float), st);
    (void)hipMemsetAsync(dbias, 0, dim * sizeof(float), st);
    return HSTU_OK;
  }
  if (!dy || !x || !weight || !mean || !rstd || !dx || !partial_ws) return set_error(HSTU_EINVAL, "layer_norm_bwd: NULL tensor");
  DISPATCH_DTYPE(dtype, ln_bwd<bf16_t>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, st),
                 ln_bwd<f16_t>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, st),
                 ln_bwd<float>(dy, x, weight, mean, rstd, dx, dweight, dbias, partial_ws, rows, dim, st));
}

static int drop_ratio_ok(float r, const char* who) {
  if (!(r >= 0.f && r < 1.f)) return set_error(HSTU_EINVAL, "%s: dropout_ratio must be in [0, 1) (got %g)", who, (double)r);
  return HSTU_OK;
}

int hstu_norm_mul_dropout_fwd(const void* attn, const void* u, const void* weight, const void* bias, void* y, float* mean,
                              float* rstd, int64_t rows, int32_t heads, int32_t head_dim, float eps, int group_norm,
                              int concat_ux, float dropout_ratio, uint64_t seed, int dtype, void* stream) {
  if (int e = drop_ratio_ok(dropout_ratio, "norm_mul_dropout_fwd")) return e;
  if (rows == 0) return HSTU_OK;
  if (!attn || !u || !weight || !bias || !y) return set_error(HSTU_EINVAL, "norm_mul_fwd: NULL tensor");
  hipStream_t st = (hipStream_t)stream;
  const DropCtx dc = make_drop_ctx(dropout_ratio, seed);
  DISPATCH_DTYPE(dtype, nm_fwd<bf16_t>(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux, dc, st),
                 nm_fwd<f16_t>(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux, dc, st),
                 nm_fwd<float>(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux, dc, st));
}

int hstu_norm_mul_fwd(const void* attn, const void* u, const void* weight, const void* bias, void* y, float* mean,
                      float* rstd, int64_t rows, int32_t heads, int32_t head_dim, float eps, int group_norm,
                      int concat_ux, int dtype, void* stream) {
  return hstu_norm_mul_dropout_fwd(attn, u, weight, bias, y, mean, rstd, rows, heads, head_dim, eps, group_norm, concat_ux,
                                   0.f, 0, dtype, stream);
}

int hstu_norm_mul_bwd(const void* dy, const void* attn, const void* u, const void* weight, const void* bias,
                      const float* mean, const float* rstd, void* dattn, void* du, float* dweight, float* dbias,
                      float* partial_ws, int64_t rows, int32_t heads, int32_t head_dim, int group_norm, int concat_ux,
                      int dtype, void* stream) {
  return hstu_norm_mul_dropout_bwd(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads,
                                   head_dim, group_norm, concat_ux, 0.f, 0, dtype, stream);
}

int hstu_norm_mul_dropout_bwd(const void* dy, const void* attn, const void* u, const void* weight, const void* bias,
                              const float* mean, const float* rstd, void* dattn, void* du, float* dweight, float* dbias,
                              float* partial_ws, int64_t rows, int32_t heads, int32_t head_dim, int group_norm,
                              int concat_ux, float dropout_ratio, uint64_t seed, int dtype, void* stream) {
  if (int e = drop_ratio_ok(dropout_ratio, "norm_mul_dropout_bwd")) return e;
  const DropCtx dc = make_drop_ctx(dropout_ratio, seed);
  hipStream_t st = (hipStream_t)stream;
  const int width = group_norm ? heads : heads * head_dim;
  if (!dweight || !dbias) return set_error(HSTU_EINVAL, "norm_mul_bwd: dweight/dbias are required");
  if (rows == 0) {
    (void)hipMemsetAsync(dweight, 0, width * sizeof(float), st);
    (void)hipMemsetAsync(dbias, 0, width * sizeof(float), st);
    return HSTU_OK;
  }
  if (!dy || !attn || !u || !weight || !bias || !mean || !rstd || !dattn || !du || !partial_ws)
    return set_error(HSTU_EINVAL, "norm_mul_bwd: NULL tensor");
  DISPATCH_DTYPE(dtype, nm_bwd<bf16_t>(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads, head_dim, group_norm, concat_ux, dc, st),
                 nm_bwd<f16_t>(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads, head_dim, group_norm, concat_ux, dc, st),
                 nm_bwd<float>(dy, attn, u, weight, bias, mean, rstd, dattn, du, dweight, dbias, partial_ws, rows, heads, head_dim, group_norm, concat_ux, dc, st));
}

int hstu_silu_fwd(const void* in, void* out, int64_t rows, int32_t cols, int64_t in_row_stride, int64_t out_row_stride,
                  int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, (silu_launch<bf16_t, false>(nullptr, in, out, rows, cols, 0, in_row_stride, out_row_stride, st)),
                 (silu_launch<f16_t, false>(nullptr, in, out, rows, cols, 0, in_row_stride, out_row_stride, st)),
                 (silu_launch<float, false>(nullptr, in, out, rows, cols, 0, in_row_stride, out_row_stride, st)));
}

int hstu_silu_bwd(const void* dout, const void* in, void* din, int64_t rows, int32_t cols, int64_t dout_row_stride,
                  int64_t in_row_stride, int64_t din_row_stride, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, (silu_launch<bf16_t, true>(dout, in, din, rows, cols, dout_row_stride, in_row_stride, din_row_stride, st)),
                 (silu_launch<f16_t, true>(dout, in, din, rows, cols, dout_row_stride, in_row_stride, din_row_stride, st)),
                 (silu_launch<float, true>(dout, in, din, rows, cols, dout_row_stride, in_row_stride, din_row_stride, st)));
}


int hstu_l2_norm_fwd(const void* x, void* y, int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows > 0 && (!x || !y)) return set_error(HSTU_EINVAL, "hstu_l2_norm_fwd: x and y must be non-NULL");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, (l2_launch<bf16_t, false>(x, nullptr, y, rows, dim, eps, st)),
                 (l2_launch<f16_t, false>(x, nullptr, y, rows, dim, eps, st)),
                 (l2_launch<float, false>(x, nullptr, y, rows, dim, eps, st)));
}

int hstu_l2_norm_bwd(const void* dy, const void* x, void* dx, int64_t rows, int32_t dim, float eps, int dtype, void* stream) {
  if (rows > 0 && (!x || !dy || !dx)) return set_error(HSTU_EINVAL, "hstu_l2_norm_bwd: dy, x and dx must be non-NULL");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DTYPE(dtype, (l2_launch<bf16_t, true>(x, dy, dx, rows, dim, eps, st)),
                 (l2_launch<f16_t, true>(x, dy, dx, rows, dim, eps, st)),
                 (l2_launch<float, true>(x, dy, dx, rows, dim, eps, st)));
}

}  // extern "C"
