// y = LayerNorm(x) . W + b for the STU layer's UVQK projection as ONE kernel (gfx950).
//   hstu_compute_uqvk (ops/hstu_compute.py:62-89: layer_norm, then addmm) and the first half of
//   _HSTUPreprocessAndAttentionFunction (ops/triton/triton_hstu_preprocess_and_attention.py:37-120);
//   kernels replaced: ops/triton/triton_layer_norm.py:77-309 + ops/triton/triton_addmm.py:185-340.
//
// The contraction length is the layer's embedding dim, K = 512: a row of x is 1 KiB, so a WAVE keeps 32 complete rows
// in registers (128 VGPRs, in the layout of the MFMA operand) -- x is read from memory exactly once (as whole 128-byte
// lines, turned into fragments through 4 KiB of wave-private LDS), the row statistics and the affine are applied in
// registers, and normed_x never exists in memory unless asked for (the NORMED instantiation: backward's recompute).  A workgroup
// (8 waves = 256 rows, two waves per SIMD) then walks the weight: W is streamed, 32 output columns at a time (32 x 512,
// K-contiguous rows = 32 KiB), through a ring of LDS tiles filled by LDS-DMA; every wave multiplies every tile by its
// own rows.  Traffic per 256 x 32 outputs: one 32 KiB tile from L2 (16 B/clk/CU with the MFMA pipe saturated; W is 2 MiB
// and stays in every XCD's L2) and one ds_read_b128 per MFMA -- against A AND B panels for a square tile.
//
// The product is formed transposed (W supplies the MFMA's A operand, x its B operand): a lane then holds, for ONE row
// of y, four runs of 4 consecutive columns; v_permlane32_swap pairs the half-waves' runs into 16-byte pieces, which cross
// the wave's LDS staging and leave as 16 rows x 64 contiguous bytes per (non-temporal) store.  The bias enters as the
// accumulator's start value (the first MFMA of a chain takes its C operand from the bias registers: no add, no zeroing);
// accumulators ping-pong: tile t is packed under the MFMAs of tile t + 1 and stored under those of tile t + 2.  One
// barrier per tile in the MIDDLE of its chain, behind a counted vmcnt (lnl step): the next tile has landed, the previous
// tile's slot is requested again, and the wave's six memory instructions of a step are spread over the chain.
//
// 421-445 us at 204,800 x 512 -> 2048 (layer norm + hipBLASLt: 82 + 577..607); what bounds it, and the arrangements
// that lost (rows of the next block requested early: LNL_PRELOAD; two workgroups per CU: hstu_ln_linear2.cuh):
// docs/EXPERIMENTS.md R4.8.
//
// Work is cut into (row block, column tile) units, dealt to the persistent workgroups (one per CU) as CONTIGUOUS runs
// of equal length: 204,800 rows = 800 blocks would leave a quarter of the chip idle in the last of 3.1 rounds; 51,200
// units are 200 per CU.  Tiles are walked cyclically, so a run may start in the middle of a block; the two workgroups
// that share a block both normalise it (and write identical mean / rstd).
//
// Row statistics: sum and sum of squares by v_dot2c_f32 on the PACKED 16-bit pairs (products exact, fp32
// accumulation; half a VALU instruction per element and statistic); when that form of the variance would cancel
// (E[x^2] > 64 var in any row of the wave) the wave recomputes the centred sum of squares the long way.
#pragma once
#include <type_traits>

#include "hstu_common.cuh"
#include "capi_internal.h"

namespace hstu {

#ifndef LNL_STAGES
#define LNL_STAGES 3       // LDS tiles in the ring: LNL_STAGES - 1 tiles of W in flight while one is multiplied
#endif
#ifndef LNL_AHEAD
#define LNL_AHEAD 4        // W fragments requested ahead of the MFMA that takes them
#endif
#ifndef LNL_DRAIN_STORES
#define LNL_DRAIN_STORES 0 // 1: every step waits for the previous step's stores of y as well (vmcnt(0))
#endif
#ifndef LNL_PRELOAD
#define LNL_PRELOAD 0      // request the next block's rows of x under the last tile of a block (a step instantiation of its own):
                           // measured 439-451 against 423-428 us -- memory instructions in the chain hold up the MFMAs behind them
#endif
#ifndef LNL_NT_STORES
#define LNL_NT_STORES 1    // y leaves with the non-temporal hint (443 -> 421 us)
#endif
#ifndef LNL_X_LINES
#define LNL_X_LINES 1      // rows of x are requested as whole 128-byte lines (8 lanes per row) and turned into MFMA fragments through
#endif                     // the wave's LDS staging; 0: fragment-shaped loads (32 rows x 32 bytes per instruction)
#ifndef LNL_ABLATE
#define LNL_ABLATE 0       // experiments: 1 no stores of y, 2 no MFMA, 4 no W DMA after the prefill, 8 skip the LN arithmetic,
                           // 16 no packing of finished tiles, 32 no bias reads, 64 no loads of x, 128 no barrier / vmcnt wait,
                           // 256 every W request twice, 512 all stores of y land in the first MiB (cache-resident)
#endif

constexpr int kLnlK = 512;
constexpr int kLnlWaves = 8;
constexpr int kLnlThreads = 64 * kLnlWaves;
constexpr int kLnlBlockRows = 32 * kLnlWaves;
constexpr int kLnlKS = kLnlK / 16;                   // MFMAs per 32 x 32 output tile
constexpr int kLnlTileBytes = 32 * kLnlK * 2;        // 32 KiB
constexpr int kLnlRingBytes = LNL_STAGES * kLnlTileBytes;
constexpr int kLnlMaxN = 4096;
constexpr int kLnlStageBytes = 4096;              // per wave: 32 rows x 128 bytes of x on their way in, 32 x 64 of y on their way out

struct LnLinearArgs {
  const void* x; const void* ln_w; const void* ln_b; const void* w; const void* bias;
  void* y; void* normed; float* mean; float* rstd;
  int64_t rows, ldx, ldy, ldn;      // leading dimensions in elements
  int64_t units;                    // row blocks x column tiles
  int n, n_tiles;
  float eps;
};

// tools-only timeline (-DLNL_TRACE): lane 0 of every wave of workgroup 100 stamps (tag, s_memtime) pairs into the buffer
// passed as `normed` (no normalised rows are written in such a build)
#ifdef LNL_TRACE
#define LNL_MARK(tag)                                                                                        \
  do {                                                                                                       \
    if (g.normed && blockIdx.x == 100 && (threadIdx.x & 63) == 0 && lnl_ti < 126) {                                      \
      ((unsigned long long*)g.normed)[(threadIdx.x >> 6) * 256 + 2 * lnl_ti] = (unsigned long long)(tag);    \
      ((unsigned long long*)g.normed)[(threadIdx.x >> 6) * 256 + 2 * lnl_ti + 1] = __builtin_readcyclecounter(); \
      ++lnl_ti;                                                                                              \
    }                                                                                                        \
  } while (0)
#else
#define LNL_MARK(tag)
#endif

static inline int lnl_smem_bytes(int n) { return kLnlRingBytes + 2 * kLnlK * 4 + n * 4 + kLnlWaves * kLnlStageBytes; }   // ring, LN tables, bias, the waves' staging

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
HSTU_DEV void lnl_dma16(uint32_t off, const char* base, uint32_t lds_base) {
  // wave-uniform by construction; said explicitly, or a control-flow join can leave them in vector registers
  const uint64_t b = (uint64_t)(uintptr_t)base;
  base = (const char*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                                   (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b));   // (the builtin returns int)
  lds_base = __builtin_amdgcn_readfirstlane(lds_base);
  // s_nop 4: a v_readfirstlane result needs 5 wait states before a memory instruction may take it as its address, and the
  // hazard pass does not look inside an asm statement
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop

HSTU_DEV void lnl_gstore(void* p, u32x4 v) {
  if (LNL_NT_STORES) gstore16_nt(p, v);
  else gstore16(p, v);
}

template <typename T> struct LnlDot;
template <> struct LnlDot<bf16_t> {
  typedef bf16_t v2 __attribute__((ext_vector_type(2)));
  static HSTU_DEV float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
  static constexpr uint32_t kOnes = 0x3f803f80u;
  static HSTU_DEV float lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
  static HSTU_DEV float hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
};
template <> struct LnlDot<f16_t> {
  typedef f16_t v2 __attribute__((ext_vector_type(2)));
  static HSTU_DEV float dot(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
  }
  static constexpr uint32_t kOnes = 0x3c003c00u;
  static HSTU_DEV float lo(uint32_t w) { return (float)__builtin_bit_cast(v2, w)[0]; }
  static HSTU_DEV float hi(uint32_t w) { return (float)__builtin_bit_cast(v2, w)[1]; }
};

// The 32 rows of the calling wave, normalised, as MFMA operand fragments: xf[ks] = elements [16 ks + 8 h, +8) of row
// `lane & 31` (h = lane >> 5).
// where lane (m, h) finds its pieces of row m of block `blk` (rows past the end: the last row, never stored)
HSTU_DEV const char* lnl_row_ptr(const LnLinearArgs& g, int64_t row0, int lane) {
  const int64_t row = row0 + (lane & 31);
  return (const char*)g.x + ((row < g.rows ? row : g.rows - 1) * g.ldx + 8 * (lane >> 5)) * 2;
}

// TAB16: the LayerNorm tables sit in LDS in the I/O type (the two-workgroup arrangement's 80 KiB budget) instead of fp32
// LN = false: the rows as they are (the plain K = 512 product of hstu_linear_k512: no statistics, no affine)
template <typename T, bool NORMED, bool TAB16 = false, bool LN = true>
HSTU_DEV void lnl_load_rows(const LnLinearArgs& g, int64_t row0, const float* gam, const float* bet, int lane,
                            u32x4 (&xf)[kLnlKS], bool preloaded, char* stage, bool x_lines = LNL_X_LINES) {
  using DT = LnlDot<T>;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool ok = row < g.rows;
  const char* xp = lnl_row_ptr(g, row0, lane);
  if (preloaded && !x_lines) {
    // the raw rows are already on their way: requested fragment by fragment under the last tile of the block before
  } else if (LNL_ABLATE & 64) {
#pragma unroll
    for (int ks = 0; ks < kLnlKS; ++ks) asm volatile("" : "=v"(xf[ks]));
  } else if (x_lines) {
    // instruction 4 c + q fetches bytes [128 c, 128 c + 128) of the rows 8 q .. 8 q + 7 of the wave: lane L = (row 8 q + (L >> 3),
    // piece L & 7) -- eight whole lines per instruction instead of 32 quarter lines.  Each 128-byte column chunk (4 MFMA steps)
    // then crosses the wave's staging: piece p of row r sits at slot p ^ ((r >> 1) & 7) of the row (conflict-free for the
    // 8-lane groups of the stores and the 16-lane groups of the loads); lane (m, h) picks up pieces 2 j + h of row m.
    const int pr = lane >> 3, pp = lane & 7;
    const char* rp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = row0 + 8 * q + pr;
      rp[q] = (const char*)g.x + (r < g.rows ? r : g.rows - 1) * g.ldx * 2 + 16 * pp;
    }
    if (!preloaded) {      // (else: requested under the last tile of the block before, the same way)
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) xf[4 * c + q] = gload16(rp[q] + 128 * c);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 8 * q + pr;
        *LDS_PTR(u32x4, stage + r * 128 + ((pp ^ ((r >> 1) & 7)) << 4)) = xf[4 * c + q];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[4 * c + j] = *LDS_PTR(const u32x4, stage + m * 128 + (((2 * j + h) ^ ((m >> 1) & 7)) << 4));
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < kLnlKS; ++ks) xf[ks] = gload16(xp + ks * 32);
  }
  if (!LN || (LNL_ABLATE & 8)) return;
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int ks = 0; ks < kLnlKS; ++ks)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s = DT::dot(xf[ks][j], DT::kOnes, s);
      q = DT::dot(xf[ks][j], xf[ks][j], q);
    }
  s += __shfl_xor(s, 32);
  q += __shfl_xor(q, 32);
  const float mean = s * (1.0f / kLnlK);
  const float ex2 = q * (1.0f / kLnlK);
  float var = ex2 - mean * mean;
  if (__builtin_amdgcn_ballot_w64(!(ex2 <= 64.f * var)) != 0) {   // rows far from zero mean (or NaN): the centred form
    float c = 0.f;
#pragma unroll
    for (int ks = 0; ks < kLnlKS; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d0 = DT::lo(xf[ks][j]) - mean, d1 = DT::hi(xf[ks][j]) - mean;
        c += d0 * d0;
        c += d1 * d1;
        if (j == 3 && ks % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      }
    c += __shfl_xor(c, 32);
    var = c * (1.0f / kLnlK);
#pragma unroll
    for (int ks = 0; ks < kLnlKS; ++ks) asm volatile("" : "+v"(xf[ks]));   // or the unpacked values are kept (spilled) for the pass below
  }
  const float rstd = 1.0f / sqrtf(var + g.eps);
  const float nmr = -mean * rstd;
  if (ok && h == 0) {
    if (g.mean) g.mean[row] = mean;
    if (g.rstd) g.rstd[row] = rstd;
  }
  const float* gp = gam + 8 * h;
  const float* bp = bet + 8 * h;
#pragma unroll
  for (int ks = 0; ks < kLnlKS; ++ks) {
    f32x4 g0, g1, b0, b1;
    if constexpr (TAB16) {
      const u32x4 gw = *LDS_PTR(const u32x4, (const T*)gam + 8 * h + 16 * ks), bw = *LDS_PTR(const u32x4, (const T*)bet + 8 * h + 16 * ks);
      g0 = f32x4{DT::lo(gw[0]), DT::hi(gw[0]), DT::lo(gw[1]), DT::hi(gw[1])};
      g1 = f32x4{DT::lo(gw[2]), DT::hi(gw[2]), DT::lo(gw[3]), DT::hi(gw[3])};
      b0 = f32x4{DT::lo(bw[0]), DT::hi(bw[0]), DT::lo(bw[1]), DT::hi(bw[1])};
      b1 = f32x4{DT::lo(bw[2]), DT::hi(bw[2]), DT::lo(bw[3]), DT::hi(bw[3])};
    } else {
      g0 = *LDS_PTR(const f32x4, gp + 16 * ks), g1 = *LDS_PTR(const f32x4, gp + 16 * ks + 4);
      b0 = *LDS_PTR(const f32x4, bp + 16 * ks), b1 = *LDS_PTR(const f32x4, bp + 16 * ks + 4);
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ga = j < 2 ? g0[2 * j] : g1[2 * j - 4], gb = j < 2 ? g0[2 * j + 1] : g1[2 * j - 3];
      const float ba = j < 2 ? b0[2 * j] : b1[2 * j - 4], bb = j < 2 ? b0[2 * j + 1] : b1[2 * j - 3];
      const float t0 = __builtin_fmaf(DT::lo(xf[ks][j]), rstd, nmr), t1 = __builtin_fmaf(DT::hi(xf[ks][j]), rstd, nmr);
      o[j] = Elem<T>::pk2(__builtin_fmaf(t0, ga, ba), __builtin_fmaf(t1, gb, bb));
    }
    xf[ks] = o;
    asm volatile("" : "+v"(xf[ks]));     // computed HERE: left to itself the compiler sinks the arithmetic towards the MFMAs and keeps the table values (spilled) until then
    if (ks % 2 == 1) __builtin_amdgcn_sched_barrier(0);   // or the scheduler hoists all 128 table reads (512 registers)
  }
#ifndef LNL_TRACE
  if (NORMED) {
    if (x_lines) {
      // the way the rows came in, backwards: each 128-byte column chunk crosses the staging and leaves as whole lines
      const int pr = lane >> 3, pp = lane & 7;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *LDS_PTR(u32x4, stage + m * 128 + (((2 * j + h) ^ ((m >> 1) & 7)) << 4)) = xf[4 * c + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 8 * q + pr;
          const u32x4 v = *LDS_PTR(const u32x4, stage + r * 128 + ((pp ^ ((r >> 1) & 7)) << 4));
          const int64_t gr = row0 + r;
          if (gr < g.rows) lnl_gstore((char*)g.normed + (gr * g.ldn + 64 * c + 8 * pp) * 2, v);
        }
      }
    } else if (ok) {
      char* np = (char*)g.normed + (row * g.ldn + 8 * h) * 2;
#pragma unroll
      for (int ks = 0; ks < kLnlKS; ++ks) gstore16(np + ks * 32, xf[ks]);
    }
  }
#endif
}

// accumulator -> rows of y.  Register r of lane (m, h) is column (r & 3) + 8 (r >> 2) + 4 h of row m of the 32 x 32 tile:
// v_permlane32_swap pairs the half-waves' 4-column runs into 16-byte pieces (lanes 0..31: columns [0, 8) and [16, 24), lanes
// 32..63: [8, 16) and [24, 32)).  Stored like that a store instruction touches 32 rows with 32 bytes each -- twice the write
// requests of whole 64-byte row segments, and the stores then hold up the weight requests queued behind them (timeline in
// docs/EXPERIMENTS.md R4.8) -- so the pieces take a turn through a wave-private 2 KiB of LDS and come back row-major: lane L
// gets piece L & 3 of rows L >> 2 (.lo) and 16 + (L >> 2) (.hi); an instruction then writes 16 rows x 64 contiguous bytes.
// Piece c of row m sits at slot c ^ ((m >> 1) & 3) of its 64-byte row: conflict-free for the ds_write_b128 lane groups (8
// consecutive rows, one piece; stores bank by 32 banks = a 128-byte window) and the ds_read_b128 groups (4 rows x 4 pieces; 64 banks).  LDS operations of one wave execute in order.
struct LnlPacked { u32x4 lo, hi; };
#ifndef LNL_STAGE_SHIFT
#define LNL_STAGE_SHIFT 1    // (2: the first version's swizzle, two-way conflicts on the stores into the staging)
#endif
HSTU_DEV uint32_t lnl_stage_off(int m, int c) { return (uint32_t)(m * 64 + ((c ^ ((m >> LNL_STAGE_SHIFT) & 3)) << 4)); }
template <typename T>
HSTU_DEV LnlPacked lnl_pack_tile(const f32x16& acc, char* stage, int lane) {
  typedef Elem<T> E;
  uint32_t p[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j][0] = E::pk2(acc[4 * j], acc[4 * j + 1]);
    p[j][1] = E::pk2(acc[4 * j + 2], acc[4 * j + 3]);
  }
#pragma unroll
  for (int jp = 0; jp < 4; jp += 2)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const auto sw = __builtin_amdgcn_permlane32_swap(p[jp][e], p[jp + 1][e], false, false);   // lanes 32..63 of the first <-> 0..31 of the second
      p[jp][e] = sw[0];
      p[jp + 1][e] = sw[1];
    }
  const int m = lane & 31, h = lane >> 5;
  *LDS_PTR(u32x4, stage + lnl_stage_off(m, h)) = u32x4{p[0][0], p[0][1], p[1][0], p[1][1]};
  *LDS_PTR(u32x4, stage + lnl_stage_off(m, 2 + h)) = u32x4{p[2][0], p[2][1], p[3][0], p[3][1]};
  LnlPacked k;
  k.lo = *LDS_PTR(const u32x4, stage + lnl_stage_off(lane >> 2, lane & 3));
  k.hi = *LDS_PTR(const u32x4, stage + lnl_stage_off(16 + (lane >> 2), lane & 3));
  return k;
}

// NORMED: the instantiation that also writes the normalised rows (its extra addressing would cost the other one 8 spilled registers)
template <typename T, bool NORMED, bool LN = true>
__global__ __launch_bounds__(kLnlThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void hstu_ln_linear_fwd_kernel(const LnLinearArgs g) {
  static_assert(LN || !NORMED, "normalised rows only exist with the LayerNorm");
  extern __shared__ __attribute__((aligned(1024))) char lnl_smem[];
  typedef Elem<T> E;
  typedef typename E::Frag Frag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  char* ring = lnl_smem;
  float* gam = (float*)(lnl_smem + kLnlRingBytes);
  float* bet = gam + kLnlK;
  float* bia = bet + kLnlK;
  if constexpr (LN) {
    for (int i = tid; i < kLnlK; i += kLnlThreads) {
      gam[i] = (float)((const T*)g.ln_w)[i];
      bet[i] = (float)((const T*)g.ln_b)[i];
    }
  }
  for (int i = tid; i < g.n; i += kLnlThreads) bia[i] = g.bias ? (float)((const T*)g.bias)[i] : 0.f;

  const int64_t u0 = g.units * blockIdx.x / gridDim.x, u1 = g.units * (blockIdx.x + 1) / gridDim.x;
  const int nsteps = (int)(u1 - u0);
  if (nsteps <= 0) return;
  int64_t blk = u0 / g.n_tiles;
  int tile = (int)(u0 - blk * g.n_tiles);

  // the wave's share of a tile's DMA: rows 4 wave .. 4 wave + 3 of the 32, one 1 KiB instruction each; the lane that
  // fills physical unit `lane` of row r fetches logical unit lane ^ swz(r) (hstu_common.cuh: tile_off)
  uint32_t uo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) uo[j] = (uint32_t)((4 * wave + j) * (kLnlK * 2) + ((lane ^ swz<64>(4 * wave + j)) << 4));
  const uint32_t ring0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ring);
  int it = tile, islot = 0, issued = 0;
  auto issue = [&]() {
    const char* base = (const char*)g.w + (int64_t)it * kLnlTileBytes;
#pragma unroll
    for (int j = 0; j < 4; ++j) lnl_dma16(uo[j], base, ring0 + islot * kLnlTileBytes + (4 * wave + j) * 1024);
    it = it + 1 == g.n_tiles ? 0 : it + 1;
    islot = islot + 1 == LNL_STAGES ? 0 : islot + 1;
    ++issued;
  };
  for (int i = 0; i < LNL_STAGES - 1 && i < nsteps; ++i) issue();
  __syncthreads();   // gam / bet / bia

  u32x4 xf[kLnlKS];
  f32x16 acc[2];
  Frag wf[LNL_AHEAD];
#ifdef LNL_TRACE
  int lnl_ti = 0;
#endif
  LNL_MARK(1);
  int cslot = 0;
  char* stage = (char*)(bia + g.n) + wave * kLnlStageBytes;
  // fragment ks of a ring tile = 16-byte unit 2 ks + h of row m, at slot (2 ks + h) ^ swz(m) of the row: the swizzle touches
  // the low four bits of the unit only, so 8 addresses (ks & 7) + an immediate 256 (ks >> 3) cover the 32 fragments
  uint32_t fa[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) fa[kk] = ring0 + (uint32_t)tile_off<64>(m, 2 * kk + h);
  auto frag_at = [&](int slot, int ks) {
    Frag f;
    const u32x4 x = *LDS_PTR(const u32x4, (uintptr_t)(fa[ks & 7] + slot * kLnlTileBytes + (ks >> 3) * 256));
    f.v = __builtin_bit_cast(typename E::vec8, x);
    return f;
  };
  auto bias_into = [&](f32x16& a, int t) {
    const float* bt = bia + t * 32 + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b4 = *LDS_PTR(const f32x4, bt + 8 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * j + e] = b4[e];
    }
  };
  auto next_slot = [&](int sl) { return sl + 1 == LNL_STAGES ? 0 : sl + 1; };

  // One column tile = a chain of 32 MFMAs on the ring tile `cslot`; the stream of fragment reads runs LNL_AHEAD MFMAs ahead
  // and straight on into the next tile.  In the middle of the chain: every wave's pieces of the NEXT tile have landed
  // (counted vmcnt + barrier; they were requested a whole step ago) and the tile before this one is dead, so its slot is
  // requested again.  The wave's six memory instructions of a step are spread over the chain (a burst of them blocks the
  // issuing waves -- and the MFMAs queued behind -- until the address unit has taken them: 1,300 of 4,150 cycles per step in
  // the first version's timeline): the two stores of the tile packed a step ago in the first half, the four requests one
  // every four MFMAs of the second; the previous tile's accumulator (`oth`) is packed there too and takes the next bias.
  LnlPacked pk;
  char* pk_dst = nullptr;
  bool ok_lo = false, ok_hi = false;     // this lane's two rows of the block exist
  int n_stores = 0;                      // store instructions of a tile that the wave really issues (none for a half without rows)
  // (PRE: the variant for the last tile of a block that requests the next block's rows; a template parameter, not a flag --
  // 32 conditional loads would cut every chain into basic blocks)
  auto step = [&](f32x16& cur, f32x16& oth, bool have_prev, bool have_packed, char* row_dst, const char* xnext, auto pre) {
    constexpr bool PRE = decltype(pre)::value;
    const int nslot = next_slot(cslot);
    const bool do_issue = issued < nsteps && !((LNL_ABLATE & 4) && issued >= LNL_STAGES);
    const char* wbase = (const char*)g.w + (int64_t)it * kLnlTileBytes;
    const uint32_t dst0 = ring0 + islot * kLnlTileBytes + 4 * wave * 1024;
    LNL_MARK(10);
#pragma unroll
    for (int ks = 0; ks < kLnlKS; ++ks) {
      Frag xb;
      xb.v = __builtin_bit_cast(typename E::vec8, xf[ks]);
      if (!(LNL_ABLATE & 2)) cur = E::mma(wf[ks % LNL_AHEAD], xb, cur);
      else cur[ks & 15] += (float)wf[ks % LNL_AHEAD].v[0] + (float)xb.v[0];
      if (ks == 3 && have_packed && ok_lo) lnl_gstore(pk_dst, pk.lo);
      if (ks == 9 && have_packed && ok_hi) lnl_gstore(pk_dst + 16 * g.ldy * 2, pk.hi);
      if (ks == kLnlKS / 2 - 1) {
        LNL_MARK(11);
        // no lgkmcnt wait: the reads in flight are of THIS tile; the slot requested below was read by MFMAs that have issued.
        // vmcnt counts in issue order: behind the next tile's four requests there are only this step's two stores
        // ... and, in the last tile of a block, the 15 fragments of the next block's rows requested so far
        const int behind = (have_packed && !LNL_DRAIN_STORES ? n_stores : 0) + (PRE ? kLnlKS / 2 - 1 : 0);
        if (LNL_ABLATE & 128) {
        } else if (behind == 17) asm volatile("s_waitcnt vmcnt(17)\n\ts_barrier" ::: "memory");
        else if (behind == 16) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        else if (behind == 15) asm volatile("s_waitcnt vmcnt(15)\n\ts_barrier" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        LNL_MARK(12);
      }
      // fragment ks of this block's rows has had its last MFMA: its registers take the next block's row pieces now, and the
      // load latency passes under the rest of the chain instead of in front of the next block
      if constexpr (PRE) xf[ks] = LNL_X_LINES ? gload16(xnext + (ks & 3) * (8 * g.ldx * 2) + 128 * (ks >> 2)) : gload16(xnext + ks * 32);
      if (ks >= kLnlKS / 2 && ks % 4 == 0 && do_issue) {
        const int j = (ks - kLnlKS / 2) / 4;
        lnl_dma16(uo[j], wbase, dst0 + j * 1024);
        if (LNL_ABLATE & 256) lnl_dma16(uo[j], wbase, dst0 + j * 1024);
      }
      wf[ks % LNL_AHEAD] = ks + LNL_AHEAD < kLnlKS ? frag_at(cslot, ks + LNL_AHEAD) : frag_at(nslot, ks + LNL_AHEAD - kLnlKS);
      if (ks == kLnlKS / 2 + 1 && have_prev && !(LNL_ABLATE & 16)) {
        pk = lnl_pack_tile<T>(oth, stage, lane);
        pk_dst = row_dst + (tile - 1) * 64;
        if (LNL_ABLATE & 512) pk_dst = (char*)g.y + ((pk_dst - (char*)g.y) & 0xFFFF0) ;
      }
      if (ks == kLnlKS / 2 + 6 && !(LNL_ABLATE & 32)) bias_into(oth, tile + 1 == g.n_tiles ? 0 : tile + 1);
#ifdef LNL_TRACE_FINE
      if (ks % 4 == 3 && ks != kLnlKS / 2 - 1) LNL_MARK(20 + ks / 4);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_issue) {
      it = it + 1 == g.n_tiles ? 0 : it + 1;
      islot = islot + 1 == LNL_STAGES ? 0 : islot + 1;
      ++issued;
    }
    cslot = nslot;
    ++tile;
  };
  auto store_packed = [&]() {
    if (ok_lo) lnl_gstore(pk_dst, pk.lo);
    if (ok_hi) lnl_gstore(pk_dst + 16 * g.ldy * 2, pk.hi);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();      // the first tile is in the ring
  int left = nsteps;
  bool preloaded = false;
  while (left > 0) {
    LNL_MARK(2);
    lnl_load_rows<T, NORMED, false, LN>(g, blk * kLnlBlockRows + wave * 32, gam, bet, lane, xf, preloaded, stage);
    LNL_MARK(3);
    const int64_t row = blk * kLnlBlockRows + wave * 32 + (lane >> 2);      // the lane's rows at store time: row, row + 16
    const bool live = !(LNL_ABLATE & 1) || g.eps == 12345.f;
    ok_lo = row < g.rows && live;
    ok_hi = row + 16 < g.rows && live;
    char* row_dst = (char*)g.y + row * g.ldy * 2 + 16 * (lane & 3);
    int nt = g.n_tiles - tile;
    if (nt > left) nt = left;
    left -= nt;
    // another block follows in this run: its rows are requested during this block's last tile
    const char* xn = nullptr;
    if (LNL_PRELOAD && left > 0) {
      if (!LNL_X_LINES) xn = lnl_row_ptr(g, (blk + 1) * kLnlBlockRows + wave * 32, lane);
      else if ((blk + 2) * kLnlBlockRows <= g.rows)      // (a last, partial block is requested when its turn comes, rows clamped)
        xn = (const char*)g.x + ((blk + 1) * kLnlBlockRows + wave * 32 + (lane >> 3)) * g.ldx * 2 + 16 * (lane & 7);
    }
    preloaded = xn != nullptr;
    bias_into(acc[0], tile);
#pragma unroll
    for (int i = 0; i < LNL_AHEAD; ++i) wf[i] = frag_at(cslot, i);
    n_stores = (__builtin_amdgcn_ballot_w64(ok_lo) != 0) + (__builtin_amdgcn_ballot_w64(ok_hi) != 0);
    const std::false_type reg{};
    const std::true_type pre{};
    const int n_reg = xn ? nt - 1 : nt;     // tiles on the regular step; the block's last one requests the next block's rows
    int k = 0;
    if (n_reg > 0) {
      step(acc[0], acc[1], false, false, row_dst, nullptr, reg);
      for (k = 1; k + 1 < n_reg; k += 2) {
        step(acc[1], acc[0], true, k > 1, row_dst, nullptr, reg);
        step(acc[0], acc[1], true, true, row_dst, nullptr, reg);
      }
      if (k < n_reg) {
        step(acc[1], acc[0], true, k > 1, row_dst, nullptr, reg);
        ++k;
      }
    }
    if (xn) {
      if (k & 1) step(acc[1], acc[0], k > 0, k > 1, row_dst, xn, pre);
      else step(acc[0], acc[1], k > 0, k > 1, row_dst, xn, pre);
      ++k;
    }
    if (nt > 1) store_packed();
    pk = (k & 1) ? lnl_pack_tile<T>(acc[0], stage, lane) : lnl_pack_tile<T>(acc[1], stage, lane);
    pk_dst = row_dst + (tile - 1) * 64;
    store_packed();
    if (tile == g.n_tiles) { tile = 0; ++blk; }
  }
}

template <typename T>
static int launch_ln_linear(const LnLinearArgs& g, hipStream_t st) {
  const int smem = lnl_smem_bytes(g.n);
#ifdef LNL_TRACE
  auto kern = hstu_ln_linear_fwd_kernel<T, false>;
#else
  auto kern = !g.ln_w ? hstu_ln_linear_fwd_kernel<T, false, false> : g.normed ? hstu_ln_linear_fwd_kernel<T, true> : hstu_ln_linear_fwd_kernel<T, false>;
#endif
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "ln_linear_fwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  const int n_cu = cu_count();
  const int grid = (int)(g.units < n_cu ? g.units : n_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kLnlThreads), smem, st, g);
  return check_launch("ln_linear_fwd");
}

}  // namespace hstu
