// HSTU attention backward, "wide" schedule for the metric shape (head dim 128, 16-bit I/O, <= 7 tiles of 32 rows):
// FOUR waves per workgroup, one per SIMD, each with the whole 512-entry register file of its SIMD.
//
// Same math, tiles, masks and LDS tile formats as the folded schedule (hstu_attn_bwd_fold.cuh); what changes is who owns
// what and when memory moves.  The folded kernel runs 8 waves of 256 registers: a wave owns ONE key tile (128 accumulator
// registers), reads both operands of every S / dP MFMA from LDS, balances the causal triangle by letting wave 7 - t open a
// second partial sum of key tile t that is handed over through LDS in a tail, and its phases -- DMA issue, pairs, dQ GEMM,
// parks, copy-outs -- run one after the other on every wave (profiles/r04_wide_v1_trace.txt: the waves of a workgroup sit in
// VMEM issue for a quarter of a problem's time).  Here wave w owns key tiles w (`lo`) AND 7 - w (`hi`) for the whole problem:
//   * dK / dV of both tiles live in the accumulator file for the whole problem: all 256 AGPRs, OWNED by this file's asm
//     statements (literal register names: hipcc's allocator never sees 256 values that must survive every branch of the
//     step loop -- left to it, it spills hundreds of registers at the control-flow joins); no second partial sum, no
//     hand-over tail; finished tiles go from the registers straight to memory (v_permlane32_swap pairs the half-waves'
//     8-byte pieces into 16-byte stores, rows past the sequence end are cut off by the buffer descriptor);
//   * the K / V row fragments of the low tile -- the B operands of S = Q K^T and dP = dO V^T for 22 of a problem's 28
//     pairs -- live in 64 VGPRs, read from LDS once per problem; the high tile's come from LDS per pair;
//   * the triangle is balanced by the same two-ended walk (step k: query tiles a = nt-1-k and b = k), now inside one
//     wave: per step a wave runs (a, lo) and one of (a, hi) | (b, lo) -- two pairs per step on every wave at 7 tiles;
//   * dQ of the step's two query tiles: wave w owns the 32 features [32 w, +32) of both tiles, a 32x32x16 GEMM over the
//     published dS' tiles and the transposed K tiles (K stays in LDS for this); its stores are issued in the NEXT step;
//   * memory moves UNDER the arithmetic: the Q / dO tiles of step k+1 (into a second stage set: the V slots of tiles 0..3,
//     dead once their fragments are in registers), the next problem's K / V tiles of the slots this problem has finished
//     with, and the dQ rows of step k-1 are issued one instruction at a time between the MFMAs of the step's first pair.
// Requires what the folded kernel requires (no contextual rows, no bias, head dims equal to the instantiated ones,
// max_seq_len <= 224, 1e-20 < |alpha| < 1e6); HSTU_BWD_WIDE=0 sends the shape back to the folded kernel.
#pragma once
#include <type_traits>
#include "hstu_attn_bwd_fold.cuh"

#ifndef WIDE_KV_REGS
#define WIDE_KV_REGS 1     // bit 1: K/V fragments of the HIGH tile in registers as well (the low tile's always are)
#endif
#ifndef WIDE_SDP_AHEAD
#define WIDE_SDP_AHEAD 4   // LDS fragments requested ahead of their MFMA in the S / dP and dV / dK streams
#endif
#ifndef WIDE_DQ_AHEAD
#define WIDE_DQ_AHEAD 2    // key tiles whose fragments are in flight ahead of the dQ GEMM's MFMAs
#endif
#ifndef WIDE_PERSIST
#define WIDE_PERSIST 2     // 0: one workgroup per (user, head); 1: one per CU walks the problems; 2: and requests the next problem's tiles
#endif
#ifndef WIDE_SIDE_IN_STREAM
#define WIDE_SIDE_IN_STREAM 1   // DMA / store instructions of a step between the MFMAs of its first pair (0: in a block before it)
#endif
#ifndef WIDE_ABLATE
#define WIDE_ABLATE 0      // timing experiments only (WRONG results): 1 no dQ stores, 2 no dk/dv stores, 4 no stage DMA after
#endif                     // step 0, 16 no K/V DMA, 32 no dQ GEMM, 64 no pairs, 128 / 256 problems alias the first 256 / 32

namespace hstu {

constexpr int kWideWaves = 4;
constexpr int kWideThreads = 256;

template <typename T, int D>
struct WideCfg {
  using B = BwdCfg<T, D, D>;
  static constexpr int DSB = 32 * 64;       // [32 keys][32 q] 16-bit tile (fold_ds_off)
  static constexpr int kMaxTiles = 7;
  static constexpr int kDsSlots = 8;        // side A key tile t -> slot t, side B key tile t -> slot 7 - t
  static constexpr int smem_bytes() { return (kMaxTiles + 2) * B::PAIR + kDsSlots * DSB; }
};


// ---- MFMAs through inline asm ----------------------------------------------------------------------------------------------
// hipcc selects ONE form for every MFMA builtin of a function; with 256 accumulator registers it is the AGPR form, and then the
// transient S / dP / dQ accumulators would have to be AGPRs as well (288 > 256).  So every MFMA of this kernel names its
// instruction itself: the transient chains accumulate in VGPRs ("+v"), dK / dV in literal AGPRs:
//     dK of the low tile a[0:63], dV of the low tile a[64:127], dK of the high tile a[128:191], dV of the high tile a[192:255]
// (block d of 32 features = 16 registers).  Every statement that touches them lists ALL AGPRs as clobbered: that makes the
// kernel descriptor allocate them and keeps the compiler's own values (spills) out of them.
// hipcc does not model the instruction inside an asm statement (guide 5.7): every statement pads the VALU-write -> MFMA-read
// hazard itself (s_nop 1: the compiler may have written an operand -- a copy, a reload -- in the instruction before),
// wide_mfma_drain() the MFMA-write -> VALU-read one after a chain's last MFMA (8-pass XDL: 11 wait states);
// tools/lint_asm_mfma.py checks the built code (no compiler instruction inside a chain touches its accumulators, no
// compiler v_accvgpr_* at all, no spills).
#define WIDE_A10(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
#define WIDE_ALL_AGPRS                                                                                                                     \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", WIDE_A10(1), WIDE_A10(2), WIDE_A10(3), WIDE_A10(4), WIDE_A10(5), WIDE_A10(6),     \
      WIDE_A10(7), WIDE_A10(8), WIDE_A10(9), WIDE_A10(10), WIDE_A10(11), WIDE_A10(12), WIDE_A10(13), WIDE_A10(14), WIDE_A10(15), WIDE_A10(16),   \
      WIDE_A10(17), WIDE_A10(18), WIDE_A10(19), WIDE_A10(20), WIDE_A10(21), WIDE_A10(22), WIDE_A10(23), WIDE_A10(24), "a250", "a251", "a252",   \
      "a253", "a254", "a255"
constexpr int kWideAccDkLo = 0, kWideAccDvLo = 64, kWideAccDkHi = 128, kWideAccDvHi = 192;

template <int I, int N, class F>
HSTU_DEV void wide_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wide_static_for<I + 1, N>(f);
  }
}

template <typename T> struct WideMma;
#define WIDE_MMA_SPEC(TYPE, NAME)                                                                                                          \
  template <> struct WideMma<TYPE> {                                                                                                       \
    /* c (VGPRs) += a b */                                                                                                                 \
    static HSTU_DEV void run(const Elem<TYPE>::Frag& a, const Elem<TYPE>::Frag& b, f32x16& c) {                                            \
      const u32x4 av = __builtin_bit_cast(u32x4, a.v), bv = __builtin_bit_cast(u32x4, b.v);                                                \
      asm volatile("s_nop 1\n\t" NAME " %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));                                                    \
    }                                                                                                                                      \
    static HSTU_DEV void run(const Elem<TYPE>::Frag& a, const u32x4& bv, f32x16& c) {                                                      \
      const u32x4 av = __builtin_bit_cast(u32x4, a.v);                                                                                     \
      asm volatile("s_nop 1\n\t" NAME " %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));                                                    \
    }                                                                                                                                      \
    /* a[BASE : BASE + 15] += a b */                                                                                                       \
    template <int BASE>                                                                                                                    \
    static HSTU_DEV void acc(const Elem<TYPE>::Frag& a, const Elem<TYPE>::Frag& b) {                                                       \
      const u32x4 av = __builtin_bit_cast(u32x4, a.v), bv = __builtin_bit_cast(u32x4, b.v);                                                \
      asm volatile("s_nop 1\n\t" NAME " a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(av), "v"(bv), "i"(BASE), "i"(BASE + 15) : WIDE_ALL_AGPRS);  \
    }                                                                                                                                      \
  };
WIDE_MMA_SPEC(bf16_t, "v_mfma_f32_32x32x16_bf16")
WIDE_MMA_SPEC(f16_t, "v_mfma_f32_32x32x16_f16")
#undef WIDE_MMA_SPEC
// after the last MFMA of a chain, before the first VALU instruction that reads its accumulators
HSTU_DEV void wide_mfma_drain(f32x16& c0, f32x16& c1) { asm volatile("s_nop 15" : "+v"(c0), "+v"(c1)); }
HSTU_DEV void wide_mfma_drain(f32x16& c0) { asm volatile("s_nop 15" : "+v"(c0)); }

// all 256 accumulator registers = 0
template <int B>
HSTU_DEV void wide_acc_zero8() {
  asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0\n\t"
               "v_accvgpr_write_b32 a[%c4], 0\n\tv_accvgpr_write_b32 a[%c5], 0\n\tv_accvgpr_write_b32 a[%c6], 0\n\tv_accvgpr_write_b32 a[%c7], 0"
               ::"i"(B), "i"(B + 1), "i"(B + 2), "i"(B + 3), "i"(B + 4), "i"(B + 5), "i"(B + 6), "i"(B + 7) : WIDE_ALL_AGPRS);
}
template <int I>
HSTU_DEV void wide_acc_zero_from() {
  if constexpr (I < 256) {
    wide_acc_zero8<I>();
    wide_acc_zero_from<I + 8>();
  }
}
HSTU_DEV void wide_acc_zero() { wide_acc_zero_from<0>(); }
// one accumulator block (16 registers from BASE) into VGPRs.  The caller has drained the MFMA pipe (wide_acc_drain).
template <int B>
HSTU_DEV void wide_acc_read8(float& x0, float& x1, float& x2, float& x3, float& x4, float& x5, float& x6, float& x7) {
  asm volatile("v_accvgpr_read_b32 %0, a[%c8]\n\tv_accvgpr_read_b32 %1, a[%c9]\n\tv_accvgpr_read_b32 %2, a[%c10]\n\tv_accvgpr_read_b32 %3, a[%c11]\n\t"
               "v_accvgpr_read_b32 %4, a[%c12]\n\tv_accvgpr_read_b32 %5, a[%c13]\n\tv_accvgpr_read_b32 %6, a[%c14]\n\tv_accvgpr_read_b32 %7, a[%c15]"
               : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7)
               : "i"(B), "i"(B + 1), "i"(B + 2), "i"(B + 3), "i"(B + 4), "i"(B + 5), "i"(B + 6), "i"(B + 7)
               : WIDE_ALL_AGPRS);
}
template <int BASE>
HSTU_DEV f32x16 wide_acc_read() {
  float x[16];
  wide_acc_read8<BASE>(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
  wide_acc_read8<BASE + 8>(x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15]);
  f32x16 r;
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = x[i];
  return r;
}
HSTU_DEV void wide_acc_drain() { asm volatile("s_nop 15" ::: "memory"); }

// LDS-DMA of one [32][D] tile by NW waves (fold_tile_dma with the wave count as a parameter)
template <typename T, int D, int NW>
HSTU_DEV void wide_tile_dma(char* tile, const char* base, int64_t row_stride_bytes, int row0, int len, int wave, int lane, bool fast) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int NCH = 32 * UPR / 64;   // 1 KiB chunks per tile
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tile);
#pragma unroll
  for (int c0 = 0; c0 < NCH; c0 += NW) {
    const int c = c0 + wave;
    const int pidx = c * 64 + lane;
    const int row = pidx / UPR, slot = pidx % UPR;
    const int unit = slot ^ swz<UPR>(row);
    const int grow = min(row0 + row, len - 1);
    if (fast) dma16_saddr(__umul24((uint32_t)grow, (uint32_t)row_stride_bytes) + unit * 16, base, lds0 + c * 1024);
    else dma16_asm(base + (int64_t)grow * row_stride_bytes + unit * 16, lds0 + c * 1024);
  }
}

template <typename T, int D, int NT>
HSTU_DEV void wide_copy_out(const char* __restrict__ tile, char* gtile, int64_t row_stride_bytes, int rows_valid, int tid) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
#pragma unroll
  for (int u0 = 0; u0 < 32 * UPR; u0 += NT) {
    const int u = u0 + tid;
    const int row = u / UPR, unit = u % UPR;
    const u32x4 v = *LDS_PTR(const u32x4, tile + tile_off<UPR>(row, unit));
    if (row < rows_valid && (!(WIDE_ABLATE & 2) || row_stride_bytes == -12345)) gstore16(gtile + row * row_stride_bytes + unit * 16, v);
  }
}

// the K (or V) row fragments of one key tile as the B operands of the S (dP) stream: fragment m covers the features
// [64 hf + 8 m, +8) of key row n32 -- what fold_pair reads from LDS for every pair.  Kept as raw 16-byte words: their only
// consumers are the "a"-constrained operands of WideMmaBA, i.e. they live in the accumulator file.
template <typename T, int D>
HSTU_DEV void wide_load_kv_frags(u32x4 (&f)[D / 16], const char* tile, int lane) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  const int n32 = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int m = 0; m < D / 16; ++m) f[m] = *LDS_PTR(const u32x4, tile + tile_off<UPR>(n32, (hf * (D / 2) + m * 8) >> 3));
}

// One (query tile i0, key tile k0) pair: S, dP, P', dS', dV +=, dK +=, publish dS' (fold_pair_x without the bias; KVREG:
// the B operands of the S / dP stream come from the register fragments kf / vf instead of the LDS tiles Kw / Vw).
struct WideNoSide {
  HSTU_DEV void operator()(int) const {}
};

// SIDE: a callable invoked with the slot numbers 0..31, once behind each of the pair's 32 MFMAs (the step's DMA and store
// instructions ride in the MFMA streams: WideSide below).
// ACC: first accumulator register of the key tile's dK (its dV follows 64 registers later).
template <typename T, int D, bool KVREG, int ACC, typename SIDE = WideNoSide>
HSTU_DEV void wide_pair(const HstuAttnParams& p, const MaskCtx& mc, const char* __restrict__ Kw, const char* __restrict__ Vw,
                        const u32x4 (&kf)[D / 16], const u32x4 (&vf)[D / 16],
                        const char* __restrict__ Qs, const char* __restrict__ dOs, char* __restrict__ myds, int i0, int k0,
                        int lane, int dmvm, const SIDE& side HSTU_TRACE_ARG) {
  using C = BwdCfg<T, D, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int n32 = lane & 31, hf = lane >> 5;
  const int len = mc.len;
  const int key = k0 + n32;
  const bool key_ok = key < len;
  f32x16 s, dp;
  Frag pb[2], dsb[2];
  int mode;   // wave-uniform: 0 = no mask needed, 1 = plain causal, 2 = general mask algebra (see fold_pair_x)
  if (mc.simple) mode = (k0 < i0 && i0 + 32 <= len) ? 0 : 1;
  else mode = (i0 + 32 <= len && mc.pair_fully_valid(i0, 32, k0, 32)) ? 0 : 2;
  const int key_id = mc.id_of(key);
  const int key_bits = key_ok ? -1 : 0;
  {
    int km = -1;
    if (mode == 1) km = ((k0 == i0) ? dmvm : -1) & ((i0 + 32 > len) ? (dmvm >> 16) : -1);
    if (mode == 2) {
      km = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = i0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
        km |= (mc.keep_bits_noctx(qi, key, key_id) & key_bits & 1) << r;
      }
    }
    const unsigned nk = ~(unsigned)km;
    const unsigned neg = __builtin_bit_cast(unsigned, p.alpha < 0.f ? 1e30f : -1e30f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = ((int)(nk << (31 - r))) >> 31;        // all ones iff masked
      s[r] = __builtin_bit_cast(float, (unsigned)m & neg);
      dp[r] = 0.f;
    }
  }
  // S and dP: one stream of 16 MFMAs alternating between the two accumulators
  {
    constexpr int NM = 2 * C::KGQ, AHEAD = WIDE_SDP_AHEAD;
    Frag fa[AHEAD + 1], fb[AHEAD + 1];
    auto load_item = [&](int m, Frag& a, Frag& bb) {
      const int e0 = hf * (D / 2) + (m >> 1) * 8;
      a = lds_row_frag<T, C::UPR_K>((m & 1) ? dOs : Qs, n32, e0);
      if constexpr (!KVREG) bb = lds_row_frag<T, C::UPR_K>((m & 1) ? Vw : Kw, n32, e0);
    };
#pragma unroll
    for (int m = 0; m < AHEAD; ++m) load_item(m, fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (m + AHEAD < NM) load_item(m + AHEAD, fa[(m + AHEAD) % (AHEAD + 1)], fb[(m + AHEAD) % (AHEAD + 1)]);
      if constexpr (KVREG) {
        if (m & 1) WideMma<T>::run(fa[m % (AHEAD + 1)], vf[m >> 1], dp);
        else WideMma<T>::run(fa[m % (AHEAD + 1)], kf[m >> 1], s);
      } else {
        if (m & 1) WideMma<T>::run(fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)], dp);
        else WideMma<T>::run(fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)], s);
      }
      side(m);
      __builtin_amdgcn_sched_barrier(0);
    }
    wide_mfma_drain(s, dp);
  }
  HSTU_MARK(11);
  auto elem = [&](const int h8) {
    float pv[8], dsv[8];
    const f32x2 a2 = {p.alpha, p.alpha};
    const f32x2 c2 = {-1.44269504088896340736f * p.alpha, -1.44269504088896340736f * p.alpha};
    const f32x2 one2 = {1.f, 1.f};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const int r = 8 * h8 + j;
      const f32x2 sv = {s[r], s[r + 1]}, dpv = {dp[r], dp[r + 1]};
      const f32x2 x = sv * a2, t = sv * c2;
      const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      const f32x2 dn = e + one2;
      const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
      const f32x2 pr = x * sg;
      const f32x2 w = x * (one2 - sg) + one2;     // 1 + x (1 - sg)
      const f32x2 dsr = dpv * sg * w;
      pv[j] = pr[0]; pv[j + 1] = pr[1];
      dsv[j] = dsr[0]; dsv[j + 1] = dsr[1];
    }
    pb[h8] = E::pack8(pv);
    dsb[h8] = E::pack8(dsv);
  };
  elem(0);
  elem(1);
  HSTU_MARK(12);
  // dV_w^T[dv][key] += dO_i^T[dv][q] P'[q][key]   and   dK_w^T[d][key] += Q_i^T[d][q] dS'[q][key]: 16 MFMAs into the tile's
  // accumulator registers (dV | dK) x k half x d block, A fragments (transposed reads of the dO / Q tile) AHEAD items ahead
  {
    constexpr int NM = 2 * 2 * C::DBQ, AHEAD = WIDE_SDP_AHEAD;
    Frag fa[AHEAD + 1];
    auto load_item = [&](int m, Frag& a) {
      const int ks = (m >> 1) & 1, d = m >> 2;
      a = lds_col_frag<T, C::UPR_K>((m & 1) ? Qs : dOs, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, lane);
    };
#pragma unroll
    for (int m = 0; m < AHEAD; ++m) load_item(m, fa[m % (AHEAD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
    wide_static_for<0, NM>([&](auto ic) {
      constexpr int m = decltype(ic)::value;
      if (m + AHEAD < NM) load_item(m + AHEAD, fa[(m + AHEAD) % (AHEAD + 1)]);
      constexpr int ks = (m >> 1) & 1, d = m >> 2;
      if constexpr ((m & 1) != 0) WideMma<T>::template acc<ACC + 16 * d>(fa[m % (AHEAD + 1)], dsb[ks]);
      else WideMma<T>::template acc<ACC + 64 + 16 * d>(fa[m % (AHEAD + 1)], pb[ks]);
      side(16 + m);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // publish dS' as [key = n32][q]: this lane holds q = 4 hf + 8 rq + (0..3) = chunk hf + 2 rq
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const u32x4 w = __builtin_bit_cast(u32x4, dsb[rq >> 1].v);
    u32x2 v2 = {w[2 * (rq & 1)], w[2 * (rq & 1) + 1]};
    *LDS_PTR(u32x2, myds + fold_ds_off(n32, hf + 2 * rq)) = v2;
  }
  HSTU_MARK(14);
}

// dQ^T[32 features of block db][32 q] of ONE query tile: sum over key tiles t = 0..N-1 of K_t^T dS'_t^T with the 32x32x16
// MFMA, two per key tile (keys 0..15, 16..31).  K_t = K/V slot t; the dS' tile of key tile t is slot t (side A) or 7 - t
// (side B): ds0 + t * ds_step.  N is a compile-time count: straight-line code, the fragments of tile t + WIDE_DQ_AHEAD
// requested before the MFMAs of tile t.
template <typename T, int D, int N>
HSTU_DEV f32x16 wide_dq_chain(const char* __restrict__ kv, const char* __restrict__ ds0, int ds_step, int db, int lane) {
  using C = BwdCfg<T, D, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int hf = lane >> 5, i16 = lane & 15, g1 = (lane >> 4) & 1;
  const int ra = 8 * hf, rb = 16 + 8 * hf;
  // B fragments: lane supplies the address of key row r + (i16 >> 2), query chunk 4 g1 + (i16 & 3) (see dsbuf_col_frag)
  const int chunk = 4 * g1 + (i16 & 3), rr = i16 >> 2;
  const int o00 = fold_ds_off(ra + rr, chunk), o01 = fold_ds_off(ra + 4 + rr, chunk);
  const int o10 = fold_ds_off(rb + rr, chunk), o11 = fold_ds_off(rb + 4 + rr, chunk);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // (the asm MFMAs pin the order: the fragments of tile t + AH are requested in front of the MFMAs of tile t)
  constexpr int AH = WIDE_DQ_AHEAD < N ? WIDE_DQ_AHEAD : N;
  Frag a0[AH + 1], a1[AH + 1], b0[AH + 1], b1[AH + 1];
  auto load_tile = [&](int t, int sl) {
    const char* Kt = kv + t * C::PAIR;
    const char* ds = ds0 + t * ds_step;
    a0[sl] = lds_col_frag<T, C::UPR_K>(Kt, ra, ra + 4, 32 * db, lane);      // K^T[d][key]
    a1[sl] = lds_col_frag<T, C::UPR_K>(Kt, rb, rb + 4, 32 * db, lane);
    b0[sl] = tr_frag16<T>(ds, o00, o01);                                     // dS'^T[key][q]
    b1[sl] = tr_frag16<T>(ds, o10, o11);
  };
#pragma unroll
  for (int t = 0; t < AH; ++t) load_tile(t, t % (AH + 1));
#pragma unroll
  for (int t = 0; t < N; ++t) {
    if (t + AH < N) load_tile(t + AH, (t + AH) % (AH + 1));
    WideMma<T>::run(a0[t % (AH + 1)], b0[t % (AH + 1)], acc);
    WideMma<T>::run(a1[t % (AH + 1)], b1[t % (AH + 1)], acc);
  }
  wide_mfma_drain(acc);
  return acc;
}

template <typename T, int D>
HSTU_DEV f32x16 wide_dq_side(const char* __restrict__ kv, const char* __restrict__ ds0, int ds_step, int n, int db, int lane) {
  switch (n) {   // wave-uniform
    case 1: return wide_dq_chain<T, D, 1>(kv, ds0, ds_step, db, lane);
    case 2: return wide_dq_chain<T, D, 2>(kv, ds0, ds_step, db, lane);
    case 3: return wide_dq_chain<T, D, 3>(kv, ds0, ds_step, db, lane);
    case 4: return wide_dq_chain<T, D, 4>(kv, ds0, ds_step, db, lane);
    case 5: return wide_dq_chain<T, D, 5>(kv, ds0, ds_step, db, lane);
    case 6: return wide_dq_chain<T, D, 6>(kv, ds0, ds_step, db, lane);
    default: return wide_dq_chain<T, D, 7>(kv, ds0, ds_step, db, lane);
  }
}

// ---- register tiles -> memory ------------------------------------------------------------------------------------------
// C layout of a transposed 32 x 32 block (dQ^T, dK^T, dV^T): column n32 = row of the output (query or key), register r =
// feature (r & 3) + 8 (r >> 2) + 4 hf of the block.  A lane's four 4-feature groups are paired up with v_permlane32_swap
// (lanes n32 and n32 + 32 hold the same output row) into two runs of 8 consecutive features: lanes 0..31 end up with the
// features [0, 8) and [16, 24) of the block, lanes 32..63 with [8, 16) and [24, 32) -- two 16-byte pieces per lane.
template <typename T>
HSTU_DEV void wide_pack_block(const f32x16& acc, float scale, u32x4& lo16, u32x4& hi16) {
  using E = Elem<T>;
  uint32_t g[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g[j][0] = E::pk2(acc[4 * j] * scale, acc[4 * j + 1] * scale);
    g[j][1] = E::pk2(acc[4 * j + 2] * scale, acc[4 * j + 3] * scale);
  }
#pragma unroll
  for (int jp = 0; jp < 4; jp += 2)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const auto sw = __builtin_amdgcn_permlane32_swap(g[jp][h], g[jp + 1][h], false, false);   // lanes 32..63 of the first <-> 0..31 of the second
      g[jp][h] = sw[0];
      g[jp + 1][h] = sw[1];
    }
  lo16 = u32x4{g[0][0], g[0][1], g[1][0], g[1][1]};
  hi16 = u32x4{g[2][0], g[2][1], g[3][0], g[3][1]};
}

// Rows of one 32-row tile of a (rows, heads, D) output as a raw buffer: base = the tile's first row at this head, the
// descriptor ends behind the last VALID row -- a store to a row past the sequence end is dropped by the range check, so the
// store instructions need no exec mask and no branch (they sit between MFMAs).
struct WideRows {
  __amdgpu_buffer_rsrc_t rsrc;
  HSTU_DEV void open(char* tile_row0, int rows_valid, int64_t row_stride_bytes) {
    const int64_t n = rows_valid > 0 ? (int64_t)min(rows_valid, 32) * row_stride_bytes : 0;
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)tile_row0, (short)0, (int)n, 0x00020000);
  }
  HSTU_DEV void store16(const u32x4& v, int byte_off) const { __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, 0); }
};

// A finished dk / dv tile leaves the wave in two moves, both by its OWNER alone (no workgroup barrier): (1) PARK: the
// transposed accumulator blocks from register ACC on (column n32 = key, registers = features), scaled and rounded to the I/O
// dtype, as a swizzled row-major [32 keys][D] tile into the LDS slot of the same key tile's K (or V), which is dead by then
// (fold_park_tile, one block's 16 registers at a time); (2) COPY OUT: whole 256-byte rows, 16 bytes per lane, four rows per
// store instruction.  (Storing the accumulators directly -- 32-byte pieces of 32 rows per instruction -- was measured: a
// quarter of the kernel's time, profiles/r04_wide_ab.txt.)
template <typename T, int D, int ACC>
HSTU_DEV void wide_park_tile(float scale, char* __restrict__ tile, int lane) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  const int n32 = lane & 31, hf = lane >> 5;
  wide_static_for<0, D / 32>([&](auto ic) {
    constexpr int d = decltype(ic)::value;
    const f32x16 blk = wide_acc_read<ACC + 16 * d>();
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      u32x2 v = {Elem<T>::pk2(blk[4 * rq] * scale, blk[4 * rq + 1] * scale), Elem<T>::pk2(blk[4 * rq + 2] * scale, blk[4 * rq + 3] * scale)};
      *LDS_PTR(u32x2, tile + tile_off<UPR>(n32, 4 * d + rq) + 8 * hf) = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// ---- side work: the memory instructions of a step, between the MFMAs of the step's first pair ---------------------------------
// A [32][128] 16-bit tile = 8 LDS-DMA chunks of 1 KiB; wave w issues chunks 4 i + w (i = 0, 1): rows 16 i + 4 w + (lane >> 4),
// and for those rows the swizzle term is ((lane >> 4) << 2) | w whatever i is -- one lane-constant unit offset per wave.
constexpr int kWideChunksPerWave = 2;
struct WideTileSrc {
  const char* base;       // row 0 of the (user, head) in the source array
  uint32_t stride;        // bytes per row
  int row0, len;          // first row of the tile, rows of the user (rows >= len fetch a clamped copy of the last row)
  uint32_t lds;           // LDS byte address of the destination tile
  bool on;
};
HSTU_DEV void wide_dma_chunk(const WideTileSrc& t, int i, int wave, int lane) {
  const int r = t.row0 + 16 * i + 4 * wave + (lane >> 4);
  const uint32_t uoff = (uint32_t)(((lane & 15) ^ (((lane >> 4) << 2) | wave)) << 4);
  const uint32_t grow = (uint32_t)min(r, t.len - 1);
  // (32-bit source offsets: the launcher sends batches with row strides >= 16 MiB or users spanning 4 GiB to the folded kernel)
  dma16_saddr_asm(__umul24(grow, t.stride) + uoff, t.base, t.lds + (uint32_t)(4 * i + wave) * 1024u);
}

// Eight DMA instructions per wave and step: the four Q / dO tiles of the next step's stage.  Everything wave-uniform except the
// lane's offsets.  Inside an MFMA stream they must be straight-line code (a branch per instruction cuts the stream into basic
// blocks): WideSideT fixes at compile time whether the second query tile's pair is on; any other combination runs in a
// block before the pair.  Placement: stage A behind every fourth MFMA of the S / dP stream, stage B of the dV / dK stream.
struct WideSide {
  WideTileSrc st[4];       // stage: Q_a, dO_a (on / off together), Q_b, dO_b (on / off together)
  int wave, lane;
  HSTU_DEV void run_all() const {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (st[j].on)
#pragma unroll
        for (int i = 0; i < kWideChunksPerWave; ++i) wide_dma_chunk(st[j], i, wave, lane);
  }
};
template <bool STB>
struct WideSideT {
  const WideSide& w;
  HSTU_DEV void operator()(int slot) const {   // slot 0..15: behind MFMA `slot` of the S / dP stream, 16..31: of the dV / dK stream
    if ((slot & 3) != 0) return;
    const int n = (slot & 15) >> 2;            // 0..3
    if (slot < 16) wide_dma_chunk(w.st[n >> 1], n & 1, w.wave, w.lane);
    else if (STB) wide_dma_chunk(w.st[2 + (n >> 1)], n & 1, w.wave, w.lane);
  }
};

// what the previous problem of this workgroup has already requested for the current one: all of its K / V tiles and the
// first step's Q / dO tiles, or nothing
struct WidePre {
  bool all;
};

// One (user, head) problem on the calling workgroup (all of its LDS).
template <typename T, int D>
HSTU_DEV void wide_problem(const HstuAttnBwdParams& bp, int tmax, int uh, char* smem, int tid, int wave, int uh_next, WidePre& pre) {
  using C = BwdCfg<T, D, D>;
  using W = WideCfg<T, D>;
  using E = Elem<T>;
  static_assert(C::EB == 2 && D == 128, "the wide backward is built for 16-bit I/O at head dim 128");
  constexpr bool REG_HI = (WIDE_KV_REGS & 2) != 0;
  // (the thread id is laundered per problem and per phase: lane-constant LDS offsets, DMA plans and mask patterns are then
  // recomputed where they are used -- a few dozen VALU instructions -- instead of being hoisted to the kernel's entry and
  // spilled around everything: 256 accumulators + 64 fragment registers leave ~190 registers for the working set)
  auto fresh = [](int x) { asm volatile("" : "+v"(x)); return x; };
  const int tid0 = tid;
  int lane = fresh(tid0) & 63;
  const HstuAttnParams& p = bp.fwd;
  const int b = user_of_slot(p, ((WIDE_ABLATE & 128) ? uh % 256 : (WIDE_ABLATE & 256) ? uh % 32 : uh) / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * tmax);
  const bool pre_all = pre.all;
  pre.all = false;
  if (len <= 0) return;     // (workgroup-uniform; nothing was requested for an empty user)
  const int b3 = uh_next >= 0 ? user_of_slot(p, uh_next / p.heads) : b, hd3 = uh_next >= 0 ? uh_next % p.heads : 0;
  const int64_t off3 = uh_next >= 0 ? load_index(p.seq_offsets, b3, p.offsets_dtype) : 0;
  const int len3 = uh_next >= 0 ? min((int)(load_index(p.seq_offsets, b3 + 1, p.offsets_dtype) - off3), 32 * tmax) : 0;
  const int nt3 = (len3 + 31) >> 5;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  HSTU_TRACE_DECL(bp.workspace, bp.workspace != nullptr && uh == 4096);
  HSTU_MARK(1);

  const int nt = (len + 31) >> 5;        // tiles of this user (<= tmax <= 7)
  const int ns = (nt + 1) >> 1;          // steps
  const int a_last = nt - ns;            // diagonal tile of the last step: tiles <= a_last are final only after its pairs
  const int lo = wave, hi = W::kMaxTiles - wave;      // the two key tiles of this wave
  char* const stage0 = smem + W::kMaxTiles * C::PAIR;  // stage set 0: two Q + dO pairs behind the K/V slots
  char* const dsbuf = stage0 + 2 * C::PAIR;
  // stage set s, side sd (0: query tile a, 1: query tile b): Q tile, dO tile.  Set 1 = the V slots of key tiles 0..3 (their
  // fragments are in registers from the top of step 0 on)
  auto q_tile = [&](int set, int sd) { return set == 0 ? stage0 + sd * C::PAIR : smem + (2 * sd) * C::PAIR + C::KT; };
  auto do_tile = [&](int set, int sd) { return set == 0 ? stage0 + sd * C::PAIR + C::KT : smem + (2 * sd + 1) * C::PAIR + C::KT; };

  const char* qbase = (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const char* dobase = (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
  char* const dq_head = (char*)bp.dq + (off0 * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB;
  char* const dk_head = (char*)bp.dk + (off0 * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
  char* const dv_head = (char*)bp.dv + (off0 * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
  const int64_t dq_rs = bp.dq_row_stride * C::EB, dk_rs = bp.dk_row_stride * C::EB, dv_rs = bp.dv_row_stride * C::EB;
  const int64_t q_rs = p.q_row_stride * C::EB, k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB,
                do_rs = bp.do_row_stride * C::EB;
  // the next problem's sources
  const char* qb3 = (const char*)p.q + (off3 * p.q_row_stride + (int64_t)hd3 * p.q_head_stride) * C::EB;
  const char* kb3 = (const char*)p.k + (off3 * p.k_row_stride + (int64_t)hd3 * p.k_head_stride) * C::EB;
  const char* vb3 = (const char*)p.v + (off3 * p.v_row_stride + (int64_t)hd3 * p.v_head_stride) * C::EB;
  const char* dob3 = (const char*)bp.dout + (off3 * bp.do_row_stride + (int64_t)hd3 * bp.do_head_stride) * C::EB;

  auto lds_of = [](const char* ptr) { return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr); };
  auto tile_src = [&](const char* base, int64_t rs, int row0, int rows, const char* dst, bool on) {
    WideTileSrc t;
    t.base = base; t.stride = (uint32_t)rs; t.row0 = row0; t.len = rows; t.lds = lds_of(dst); t.on = on;
    return t;
  };
  auto dma_tile = [&](const WideTileSrc& t) {   // a whole tile at once (prologue, end of a problem)
    const int ln = fresh(tid0) & 63;
    if (t.on)
#pragma unroll
      for (int i = 0; i < kWideChunksPerWave; ++i) wide_dma_chunk(t, i, wave, ln);
  };
  auto k_src = [&](const char* base, int t, int rows, bool on) { return tile_src(base, k_rs, 32 * t, rows, smem + t * C::PAIR, on); };
  auto v_src = [&](const char* base, int t, int rows, bool on) { return tile_src(base, v_rs, 32 * t, rows, smem + t * C::PAIR + C::KT, on); };

  // ---- prologue of a workgroup's first problem (every later one finds its tiles requested by its predecessor)
  if (!pre_all) {
    if (!(WIDE_ABLATE & 16))
      for (int t = 0; t < nt; ++t) {
        dma_tile(k_src(kbase, t, len, true));
        dma_tile(v_src(vbase, t, len, true));
      }
    dma_tile(tile_src(qbase, q_rs, 32 * (nt - 1), len, q_tile(0, 0), true));
    dma_tile(tile_src(dobase, do_rs, 32 * (nt - 1), len, do_tile(0, 0), true));
    dma_tile(tile_src(qbase, q_rs, 0, len, q_tile(0, 1), nt > 1));
    dma_tile(tile_src(dobase, do_rs, 0, len, do_tile(0, 1), nt > 1));
  }
  HSTU_MARK(2);

  wide_acc_zero();
  u32x4 kf_lo[D / 16], vf_lo[D / 16], kf_hi[D / 16], vf_hi[D / 16];
  const float scale_v = attn_scale_of(p);
  const float ds_scale = scale_v * p.alpha;

  // lane-constant mask patterns of the plain-causal case (see fold_pair_x)
  int dmvm = 0;
  {
    const int n32 = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hf;
      dmvm |= (n32 <= row ? 1 : 0) << r;
      dmvm |= (32 * (nt - 1) + row < len ? 1 : 0) << (16 + r);
    }
  }
  auto owner_of = [](int t) { return t < kWideWaves ? t : W::kMaxTiles - t; };
  const bool nxt = WIDE_PERSIST >= 2 && nt3 > 0 && !(WIDE_ABLATE & 16);
  // Everything below that touches K/V slot t outside the pairs and the dQ GEMM is done by ONE wave, owner_of(t): parking the
  // finished dk / dv tile in the slot, copying it out, and then requesting the next problem's K / V tile t into it -- the
  // owner's own program order is all the synchronisation these three need.  `whole`: all 8 chunks of a tile by this wave.
  auto dma_whole = [&](const WideTileSrc& t) {
    const int ln = fresh(tid0) & 63;
#pragma unroll
    for (int w2 = 0; w2 < kWideWaves; ++w2)
#pragma unroll
      for (int i = 0; i < kWideChunksPerWave; ++i) wide_dma_chunk(t, i, w2, ln);
  };
  auto next_k = [&](int t) { if (nxt && t < nt3) dma_whole(k_src(kb3, t, len3, true)); };
  auto next_v = [&](int t) { if (nxt && t < nt3) dma_whole(v_src(vb3, t, len3, true)); };
  // dk (which = 0) or dv (1) of this wave's key tile t: registers -> the dead LDS tile `slot` -> memory
  auto retire = [&](int t, int which, char* slot) {
    if (WIDE_ABLATE & 2) return;
    const int ln = fresh(tid0) & 63;
    wide_acc_drain();
    if (t < kWideWaves) {
      if (which) wide_park_tile<T, D, kWideAccDvLo>(scale_v, slot, ln);
      else wide_park_tile<T, D, kWideAccDkLo>(ds_scale, slot, ln);
    } else {
      if (which) wide_park_tile<T, D, kWideAccDvHi>(scale_v, slot, ln);
      else wide_park_tile<T, D, kWideAccDkHi>(ds_scale, slot, ln);
    }
    char* const g = which ? dv_head + (int64_t)(32 * t) * dv_rs : dk_head + (int64_t)(32 * t) * dk_rs;
    wide_copy_out<T, D, 64>(slot, g, which ? dv_rs : dk_rs, len - 32 * t, ln);
  };

  WideSide side;
  side.wave = wave;
  bool st0_req = false;        // the next problem's first Q / dO tiles are requested

  for (int k = 0; k < ns; ++k) {
    const int a = nt - 1 - k, bq = k;
    const bool b_on = bq < a, last = k + 1 == ns;
    const int set = k & 1;
    // this step's Q/dO tiles (first time: K/V) have landed -- every wave has waited for its own DMA instructions: at the end of
    // the previous step, or here for what the prologue / the previous problem issued --, dS' of the last step is consumed
    if (k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    HSTU_MARK(10);
    lane = fresh(tid0) & 63;
    if (k == 0) {
      if (lo < nt) {
        wide_load_kv_frags<T, D>(kf_lo, smem + lo * C::PAIR, lane);
        wide_load_kv_frags<T, D>(vf_lo, smem + lo * C::PAIR + C::KT, lane);
      }
      if (REG_HI && hi < nt) {
        wide_load_kv_frags<T, D>(kf_hi, smem + hi * C::PAIR, lane);
        wide_load_kv_frags<T, D>(vf_hi, smem + hi * C::PAIR + C::KT, lane);
      }
      lds_barrier();           // from here on the V slots of tiles 0..3 are stage set 1
      // slots this problem does not use: the next problem's tiles right away (the V slots of tiles 0..3 at the end: stage set 1)
      if (lo >= nt) next_k(lo);
      if (hi >= nt) { next_k(hi); next_v(hi); }
    } else if (a + 1 > a_last && wave == owner_of(a + 1)) {
      // the key tile whose diagonal pair was in the previous step is final (unless the other end of the walk still reaches
      // it: the tiles <= a_last, retired after the last step's pairs); every dQ GEMM that read its K tile is done
      // (both through the K slot, one after the other: the V slots of tiles 0..3 may be holding this step's Q / dO tiles)
      char* const kslot = smem + (a + 1) * C::PAIR;
      retire(a + 1, 1, kslot);
      retire(a + 1, 0, kslot);
      next_k(a + 1);
      if (a + 1 >= kWideWaves) next_v(a + 1);
    }
    // ---- the step's stage requests: the next step's Q / dO tiles into the other stage set; in the last step, if that set is
    // stage set 0, the next problem's first tiles
    side.lane = lane;
    if (!last) {
      const int qa = a - 1, qb = bq + 1;
      const bool bn = qb < qa;
      const bool on = !(WIDE_ABLATE & 4);
      side.st[0] = tile_src(qbase, q_rs, 32 * qa, len, q_tile(set ^ 1, 0), on);
      side.st[1] = tile_src(dobase, do_rs, 32 * qa, len, do_tile(set ^ 1, 0), on);
      side.st[2] = tile_src(qbase, q_rs, 32 * qb, len, q_tile(set ^ 1, 1), on && bn);
      side.st[3] = tile_src(dobase, do_rs, 32 * qb, len, do_tile(set ^ 1, 1), on && bn);
    } else {
      const bool on = nxt && set == 1;
      side.st[0] = tile_src(qb3, q_rs, 32 * (nt3 - 1), len3, q_tile(0, 0), on);
      side.st[1] = tile_src(dob3, do_rs, 32 * (nt3 - 1), len3, do_tile(0, 0), on);
      side.st[2] = tile_src(qb3, q_rs, 0, len3, q_tile(0, 1), on && nt3 > 1);
      side.st[3] = tile_src(dob3, do_rs, 0, len3, do_tile(0, 1), on && nt3 > 1);
      if (on) st0_req = true;
    }
    // ---- phase 1: this wave's pairs of the step: (a, lo) carrying the stage requests, then (a, hi) or (b, lo).  (A pair the
    // attention window rules out entirely is run like any other: every element masked, exact zeros published and accumulated.)
    const bool run_pairs = !(WIDE_ABLATE & 64);
    {
      const bool sta_on = side.st[0].on, stb_on = side.st[2].on;
      const int combo = (WIDE_SIDE_IN_STREAM && lo <= a && run_pairs && sta_on) ? (stb_on ? 2 : 1) : 0;
      if (combo == 0) side.run_all();
      if (lo <= a && run_pairs) {
        char* const myds = dsbuf + lo * W::DSB;
#define WIDE_PAIR_X(SIDE_) wide_pair<T, D, true, kWideAccDkLo>(p, mc, nullptr, nullptr, kf_lo, vf_lo, q_tile(set, 0), do_tile(set, 0), myds, 32 * a, 32 * lo, lane, dmvm, SIDE_ HSTU_TRACE_PASS)
        if (combo == 2) WIDE_PAIR_X((WideSideT<true>{side}));
        else if (combo == 1) WIDE_PAIR_X((WideSideT<false>{side}));
        else WIDE_PAIR_X(WideNoSide());
#undef WIDE_PAIR_X
      }
    }
    if (run_pairs) {
      if (hi <= a) {
        wide_pair<T, D, REG_HI, kWideAccDkHi>(p, mc, smem + hi * C::PAIR, smem + hi * C::PAIR + C::KT, kf_hi, vf_hi, q_tile(set, 0), do_tile(set, 0),
                                              dsbuf + hi * W::DSB, 32 * a, 32 * hi, lane, dmvm, WideNoSide() HSTU_TRACE_PASS);
      } else if (b_on && lo <= bq) {
        wide_pair<T, D, true, kWideAccDkLo>(p, mc, nullptr, nullptr, kf_lo, vf_lo, q_tile(set, 1), do_tile(set, 1), dsbuf + (W::kDsSlots - 1 - lo) * W::DSB,
                                            32 * bq, 32 * lo, lane, dmvm, WideNoSide() HSTU_TRACE_PASS);
      }
    }
    HSTU_MARK(13);
    lds_barrier();     // dS' of this step published; reads of this step's stage set done
    HSTU_MARK(15);
    lane = fresh(tid0) & 63;
    if (last) {
      // both stage sets and every V slot are dead: dv of the tiles that were waiting for the last pairs, then the next problem's
      // V tile of the slot (private to its owner from here on) and its first Q / dO tiles
      if (lo <= a_last) retire(lo, 1, smem + lo * C::PAIR + C::KT);
      next_v(lo);
      if (nxt && !st0_req) {
        dma_tile(tile_src(qb3, q_rs, 32 * (nt3 - 1), len3, q_tile(0, 0), true));
        dma_tile(tile_src(dob3, do_rs, 32 * (nt3 - 1), len3, do_tile(0, 0), true));
        dma_tile(tile_src(qb3, q_rs, 0, len3, q_tile(0, 1), nt3 > 1));
        dma_tile(tile_src(dob3, do_rs, 0, len3, do_tile(0, 1), nt3 > 1));
        st0_req = true;
      }
    }
    HSTU_MARK(18);
    // ---- phase 2: dQ of the two query tiles, 32 feature columns per wave.  Between the GEMM and its stores the wave waits for
    // the DMA instructions it issued in this step (a pair and a GEMM ago): the next step starts with a bare barrier, and the
    // stores issued here have a whole step to drain before anything waits on the counter again.
    {
      u32x4 qx[2], qy[2];
      if (!(WIDE_ABLATE & 32)) {
        const f32x16 qa = wide_dq_side<T, D>(smem, dsbuf, W::DSB, a + 1, wave, lane);
        HSTU_MARK(16);
        wide_pack_block<T>(qa, ds_scale, qx[0], qy[0]);
        if (b_on) {
          const f32x16 qb = wide_dq_side<T, D>(smem, dsbuf + (W::kDsSlots - 1) * W::DSB, -W::DSB, bq + 1, wave, lane);
          wide_pack_block<T>(qb, ds_scale, qx[1], qy[1]);
        }
      }
      if (!last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last step's requests are the next problem's: waited for at its top)
      if (!(WIDE_ABLATE & 32) && !(WIDE_ABLATE & 1)) {
        const int off = (lane & 31) * (int)dq_rs + (32 * wave + 8 * (lane >> 5)) * C::EB;
        WideRows ra;
        ra.open(dq_head + (int64_t)(32 * a) * dq_rs, len - 32 * a, dq_rs);
        ra.store16(qx[0], off);
        ra.store16(qy[0], off + 16 * C::EB);
        if (b_on) {
          WideRows rb;
          rb.open(dq_head + (int64_t)(32 * bq) * dq_rs, len - 32 * bq, dq_rs);
          rb.store16(qx[1], off);
          rb.store16(qy[1], off + 16 * C::EB);
        }
      }
    }
    HSTU_MARK(17);
  }
  HSTU_MARK(20);
  // ---- the K tiles the last dQ GEMM was still reading: dk of the tiles retired at the end, then the next problem's K tile
  lds_barrier();
  if (lo <= a_last) {
    retire(lo, 0, smem + lo * C::PAIR);
    next_k(lo);
  }
  pre.all = nxt;
  HSTU_MARK(21);
}

template <typename T, int D>
__global__ __launch_bounds__(kWideThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void hstu_attn_bwd_wide_kernel(const HstuAttnBwdParams bp, int tmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = bp.fwd.batch * bp.fwd.heads;
  WidePre pre;
  pre.all = false;
  if (WIDE_PERSIST) {
    for (int uh = blockIdx.x; uh < total; uh += gridDim.x) {
      int uh_l = uh;
      asm volatile("" : "+s"(uh_l));     // nothing of problem i+1 is hoisted into problem i
      const int uh_n = (WIDE_PERSIST >= 2 && uh_l + (int)gridDim.x < total) ? uh_l + (int)gridDim.x : -1;
      wide_problem<T, D>(bp, tmax, uh_l, smem, tid, wave, uh_n, pre);
    }
  } else {
    wide_problem<T, D>(bp, tmax, blockIdx.x, smem, tid, wave, -1, pre);
  }
}

}  // namespace hstu
