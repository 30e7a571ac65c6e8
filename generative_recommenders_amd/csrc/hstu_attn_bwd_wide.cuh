// HSTU attention backward, "wide" schedule for the metric shape (head dim 128, 16-bit I/O, <= 7 tiles of 32 rows):
// FOUR waves per workgroup, one per SIMD, each with the whole 512-entry register file of its SIMD.
//
// Same math, tiles, masks and LDS formats as the folded schedule (hstu_attn_bwd_fold.cuh); what changes is who owns
// what.  The folded kernel runs 8 waves of 256 registers: a wave owns ONE key tile (128 accumulator registers), reads
// both operands of every S / dP MFMA from LDS (the stream runs at the LDS rate, not the matrix rate), and the causal
// triangle is balanced by letting wave 7 - t open a second partial sum of key tile t that is handed over through LDS in
// a tail.  Here wave w owns key tiles w AND 7 - w for the whole problem:
//   * dK / dV of both tiles live in the accumulator half of the register file (256 AGPRs), for the whole problem: no
//     second partial sum, no hand-over tail;
//   * the K / V row fragments of both tiles -- the B operands of S = Q K^T and dP = dO V^T -- live in 128 VGPRs, read
//     from LDS once per problem: the S / dP stream reads only its A operand (the streamed Q / dO tile) from LDS;
//   * the triangle is balanced by the same two-ended walk (step k: query tiles a = nt-1-k and b = k), now inside one
//     wave: per step a wave runs (a, w) and one of (a, 7 - w) | (b, w) -- two pairs per step on every wave at 7 tiles;
//   * dQ of the step's two query tiles: wave w owns the 32 features [32 w, +32) of both tiles, a 32x32x16 GEMM over the
//     published dS' tiles and the transposed K tiles (K stays in LDS for this), stored as 16-byte pieces
//     (v_permlane32_swap).
// Requires what the folded kernel requires (no contextual rows, no bias, head dims equal to the instantiated ones,
// max_seq_len <= 224, 1e-20 < |alpha| < 1e6); HSTU_BWD_WIDE=0 sends the shape back to the folded kernel.
#pragma once
#include "hstu_attn_bwd_fold.cuh"

#ifndef WIDE_KV_REGS
#define WIDE_KV_REGS 1     // bit 0: K/V fragments of the low tile in registers, bit 1: of the high tile (0: both operands from LDS)
#endif
#ifndef WIDE_ACC_HI_VGPR
#define WIDE_ACC_HI_VGPR 1 // dK / dV of the high tile in architectural registers (asm MFMAs), of the low tile in the accumulator file
#endif
#ifndef WIDE_SDP_AHEAD
#define WIDE_SDP_AHEAD 4   // LDS fragments requested ahead of their MFMA in the S / dP and dV / dK streams
#endif
#ifndef WIDE_DQ_AHEAD
#define WIDE_DQ_AHEAD 2    // key tiles whose fragments are in flight ahead of the dQ GEMM's MFMAs
#endif
#ifndef WIDE_PERSIST
#define WIDE_PERSIST 2     // as FOLD_PERSIST
#endif
#ifndef WIDE_ABLATE
#define WIDE_ABLATE 0      // timing experiments only (WRONG results): bits as FOLD_ABLATE
#endif

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


// 32x32x16 MFMA with its accumulator in the ARCHITECTURAL registers, through inline asm.  hipcc selects ONE form for every
// MFMA builtin of a function: with 256 accumulator registers (the two tiles' dK / dV) it is the AGPR form, and then the S / dP
// and dQ accumulators would have to be AGPRs as well -- 288 > 256: hundreds of spills and copies.  These chains therefore
// name their instruction themselves ("+v"): dK / dV own the whole accumulator file, everything transient stays in VGPRs.
// hipcc does not model the instruction inside an asm statement (guide 5.7): `FIRST` (a constant after unrolling) pads the VALU-write -> MFMA-read hazard
// of a freshly initialised accumulator, wide_mfma_drain() the MFMA-write -> VALU-read one after the chain's last MFMA
// (8-pass XDL: 11 wait states); back-to-back accumulation on one register tuple needs none.
template <typename T> struct WideMma;
template <> struct WideMma<bf16_t> {
  static HSTU_DEV void run(bool FIRST, const Elem<bf16_t>::Frag& a, const Elem<bf16_t>::Frag& b, f32x16& c) {
    const u32x4 av = __builtin_bit_cast(u32x4, a.v), bv = __builtin_bit_cast(u32x4, b.v);
    if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
  }
};
template <> struct WideMma<f16_t> {
  static HSTU_DEV void run(bool FIRST, const Elem<f16_t>::Frag& a, const Elem<f16_t>::Frag& b, f32x16& c) {
    const u32x4 av = __builtin_bit_cast(u32x4, a.v), bv = __builtin_bit_cast(u32x4, b.v);
    if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
  }
};
// the same with the B operand (a resident K / V fragment) in the accumulator file: MFMA A / B operands may be AGPRs
template <typename T> struct WideMmaBA;
template <> struct WideMmaBA<bf16_t> {
  static HSTU_DEV void run(bool FIRST, const Elem<bf16_t>::Frag& a, const u32x4& b_agpr, f32x16& c) {
    const u32x4 av = __builtin_bit_cast(u32x4, a.v);
    if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(b_agpr));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(b_agpr));
  }
};
template <> struct WideMmaBA<f16_t> {
  static HSTU_DEV void run(bool FIRST, const Elem<f16_t>::Frag& a, const u32x4& b_agpr, f32x16& c) {
    const u32x4 av = __builtin_bit_cast(u32x4, a.v);
    if (FIRST) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(b_agpr));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(av), "a"(b_agpr));
  }
};
// after the last MFMA of an asm chain, before the first VALU instruction that reads its accumulators
HSTU_DEV void wide_mfma_drain(f32x16& c0, f32x16& c1) { asm volatile("s_nop 15" : "+v"(c0), "+v"(c1)); }
HSTU_DEV void wide_mfma_drain(f32x16& c0) { asm volatile("s_nop 15" : "+v"(c0)); }

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
template <typename T, int D, bool KVREG, bool ACCV>
HSTU_DEV void wide_pair(const HstuAttnParams& p, const MaskCtx& mc, const char* __restrict__ Kw, const char* __restrict__ Vw,
                        const u32x4 (&kf)[D / 16], const u32x4 (&vf)[D / 16],
                        const char* __restrict__ Qs, const char* __restrict__ dOs, char* __restrict__ myds, int i0, int k0,
                        f32x16 (&dk_acc)[D / 32], f32x16 (&dv_acc)[D / 32], int lane, int dmvm HSTU_TRACE_ARG) {
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
        if (m & 1) WideMmaBA<T>::run(m < 2, fa[m % (AHEAD + 1)], vf[m >> 1], dp);
        else WideMmaBA<T>::run(m < 2, fa[m % (AHEAD + 1)], kf[m >> 1], s);
      } else {
        if (m & 1) WideMma<T>::run(m < 2, fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)], dp);
        else WideMma<T>::run(m < 2, fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)], s);
      }
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
  // dV_w^T[dv][key] += dO_i^T[dv][q] P'[q][key]   and   dK_w^T[d][key] += Q_i^T[d][q] dS'[q][key]
  {
    constexpr int NM = 2 * 2 * C::DBQ, AHEAD = WIDE_SDP_AHEAD;   // (dV | dK) x d block x k half
    Frag fa[AHEAD + 1];
    auto load_item = [&](int m, Frag& a) {
      const int ks = (m >> 1) & 1, d = m >> 2;
      a = lds_col_frag<T, C::UPR_K>((m & 1) ? Qs : dOs, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, lane);
    };
#pragma unroll
    for (int m = 0; m < AHEAD; ++m) load_item(m, fa[m % (AHEAD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (m + AHEAD < NM) load_item(m + AHEAD, fa[(m + AHEAD) % (AHEAD + 1)]);
      const int ks = (m >> 1) & 1, d = m >> 2;
      if constexpr (ACCV) {   // accumulators in VGPRs (the high tile): the asm form; pb / dsb were written by VALU just now
        if (m & 1) WideMma<T>::run(m < 4, fa[m % (AHEAD + 1)], dsb[ks], dk_acc[d]);
        else WideMma<T>::run(m < 4, fa[m % (AHEAD + 1)], pb[ks], dv_acc[d]);
      } else {
        if (m & 1) dk_acc[d] = E::mma(fa[m % (AHEAD + 1)], dsb[ks], dk_acc[d]);
        else dv_acc[d] = E::mma(fa[m % (AHEAD + 1)], pb[ks], dv_acc[d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // publish dS' as [key = n32][q]: this lane holds q = 4 hf + 8 rq + (0..3) = chunk hf + 2 rq
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const u32x4 w = __builtin_bit_cast(u32x4, dsb[rq >> 1].v);
    u32x2 v2 = {w[2 * (rq & 1)], w[2 * (rq & 1) + 1]};
    *LDS_PTR(u32x2, myds + fold_ds_off(n32, hf + 2 * rq)) = v2;
  }
}

// a pair the attention window rules out entirely publishes zeros (the dQ GEMM reads every tile on or below the diagonal)
HSTU_DEV void wide_publish_zero(char* __restrict__ myds, int lane) {
  const int n32 = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) *LDS_PTR(u32x2, myds + fold_ds_off(n32, hf + 2 * rq)) = u32x2{0u, 0u};
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
    WideMma<T>::run(t == 0, a0[t % (AH + 1)], b0[t % (AH + 1)], acc);
    WideMma<T>::run(false, a1[t % (AH + 1)], b1[t % (AH + 1)], acc);
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

// C layout of a dQ^T block: column n32 = query row, register r = feature (r & 3) + 8 (r >> 2) + 4 hf of the wave's 32.
// A lane's four 4-feature groups are paired up with v_permlane32_swap (lanes n32 and n32 + 32 hold the same query row) into
// two runs of 8 consecutive features: lanes 0..31 store features [0, 8) and [16, 24) of the block, lanes 32..63 [8, 16) and
// [24, 32) -- two 16-byte stores per lane instead of four scattered 8-byte ones.
template <typename T, int D>
HSTU_DEV void wide_dq_store(const HstuAttnBwdParams& bp, const f32x16& acc, float ds_scale, int q0, int len, int64_t off0, int hd,
                            int db, int lane) {
  using C = BwdCfg<T, D, D>;
  using E = Elem<T>;
  const int n32 = lane & 31, hf = lane >> 5;
  uint32_t g[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    g[j][0] = E::pk2(acc[4 * j] * ds_scale, acc[4 * j + 1] * ds_scale);
    g[j][1] = E::pk2(acc[4 * j + 2] * ds_scale, acc[4 * j + 3] * ds_scale);
  }
#pragma unroll
  for (int jp = 0; jp < 4; jp += 2)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // lanes 32..63 of the first operand <-> lanes 0..31 of the second
      const auto sw = __builtin_amdgcn_permlane32_swap(g[jp][h], g[jp + 1][h], false, false);
      g[jp][h] = sw[0];
      g[jp + 1][h] = sw[1];
    }
  const int qrow = q0 + n32;
  if (qrow < len && (!(WIDE_ABLATE & 1) || bp.total_rows == -12345)) {
    char* dqrow = (char*)bp.dq + ((off0 + qrow) * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB;
    char* dst = dqrow + (32 * db + 8 * hf) * C::EB;
    gstore16(dst, u32x4{g[0][0], g[0][1], g[1][0], g[1][1]});
    gstore16(dst + 16 * C::EB, u32x4{g[2][0], g[2][1], g[3][0], g[3][1]});
  }
}

// One (user, head) problem on the calling workgroup (all of its LDS).
template <typename T, int D>
HSTU_DEV void wide_problem(const HstuAttnBwdParams& bp, int tmax, int uh, char* smem, int tid, int wave, int uh_next,
                           int& pre_lo) {
  using C = BwdCfg<T, D, D>;
  using W = WideCfg<T, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  int lane;
  static_assert(C::EB == 2 && D == 128, "the wide backward is built for 16-bit I/O at head dim 128");
  constexpr bool REG_LO = (WIDE_KV_REGS & 1) != 0, REG_HI = (WIDE_KV_REGS & 2) != 0, ACC_HI_V = WIDE_ACC_HI_VGPR != 0;
  // (the thread id is laundered per problem and per phase: lane-constant LDS offsets, DMA plans and mask patterns are then
  // recomputed where they are used -- a few dozen VALU instructions -- instead of being hoisted to the kernel's entry and
  // spilled around everything: 256 accumulators + 128 fragment registers leave ~100 VGPRs for the working set)
  auto fresh = [](int x) { asm volatile("" : "+v"(x)); return x; };
  const int tid0 = tid;
  tid = fresh(tid0);
  lane = tid & 63;
  const HstuAttnParams& p = bp.fwd;
  const int b = user_of_slot(p, ((WIDE_ABLATE & 128) ? uh % 256 : (WIDE_ABLATE & 256) ? uh % 32 : uh) / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * tmax);
  const int pre_in = pre_lo;             // K/V tiles >= pre_in of THIS problem were issued by the previous problem's tail
  pre_lo = W::kMaxTiles;
  if (len <= 0) return;
  const int b3 = uh_next >= 0 ? user_of_slot(p, uh_next / p.heads) : b, hd3 = uh_next >= 0 ? uh_next % p.heads : 0;
  const int64_t off3 = uh_next >= 0 ? load_index(p.seq_offsets, b3, p.offsets_dtype) : 0;
  const int len3 = uh_next >= 0 ? min((int)(load_index(p.seq_offsets, b3 + 1, p.offsets_dtype) - off3), 32 * tmax) : 0;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  HSTU_TRACE_DECL(bp.workspace, bp.workspace != nullptr && uh == 4096);
  HSTU_MARK(1);

  const int nt = (len + 31) >> 5;        // tiles of this user (<= tmax <= 7)
  const int ns = (nt + 1) >> 1;          // steps
  const int a_last = nt - ns;            // diagonal tile of the last step: tiles <= a_last are final only after the loop
  const int lo = wave, hi = W::kMaxTiles - wave;      // the two key tiles of this wave
  char* const stageA = smem + W::kMaxTiles * C::PAIR;
  char* const stageB = stageA + C::PAIR;
  char* const dsbuf = stageB + C::PAIR;

  const char* qbase = (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const char* dobase = (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
  char* const dk_head = (char*)bp.dk + (off0 * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
  char* const dv_head = (char*)bp.dv + (off0 * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
  const int64_t dk_rs = bp.dk_row_stride * C::EB, dv_rs = bp.dv_row_stride * C::EB;
  const int64_t q_rs = p.q_row_stride * C::EB, k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB,
                do_rs = bp.do_row_stride * C::EB;

  const int len_max = 32 * tmax;
  const bool dma_fast = FOLD_DMA_FAST && q_rs < (1 << 24) && k_rs < (1 << 24) && v_rs < (1 << 24) && do_rs < (1 << 24) &&
                        (int64_t)len_max * q_rs < (1LL << 32) && (int64_t)len_max * k_rs < (1LL << 32) &&
                        (int64_t)len_max * v_rs < (1LL << 32) && (int64_t)len_max * do_rs < (1LL << 32);
  auto stage_dma = [&](int qa, int qb, bool b_on) {
    const int ln = fresh(tid0) & 63;
    wide_tile_dma<T, D, kWideWaves>(stageA, qbase, q_rs, 32 * qa, len, wave, ln, dma_fast);
    wide_tile_dma<T, D, kWideWaves>(stageA + C::KT, dobase, do_rs, 32 * qa, len, wave, ln, dma_fast);
    if (b_on) {
      wide_tile_dma<T, D, kWideWaves>(stageB, qbase, q_rs, 32 * qb, len, wave, ln, dma_fast);
      wide_tile_dma<T, D, kWideWaves>(stageB + C::KT, dobase, do_rs, 32 * qb, len, wave, ln, dma_fast);
    }
  };

  // ---- prologue: the K/V tiles the previous problem's tail has not already requested, and the first two query tiles
  for (int t = 0; t < ((WIDE_ABLATE & 16) ? 0 : min(nt, pre_in)); ++t) {
    char* dst = smem + t * C::PAIR;
    wide_tile_dma<T, D, kWideWaves>(dst, kbase, k_rs, 32 * t, len, wave, lane, dma_fast);
    wide_tile_dma<T, D, kWideWaves>(dst + C::KT, vbase, v_rs, 32 * t, len, wave, lane, dma_fast);
  }
  stage_dma(nt - 1, 0, 0 < nt - 1);
  HSTU_MARK(2);

  f32x16 dk_lo[C::DBQ], dv_lo[C::DBV], dk_hi[C::DBQ], dv_hi[C::DBV];
#pragma unroll
  for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_lo[d][r] = 0.f; dv_lo[d][r] = 0.f; dk_hi[d][r] = 0.f; dv_hi[d][r] = 0.f; }
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
  // owner wave of key tile t
  auto owner_of = [](int t) { return t < kWideWaves ? t : W::kMaxTiles - t; };

  for (int k = 0; k < ns; ++k) {
    const int a = nt - 1 - k, bq = k;
    const bool b_on = bq < a;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // Q/dO tiles of this step (and, first time, K/V) landed; dS' of the last step consumed
    HSTU_MARK(10);
    lane = fresh(tid0) & 63;
    if (k == 0) {
      if (REG_LO && lo < nt) {
        wide_load_kv_frags<T, D>(kf_lo, smem + lo * C::PAIR, lane);
        wide_load_kv_frags<T, D>(vf_lo, smem + lo * C::PAIR + C::KT, lane);
      }
      if (REG_HI && hi < nt) {
        wide_load_kv_frags<T, D>(kf_hi, smem + hi * C::PAIR, lane);
        wide_load_kv_frags<T, D>(vf_hi, smem + hi * C::PAIR + C::KT, lane);
      }
    }
    if (k > 0 && a + 1 > a_last && wave == owner_of(a + 1)) {
      // owner of the previous step's diagonal tile: its K tile is dead now (every dQ GEMM of that step is done): park dK in
      // its place (dV was parked at the end of that step)
      if (a + 1 < kWideWaves) fold_park_tile<T, D>(dk_lo, ds_scale, smem + (a + 1) * C::PAIR, lane);
      else fold_park_tile<T, D>(dk_hi, ds_scale, smem + (a + 1) * C::PAIR, lane);   // (its last MFMA is a step behind)
    }
    // ---- phase 1: this wave's pairs of the step: (a, lo), then (a, hi) or (b, lo)
    if (!(WIDE_ABLATE & 64)) {
      if (lo <= a) {
        char* const myds = dsbuf + lo * W::DSB;
        if (mc.win == 0 || mc.pair_may_be_active(32 * a, 32, 32 * lo, 32))
          wide_pair<T, D, REG_LO, false>(p, mc, smem + lo * C::PAIR, smem + lo * C::PAIR + C::KT, kf_lo, vf_lo, stageA, stageA + C::KT, myds,
                                  32 * a, 32 * lo, dk_lo, dv_lo, lane, dmvm HSTU_TRACE_PASS);
        else wide_publish_zero(myds, lane);
      }
      if (hi <= a) {
        char* const myds = dsbuf + hi * W::DSB;
        if (mc.win == 0 || mc.pair_may_be_active(32 * a, 32, 32 * hi, 32))
          wide_pair<T, D, REG_HI, ACC_HI_V>(p, mc, smem + hi * C::PAIR, smem + hi * C::PAIR + C::KT, kf_hi, vf_hi, stageA, stageA + C::KT, myds,
                                  32 * a, 32 * hi, dk_hi, dv_hi, lane, dmvm HSTU_TRACE_PASS);
        else wide_publish_zero(myds, lane);
      } else if (b_on && lo <= bq) {
        char* const myds = dsbuf + (W::kDsSlots - 1 - lo) * W::DSB;
        if (mc.win == 0 || mc.pair_may_be_active(32 * bq, 32, 32 * lo, 32))
          wide_pair<T, D, REG_LO, false>(p, mc, smem + lo * C::PAIR, smem + lo * C::PAIR + C::KT, kf_lo, vf_lo, stageB, stageB + C::KT, myds,
                                  32 * bq, 32 * lo, dk_lo, dv_lo, lane, dmvm HSTU_TRACE_PASS);
        else wide_publish_zero(myds, lane);
      }
    }
    HSTU_MARK(13);
    __syncthreads();   // dS' of this step published; stage reads done
    HSTU_MARK(15);
    tid = fresh(tid0);
    lane = tid & 63;
    if (k + 1 < ns && !(WIDE_ABLATE & 4)) stage_dma(a - 1, bq + 1, bq + 1 < a - 1);
    if (k > 0 && a + 1 > a_last) {   // dk / dv of the previous step's diagonal key tile (parked in K/V slot a + 1): out, by all waves
      const int kt1 = a + 1;
      wide_copy_out<T, D, kWideThreads>(smem + kt1 * C::PAIR, dk_head + (int64_t)(32 * kt1) * dk_rs, dk_rs, len - 32 * kt1, tid);
      wide_copy_out<T, D, kWideThreads>(smem + kt1 * C::PAIR + C::KT, dv_head + (int64_t)(32 * kt1) * dv_rs, dv_rs, len - 32 * kt1, tid);
    }
    HSTU_MARK(18);
    // ---- phase 2: dQ of the two query tiles, 32 feature columns per wave
    if (!(WIDE_ABLATE & 32)) {
      const f32x16 qa = wide_dq_side<T, D>(smem, dsbuf, W::DSB, a + 1, wave, lane);
      wide_dq_store<T, D>(bp, qa, ds_scale, 32 * a, len, off0, hd, wave, lane);
      if (b_on) {
        const f32x16 qb = wide_dq_side<T, D>(smem, dsbuf + (W::kDsSlots - 1) * W::DSB, -W::DSB, bq + 1, wave, lane);
        wide_dq_store<T, D>(bp, qb, ds_scale, 32 * bq, len, off0, hd, wave, lane);
      }
    }
    HSTU_MARK(17);
    lane = fresh(tid0) & 63;
    if (a > a_last && wave == owner_of(a)) {
      // diagonal step of a tile no later query tile reaches: dV is parked right away (V tiles are read by their owner only);
      // the K tile may still be read by other waves' dQ GEMM: dK follows after the next barrier
      if (a < kWideWaves) fold_park_tile<T, D>(dv_lo, scale_v, smem + a * C::PAIR + C::KT, lane);
      else fold_park_tile<T, D>(dv_hi, scale_v, smem + a * C::PAIR + C::KT, lane);
    }
    HSTU_MARK(23);
  }
  HSTU_MARK(20);
  if (WIDE_ABLATE & 8) return;
  // ---- tail: key tiles 0..a_last (all owned as `lo` tiles: a_last <= 3) are final now
  __syncthreads();     // K/V tiles, stages and dS' buffers are dead from here on
  tid = fresh(tid0);
  lane = tid & 63;
  if (len3 > 0) {
    // K/V slots above a_last are not touched by the tail: the next problem's tiles of those slots stream in under it
    const char* kb3 = (const char*)p.k + (off3 * p.k_row_stride + (int64_t)hd3 * p.k_head_stride) * C::EB;
    const char* vb3 = (const char*)p.v + (off3 * p.v_row_stride + (int64_t)hd3 * p.v_head_stride) * C::EB;
    const int nt3 = (len3 + 31) >> 5;
    for (int t = a_last + 1; t < nt3; ++t) {
      char* dst = smem + t * C::PAIR;
      wide_tile_dma<T, D, kWideWaves>(dst, kb3, k_rs, 32 * t, len3, wave, lane, dma_fast);
      wide_tile_dma<T, D, kWideWaves>(dst + C::KT, vb3, v_rs, 32 * t, len3, wave, lane, dma_fast);
    }
    pre_lo = a_last + 1;
  }
  if (lo <= a_last) {
    fold_park_tile<T, D>(dk_lo, ds_scale, smem + lo * C::PAIR, lane);
    fold_park_tile<T, D>(dv_lo, scale_v, smem + lo * C::PAIR + C::KT, lane);
  }
  __syncthreads();
  HSTU_MARK(24);
  tid = fresh(tid0);
  for (int t = 0; t <= a_last; ++t) {
    wide_copy_out<T, D, kWideThreads>(smem + t * C::PAIR, dk_head + (int64_t)(32 * t) * dk_rs, dk_rs, len - 32 * t, tid);
    wide_copy_out<T, D, kWideThreads>(smem + t * C::PAIR + C::KT, dv_head + (int64_t)(32 * t) * dv_rs, dv_rs, len - 32 * t, tid);
  }
  HSTU_MARK(21);
}

template <typename T, int D>
__global__ __launch_bounds__(kWideThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void hstu_attn_bwd_wide_kernel(const HstuAttnBwdParams bp, int tmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = bp.fwd.batch * bp.fwd.heads;
  int pre_lo = WideCfg<T, D>::kMaxTiles;
  if (WIDE_PERSIST) {
    for (int uh = blockIdx.x; uh < total; uh += gridDim.x) {
      int uh_l = uh;
      asm volatile("" : "+s"(uh_l));     // nothing of problem i+1 is hoisted into problem i
      const int uh_n = (WIDE_PERSIST >= 2 && uh_l + (int)gridDim.x < total) ? uh_l + (int)gridDim.x : -1;
      wide_problem<T, D>(bp, tmax, uh_l, smem, tid, wave, uh_n, pre_lo);
      __syncthreads();                   // the tail's LDS reads are done before the next prologue's DMA lands
    }
  } else {
    wide_problem<T, D>(bp, tmax, blockIdx.x, smem, tid, wave, -1, pre_lo);
  }
}

}  // namespace hstu
