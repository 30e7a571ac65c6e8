// HSTU attention backward, "folded" schedule for short sequences (the metric shape): one workgroup
// of 8 waves per (user, head), the whole sequence (<= 7 tiles of 32 rows) in ONE key block.
//
// Same math, fragments and LDS tiles as hstu_attn_bwd.cuh (owner wave of a key tile keeps dK/dV in
// registers, dS' goes through LDS, dQ is a GEMM over the key tiles).  What changes is the schedule.
// The causal triangle gives key tile w one (query tile, key tile) pair per query tile >= w, so in the
// plain schedule (one query tile per step) the owner of key tile 0 works in every step while the
// others idle: 7 steps for 28 pairs on 8 waves.  Here every step takes TWO query tiles, one from
// each end of the sequence:
//     step k:   side A = query tile a = nt-1-k  (needs key tiles 0..a   -> a+1 owner waves 0..a)
//               side B = query tile b = k       (needs key tiles 0..b   -> b+1 owner waves 7-b..7)
// a + b = nt-1 <= 6, so the two sides always fit the 8 waves with all of them busy, and the loop
// has ceil(nt/2) steps instead of nt.  A wave first owns key tile w on side A (until the diagonal
// step a == w, after which that tile's dK/dV are final unless side B also reaches it: it is stored
// on the spot), then may own key tile 7-w on side B.  The side-B partial sums of key tiles
// 0..floor(nt/2)-1 are handed to their side-A owners through LDS after the loop.
// dQ of both query tiles of a step is complete within the step (all their key tiles are visited in
// it): a 16x16x32-MFMA GEMM in which every wave owns 16 feature columns of both tiles (balanced for any
// split between the sides), straight to HBM in the I/O dtype.
//
// LDS (16-bit I/O, DQK = DV = 128): 7 K/V tile pairs (112 KiB) + 2 Q/dO stages (32 KiB) + 8 dS'
// tiles (16 KiB, unpadded, 8-byte chunks XOR-swizzled) = exactly 160 KiB.  All global -> LDS traffic
// is LDS-DMA; the Q/dO tiles of step k+1 are fetched while the dQ GEMM of step k runs.
// Requires: no contextual rows (pair (i, w) active only for w <= i), head dims equal to the
// instantiated ones (LDS-DMA cannot zero-fill), 16-bit I/O, max_seq_len <= 224.
#pragma once
#include "hstu_attn_bwd.cuh"

#ifndef FOLD_SDP_AHEAD
#define FOLD_SDP_AHEAD 3   // MFMAs whose LDS operands are requested ahead in the S / dP and dV / dK streams
#endif
#ifndef FOLD_DQ32_AHEAD
#define FOLD_DQ32_AHEAD 2  // key tiles whose fragments are requested ahead of the 32x32x16 dQ chain's MFMAs
#endif
#ifndef FOLD_ABLATE
#define FOLD_ABLATE 0      // timing experiments only (WRONG results; tools/ab_bwd.py): 1 no dQ stores, 2 no dk/dv stores, 4 no stage DMA after
#endif                     // step 0, 8 no tail, 16 no K/V DMA, 32 no dQ GEMM, 64 no pairs, 128 / 256 every problem aliases one of the first
                           // 256 / 32 (Infinity-Cache / L2 resident data)
// The compile-time experiments of rounds 2-5 that lost (conflict-free parks, copy-out split, L2 touch, counted vmcnt, wave priorities,
// staggered starts, static dQ slots ...) are out of this file: docs/experiments/r05_hstu_attn_bwd_fold_with_switches.cuh.txt holds the
// last version that carried them, docs/EXPERIMENTS.md what each measured.

namespace hstu {

template <typename T, int DQK, int DV>
struct FoldCfg {
  using B = BwdCfg<T, DQK, DV>;
  static constexpr int DSB = 32 * 64;                    // [32 keys][32 q] 16-bit tile, 64-byte rows
  static constexpr int kMaxTiles = kBwdWaves - 1;
  // all kMaxTiles K/V slots are always allocated (the dQ GEMM reads every slot, used or not)
  static constexpr int smem_bytes() { return (kMaxTiles + 2) * B::PAIR + kBwdWaves * DSB; }
};

// LDS-DMA of one [32][D] tile (as tile_dma in hstu_attn_fwd.cuh), issued through inline asm.  Why asm: for the
// builtin, hipcc's waitcnt pass assumes that ANY later LDS read may alias the tile in flight and puts an
// s_waitcnt vmcnt(0) in front of it -- here that is the first read of the dQ GEMM, which only touches K tiles and
// dS' buffers, so the whole HBM latency of the next step's Q/dO tiles (the very thing the DMA is issued early to
// hide) would be waited out on the spot.  The asm form is invisible to that pass; the kernel orders its consumers
// by hand: s_waitcnt vmcnt(0) + barrier at the top of every step, before the first read of a DMA'd tile.
// (The compiler's own vmcnt bookkeeping stays safe: unknown extra operations in flight only make a counted wait
// stricter, memory operations complete in order.)  M0 carries the LDS base of the 1 KiB chunk.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
HSTU_DEV void dma16_asm(const char* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" HSTU_DMA_POLICY ::"v"(g), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop

// Source address of a lane's 16 bytes: (scalar tile base) + 32-bit offset = row x stride + swizzled unit x 16 -- one min, one
// 24-bit multiply-add per chunk and the 64-bit base from SGPRs, instead of a 64-bit multiply-add chain of ~20 VALU
// instructions (hstu_attn_fwd.cuh, tile_dma_fast).  `fast` (workgroup-uniform): strides < 16 MiB and the user's rows within
// 4 GiB of the base; otherwise 64-bit addresses.
HSTU_DEV void dma16_saddr(uint32_t off, const char* base, uint32_t lds_base) { dma16_saddr_asm(off, base, lds_base); }

template <typename T, int D>
HSTU_DEV void fold_tile_dma(char* tile, const char* base, int64_t row_stride_bytes, int row0, int len, int wave, int lane, bool fast) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int NCH = 32 * UPR / 64;   // 1 KiB chunks per tile
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tile);
  for (int c = wave; c < NCH; c += kBwdWaves) {
    const int pidx = c * 64 + lane;
    const int row = pidx / UPR, slot = pidx % UPR;
    const int unit = slot ^ swz<UPR>(row);
    const int grow = min(row0 + row, len - 1);
    if (fast) dma16_saddr(__umul24((uint32_t)grow, (uint32_t)row_stride_bytes) + unit * 16, base, lds0 + c * 1024);
    else dma16_asm(base + (int64_t)grow * row_stride_bytes + unit * 16, lds0 + c * 1024);
  }
}

// byte offset of the 8-byte chunk `chunk` (4 query columns) of key row `row` in a dS' tile
// The XOR term (row >> 1) & 7 makes all three users conflict-free: the owners' ds_write_b64 (16 rows, one chunk
// column: 16 distinct 8-byte positions of the 128-byte store window), and the 16x16x32 transposed reads of the
// dQ GEMM (a 32-lane group reads rows 8g..8g+3 of g = 0,1: four 64-byte quarters x two different 32-byte halves).
HSTU_DEV int fold_ds_off(int row, int chunk) { return (row << 6) + ((chunk ^ ((row >> 1) & 7)) << 3); }

// Research-path bias inside the folded schedule (hstu_attn_bwd_fold_bias_kernel): the per-workgroup state the pairs need.
// FoldNoBias compiles every use away.
struct FoldNoBias {
  static constexpr bool on = false;
};
struct FoldBias {
  static constexpr bool on = true;
  BiasCtx bc;          // staged tables + this user's timestamps (LDS)
  float* hpos;         // LDS histograms of dS': position bins, then time buckets x ts_copies
  float* hts;
  char* bcache;        // one byte per element of the causal triangle: the time bucket (user-invariant across heads)
  bool cached;         // the bucket bytes of this user are in place (heads after the first)
  TsRun ts_run;        // running sum of the current time bucket (hstu_common.cuh)
};

// One (query tile i0, key tile k0) pair on the owner wave: S, dP, P', dS', dV += , dK +=, publish dS'.
// PUBLISH = false (the long-sequence dK / dV kernel, hstu_attn_bwd_long.cuh: dQ comes from a kernel of its own): dS' stays in registers.
// CTXM (that kernel's instantiation for contextual_seq_len > 0): partly valid tiles take the general predicate, by compares.
template <typename T, int DQK, int DV, typename BX = FoldNoBias, bool PUBLISH = true, bool CTXM = false>
HSTU_DEV void fold_pair_x(const HstuAttnParams& p, const MaskCtx& mc, const char* __restrict__ Kw, const char* __restrict__ Vw,
                        const char* __restrict__ Qs, const char* __restrict__ dOs, char* __restrict__ myds, int i0, int k0, f32x16 (&dk_acc)[DQK / 32],
                        f32x16 (&dv_acc)[DV / 32], int lane, int dmvm, BX& bx HSTU_TRACE_ARG) {
  using C = BwdCfg<T, DQK, DV>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int n32 = lane & 31, hf = lane >> 5;
  const int len = mc.len;
  const int key = k0 + n32;
  const bool key_ok = key < len;
  f32x16 s, dp;
  // C layout of S / dP: column n32 = key, register r = query row (r&3) + 8 (r>>2) + 4 hf.  Query rows >= len hold a
  // clamped copy of a real row (LDS-DMA cannot zero-fill), so only tiles entirely inside the sequence need no mask.
  Frag pb[2], dsb[2];
  int mode;   // wave-uniform: 0 = no mask needed, 1 = plain causal, 2 = general mask algebra
  // Target rows only (no window, no contextual rows: what DLRM-v3 runs, modules/dlrm_hstu.py:207-214): a query tile whose rows all lie
  // in front of the first target sees plain causal masks -- its keys are <= its rows -- and takes the plain path (two compares and the
  // lane-constant patterns instead of the general tile predicate's ~100 scalar instructions and 16 x 8 vector instructions of mask
  // bits); of a 7-tile user with <= 20 targets that is every pair but the last one or two query tiles'.
  const bool plain = mc.simple || (HSTU_TARGETS_PLAIN && mc.has_targets && mc.win == 0 && mc.ctx == 0 && i0 + 32 <= min(len, mc.max_id));
  if (plain) mode = (k0 < i0 && i0 + 32 <= len) ? 0 : 1;    // strictly below the diagonal and all rows real
  else mode = (i0 + 32 <= len && mc.pair_fully_valid(i0, 32, k0, 32)) ? 0 : 2;
  const int key_id = mc.id_of(key);
  const int key_bits = key_ok ? -1 : 0;
  // The mask of a pair is 16 bits per lane (bit r = register r survives) and it is applied by
  // STARTING the S accumulator of a masked element at -1e30 instead of 0: then x = alpha S is hugely negative,
  // exp2(-x log2 e) = +inf, sigmoid = 1/inf = 0 exactly, and P' = x * 0 = -0, dS' = dP * 0 * (x + 1) = +-0 -- the
  // element-wise block needs no mask code at all, is the same for every pair and is ONE basic block (which is what
  // lets the scheduler weave it into the MFMAs).  Two instructions per element (bit-field extract, AND) in place of
  // the accumulator's v_mov 0.  Needs 1e-28 < |alpha| < 3e8 (alpha = 0 zeroes dq, dk and P' by itself); the
  // dispatcher sends anything else to the general kernel.
  // Plain causal: the pattern depends on the pair only through two wave-uniform facts -- is this the diagonal tile
  // (keep key <= query) and is it the sequence's last, partial query tile (keep query < len; every key of an earlier
  // tile is then < len too) -- both patterns live in one lane-constant register of the kernel (dmvm).  Any other
  // mask: the bits are collected from the general predicate.
  auto mask_init = [&]() {
    int km = -1;
    if (mode == 1) km = ((k0 == i0) ? dmvm : -1) & ((i0 + 32 > len) ? (dmvm >> 16) : -1);
    if (mode == 2) {
      km = 0;
      bool ctx_done = false;
      if constexpr (CTXM) {           // (the long-sequence dK / dV kernel also takes contextual rows; the folded kernels never see them)
        if (mc.ctx > 0) {             // wave-uniform: the general predicate by compares
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int qi = i0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            km |= (((qi < len) & key_ok & mc.valid_ids(qi, key, mc.id_of(qi), key_id)) ? 1 : 0) << r;
          }
          ctx_done = true;
        }
      }
      if (!ctx_done) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qi = i0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          km |= (mc.keep_bits_noctx(qi, key, key_id) & key_bits & 1) << r;
        }
      }
    }
    const unsigned nk = ~(unsigned)km;
    const unsigned neg = __builtin_bit_cast(unsigned, p.alpha < 0.f ? 1e30f : -1e30f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = ((int)(nk << (31 - r))) >> 31;        // all ones iff masked
      s[r] = __builtin_bit_cast(float, (unsigned)m & neg);
    }
  };
  // Pairs that need no mask (strictly below the diagonal, all rows real: 21 of a 7-tile problem's 28) start their S chain from the
  // MFMA's literal-zero C operand instead of the 3-instructions-per-element mask set-up (with the bias too: x = alpha S + b of an
  // unmasked element needs no start value).
#pragma unroll
  for (int r = 0; r < 16; ++r) dp[r] = 0.f;
  // S and dP as ONE stream of 16 MFMAs alternating between the two accumulators (no back-to-back dependency), with
  // the LDS reads of item m + AHEAD issued before the MFMA of item m.  The order is pinned with scheduling
  // barriers: left alone, hipcc emits read, read, wait, MFMA per item into the same registers, i.e. one full LDS
  // latency per MFMA, and the pair is latency bound whoever shares the SIMD.
  static_assert(DQK == DV, "interleaved S / dP stream");
  {
    constexpr int NM = 2 * C::KGQ, AHEAD = FOLD_SDP_AHEAD;
    Frag fa[AHEAD + 1], fb[AHEAD + 1];
    auto load_item = [&](int m, Frag& a, Frag& bb) {
      const int e0 = hf * (DQK / 2) + (m >> 1) * 8;
      a = lds_row_frag<T, C::UPR_K>((m & 1) ? dOs : Qs, n32, e0);
      bb = lds_row_frag<T, C::UPR_K>((m & 1) ? Vw : Kw, n32, e0);
    };
#pragma unroll
    for (int m = 0; m < AHEAD; ++m) load_item(m, fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (m + AHEAD < NM) load_item(m + AHEAD, fa[(m + AHEAD) % (AHEAD + 1)], fb[(m + AHEAD) % (AHEAD + 1)]);
      if (m & 1) dp = E::mma(fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)], dp);
      else if (m == 0) {
        // (wave-uniform) no mask: the chain starts from the MFMA's literal-zero C operand; otherwise from the mask pattern
        if (mode == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          s = E::mma(fa[0], fb[0], z);
        } else {
          mask_init();
          s = E::mma(fa[0], fb[0], s);
        }
      } else s = E::mma(fa[m % (AHEAD + 1)], fb[m % (AHEAD + 1)], s);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  HSTU_MARK(11);
  int t_k32 = 0;
  if constexpr (BX::on) {
    if (bx.bc.small) t_k32 = bx.bc.t32_at(key);
  }
  // (bias: dS' in fp32, the buckets and the first position bin of both halves stay live for the histograms, which run
  // AFTER the dV / dK stream: LDS operations complete in order, and a ds_add in front of the stream's fragment reads
  // delays every one of them by the atomic's latency)
  float ds_keep[BX::on ? 16 : 1];
  int bkt_keep[BX::on ? 16 : 1];
  int pbase_keep[2] = {0, 0};
  auto elem = [&](const int h8) {
    float pv[8], dsv[8];
    float xb[8];
    int bkt[8];
    int pbase = 0;       // position bin of the half's first element; row j + 8 gg of the half: pbase - j - 8 gg
    if constexpr (BX::on) {
      // bias term of the 8 elements of this half (hstu_attn_bwd.cuh, BIAS): time buckets from the user's byte matrix
      // (heads after the first) or computed and left there (first head)
      const BiasCtx& bc = bx.bc;
      const int qt = i0 >> 5, kt = k0 >> 5;
      char* const bslot = bx.bcache + ((((qt * (qt + 1)) >> 1) + kt) * 2 + h8) * 512 + 8 * lane;
      if (bx.cached) {
        const u32x2 w = *LDS_PTR(const u32x2, bslot);
#pragma unroll
        for (int j = 0; j < 8; ++j) bkt[j] = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
      } else {
        if (bc.small) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const auto t4 = bc.t32x4_next(i0 + 8 * (2 * h8 + g) + 4 * hf);
#pragma unroll
            for (int j = 0; j < 4; ++j) bkt[4 * g + j] = bc.bucket32(t4[j], t_k32);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            bkt[j] = bc.bucket(bc.ts_at(i0 + (r & 3) + 8 * (r >> 2) + 4 * hf + 1), bc.ts_at(key));
          }
        }
        u32x2 w = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j >> 2] |= (unsigned)bkt[j] << (8 * (j & 3));
        *LDS_PTR(u32x2, bslot) = w;
      }
      pbase = bc.pos_index(i0 + 16 * h8 + 4 * hf, key);
#pragma unroll
      for (int j = 0; j < 8; ++j) xb[j] = (BIAS_ABLATE & 4) ? 0.f : bc.value(pbase - (j & 3) - 8 * (j >> 2), bkt[j]);
    }
    {   // two elements per VALU instruction where the ISA has a packed fp32 form (mul / add / fma): -1.6 % kernel time
      const f32x2 a2 = {p.alpha, p.alpha};
      const f32x2 c2 = {-1.44269504088896340736f * p.alpha, -1.44269504088896340736f * p.alpha};
      const f32x2 nl2 = {-1.44269504088896340736f, -1.44269504088896340736f};
      const f32x2 one2 = {1.f, 1.f};
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const int r = 8 * h8 + j;
        const f32x2 sv = {s[r], s[r + 1]}, dpv = {dp[r], dp[r + 1]};
        f32x2 x = sv * a2, t = sv * c2;
        if constexpr (BX::on) {
          const f32x2 bv = {xb[j], xb[j + 1]};
          x = x + bv;
          t = x * nl2;
        }
        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        const f32x2 dn = e + one2;
        const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
        const f32x2 pr = x * sg;
        const f32x2 dsr = dpv * (pr * (one2 - sg) + sg);     // sg (1 + x (1 - sg)) = sg + P' (1 - sg)
        pv[j] = pr[0]; pv[j + 1] = pr[1];
        dsv[j] = dsr[0]; dsv[j + 1] = dsr[1];
      }
    }
    if constexpr (BX::on) {
      pbase_keep[h8] = pbase;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ds_keep[8 * h8 + j] = dsv[j]; bkt_keep[8 * h8 + j] = bkt[j]; }
    }
    pb[h8] = E::pack8(pv);
    dsb[h8] = E::pack8(dsv);
  };
  elem(0);
  elem(1);
  HSTU_MARK(12);
  // dV_w^T[dv][key] += dO_i^T[dv][q] P'[q][key]   and   dK_w^T[d][key] += Q_i^T[d][q] dS'[q][key]:
  // one stream of 16 MFMAs alternating between the dV and dK accumulators, A fragments (transposed LDS reads of
  // the dO / Q tile) requested AHEAD items before their MFMA, order pinned as above
  {
    constexpr int NM = 2 * 2 * C::DBQ, AHEAD = FOLD_SDP_AHEAD;   // (dV | dK) x d block x k half
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
      if (m & 1) dk_acc[d] = E::mma(fa[m % (AHEAD + 1)], dsb[ks], dk_acc[d]);
      else dv_acc[d] = E::mma(fa[m % (AHEAD + 1)], pb[ks], dv_acc[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // publish dS' as [key = n32][q]: this lane holds q = 4 hf + 8 rq + (0..3) = chunk hf + 2 rq
#pragma unroll
  for (int rq = 0; PUBLISH && rq < 4; ++rq) {
    const u32x4 w = __builtin_bit_cast(u32x4, dsb[rq >> 1].v);
    u32x2 v2 = {w[2 * (rq & 1)], w[2 * (rq & 1) + 1]};
    *LDS_PTR(u32x2, myds + fold_ds_off(n32, hf + 2 * rq)) = v2;
  }
  if constexpr (BX::on) {
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
      const float* dsv_ = ds_keep + 8 * h8;
      const int* bkt_ = bkt_keep + 8 * h8;
      const int pbase_ = pbase_keep[h8];
      // d bias = dS' (masked elements carry exact zeros): position histogram (4 consecutive rows x 4 consecutive keys
      // share a diagonal: three DPP row shifts, one atomic), time histogram (running sum per lane), as hstu_attn_bwd.cuh
      const int p16 = lane & 15;
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        float t = dsv_[4 * gg + 3];
#pragma unroll
        for (int j = 2; j >= 0; --j)
          t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x101, 0xf, 0xf, true)) + dsv_[4 * gg + j];
        if (!(BIAS_ABLATE & 1) && t != 0.f) atomicAdd(bx.hpos + pbase_ - 8 * gg, t);
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (!(BIAS_ABLATE & 1) && p16 < j && dsv_[4 * gg + j] != 0.f) atomicAdd(bx.hpos + pbase_ - 8 * gg - j, dsv_[4 * gg + j]);
      }
      if (!(BIAS_ABLATE & 2) && bx.bc.lts) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bx.ts_run.add(bkt_[j], dsv_[j]);
      }
    }
  }
}

template <typename T, int DQK, int DV>
HSTU_DEV void fold_pair(const HstuAttnParams& p, const MaskCtx& mc, const char* __restrict__ Kw, const char* __restrict__ Vw,
                        const char* __restrict__ Qs, const char* __restrict__ dOs, char* __restrict__ myds, int i0, int k0, f32x16 (&dk_acc)[DQK / 32],
                        f32x16 (&dv_acc)[DV / 32], int lane, int dmvm HSTU_TRACE_ARG) {
  FoldNoBias nb;
  fold_pair_x<T, DQK, DV, FoldNoBias>(p, mc, Kw, Vw, Qs, dOs, myds, i0, k0, dk_acc, dv_acc, lane, dmvm, nb HSTU_TRACE_PASS);
}

// Finished dk / dv tiles leave the workgroup in two moves.  (1) PARK: the owner wave writes its transposed
// accumulators (column n32 = key, registers = features), scaled and rounded to the I/O dtype, as a swizzled
// row-major [32 keys][D] tile into LDS -- in place of the K (or V) tile of the same keys, which is dead by then.
// (2) COPY OUT: all 8 waves move parked tiles to global memory, one 16-byte unit per thread, 16 consecutive lanes
// per 256-byte row: one dwordx4 store per wave per tile.  Storing the accumulators directly takes D/4 dwordx2
// stores per lane, each touching 64 different rows, all issued by ONE wave: that is store-issue bound (some 4k
// cycles per tile pair on the wave the others are waiting for).
template <typename T, int D>
HSTU_DEV void fold_park_tile(const f32x16 (&acc)[D / 32], float scale, char* __restrict__ tile, int lane) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  const int n32 = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int d = 0; d < D / 32; ++d)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      u32x2 v = {Elem<T>::pk2(acc[d][4 * rq] * scale, acc[d][4 * rq + 1] * scale),
                 Elem<T>::pk2(acc[d][4 * rq + 2] * scale, acc[d][4 * rq + 3] * scale)};
      // (256-byte rows: a 2-way bank conflict per ds_write_b64 group; the conflict-free variant -- 8-byte halves flipped on rows with
      // bit 1 set -- measured 1.6 % SLOWER: docs/EXPERIMENTS.md A.4)
      *LDS_PTR(u32x2, tile + tile_off<UPR>(n32, 4 * d + rq) + 8 * hf) = v;
    }
}
template <typename T, int D, int NT = kBwdThreads>
HSTU_DEV void fold_copy_out(const char* __restrict__ tile, char* gtile, int64_t row_stride_bytes, int rows_valid, int tid) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  for (int u = tid; u < 32 * UPR; u += NT) {
    const int row = u / UPR, unit = u % UPR;
    const u32x4 v = *LDS_PTR(const u32x4, tile + tile_off<UPR>(row, unit));
    // (non-temporal: -0.2 .. -0.9 %; the same hint on the dQ stores +2.7 %: profiles/r04_fold_nt_stores.txt)
    if (row < rows_valid && (!(FOLD_ABLATE & 2) || row_stride_bytes == -12345)) gstore16_nt(gtile + row * row_stride_bytes + unit * 16, v);
  }
}

// lane-linear fp32 dump / reload-and-add of an accumulator set (hand-over of a partial sum between two waves)
template <int NB>
HSTU_DEV void fold_dump(const f32x16 (&acc)[NB], char* __restrict__ region, int lane) {
#pragma unroll
  for (int d = 0; d < NB; ++d)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 v4 = {acc[d][4 * c], acc[d][4 * c + 1], acc[d][4 * c + 2], acc[d][4 * c + 3]};
      *LDS_PTR(f32x4, region + ((d * 4 + c) * 64 + lane) * 16) = v4;
    }
}
template <int NB>
HSTU_DEV void fold_add(f32x16 (&acc)[NB], const char* __restrict__ region, int lane) {
#pragma unroll
  for (int d = 0; d < NB; ++d)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v4 = *LDS_PTR(const f32x4, region + ((d * 4 + c) * 64 + lane) * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[d][4 * c + e] += v4[e];
    }
}

// 8-element fragment of a 16x16x32 MFMA for a contraction ACROSS the rows of a row-major LDS tile:
// lane (i16 = lane & 15, g = lane >> 4) gets rows 8g..8g+7 of one 16-bit column; `off_lo` / `off_hi` are this
// lane's byte offsets for rows 8g + (i16 >> 2) and 8g + 4 + (i16 >> 2) (transpose reads, see lds_col_frag).
template <typename T>
HSTU_DEV typename Elem<T>::Frag tr_frag16(const char* base, int off_lo, int off_hi) {
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + off_lo));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + off_hi));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  typename Elem<T>::Frag f;
  f.v = __builtin_bit_cast(typename Elem<T>::vec8, ab);
  return f;
}

// dQ of the two query tiles of a step.  dQ^T[d][q] = sum over key tiles of K_t^T[d][key] dS'_t^T[key][q] with the
// 16x16x32 MFMA (one key tile = one contraction): wave w < DQK/16 owns the 16 feature columns [16 w, +16) of BOTH
// query tiles (2 x 2 accumulators of 16 q rows), so every wave runs (a+1) + (b+1) = nt + 1 tile contractions per step
// whatever the split between the sides.  The dS' tile of key tile t was published by its owner wave: wave t on
// side A, wave 7 - t on side B.  Fragments of the next contraction are in flight under the MFMAs of this one.
// NA / NB: compile-time slot counts of the two sides (key tiles 0..NA-1 of side A, 0..NB-1 of side B).  The generic
// instance (7, 4) covers any step by zeroing the dS' fragment of idle slots; the steps of the full-length schedules
// (7 and 6 tiles) get exact counts: 8 (7) contractions per step instead of 11.
template <typename T, int DQK, int DV, int NA, int NB>
HSTU_DEV void fold_dq_slots(char* dq_head, int64_t dq_rs, const MaskCtx& mc, const char* __restrict__ kv,
                            const char* __restrict__ dsbuf, int a, int bq, bool b_on, int wave,
                            float ds_scale, int lane HSTU_TRACE_ARG) {
  using C = BwdCfg<T, DQK, DV>;
  using F = FoldCfg<T, DQK, DV>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  // Wave w owns the 32 feature columns [32 (w & 3), +32) of query rows [16 (w >> 2), +16) of BOTH tiles: two
  // accumulators per side.  The A rows (features) of the two MFMAs are interleaved in groups of 4 -- MFMA h covers
  // features 8 m' + 4 h + r (m' = 0..3 = the C layout's lane group, r = 0..3 = its register) -- so that a lane ends
  // up with 8 CONSECUTIVE features of one query row: one 16-byte store per side instead of two scattered 8-byte
  // ones (scattered 8-byte stores are store-issue bound).  The interleave costs nothing: a transposed LDS read takes
  // its four 4-element pieces per row from arbitrary addresses.
  const int db = wave & 3, qb = wave >> 2;
  if (db >= DQK / 32) return;
  const int i16 = lane & 15, g = lane >> 4;
  const int row_lo = 8 * g + (i16 >> 2), row_hi = row_lo + 4;
  // Which 16 features an MFMA covers.  Interleaved in groups of 4 (MFMA h: features 8 m' + 4 h + r), a lane ends up with 8
  // consecutive features -- but then MFMA h's transposed K reads touch only half h of every 16-byte unit, the two 16-lane
  // groups of a 32-lane read pass land on the same 16 eight-byte slots of the bank window, and EVERY K read of the dQ GEMM is
  // a 2-way bank conflict (two thirds of the kernel's SQ_LDS_BANK_CONFLICT, profiles/r02_*).  With 256-byte rows (head dim
  // 128) MFMA h takes 16 CONTIGUOUS features instead (32 bytes per row: the pass covers all 256 bytes of the window, no
  // conflict), and the results of the two MFMAs are exchanged between lane groups g and g ^ 1 with v_permlane16_swap so that a
  // lane still stores 8 consecutive features (fold_dq_store).  Same contractions in the same order: bit-identical.
  constexpr bool kContig = C::UPR_K == 16;
  const int colK0 = kContig ? 32 * db + 4 * (i16 & 3) : 32 * db + 8 * (i16 & 3), colK1 = colK0 + (kContig ? 16 : 4);
  const int k0_lo = tile_off<C::UPR_K>(row_lo, colK0 >> 3) + ((colK0 & 7) << 1);
  const int k0_hi = tile_off<C::UPR_K>(row_hi, colK0 >> 3) + ((colK0 & 7) << 1);
  const int k1_lo = tile_off<C::UPR_K>(row_lo, colK1 >> 3) + ((colK1 & 7) << 1);
  const int k1_hi = tile_off<C::UPR_K>(row_hi, colK1 >> 3) + ((colK1 & 7) << 1);
  const int d_lo = fold_ds_off(row_lo, 4 * qb + (i16 & 3)), d_hi = fold_ds_off(row_hi, 4 * qb + (i16 & 3));
  // The key-tile loops are STATIC (7 slots for side A, 4 for side B -- b <= 3 -- fully unrolled, no branches): a
  // slot without work reads whatever its LDS slot holds and gets its dS' fragment zeroed (an idle K/V slot only ever holds
  // finite values: K rows, parked dk tiles, or the zeros the kernel starts it with).  Straight-line code is what lets hipcc keep the LDS reads of the
  // next slots in flight under the MFMAs of this one (counted lgkmcnt waits); a runtime work list makes it drain
  // lgkmcnt to 0 at every block boundary and serialises (LDS latency + MFMA) per key tile.
  // (Scalar work is not free either: without an attention window every tile on or below the diagonal is active,
  // so the per-tile predicate -- some 50 SALU instructions -- is only evaluated when there is a window.)
  unsigned on_a = (1u << (a + 1)) - 1u, on_b = b_on ? (1u << (bq + 1)) - 1u : 0u;
  if (mc.win != 0) {
    for (int t = 0; t <= a; ++t)
      if (!mc.pair_may_be_active(32 * a, 32, 32 * t, 32)) on_a &= ~(1u << t);
    if (b_on)
      for (int t = 0; t <= bq; ++t)
        if (!mc.pair_may_be_active(32 * bq, 32, 32 * t, 32)) on_b &= ~(1u << t);
  }
  HSTU_MARK(19);
  f32x4 acc[2][2];
#pragma unroll
  for (int sd = 0; sd < 2; ++sd)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[sd][h] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
#pragma unroll
    for (int t = 0; t < (sd ? NB : NA); ++t) {
      const bool on = ((sd ? on_b : on_a) >> t) & 1u;
      const char* Kt = kv + t * C::PAIR;                          // all 7 slots are always allocated
      const char* ds = dsbuf + (sd ? (kBwdWaves - 1 - t) : t) * F::DSB;
      const Frag fk0 = tr_frag16<T>(Kt, k0_lo, k0_hi), fk1 = tr_frag16<T>(Kt, k1_lo, k1_hi);
      Frag fd = tr_frag16<T>(ds, d_lo, d_hi);
      fd.v = __builtin_bit_cast(typename E::vec8, on ? __builtin_bit_cast(u32x4, fd.v) : zero4);
      acc[sd][0] = E::mma16(fk0, fd, acc[sd][0]);
      acc[sd][1] = E::mma16(fk1, fd, acc[sd][1]);
    }
  }
  // requested instruction order (hipcc would otherwise, short of registers, serialise read -> wait -> MFMA per
  // slot): the 6 transposed reads of slot s+1 are issued ahead of the MFMA pair of slot s
  {
    constexpr int NSLOT = NA + NB;
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      if (sl + 1 < NSLOT) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // zeroing of an idle slot's dS' fragment
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  }
  HSTU_MARK(16);
  // C layout of MFMA h: column i16 = query row, register r = feature 32 db + 8 g + 4 h + r (interleaved) or
  // 32 db + 16 h + 4 g + r (contiguous)
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    if (sd == 1 && !b_on) break;     // (wave-uniform: the lane exchange below runs with all lanes)
    uint32_t x0 = E::pk2(acc[sd][0][0] * ds_scale, acc[sd][0][1] * ds_scale), x1 = E::pk2(acc[sd][0][2] * ds_scale, acc[sd][0][3] * ds_scale);
    uint32_t y0 = E::pk2(acc[sd][1][0] * ds_scale, acc[sd][1][1] * ds_scale), y1 = E::pk2(acc[sd][1][2] * ds_scale, acc[sd][1][3] * ds_scale);
    int f0 = 8 * g;                  // first of the lane's 8 consecutive features
    if constexpr (kContig) {
      // MFMA 0 holds features 4 g + r, MFMA 1 features 16 + 4 g + r of the wave's 32.  v_permlane16_swap exchanges the odd lane
      // groups of its first operand with the even ones of its second: afterwards lane group g holds the 8 features from
      // 8 (2 (g & 1) + (g >> 1)) on -- x: the lower four, y: the upper four
      const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      x0 = s0[0]; y0 = s0[1]; x1 = s1[0]; y1 = s1[1];
      f0 = 8 * (2 * (g & 1) + (g >> 1));
    }
    const int qrow = 32 * (sd ? bq : a) + 16 * qb + i16;
    if (qrow < mc.len && (!(FOLD_ABLATE & 1) || dq_rs == -12345))
      gstore16(dq_head + qrow * dq_rs + (32 * db + f0) * C::EB, u32x4{x0, x1, y0, y1});
  }
}

template <typename T, int DQK, int DV>
HSTU_DEV void fold_dq_phase(char* dq_head, int64_t dq_rs, const MaskCtx& mc, const char* __restrict__ kv,
                            const char* __restrict__ dsbuf, int a, int bq, bool b_on, int wave,
                            float ds_scale, int lane HSTU_TRACE_ARG) {
#define HSTU_FOLD_DQ(NA_, NB_) return fold_dq_slots<T, DQK, DV, NA_, NB_>(dq_head, dq_rs, mc, kv, dsbuf, a, bq, b_on, wave, ds_scale, lane HSTU_TRACE_PASS)
  const int code = 8 * (a + 1) + (b_on ? bq + 1 : 0);   // wave-uniform
  switch (code) {
    case 8 * 7 + 1: HSTU_FOLD_DQ(7, 1);
    case 8 * 6 + 2: HSTU_FOLD_DQ(6, 2);
    case 8 * 5 + 3: HSTU_FOLD_DQ(5, 3);
    case 8 * 4 + 0: HSTU_FOLD_DQ(4, 0);
    default: HSTU_FOLD_DQ(7, 4);
  }
#undef HSTU_FOLD_DQ
}

// ---- round 4: the dQ GEMM of head dim 128 as 32x32x16 chains -------------------------------------------------------------------
// The 16x16x32 phase above gives every wave (32 features x 16 query rows) of BOTH query tiles: 8 key-tile slots per step on
// every wave, 6 transposed reads for 2 small MFMAs per slot, one slot of read-ahead -- 2 K cycles per step of pure LDS latency
// (removing the phase shortens the kernel by 17 %, `FOLD_ABLATE` 32).  Here wave (db = wave & 3, side = wave >> 2) owns the 32
// features of block db of ONE query tile (side 0: tile a, key tiles 0..a; side 1: tile b, key tiles 0..b): per key tile two
// 32x32x16 MFMAs (keys 0..15, 16..31) on 8 transposed reads -- half the reads per MAC, the fragments of tile t + 2 requested
// before the MFMAs of tile t (straight-line code per tile count).  The sides are unequal (a + 1 against b + 1 tiles) but the
// longer chain, 7 / 6 / 5 / 4 tiles over the four steps of a full-length problem, is shorter than the 8 slots everybody ran.
// Results differ from the 16x16x32 phase in summation order only.  Needs every tile on or below the diagonal published
// (no attention window: a pair the window rules out is skipped and leaves its dS' slot stale).
template <typename T, int D, int N>
HSTU_DEV f32x16 fold_dq_chain32(const char* __restrict__ kv, const char* __restrict__ ds0, int ds_step, int db, int lane) {
  using C = BwdCfg<T, D, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int hf = lane >> 5, i16 = lane & 15, g1 = (lane >> 4) & 1;
  const int ra = 8 * hf, rb = 16 + 8 * hf;
  // B fragments (dS'^T[key][q]): the lane supplies the address of key row r + (i16 >> 2), query chunk 4 g1 + (i16 & 3)
  const int chunk = 4 * g1 + (i16 & 3), rr = i16 >> 2;
  const int o00 = fold_ds_off(ra + rr, chunk), o01 = fold_ds_off(ra + 4 + rr, chunk);
  const int o10 = fold_ds_off(rb + rr, chunk), o11 = fold_ds_off(rb + 4 + rr, chunk);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int t = 0; t < N; ++t) {
    const char* Kt = kv + t * C::PAIR;
    const char* ds = ds0 + t * ds_step;
    const Frag a0 = lds_col_frag<T, C::UPR_K>(Kt, ra, ra + 4, 32 * db, lane);      // K^T[d][key]
    const Frag a1 = lds_col_frag<T, C::UPR_K>(Kt, rb, rb + 4, 32 * db, lane);
    const Frag b0 = tr_frag16<T>(ds, o00, o01);
    const Frag b1 = tr_frag16<T>(ds, o10, o11);
    acc = E::mma(a0, b0, acc);
    acc = E::mma(a1, b1, acc);
  }
  // requested order: the 8 transposed reads of tile t + AHEAD in front of the MFMA pair of tile t
  {
    constexpr int AH = FOLD_DQ32_AHEAD < N ? FOLD_DQ32_AHEAD : N;
    __builtin_amdgcn_sched_group_barrier(0x100, 8 * AH, 0);
#pragma unroll
    for (int t = 0; t < N; ++t) {
      if (t + AH < N) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  }
  return acc;
}

template <typename T, int D>
HSTU_DEV void fold_dq_phase32(char* dq_head, int64_t dq_rs, const MaskCtx& mc, const char* __restrict__ kv, const char* __restrict__ dsbuf,
                              int a, int bq, bool b_on, int wave, float ds_scale, int lane HSTU_TRACE_ARG) {
  using C = BwdCfg<T, D, D>;
  using F = FoldCfg<T, D, D>;
  using E = Elem<T>;
  static_assert(D == 128, "four feature blocks x two sides = eight waves");
  const int db = wave & 3, side = wave >> 2;
  if (side == 1 && !b_on) return;
  const int n = side ? bq + 1 : a + 1;                  // key tiles 0 .. n - 1
  const int q0 = 32 * (side ? bq : a);
  const char* ds0 = side ? dsbuf + (kBwdWaves - 1) * F::DSB : dsbuf;     // side A: slot t, side B: slot 7 - t
  const int ds_step = side ? -F::DSB : F::DSB;
  HSTU_MARK(19);
  f32x16 acc;
  switch (n) {   // wave-uniform
    case 1: acc = fold_dq_chain32<T, D, 1>(kv, ds0, ds_step, db, lane); break;
    case 2: acc = fold_dq_chain32<T, D, 2>(kv, ds0, ds_step, db, lane); break;
    case 3: acc = fold_dq_chain32<T, D, 3>(kv, ds0, ds_step, db, lane); break;
    case 4: acc = fold_dq_chain32<T, D, 4>(kv, ds0, ds_step, db, lane); break;
    case 5: acc = fold_dq_chain32<T, D, 5>(kv, ds0, ds_step, db, lane); break;
    case 6: acc = fold_dq_chain32<T, D, 6>(kv, ds0, ds_step, db, lane); break;
    default: acc = fold_dq_chain32<T, D, 7>(kv, ds0, ds_step, db, lane); break;
  }
  HSTU_MARK(16);
  // C layout: column n32 = query row, register r = feature (r & 3) + 8 (r >> 2) + 4 hf of the block.  Lanes n32 and n32 + 32 hold
  // the same row: v_permlane32_swap pairs their 4-feature groups into runs of 8 features -- lanes 0..31 store the features
  // [0, 8) and [16, 24) of the block, lanes 32..63 [8, 16) and [24, 32): two 16-byte stores per lane
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
      const auto sw = __builtin_amdgcn_permlane32_swap(g[jp][h], g[jp + 1][h], false, false);
      g[jp][h] = sw[0];
      g[jp + 1][h] = sw[1];
    }
  const int qrow = q0 + n32;
  if (qrow < mc.len && (!(FOLD_ABLATE & 1) || dq_rs == -12345)) {
    char* dst = dq_head + qrow * dq_rs + (32 * db + 8 * hf) * C::EB;
    gstore16(dst, u32x4{g[0][0], g[0][1], g[1][0], g[1][1]});
    gstore16(dst + 16 * C::EB, u32x4{g[2][0], g[2][1], g[3][0], g[3][1]});
  }
}


// ---- scalar state of the persistent kernels -------------------------------------------------------------------------------------
// One workgroup per CU walks the problems; everything it knows about a problem is wave-uniform and belongs in SGPRs.  Through round 5
// the kernels took their parameter block (44 fields) as by-value kernel arguments and hipcc loaded all of it at kernel entry and kept
// it for the kernel's life: 162 scalar values spilled into vector lanes, 764 v_readlane_b32 / 162 v_writelane_b32 in the d = 128
// kernel's code (1,159 / 281 with the bias), and every problem began with two VECTOR loads of its offsets and an s_waitcnt vmcnt(0)
// behind them -- a wait for every store and LDS-DMA request of the previous problem's tail.  Now:
//  * the parameter block is RE-READ from the kernel-argument segment (constant address space: s_load) once per problem, through a
//    pointer laundered in an empty asm statement so that the loads cannot be hoisted out of the problem loop: a field lives from its
//    load to its last use in the problem, not for the kernel's life; the tail re-reads the four fields it needs for the next
//    problem's K/V requests instead of keeping them across the step loop;
//  * a problem's offsets come through scalar loads (sload_index), are requested one problem AHEAD and handed from iteration to
//    iteration (FoldWork): no vector-memory wait between two problems;
//  * dq / dk / dv / q / dO are addressed from per-problem base pointers (2 SGPRs each) instead of pointer + offset + two strides.
typedef const __attribute__((address_space(4))) uint32_t* kargw_t;
#define HSTU_KARG(k, field) (((const __attribute__((address_space(4))) HstuAttnBwdParams*)(k))->field)

HSTU_DEV HstuAttnBwdParams reload_bwd_params(kargw_t k) {
  static_assert(sizeof(HstuAttnBwdParams) % 4 == 0, "word copy");
  HstuAttnBwdParams r;
  uint32_t* d = (uint32_t*)&r;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(HstuAttnBwdParams) / 4); ++i) d[i] = k[i];
  return r;
}

struct FoldWork {
  int64_t off0;   // first row of the user
  int len;        // its rows, clamped to 32 tmax (a longer user is a caller error: the reference's padded path would truncate it
                  // too; rows past 32 tmax are ignored, their gradient rows not written); <= 0: nothing to do
  int b, hd;      // user, head
};

HSTU_DEV FoldWork fold_work(const HstuAttnParams& p, int uh, int tmax) {
  FoldWork w;
  // (FOLD_ABLATE 128 / 256: every problem aliases one of the first 256 / 32: cache-resident data)
  w.b = user_of_slot_s(p, ((FOLD_ABLATE & 128) ? uh % 256 : (FOLD_ABLATE & 256) ? uh % 32 : uh) / p.heads);
  w.hd = uh % p.heads;
  w.off0 = sload_index(p.seq_offsets, w.b, p.offsets_dtype);
  w.len = min((int)(sload_index(p.seq_offsets, w.b + 1, p.offsets_dtype) - w.off0), 32 * tmax);
  return w;
}

// One (user, head) problem `cur` on the calling workgroup (all of its LDS).  `nxt` (len > 0): the problem this workgroup takes next --
// its K/V tiles of the slots that are free during this problem's tail are requested there; `pre_lo`: in: K/V tiles >= pre_lo of THIS
// problem were requested by the previous problem's tail; out: the same for the next one.  `kargs`: the kernel-argument segment.
template <typename T, int DQK, int DV, typename BX = FoldNoBias>
HSTU_DEV void fold_problem_x(const HstuAttnBwdParams& bp, kargw_t kargs, int tmax, const FoldWork& cur, const FoldWork& nxt, char* smem,
                             int tid, int lane, int wave, int& pre_lo, BX& bx) {
  using C = BwdCfg<T, DQK, DV>;
  using F = FoldCfg<T, DQK, DV>;
  static_assert(C::EB == 2, "the folded backward is built for 16-bit I/O");
  static_assert(DQK / 32 <= 4, "dQ GEMM: 32 feature columns x 16 query rows per wave");
  static_assert(DQK == DV, "hand-over regions assume equal K and V tile sizes");
  const HstuAttnParams& p = bp.fwd;
  const int64_t off0 = cur.off0;
  const int len = cur.len, hd = cur.hd;
  const int pre_in = pre_lo;
  pre_lo = F::kMaxTiles;
  if (len <= 0) return;
  const MaskCtx mc = make_mask_ctx<true>(p, cur.b, len);
  HSTU_TRACE_DECL(bp.workspace, bp.workspace != nullptr && cur.b * p.heads + hd == 4096);
  HSTU_MARK(1);

  const int nt = (len + 31) >> 5;        // tiles of this user (<= tmax <= 7)
  const int ns = (nt + 1) >> 1;          // steps
  char* const stageA = smem + F::kMaxTiles * C::PAIR;
  char* const stageB = stageA + C::PAIR;
  char* const dsbuf = stageB + C::PAIR;

  const char* qbase = (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
  const char* dobase = (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
  char* const dq_head = (char*)bp.dq + (off0 * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB;
  char* const dk_head = (char*)bp.dk + (off0 * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
  char* const dv_head = (char*)bp.dv + (off0 * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
  const int64_t dq_rs = bp.dq_row_stride * C::EB, dk_rs = bp.dk_row_stride * C::EB, dv_rs = bp.dv_row_stride * C::EB;
  const int64_t q_rs = p.q_row_stride * C::EB, k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB,
                do_rs = bp.do_row_stride * C::EB;

  // LDS-DMA source addresses as scalar base + 32-bit lane offset (3 instead of ~20 VALU per chunk; fold_tile_dma): strides < 16 MiB
  // and a user's rows within 4 GiB of its first (the same test covers the next problem's K/V rows requested from this one's tail)
  const int len_max = 32 * tmax;
  const bool dma_fast = q_rs < (1 << 24) && k_rs < (1 << 24) && v_rs < (1 << 24) && do_rs < (1 << 24) &&
                        (int64_t)len_max * q_rs < (1LL << 32) && (int64_t)len_max * k_rs < (1LL << 32) &&
                        (int64_t)len_max * v_rs < (1LL << 32) && (int64_t)len_max * do_rs < (1LL << 32);
  auto stage_dma = [&](int qa, int qb, bool b_on) {
    fold_tile_dma<T, DQK>(stageA, qbase, q_rs, 32 * qa, len, wave, lane, dma_fast);
    fold_tile_dma<T, DV>(stageA + C::KT, dobase, do_rs, 32 * qa, len, wave, lane, dma_fast);
    if (b_on) {
      fold_tile_dma<T, DQK>(stageB, qbase, q_rs, 32 * qb, len, wave, lane, dma_fast);
      fold_tile_dma<T, DV>(stageB + C::KT, dobase, do_rs, 32 * qb, len, wave, lane, dma_fast);
    }
  };

  // ---- prologue: the K/V tiles the previous tail has not requested and the first two query tiles, all by LDS-DMA
  {
    const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
    const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
    for (int t = 0; t < ((FOLD_ABLATE & 16) ? 0 : min(nt, pre_in)); ++t) {
      char* dst = smem + t * C::PAIR;
      fold_tile_dma<T, DQK>(dst, kbase, k_rs, 32 * t, len, wave, lane, dma_fast);
      fold_tile_dma<T, DV>(dst + C::KT, vbase, v_rs, 32 * t, len, wave, lane, dma_fast);
    }
  }
  stage_dma(nt - 1, 0, 0 < nt - 1);
  for (int i = tid; i < kBwdWaves * F::DSB / 16; i += kBwdThreads) *LDS_PTR(u32x4, dsbuf + 16 * i) = u32x4{0u, 0u, 0u, 0u};
  // K tiles of slots this user does not fill: the dQ GEMM reads every slot (times a zeroed dS' fragment), so they
  // must hold finite values
  for (int t = nt; t < F::kMaxTiles; ++t)
    for (int i = tid; i < C::KT / 16; i += kBwdThreads) *LDS_PTR(u32x4, smem + t * C::PAIR + 16 * i) = u32x4{0u, 0u, 0u, 0u};
  HSTU_MARK(2);

  f32x16 dk_acc[C::DBQ], dv_acc[C::DBV];
#pragma unroll
  for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk_acc[d][r] = 0.f;
#pragma unroll
  for (int d = 0; d < C::DBV; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv_acc[d][r] = 0.f;
  const float scale_v = attn_scale_of(p);
  const float ds_scale = scale_v * p.alpha;

  // lane-constant mask patterns of the plain-causal case (see fold_pair): low half = diagonal tile, bit r set iff
  // key n32 <= query row (r&3) + 8 (r>>2) + 4 hf; high half = last query tile, bit r set iff that row is < len
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

  for (int k = 0; k < ns; ++k) {
    const int a = nt - 1 - k, bq = k;
    const bool b_on = bq < a;
    // (leaving the previous step's stores in flight across the barrier -- a counted vmcnt -- measured 1-2 % SLOWER: EXPERIMENTS R5.3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // Q/dO tiles of this step (and, first time, K/V) landed; dS' of the last step consumed
    HSTU_MARK(10);
    if (k > 0 && wave == a + 1) {
      // owner of the previous step's diagonal tile: its K tile is dead now (every dQ GEMM of that step is done):
      // park dK in its place; dV was parked before the barrier, so the accumulators are free for a side-B tile
      int lane3 = lane;
      asm volatile("" : "+v"(lane3));
      fold_park_tile<T, DQK>(dk_acc, ds_scale, smem + wave * C::PAIR, lane3);
#pragma unroll
      for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk_acc[d][r] = 0.f;
    }
    // ---- phase 1: this wave's pair of the step
    int kt = -1, qt = 0;
    const char* st = stageA;
    if (wave <= a) { kt = wave; qt = a; }
    else if (b_on && kBwdWaves - 1 - wave <= bq) { kt = kBwdWaves - 1 - wave; qt = bq; st = stageB; }
    if (kt >= 0 && !(FOLD_ABLATE & 64) && (mc.win == 0 || mc.pair_may_be_active(32 * qt, 32, 32 * kt, 32))) {
      const char* Kw = smem + kt * C::PAIR;
      // (the lane id is laundered per phase: LDS offsets derived from it are then recomputed where they are used --
      // a few dozen VALU instructions -- instead of being hoisted out of the step loop, where some 60 of them,
      // alive across both phases next to the 128 accumulator registers, push the kernel into spilling)
      int lane1 = lane;
      asm volatile("" : "+v"(lane1));
      fold_pair_x<T, DQK, DV, BX>(p, mc, Kw, Kw + C::KT, st, st + C::KT, dsbuf + wave * F::DSB, 32 * qt, 32 * kt, dk_acc,
                                  dv_acc, lane1, dmvm, bx HSTU_TRACE_PASS);
    }
    HSTU_MARK(13);
    HSTU_MARK(14);
    __syncthreads();   // dS' of this step published; stage reads done
    HSTU_MARK(15);
    if (k + 1 < ns && !(FOLD_ABLATE & 4)) stage_dma(a - 1, bq + 1, bq + 1 < a - 1);
    // dk / dv of the previous step's diagonal key tile (parked in K/V slot a + 1): out, by all waves
    if (k > 0) {
      const int kt1 = a + 1;
      fold_copy_out<T, DQK>(smem + kt1 * C::PAIR, dk_head + (int64_t)(32 * kt1) * dk_rs, dk_rs, len - 32 * kt1, tid);
      fold_copy_out<T, DV>(smem + kt1 * C::PAIR + C::KT, dv_head + (int64_t)(32 * kt1) * dv_rs, dv_rs, len - 32 * kt1, tid);
    }
    HSTU_MARK(18);
    // ---- phase 2: dQ of the two query tiles
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    if (!(FOLD_ABLATE & 32)) {
      bool done32 = false;
      if constexpr (DQK == 128 && DV == 128 && !BX::on) {
        if (mc.win == 0) {       // 32x32x16 chains: needs every tile on or below the diagonal published
          fold_dq_phase32<T, DQK>(dq_head, dq_rs, mc, smem, dsbuf, a, bq, b_on, wave, ds_scale, lane2 HSTU_TRACE_PASS);
          done32 = true;
        }
      }
      if (!done32) fold_dq_phase<T, DQK, DV>(dq_head, dq_rs, mc, smem, dsbuf, a, bq, b_on, wave, ds_scale, lane2 HSTU_TRACE_PASS);
    }
    HSTU_MARK(17);
    if (kt == wave && wave == a) {
      // diagonal step of side A: no later query tile reaches key tile `wave`, its dK/dV are final (a >= nt/2 in
      // every step, so it is never one of the side-B tiles 0..nt/2-1).  dV is parked right away: V tiles are only
      // read by their owner's pairs and this owner has just done its last one.  The K tile may still be read by
      // other waves' dQ GEMM: dK follows after the next barrier.
      int lane3 = lane;
      asm volatile("" : "+v"(lane3));
      fold_park_tile<T, DV>(dv_acc, scale_v, smem + wave * C::PAIR + C::KT, lane3);
#pragma unroll
      for (int d = 0; d < C::DBV; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dv_acc[d][r] = 0.f;
    }
    HSTU_MARK(23);
  }
  HSTU_MARK(20);
  if (FOLD_ABLATE & 8) return;
  // ---- tail.  Key tiles 0..nb-1 have two partial sums: side A (wave t) and side B (wave 7 - t).  Each of the two
  // waves finishes HALF of the tile: the side-A owner hands its dV partial over and finishes dK, the side-B owner
  // hands its dK partial over and finishes dV.  Hand-over regions (fp32, lane-linear, one K/V slot's size each):
  // side A -> K/V slot t, side B -> stage A / stage B / dS' buffers for t = 0 / 1 / 2 (all dead by now).  The
  // finished halves are parked in the region the wave has just read, then everything parked goes out together.
  const int nb = nt >> 1;
  const int a_last = nt - ns;                      // diagonal tile of the last step: dV parked, dK still in registers
  const int bt = kBwdWaves - 1 - wave;
  const bool a_fin = wave < nb, b_fin = bt < nb;
  const int tb = a_fin ? wave : bt;
  char* const reg_a = smem + tb * C::PAIR;
  char* const reg_b = tb == 0 ? stageA : (tb == 1 ? stageB : dsbuf);
  __syncthreads();     // K/V tiles, stages and dS' buffers are dead from here on
  if (nxt.len > 0) {
    // K/V slots above the last step's diagonal tile are not touched by the tail: the next problem's tiles of those
    // slots stream in under it.  (k / v and their strides are re-read from the kernel arguments here instead of living in
    // scalar registers across the step loop.)
    kargw_t k2 = kargs;
    asm volatile("" : "+s"(k2));
    const int64_t krs3 = HSTU_KARG(k2, fwd.k_row_stride), vrs3 = HSTU_KARG(k2, fwd.v_row_stride);
    const char* kb3 = (const char*)HSTU_KARG(k2, fwd.k) + (nxt.off0 * krs3 + (int64_t)nxt.hd * HSTU_KARG(k2, fwd.k_head_stride)) * C::EB;
    const char* vb3 = (const char*)HSTU_KARG(k2, fwd.v) + (nxt.off0 * vrs3 + (int64_t)nxt.hd * HSTU_KARG(k2, fwd.v_head_stride)) * C::EB;
    const int nt3 = (nxt.len + 31) >> 5;
    for (int t = a_last + 1; t < nt3; ++t) {
      char* dst = smem + t * C::PAIR;
      fold_tile_dma<T, DQK>(dst, kb3, krs3 * C::EB, 32 * t, nxt.len, wave, lane, dma_fast);
      fold_tile_dma<T, DV>(dst + C::KT, vb3, vrs3 * C::EB, 32 * t, nxt.len, wave, lane, dma_fast);
    }
    pre_lo = a_last + 1;
  }
  int lane4 = lane;
  asm volatile("" : "+v"(lane4));
  if (wave == a_last) fold_park_tile<T, DQK>(dk_acc, ds_scale, smem + wave * C::PAIR, lane4);
  if (a_fin) fold_dump<C::DBV>(dv_acc, reg_a, lane4);
  if (b_fin) fold_dump<C::DBQ>(dk_acc, reg_b, lane4);
  __syncthreads();
  HSTU_MARK(22);
  if (a_fin) {
    fold_add<C::DBQ>(dk_acc, reg_b, lane4);
    fold_park_tile<T, DQK>(dk_acc, ds_scale, reg_b, lane4);
  }
  if (b_fin) {
    fold_add<C::DBV>(dv_acc, reg_a, lane4);
    fold_park_tile<T, DV>(dv_acc, scale_v, reg_a + C::KT, lane4);
  }
  __syncthreads();
  HSTU_MARK(24);
  fold_copy_out<T, DQK>(smem + a_last * C::PAIR, dk_head + (int64_t)(32 * a_last) * dk_rs, dk_rs, len - 32 * a_last, tid);
  fold_copy_out<T, DV>(smem + a_last * C::PAIR + C::KT, dv_head + (int64_t)(32 * a_last) * dv_rs, dv_rs, len - 32 * a_last, tid);
  for (int t = 0; t < nb; ++t) {
    const char* rb = t == 0 ? stageA : (t == 1 ? stageB : dsbuf);
    fold_copy_out<T, DQK>(rb, dk_head + (int64_t)(32 * t) * dk_rs, dk_rs, len - 32 * t, tid);
    fold_copy_out<T, DV>(smem + t * C::PAIR + C::KT, dv_head + (int64_t)(32 * t) * dv_rs, dv_rs, len - 32 * t, tid);
  }
  HSTU_MARK(21);
}

// One persistent workgroup per CU walks the problems blockIdx.x, blockIdx.x + gridDim.x, ... (no workgroup relaunch between two
// problems of a CU: the dispatch gap, the kernel-argument loads and the wave start-up are paid once) and requests the next problem's
// K/V tiles of the slots its own tail does not use.
template <typename T, int DQK, int DV>
__global__ __launch_bounds__(kBwdThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void hstu_attn_bwd_fold_kernel(const HstuAttnBwdParams bp_arg, int tmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const kargw_t kargs = (kargw_t)__builtin_amdgcn_kernarg_segment_ptr();     // (bp_arg is the first argument: offset 0)
  const int total = HSTU_KARG(kargs, fwd.batch) * HSTU_KARG(kargs, fwd.heads);
  int pre_lo = 7;
  FoldNoBias nb;
  FoldWork cur;
  {
    const HstuAttnBwdParams bp0 = reload_bwd_params(kargs);
    cur = fold_work(bp0.fwd, blockIdx.x, tmax);
  }
  for (int uh = blockIdx.x; uh < total; uh += gridDim.x) {
    kargw_t kl = kargs;
    asm volatile("" : "+s"(kl));       // nothing of problem i+1 is hoisted into problem i, no argument outlives its problem
    const HstuAttnBwdParams bp = reload_bwd_params(kl);
    FoldWork nxt;
    nxt.len = 0; nxt.off0 = 0; nxt.b = 0; nxt.hd = 0;
    if (uh + (int)gridDim.x < total) nxt = fold_work(bp.fwd, uh + (int)gridDim.x, tmax);
    fold_problem_x<T, DQK, DV, FoldNoBias>(bp, kl, tmax, cur, nxt, smem, tid, lane, wave, pre_lo, nb);
    cur = nxt;
    __syncthreads();                   // the tail's LDS reads are done before the next prologue's DMA lands
  }
}

// Research-path backward (relative position / time bias, hstu_attn_bwd.cuh BIAS) on the folded schedule: one persistent
// workgroup per CU walks USERS, and for each user its heads.  Per user: tables and timestamps staged once, the time-bucket
// matrix computed by the first head and kept as bytes; per workgroup: ONE pair of histograms for everything it
// processes, flushed to its row of `bias_partial` at the end (gridDim rows for the reduce instead of batch x heads).
// LDS behind the folded kernel's own region: [pos histogram 2N][time histogram (nb+1) x ts_copies][tables][bucket bytes].
template <typename T, int D>
__global__ __launch_bounds__(kBwdThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void hstu_attn_bwd_fold_bias_kernel(
    const HstuAttnBwdParams bp_arg, int tmax, float* bias_partial, int ts_copies, int hist_bytes, int table_bytes, int* next_user) {
  using F = FoldCfg<T, D, D>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const kargw_t kargs = (kargw_t)__builtin_amdgcn_kernarg_segment_ptr();
  const int batch = HSTU_KARG(kargs, fwd.batch), heads = HSTU_KARG(kargs, fwd.heads);
  const int max_seq_len = HSTU_KARG(kargs, fwd.max_seq_len), num_buckets = HSTU_KARG(kargs, fwd.num_buckets);
  FoldBias bx;
  bx.hpos = (float*)(smem + F::smem_bytes());
  bx.hts = bx.hpos + 2 * max_seq_len;
  char* const tables = (char*)bx.hpos + hist_bytes;
  bx.bcache = tables + table_bytes;
  bx.ts_run.init(bx.hts, ts_copies);
  bx.cached = false;
  const int hist_floats = 2 * max_seq_len + (num_buckets + 1) * ts_copies;
  for (int i = tid; i < hist_floats; i += kBwdThreads) bx.hpos[i] = 0.f;
  int pre_lo = 7;
  FoldWork cur;
  {
    const HstuAttnBwdParams bp0 = reload_bwd_params(kargs);
    cur = fold_work(bp0.fwd, (int)blockIdx.x * heads, tmax);
  }
  // Users are handed out DYNAMICALLY (round 6): workgroup b starts with user b of the launch order and takes its next one from a
  // counter (`next_user`, zeroed by the launcher; thread 0 adds, the value crosses the workgroup in the unused last word of the position
  // histogram, between the two barriers every user has anyway).  Walking users b, b + grid, ... left the launch waiting for whichever
  // workgroup had drawn the longest users: ML-20M lengths (uniform in 1 .. 211), 8192 users: 2.78 ms, the SAME batch with its users
  // sorted by length 2.38 (tools/c2_length_order_probe.py).
  volatile int* const slot = (volatile int*)(bx.hpos + 2 * max_seq_len - 1);
  // (the ticket is drawn one user AHEAD: thread 0 publishes the one it drew during the previous user and draws the next right away -- the
  // atomic's round trip passes under a whole user instead of in front of the barrier every wave waits at: C2 backward -0.6 %.  Requesting
  // the next user's timestamps ahead as well -- registers across the head loop -- gave that back: not adopted)
  int ticket = 0;
  if (tid == 0) ticket = (int)gridDim.x + atomicAdd(next_user, 1);
  for (int u = blockIdx.x; u < batch;) {
    if (tid == 0) {
      *slot = ticket;
      ticket = (int)gridDim.x + atomicAdd(next_user, 1);
    }
    __syncthreads();                       // the previous user's pairs have read their last table entry
    {
      kargw_t kl = kargs;
      asm volatile("" : "+s"(kl));
      const HstuAttnBwdParams bp = reload_bwd_params(kl);
      bx.bc = stage_bias_tables(bp.fwd, cur.b, tables, tid, kBwdThreads, /*user_only=*/u != (int)blockIdx.x);   // (the weight tables do not depend on the user: staged with the workgroup's first one)
    }
    __syncthreads();
    bx.bc.finish(kBwdWaves);
    const int u_next = __builtin_amdgcn_readfirstlane(*slot);      // (rewritten only behind the barriers of the head loop below)
    for (int hd = 0; hd < heads; ++hd) {
      kargw_t kl = kargs;
      asm volatile("" : "+s"(kl));
      const HstuAttnBwdParams bp = reload_bwd_params(kl);
      const int uh = u * heads + hd;
      const int uh_n = hd + 1 < heads ? uh + 1 : (u_next < batch ? u_next * heads : -1);
      FoldWork nxt;
      nxt.len = 0; nxt.off0 = 0; nxt.b = 0; nxt.hd = 0;
      if (hd + 1 < heads) { nxt = cur; nxt.hd = hd + 1; }            // the same user's next head: no load
      else if (uh_n >= 0) nxt = fold_work(bp.fwd, uh_n, tmax);
      bx.cached = hd > 0;
      fold_problem_x<T, D, D, FoldBias>(bp, kl, tmax, cur, nxt, smem, tid, lane, wave, pre_lo, bx);
      cur = nxt;
      __syncthreads();
    }
    u = u_next;
  }
  bx.ts_run.flush();
  __syncthreads();
  const HstuAttnBwdParams bp = reload_bwd_params(kargs);
  const float scale_v = attn_scale_of(bp.fwd);
  float* row = bias_partial + (int64_t)blockIdx.x * (2 * max_seq_len + num_buckets);
  const int npos = 2 * max_seq_len - 1;
  for (int i = tid; i < 2 * max_seq_len + num_buckets; i += kBwdThreads) {
    float v;
    if (i < npos) {
      v = bx.hpos[i];
    } else {
      v = 0.f;
      const float* cp = bx.hts + (i - npos) * ts_copies;
      for (int c = 0; c < ts_copies; ++c) v += cp[c];
    }
    row[i] = v * scale_v;
  }
}

}  // namespace hstu
