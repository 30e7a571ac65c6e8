// HSTU attention backward for gfx950: one pass, dq/dk/dv from dout.
//
//   S = alpha Q K^T,  sg = sigmoid(S),  P = S sg scale M
//   dV = P^T dO,  dP = dO V^T,  dS = dP M scale sg (1 + S (1 - sg))
//   dQ = alpha dS K,  dK = alpha dS^T Q                       
//
// Replaces triton_hstu_attention_bwd (ops/triton/triton_hstu_attention.py:1849-1948,
// kernels :899-1764) and hstu::hstu_mha_bwd (ops/cpp/hstu_attention/flash_api.cpp:111-141).
//
// Mapping.  One workgroup of 8 waves owns one (user, head, block of NW*32 keys), NW<=8.
// The K/V rows of the block stay RESIDENT in LDS (row-major, swizzled); query/dO tiles
// of 32 rows stream through one LDS stage (next tile prefetched into registers).  For
// every query tile:
//   phase 1  wave w < NW (owner of key tile w):
//            S  = Q_i K_w^T, dP = dO_i V_w^T            (keys on the lane axis)
//            P, dS element-wise in registers; C layout == B layout of the next MFMAs
//            dV_w^T += dO_i^T P      dK_w^T += Q_i^T dS   (A through LDS transpose reads;
//            accumulators live in registers for the whole kernel: no atomics)
//            dS (already scaled by alpha) is published to LDS as [key][q]
//   phase 2  wave d < DQK/32: dQ_i^T[32d..32d+32) = sum over key tiles K_w^T dS_w^T
//            (a plain GEMM over all keys of the block: the cross-wave reduction of dQ
//            happens in the MFMA accumulator instead of in memory)
// When one block covers the user's whole sequence (the common case: L <= 32*NW) dQ is
// written once, in the I/O dtype.  Longer sequences use several key blocks and add
// their fp32 dQ partials into a zeroed workspace that a second tiny kernel converts
// (the CUDA reference does this for every shape: flash_common.cpp:806-816).
// q, k, v and dout are each read from HBM exactly once per (user, head) in the
// single-block case; dq, dk, dv are written once.
#pragma once
#include <type_traits>
#include "hstu_attn_fwd.cuh"

#ifndef BIAS_ABLATE
#define BIAS_ABLATE 0   // timing experiments only (wrong results): 1 no position histogram, 2 no time histogram, 4 bias value 0
#endif

namespace hstu {

constexpr int kBwdThreads = 512;
constexpr int kBwdWaves = 8;

template <typename T, int DQK, int DV>
struct BwdCfg {
  static constexpr int EB = Elem<T>::kBytes;
  static constexpr int EPU = 16 / EB;
  static constexpr int UPR_K = DQK * EB / 16;
  static constexpr int UPR_V = DV * EB / 16;
  static constexpr int KT = 32 * DQK * EB;
  static constexpr int VT = 32 * DV * EB;
  static constexpr int PAIR = KT + VT;            // K+V tile, also Q+dO stage
  static constexpr int DSROW = (EB == 2) ? 72 : 144;  // padded [key][32 q] row of the dS buffer
  static constexpr int DSBUF = 32 * DSROW;
  static constexpr int KGQ = DQK / 16;
  static constexpr int KGV = DV / 16;
  static constexpr int DBQ = DQK / 32;
  static constexpr int DBV = DV / 32;
  static constexpr int NQU = (32 * UPR_K + kBwdThreads - 1) / kBwdThreads;
  static constexpr int NOU = (32 * UPR_V + kBwdThreads - 1) / kBwdThreads;
  static constexpr int smem_bytes(int nw, int extra = 0) { return nw * PAIR + PAIR + 2 * nw * DSBUF + extra; }
  // at most 7 key tiles per block: the 8th wave never owns a tile, so every step has a
  // helper wave for the dQ GEMM of the previous step
  static constexpr int max_tiles(int lds_budget) {
    int nw = (lds_budget - PAIR) / (PAIR + 2 * DSBUF);
    return nw > kBwdWaves - 1 ? kBwdWaves - 1 : nw;
  }
};

// Column fragment from the (unswizzled, padded) dS buffer: slot j<4 -> buf[rowA+j][n32],
// slot j>=4 -> buf[rowB+j-4][n32].
template <typename T, int ROWB>
HSTU_DEV typename Elem<T>::Frag dsbuf_col_frag(const char* buf, int rowA, int rowB, int lane) {
  typename Elem<T>::Frag f;
  if constexpr (Elem<T>::kBytes == 2) {
    const int i16 = lane & 15;
    const int col = (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2);
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, buf + (rowA + (i16 >> 2)) * ROWB + col * 2));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, buf + (rowB + (i16 >> 2)) * ROWB + col * 2));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    f.v = __builtin_bit_cast(typename Elem<T>::vec8, ab);
  } else {
    const int col = lane & 31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.v[j] = *LDS_PTR(const float, buf + (rowA + j) * ROWB + col * 4);
      f.v[4 + j] = *LDS_PTR(const float, buf + (rowB + j) * ROWB + col * 4);
    }
  }
  return f;
}

// dQ_i^T blocks [32 db, +32) for db = db0, db0 + dstep, ... (NB at a time) of query tile i0:
// sum over the block's key tiles of K_w^T dS'_w^T, then scale and store (or atomically add to
// the fp32 workspace).  Executed by ONE wave; the dS' fragments are read once per key tile and
// shared by the NB output blocks, and NB independent MFMA chains hide each other's latency.
template <typename T, int DQK, int DV, int NB>
HSTU_DEV void bwd_dq_blocks(const HstuAttnBwdParams& bp, const MaskCtx& mc, const char* smem, const char* ds_base,
                            int nw, int kb0, int i0, int db0, int dstep, int64_t off0, int hd, float ds_scale,
                            float* dq_accum, char* dq_scratch, int lane, bool dq_plain = false) {
  using C = BwdCfg<T, DQK, DV>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const HstuAttnParams& p = bp.fwd;
  const int n32 = lane & 31, hf = lane >> 5;
  const int len = mc.len;
  f32x16 acc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int w2 = 0; w2 < nw; ++w2) {
    const int k0 = kb0 + 32 * w2;
    if (k0 >= len || !mc.pair_may_be_active(i0, 32, k0, 32)) continue;   // wave-uniform
    const char* Kt = smem + w2 * C::PAIR;
    const char* ds = ds_base + w2 * C::DSBUF;
    const int ra = 8 * hf, rb = 16 + 8 * hf;
    Frag b0 = dsbuf_col_frag<T, C::DSROW>(ds, ra, ra + 4, lane);             // dS'^T[key][q]
    Frag b1 = dsbuf_col_frag<T, C::DSROW>(ds, rb, rb + 4, lane);
    Frag a0[NB], a1[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int db = db0 + n * dstep;
      a0[n] = lds_col_frag<T, C::UPR_K>(Kt, ra, ra + 4, 32 * db, lane);      // K^T[d][key]
      a1[n] = lds_col_frag<T, C::UPR_K>(Kt, rb, rb + 4, 32 * db, lane);
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = E::mma(a0[n], b0, acc[n]);
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[n] = E::mma(a1[n], b1, acc[n]);
  }
  // C layout: column n32 = query row, register r = d within the 32-block
  const int qrow = i0 + n32;
  if (dq_accum == nullptr && qrow < len) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int db = db0 + n * dstep;
      {
        char* dqrow = (char*)bp.dq + ((off0 + qrow) * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = 32 * db + 8 * rq + 4 * hf;
          if (d0 < p.dqk)
            store4<T>(dqrow, d0, acc[n][4 * rq] * ds_scale, acc[n][4 * rq + 1] * ds_scale, acc[n][4 * rq + 2] * ds_scale,
                      acc[n][4 * rq + 3] * ds_scale);
        }
      }
    }
  }
  if (dq_accum != nullptr) {
    // several key blocks: fp32 partials are ADDED to the workspace.  In the C layout a lane holds one query row, so
    // one atomic instruction would touch 64 different cache lines (measured: ~1100 cycles per instruction, a 9x
    // cliff at the first sequence length that needs two key blocks).  Through a per-wave LDS tile instead: written
    // in the C layout, read back with a lane = one feature column, so an instruction adds two whole 128-byte rows.
    // (No predication: rows >= len / columns >= dqk add 0.0 to a clamped, valid address -- a masked atomic makes
    // hipcc reload spilled addresses and drain vmcnt in front of every single one.)
    const int hsel = lane >> 5, col = lane & 31;
    float* const base = dq_accum + ((off0 + i0) * p.heads + hd) * (int64_t)p.dqk;   // wave-uniform
    const int rstride = p.heads * p.dqk;
    const int rmax = len - 1 - i0;                                                    // last real row of the tile (>= 0)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int db = db0 + n * dstep;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 v4 = {acc[n][4 * rq] * ds_scale, acc[n][4 * rq + 1] * ds_scale, acc[n][4 * rq + 2] * ds_scale,
                    acc[n][4 * rq + 3] * ds_scale};
        *LDS_PTR(f32x4, dq_scratch + n32 * 128 + (((2 * rq + hf) ^ (n32 & 7)) << 4)) = v4;
      }
      const int d = 32 * db + col;
      const bool col_ok = d < p.dqk;
      const int dc = col_ok ? d : 0;
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int row = 2 * i + hsel;
        float v = *LDS_PTR(const float, dq_scratch + row * 128 + ((((col >> 2) ^ (row & 7)) << 4) | ((col & 3) << 2)));
        if (dq_plain) {
          // deterministic: this key block's slab -- every element is written by exactly one wave, once (plain, predicated store)
          if (col_ok && row <= rmax) base[row * rstride + dc] = v;
          continue;
        }
        v = (col_ok && row <= rmax) ? v : 0.f;
        if (!(BIAS_ABLATE & 16) || bp.total_rows == -12345) atomicAdd(base + min(row, rmax) * rstride + dc, v);   // (16: timing experiment, no dq adds)
      }
    }
  }
}

// all DBQ output blocks of one query tile, dealt to `n_help` waves; this wave has rank `rank`
template <typename T, int DQK, int DV>
HSTU_DEV void bwd_dq_tile(const HstuAttnBwdParams& bp, const MaskCtx& mc, const char* smem, const char* ds_base, int nw,
                          int kb0, int i0, int rank, int n_help, int64_t off0, int hd, float ds_scale, float* dq_accum,
                          char* dq_scratch, int lane, bool dq_plain = false) {
  constexpr int DBQ = DQK / 32;
  if constexpr (DBQ >= 2) {
    if (2 * n_help <= DBQ) {   // few helpers: each takes pairs of blocks (rank, rank + n_help), ...
      for (int db = rank; db + n_help < DBQ; db += 2 * n_help)
        bwd_dq_blocks<T, DQK, DV, 2>(bp, mc, smem, ds_base, nw, kb0, i0, db, n_help, off0, hd, ds_scale, dq_accum, dq_scratch, lane, dq_plain);
      if ((DBQ / n_help) & 1)  // odd number of rounds: one single block left per helper
        for (int db = rank + (DBQ / n_help - 1) * n_help; db < DBQ; db += n_help)
          bwd_dq_blocks<T, DQK, DV, 1>(bp, mc, smem, ds_base, nw, kb0, i0, db, 1, off0, hd, ds_scale, dq_accum, dq_scratch, lane, dq_plain);
      return;
    }
  }
  for (int db = rank; db < DBQ; db += n_help)
    bwd_dq_blocks<T, DQK, DV, 1>(bp, mc, smem, ds_base, nw, kb0, i0, db, 1, off0, hd, ds_scale, dq_accum, dq_scratch, lane, dq_plain);
}

template <typename T, int DQK, int DV, bool BIAS = false>
__global__ __launch_bounds__(kBwdThreads) void hstu_attn_bwd_kernel(const HstuAttnBwdParams bp, int nkb, int nw,
                                                                    float* dq_accum, float* bias_partial, int ts_copies,
                                                                    int bucket_cache_off, int64_t dq_slab_elems) {
  using C = BwdCfg<T, DQK, DV>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const HstuAttnParams& p = bp.fwd;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31;

  // ---- work decode (same grouping as forward: key blocks of one (user, head) adjacent)
  const int bid = blockIdx.x;
  const int grp = bid / (8 * nkb), rem = bid % (8 * nkb);
  const int kb = rem / 8;
  const int uh = grp * 8 + (rem & 7);
  // bucket_cache_off > 0 (research-path bias, one key block): ONE workgroup walks all heads of a user.  The time-bucket
  // matrix depends on the user only: the first head computes it (hardware log2 + the exactness check, ~12 VALU
  // instructions and a timestamp read per element) and leaves it as one byte per element in LDS, the other heads read
  // 8 bytes per lane and half tile; the tables are staged, the histograms flushed and reduced once per user.
  const bool head_loop = BIAS && bucket_cache_off > 0;
  if (uh >= (head_loop ? p.batch : p.batch * p.heads)) return;
  const int b = user_of_slot(p, head_loop ? uh : uh / p.heads), hd_first = head_loop ? 0 : uh % p.heads;
  const int n_heads = head_loop ? p.heads : 1;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = (int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0);
  const int kb0 = kb * 32 * nw;
  if (kb0 >= len) return;
  // deterministic mode: key block kb owns slab kb of the workspace and STORES its partial there (hstu_dq_convert_kernel adds
  // the slabs in block order); otherwise all key blocks add into one accumulator with atomics
  // (never with the relative bias: the launcher refuses the combination, and the BIAS instantiations sit at the register limit)
  const bool dq_plain = !BIAS && dq_slab_elems != 0;
  if (dq_plain) dq_accum += (int64_t)kb * dq_slab_elems;
  const MaskCtx mc = make_mask_ctx(p, b, len);
#ifdef HSTU_TRACE
  // one key block: the (otherwise unused) workspace pointer carries the trace buffer; several key blocks: the
  // forward kernel's trace pointer (set with hstu_trace_set_fwd) is borrowed
  HSTU_TRACE_DECL(dq_accum == nullptr ? bp.workspace : (void*)g_hstu_trace_fwd,
                  (dq_accum == nullptr ? bp.workspace != nullptr : g_hstu_trace_fwd != nullptr) && blockIdx.x == 4096);
#endif
  HSTU_MARK(1);

  char* const stage = smem + nw * C::PAIR;           // Q_i tile then dO_i tile
  char* const dsbuf = stage + C::PAIR;               // 2 x nw buffers of [32 keys][32 q] (double buffered)
  const int k0w = kb0 + 32 * wave;                   // first key of this wave's tile
  const bool tile_owner = wave < nw && k0w < len;
  // research-path bias: per-workgroup fp32 histograms of dS' over (j - i) and over the time bucket,
  // flushed to this workgroup's row of `bias_partial` and summed by a second kernel
  BiasCtx bc;
  // several key blocks: a [32 q][32 d] fp32 tile per wave for the coalesced dq adds, then the bias histograms
  char* const dq_scratch = dsbuf + 2 * nw * C::DSBUF + wave * (kDqScratchBytes / kBwdWaves);
  // [pos histogram, 2N floats][time-bucket histogram, (nb+1) x ts_copies][staged tables]
  float* const hpos = (float*)(dsbuf + 2 * nw * C::DSBUF + (nkb > 1 ? kDqScratchBytes : 0));
  float* const hts = hpos + 2 * p.max_seq_len;
  const int hist_floats = 2 * p.max_seq_len + (p.num_buckets + 1) * ts_copies;
  if constexpr (BIAS) {
    bc = stage_bias_tables(p, b, (char*)hpos + (hist_floats * 4 + 15) / 16 * 16, tid, kBwdThreads);
    for (int i = tid; i < hist_floats; i += kBwdThreads) hpos[i] = 0.f;
  }

  char* const bcache = (char*)hpos + bucket_cache_off;
  const float scale_v = attn_scale_of(p);
  const float ds_scale = scale_v * p.alpha;
  const int key = k0w + n32;
  const bool key_ok = tile_owner && key < len;
  int t_k32 = 0;
  TsRun ts_run;   // running sum of the current time bucket (hstu_common.cuh)
  ts_run.init(hts, ts_copies);

  const int tid_wg = tid;
  for (int hi = 0; hi < n_heads; ++hi) {
  const int hd = hd_first + hi;
  // (the thread id is laundered per head: the per-thread offsets of the prologue and the epilogue are recomputed there
  // instead of being hoisted out of the head loop and kept alive across the query-tile loop -- 70 registers)
  int tid_h = tid_wg;
  if constexpr (BIAS) asm volatile("" : "+v"(tid_h));
  const int tid = tid_h, lane = tid_h & 63, n32 = lane & 31, hf = lane >> 5;
  const char* qbase = (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const char* dobase = (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
  const int64_t q_rs = p.q_row_stride * C::EB, k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB,
                do_rs = bp.do_row_stride * C::EB;

  // ---- query tiles are visited in DESCENDING order.  Tile i needs key tiles <= i (causal), so
  // the steps with many active owner waves come first and every later step frees one more
  // wave: those idle waves run the dQ GEMM of the PREVIOUS step (whose cost shrinks at the same
  // pace), and an owner that has seen its last query tile stores dK/dV while the others go on.
  const int kt0 = kb0 >> 5;
  const int it_lo = (mc.ctx > 0) ? 0 : kt0;          // contextual rows (id 0) see every key
  const int it_hi = (len + 31) >> 5;

  // ---- resident K/V block by LDS-DMA: every 1 KiB chunk of the block is in flight at once, no
  // registers and no ds_write involved (rows past len receive a clamped copy: masked, never used)
  for (int w2 = 0; w2 < nw; ++w2)
    if (kb0 + 32 * w2 < len) {
      char* dst = smem + w2 * C::PAIR;
      tile_dma<T, DQK>(dst, kbase, k_rs, kb0 + 32 * w2, len, p.dqk, wave, kBwdWaves, lane);
      tile_dma<T, DV>(dst + C::KT, vbase, v_rs, kb0 + 32 * w2, len, p.dv, wave, kBwdWaves, lane);
    }
  HSTU_MARK(2);
  u32x4 qreg[C::NQU], oreg[C::NOU];
  tile_gload<T, DQK, C::NQU, kBwdThreads>(qreg, qbase, q_rs, (it_hi - 1) * 32, len, p.dqk, tid);
  tile_gload<T, DV, C::NOU, kBwdThreads>(oreg, dobase, do_rs, (it_hi - 1) * 32, len, p.dv, tid);
  tile_lds_write<T, DQK, C::NQU, kBwdThreads>(qreg, stage, (it_hi - 1) * 32, len, p.dqk, tid);
  tile_lds_write<T, DV, C::NOU, kBwdThreads>(oreg, stage + C::KT, (it_hi - 1) * 32, len, p.dv, tid);

  f32x16 dk_acc[C::DBQ], dv_acc[C::DBV];
#pragma unroll
  for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk_acc[d][r] = 0.f;
#pragma unroll
  for (int d = 0; d < C::DBV; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dv_acc[d][r] = 0.f;
  __syncthreads();
  HSTU_MARK(3);

  if constexpr (BIAS) {   // (after the barrier above: the staged tables are visible)
    if (hi == 0) {
      bc.finish(kBwdWaves);
      if (bc.small) t_k32 = bc.t32_at(key);
    }
  }
  const bool bkt_cached = head_loop && hi > 0;

  for (int it = it_hi - 1; it >= it_lo; --it) {
    const int i0 = it << 5;
    const bool more = it > it_lo;
    if (more) {
      tile_gload<T, DQK, C::NQU, kBwdThreads>(qreg, qbase, q_rs, i0 - 32, len, p.dqk, tid);
      tile_gload<T, DV, C::NOU, kBwdThreads>(oreg, dobase, do_rs, i0 - 32, len, p.dv, tid);
    }
    HSTU_MARK(10);
    const bool active = tile_owner && mc.pair_may_be_active(i0, 32, k0w, 32);
    if (active) {
      // ------------------------------ phase 1 (owner of key tile `wave`) ------------------------------
      // (lane id laundered per phase: LDS offsets derived from it are recomputed here instead of being hoisted
      // out of the query-tile loop and kept alive next to the 128 accumulator registers -- see hstu_attn_bwd_fold.cuh)
      int lane_p = lane;
      asm volatile("" : "+v"(lane_p));
      const int n32 = lane_p & 31, hf = lane_p >> 5;
      const char* Kw = smem + wave * C::PAIR;
      const char* Vw = Kw + C::KT;
      const char* Qs = stage;
      const char* dOs = stage + C::KT;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int kg = 0; kg < C::KGQ; ++kg) {
        const int e0 = hf * (DQK / 2) + kg * 8;
        Frag a = lds_row_frag<T, C::UPR_K>(Qs, n32, e0);
        Frag bb = lds_row_frag<T, C::UPR_K>(Kw, n32, e0);
        s = E::mma(a, bb, s);
      }
#pragma unroll
      for (int kg = 0; kg < C::KGV; ++kg) {
        const int e0 = hf * (DV / 2) + kg * 8;
        Frag a = lds_row_frag<T, C::UPR_V>(dOs, n32, e0);
        Frag bb = lds_row_frag<T, C::UPR_V>(Vw, n32, e0);
        dp = E::mma(a, bb, dp);
      }
      HSTU_MARK(11);
      // C layout: column n32 = key, register r = query row (r&3) + 8 (r>>2) + 4 hf.
      // P' = silu(x), dS' = dP silu'(x) with x = alpha S; scale and alpha are applied in fp32
      // to the accumulators at the very end (dV *= scale; dK, dQ *= scale * alpha)
      Frag pb[2], dsb[2];
      const int mode = mc.pair_fully_valid(i0, 32, k0w, 32) ? 0 : (mc.simple ? 1 : 2);   // wave-uniform
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        float pv[8], dsv[8];
        int pidx[8], bkt[8];
        float xb[8];
        int tq4[8];
        // this lane's 8 bucket bytes of the half tile (pairs of one key block: it >= wave)
        const int bslot_off = ((((it * (it + 1)) >> 1) + wave) * 2 + h8) * 512;    // (scalar; the lane part is added at the use)
        if constexpr (BIAS) {
          if (bc.small && !bkt_cached) {   // next-item timestamps of the 4 consecutive query rows of a register group: one 16-byte LDS read
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const auto t4 = bc.t32x4_next(i0 + 8 * (2 * h8 + g) + 4 * hf);
#pragma unroll
              for (int j = 0; j < 4; ++j) tq4[4 * g + j] = t4[j];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xb[j] = 0.f;
        if constexpr (BIAS) {
          if (bkt_cached) {
            const u32x2 w = *LDS_PTR(const u32x2, bcache + bslot_off + 8 * lane_p);
#pragma unroll
            for (int j = 0; j < 8; ++j) bkt[j] = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
          } else {
            if (bc.small) {   // (workgroup-uniform: one branch per half tile, not one per element)
#pragma unroll
              for (int j = 0; j < 8; ++j) bkt[j] = bc.bucket32(tq4[j], t_k32);
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const int r = 8 * h8 + j;
                bkt[j] = bc.bucket(bc.ts_at(i0 + (r & 3) + 8 * (r >> 2) + 4 * hf + 1), bc.ts_at(key));
              }
            }
            if (head_loop) {
              u32x2 w = {0u, 0u};
#pragma unroll
              for (int j = 0; j < 8; ++j) w[j >> 2] |= (unsigned)bkt[j] << (8 * (j & 3));
              *LDS_PTR(u32x2, bcache + bslot_off + 8 * lane_p) = w;
            }
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            const int qi = i0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            pidx[j] = bc.pos_index(qi, key);
            xb[j] = (BIAS_ABLATE & 4) ? 0.f : bc.value(pidx[j], bkt[j]);
          }
        }
        {   // two elements per VALU instruction where the ISA has a packed fp32 form (mul / add / fma)
          const f32x2 a2 = {p.alpha, p.alpha}, one2 = {1.f, 1.f}, nl2 = {-1.44269504088896340736f, -1.44269504088896340736f};
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const int r = 8 * h8 + j;
            const f32x2 sv = {s[r], s[r + 1]}, dpv = {dp[r], dp[r + 1]}, bv = {xb[j], xb[j + 1]};
            const f32x2 x = sv * a2 + bv, t = x * nl2;
            const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            const f32x2 dn = e + one2;
            const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
            const f32x2 pr = x * sg, w = x * (one2 - sg) + one2, dsr = dpv * sg * w;
            pv[j] = pr[0]; pv[j + 1] = pr[1];
            dsv[j] = dsr[0]; dsv[j + 1] = dsr[1];
          }
        }
        if (mode == 1) {          // plain causal, no targets: key <= query (and both in range)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            const int qi = i0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const bool ok = key_ok & (qi < len) & (key <= qi);
            pv[j] = ok ? pv[j] : 0.f;
            dsv[j] = ok ? dsv[j] : 0.f;
          }
        } else if (mode == 2) {   // general mask algebra
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            const int qi = i0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const bool ok = key_ok & (qi < len) & mc.valid_ids(qi, key, mc.id_of(qi), mc.id_of(key));
            pv[j] = ok ? pv[j] : 0.f;
            dsv[j] = ok ? dsv[j] : 0.f;
          }
        }
        if constexpr (BIAS) {
          // d bias = dS (summed over heads / users later); masked-out elements carry exact zeros.
          // Position histogram: bin = n-1 + key - query.  The 4 consecutive query rows of a register group (j = 0..3)
          // of the 4 consecutive keys held by lanes l .. l+3 lie on ONE diagonal: lane l collects them with three DPP
          // row shifts and issues one atomic instead of four (LDS float atomics cost ~200 cycles per instruction
          // here).  What a shift pushes out of a 16-lane row (the first j lanes of a row for element j) is added by its
          // owner: three more instructions with 4 / 8 / 12 active lanes.
          const int p16 = lane & 15;
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            float t = dsv[4 * gg + 3];
#pragma unroll
            for (int j = 2; j >= 0; --j)
              t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x101, 0xf, 0xf, true)) +
                  dsv[4 * gg + j];
            if (!(BIAS_ABLATE & 1) && t != 0.f) atomicAdd(hpos + pidx[4 * gg], t);
#pragma unroll
            for (int j = 1; j < 4; ++j)
              if (!(BIAS_ABLATE & 1) && p16 < j && dsv[4 * gg + j] != 0.f) atomicAdd(hpos + pidx[4 * gg + j], dsv[4 * gg + j]);
            // (collecting the pushed-out elements on lanes 0..2 with four more row shifts -- one atomic instead of three --
            // was measured: -1.4 %, and one spilled register at head dim 128)
          }
          if (!(BIAS_ABLATE & 2) && bc.lts) {
#pragma unroll
            for (int j = 0; j < 8; ++j) ts_run.add(bkt[j], dsv[j]);
          }
        }
        pb[h8] = E::pack8(pv);
        dsb[h8] = E::pack8(dsv);
      }
      HSTU_MARK(12);
      // dV_w^T[dv][key] += dO_i^T[dv][q] P'[q][key]
#pragma unroll
      for (int d = 0; d < C::DBV; ++d)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Frag a = lds_col_frag<T, C::UPR_V>(dOs, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, lane_p);
          dv_acc[d] = E::mma(a, pb[ks], dv_acc[d]);
        }
      // dK_w^T[d][key] += Q_i^T[d][q] dS'[q][key]
#pragma unroll
      for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Frag a = lds_col_frag<T, C::UPR_K>(Qs, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, lane_p);
          dk_acc[d] = E::mma(a, dsb[ks], dk_acc[d]);
        }
      HSTU_MARK(13);
      // publish dS' as [key = n32][q]: this lane holds q = 4 hf + 8 rq + (0..3), rq = 0..3,
      // i.e. slots 4 (rq & 1) .. +3 of dsb[rq >> 1]
      char* myds = dsbuf + ((it & 1) * nw + wave) * C::DSBUF + n32 * C::DSROW;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int qloc = 4 * hf + 8 * rq;
        if constexpr (C::EB == 2) {
          const u32x4 w = __builtin_bit_cast(u32x4, dsb[rq >> 1].v);
          u32x2 v2 = {w[2 * (rq & 1)], w[2 * (rq & 1) + 1]};
          *LDS_PTR(u32x2, myds + qloc * 2) = v2;
        } else {
          f32x4 v4 = {dsb[rq >> 1].v[(rq & 1) * 4 + 0], dsb[rq >> 1].v[(rq & 1) * 4 + 1], dsb[rq >> 1].v[(rq & 1) * 4 + 2],
                      dsb[rq >> 1].v[(rq & 1) * 4 + 3]};
          *LDS_PTR(f32x4, myds + qloc * 4) = v4;
        }
      }
    } else if (it + 1 < it_hi) {
      // ------------------------------ dQ GEMM of the PREVIOUS step (query tile it+1), on an idle wave ------------------------------
      // helpers = waves without phase-1 work in this step (wave 7 never owns a tile, so there is
      // always one); the DQK/32 output blocks are dealt round-robin to them.
      int n_help = 0, my_rank = 0;
      for (int w2 = 0; w2 < kBwdWaves; ++w2) {
        const bool act = w2 < nw && (kb0 + 32 * w2) < len && mc.pair_may_be_active(i0, 32, kb0 + 32 * w2, 32);
        if (!act) {
          if (w2 < wave) ++my_rank;
          ++n_help;
        }
      }
      const char* ds_prev = dsbuf + (((it + 1) & 1) * nw) * C::DSBUF;
      int lane_h = lane;
      asm volatile("" : "+v"(lane_h));
      bwd_dq_tile<T, DQK, DV>(bp, mc, smem, ds_prev, nw, kb0, i0 + 32, my_rank, n_help, off0, hd, ds_scale, dq_accum, dq_scratch, lane_h, dq_plain);
    }
    HSTU_MARK(14);
    lds_barrier();  // stage reads done; dS'(it) complete; dS'(it+1) consumed (LDS-only: the dq adds stay in flight)
    HSTU_MARK(15);
    if (more) {
      tile_lds_write<T, DQK, C::NQU, kBwdThreads>(qreg, stage, i0 - 32, len, p.dqk, tid);
      tile_lds_write<T, DV, C::NOU, kBwdThreads>(oreg, stage + C::KT, i0 - 32, len, p.dv, tid);
    }
    HSTU_MARK(16);
    lds_barrier();  // next Q/dO tile visible
    HSTU_MARK(18);
  }
  HSTU_MARK(20);
  // ---- dQ of the last visited query tile: every wave is idle now
  if (it_hi > it_lo && wave < C::DBQ) {
    const char* ds_prev = dsbuf + ((it_lo & 1) * nw) * C::DSBUF;
    bwd_dq_blocks<T, DQK, DV, 1>(bp, mc, smem, ds_prev, nw, kb0, it_lo << 5, wave, 1, off0, hd, ds_scale, dq_accum, dq_scratch, lane, dq_plain);
  }
  HSTU_MARK(21);
  // ---- epilogue: dK_w^T / dV_w^T accumulators (column n32 = key) -> rows of dk / dv.  16-bit I/O with the
  // instantiated head dims: through LDS -- each owner wave writes its two [32 keys][D] tiles over its own (dead) K/V
  // tiles in the swizzled row-major layout, reads 16-byte units back with 16 consecutive lanes per row and stores
  // whole rows with dwordx4 (storing the accumulators directly is D/4 scattered dwordx2 stores per lane: store-issue
  // bound, see hstu_attn_fwd.cuh).
  if constexpr (C::EB == 2) {
    if (p.dqk == DQK && p.dv == DV) {   // wave-uniform
      __syncthreads();                  // the last dQ GEMM has read every K tile
      if (tile_owner) {
        char* kt = smem + wave * C::PAIR;
        char* vt = kt + C::KT;
#pragma unroll
        for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            u32x2 v = {E::pk2(dk_acc[d][4 * rq] * ds_scale, dk_acc[d][4 * rq + 1] * ds_scale),
                       E::pk2(dk_acc[d][4 * rq + 2] * ds_scale, dk_acc[d][4 * rq + 3] * ds_scale)};
            *LDS_PTR(u32x2, kt + tile_off<C::UPR_K>(n32, 4 * d + rq) + 8 * hf) = v;
          }
#pragma unroll
        for (int d = 0; d < C::DBV; ++d)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            u32x2 v = {E::pk2(dv_acc[d][4 * rq] * scale_v, dv_acc[d][4 * rq + 1] * scale_v),
                       E::pk2(dv_acc[d][4 * rq + 2] * scale_v, dv_acc[d][4 * rq + 3] * scale_v)};
            *LDS_PTR(u32x2, vt + tile_off<C::UPR_V>(n32, 4 * d + rq) + 8 * hf) = v;
          }
        const int rows_valid = len - k0w;
        char* dkt = (char*)bp.dk + ((off0 + k0w) * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
        char* dvt = (char*)bp.dv + ((off0 + k0w) * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
#pragma unroll
        for (int i = 0; i < 32 * C::UPR_K / 64; ++i) {
          const int idx = i * 64 + lane;
          const int row = idx / C::UPR_K, unit = idx % C::UPR_K;
          const u32x4 v = *LDS_PTR(const u32x4, kt + tile_off<C::UPR_K>(row, unit));
          if (row < rows_valid) gstore16(dkt + (int64_t)row * bp.dk_row_stride * C::EB + unit * 16, v);
        }
#pragma unroll
        for (int i = 0; i < 32 * C::UPR_V / 64; ++i) {
          const int idx = i * 64 + lane;
          const int row = idx / C::UPR_V, unit = idx % C::UPR_V;
          const u32x4 v = *LDS_PTR(const u32x4, vt + tile_off<C::UPR_V>(row, unit));
          if (row < rows_valid) gstore16(dvt + (int64_t)row * bp.dv_row_stride * C::EB + unit * 16, v);
        }
      }
      if (hi + 1 < n_heads) __syncthreads();   // copy-out reads done before the next head's K/V land
      continue;
    }
  }
  if (key_ok) {
      char* dkrow = (char*)bp.dk + ((off0 + key) * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
      char* dvrow = (char*)bp.dv + ((off0 + key) * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
#pragma unroll
      for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = 32 * d + 8 * rq + 4 * hf;
          if (d0 < p.dqk)
            store4<T>(dkrow, d0, dk_acc[d][4 * rq] * ds_scale, dk_acc[d][4 * rq + 1] * ds_scale,
                      dk_acc[d][4 * rq + 2] * ds_scale, dk_acc[d][4 * rq + 3] * ds_scale);
        }
#pragma unroll
      for (int d = 0; d < C::DBV; ++d)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = 32 * d + 8 * rq + 4 * hf;
          if (d0 < p.dv)
            store4<T>(dvrow, d0, dv_acc[d][4 * rq] * scale_v, dv_acc[d][4 * rq + 1] * scale_v, dv_acc[d][4 * rq + 2] * scale_v,
                      dv_acc[d][4 * rq + 3] * scale_v);
        }
    }
  if (hi + 1 < n_heads) __syncthreads();
  }   // heads
  if constexpr (BIAS) {
    ts_run.flush();
    __syncthreads();
    float* row = bias_partial + (int64_t)blockIdx.x * (2 * p.max_seq_len + p.num_buckets);
    const int npos = 2 * p.max_seq_len - 1;
    for (int i = tid; i < 2 * p.max_seq_len + p.num_buckets; i += kBwdThreads) {
      float v;
      if (i < npos) {
        v = hpos[i];
      } else {
        v = 0.f;
        const float* cp = hts + (i - npos) * ts_copies;
        for (int c = 0; c < ts_copies; ++c) v += cp[c];
      }
      row[i] = v * scale_v;
    }
  }
}

// fp32 dq accumulator (rows, H, dqk) -> dq in the I/O dtype (strided).  One thread per 16 bytes of OUTPUT: 8 features
// for 16-bit I/O, 4 for fp32 (dqk is a multiple of that -- the boundary checks it -- but not necessarily of 8).
template <typename T>
__global__ __launch_bounds__(256) void hstu_dq_convert_kernel(const float* acc, void* dq, int64_t rows, int heads, int dqk,
                                                              int64_t row_stride, int64_t head_stride, int n_slabs, int64_t slab_elems) {
  constexpr int VEC = 16 / Elem<T>::kBytes;
  const int pph = dqk / VEC;                // pieces per (row, head)
  const int ppr = heads * pph;              // pieces per row
  const int64_t n = rows * ppr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ppr;
    const int rem = (int)(i - r * ppr);
    const int h = rem / pph, c = (rem - h * pph) * VEC;
    const float* src = acc + i * VEC;       // the accumulator is dense: (row, head, feature)
    T* out = (T*)dq + r * row_stride + h * head_stride + c;
    f32x4 a = *reinterpret_cast<const f32x4*>(src);
    for (int sl = 1; sl < n_slabs; ++sl) a += *reinterpret_cast<const f32x4*>(src + sl * slab_elems);   // (deterministic: key blocks in order)
    if constexpr (Elem<T>::kBytes == 2) {
      f32x4 b = *reinterpret_cast<const f32x4*>(src + 4);
      for (int sl = 1; sl < n_slabs; ++sl) b += *reinterpret_cast<const f32x4*>(src + 4 + sl * slab_elems);
      u32x4 v = {Elem<T>::pk2(a[0], a[1]), Elem<T>::pk2(a[2], a[3]), Elem<T>::pk2(b[0], b[1]), Elem<T>::pk2(b[2], b[3])};
      *reinterpret_cast<u32x4*>(out) = v;
    } else {
      *reinterpret_cast<f32x4*>(out) = a;
    }
  }
}

}  // namespace hstu
