// HSTU attention backward, short sequences at head dim 128: ONE workgroup of SIXTEEN waves per CU (four per SIMD, 128
// registers each), every wave the owner of 16 keys for the whole problem.
//
// Why (round-4 finding, docs/EXPERIMENTS.md R4.2): the folded kernel (hstu_attn_bwd_fold.cuh) runs two 256-register waves
// per SIMD; in its memory phases the waves sit blocked on ISSUING LDS-DMA / stores and nothing else is on the SIMD to
// use the issue slots (the one-wave-per-SIMD experiment made it worse: 44 % of the slots used against 72 %).  Here four
// waves share a SIMD, and the structure changes with the ownership:
//   * wave (t, h) = (wave >> 1, wave & 1) owns keys [32 t + 16 h, +16) of the user -- 14 of the 16 waves own all 224 rows a
//     problem can have -- and keeps their dK^T / dV^T in 64 registers from the first step to the last: no fold, no second
//     partial sum, no hand-over tail.  All MFMAs are 16x16x32 (C layout: lane = key, registers = 4 query rows, so P' and
//     dS' feed the second MFMA straight from registers as in the folded kernel).
//   * the steps take ONE query tile each, last tile first (tile i needs key tiles 0..i): only one Q/dO stage is live, so the
//     32 KiB of stage LDS are a DOUBLE BUFFER -- tile i-1 is requested at the top of step i and has the whole step to land
//     (the folded kernel needs both stages in every step and can only request after its pairs have finished reading).
//   * key tile t is final after step t: dV is parked over the dead V tile at once, dK over the K tile one step later, the
//     rows leave in step t-1, and from step t-2 on the slot takes the NEXT problem's K/V tile t -- loads, stores and
//     arithmetic of neighbouring steps / problems overlap; per step 32 KiB come in and 24 KiB go out whatever the step.
//   * dQ of tile i: all sixteen waves, (16 features, 16 query rows) each, one 16x16x32 MFMA per key tile.
// LDS: 7 K/V pairs (112 KiB) + 2 Q/dO stages (32 KiB) + 7 dS' tiles (14 KiB) = 158 KiB.  Tiles are row-major [32][128]
// 16-bit with their 16-byte units XOR-swizzled by w16_swz (NOT hstu_common's swz: this kernel's transposed reads take 16
// columns x 8 consecutive rows per pass); every LDS access of the kernel is conflict-free by construction (checked in the
// comments at each access).  Same math, masks (S accumulator start value) and rounding points as the folded kernel.
// Requires what the folded kernel requires (attn_bwd_fold_applicable), head dims 128 and no attention window.
#pragma once
#include "hstu_attn_bwd_fold.cuh"

#ifndef W16_ABLATE
#define W16_ABLATE 0       // timing experiments only (WRONG results): 1 no dQ stores, 2 no dk/dv stores, 4 no stage DMA after the
#endif                     // first, 16 no K/V DMA, 32 no dQ GEMM, 64 no pairs
#ifndef W16_NEXT_KV
#define W16_NEXT_KV 1      // the next problem's K/V tiles stream into the slots this problem has finished with
#endif

namespace hstu {

constexpr int kW16Waves = 16;
constexpr int kW16Threads = 1024;

template <typename T, int D>
struct W16Cfg {
  using B = BwdCfg<T, D, D>;
  static constexpr int DSB = 32 * 64;                 // [32 keys][32 q] 16-bit dS' tile
  static constexpr int kMaxTiles = 7;
  static constexpr int smem_bytes() { return kMaxTiles * B::PAIR + 2 * B::PAIR + kMaxTiles * DSB; }
};

// unit swizzle of a 256-byte row: a bijection of r & 15 (ds_read_b128 of one unit column from 16 rows: 16 different slots),
// and rows 8 m .. 8 m + 7 get 8 different unit PAIRS (ds_read_b64_tr_b16 of a 32-byte column block from 8 consecutive rows:
// 32 lanes, 32 different 8-byte bank slots)
HSTU_DEV int w16_swz(int r) { return ((r & 7) << 1) | ((r >> 3) & 1); }
HSTU_DEV int w16_toff(int r, int u) { return (r * 16 + (u ^ w16_swz(r))) << 4; }
// dS' tile: 64-byte rows, 8-byte chunks (4 query columns) XOR-swizzled so that both the owners' ds_write_b64 (16 rows x 2
// chunks per 32-lane pass) and the dQ GEMM's transposed reads (8 consecutive rows x 4 chunks per pass) are conflict-free
HSTU_DEV int w16_ds_off(int row, int chunk) {
  const int f = (((row >> 2) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 4) & 1);
  return (row << 6) + ((chunk ^ f) << 3);
}

// LDS-DMA of a tile PAIR (Q + dO, or K + V: 2 x [32][D] = 16 chunks of 1 KiB): wave w moves chunk w & 7 of tile w >> 3
template <typename T, int D>
HSTU_DEV void w16_pair_dma(char* pair, const char* base0, int64_t rs0, const char* base1, int64_t rs1, int row0, int len,
                           int wave, int lane, bool fast) {
  static_assert(D * Elem<T>::kBytes == 256, "256-byte rows");
  const bool second = wave >= 8;                       // wave-uniform
  const char* base = second ? base1 : base0;
  const int64_t rs = second ? rs1 : rs0;
  const int c = wave & 7;
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(pair + (second ? 32 * 256 : 0)));
  const int pidx = c * 64 + lane;
  const int row = pidx >> 4, slot = pidx & 15;
  const int unit = slot ^ w16_swz(row);
  const int grow = min(row0 + row, len - 1);
  if (fast) dma16_saddr(__umul24((uint32_t)grow, (uint32_t)rs) + unit * 16, base, lds0 + c * 1024);
  else dma16_asm(base + (int64_t)grow * rs + unit * 16, lds0 + c * 1024);
}

// One (query tile i0, 16 keys k0 + 16 h ..) half pair on the owner wave.
//   S[q][key] / dP[q][key]: A = Q / dO rows (m = q), B = K / V rows (n = key); C: lane (key = i16, g), register r of block qb
//   <-> query row 16 qb + 4 g + r.  P' and dS' packed in that order ARE the B operand (k = q) of
//   dV^T[dv][key] += dO^T[dv][q] P'[q][key],  dK^T[d][key] += Q^T[d][q] dS'[q][key]   (A: transposed reads, rows
//   {4 g + j} u {16 + 4 g + j}: the same k order).
template <typename T, int D>
HSTU_DEV void w16_pair(const HstuAttnParams& p, const MaskCtx& mc, const char* __restrict__ Kw, const char* __restrict__ Vw,
                       const char* __restrict__ Qs, const char* __restrict__ dOs, char* __restrict__ myds, int i0, int k0, int h,
                       f32x4 (&dk_acc)[D / 16], f32x4 (&dv_acc)[D / 16], int lane, int dmvm) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  static_assert(D == 128, "eight 16-feature blocks, four 32-wide contraction slices");
  const int i16 = lane & 15, g = lane >> 4;
  const int len = mc.len;
  const int krow = 16 * h + i16;
  const int key = k0 + krow;
  f32x4 s[2], dp[2];
  {
    // mask bits (bit e = 4 qb + r <-> query row 16 qb + 4 g + r survives), applied as the S accumulator's start value
    // (-1e30 -> sigmoid = 0 exactly: fold_pair_x)
    int mode;   // wave-uniform: 0 = no mask needed, 1 = plain causal, 2 = general mask algebra
    if (mc.simple) mode = (k0 < i0 && i0 + 32 <= len) ? 0 : 1;
    else mode = (i0 + 32 <= len && mc.pair_fully_valid(i0, 32, k0, 32)) ? 0 : 2;
    int km = -1;
    if (mode == 1) km = ((k0 == i0) ? dmvm : -1) & ((i0 + 32 > len) ? (dmvm >> 8) : -1);
    if (mode == 2) {
      const int key_id = mc.id_of(key);
      const int key_bits = key < len ? -1 : 0;
      km = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int qi = i0 + 16 * (e >> 2) + 4 * g + (e & 3);
        km |= (mc.keep_bits_noctx(qi, key, key_id) & key_bits & 1) << e;
      }
    }
    const unsigned nk = ~(unsigned)km;
    const unsigned neg = __builtin_bit_cast(unsigned, p.alpha < 0.f ? 1e30f : -1e30f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = ((int)(nk << (31 - e))) >> 31;        // all ones iff masked
      s[e >> 2][e & 3] = __builtin_bit_cast(float, (unsigned)m & neg);
      dp[e >> 2][e & 3] = 0.f;
    }
  }
  // ---- S and dP: 8 half items (slice ks of the head dim x {S, dP}), each 3 row fragments (ds_read_b128: the 16 lanes of a
  // group read one unit column of 16 rows -- conflict-free, w16_swz is a bijection of r & 15) and 2 MFMAs; the reads of half
  // item n + 1 are issued before the MFMAs of half item n.
  {
    const int xs = w16_swz(i16) ^ g;                       // unit 4 ks + g of a row r with r & 15 == i16 sits in slot (4 ks) ^ xs
    const int rk = krow * 256, rq0 = i16 * 256, rq1 = (16 + i16) * 256;
    auto load_half = [&](int n, Frag& fb, Frag& fa0, Frag& fa1) {
      const int ks = n >> 1;
      const int uo = ((4 * ks) ^ xs) << 4;
      const char* bt = (n & 1) ? Vw : Kw;
      const char* at = (n & 1) ? dOs : Qs;
      fb.v = __builtin_bit_cast(typename E::vec8, *LDS_PTR(const u32x4, bt + rk + uo));
      fa0.v = __builtin_bit_cast(typename E::vec8, *LDS_PTR(const u32x4, at + rq0 + uo));
      fa1.v = __builtin_bit_cast(typename E::vec8, *LDS_PTR(const u32x4, at + rq1 + uo));
    };
    Frag fb[2], fa0[2], fa1[2];
    load_half(0, fb[0], fa0[0], fa1[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      if (n + 1 < 8) load_half(n + 1, fb[(n + 1) & 1], fa0[(n + 1) & 1], fa1[(n + 1) & 1]);
      if (n & 1) {
        dp[0] = E::mma16(fa0[n & 1], fb[n & 1], dp[0]);
        dp[1] = E::mma16(fa1[n & 1], fb[n & 1], dp[1]);
      } else {
        s[0] = E::mma16(fa0[n & 1], fb[n & 1], s[0]);
        s[1] = E::mma16(fa1[n & 1], fb[n & 1], s[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- element-wise: P' = x sigmoid(x), dS' = dP sigmoid(x) (1 + x (1 - sigmoid(x))), x = alpha S (packed fp32 pairs)
  Frag pb, dsb;
  {
    float pv[8], dsv[8];
    const f32x2 a2 = {p.alpha, p.alpha};
    const f32x2 c2 = {-1.44269504088896340736f * p.alpha, -1.44269504088896340736f * p.alpha};
    const f32x2 one2 = {1.f, 1.f};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      const f32x2 sv = {s[e >> 2][e & 3], s[e >> 2][(e & 3) + 1]}, dpv = {dp[e >> 2][e & 3], dp[e >> 2][(e & 3) + 1]};
      const f32x2 x = sv * a2, t = sv * c2;
      const f32x2 ex = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
      const f32x2 dn = ex + one2;
      const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
      const f32x2 pr = x * sg;
      const f32x2 w = x * (one2 - sg) + one2;
      const f32x2 dsr = dpv * sg * w;
      pv[e] = pr[0]; pv[e + 1] = pr[1];
      dsv[e] = dsr[0]; dsv[e + 1] = dsr[1];
    }
    pb = E::pack8(pv);
    dsb = E::pack8(dsv);
  }
  // ---- dV^T += dO^T P',  dK^T += Q^T dS': 16 items (feature block db x {dV, dK}), A by two transposed reads (32-lane pass =
  // lane groups g, g + 1: rows 4 g .. 4 g + 7 x 32 bytes -- 8 different unit pairs: conflict-free), two items ahead
  {
    const int rr = i16 >> 2, c4 = i16 & 3;
    const int ra = 4 * g + rr;
    const int xa = (c4 >> 1) ^ w16_swz(ra);                // w16_swz(16 + ra) == w16_swz(ra)
    const int oa = ra * 256 + ((c4 & 1) << 3), ob = oa + 16 * 256;
    constexpr int NM = 16, AHEAD = 2;
    Frag fa[AHEAD + 1];
    auto load_item = [&](int m, Frag& a) {
      const int db = m >> 1;
      const int uo = ((2 * db) ^ xa) << 4;
      a = tr_frag16<T>((m & 1) ? Qs : dOs, oa + uo, ob + uo);
    };
#pragma unroll
    for (int m = 0; m < AHEAD; ++m) load_item(m, fa[m % (AHEAD + 1)]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (m + AHEAD < NM) load_item(m + AHEAD, fa[(m + AHEAD) % (AHEAD + 1)]);
      const int db = m >> 1;
      if (m & 1) dk_acc[db] = E::mma16(fa[m % (AHEAD + 1)], dsb, dk_acc[db]);
      else dv_acc[db] = E::mma16(fa[m % (AHEAD + 1)], pb, dv_acc[db]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- publish dS' as [key][q]: this lane holds q = 16 qb + 4 g + (0..3) = chunk 4 qb + g of key row krow
  {
    const u32x4 w = __builtin_bit_cast(u32x4, dsb.v);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) *LDS_PTR(u32x2, myds + w16_ds_off(krow, 4 * qb + g)) = u32x2{w[2 * qb], w[2 * qb + 1]};
  }
}

// finished dK^T / dV^T of this wave's 16 keys -> rows 16 h .. 16 h + 15 of a row-major [32][D] tile (ds_write_b64: a 32-lane
// pass = 16 rows x one unit x both halves: 32 different 8-byte slots)
template <typename T, int D>
HSTU_DEV void w16_park(const f32x4 (&acc)[D / 16], float scale, char* __restrict__ tile, int h, int lane) {
  const int i16 = lane & 15, g = lane >> 4;
  const int krow = 16 * h + i16;
#pragma unroll
  for (int db = 0; db < D / 16; ++db) {
    const u32x2 v = {Elem<T>::pk2(acc[db][0] * scale, acc[db][1] * scale), Elem<T>::pk2(acc[db][2] * scale, acc[db][3] * scale)};
    *LDS_PTR(u32x2, tile + w16_toff(krow, 2 * db + (g >> 1)) + 8 * (g & 1)) = v;
  }
}

// both parked tiles of a K/V slot (dk over K, dv over V) -> global memory: 1024 threads, one 16-byte unit each, 16 lanes per
// 256-byte row
template <typename T, int D>
HSTU_DEV void w16_copy_out(const char* __restrict__ pair, char* dk_rows, int64_t dk_rs, char* dv_rows, int64_t dv_rs, int rows_valid,
                           int tid) {
  const int second = tid >> 9, u = tid & 511;
  const int row = u >> 4, unit = u & 15;
  const u32x4 v = *LDS_PTR(const u32x4, pair + second * (32 * 256) + w16_toff(row, unit));
  char* dst = second ? dv_rows + row * dv_rs : dk_rows + row * dk_rs;
  if (row < rows_valid && (!(W16_ABLATE & 2) || dk_rs == -12345)) gstore16_nt(dst + unit * 16, v);
}

// dQ of query tile qt: wave (db8 = wave & 7, qh = wave >> 3) owns features [16 db8, +16) of query rows [16 qh, +16):
// dQ^T[d][q] = sum over key tiles t < N of K_t^T[d][key] dS'_t^T[key][q], one 16x16x32 MFMA per key tile (k order: lane group g
// <-> keys {4 g + j} u {16 + 4 g + j} for both operands), two accumulators (even / odd tiles), fragments of tile t + 2
// requested ahead.  Transposed reads: 8 consecutive rows per 32-lane pass, conflict-free for K (w16_swz) and dS' (w16_ds_off).
template <typename T, int D, int N>
HSTU_DEV f32x4 w16_dq_chain(const char* __restrict__ kv, const char* __restrict__ dsbuf, int db8, int qh, int lane) {
  using C = BwdCfg<T, D, D>;
  using F = W16Cfg<T, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int i16 = lane & 15, g = lane >> 4;
  const int rr = i16 >> 2, c4 = i16 & 3;
  const int ra = 4 * g + rr;
  const int ka = ra * 256 + ((((2 * db8 + (c4 >> 1)) ^ w16_swz(ra))) << 4) + ((c4 & 1) << 3), kb = ka + 16 * 256;
  const int da = w16_ds_off(ra, 4 * qh + c4), dh = w16_ds_off(16 + ra, 4 * qh + c4);
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int t = 0; t < N; ++t) {
    const Frag a = tr_frag16<T>(kv + t * C::PAIR, ka, kb);
    const Frag b = tr_frag16<T>(dsbuf + t * F::DSB, da, dh);
    acc[t & 1] = E::mma16(a, b, acc[t & 1]);
  }
  {
    constexpr int AH = 2 < N ? 2 : N;
    __builtin_amdgcn_sched_group_barrier(0x100, 4 * AH, 0);
#pragma unroll
    for (int t = 0; t < N; ++t) {
      if (t + AH < N) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
  }
  return acc[0] + acc[1];
}

template <typename T, int D>
HSTU_DEV void w16_dq_phase(const HstuAttnBwdParams& bp, const MaskCtx& mc, const char* __restrict__ kv, const char* __restrict__ dsbuf,
                           int qt, int wave, int64_t off0, int hd, float ds_scale, int lane) {
  using C = BwdCfg<T, D, D>;
  using E = Elem<T>;
  const int db8 = wave & 7, qh = wave >> 3;
  f32x4 acc;
  switch (qt) {   // wave-uniform: key tiles 0 .. qt
    case 0: acc = w16_dq_chain<T, D, 1>(kv, dsbuf, db8, qh, lane); break;
    case 1: acc = w16_dq_chain<T, D, 2>(kv, dsbuf, db8, qh, lane); break;
    case 2: acc = w16_dq_chain<T, D, 3>(kv, dsbuf, db8, qh, lane); break;
    case 3: acc = w16_dq_chain<T, D, 4>(kv, dsbuf, db8, qh, lane); break;
    case 4: acc = w16_dq_chain<T, D, 5>(kv, dsbuf, db8, qh, lane); break;
    case 5: acc = w16_dq_chain<T, D, 6>(kv, dsbuf, db8, qh, lane); break;
    default: acc = w16_dq_chain<T, D, 7>(kv, dsbuf, db8, qh, lane); break;
  }
  // C layout: column i16 = query row 16 qh + i16, register r = feature 16 db8 + 4 g + r: 8 bytes per lane
  const int i16 = lane & 15, g = lane >> 4;
  const int qrow = 32 * qt + 16 * qh + i16;
  if (qrow < mc.len && (!(W16_ABLATE & 1) || bp.total_rows == -12345)) {
    char* dst = (char*)bp.dq + ((off0 + qrow) * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB + (16 * db8 + 4 * g) * C::EB;
    *reinterpret_cast<u32x2*>(dst) = u32x2{E::pk2(acc[0] * ds_scale, acc[1] * ds_scale), E::pk2(acc[2] * ds_scale, acc[3] * ds_scale)};
  }
}

// 32-bit LDS-DMA offsets (fold_tile_dma): strides < 16 MiB and a user's rows within 4 GiB of its first row
HSTU_DEV bool dma_fast_of(const HstuAttnParams& p, const HstuAttnBwdParams& bp, int tmax) {
  const int64_t len_max = 32 * tmax;
  const int64_t q_rs = p.q_row_stride * 2, k_rs = p.k_row_stride * 2, v_rs = p.v_row_stride * 2, do_rs = bp.do_row_stride * 2;
  return FOLD_DMA_FAST && q_rs < (1 << 24) && k_rs < (1 << 24) && v_rs < (1 << 24) && do_rs < (1 << 24) && len_max * q_rs < (1LL << 32) &&
         len_max * k_rs < (1LL << 32) && len_max * v_rs < (1LL << 32) && len_max * do_rs < (1LL << 32);
}

// What a problem finds already requested by its predecessor on this workgroup: K/V tiles >= kv_lo, and (stage_first) its first
// Q/dO tile in stage `stg`.
struct W16Pre {
  int kv_lo;
  int stage_first;
};

// One (user, head) problem `uh` on the calling workgroup.  `stg`: the stage the CURRENT step's Q/dO tile lives in (toggles
// every step, across problems).
template <typename T, int D>
HSTU_DEV void w16_problem(const HstuAttnBwdParams& bp, int tmax, int uh, char* smem, int tid, int lane, int wave, int uh_next,
                          W16Pre& pre, int& stg) {
  using C = BwdCfg<T, D, D>;
  using F = W16Cfg<T, D>;
  static_assert(C::EB == 2, "16-bit I/O");
  const HstuAttnParams& p = bp.fwd;
  const int b = user_of_slot(p, uh / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * tmax);
  const W16Pre pre_in = pre;
  pre.kv_lo = F::kMaxTiles;
  pre.stage_first = 0;
  if (len <= 0) return;
  const int b3 = uh_next >= 0 ? user_of_slot(p, uh_next / p.heads) : b, hd3 = uh_next >= 0 ? uh_next % p.heads : 0;
  const int64_t off3 = uh_next >= 0 ? load_index(p.seq_offsets, b3, p.offsets_dtype) : 0;
  const int len3 = uh_next >= 0 ? min((int)(load_index(p.seq_offsets, b3 + 1, p.offsets_dtype) - off3), 32 * tmax) : 0;
  const int nt3 = (len3 + 31) >> 5;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  const int nt = (len + 31) >> 5;        // <= tmax <= 7
  char* const stage0 = smem + F::kMaxTiles * C::PAIR;
  char* const dsbuf = stage0 + 2 * C::PAIR;

  const char* qbase = (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const char* dobase = (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
  // (the next problem's base pointers are formed where they are used: eight more live SGPR pairs spill)
  auto next_kv = [&](int t, int lane_) {
    const char* kb3 = (const char*)p.k + (off3 * p.k_row_stride + (int64_t)hd3 * p.k_head_stride) * C::EB;
    const char* vb3 = (const char*)p.v + (off3 * p.v_row_stride + (int64_t)hd3 * p.v_head_stride) * C::EB;
    w16_pair_dma<T, D>(smem + t * C::PAIR, kb3, p.k_row_stride * C::EB, vb3, p.v_row_stride * C::EB, 32 * t, len3, wave, lane_, dma_fast_of(p, bp, tmax));
  };
  char* const dk_head = (char*)bp.dk + (off0 * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
  char* const dv_head = (char*)bp.dv + (off0 * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
  const int64_t dk_rs = bp.dk_row_stride * C::EB, dv_rs = bp.dv_row_stride * C::EB;
  const int64_t q_rs = p.q_row_stride * C::EB, k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB,
                do_rs = bp.do_row_stride * C::EB;
  const bool dma_fast = dma_fast_of(p, bp, tmax);
  const bool prefetch = W16_NEXT_KV && len3 > 0;

  // ---- prologue: what the predecessor has not requested -- K/V tiles below pre_in.kv_lo, the first Q/dO tile
  {
    int lane0 = lane;
    asm volatile("" : "+v"(lane0));    // (per-lane DMA plans are recomputed where they are used, not kept across the step loop)
    if (!(W16_ABLATE & 16))
      for (int t = 0; t < min(nt, pre_in.kv_lo); ++t)
        w16_pair_dma<T, D>(smem + t * C::PAIR, kbase, k_rs, vbase, v_rs, 32 * t, len, wave, lane0, dma_fast);
    if (!pre_in.stage_first) w16_pair_dma<T, D>(stage0 + stg * C::PAIR, qbase, q_rs, dobase, do_rs, 32 * (nt - 1), len, wave, lane0, dma_fast);
  }

  const int kt = wave >> 1, h = wave & 1;          // this wave's keys: [32 kt + 16 h, +16)
  f32x4 dk_acc[D / 16], dv_acc[D / 16];
#pragma unroll
  for (int d = 0; d < D / 16; ++d) {
    dk_acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv_acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float scale_v = attn_scale_of(p);
  const float ds_scale = scale_v * p.alpha;
  // lane-constant mask patterns of the plain-causal case: bits 0..7 = diagonal tile (key row <= query row), bits 8..15 = last
  // query tile (row < len); bit e = 4 qb + r <-> query row 16 qb + 4 g + r
  int dmvm = 0;
  {
    const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = 16 * (e >> 2) + 4 * g + (e & 3);
      dmvm |= (16 * h + i16 <= row ? 1 : 0) << e;
      dmvm |= (32 * (nt - 1) + row < len ? 1 : 0) << (8 + e);
    }
    dmvm |= (int)0xffff0000u;       // (`dmvm >> 8` keeps the unused high bits set)
  }

  for (int i = nt - 1; i >= 0; --i) {
    char* const cur = stage0 + stg * C::PAIR;
    char* const oth = stage0 + (stg ^ 1) * C::PAIR;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // A: Q/dO tile i (and, first step, K/V) landed; dQ GEMM of step i+1 done; copy-out of tile i+2 read
    // requests of this step, a whole step ahead of their use: the next Q/dO tile into the other stage (free since barrier B of
    // step i+1) -- at step 0 the NEXT problem's first tile --, and the next problem's K/V tile of the slot that has just been
    // copied out (tile i+2; at the first step every slot this problem does not use)
    int lane0 = lane;
    asm volatile("" : "+v"(lane0));
    if (i > 0) {
      if (!(W16_ABLATE & 4)) w16_pair_dma<T, D>(oth, qbase, q_rs, dobase, do_rs, 32 * (i - 1), len, wave, lane0, dma_fast);
    } else if (prefetch) {
      const char* qb3 = (const char*)p.q + (off3 * p.q_row_stride + (int64_t)hd3 * p.q_head_stride) * C::EB;
      const char* dob3 = (const char*)bp.dout + (off3 * bp.do_row_stride + (int64_t)hd3 * bp.do_head_stride) * C::EB;
      w16_pair_dma<T, D>(oth, qb3, q_rs, dob3, do_rs, 32 * (nt3 - 1), len3, wave, lane0, dma_fast);
      pre.stage_first = 1;
    }
    if (prefetch && !(W16_ABLATE & 16)) {
      const int t_hi = (i == nt - 1) ? nt3 - 1 : i + 2;      // first step: every tile from nt + 1 up
      for (int t = i + 2; t <= t_hi && t < nt3; ++t) next_kv(t, lane0);
    }
    if (i + 1 < nt && kt == i + 1) {
      // owner of the previous step's diagonal tile: its K tile is dead now (every dQ GEMM that reads it is done)
      int lane3 = lane;
      asm volatile("" : "+v"(lane3));
      w16_park<T, D>(dk_acc, ds_scale, smem + kt * C::PAIR, h, lane3);
    }
    // ---- phase 1: this wave's half pair of the step
    if (kt <= i && !(W16_ABLATE & 64)) {      // (no attention window here: every tile on or below the diagonal is active)
      const char* Kw = smem + kt * C::PAIR;
      int lane1 = lane;
      asm volatile("" : "+v"(lane1));
      w16_pair<T, D>(p, mc, Kw, Kw + C::KT, cur, cur + C::KT, dsbuf + kt * F::DSB, 32 * i, 32 * kt, h, dk_acc, dv_acc, lane1, dmvm);
    }
    __syncthreads();   // B: dS' of this step published; stage reads done; dK of tile i+1 parked
    if (i + 1 < nt) {
      const int t1 = i + 1;
      int tid1 = tid;
      asm volatile("" : "+v"(tid1));
      w16_copy_out<T, D>(smem + t1 * C::PAIR, dk_head + (int64_t)(32 * t1) * dk_rs, dk_rs, dv_head + (int64_t)(32 * t1) * dv_rs, dv_rs,
                         len - 32 * t1, tid1);
    }
    // ---- phase 2: dQ of query tile i
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    if (!(W16_ABLATE & 32)) w16_dq_phase<T, D>(bp, mc, smem, dsbuf, i, wave, off0, hd, ds_scale, lane2);
    if (kt == i) {
      // key tile i is final (no earlier query tile reaches it).  V tiles are only read by their owners' pairs, and this was
      // their last one: dV is parked right away; the K tile is still read by this step's dQ GEMM: dK follows after barrier A
      int lane3 = lane;
      asm volatile("" : "+v"(lane3));
      w16_park<T, D>(dv_acc, scale_v, smem + kt * C::PAIR + C::KT, h, lane3);
    }
    stg ^= 1;
  }
  // ---- tile 0: dV parked, dK still in the registers of waves 0 and 1
  __syncthreads();     // C: every dQ GEMM is done (K tile 0 is dead); copy-out of tile 1 has read its slot
  if (prefetch && !(W16_ABLATE & 16) && 1 < nt3) {
    int lane0 = lane;
    asm volatile("" : "+v"(lane0));
    next_kv(1, lane0);
  }
  if (kt == 0) {
    int lane4 = lane;
    asm volatile("" : "+v"(lane4));
    w16_park<T, D>(dk_acc, ds_scale, smem, h, lane4);
  }
  __syncthreads();     // D
  {
    int tid1 = tid;
    asm volatile("" : "+v"(tid1));
    w16_copy_out<T, D>(smem, dk_head, dk_rs, dv_head, dv_rs, len, tid1);
  }
  if (prefetch) pre.kv_lo = 1;         // (tiles 2 .. nt3-1 during the steps, tile 1 just now)
}

template <typename T, int D>
__global__ __launch_bounds__(kW16Threads) __attribute__((amdgpu_waves_per_eu(4, 4))) void hstu_attn_bwd_w16_kernel(const HstuAttnBwdParams bp, int tmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int total = bp.fwd.batch * bp.fwd.heads;
  W16Pre pre = {W16Cfg<T, D>::kMaxTiles, 0};
  int stg = 0;
  for (int uh = blockIdx.x; uh < total; uh += gridDim.x) {
    int uh_l = uh;
    asm volatile("" : "+s"(uh_l));     // nothing of problem i+1 is hoisted into problem i
    const int uh_n = (uh_l + (int)gridDim.x < total) ? uh_l + (int)gridDim.x : -1;
    w16_problem<T, D>(bp, tmax, uh_l, smem, tid, lane, wave, uh_n, pre, stg);
    __syncthreads();                   // E: the last copy-out has read its slot before the next prologue's DMA lands
  }
}

}  // namespace hstu
