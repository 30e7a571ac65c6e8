// HSTU attention for SHORT sequences (max_seq_len <= 64, head dims <= 32): one WAVE per (user, head) problem.
//
// Why.  Long-tailed batches (Amazon-Books: N = 61, 4 heads of 16, 95 % of the users below 30 rows) are all fixed cost
// for the other kernels: a workgroup of 4 or 8 waves, a K/V ring or block, barriers between phases -- per (user, head)
// of a dozen rows.  Here a wave owns a whole problem (at most 2 tiles of 32 rows): it stages q, k, v (dO) in its
// PRIVATE slice of LDS, runs every (query tile, key tile) pair, and writes its rows -- no barrier anywhere, waves never
// wait for each other, and a workgroup of four waves walks the problems four at a time (persistent: slot s takes
// problems s, s + slots, ...).  Staging goes through registers with zero fill (rows past the sequence, columns past
// the real head dim), so head dims 8..32 share one instantiation and nothing needs a clamp.
// Same fragments, LDS tile layout and element-wise math as the other kernels (S accumulator start value for masks).
#pragma once
#include "hstu_attn_bwd_quad.cuh"

namespace hstu {

constexpr int kSoloWaves = 4;
constexpr int kSoloThreads = 256;
constexpr int kSoloD = 32;          // instantiated head dim (16-bit: 64-byte rows, 2 KiB tiles)
constexpr int kSoloMaxLen = 64;

template <typename T>
struct SoloCfg {
  static constexpr int TILE = 32 * kSoloD * Elem<T>::kBytes;      // 2048
  static constexpr int UPR = kSoloD * Elem<T>::kBytes / 16;        // 4
  // per wave: K, V, Q (2 tiles each; forward: 12 KiB -> 3 workgroups per CU) + dO and 2 dS' tiles (backward: 20 KiB)
  static constexpr int bwd_slice() { return 8 * TILE + 2 * 32 * 64; }
  static constexpr int fwd_slice() { return 6 * TILE; }   // the output tile is parked over the query tile it came from
};

// [rows of one tensor] -> this wave's LDS tiles, zero filled: 2 units (16 B) per lane and tile.  Two halves: the
// loads (issue) and the LDS writes (commit), so that all the loads of a problem are in flight before the first write.
template <typename T>
HSTU_DEV void solo_issue(u32x4 (&reg)[4], const char* base, int64_t row_stride_bytes, int len, int real_d, int nt, int lane) {
  using S = SoloCfg<T>;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                   // j = 2 * tile + half
    const int u = (j & 1) * 64 + lane, row = 32 * (j >> 1) + u / S::UPR, unit = u % S::UPR;
    const bool ok = (j >> 1) < nt && row < len && unit * EPU < real_d;
    reg[j] = gload16(base + (int64_t)(ok ? row : 0) * row_stride_bytes + (ok ? unit : 0) * 16);
  }
}
template <typename T>
HSTU_DEV void solo_commit(const u32x4 (&reg)[4], char* tiles, int len, int real_d, int nt, int lane) {
  using S = SoloCfg<T>;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = (j & 1) * 64 + lane, row = 32 * (j >> 1) + u / S::UPR, unit = u % S::UPR;
    const bool ok = (j >> 1) < nt && row < len && unit * EPU < real_d;
    *LDS_PTR(u32x4, tiles + (j >> 1) * S::TILE + tile_off<S::UPR>(u / S::UPR, unit)) = ok ? reg[j] : u32x4{0u, 0u, 0u, 0u};
  }
}
template <typename T>
HSTU_DEV void solo_stage(char* tiles, const char* base, int64_t row_stride_bytes, int len, int real_d, int nt, int lane) {
  u32x4 reg[4];
  solo_issue<T>(reg, base, row_stride_bytes, len, real_d, nt, lane);
  solo_commit<T>(reg, tiles, len, real_d, nt, lane);
}

// rows of a parked [32][32] tile -> global (only the real head dim's units)
template <typename T>
HSTU_DEV void solo_copy_out(const char* tile, char* gtile, int64_t row_stride_bytes, int rows_valid, int real_d, int lane) {
  using S = SoloCfg<T>;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = j * 64 + lane, row = u / S::UPR, unit = u % S::UPR;
    const u32x4 v = *LDS_PTR(const u32x4, tile + tile_off<S::UPR>(row, unit));
    if (row < rows_valid && unit * EPU < real_d) gstore16(gtile + row * row_stride_bytes + unit * 16, v);
  }
}

// 16-bit mask of the pair (query tile i0, key tile j0) for the lane layout "registers = rows, lanes = columns" of a
// 32x32 C tile: bit r = element (row i0/j0 + (r&3) + 8 (r>>2) + 4 hf, column n32) survives.  `rows_are_keys`: forward
// (S^T: rows = keys, lanes = queries); otherwise rows = queries, lanes = keys (backward, fold_pair computes its own).
HSTU_DEV int solo_mask_bits_fwd(const MaskCtx& mc, int i0, int j0, int lane) {
  const int n32 = lane & 31, hf = lane >> 5;
  const int qi = i0 + n32;
  const int qi_id = mc.id_of(qi);
  int km = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
    const bool ok = (qi < mc.len) & (key < mc.len) & mc.valid_ids(qi, key, qi_id, mc.id_of(key));
    km |= (ok ? 1 : 0) << r;
  }
  return km;
}

// ------------------------------------------------------------------------------------------------ forward
struct SoloProb { int b, hd, len; int64_t off0; };
HSTU_DEV SoloProb solo_prob(const HstuAttnParams& p, int uh) {
  SoloProb r;
  r.b = user_of_slot(p, uh / p.heads);
  r.hd = uh % p.heads;
  r.off0 = load_index(p.seq_offsets, r.b, p.offsets_dtype);
  r.len = min((int)(load_index(p.seq_offsets, r.b + 1, p.offsets_dtype) - r.off0), kSoloMaxLen);
  return r;
}
template <typename T>
HSTU_DEV void solo_fwd_issue(const HstuAttnParams& p, const SoloProb& pr, u32x4 (&rk)[4], u32x4 (&rv)[4], u32x4 (&rq)[4], int lane) {
  constexpr int EB = Elem<T>::kBytes;
  const int nt = (pr.len + 31) >> 5;
  solo_issue<T>(rk, (const char*)p.k + (pr.off0 * p.k_row_stride + (int64_t)pr.hd * p.k_head_stride) * EB, p.k_row_stride * EB, pr.len, p.dqk, nt, lane);
  solo_issue<T>(rv, (const char*)p.v + (pr.off0 * p.v_row_stride + (int64_t)pr.hd * p.v_head_stride) * EB, p.v_row_stride * EB, pr.len, p.dv, nt, lane);
  solo_issue<T>(rq, (const char*)p.q + (pr.off0 * p.q_row_stride + (int64_t)pr.hd * p.q_head_stride) * EB, p.q_row_stride * EB, pr.len, p.dqk, nt, lane);
}

// everything after the staging: the pairs of one problem from the wave's LDS slice, rows out
template <typename T>
HSTU_DEV void solo_fwd_compute(const HstuAttnParams& p, const SoloProb& pr, char* slice, int lane) {
  using S = SoloCfg<T>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  constexpr int EB = E::kBytes;
  const int b = pr.b, hd = pr.hd, len = pr.len;
  const int64_t off0 = pr.off0;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  const int nt = (len + 31) >> 5;
  const int n32 = lane & 31, hf = lane >> 5;
  char* Kt = slice, *Vt = slice + 2 * S::TILE, *Qt = slice + 4 * S::TILE;
  const float scale_v = attn_scale_of(p);
  const unsigned neg = __builtin_bit_cast(unsigned, p.alpha < 0.f ? 1e30f : -1e30f);
  for (int i = 0; i < nt; ++i) {
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    // contextual rows (id 0) see every non-target key, also the ones after them
    const int t_hi = (mc.ctx > 0 && 32 * i < mc.ctx) ? nt - 1 : i;
    for (int t = 0; t <= t_hi; ++t) {
      if (!mc.simple && !mc.pair_may_be_active(32 * i, 32, 32 * t, 32)) continue;
      // S^T[key][q] = K_t Q_i^T, masked elements start at -1e30 (sigmoid -> exactly 0)
      const unsigned nk = ~(unsigned)solo_mask_bits_fwd(mc, 32 * i, 32 * t, lane);
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = __builtin_bit_cast(float, (unsigned)(((int)(nk << (31 - r))) >> 31) & neg);
#pragma unroll
      for (int kg = 0; kg < kSoloD / 16; ++kg) {
        const Frag a = lds_row_frag<T, S::UPR>(Kt + t * S::TILE, n32, hf * (kSoloD / 2) + kg * 8);
        const Frag bq = lds_row_frag<T, S::UPR>(Qt + i * S::TILE, n32, hf * (kSoloD / 2) + kg * 8);
        s = E::mma(a, bq, s);
      }
      Frag pb[2];
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = s[8 * h8 + j] * p.alpha;
          pv[j] = x * fast_sigmoid(x);
        }
        pb[h8] = E::pack8(pv);
      }
      // O^T[dv][q] += V_t^T[dv][key] P'^T[key][q]
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const Frag a = lds_col_frag<T, S::UPR>(Vt + t * S::TILE, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 0, lane);
        oacc = E::mma(a, pb[ks], oacc);
      }
    }
    // C layout: column n32 = query row, registers = features 8 rq + 4 hf + (0..3): park as a row-major tile (over Q_i,
    // which no later pair reads), copy rows out
    char* Ot = Qt + i * S::TILE;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      u32x2 v = {E::pk2(oacc[4 * rq] * scale_v, oacc[4 * rq + 1] * scale_v), E::pk2(oacc[4 * rq + 2] * scale_v, oacc[4 * rq + 3] * scale_v)};
      *LDS_PTR(u32x2, Ot + tile_off<S::UPR>(n32, rq) + 8 * hf) = v;
    }
    char* obase = (char*)p.out + ((off0 + 32 * i) * p.o_row_stride + (int64_t)hd * p.o_head_stride) * EB;
    solo_copy_out<T>(Ot, obase, p.o_row_stride * EB, len - 32 * i, p.dv, lane);
  }
}

template <typename T>
__global__ __launch_bounds__(kSoloThreads) void hstu_attn_fwd_solo_kernel(const HstuAttnParams p) {
  using S = SoloCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slice = smem + wave * S::fwd_slice();
  const int total = p.batch * p.heads;
  for (int uh = blockIdx.x * kSoloWaves + wave; uh < total; uh += gridDim.x * kSoloWaves) {
    int uh_l = uh;
    asm volatile("" : "+s"(uh_l));
    const SoloProb cur = solo_prob(p, uh_l);
    if (cur.len <= 0) continue;
    // (issuing the NEXT problem's loads before computing this one -- 48 more live registers -- was measured: 56 -> 59 us
    // on the Amazon-Books batch; a wave's time per problem is instruction issue, not the HBM round trip)
    u32x4 rk[4], rv[4], rq[4];
    solo_fwd_issue<T>(p, cur, rk, rv, rq, lane);
    const int nt = (cur.len + 31) >> 5;
    solo_commit<T>(rk, slice, cur.len, p.dqk, nt, lane);
    solo_commit<T>(rv, slice + 2 * S::TILE, cur.len, p.dv, nt, lane);
    solo_commit<T>(rq, slice + 4 * S::TILE, cur.len, p.dqk, nt, lane);
    solo_fwd_compute<T>(p, cur, slice, lane);
  }
}

// ------------------------------------------------------------------------------------------------ backward
// dQ of query tile qt (rows 16 qb .. +16) from the wave's own dS' tiles: two interleaved 16x16x32 MFMAs per key tile
// (quad_dq_phase with one feature block of 32)
template <typename T>
HSTU_DEV void solo_dq(const HstuAttnBwdParams& bp, const MaskCtx& mc, const char* Kt, const char* ds, int qt, int qb, int64_t off0, int hd,
                      float ds_scale, int lane) {
  using S = SoloCfg<T>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  const int i16 = lane & 15, g = lane >> 4;
  const int row_lo = 8 * g + (i16 >> 2), row_hi = row_lo + 4;
  const int colK0 = 8 * (i16 & 3), colK1 = colK0 + 4;
  const int k0_lo = tile_off<S::UPR>(row_lo, colK0 >> 3) + ((colK0 & 7) << 1), k0_hi = tile_off<S::UPR>(row_hi, colK0 >> 3) + ((colK0 & 7) << 1);
  const int k1_lo = tile_off<S::UPR>(row_lo, colK1 >> 3) + ((colK1 & 7) << 1), k1_hi = tile_off<S::UPR>(row_hi, colK1 >> 3) + ((colK1 & 7) << 1);
  const int d_lo = fold_ds_off(row_lo, 4 * qb + (i16 & 3)), d_hi = fold_ds_off(row_hi, 4 * qb + (i16 & 3));
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int t = 0; t <= qt; ++t) {
    if (mc.win != 0 && !mc.pair_may_be_active(32 * qt, 32, 32 * t, 32)) continue;
    const Frag fk0 = tr_frag16<T>(Kt + t * S::TILE, k0_lo, k0_hi), fk1 = tr_frag16<T>(Kt + t * S::TILE, k1_lo, k1_hi);
    const Frag fd = tr_frag16<T>(ds + t * 32 * 64, d_lo, d_hi);
    acc[0] = E::mma16(fk0, fd, acc[0]);
    acc[1] = E::mma16(fk1, fd, acc[1]);
  }
  const int qrow = 32 * qt + 16 * qb + i16;
  if (qrow < mc.len && 8 * g < bp.fwd.dqk) {
    char* dqrow = (char*)bp.dq + ((off0 + qrow) * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * E::kBytes;
    u32x4 v = {E::pk2(acc[0][0] * ds_scale, acc[0][1] * ds_scale), E::pk2(acc[0][2] * ds_scale, acc[0][3] * ds_scale),
               E::pk2(acc[1][0] * ds_scale, acc[1][1] * ds_scale), E::pk2(acc[1][2] * ds_scale, acc[1][3] * ds_scale)};
    gstore16(dqrow + 8 * g * E::kBytes, v);
  }
}

template <typename T>
HSTU_DEV void solo_bwd_problem(const HstuAttnBwdParams& bp, int uh, char* slice, int lane) {
  using S = SoloCfg<T>;
  using C = BwdCfg<T, kSoloD, kSoloD>;
  constexpr int EB = Elem<T>::kBytes;
  const HstuAttnParams& p = bp.fwd;
  const int b = user_of_slot(p, uh / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), kSoloMaxLen);
  if (len <= 0) return;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  HSTU_TRACE_DECL(bp.workspace, false);
  const int nt = (len + 31) >> 5;
  char* Kt = slice, *Vt = slice + 2 * S::TILE, *Qt = slice + 4 * S::TILE, *dOt = slice + 6 * S::TILE, *ds = slice + 8 * S::TILE;
  solo_stage<T>(Kt, (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * EB, p.k_row_stride * EB, len, p.dqk, nt, lane);
  solo_stage<T>(Vt, (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * EB, p.v_row_stride * EB, len, p.dv, nt, lane);
  solo_stage<T>(Qt, (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * EB, p.q_row_stride * EB, len, p.dqk, nt, lane);
  solo_stage<T>(dOt, (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * EB, bp.do_row_stride * EB, len, p.dv, nt, lane);
  f32x16 dk0[1], dv0[1], dk1[1], dv1[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[0][r] = 0.f; dv0[0][r] = 0.f; dk1[0][r] = 0.f; dv1[0][r] = 0.f; }
  const float scale_v = attn_scale_of(p);
  const float ds_scale = scale_v * p.alpha;
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
  for (int i = nt - 1; i >= 0; --i) {
    if (i >= 1 && (mc.win == 0 || mc.pair_may_be_active(32 * i, 32, 32, 32)))
      fold_pair<T, kSoloD, kSoloD>(p, mc, Kt + S::TILE, Vt + S::TILE, Qt + i * S::TILE, dOt + i * S::TILE, ds + 32 * 64, 32 * i, 32, dk1, dv1, lane,
                                   dmvm HSTU_TRACE_PASS);
    if (mc.win == 0 || mc.pair_may_be_active(32 * i, 32, 0, 32))
      fold_pair<T, kSoloD, kSoloD>(p, mc, Kt, Vt, Qt + i * S::TILE, dOt + i * S::TILE, ds, 32 * i, 0, dk0, dv0, lane, dmvm HSTU_TRACE_PASS);
    solo_dq<T>(bp, mc, Kt, ds, i, 0, off0, hd, ds_scale, lane);
    solo_dq<T>(bp, mc, Kt, ds, i, 1, off0, hd, ds_scale, lane);
  }
  // dk / dv: park over the dead K / V tiles, copy the rows out
  char* const dk_head = (char*)bp.dk + (off0 * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * EB;
  char* const dv_head = (char*)bp.dv + (off0 * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * EB;
  fold_park_tile<T, kSoloD>(dk0, ds_scale, Kt, lane);
  fold_park_tile<T, kSoloD>(dv0, scale_v, Vt, lane);
  solo_copy_out<T>(Kt, dk_head, bp.dk_row_stride * EB, len, p.dqk, lane);
  solo_copy_out<T>(Vt, dv_head, bp.dv_row_stride * EB, len, p.dv, lane);
  if (nt > 1) {
    fold_park_tile<T, kSoloD>(dk1, ds_scale, Kt + S::TILE, lane);
    fold_park_tile<T, kSoloD>(dv1, scale_v, Vt + S::TILE, lane);
    solo_copy_out<T>(Kt + S::TILE, dk_head + 32 * bp.dk_row_stride * EB, bp.dk_row_stride * EB, len - 32, p.dqk, lane);
    solo_copy_out<T>(Vt + S::TILE, dv_head + 32 * bp.dv_row_stride * EB, bp.dv_row_stride * EB, len - 32, p.dv, lane);
  }
  (void)C::EB;
}

template <typename T>
__global__ __launch_bounds__(kSoloThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void hstu_attn_bwd_solo_kernel(const HstuAttnBwdParams bp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slice = smem + wave * SoloCfg<T>::bwd_slice();
  const int total = bp.fwd.batch * bp.fwd.heads;
  for (int uh = blockIdx.x * kSoloWaves + wave; uh < total; uh += gridDim.x * kSoloWaves) {
    int uh_l = uh;
    asm volatile("" : "+s"(uh_l));
    solo_bwd_problem<T>(bp, uh_l, slice, lane);
  }
}

}  // namespace hstu
