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
  static constexpr int bwd_slice(int tpt = 2) { return tpt * (4 * TILE + 32 * 64); }
  static constexpr int fwd_slice(int tpt = 2) { return 3 * tpt * TILE; }   // the output tile is parked over the query tile it came from
};

// [rows of one tensor] -> this wave's LDS tiles, zero filled: 2 units (16 B) per lane and tile.  Two halves: the
// loads (issue) and the LDS writes (commit), so that all the loads of a problem are in flight before the first write.
template <typename T, int TPT = 2>
HSTU_DEV void solo_issue(u32x4 (&reg)[2 * TPT], const char* base, int64_t row_stride_bytes, int len, int real_d, int nt, int lane) {
  using S = SoloCfg<T>;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int j = 0; j < 2 * TPT; ++j) {             // j = 2 * tile + half
    const int u = (j & 1) * 64 + lane, row = 32 * (j >> 1) + u / S::UPR, unit = u % S::UPR;
    const bool ok = (j >> 1) < nt && row < len && unit * EPU < real_d;
    reg[j] = gload16(base + (int64_t)(ok ? row : 0) * row_stride_bytes + (ok ? unit : 0) * 16);
  }
}
// TPT: tiles per tensor of the slice (2; 1 in the launches that take users of <= 32 rows only: a tensor's second tile does not exist there)
template <typename T, int TPT = 2>
HSTU_DEV void solo_commit(const u32x4 (&reg)[2 * TPT], char* tiles, int len, int real_d, int nt, int lane) {
  using S = SoloCfg<T>;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int j = 0; j < 2 * TPT; ++j) {
    const int u = (j & 1) * 64 + lane, row = 32 * (j >> 1) + u / S::UPR, unit = u % S::UPR;
    const bool ok = (j >> 1) < nt && row < len && unit * EPU < real_d;
    *LDS_PTR(u32x4, tiles + (j >> 1) * S::TILE + tile_off<S::UPR>(u / S::UPR, unit)) = ok ? reg[j] : u32x4{0u, 0u, 0u, 0u};
  }
}
template <typename T, int TPT = 2>
HSTU_DEV void solo_stage(char* tiles, const char* base, int64_t row_stride_bytes, int len, int real_d, int nt, int lane) {
  u32x4 reg[2 * TPT];
  solo_issue<T, TPT>(reg, base, row_stride_bytes, len, real_d, nt, lane);
  solo_commit<T, TPT>(reg, tiles, len, real_d, nt, lane);
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
  if (mc.simple) {
    // plain causal (wave-uniform): bit r = key j0 + rho(r) <= query AND key < len AND query < len, rho(r) = (r&3) + 8 (r>>2) + 4 hf.
    // The keys of a register set come in groups of four, eight apart: "rho(r) < m" is a run of whole groups and a partial one --
    // a dozen instructions instead of the general predicate's ~250 (a third of a one-pair problem's instructions; Amazon-Books:
    // every problem is one or three pairs).
    const int m = min(min(qi + 1, mc.len) - j0, 32) - 4 * hf;       // keys of this lane's half below m survive
    const int g = m >> 3, rem = min(m & 7, 4);
    const int run = m <= 0 ? 0 : ((1 << (4 * g)) - 1) | (((1 << rem) - 1) << (4 * g));
    return qi < mc.len ? (run & 0xffff) : 0;
  }
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
template <typename T, int TPT = 2>
HSTU_DEV void solo_fwd_issue(const HstuAttnParams& p, const SoloProb& pr, u32x4 (&rk)[2 * TPT], u32x4 (&rv)[2 * TPT], u32x4 (&rq)[2 * TPT], int lane) {
  constexpr int EB = Elem<T>::kBytes;
  const int nt = (pr.len + 31) >> 5;
  solo_issue<T, TPT>(rk, (const char*)p.k + (pr.off0 * p.k_row_stride + (int64_t)pr.hd * p.k_head_stride) * EB, p.k_row_stride * EB, pr.len, p.dqk, nt, lane);
  solo_issue<T, TPT>(rv, (const char*)p.v + (pr.off0 * p.v_row_stride + (int64_t)pr.hd * p.v_head_stride) * EB, p.v_row_stride * EB, pr.len, p.dv, nt, lane);
  solo_issue<T, TPT>(rq, (const char*)p.q + (pr.off0 * p.q_row_stride + (int64_t)pr.hd * p.q_head_stride) * EB, p.q_row_stride * EB, pr.len, p.dqk, nt, lane);
}

// Time buckets of one user as BYTES in LDS, computed once per user by the workgroup's four waves and read by every head:
// 512 bytes per half tile (8 per lane), half tile (pair, h8) at ((pair * 2 + h8) * 512), pair = i (i + 1) / 2 + t for
// query tile i, key tile t <= i.  Two layouts, each what its consumer's registers hold: forward S^T (lane = query,
// registers = keys), backward S (lane = key, registers = queries: the layout of fold_pair_x's byte cache).
constexpr int kSoloBucketBytes = 3 * 1024;
template <bool BWD>
HSTU_DEV void solo_bucket_bytes(const BiasCtx& bc, char* bcache, int len, int wave, int lane, int nwaves = kSoloWaves) {
  const int n32 = lane & 31, hf = lane >> 5;
  const int npairs = len > 32 ? 3 : 1;
  for (int item = wave; item < 2 * npairs; item += nwaves) {     // wave-uniform
    const int pair = item >> 1, h8 = item & 1;
    const int i = pair == 0 ? 0 : 1, t = pair == 2 ? 1 : 0;
    int bkt[8];
    if constexpr (BWD) {
      const int key = 32 * t + n32;
      if (bc.small) {
        const int t_k32 = bc.t32_at(key);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const auto t4 = bc.t32x4_next(32 * i + 8 * (2 * h8 + g) + 4 * hf);
#pragma unroll
          for (int j = 0; j < 4; ++j) bkt[4 * g + j] = bc.bucket32(t4[j], t_k32);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * h8 + j;
          bkt[j] = bc.bucket(bc.ts_at(32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf + 1), bc.ts_at(key));
        }
      }
    } else {
      const int qi = 32 * i + n32;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * h8 + j;
        const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;
        bkt[j] = bc.small ? bc.bucket32(bc.t32_at(qi + 1), bc.t32_at(key)) : bc.bucket(bc.ts_at(qi + 1), bc.ts_at(key));
      }
    }
    u32x2 w = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j >> 2] |= (unsigned)bkt[j] << (8 * (j & 3));
    *LDS_PTR(u32x2, bcache + item * 512 + 8 * lane) = w;
  }
}

// everything after the staging: the pairs of one problem from the wave's LDS slice, rows out.  BIAS: the research path's
// relative position / time-bucket term (hstu_attn_fwd.cuh, BIAS) from the workgroup's staged tables.
struct SoloNoBias {};
template <typename T, bool BIAS = false, typename BC = SoloNoBias, int TPT = 2>
HSTU_DEV void solo_fwd_compute(const HstuAttnParams& p, const SoloProb& pr, char* slice, int lane, const BC& bc = BC(),
                               const char* bcache = nullptr) {
  using S = SoloCfg<T>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  constexpr int EB = E::kBytes;
  const int b = pr.b, hd = pr.hd, len = pr.len;
  const int64_t off0 = pr.off0;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  const int nt = TPT == 1 ? 1 : (len + 31) >> 5;
  const int n32 = lane & 31, hf = lane >> 5;
  char* Kt = slice, *Vt = slice + TPT * S::TILE, *Qt = slice + 2 * TPT * S::TILE;
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
        [[maybe_unused]] u32x2 bw = {0u, 0u};
        if constexpr (BIAS) bw = *LDS_PTR(const u32x2, bcache + ((((i * (i + 1)) >> 1) + t) * 2 + h8) * 512 + 8 * lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float x = s[8 * h8 + j] * p.alpha;
          if constexpr (BIAS) {
            const int r = 8 * h8 + j;
            const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const int bkt = (int)((bw[j >> 2] >> (8 * (j & 3))) & 255u);
            x += bc.value(bc.pos_index(32 * i + n32, key), bkt);
          }
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

// TPT = 1: the launch that takes the problems of <= 32 rows only (slices of one tile per tensor: 6 KiB per wave instead of 12 -- as many
// workgroups per CU as the registers allow); the other launch takes the longer ones
template <typename T, int TPT>
__global__ __launch_bounds__(kSoloThreads) void hstu_attn_fwd_solo_kernel(const HstuAttnParams p, int len_lo, int len_hi) {
  using S = SoloCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slice = smem + wave * S::fwd_slice(TPT);
  const int total = p.batch * p.heads;
  for (int uh = blockIdx.x * kSoloWaves + wave; uh < total; uh += gridDim.x * kSoloWaves) {
    int uh_l = uh;
    asm volatile("" : "+s"(uh_l));
    const SoloProb cur = solo_prob(p, uh_l);
    if (cur.len <= len_lo || cur.len > len_hi) continue;      // (empty, or the other launch's)
    // (issuing the NEXT problem's loads before computing this one -- 48 more live registers -- was measured: 56 -> 59 us
    // on the Amazon-Books batch; a wave's time per problem is instruction issue, not the HBM round trip)
    u32x4 rk[2 * TPT], rv[2 * TPT], rq[2 * TPT];
    solo_fwd_issue<T, TPT>(p, cur, rk, rv, rq, lane);
    const int nt = (cur.len + 31) >> 5;
    solo_commit<T, TPT>(rk, slice, cur.len, p.dqk, nt, lane);
    solo_commit<T, TPT>(rv, slice + TPT * S::TILE, cur.len, p.dv, nt, lane);
    solo_commit<T, TPT>(rq, slice + 2 * TPT * S::TILE, cur.len, p.dqk, nt, lane);
    solo_fwd_compute<T, false, SoloNoBias, TPT>(p, cur, slice, lane);
  }
}

// Research path (relative position + time bias) at the short-sequence shapes (Amazon-Books: N = 61, 4 heads of 16).
// Helpers of the BACKWARD kernel's software pipeline over the users (further down): the user part of stage_bias_tables
// split into its loads (registers) and its LDS writes.
struct SoloTs { int64_t t, tn, t0, tl; };     // thread i: t[i], t[i + 1], t[0], t[n - 1] of the user's timestamp row (clamped)
HSTU_DEV void solo_ts_issue(const HstuAttnParams& p, int b, int tid, SoloTs& r) {
  const int64_t* ts_row = bias_ts_row(p, b);
  if (!ts_row) return;                          // (kernel-uniform)
  const int n = p.max_seq_len;
  r.t = ts_row[min(tid, n - 1)];
  r.tn = ts_row[min(tid + 1, n - 1)];
  r.t0 = ts_row[0];
  r.tl = ts_row[n - 1];
}
// the user part of stage_bias_tables from registers (max_seq_len + 32 <= the workgroup's threads: thread i writes entry i)
HSTU_DEV BiasCtx solo_ts_commit(const HstuAttnParams& p, char* lds, int tid, const SoloTs& r) {
  const bool has_ts = p.ts_w && p.timestamps;
  const BiasCtx c = bias_ctx_at(p, has_ts, lds);
  if (has_ts) {
    const int n = p.max_seq_len;
    bool big = false;
    if (tid < c.npad) {
      const int64_t o = r.t - r.t0;
      if (tid < n) {
        *LDS_PTR(int64_t, c.ltime + 8 * tid) = r.t;
        big = o >= (1LL << 30) || o <= -(1LL << 30);
      }
      *LDS_PTR(int, c.lt32 + 4 * tid) = (int)o;                                   // (entries >= n: the last timestamp, clamped above)
      *LDS_PTR(int, c.lt32 + 4 * (c.npad + tid)) = (int)(r.tn - r.t0);           // shifted copy: entry i = t[i + 1]
    }
    const bool wave_big = __builtin_amdgcn_ballot_w64(big) != 0;
    if ((tid & 63) == 0) *LDS_PTR(int, c.lt32 + 4 * (2 * c.npad + (tid >> 6))) = wave_big ? 1 : 0;
  }
  return c;
}
HSTU_DEV SoloProb solo_user(const HstuAttnParams& p, int u, int hd) {      // u clamped: every load is a valid one
  SoloProb r;
  r.b = user_of_slot(p, min(u, p.batch - 1));
  r.hd = hd;
  r.off0 = load_index(p.seq_offsets, r.b, p.offsets_dtype);
  r.len = min((int)(load_index(p.seq_offsets, r.b + 1, p.offsets_dtype) - r.off0), kSoloMaxLen);
  return r;
}

// Forward: NOTHING is shared between the waves.  A wave owns a USER -- its own copy of the tables, its own bucket bytes
// (computed once, read by every head), its slice -- and walks the user's heads, the next head's rows requested under the
// current head's pairs; no barrier anywhere, the waves of a CU drift apart and fill each other's waits (as in the plain
// short-sequence kernel).  LDS per wave: slice + tables + 3 KiB of bucket bytes.  (A workgroup per user with its waves on
// the heads, software-pipelined over the users as the backward below, was measured too: 143 vs 135 us on the Amazon-Books
// batch.  By removal, of those 135-144 us the pairs are 71, the bucket bytes 37, the row loads 18, the rest 24.)
#ifndef SOLO_FWD_BIAS_SHORT_WAVES
#define SOLO_FWD_BIAS_SHORT_WAVES 3
#endif
template <typename T, int TPT>
__global__ __launch_bounds__(kSoloThreads) __attribute__((amdgpu_waves_per_eu(TPT == 1 ? SOLO_FWD_BIAS_SHORT_WAVES : 2, TPT == 1 ? SOLO_FWD_BIAS_SHORT_WAVES : 2)))
void hstu_attn_fwd_solo_bias_kernel(const HstuAttnParams p, int table_bytes, int len_lo, int len_hi) {
  using S = SoloCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per_wave = S::fwd_slice(TPT) + table_bytes + (TPT == 1 ? 1024 : kSoloBucketBytes);
  char* slice = smem + wave * per_wave;
  char* const tables = slice + S::fwd_slice(TPT);
  char* const bcache = tables + table_bytes;
  bool first = true;
  for (int u = blockIdx.x * kSoloWaves + wave; u < p.batch; u += gridDim.x * kSoloWaves) {
    int u_l = u;
    asm volatile("" : "+s"(u_l));
    SoloProb cur = solo_user(p, u_l, 0);
    if (cur.len <= len_lo || cur.len > len_hi) continue;      // (empty, or the other launch's)
    u32x4 rk[2 * TPT], rv[2 * TPT], rq[2 * TPT];
    solo_fwd_issue<T, TPT>(p, cur, rk, rv, rq, lane);                // head 0: under the staging of the user's tables
    BiasCtx bc = stage_bias_tables(p, cur.b, tables, lane, 64, !first);
    first = false;
    bc.finish(1);                                                    // (LDS operations of a wave complete in order: no barrier)
    if (bc.lts) solo_bucket_bytes<false>(bc, bcache, cur.len, 0, lane, 1);
    const int nt = (cur.len + 31) >> 5;
    for (int hd = 0; hd < p.heads; ++hd) {
      cur.hd = hd;
      solo_commit<T, TPT>(rk, slice, cur.len, p.dqk, nt, lane);
      solo_commit<T, TPT>(rv, slice + TPT * S::TILE, cur.len, p.dv, nt, lane);
      solo_commit<T, TPT>(rq, slice + 2 * TPT * S::TILE, cur.len, p.dqk, nt, lane);
      if (hd + 1 < p.heads) {                                        // the next head's rows: in flight during this head's pairs
        SoloProb nx = cur;
        nx.hd = hd + 1;
        solo_fwd_issue<T, TPT>(p, nx, rk, rv, rq, lane);
      }
      solo_fwd_compute<T, true, BiasCtx, TPT>(p, cur, slice, lane, bc, bcache);
    }
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

// the four row blocks of one (user, head): requested (registers) / written to the wave's slice
template <typename T, int TPT = 2>
HSTU_DEV void solo_bwd_issue(const HstuAttnBwdParams& bp, const SoloProb& pr, u32x4 (&rk)[2 * TPT], u32x4 (&rv)[2 * TPT], u32x4 (&rq)[2 * TPT],
                             u32x4 (&rd)[2 * TPT], int lane) {
  constexpr int EB = Elem<T>::kBytes;
  const HstuAttnParams& p = bp.fwd;
  const int nt = (pr.len + 31) >> 5;
  solo_issue<T, TPT>(rk, (const char*)p.k + (pr.off0 * p.k_row_stride + (int64_t)pr.hd * p.k_head_stride) * EB, p.k_row_stride * EB, pr.len, p.dqk, nt, lane);
  solo_issue<T, TPT>(rv, (const char*)p.v + (pr.off0 * p.v_row_stride + (int64_t)pr.hd * p.v_head_stride) * EB, p.v_row_stride * EB, pr.len, p.dv, nt, lane);
  solo_issue<T, TPT>(rq, (const char*)p.q + (pr.off0 * p.q_row_stride + (int64_t)pr.hd * p.q_head_stride) * EB, p.q_row_stride * EB, pr.len, p.dqk, nt, lane);
  solo_issue<T, TPT>(rd, (const char*)bp.dout + (pr.off0 * bp.do_row_stride + (int64_t)pr.hd * bp.do_head_stride) * EB, bp.do_row_stride * EB, pr.len, p.dv, nt, lane);
}
template <typename T, int TPT = 2>
HSTU_DEV void solo_bwd_commit(const HstuAttnBwdParams& bp, const SoloProb& pr, const u32x4 (&rk)[2 * TPT], const u32x4 (&rv)[2 * TPT],
                              const u32x4 (&rq)[2 * TPT], const u32x4 (&rd)[2 * TPT], char* slice, int lane) {
  using S = SoloCfg<T>;
  const HstuAttnParams& p = bp.fwd;
  const int nt = (pr.len + 31) >> 5;
  solo_commit<T, TPT>(rk, slice, pr.len, p.dqk, nt, lane);
  solo_commit<T, TPT>(rv, slice + TPT * S::TILE, pr.len, p.dv, nt, lane);
  solo_commit<T, TPT>(rq, slice + 2 * TPT * S::TILE, pr.len, p.dqk, nt, lane);
  solo_commit<T, TPT>(rd, slice + 3 * TPT * S::TILE, pr.len, p.dv, nt, lane);
}
template <typename T, int TPT = 2>
HSTU_DEV void solo_bwd_stage(const HstuAttnBwdParams& bp, int b, int hd, char* slice, int lane) {
  using S = SoloCfg<T>;
  constexpr int EB = Elem<T>::kBytes;
  const HstuAttnParams& p = bp.fwd;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * TPT);
  if (len <= 0) return;
  const int nt = (len + 31) >> 5;
  char* Kt = slice, *Vt = slice + TPT * S::TILE, *Qt = slice + 2 * TPT * S::TILE, *dOt = slice + 3 * TPT * S::TILE;
  solo_stage<T, TPT>(Kt, (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * EB, p.k_row_stride * EB, len, p.dqk, nt, lane);
  solo_stage<T, TPT>(Vt, (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * EB, p.v_row_stride * EB, len, p.dv, nt, lane);
  solo_stage<T, TPT>(Qt, (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * EB, p.q_row_stride * EB, len, p.dqk, nt, lane);
  solo_stage<T, TPT>(dOt, (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * EB, bp.do_row_stride * EB, len, p.dv, nt, lane);
}

// `staged`: the caller has run solo_bwd_stage for this problem already
template <typename T, typename BX = FoldNoBias, int TPT = 2>
HSTU_DEV void solo_bwd_problem_x(const HstuAttnBwdParams& bp, int b, int hd, char* slice, int lane, BX& bx, bool staged = false) {
  using S = SoloCfg<T>;
  using C = BwdCfg<T, kSoloD, kSoloD>;
  constexpr int EB = Elem<T>::kBytes;
  const HstuAttnParams& p = bp.fwd;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * TPT);
  if (len <= 0) return;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  HSTU_TRACE_DECL(bp.workspace, false);
  const int nt = TPT == 1 ? 1 : (len + 31) >> 5;
  char* Kt = slice, *Vt = slice + TPT * S::TILE, *Qt = slice + 2 * TPT * S::TILE, *dOt = slice + 3 * TPT * S::TILE, *ds = slice + 4 * TPT * S::TILE;
  if (!staged) solo_bwd_stage<T, TPT>(bp, b, hd, slice, lane);
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
    if (TPT > 1 && i >= 1 && (mc.win == 0 || mc.pair_may_be_active(32 * i, 32, 32, 32)))
      fold_pair_x<T, kSoloD, kSoloD, BX>(p, mc, Kt + S::TILE, Vt + S::TILE, Qt + i * S::TILE, dOt + i * S::TILE, ds + 32 * 64, 32 * i, 32, dk1, dv1,
                                         lane, dmvm, bx HSTU_TRACE_PASS);
    if (mc.win == 0 || mc.pair_may_be_active(32 * i, 32, 0, 32))
      fold_pair_x<T, kSoloD, kSoloD, BX>(p, mc, Kt, Vt, Qt + i * S::TILE, dOt + i * S::TILE, ds, 32 * i, 0, dk0, dv0, lane, dmvm, bx HSTU_TRACE_PASS);
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
  if (TPT > 1 && nt > 1) {
    fold_park_tile<T, kSoloD>(dk1, ds_scale, Kt + S::TILE, lane);
    fold_park_tile<T, kSoloD>(dv1, scale_v, Vt + S::TILE, lane);
    solo_copy_out<T>(Kt + S::TILE, dk_head + 32 * bp.dk_row_stride * EB, bp.dk_row_stride * EB, len - 32, p.dqk, lane);
    solo_copy_out<T>(Vt + S::TILE, dv_head + 32 * bp.dv_row_stride * EB, bp.dv_row_stride * EB, len - 32, p.dv, lane);
  }
  (void)C::EB;
}
// Research-path backward at the short-sequence shapes: as the forward above -- the workgroup walks users, its waves the
// heads -- with the bias term and the two histograms of dS' of the folded research kernel (fold_pair_x<FoldBias>): ONE pair of
// LDS histograms per workgroup for everything it processes, flushed to its row of `bias_partial` at the end; the user's
// bucket bytes computed once by the four waves (fold_pair_x then always reads its byte cache).
// LDS: [4 slices][pos histogram 2N | time histogram (nb+1) x ts_copies][2 x (tables | bucket bytes)].
//
// Round 6: TWO launches by length class.  A slice sized for 64 rows (20 KiB per wave) left room for one workgroup per CU -- one wave
// per SIMD, every latency of a user's chain exposed, two workgroup barriers per user with nothing else on the CU -- while 95 % of an
// Amazon-Books batch is shorter than 33 rows.  The launch that takes the users of <= 32 rows lays its slices out with ONE tile per
// tensor (`tpt` = 1: 10 KiB per wave, 1 KiB of bucket bytes): two independent workgroups per CU (the lesson of every experiment of
// rounds 4-6: independent workgroups hide each other's latencies, a lone lock-stepped one hides nothing); the other launch takes the
// users of 33 .. 64 rows as before.  A launch walks the whole launch order and skips the users of the other class (offsets through
// scalar loads: no wait on the prefetched rows).  `row0`: first row of `bias_partial` of this launch.
struct SoloSlot { SoloProb pr; int slot; };
HSTU_DEV SoloSlot solo_next_user(const HstuAttnParams& p, int slot, int stride, int hd, int len_lo, int len_hi) {
  SoloSlot r;
  r.pr.b = 0; r.pr.hd = hd; r.pr.off0 = 0; r.pr.len = 0;
  for (; slot < p.batch; slot += stride) {
    r.pr.b = user_of_slot_s(p, slot);
    r.pr.off0 = sload_index(p.seq_offsets, r.pr.b, p.offsets_dtype);
    r.pr.len = min((int)(sload_index(p.seq_offsets, r.pr.b + 1, p.offsets_dtype) - r.pr.off0), kSoloMaxLen);
    if (r.pr.len > len_lo && r.pr.len <= len_hi) break;
  }
  if (slot >= p.batch) {      // none left: a harmless problem (every load of it is a valid one, nothing is computed for it)
    r.pr.b = user_of_slot_s(p, 0);
    r.pr.off0 = sload_index(p.seq_offsets, r.pr.b, p.offsets_dtype);
    r.pr.len = 0;
  }
  r.slot = slot;
  return r;
}

template <typename T, int TPT>
__global__ __launch_bounds__(kSoloThreads) __attribute__((amdgpu_waves_per_eu(TPT == 1 ? 2 : 1, TPT == 1 ? 2 : 1))) void hstu_attn_bwd_solo_bias_kernel(
    const HstuAttnBwdParams bp, float* bias_partial, int ts_copies, int hist_bytes, int table_bytes, int len_lo, int len_hi, int row0) {
  using S = SoloCfg<T>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const HstuAttnParams& p = bp.fwd;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int slice_bytes = S::bwd_slice(TPT);
  char* slice = smem + wave * slice_bytes;
  FoldBias bx;
  bx.hpos = (float*)(smem + kSoloWaves * slice_bytes);
  bx.hts = bx.hpos + 2 * p.max_seq_len;
  char* const tab0 = (char*)bx.hpos + hist_bytes;
  const int tab_stride = table_bytes + (TPT == 1 ? 1024 : kSoloBucketBytes);
  bx.ts_run.init(bx.hts, ts_copies);
  bx.cached = true;
  const int hist_floats = 2 * p.max_seq_len + (p.num_buckets + 1) * ts_copies;
  for (int i = tid; i < hist_floats; i += kSoloThreads) bx.hpos[i] = 0.f;
  const int hd0 = min(wave, p.heads - 1);
  const int grid = gridDim.x;
  {
    const int b0 = user_of_slot(p, min((int)blockIdx.x, p.batch - 1));
    stage_bias_tables(p, b0, tab0, tid, kSoloThreads);
    stage_bias_tables(p, b0, tab0 + tab_stride, tid, kSoloThreads);
  }
  // the same software pipeline over the users as the forward kernel: rows / timestamps of the next user, offsets of the
  // one after, requested while this one is worked on; tables and bucket bytes double-buffered
  SoloSlot cur = solo_next_user(p, blockIdx.x, grid, hd0, len_lo, len_hi);
  SoloSlot nxt = solo_next_user(p, cur.slot + grid, grid, hd0, len_lo, len_hi);
  u32x4 rk[2 * TPT], rv[2 * TPT], rq[2 * TPT], rd[2 * TPT];
  SoloTs ts = {0, 0, 0, 0};
  solo_bwd_issue<T, TPT>(bp, cur.pr, rk, rv, rq, rd, lane);
  solo_ts_issue(p, cur.pr.b, tid, ts);
  __syncthreads();                         // histograms zeroed, tables in place
  int buf = 0;
  while (cur.slot < p.batch) {
    char* const tables = tab0 + buf * tab_stride;
    bx.bcache = tables + table_bytes;
    solo_bwd_commit<T, TPT>(bp, cur.pr, rk, rv, rq, rd, slice, lane);
    bx.bc = solo_ts_commit(p, tables, tid, ts);
    const SoloProb pf = nxt.slot < p.batch ? nxt.pr : cur.pr;
    solo_bwd_issue<T, TPT>(bp, pf, rk, rv, rq, rd, lane);
    solo_ts_issue(p, pf.b, tid, ts);
    const SoloSlot nn = solo_next_user(p, nxt.slot + grid, grid, hd0, len_lo, len_hi);
    __syncthreads();
    bx.bc.finish(kSoloWaves);
    solo_bucket_bytes<true>(bx.bc, bx.bcache, cur.pr.len, wave, lane);
    __syncthreads();
    for (int hd = wave; hd < p.heads; hd += kSoloWaves) solo_bwd_problem_x<T, FoldBias, TPT>(bp, cur.pr.b, hd, slice, lane, bx, hd == wave);
    cur = nxt;
    nxt = nn;
    buf ^= 1;
  }
  bx.ts_run.flush();
  __syncthreads();
  const float scale_v = attn_scale_of(p);
  float* row = bias_partial + (int64_t)(row0 + (int)blockIdx.x) * (2 * p.max_seq_len + p.num_buckets);
  const int npos = 2 * p.max_seq_len - 1;
  for (int i = tid; i < 2 * p.max_seq_len + p.num_buckets; i += kSoloThreads) {
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

// TPT = 1: the launch that takes the (user, head) problems of <= 32 rows only -- slices of one tile per tensor (10 KiB per wave instead of
// 20), half the accumulators and staging registers (128 instead of 213): four workgroups per CU instead of two (the other launch takes
// the longer users)
#ifndef SOLO_BWD_SHORT_WAVES
#define SOLO_BWD_SHORT_WAVES 4
#endif
template <typename T, int TPT>
__global__ __launch_bounds__(kSoloThreads) __attribute__((amdgpu_waves_per_eu(TPT == 1 ? SOLO_BWD_SHORT_WAVES : 2, TPT == 1 ? SOLO_BWD_SHORT_WAVES : 2)))
void hstu_attn_bwd_solo_kernel(const HstuAttnBwdParams bp, int len_lo, int len_hi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* slice = smem + wave * SoloCfg<T>::bwd_slice(TPT);
  const HstuAttnParams& p = bp.fwd;
  const int total = p.batch * p.heads;
  for (int uh = blockIdx.x * kSoloWaves + wave; uh < total; uh += gridDim.x * kSoloWaves) {
    int uh_l = uh;
    asm volatile("" : "+s"(uh_l));
    const int b = user_of_slot_s(p, uh_l / p.heads);
    const int len = min((int)(sload_index(p.seq_offsets, b + 1, p.offsets_dtype) - sload_index(p.seq_offsets, b, p.offsets_dtype)), kSoloMaxLen);
    if (len <= len_lo || len > len_hi) continue;       // (the other launch's)
    FoldNoBias nb;
    solo_bwd_problem_x<T, FoldNoBias, TPT>(bp, b, uh_l % p.heads, slice, lane, nb);
  }
}

}  // namespace hstu
