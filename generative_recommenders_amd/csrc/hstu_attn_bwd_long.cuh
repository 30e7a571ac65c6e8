// HSTU attention backward for LONG sequences (more than one key block: max_seq_len > 224), 16-bit I/O, no bias, no contextual rows:
// TWO kernels, no atomics, no fp32 workspace, bit-deterministic.
//
// Why.  The general kernel (hstu_attn_bwd.cuh) keeps a block of <= 5 key tiles resident, runs one pair per owner wave and step, reduces
// dQ over the block's key tiles with a second GEMM on the idle waves and ADDS the block's fp32 dQ partial into a workspace with atomics
// that a third kernel converts: two barriers, a dS' round trip through LDS and 16 atomic instructions per pair.  At the reference's own
// benchmark lengths (ops/benchmarks/hstu_attention_bench.py:139, seq_len 2^8 .. 2^12) it runs at 0.10 of the MFMA peak and LOSES to
// the reference's Triton kernel on the same MI355X (N = 2048, 512 users x 4 heads of 128: 24.7 ms against 16.2;
// profiles/r06_reference_triton_long.txt).  Long sequences are compute bound (the causal triangle of N = 2048 is 1891 tile pairs
// of 1.3 MFLOP per user and head), so the split that FlashAttention-2 style backward passes use pays here: recompute S and dP once more
// (7 GEMMs per pair instead of 5) and get two kernels with the forward's structure -- operands of one side in registers for the
// whole workgroup's life, tiles of the other side streaming through an LDS ring, ONE barrier per tile, nothing published, nothing added
// in memory:
//
//   hstu_attn_bwd_dkv_kernel   workgroup = (user, head, block of 7 key tiles), 8 waves; wave w < 7 owns key tile w: dK^T / dV^T
//                              accumulators in registers (128 VGPRs), its K / V tile resident in LDS; the Q / dO tiles of the rows
//                              on or below the block stream through a 3-deep ring (LDS-DMA, counted vmcnt); per tile the pair code
//                              of the folded kernel (fold_pair_x) without the dS' hand-over.  7 x 16 + 3 x 16 KiB = 160 KiB.
//   hstu_attn_bwd_dq_kernel    the forward kernel's mapping: workgroup = (user, head, 128 query rows), wave = 32 rows, Q and dO
//                              fragments in registers, K / V tiles through the 3-deep ring; everything transposed (query on the lane
//                              axis):  S^T = K Q^T,  dP^T = V dO^T,  dS'^T element-wise (the C layout is the next MFMA's B layout),
//                              dQ^T += K^T dS'^T with K through the LDS transpose read.  Two workgroups per CU.
//
// Math (ops/pytorch/pt_hstu_attention.py:87-168 differentiated):  x = alpha S,  sg = sigmoid(x),  P' = x sg,
// dS' = dP (sg + P' (1 - sg));  dV = scale P'^T dO,  dK = scale alpha dS'^T Q,  dQ = scale alpha dS' K  (masked elements: zero).
// Replaces ops/triton/triton_hstu_attention.py:899-1764 (`_hstu_attn_bwd*`) at these lengths.
#pragma once
#include "hstu_attn_bwd_fold.cuh"

namespace hstu {

#ifndef LONG_NW
#define LONG_NW 7
#endif
#ifndef LONG_STAGES
#define LONG_STAGES 3
#endif
#ifndef LONG_LAUNDER
#define LONG_LAUNDER 0   // 1: lane id laundered in front of the pair (LDS offsets recomputed per pair: 214 instead of 233 registers, 3-5 % slower)
#endif
constexpr int kLongNW = LONG_NW;            // key tiles (owner waves) per block of the dK / dV kernel
constexpr int kLongStages = LONG_STAGES;    // its Q / dO ring
static_assert(kLongNW <= kBwdWaves && kLongStages >= 2 && (kLongNW + kLongStages) * 16 <= 160, "LDS");

template <typename T, int D>
struct LongCfg {
  using B = BwdCfg<T, D, D>;
  static constexpr int smem_dkv() { return (kLongNW + kLongStages) * B::PAIR; }
};

// -------------------------------------------------------------------------------------------------------------------------------
// dK / dV
// -------------------------------------------------------------------------------------------------------------------------------
// CTX: the instantiation for contextual_seq_len > 0 (query tiles with contextual rows in front of the block's own; general predicate)
template <typename T, int D, bool CTX = false>
__global__ __launch_bounds__(kBwdThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void hstu_attn_bwd_dkv_kernel(const HstuAttnBwdParams bp, int nkb) {
  using C = BwdCfg<T, D, D>;
  static_assert(C::EB == 2, "16-bit I/O");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const HstuAttnParams& p = bp.fwd;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // work decode (as the other attention kernels: the blocks of 8 consecutive (user, head) pairs form a dispatch group, so that the key
  // blocks of one (user, head) land on one XCD -- block id mod 8 -- and re-read its Q / dO rows from that XCD's L2); the block next to
  // the sequence start sees every query tile and goes first
  const int bid = blockIdx.x;
  const int grp = bid / (8 * nkb), rem = bid % (8 * nkb);
  const int kb = rem / 8;
  const int uh = grp * 8 + (rem & 7);
  if (uh >= p.batch * p.heads) return;
  const int b = user_of_slot(p, uh / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  // (a user longer than max_seq_len is a caller error; as the folded kernel: rows past the last tile of max_seq_len are ignored)
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * ((p.max_seq_len + 31) >> 5));
  const int kt0 = kb * kLongNW;
  if (32 * kt0 >= len) return;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  const int nt = (len + 31) >> 5;
  const int nw = min(kLongNW, nt - kt0);      // key tiles of this block
  // query tiles that can see a key of the block: from the block's first tile on (causal), up to the reach of the attention window
  // when there is one and nothing lifts it (min_full_attn_seq_len, target rows: their ids are clamped, contextual rows: ids shifted);
  // in front of them the tiles that hold contextual rows (id 0: they see every key).  Step j of the loop is query tile tile_of(j).
  int it_hi = nt;
  if (mc.win > 0 && mc.full == 0 && !mc.has_targets && mc.ctx == 0) it_hi = min(nt, ((32 * (kt0 + nw) - 1 + mc.win) >> 5) + 1);
  const int n_pre = (CTX && mc.ctx > 0) ? min((mc.ctx + 31) >> 5, kt0) : 0;
  const int n_steps = n_pre + it_hi - kt0;
  auto tile_of = [&](int j) { return (CTX && j < n_pre) ? j : kt0 + (j - n_pre); };

  const char* qbase = (const char*)p.q + (off0 * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const char* dobase = (const char*)bp.dout + (off0 * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
  const int64_t q_rs = p.q_row_stride * C::EB, k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB, do_rs = bp.do_row_stride * C::EB;
  // LDS-DMA source addresses as scalar base + 32-bit lane offset (fold_tile_dma): strides < 16 MiB, the user's rows within 4 GiB
  const bool dma_fast = q_rs < (1 << 24) && k_rs < (1 << 24) && v_rs < (1 << 24) && do_rs < (1 << 24) && (int64_t)len * q_rs < (1LL << 32) &&
                        (int64_t)len * k_rs < (1LL << 32) && (int64_t)len * v_rs < (1LL << 32) && (int64_t)len * do_rs < (1LL << 32);

  char* const ring = smem + kLongNW * C::PAIR;
  auto stage_dma = [&](int it, int slot) {
    char* st = ring + slot * C::PAIR;
    fold_tile_dma<T, D>(st, qbase, q_rs, 32 * it, len, wave, lane, dma_fast);
    fold_tile_dma<T, D>(st + C::KT, dobase, do_rs, 32 * it, len, wave, lane, dma_fast);
  };
  // DMA instructions of one wave per ring stage (Q tile + dO tile, chunks dealt round-robin to the 8 waves; a wave without a chunk
  // -- head dim 64: waves 4..7 -- has nothing pending and its counted wait is a no-op)
  constexpr int PER = 2 * ((32 * C::UPR_K / 64 + kBwdWaves - 1) / kBwdWaves);

  // ---- prologue: the block's K / V tiles and the first two query tiles, all by LDS-DMA
  for (int t = 0; t < nw; ++t) {
    char* dst = smem + t * C::PAIR;
    fold_tile_dma<T, D>(dst, kbase, k_rs, 32 * (kt0 + t), len, wave, lane, dma_fast);
    fold_tile_dma<T, D>(dst + C::KT, vbase, v_rs, 32 * (kt0 + t), len, wave, lane, dma_fast);
  }
  for (int j = 0; j < kLongStages - 1 && j < n_steps; ++j) stage_dma(tile_of(j), j);

  f32x16 dk_acc[C::DBQ], dv_acc[C::DBV];
#pragma unroll
  for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk_acc[d][r] = 0.f; dv_acc[d][r] = 0.f; }
  const float scale_v = attn_scale_of(p);
  const float ds_scale = scale_v * p.alpha;
  // lane-constant mask patterns of the plain-causal case (fold_pair_x): low half = diagonal tile, bit r set iff key n32 <= query row
  // (r&3) + 8 (r>>2) + 4 hf; high half = the sequence's last query tile, bit r set iff that row is < len
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
  const int kt = kt0 + wave;
  const bool owner = wave < nw;
  FoldNoBias nb;
  HSTU_TRACE_DECL(nullptr, false);

  for (int j = 0; j < n_steps; ++j) {
    const int it = tile_of(j);
    const int slot = j % kLongStages;
    // the tile of step j has landed once only the tiles requested after it are pending
    const int newer = min(kLongStages - 2, n_steps - 1 - j);      // tiles requested after this one so far
    static_assert(kLongStages <= 4, "counted waits below");
    if (newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave's chunks of this tile (first time: of the K / V block) landed; the stage of the step before is free
    asm volatile("" ::: "memory");
    if (j + kLongStages - 1 < n_steps) stage_dma(tile_of(j + kLongStages - 1), (slot + kLongStages - 1) % kLongStages);
    if (owner && (mc.simple ? it >= kt : mc.pair_may_be_active(32 * it, 32, 32 * kt, 32))) {
      const char* Kw = smem + wave * C::PAIR;
      const char* st = ring + slot * C::PAIR;
      // (LONG_LAUNDER: the lane id laundered, LDS offsets derived from it recomputed per pair instead of living across the loop --
      // what the folded kernel needs next to its dQ phase; here they fit: 233 registers, 3-5 % faster)
      int lane1 = lane;
      if (LONG_LAUNDER) asm volatile("" : "+v"(lane1));
      fold_pair_x<T, D, D, FoldNoBias, false, CTX>(p, mc, Kw, Kw + C::KT, st, st + C::KT, nullptr, 32 * it, 32 * kt, dk_acc, dv_acc, lane1, dmvm, nb HSTU_TRACE_PASS);
    }
  }
  // ---- epilogue: every owner parks its two tiles over its own K / V tile (nobody else reads them in this kernel) and copies the
  // rows out itself: no barrier
  if (owner) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    char* const mine = smem + wave * C::PAIR;
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    fold_park_tile<T, D>(dk_acc, ds_scale, mine, lane2);
    fold_park_tile<T, D>(dv_acc, scale_v, mine + C::KT, lane2);
    char* const dk_head = (char*)bp.dk + (off0 * bp.dk_row_stride + (int64_t)hd * bp.dk_head_stride) * C::EB;
    char* const dv_head = (char*)bp.dv + (off0 * bp.dv_row_stride + (int64_t)hd * bp.dv_head_stride) * C::EB;
    const int64_t dk_rs = bp.dk_row_stride * C::EB, dv_rs = bp.dv_row_stride * C::EB;
    fold_copy_out<T, D, 64>(mine, dk_head + (int64_t)(32 * kt) * dk_rs, dk_rs, len - 32 * kt, lane2);
    fold_copy_out<T, D, 64>(mine + C::KT, dv_head + (int64_t)(32 * kt) * dv_rs, dv_rs, len - 32 * kt, lane2);
  }
}

// -------------------------------------------------------------------------------------------------------------------------------
// dQ
// -------------------------------------------------------------------------------------------------------------------------------
template <typename T, int D, bool CTX = false>
__global__ __launch_bounds__(kFwdThreads, 2) void hstu_attn_bwd_dq_kernel(const HstuAttnBwdParams bp, int nqb) {
  using C = FwdCfg<T, D, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  static_assert(C::EB == 2 && C::NS == 3 && C::COUNTED, "16-bit I/O, three-deep ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const HstuAttnParams& p = bp.fwd;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31, hf = lane >> 5;

  // work decode: hstu_attn_fwd_kernel's (heavier -- later -- query blocks first)
  const int bid = blockIdx.x;
  const int grp = bid / (8 * nqb), rem = bid % (8 * nqb);
  const int qb = nqb - 1 - rem / 8;
  const int uh = grp * 8 + (rem & 7);
  if (uh >= p.batch * p.heads) return;
  const int b = user_of_slot(p, uh / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  // (a user longer than max_seq_len is a caller error; as the folded kernel: rows past the last tile of max_seq_len are ignored)
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * ((p.max_seq_len + 31) >> 5));
  const int q0 = qb * kFwdRowsPerBlock;
  if (q0 >= len) return;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  const float ds_scale = attn_scale_of(p) * p.alpha;

  const int na_blk = (min(kFwdRowsPerBlock, len - q0) + 31) >> 5;
  const int vw = ((qb & 1) && wave < na_blk) ? na_blk - 1 - wave : wave;      // (odd blocks hand their row tiles out in reverse: SIMD balance)
  const int r0 = q0 + 32 * vw;
  const bool wave_active = r0 < len;
  const int qi = r0 + n32;                     // this lane's query row
  const bool row_ok = qi < len;

  // key range of the workgroup (conservative; the element mask is exact); contextual rows (id 0) see every key
  const int i_last = min(q0 + kFwdRowsPerBlock, len) - 1;
  const bool ctx_rows = CTX && mc.ctx > 0 && q0 < mc.ctx;
  const int kv_hi = ctx_rows ? len : min(len, i_last + 1);
  int kv_lo = 0;
  if (mc.win > 0 && mc.full == 0 && !ctx_rows) {
    const int x = mc.id_of(q0) - mc.win;
    const int pos = x <= 0 ? 0 : ((CTX && mc.ctx > 0) ? x + mc.ctx - 1 : x);
    kv_lo = (pos >> 5) << 5;
  }
  const int ntiles = (kv_hi - kv_lo + 31) >> 5;

  // Q and dO fragments of the wave's rows (B operands): lane (q = n32, hf) holds elements hf D/2 + 8 kg .. + 8 of its row; rows past
  // the sequence end are zero: S = 0, dP = 0, dS' = 0 without a mask
  Frag qf[C::KG], dof[C::KG];
  {
    const int ld_row = min(qi, len - 1);
    const char* qrow = (const char*)p.q + ((off0 + ld_row) * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;
    const char* dorow = (const char*)bp.dout + ((off0 + ld_row) * bp.do_row_stride + (int64_t)hd * bp.do_head_stride) * C::EB;
    RawFrag<T> rq[C::KG], rd[C::KG];
#pragma unroll
    for (int kg = 0; kg < C::KG; ++kg) {
      rq[kg] = global_row_frag_issue<T>(qrow, hf * (D / 2) + kg * 8, true);
      rd[kg] = global_row_frag_issue<T>(dorow, hf * (D / 2) + kg * 8, true);
    }
#pragma unroll
    for (int kg = 0; kg < C::KG; ++kg) {
      qf[kg] = finish_row_frag<T>(rq[kg], row_ok, true);
      dof[kg] = finish_row_frag<T>(rd[kg], row_ok, true);
    }
  }
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const int64_t k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB;

  f32x16 acc[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // K / V ring: hstu_attn_fwd_kernel's (two tiles in flight, one raw barrier per tile, counted vmcnt, planned DMA addresses)
  constexpr int NIK = C::NCH_K / 4;
  const bool dma_fast = k_rs < (1 << 24) && v_rs < (1 << 24) && (int64_t)len * k_rs < (1LL << 32) && (int64_t)len * v_rs < (1LL << 32);
  DmaPlan<NIK> pl;
  if (dma_fast) dma_plan<T, D, NIK>(pl, D, wave, 4, lane);
  auto issue_tile = [&](int t, int slot) {
    char* st = smem + slot * C::STAGE;
    if (dma_fast) {
      tile_dma_fast<NIK>(st, kbase, (uint32_t)k_rs, kv_lo + 32 * t, len, pl, wave, 4);
      tile_dma_fast<NIK>(st + C::KT, vbase, (uint32_t)v_rs, kv_lo + 32 * t, len, pl, wave, 4);
      return;
    }
    tile_dma<T, D>(st, kbase, k_rs, kv_lo + 32 * t, len, D, wave, 4, lane);
    tile_dma<T, D>(st + C::KT, vbase, v_rs, kv_lo + 32 * t, len, D, wave, 4, lane);
  };
  for (int t = 0; t < C::NS - 1 && t < ntiles; ++t) issue_tile(t, t);
  // (target rows only and every row of this wave in front of the first target: the plain causal path, as in hstu_attn_fwd_kernel)
  const bool wave_plain = mc.simple || (HSTU_TARGETS_PLAIN && mc.has_targets && mc.win == 0 && mc.ctx == 0 && r0 + 32 <= min(len, mc.max_id));

  for (int t = 0; t < ntiles; ++t) {
    const int slot = t % C::NS;
    const int j0 = kv_lo + (t << 5);
    if (min(C::NS - 2, ntiles - 1 - t) >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + C::NS - 1 < ntiles) issue_tile(t + C::NS - 1, (slot + C::NS - 1) % C::NS);
    bool tile_act, tile_full;
    if (wave_plain) {
      tile_act = r0 < len && j0 <= min(r0 + 31, len - 1);
      tile_full = j0 + 32 <= r0;        // strictly below this wave's first row
    } else {
      tile_act = mc.pair_may_be_active(r0, 32, j0, 32);
      tile_full = tile_act && mc.pair_fully_valid(r0, 32, j0, 32);
    }
    if (!(wave_active && tile_act)) continue;
    const char* Kt = smem + slot * C::STAGE;
    const char* Vt = Kt + C::KT;
    // mode (wave-uniform): 0 no mask, 4 plain causal (the only partly masked tile is the aligned diagonal one: a lane-constant
    // pattern, put into S itself -- a masked element is -1e30, alpha S hugely negative, exp2 gives +inf, the sigmoid exactly 0 and
    // dS' = dP * 0), 3 targets / window by integer arithmetic, 2 contextual rows: the general predicate by compares
    const int mode = tile_full ? 0 : (wave_plain ? 4 : ((!CTX || mc.ctx == 0) ? 3 : 2));
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    // S^T and dP^T as one stream alternating between the two accumulators (no back-to-back dependency)
#pragma unroll
    for (int kg = 0; kg < C::KG; ++kg) {
      const Frag a = lds_row_frag<T, C::UPR_K>(Kt, n32, hf * (D / 2) + kg * 8);
      const Frag c = lds_row_frag<T, C::UPR_V>(Vt, n32, hf * (D / 2) + kg * 8);
      s = E::mma(a, qf[kg], s);
      dp = E::mma(c, dof[kg], dp);
    }
    if (mode == 4) {
      const float neg = p.alpha < 0.f ? 1e30f : -1e30f;
      const int x = n32 - 4 * hf;           // key (r&3) + 8 (r>>2) + 4 hf > query n32  <=>  (r&3) + 8 (r>>2) > x
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = ((r & 3) + 8 * (r >> 2) > x) ? neg : s[r];
    }
    Frag dsb[2];
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
      float dsv[8];
      const f32x2 a2 = {p.alpha, p.alpha};
      const f32x2 c2 = {-1.44269504088896340736f * p.alpha, -1.44269504088896340736f * p.alpha};
      const f32x2 one2 = {1.f, 1.f};
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const int r = 8 * h8 + j;
        const f32x2 sv = {s[r], s[r + 1]}, dpv = {dp[r], dp[r + 1]};
        const f32x2 x = sv * a2, tt = sv * c2;
        const f32x2 e = {__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])};
        const f32x2 dn = e + one2;
        const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
        const f32x2 pr = x * sg;
        const f32x2 dsr = dpv * (pr * (one2 - sg) + sg);     // sg (1 + x (1 - sg)) = sg + P' (1 - sg)
        dsv[j] = dsr[0];
        dsv[j + 1] = dsr[1];
      }
      if (mode == 3) {
        const int i_eff = row_ok ? qi : -1;
        const int idi = mc.has_targets ? min(i_eff, mc.max_id) : i_eff;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * h8 + j;
          const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          const int idj = mc.has_targets ? min(key, mc.max_id) : key;
          const int keep = mc.keep_bits_row(i_eff, idi, key, idj) & ((key - len) >> 31);
          dsv[j] = __builtin_bit_cast(float, __builtin_bit_cast(int, dsv[j]) & keep);
        }
      }
      if (CTX && mode == 2) {
        const int qi_id = mc.id_of(qi);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * h8 + j;
          const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
          const bool ok = row_ok & (key < len) & mc.valid_ids(qi, key, qi_id, mc.id_of(key));
          dsv[j] = ok ? dsv[j] : 0.f;
        }
      }
      dsb[h8] = E::pack8(dsv);
    }
    // dQ^T[d][q] += K^T[d][key] dS'^T[key][q]
#pragma unroll
    for (int d = 0; d < C::DB; ++d) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const Frag a = lds_col_frag<T, C::UPR_K>(Kt, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, lane);
        acc[d] = E::mma(a, dsb[ks], acc[d]);
      }
    }
  }

  // ---- epilogue (hstu_attn_fwd_kernel's barrier-free one): the dQ^T accumulators (column n32 = query row, registers = features) are
  // parked as a row-major tile in a ring slot that is dead after the last step's barrier and leave as whole rows
  {
    const int dead = ((ntiles > 0 ? ntiles - 1 : 0) + 1 + (wave >> 1)) % C::NS;
    char* tile = smem + dead * C::STAGE + (wave & 1) * C::VT;
    if (wave_active) {
#pragma unroll
      for (int d = 0; d < C::DB; ++d)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          u32x2 v = {E::pk2(acc[d][4 * rq] * ds_scale, acc[d][4 * rq + 1] * ds_scale),
                     E::pk2(acc[d][4 * rq + 2] * ds_scale, acc[d][4 * rq + 3] * ds_scale)};
          *LDS_PTR(u32x2, tile + tile_off<C::UPR_V>(n32, 4 * d + rq) + 8 * hf) = v;
        }
      char* obase = (char*)bp.dq + ((off0 + r0) * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB;
      const int rows_valid = len - r0;
#pragma unroll
      for (int i = 0; i < 32 * C::UPR_V / 64; ++i) {
        const int idx = i * 64 + lane;
        const int row = idx / C::UPR_V, unit = idx % C::UPR_V;
        const u32x4 v = *LDS_PTR(const u32x4, tile + tile_off<C::UPR_V>(row, unit));
        if (row < rows_valid) gstore16(obase + (int64_t)row * bp.dq_row_stride * C::EB + unit * 16, v);
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------------------------------------
template <typename T, int D>
static int launch_bwd_long_inst(const HstuAttnBwdParams& bp, hipStream_t st) {
  const HstuAttnParams& p = bp.fwd;
  const int tmax = (p.max_seq_len + 31) / 32;
  const int groups = (p.batch * p.heads + 7) / 8;
  {
    const int nkb = (tmax + kLongNW - 1) / kLongNW;
    const int smem = LongCfg<T, D>::smem_dkv();
    auto kern = p.contextual_seq_len > 0 ? hstu_attn_bwd_dkv_kernel<T, D, true> : hstu_attn_bwd_dkv_kernel<T, D, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_attn_bwd(long): cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(groups * 8 * nkb), dim3(kBwdThreads), smem, st, bp, nkb);
    if (int rc = check_launch("hstu_attn_bwd(long, dk/dv)")) return rc;
  }
  {
    using C = FwdCfg<T, D, D>;
    const int nqb = (p.max_seq_len + kFwdRowsPerBlock - 1) / kFwdRowsPerBlock;
    auto kern = p.contextual_seq_len > 0 ? hstu_attn_bwd_dq_kernel<T, D, true> : hstu_attn_bwd_dq_kernel<T, D, false>;
    hipLaunchKernelGGL(kern, dim3(groups * 8 * nqb), dim3(kFwdThreads), C::SMEM, st, bp, nqb);
    return check_launch("hstu_attn_bwd(long, dq)");
  }
}

template <typename T>
static int launch_bwd_long_dtype(const HstuAttnBwdParams& bp, hipStream_t st) {
  if (bp.fwd.dqk == 128) return launch_bwd_long_inst<T, 128>(bp, st);
  if (bp.fwd.dqk == 64) return launch_bwd_long_inst<T, 64>(bp, st);
  return set_error(HSTU_EUNSUPPORTED, "hstu_attn_bwd(long): head dim %d not instantiated", bp.fwd.dqk);
}

}  // namespace hstu
