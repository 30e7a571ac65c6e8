// HSTU attention backward for short sequences at head dim 64: workgroups of FOUR waves, TWO of them per CU.
//
// Why a second short-sequence kernel.  The folded kernel (hstu_attn_bwd_fold.cuh) is one 8-wave workgroup per CU: its
// phases -- K/V block streaming in, pairs (MFMA / VALU), dQ GEMM (LDS), parked tiles streaming out -- run one after
// the other and nothing else is on the CU to fill them (removing any one phase shortens the launch by almost its
// whole duration: profiles/r02_fold_ablation.txt).  At head dim 128 that is forced: the dK/dV accumulators of one
// (user, head) are 224 KiB of the 512 KiB register file and its K/V block 112 KiB of the 160 KiB LDS.  At head dim 64
// both halve: a workgroup of 4 waves (256 VGPRs each: two key tiles' accumulators = 128 registers) with 78 KiB of
// LDS holds a whole problem, and two such workgroups -- two independent problems, each in whatever phase it happens to
// be -- share a CU.
//
// Schedule (per workgroup; the plain descending one, no fold, no hand-over):
//   wave w owns key tiles w and 6 - w (wave 3: tile 3 only): dK/dV of both live in registers;
//   step i = nt-1 .. 0 takes query tile i:  every wave runs the pairs (i, t) of its tiles t <= i (S, dP, P', dS',
//   dV +=, dK +=; dS' published to LDS), barrier, then the dQ GEMM of tile i over all key tiles with the waves
//   splitting the 64 features x 32 rows four ways, while the next query tile streams in;
//   key tile i is final after step i: its owner parks dV (then dK) over the dead V (K) tile and all waves copy the
//   rows out -- every tile leaves during the loop, tile 0 right after it.
// Same fragments, LDS tile layout, mask handling (S accumulator start value) and byte counts as the folded kernel,
// whose building blocks it uses.
#pragma once
#include "hstu_attn_bwd_fold.cuh"

#ifndef QUAD_ABLATE
#define QUAD_ABLATE 0      // timing experiments only (WRONG results): 1 no dQ stores, 2 no dk/dv stores, 16 no K/V DMA, 32 no dQ
#endif                     // GEMM, 64 no pairs

namespace hstu {

constexpr int kQuadWaves = 4;
constexpr int kQuadThreads = 256;

template <typename T, int D>
struct QuadCfg {
  using B = BwdCfg<T, D, D>;
  static constexpr int DSB = 32 * 64;                 // [32 keys][32 q] 16-bit dS' tile
  static constexpr int kMaxTiles = 7;
  // 7 K/V pairs + one Q/dO stage + 7 dS' tiles
  static constexpr int smem_bytes() { return kMaxTiles * B::PAIR + B::PAIR + kMaxTiles * DSB; }
};

template <typename T, int D>
HSTU_DEV void quad_tile_dma(char* tile, const char* base, int64_t row_stride_bytes, int row0, int len, int wave, int lane, bool fast = false) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int NCH = 32 * UPR / 64;   // 1 KiB chunks per tile
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tile);
  for (int c = wave; c < NCH; c += kQuadWaves) {
    const int pidx = c * 64 + lane;
    const int row = pidx / UPR, slot = pidx % UPR;
    const int unit = slot ^ swz<UPR>(row);
    const int grow = min(row0 + row, len - 1);
    if (fast) dma16_saddr(__umul24((uint32_t)grow, (uint32_t)row_stride_bytes) + unit * 16, base, lds0 + c * 1024);   // (fold_tile_dma)
    else dma16_asm(base + (int64_t)grow * row_stride_bytes + unit * 16, lds0 + c * 1024);
  }
}

template <typename T, int D>
HSTU_DEV void quad_copy_out(const char* __restrict__ tile, char* gtile, int64_t row_stride_bytes, int rows_valid, int tid) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  for (int u = tid; u < 32 * UPR; u += kQuadThreads) {
    const int row = u / UPR, unit = u % UPR;
    const u32x4 v = *LDS_PTR(const u32x4, tile + tile_off<UPR>(row, unit));
    if (row < rows_valid && (!(QUAD_ABLATE & 2) || row_stride_bytes == -12345)) gstore16(gtile + row * row_stride_bytes + unit * 16, v);
  }
}

// dQ of query tile qt: dQ^T[d][q] = sum over key tiles of K_t^T[d][key] dS'_t^T[key][q] (16x16x32 MFMA, one key tile =
// one contraction; NSLOT = qt + 1 of them, a compile-time count: the causal triangle needs 28 contractions per problem, a
// static 7-slot loop with zeroed idle slots runs 49).  Wave w owns the 32 feature columns [32 (w & 1), +32) of query rows [16 (w >> 1), +16): two
// MFMAs whose A rows interleave the features in groups of 4, so that a lane ends up with 8 consecutive features of
// one query row (one 16-byte store; see fold_dq_phase).  Static 7-slot loop, idle slots get a zeroed dS' fragment.
template <typename T, int D, int NSLOT>
HSTU_DEV void quad_dq_slots(const HstuAttnBwdParams& bp, const MaskCtx& mc, const char* __restrict__ kv,
                            const char* __restrict__ dsbuf, int qt, int wave, int64_t off0, int hd, float ds_scale, int lane) {
  using C = BwdCfg<T, D, D>;
  using Q = QuadCfg<T, D>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  static_assert(D == 64, "feature split: 2 x 32 columns");
  const int db = wave & 1, qb = wave >> 1;
  const int i16 = lane & 15, g = lane >> 4;
  const int row_lo = 8 * g + (i16 >> 2), row_hi = row_lo + 4;
  const int colK0 = 32 * db + 8 * (i16 & 3), colK1 = colK0 + 4;
  const int k0_lo = tile_off<C::UPR_K>(row_lo, colK0 >> 3) + ((colK0 & 7) << 1);
  const int k0_hi = tile_off<C::UPR_K>(row_hi, colK0 >> 3) + ((colK0 & 7) << 1);
  const int k1_lo = tile_off<C::UPR_K>(row_lo, colK1 >> 3) + ((colK1 & 7) << 1);
  const int k1_hi = tile_off<C::UPR_K>(row_hi, colK1 >> 3) + ((colK1 & 7) << 1);
  const int d_lo = fold_ds_off(row_lo, 4 * qb + (i16 & 3)), d_hi = fold_ds_off(row_hi, 4 * qb + (i16 & 3));
  unsigned on = (1u << (qt + 1)) - 1u;
  if (mc.win != 0)
    for (int t = 0; t <= qt; ++t)
      if (!mc.pair_may_be_active(32 * qt, 32, 32 * t, 32)) on &= ~(1u << t);
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int t = 0; t < NSLOT; ++t) {
    const char* Kt = kv + t * C::PAIR;
    const Frag fk0 = tr_frag16<T>(Kt, k0_lo, k0_hi), fk1 = tr_frag16<T>(Kt, k1_lo, k1_hi);
    Frag fd = tr_frag16<T>(dsbuf + t * Q::DSB, d_lo, d_hi);
    fd.v = __builtin_bit_cast(typename E::vec8, ((on >> t) & 1u) ? __builtin_bit_cast(u32x4, fd.v) : zero4);
    acc[0] = E::mma16(fk0, fd, acc[0]);
    acc[1] = E::mma16(fk1, fd, acc[1]);
  }
  // requested order: the 6 transposed reads of slot s+1 ahead of the MFMA pair of slot s
  __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl) {
    if (sl + 1 < NSLOT) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
  }
  // C layout of MFMA h: column i16 = query row, register r = feature 32 db + 8 g + 4 h + r
  const int qrow = 32 * qt + 16 * qb + i16;
  if (qrow < mc.len && (!(QUAD_ABLATE & 1) || bp.total_rows == -12345)) {
    char* dqrow = (char*)bp.dq + ((off0 + qrow) * bp.dq_row_stride + (int64_t)hd * bp.dq_head_stride) * C::EB;
    u32x4 v = {E::pk2(acc[0][0] * ds_scale, acc[0][1] * ds_scale), E::pk2(acc[0][2] * ds_scale, acc[0][3] * ds_scale),
               E::pk2(acc[1][0] * ds_scale, acc[1][1] * ds_scale), E::pk2(acc[1][2] * ds_scale, acc[1][3] * ds_scale)};
    gstore16(dqrow + (32 * db + 8 * g) * C::EB, v);
  }
}

template <typename T, int D>
HSTU_DEV void quad_dq_phase(const HstuAttnBwdParams& bp, const MaskCtx& mc, const char* __restrict__ kv,
                            const char* __restrict__ dsbuf, int qt, int wave, int64_t off0, int hd, float ds_scale, int lane) {
  switch (qt) {   // (wave-uniform)
    case 0: return quad_dq_slots<T, D, 1>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
    case 1: return quad_dq_slots<T, D, 2>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
    case 2: return quad_dq_slots<T, D, 3>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
    case 3: return quad_dq_slots<T, D, 4>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
    case 4: return quad_dq_slots<T, D, 5>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
    case 5: return quad_dq_slots<T, D, 6>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
    default: return quad_dq_slots<T, D, 7>(bp, mc, kv, dsbuf, qt, wave, off0, hd, ds_scale, lane);
  }
}

#ifndef QUAD_PERSIST
#define QUAD_PERSIST 0     // 1: two workgroups per CU walk the problems (grid = 2 x CUs) instead of one workgroup per problem
#endif

template <typename T, int D>
HSTU_DEV void quad_problem(const HstuAttnBwdParams& bp, int tmax, int uh, char* smem, int tid, int lane, int wave) {
  using C = BwdCfg<T, D, D>;
  using Q = QuadCfg<T, D>;
  static_assert(C::EB == 2, "16-bit I/O");
  const HstuAttnParams& p = bp.fwd;
  const int b = user_of_slot(p, uh / p.heads), hd = uh % p.heads;
  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = min((int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0), 32 * tmax);
  if (len <= 0) return;
  const MaskCtx mc = make_mask_ctx(p, b, len);
  HSTU_TRACE_DECL(bp.workspace, false);
  const int nt = (len + 31) >> 5;        // <= tmax <= 7
  char* const stage = smem + Q::kMaxTiles * C::PAIR;
  char* const dsbuf = stage + C::PAIR;

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
  const bool dma_fast = q_rs < (1 << 24) && k_rs < (1 << 24) && v_rs < (1 << 24) && do_rs < (1 << 24) &&
                        (int64_t)len_max * q_rs < (1LL << 32) && (int64_t)len_max * k_rs < (1LL << 32) &&
                        (int64_t)len_max * v_rs < (1LL << 32) && (int64_t)len_max * do_rs < (1LL << 32);
  auto stage_dma = [&](int qt) {
    quad_tile_dma<T, D>(stage, qbase, q_rs, 32 * qt, len, wave, lane, dma_fast);
    quad_tile_dma<T, D>(stage + C::KT, dobase, do_rs, 32 * qt, len, wave, lane, dma_fast);
  };
  // ---- prologue: the whole K/V block and the first query tile, all by LDS-DMA
  for (int t = 0; t < ((QUAD_ABLATE & 16) ? 0 : nt); ++t) {
    char* dst = smem + t * C::PAIR;
    quad_tile_dma<T, D>(dst, kbase, k_rs, 32 * t, len, wave, lane, dma_fast);
    quad_tile_dma<T, D>(dst + C::KT, vbase, v_rs, 32 * t, len, wave, lane, dma_fast);
  }
  stage_dma(nt - 1);
  for (int i = tid; i < Q::kMaxTiles * Q::DSB / 16; i += kQuadThreads) *LDS_PTR(u32x4, dsbuf + 16 * i) = u32x4{0u, 0u, 0u, 0u};
  // K tiles of unused slots: the dQ GEMM reads every slot (times a zeroed dS' fragment): finite values only
  for (int t = nt; t < Q::kMaxTiles; ++t)
    for (int i = tid; i < C::KT / 16; i += kQuadThreads) *LDS_PTR(u32x4, smem + t * C::PAIR + 16 * i) = u32x4{0u, 0u, 0u, 0u};

  // key tiles of this wave: A = wave, B = 6 - wave (waves 0..2); their accumulators are two separate register sets
  const int tA = wave, tB = wave < 3 ? 6 - wave : -1;
  f32x16 dkA[C::DBQ], dvA[C::DBV], dkB[C::DBQ], dvB[C::DBV];
#pragma unroll
  for (int d = 0; d < C::DBQ; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkA[d][r] = 0.f; dvA[d][r] = 0.f; dkB[d][r] = 0.f; dvB[d][r] = 0.f; }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // Q/dO tile i (and, first time, K/V) landed; every dQ GEMM of step i+1 is done
    const int fin = i + 1;          // key tile that became final in the previous step
    if (fin < nt) {
      // its dV was parked at the end of that step; its K tile is dead now: dK follows
      int lane3 = lane;
      asm volatile("" : "+v"(lane3));
      if (fin == tA) fold_park_tile<T, D>(dkA, ds_scale, smem + fin * C::PAIR, lane3);
      else if (fin == tB) fold_park_tile<T, D>(dkB, ds_scale, smem + fin * C::PAIR, lane3);
    }
    // ---- phase 1: the pairs (i, t) of this wave's tiles
    if (!(QUAD_ABLATE & 64) && tA <= i && (mc.win == 0 || mc.pair_may_be_active(32 * i, 32, 32 * tA, 32))) {
      const char* Kw = smem + tA * C::PAIR;
      int lane1 = lane;
      asm volatile("" : "+v"(lane1));
      fold_pair<T, D, D>(p, mc, Kw, Kw + C::KT, stage, stage + C::KT, dsbuf + tA * Q::DSB, 32 * i, 32 * tA, dkA, dvA, lane1, dmvm HSTU_TRACE_PASS);
    }
    if (!(QUAD_ABLATE & 64) && tB >= 0 && tB <= i && (mc.win == 0 || mc.pair_may_be_active(32 * i, 32, 32 * tB, 32))) {
      const char* Kw = smem + tB * C::PAIR;
      int lane1 = lane;
      asm volatile("" : "+v"(lane1));
      fold_pair<T, D, D>(p, mc, Kw, Kw + C::KT, stage, stage + C::KT, dsbuf + tB * Q::DSB, 32 * i, 32 * tB, dkB, dvB, lane1, dmvm HSTU_TRACE_PASS);
    }
    __syncthreads();   // dS' of this step published; stage reads done; dK of tile i+1 parked
    if (i > 0) stage_dma(i - 1);
    if (fin < nt) {
      quad_copy_out<T, D>(smem + fin * C::PAIR, dk_head + (int64_t)(32 * fin) * dk_rs, dk_rs, len - 32 * fin, tid);
      quad_copy_out<T, D>(smem + fin * C::PAIR + C::KT, dv_head + (int64_t)(32 * fin) * dv_rs, dv_rs, len - 32 * fin, tid);
    }
    // ---- phase 2: dQ of query tile i
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    if (!(QUAD_ABLATE & 32)) quad_dq_phase<T, D>(bp, mc, smem, dsbuf, i, wave, off0, hd, ds_scale, lane2);
    // key tile i is final: V tiles are read by their owner's pairs only, and this was its last one
    {
      int lane3 = lane;
      asm volatile("" : "+v"(lane3));
      if (i == tA) fold_park_tile<T, D>(dvA, scale_v, smem + i * C::PAIR + C::KT, lane3);
      else if (i == tB) fold_park_tile<T, D>(dvB, scale_v, smem + i * C::PAIR + C::KT, lane3);
    }
  }
  // ---- tile 0: dV parked, dK still in wave 0's registers
  __syncthreads();     // every dQ GEMM is done: K tile 0 is dead
  if (wave == 0) {
    int lane4 = lane;
    asm volatile("" : "+v"(lane4));
    fold_park_tile<T, D>(dkA, ds_scale, smem, lane4);
  }
  __syncthreads();
  quad_copy_out<T, D>(smem, dk_head, dk_rs, len, tid);
  quad_copy_out<T, D>(smem + C::KT, dv_head, dv_rs, len, tid);
}

template <typename T, int D>
__global__ __launch_bounds__(kQuadThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void hstu_attn_bwd_quad_kernel(const HstuAttnBwdParams bp, int tmax) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (QUAD_PERSIST) {
    const int total = bp.fwd.batch * bp.fwd.heads;
    for (int uh = blockIdx.x; uh < total; uh += gridDim.x) {
      int uh_l = uh;
      asm volatile("" : "+s"(uh_l));
      quad_problem<T, D>(bp, tmax, uh_l, smem, tid, lane, wave);
      __syncthreads();
    }
  } else {
    quad_problem<T, D>(bp, tmax, blockIdx.x, smem, tid, lane, wave);
  }
}

}  // namespace hstu
