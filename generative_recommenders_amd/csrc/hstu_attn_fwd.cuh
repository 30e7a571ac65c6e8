// HSTU attention forward for gfx950 (hand-written, MFMA 32x32, wave64).
//
//   O[i,:] = sum_j silu(alpha <q_i,k_j>) * scale * M[i,j] * v_j        (per user, per head)
//
// Replaces triton_hstu_attention_fwd / triton_cached_hstu_mha
// (ops/triton/triton_hstu_attention.py:1767-1846, 2095-2170) and hstu::hstu_mha_fwd
// (ops/cpp/hstu_attention/flash_api.cpp:34-110).  Semantics follow the reference
// PyTorch path, ops/pytorch/pt_hstu_attention.py:129-235.
//
// Mapping.  One workgroup (4 waves) = one (user, head, block of 128 query rows);
// wave w owns query rows [q0+32w, q0+32w+32).  Everything is computed "transposed"
// so that the query index lives on the lane axis of every fragment:
//     S^T  = K_j  Q^T      A = K tile rows (LDS, b128), B = Q fragment (registers)
//     P^T  = silu(alpha S^T) * scale * M      in the MFMA C layout == the B layout
//                                              of the next MFMA (no shuffles, no LDS)
//     O^T += V_j^T P^T     A = V tile through the LDS transpose read
// K/V tiles of 32 keys stream through a 3-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4 through inline asm,
// source addresses from a 2-register per-lane plan): two tiles in flight per workgroup, one raw s_barrier per tile with a
// counted s_waitcnt vmcnt; three workgroups per CU (165 VGPRs).  (2 stages when a stage exceeds 16 KiB: FwdCfg.)
// HSTU has no softmax, so there is no running max / rescale: tiles are independent.
#pragma once
#include "hstu_common.cuh"

namespace hstu {

constexpr int kFwdThreads = 256;
#ifndef FWD_ABLATE
#define FWD_ABLATE 0   // timing experiments only (wrong results): 1 no output stores, 2 no MFMA / element-wise work (loads and barriers only), 4 no K/V loads
#endif
#ifndef HSTU_FWD_MIN_WAVES
#define HSTU_FWD_MIN_WAVES 2
#endif
// (round 6: the measured-and-lost switches of this kernel -- K tiles three ahead behind a second barrier, first tiles requested in front of
// the Q rows, wave priority by phase, the conflict-free park of the output tile, the builtin form of the DMA instruction -- left the
// source; the version that carried them: docs/experiments/r06_hstu_attn_fwd_with_switches.cuh.txt, results docs/EXPERIMENTS.md A.3, R5.7)
constexpr int kFwdRowsPerBlock = 128;

template <typename T, int DQK, int DV>
struct FwdCfg {
  static constexpr int EB = Elem<T>::kBytes;
  static constexpr int EPU = 16 / EB;            // elements per 16-byte unit
  static constexpr int UPR_K = DQK * EB / 16;    // units per K row
  static constexpr int UPR_V = DV * EB / 16;
  static constexpr int KT = 32 * DQK * EB;       // bytes of a 32-row K tile
  static constexpr int VT = 32 * DV * EB;
  static constexpr int STAGE = KT + VT;
  static constexpr int KG = DQK / 16;            // 16-wide contraction groups of QK^T
  static constexpr int DB = DV / 32;             // 32-wide output blocks
  static constexpr int NKU = (32 * UPR_K + kFwdThreads - 1) / kFwdThreads;  // staged units / thread
  static constexpr int NVU = (32 * UPR_V + kFwdThreads - 1) / kFwdThreads;
  // K/V ring filled by LDS-DMA.  Counted vmcnt waits need every wave to issue the same number of
  // DMA instructions per tile (whole multiples of the 4 waves); otherwise depth 2 with full drains.
  static constexpr int NCH_K = 32 * UPR_K / 64, NCH_V = 32 * UPR_V / 64;
  static constexpr bool COUNTED = (NCH_K % 4 == 0) && (NCH_V % 4 == 0);
  static constexpr int PER_TILE = NCH_K / 4 + NCH_V / 4;      // DMA instructions per wave per tile
  static constexpr int NS = (COUNTED && STAGE <= 16384) ? 3 : 2;  // ring depth (= tiles in flight + 1)
  static constexpr int SMEM = NS * STAGE;
};

// Cooperative global -> register load of one [32][D] tile and the matching register -> LDS
// write.  The loads are UNCONDITIONAL on clamped (always valid) addresses so that nothing
// consumes the loaded registers until tile_lds_write: the global loads stay in flight across
// the whole compute phase (a predicated `ok ? load : 0` forces an s_waitcnt right at the load).
// Zero-fill (rows >= len, columns >= the real head dim) is applied at write time.
template <typename T, int D, int NU, int NTHREADS>
HSTU_DEV void tile_gload(u32x4 (&reg)[NU], const char* base, int64_t row_stride_bytes, int row0, int len,
                         int real_d, int tid) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int t = 0; t < NU; ++t) {
    const int u = min(tid + t * NTHREADS, 32 * UPR - 1);
    const int row = min(row0 + u / UPR, len - 1);
    const int unit = ((u % UPR) * EPU < real_d) ? (u % UPR) : 0;
    reg[t] = gload16(base + (int64_t)row * row_stride_bytes + unit * 16);
  }
}

template <typename T, int D, int NU, int NTHREADS>
HSTU_DEV void tile_lds_write(const u32x4 (&reg)[NU], char* tile, int row0, int len, int real_d, int tid) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int EPU = 16 / Elem<T>::kBytes;
#pragma unroll
  for (int t = 0; t < NU; ++t) {
    const int u = tid + t * NTHREADS;
    if (u < 32 * UPR) {
      const int row = u / UPR, unit = u % UPR;
      const bool ok = (row0 + row < len) & (unit * EPU < real_d);
      const u32x4 z = {0u, 0u, 0u, 0u};
      *LDS_PTR(u32x4, tile + tile_off<UPR>(row, unit)) = ok ? reg[t] : z;
    }
  }
}

// The DMA instruction itself goes through inline asm (M0 = LDS address of the 1 KiB chunk, 32-bit lane offset, 64-bit
// scalar base).  Why: for the builtin, hipcc's waitcnt pass assumes that ANY later LDS read may alias the chunk in flight
// and puts an `s_waitcnt vmcnt(0)` in front of it -- in the forward's loop that was the first transposed V read of every
// key tile, i.e. each wave waited for the two tiles it had just requested before finishing the current one, and the
// three-deep ring never had more than the current tile's latency of cover.  The asm form is invisible to that pass; the
// kernel orders the consumers itself (counted vmcnt + barrier at the top of every tile).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// HSTU_DMA_NT: the non-temporal hint on the tile requests -- every byte of q / k / v / dO is read once, by one CU
#ifndef HSTU_DMA_NT
#define HSTU_DMA_NT 0
#endif
#if HSTU_DMA_NT
#define HSTU_DMA_POLICY " nt"
#else
#define HSTU_DMA_POLICY ""
#endif
HSTU_DEV void dma16_saddr_asm(uint32_t off, const char* base, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" HSTU_DMA_POLICY ::"v"(off), "s"(base), "s"(lds_base) : "memory", "m0");
}
HSTU_DEV void dma16_vaddr_asm(const char* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" HSTU_DMA_POLICY ::"v"(g), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop

// LDS-DMA variant (global_load_lds_dwordx4): wave `wave` of `nwaves` moves 1 KiB chunks of a
// [32][D] tile from global memory straight into LDS -- no VGPRs, no ds_write, completion counted
// by vmcnt (the compiler waits for it before the next __syncthreads()).  The hardware writes lane
// l's 16 bytes to (wave-uniform base) + 16 l, so the XOR swizzle of tile_off is applied on the
// SOURCE side: the lane that fills physical slot s of a row fetches logical unit s ^ swizzle(row).
// No zero fill is possible: rows past `len` / columns past the real head dim receive a clamped
// (valid, finite) copy, so callers must MASK such keys instead of relying on zeros.
template <typename T, int D>
HSTU_DEV void tile_dma(char* tile, const char* base, int64_t row_stride_bytes, int row0, int len, int real_d,
                       int wave, int nwaves, int lane) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int EPU = 16 / Elem<T>::kBytes;
  constexpr int NCH = 32 * UPR / 64;   // 1 KiB chunks per tile
  for (int c = wave; c < NCH; c += nwaves) {
    const int pidx = c * 64 + lane;
    const int row = pidx / UPR, slot = pidx % UPR;
    const int unit = slot ^ swz<UPR>(row);
    const int grow = min(row0 + row, len - 1);
    const int gunit = (unit * EPU < real_d) ? unit : 0;
    const char* g = base + (int64_t)grow * row_stride_bytes + gunit * 16;
    dma16_vaddr_asm(g, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tile) + c * 1024);
  }
}

// The same with the lane-constant part of the address planned ahead (NI chunks per wave: c = wave + i nwaves): the lane's row
// inside the tile and the byte offset of its (swizzled, clamped) unit.  A chunk then costs add + min + a 24-bit multiply-add,
// and the load takes its 64-bit base from SGPRs (global_load_lds ... v_off32, s[base]).
// Chunk i of a wave is chunk `wave + i nwaves`: with 4 waves and 16-unit rows that is 16 i rows further down, and the swizzle
// only looks at the row modulo 16 -- the unit offset is the same for every chunk and the row advances by a constant, so the
// plan is TWO registers per tensor (one set for K and V when their head dims agree).
template <int NI> struct DmaPlan { int rl0; uint32_t uo0; int rstep; };

template <typename T, int D, int NI>
HSTU_DEV void dma_plan(DmaPlan<NI>& pl, int real_d, int wave, int nwaves, int lane) {
  constexpr int UPR = D * Elem<T>::kBytes / 16;
  constexpr int EPU = 16 / Elem<T>::kBytes;
  const int pidx = wave * 64 + lane;
  const int row = pidx / UPR, slot = pidx % UPR;
  const int unit = slot ^ swz<UPR>(row);
  pl.rl0 = row;
  pl.uo0 = (unit * EPU < real_d) ? unit * 16 : 0;
  pl.rstep = nwaves * 64 / UPR;
}

template <int NI>
HSTU_DEV void tile_dma_fast(char* tile, const char* base, uint32_t row_stride_bytes, int row0, int len, const DmaPlan<NI>& pl,
                            int wave, int nwaves) {
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tile);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const uint32_t grow = (uint32_t)min(row0 + i * pl.rstep + pl.rl0, len - 1);
    const uint32_t off = __umul24(grow, row_stride_bytes) + pl.uo0;
    dma16_saddr_asm(off, base, lds0 + (wave + i * nwaves) * 1024);
  }
}

// Row fragment straight from global memory (unconditional, caller clamps the address):
// raw 16-byte pieces first, converted / zeroed by finish_row_frag once all are in flight.
template <typename T> struct RawFrag { u32x4 x0, x1; };

template <typename T>
HSTU_DEV RawFrag<T> global_row_frag_issue(const char* row_ptr, int e0, bool second_ok) {
  RawFrag<T> r;
  if constexpr (Elem<T>::kBytes == 2) {
    r.x0 = gload16(row_ptr + e0 * 2);
    r.x1 = r.x0;
  } else {
    r.x0 = gload16(row_ptr + e0 * 4);
    r.x1 = gload16(row_ptr + e0 * 4 + (second_ok ? 16 : 0));   // never read past the row
  }
  return r;
}

// `ok1` covers the second 16-byte piece of an fp32 fragment (the real head dim is a multiple of
// 4 elements only, so a fragment may straddle it; the other operand is NOT zero padded when it
// comes through LDS-DMA, so this one must be).
template <typename T>
HSTU_DEV typename Elem<T>::Frag finish_row_frag(const RawFrag<T>& r, bool ok, bool ok1) {
  typename Elem<T>::Frag f;
  const u32x4 z = {0u, 0u, 0u, 0u};
  if constexpr (Elem<T>::kBytes == 2) {
    f.v = __builtin_bit_cast(typename Elem<T>::vec8, ok ? r.x0 : z);
  } else {
    f32x4 a = __builtin_bit_cast(f32x4, ok ? r.x0 : z), b = __builtin_bit_cast(f32x4, (ok && ok1) ? r.x1 : z);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
  }
  return f;
}

// Store a transposed accumulator block: this lane holds, for output row `row_ptr`,
// the 4 consecutive columns d0..d0+3 in acc[4*rq .. 4*rq+3].
template <typename T>
HSTU_DEV void store4(char* row_ptr, int d0, float x0, float x1, float x2, float x3) {
  if constexpr (Elem<T>::kBytes == 2) {
    u32x2 v = {Elem<T>::pk2(x0, x1), Elem<T>::pk2(x2, x3)};
    *reinterpret_cast<u32x2*>(row_ptr + d0 * 2) = v;
  } else {
    f32x4 v = {x0, x1, x2, x3};
    *reinterpret_cast<f32x4*>(row_ptr + d0 * 4) = v;
  }
}

// PRECISE (HSTU_ATTN_PRECISE=1; 16-bit I/O without bias): P' enters the second MFMA as TWO 16-bit fragments, its rounded value
// and the rounding's remainder -- the kernel's own error (P' rounded to the I/O dtype before O += V^T P'^T, 1.66e-3 relative
// for bf16: as large as the rounding of the output itself) drops out, at the price of 8 more registers (two waves per SIMD
// instead of three) and a second PV MFMA per fragment.
template <typename T, int DQK, int DV, bool BIAS = false, bool HEADS = false, bool PRECISE = false>
__global__ __launch_bounds__(kFwdThreads, PRECISE ? 2 : HSTU_FWD_MIN_WAVES) void hstu_attn_fwd_kernel(const HstuAttnParams p, int nqb, int bucket_cache_off) {
  using C = FwdCfg<T, DQK, DV>;
  using E = Elem<T>;
  using Frag = typename E::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31;

  // ---- work decode: 8 consecutive (user,head) pairs share a dispatch group so that the
  // query blocks of one (user,head) land on the same XCD (block id mod 8) back to back and
  // re-read K/V from that XCD's L2; heavier (later) query blocks are dispatched first.
  const int bid = blockIdx.x;
  const int grp = bid / (8 * nqb), rem = bid % (8 * nqb);
  const int qb = nqb - 1 - rem / 8;
  const int uh = grp * 8 + (rem & 7);
  // bucket_cache_off > 0 (research-path bias, short sequences): ONE workgroup walks all heads of a (user, query block).
  // The time bucket of an element depends on the user only; computing it (timestamp read, hardware log2, the exactness
  // check: ~12 VALU instructions) is 0.64 of the 0.87 ms the bias adds to this kernel at the ML-20M shape, the table
  // lookups 0.08.  The first head leaves one byte per element in LDS (wave w of query block qb keeps its 4 qb + w + 1
  // key tiles: 1 KiB each), the other heads read 8 bytes per lane and half tile; the tables are staged once.
  constexpr bool head_loop = BIAS && HEADS;   // (its own instantiation: the plain bias kernel keeps its 125 registers / 4 waves per SIMD)
  if (uh >= (head_loop ? p.batch : p.batch * p.heads)) return;
  const int b = user_of_slot(p, head_loop ? uh : uh / p.heads), hd_first = head_loop ? 0 : uh % p.heads;
  const int n_heads = head_loop ? p.heads : 1;

  const int64_t off0 = load_index(p.seq_offsets, b, p.offsets_dtype);
  const int len = (int)(load_index(p.seq_offsets, b + 1, p.offsets_dtype) - off0);
  const int nq_rows = p.delta_q > 0 ? min(p.delta_q, len) : len;
  const int i_shift = p.delta_q > 0 ? len - nq_rows : 0;      // logical position of q row 0
  const int64_t q_base = p.delta_q > 0 ? (int64_t)b * p.delta_q + (p.delta_q - nq_rows) : off0;
  const int q0 = qb * kFwdRowsPerBlock;
  if (q0 >= nq_rows) return;

  const MaskCtx mc = make_mask_ctx(p, b, len);
  const float scale_v = attn_scale_of(p);
#ifdef HSTU_TRACE
  // blocks 4096 / 4104: the heavy and the light query block of one (user, head); rows 0-3 / 4-7 of the trace buffer
  HSTU_TRACE_DECL(g_hstu_trace_fwd + (blockIdx.x == 4104 ? 4 * 256 : 0), g_hstu_trace_fwd != nullptr && (blockIdx.x == 4096 || blockIdx.x == 4104));
#endif
  HSTU_MARK(1);
  // Which 32 rows of the block the wave owns.  Wave w of a workgroup runs on SIMD w, and with plain causal masks the
  // row tile 4 qb + w costs 4 qb + w + 1 key tiles: every workgroup puts its lightest tile on SIMD 0 and its heaviest on
  // SIMD 3.  Odd query blocks hand their row tiles out in reverse (among the waves that have rows), so
  // that the two blocks of a 200-row sequence load the SIMDs 1+7, 2+6, 3+5, 4+0 instead of 1+5, 2+6, 3+7, 4+0.
  const int na_blk = (min(kFwdRowsPerBlock, nq_rows - q0) + 31) >> 5;
  const int vw = ((qb & 1) && wave < na_blk) ? na_blk - 1 - wave : wave;
  const int r0 = q0 + 32 * vw;                   // first q row of this wave
  const bool wave_active = r0 < nq_rows;
  const int my_row = r0 + n32;
  const bool row_ok = my_row < nq_rows;
  const int qi = my_row + i_shift;               // logical position of this lane's query

  const int qi_id = mc.id_of(qi);
  BiasCtx bc;
  int64_t t_q1 = 0;
  int t_q32 = 0;
  if constexpr (BIAS) {
    // tables + this user's timestamps -> LDS behind the K/V ring (every thread of the workgroup is here: the early
    // returns above are workgroup-uniform)
    bc = stage_bias_tables(p, b, smem + C::SMEM, tid, kFwdThreads);
    const int64_t* tr = bias_ts_row(p, b);
    t_q1 = tr ? tr[min(max(qi + 1, 0), p.max_seq_len - 1)] : 0;   // the row uses the NEXT item's timestamp
    lds_barrier();
    bc.finish(kFwdThreads / 64);
    if (bc.small) t_q32 = bc.t32_at(qi + 1);
  }

  // ---- key range visited by this workgroup (conservative; the per-element mask is exact)
  const int i_first = q0 + i_shift;
  const int i_last = min(q0 + kFwdRowsPerBlock, nq_rows) - 1 + i_shift;
  const bool ctx_rows = mc.ctx > 0 && i_first < mc.ctx;
  const int kv_hi = ctx_rows ? len : min(len, i_last + 1);
  int kv_lo = 0;
  if (mc.win > 0 && mc.full == 0 && !ctx_rows) {
    const int x = mc.id_of(i_first) - mc.win;
    const int pos = x <= 0 ? 0 : (mc.ctx > 0 ? x + mc.ctx - 1 : x);
    kv_lo = (pos >> 5) << 5;
  }
  const int ntiles = (kv_hi - kv_lo + 31) >> 5;

  // bucket bytes of this wave: tiles 0 .. (4 qb + wave) of its query tile, behind those of the waves before it
  char* const bcache = smem + C::SMEM + bucket_cache_off + (vw * (4 * qb + 1) + ((vw * (vw - 1)) >> 1)) * 1024;

  const int lane_wg = lane;
  for (int hi = 0; hi < n_heads; ++hi) {
  const int hd = hd_first + hi;
  const bool bkt_cached = head_loop && hi > 0;
  // (the lane id is laundered per head: per-lane LDS offsets are recomputed there instead of being hoisted out of the head
  // loop and kept alive across the tile loop)
  int lane_h = lane_wg;
  if constexpr (head_loop) asm volatile("" : "+v"(lane_h));
  const int lane = lane_h, n32 = lane & 31, hf = lane >> 5;
  Frag qf[C::KG];
#define HSTU_FWD_LOAD_Q()                                                                                                   \
  {                                                                                                                          \
    const int ld_row = min(my_row, nq_rows - 1); /* clamped: always a valid row of this user */                              \
    const char* qrow = (const char*)p.q + ((q_base + ld_row) * p.q_row_stride + (int64_t)hd * p.q_head_stride) * C::EB;     \
    RawFrag<T> raw[C::KG];                                                                                                   \
    _Pragma("unroll") for (int kg = 0; kg < C::KG; ++kg) {                                                                   \
      const int e0 = hf * (DQK / 2) + kg * 8;                                                                                \
      raw[kg] = global_row_frag_issue<T>(qrow, e0 < p.dqk ? e0 : 0, e0 + 4 < p.dqk);                                        \
    }                                                                                                                        \
    _Pragma("unroll") for (int kg = 0; kg < C::KG; ++kg) {                                                                   \
      const int e0 = hf * (DQK / 2) + kg * 8;                                                                                \
      qf[kg] = finish_row_frag<T>(raw[kg], row_ok && e0 < p.dqk, e0 + 4 < p.dqk);                                           \
    }                                                                                                                        \
  }
  // ---- Q fragment of this wave (B operand of S^T = K Q^T): lane (q = n32, hf) holds elements hf*DQK/2 + 8*kg .. +8 of its
  // row: one contiguous half row per lane.  (Requesting the first K/V tiles BEFORE the Q rows measured 1.6 % slower, twice.)
  HSTU_FWD_LOAD_Q()
  HSTU_MARK(2);
  const char* kbase = (const char*)p.k + (off0 * p.k_row_stride + (int64_t)hd * p.k_head_stride) * C::EB;
  const char* vbase = (const char*)p.v + (off0 * p.v_row_stride + (int64_t)hd * p.v_head_stride) * C::EB;
  const int64_t k_rs = p.k_row_stride * C::EB, v_rs = p.v_row_stride * C::EB;

  f32x16 oacc[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;

  // plain causal, rows aligned with the key tiles (no delta_q shift, no sliding-window start), |alpha| in the range where
  // alpha * 1e30 neither overflows nor loses the mask: the diagonal tile's mask is a lane constant (mode 4 below)
  const float aabs = fabsf(p.alpha);
  // (target rows only -- no window, no contextual rows, no delta: DLRM-v3's call -- and every row of this wave in front of the first
  // target: the wave's masks are the plain causal ones, it takes the plain path's two compares per tile instead of the general
  // predicates' ~100 scalar instructions, and its diagonal tile the lane-constant pattern)
  const bool wave_plain = mc.simple || (HSTU_TARGETS_PLAIN && mc.has_targets && mc.win == 0 && mc.ctx == 0 && i_shift == 0 && r0 + 32 <= min(len, mc.max_id));
  const bool diag_fast = !BIAS && wave_plain && i_shift == 0 && kv_lo == 0 && aabs > 1e-20f && aabs < 1e6f;

  // ---- K/V tiles stream through an NS-deep LDS ring filled by LDS-DMA: tiles t+1 .. t+NS-1 are in
  // flight while tile t is computed; one raw barrier per tile, loads are never drained in the loop
  // The source address of a chunk is (uniform tile base) + (row of the lane) x (row stride) + (swizzled unit) x 16.  Written
  // naively that is ~20 VALU instructions per chunk -- a 64-bit multiply-add among them -- i.e. 80 per key tile and wave next
  // to the ~90 of the tile's element-wise block (docs/EXPERIMENTS.md A.2: 1.400 -> 1.336 ms).
  // The lane's row inside the tile and its unit's byte offset never change: they are planned once per head, and a chunk
  // costs an add, a min and one 24-bit multiply-add into a 32-bit offset from the head's (scalar) base pointer.  Needs the
  // user's rows to span < 4 GiB and strides < 16 MiB (else the general path).
  constexpr int NIK = C::COUNTED ? C::NCH_K / 4 : 1, NIV = C::COUNTED ? C::NCH_V / 4 : 1;
  // (the two-register plan needs a wave's chunks to lie a multiple of 16 rows apart: 16-bit head dims 64 / 128, not fp32 rows)
  constexpr bool plan_ok = (NIK == 1 || (4 * 64 / C::UPR_K) % 16 == 0) && (NIV == 1 || (4 * 64 / C::UPR_V) % 16 == 0);
  const bool dma_fast = C::COUNTED && plan_ok && k_rs < (1 << 24) && v_rs < (1 << 24) &&
                        (int64_t)len * k_rs < (1LL << 32) && (int64_t)len * v_rs < (1LL << 32);   // workgroup-uniform
  DmaPlan<NIK> plk;
  DmaPlan<NIV> plv;
  if (dma_fast) {
    dma_plan<T, DQK, NIK>(plk, p.dqk, wave, 4, lane);
    if (DQK == DV && p.dqk == p.dv) plv = DmaPlan<NIV>{plk.rl0, plk.uo0, plk.rstep};   // (the same registers)
    else dma_plan<T, DV, NIV>(plv, p.dv, wave, 4, lane);
  }
  auto issue_tile = [&](int t, int slot) {
    if (FWD_ABLATE & 4) return;   // (timing experiment: no K/V loads)
    char* st = smem + slot * C::STAGE;
    if (dma_fast) {
      tile_dma_fast<NIK>(st, kbase, (uint32_t)k_rs, kv_lo + 32 * t, len, plk, wave, 4);
      tile_dma_fast<NIV>(st + C::KT, vbase, (uint32_t)v_rs, kv_lo + 32 * t, len, plv, wave, 4);
      return;
    }
    tile_dma<T, DQK>(st, kbase, k_rs, kv_lo + 32 * t, len, p.dqk, wave, 4, lane);
    tile_dma<T, DV>(st + C::KT, vbase, v_rs, kv_lo + 32 * t, len, p.dv, wave, 4, lane);
  };
  // (K tiles three ahead -- K(t+3) requested in the middle of step t behind a second barrier, +12 % bytes in flight in the same LDS --
  // measured bit-identical and 0.7 % / 3 % SLOWER: the second barrier costs more than the lead buys, profiles/r03_ab_fwd_k_early.txt)
  for (int t = 0; t < C::NS - 1 && t < ntiles; ++t) issue_tile(t, t);
#undef HSTU_FWD_LOAD_Q
  HSTU_MARK(3);

  // one key tile; `slot` = t % NS, as a compile-time constant in the unrolled loop (SLOT >= 0) or at run time
  // (unrolling this loop by the ring depth -- ring slots as immediate LDS offsets -- triples the code and measured nothing)
  for (int t = 0; t < ntiles; ++t) {
    const int slot = t % C::NS;
    const int j0 = kv_lo + (t << 5);
    // tile t has landed once at most (tiles issued after it) * PER_TILE DMA instructions are pending
    if constexpr (C::COUNTED) {
      const int newer = min(C::NS - 2, ntiles - 1 - t);     // tiles issued after tile t so far
      if (newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_TILE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // every wave's chunks of tile t landed; stage (t-1) % NS is free
    asm volatile("" ::: "memory");
    if (t + C::NS - 1 < ntiles) issue_tile(t + C::NS - 1, (slot + C::NS - 1) % C::NS);
    HSTU_MARK(10);
    // (scalar work is not free: the general tile predicates cost ~100 SALU instructions per tile; plain-causal
    // batches -- no targets, window or contextual rows -- take two compares instead)
    const int i0w = r0 + i_shift;
    bool tile_act, tile_full;
    if (wave_plain) {
      tile_act = i0w < len && j0 <= min(i0w + 31, len - 1);
      tile_full = j0 + 32 <= i0w;     // strictly below this wave's first row (then also j0 + 32 <= len)
    } else {
      tile_act = mc.pair_may_be_active(i0w, 32, j0, 32);
      tile_full = tile_act && mc.pair_fully_valid(i0w, 32, j0, 32);
    }
    if (wave_active && tile_act && !(FWD_ABLATE & 2)) {
      const char* Kt = smem + slot * C::STAGE;
      const char* Vt = Kt + C::KT;
      // ONE accumulator chain: back-to-back dependent MFMAs forward their result, and the VALU cycles a second chain
      // costs (16 adds per tile and lane) are what the long-sequence forward is bound by (N = 8192: +2..4 %)
      // mode (wave-uniform): 0 no mask needed, 1 plain causal by compares, 2 general mask algebra, 3 targets / window by integer
      // arithmetic, 4 plain causal with tile-aligned rows: the diagonal tile's mask is put into S itself
      // (as hstu_attn_bwd_fold.cuh does through the accumulator's start value: a masked element is -1e30, alpha S is hugely negative, exp2 gives +inf,
      // the sigmoid exactly 0 and P' = x * 0 = -0 -- the element-wise block needs no mask code; the predicate, key
      // (r&3) + 8 (r>>2) + 4 hf <= query n32, is a compare against a lane constant).  Every other tile such a wave visits lies
      // strictly below its rows: no mask at all (rows past the sequence end are zero-filled: silu(0) = 0).
      const int mode = tile_full ? 0 : (wave_plain ? (diag_fast ? 4 : 1) : (mc.ctx == 0 ? 3 : 2));
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int kg = 0; kg < C::KG; ++kg) {
        Frag a = lds_row_frag<T, C::UPR_K>(Kt, n32, hf * (DQK / 2) + kg * 8);
        s = E::mma(a, qf[kg], s);
      }
      // (requesting the K fragments 2 / 3 / 4 / 8 ahead of the chain's MFMAs, or V fragments before the element-wise block:
      // measured, no gain or a 4th register bank -- 169 registers cost a wave per SIMD and 22 %: docs/EXPERIMENTS.md)
      if (mode == 4) {
        const float neg = p.alpha < 0.f ? 1e30f : -1e30f;
        const int x = n32 - 4 * hf;           // key (r&3) + 8 (r>>2) + 4 hf > query n32  <=>  (r&3) + 8 (r>>2) > x
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = ((r & 3) + 8 * (r >> 2) > x) ? neg : s[r];
      }
      HSTU_MARK(11);
      Frag pb[2];
      [[maybe_unused]] Frag pbl[2];   // PRECISE: the remainders of P' after its rounding to the I/O dtype
#pragma unroll
      for (int h8 = 0; h8 < 2; ++h8) {   // two halves keep only 8 fp32 temporaries live
        float pv[8];
        if constexpr (BIAS) {
#ifndef FWD_BIAS_ABLATE
#define FWD_BIAS_ABLATE 0   // timing experiments only (wrong results): 1 bucket 0 for every element, 2 no table lookups
#endif
          // two straight-line variants of the half tile (wave-uniform choice): buckets read from the user's byte matrix,
          // or computed and left there -- the bucket goes straight into its element's value, no array of 8 stays live
          char* const bslot = bcache + (2 * t + h8) * 512 + 8 * lane;
          if (bkt_cached) {
            const u32x2 w = *LDS_PTR(const u32x2, bslot);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int r = 8 * h8 + j;
              const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
              const int bkt = (int)((w[j >> 2] >> (8 * (j & 3))) & 255u);
              float x = s[r] * p.alpha;
              if (!(FWD_BIAS_ABLATE & 2)) x += bc.value(bc.pos_index(qi, key), bkt);
              pv[j] = x * fast_sigmoid(x);
            }
          } else {
            u32x2 w = {0u, 0u};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int r = 8 * h8 + j;
              const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
              const int bkt = (FWD_BIAS_ABLATE & 1) ? 0 : (bc.small ? bc.bucket32(t_q32, bc.t32_at(key)) : bc.bucket(t_q1, bc.ts_at(key)));   // wave-uniform choice
              w[j >> 2] |= (unsigned)bkt << (8 * (j & 3));
              float x = s[r] * p.alpha;
              if (!(FWD_BIAS_ABLATE & 2)) x += bc.value(bc.pos_index(qi, key), bkt);
              pv[j] = x * fast_sigmoid(x);
            }
            if (head_loop) *LDS_PTR(u32x2, bslot) = w;
          }
        } else {
          // two elements per instruction where the ISA has a packed fp32 form, and the exponent's argument straight from S
          // (one multiply by -alpha log2 e instead of two): 4 VALU instructions per element, two of them transcendental
          const f32x2 a2 = {p.alpha, p.alpha};
          const f32x2 c2 = {-1.44269504088896340736f * p.alpha, -1.44269504088896340736f * p.alpha};
          const f32x2 one2 = {1.f, 1.f};
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const f32x2 sv = {s[8 * h8 + j], s[8 * h8 + j + 1]};
            const f32x2 x = sv * a2, tt = sv * c2;
            const f32x2 e = {__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])};
            const f32x2 dn = e + one2;
            const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
            const f32x2 pr = x * sg;
            pv[j] = pr[0];
            pv[j + 1] = pr[1];
          }
        }
        if (mode == 1) {          // plain causal: key <= query, both in range
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            pv[j] = (row_ok & (key < len) & (key <= qi)) ? pv[j] : 0.f;
          }
        } else if (mode == 2) {   // general mask algebra
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const bool ok = row_ok & (key < len) & mc.valid_ids(qi, key, qi_id, mc.id_of(key));
            pv[j] = ok ? pv[j] : 0.f;
          }
        } else if (mode == 3) {   // targets / window without contextual rows: integer arithmetic, no compare chains
          const int i_eff = row_ok ? qi : -1;
          const int idi = mc.has_targets ? min(i_eff, mc.max_id) : i_eff;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int r = 8 * h8 + j;
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            const int idj = mc.has_targets ? min(key, mc.max_id) : key;
            const int keep = mc.keep_bits_row(i_eff, idi, key, idj) & ((key - len) >> 31);
            pv[j] = __builtin_bit_cast(float, __builtin_bit_cast(int, pv[j]) & keep);
          }
        }
        pb[h8] = E::pack8(pv);
        if constexpr (PRECISE) {
          float lo[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) lo[j] = pv[j] - (float)pb[h8].v[j];
          pbl[h8] = E::pack8(lo);
        }
      }
      HSTU_MARK(12);
#pragma unroll
      for (int d = 0; d < C::DB; ++d) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          Frag a = lds_col_frag<T, C::UPR_V>(Vt, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, lane);
          oacc[d] = E::mma(a, pb[ks], oacc[d]);
          if constexpr (PRECISE) oacc[d] = E::mma(a, pbl[ks], oacc[d]);
        }
      }
    }
    HSTU_MARK(14);
  }
  HSTU_MARK(20);

  // ---- epilogue: O^T accumulators (column n32 = query row, registers = features) -> out rows.  16-bit I/O with
  // the instantiated head dim: through LDS (the K/V ring is dead: each wave writes its [32][DV] tile in the swizzled
  // row-major layout, reads 16-byte units back with 16 consecutive lanes per row and stores whole rows with dwordx4);
  // storing the accumulators directly is DV/4 dwordx2 stores per lane that touch 64 rows each (store-issue bound).
  if constexpr (C::EB == 2 && C::SMEM >= 4 * C::VT) {
    if (p.dv == DV) {     // wave-uniform
      // Where a wave parks its tile.  After the barrier of the LAST step every wave has finished the step before it, so of
      // the ring's three slots only the last tile's is still being read: the two others (a K/V pair = two output tiles
      // each) are dead and each wave has a private tile there -- no workgroup barrier, a wave that is done early (the
      // causal triangle) stores its rows while the others still compute.  Otherwise: barrier, then the ring's start.
      constexpr bool epi_free = C::NS == 3 && C::STAGE >= 2 * C::VT && kFwdThreads / 64 <= 4;
      char* tile;
      if constexpr (epi_free) {
        const int dead = ((ntiles > 0 ? ntiles - 1 : 0) + 1 + (wave >> 1)) % C::NS;   // the slots after the last tile's, cyclically
        tile = smem + dead * C::STAGE + (wave & 1) * C::VT;
      } else {
        __syncthreads();    // every wave is done with the ring
        tile = smem + wave * C::VT;
      }
      // (the 8-byte park writes are 2-way bank-conflicted; flipping the halves on rows with bit 1 set removes the conflicts and costs 1 %:
      // profiles/r03_ab_fwd_epilogue_swap.txt)
      if (wave_active) {
#pragma unroll
        for (int d = 0; d < C::DB; ++d)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            u32x2 v = {E::pk2(oacc[d][4 * rq] * scale_v, oacc[d][4 * rq + 1] * scale_v),
                       E::pk2(oacc[d][4 * rq + 2] * scale_v, oacc[d][4 * rq + 3] * scale_v)};
            *LDS_PTR(u32x2, tile + tile_off<C::UPR_V>(n32, 4 * d + rq) + 8 * hf) = v;
          }
        char* obase = (char*)p.out + ((q_base + r0) * p.o_row_stride + (int64_t)hd * p.o_head_stride) * C::EB;
        const int rows_valid = nq_rows - r0;
#pragma unroll
        for (int i = 0; i < 32 * C::UPR_V / 64; ++i) {
          const int idx = i * 64 + lane;
          const int row = idx / C::UPR_V, unit = idx % C::UPR_V;
          const u32x4 v = *LDS_PTR(const u32x4, tile + tile_off<C::UPR_V>(row, unit));
          if (row < rows_valid && (!(FWD_ABLATE & 1) || p.batch == -12345)) gstore16_nt(obase + (int64_t)row * p.o_row_stride * C::EB + unit * 16, v);
        }
      }
      HSTU_MARK(21);
      if (hi + 1 < n_heads) __syncthreads();   // the output tiles (in the ring's place) are read before the next head's K/V land
      continue;
    }
  }
  if (row_ok) {
    char* orow = (char*)p.out + ((q_base + my_row) * p.o_row_stride + (int64_t)hd * p.o_head_stride) * C::EB;
#pragma unroll
    for (int d = 0; d < C::DB; ++d) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int d0 = 32 * d + 8 * rq + 4 * hf;
        if (d0 < p.dv)
          store4<T>(orow, d0, oacc[d][4 * rq] * scale_v, oacc[d][4 * rq + 1] * scale_v, oacc[d][4 * rq + 2] * scale_v,
                    oacc[d][4 * rq + 3] * scale_v);
      }
    }
  }
  HSTU_MARK(21);
  if (hi + 1 < n_heads) __syncthreads();
  }   // heads
}

}  // namespace hstu
