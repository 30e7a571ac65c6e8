// Shared device-side building blocks for the gfx950 HSTU kernels.
//
// Conventions used by every attention kernel in this directory
// ------------------------------------------------------------
// * One wavefront = 64 lanes.  lane = threadIdx.x & 63, `n32 = lane & 31`,
//   `hf = lane >> 5` (which half of the wave), `g16 = lane >> 4`, `i16 = lane & 15`.
// * The matrix instruction is the 32x32 MFMA (v_mfma_f32_32x32x16_{bf16,f16}; for
//   fp32 I/O eight v_mfma_f32_32x32x2_f32).  A "fragment" is 8 elements per lane:
//       A operand:  A[row = n32][k = kk(hf, j)]      j = 0..7
//       B operand:  B[k = kk(hf, j)][col = n32]
//       C/D      :  C[row = (r&3) + 8*(r>>2) + 4*hf][col = n32]   r = 0..15
//   The hardware pairs A's (hf, j) with B's (hf, j); which logical contraction
//   index that slot carries is OUR choice as long as both operands agree.
// * Tiles in LDS are row-major [32 rows][D] with the 16-byte units of each row
//   XOR-swizzled (tile_off) so that a 16-lane group reading one unit column of 16
//   different rows hits 16 different 16-byte bank slots.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hstu_hip.h"

namespace hstu {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define HSTU_DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLOBAL_PTR(T, p) ((__attribute__((address_space(1))) T*)(uintptr_t)(p))   // (see gload16 / gstore16)

// ---------------------------------------------------------------------------
// element-type traits
// ---------------------------------------------------------------------------
template <typename T> struct Elem;

template <> struct Elem<bf16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kDtype = HSTU_DTYPE_BF16;
  typedef bf16_t vec8 __attribute__((ext_vector_type(8)));
  struct Frag { vec8 v; };
  static HSTU_DEV f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
  }
  // 16x16x32: A[m = lane&15][k = 8 (lane>>4) + j], B[k][n = lane&15], C[m = 4 (lane>>4) + r][n = lane&15]
  static HSTU_DEV f32x4 mma16(const Frag& a, const Frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
  }
  static HSTU_DEV uint32_t pk2(float a, float b) {   // one v_cvt_pk_bf16_f32
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef bf16_t h2 __attribute__((ext_vector_type(2)));
    f2 x = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, h2));
  }
  static HSTU_DEV Frag pack8(const float* x) {
    u32x4 w = {pk2(x[0], x[1]), pk2(x[2], x[3]), pk2(x[4], x[5]), pk2(x[6], x[7])};
    Frag f;
    f.v = __builtin_bit_cast(vec8, w);
    return f;
  }
};

template <> struct Elem<f16_t> {
  static constexpr int kBytes = 2;
  static constexpr int kDtype = HSTU_DTYPE_F16;
  typedef f16_t vec8 __attribute__((ext_vector_type(8)));
  struct Frag { vec8 v; };
  static HSTU_DEV f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, c, 0, 0, 0);
  }
  static HSTU_DEV f32x4 mma16(const Frag& a, const Frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
  }
  static HSTU_DEV uint32_t pk2(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef f16_t h2 __attribute__((ext_vector_type(2)));
    f2 x = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, h2));
  }
  static HSTU_DEV Frag pack8(const float* x) {
    u32x4 w = {pk2(x[0], x[1]), pk2(x[2], x[3]), pk2(x[4], x[5]), pk2(x[6], x[7])};
    Frag f;
    f.v = __builtin_bit_cast(vec8, w);
    return f;
  }
};

template <> struct Elem<float> {
  static constexpr int kBytes = 4;
  static constexpr int kDtype = HSTU_DTYPE_F32;
  typedef float vec8 __attribute__((ext_vector_type(8)));
  struct Frag { vec8 v; };
  // exact fp32: slot (hf, j) of the 16-wide k-group is fed to the j-th 32x32x2 MFMA.
  static HSTU_DEV f32x16 mma(const Frag& a, const Frag& b, f32x16 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], c, 0, 0, 0);
    return c;
  }
  static HSTU_DEV Frag pack8(const float* x) {
    Frag f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = x[j];
    return f;
  }
};

template <typename T> HSTU_DEV float to_f32(T x) { return (float)x; }

// ---------------------------------------------------------------------------
// LDS tile layout
// ---------------------------------------------------------------------------
// Units are 16 bytes.  UPR = units per row (power of two).  tile_off returns the byte offset of
// unit `u` of row `r` inside a tile of 32 (or more) rows: unit u sits in slot u ^ swz(r) of its row.
// swz is chosen for BOTH access patterns of the kernels (LDS = 64 banks x 4 B = a 256-byte window):
//  * ds_read_b128 of one unit column from 16 different rows (16-lane groups whose rows have 16
//    distinct values of r & 15): swz must be a bijection of r & 15 onto the slots sharing a window;
//  * ds_read_b64_tr_b16 (32-lane groups = 4 consecutive rows 4m..4m+3 x one 64-byte, 4-unit-aligned
//    column block): the four rows must land in four DIFFERENT 64-byte quarters of the window, i.e.
//    r & 3 must drive the quarter.  (A plain u ^ (r & 15) puts all four rows in the same quarter:
//    a 4-way conflict on every transposed read.)
template <int UPR> HSTU_DEV int swz(int r) {
  static_assert((UPR & (UPR - 1)) == 0, "UPR must be a power of two");
  if constexpr (UPR >= 16) return ((r & 3) << 2) | ((r >> 2) & 3);             // 256-byte rows (or longer)
  else if constexpr (UPR == 8) return (((r >> 1) & 1) << 2) | ((r >> 2) & 3);  // 2 rows per window: quarter = (r&1, r>>1&1)
  else return (r / (16 / UPR)) & (UPR - 1);                                     // >= 4 rows per window: quarters differ already
}
template <int UPR> HSTU_DEV int tile_off(int r, int u) { return (r * UPR + (u ^ swz<UPR>(r))) << 4; }

// Fragment of a row-major tile for a contraction along the row:
// elements [e0, e0+8) of row `row`.  16-bit: one unit; fp32: two units.
template <typename T, int UPR>
HSTU_DEV typename Elem<T>::Frag lds_row_frag(const char* tile, int row, int e0) {
  typename Elem<T>::Frag f;
  if constexpr (Elem<T>::kBytes == 2) {
    u32x4 x = *LDS_PTR(const u32x4, tile + tile_off<UPR>(row, e0 >> 3));
    f.v = __builtin_bit_cast(typename Elem<T>::vec8, x);
  } else {
    u32x4 x0 = *LDS_PTR(const u32x4, tile + tile_off<UPR>(row, (e0 >> 2)));
    u32x4 x1 = *LDS_PTR(const u32x4, tile + tile_off<UPR>(row, (e0 >> 2) + 1));
    f32x4 a = __builtin_bit_cast(f32x4, x0), b = __builtin_bit_cast(f32x4, x1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f.v[j] = a[j]; f.v[4 + j] = b[j]; }
  }
  return f;
}

// Fragment for a contraction ACROSS rows ("transposed" use of a row-major tile):
// slot j<4 -> tile[rowA + j][col], slot j>=4 -> tile[rowB + j-4][col], where
// col = colblk + n32.  16-bit types use the hardware transpose read
// (ds_read_b64_tr_b16: the 16 lanes of a group fetch a [4 rows][16 cols] block,
// lane i supplying the address of row i/4, cols 4*(i%4)..+3, and receive column i
// of the 4 rows); fp32 gathers with eight ds_read_b32.
template <typename T, int UPR>
HSTU_DEV typename Elem<T>::Frag lds_col_frag(const char* tile, int rowA, int rowB, int colblk, int lane) {
  typename Elem<T>::Frag f;
  if constexpr (Elem<T>::kBytes == 2) {
    const int i16 = lane & 15;
    const int col = colblk + (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2);  // first of this lane's 4 cols
    const int sub = (col & 7) << 1;                                         // byte offset inside the unit
    const int ra = rowA + (i16 >> 2), rb = rowB + (i16 >> 2);
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off<UPR>(ra, col >> 3) + sub));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off<UPR>(rb, col >> 3) + sub));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    f.v = __builtin_bit_cast(typename Elem<T>::vec8, ab);
  } else {
    const int col = colblk + (lane & 31);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f.v[j] = *LDS_PTR(const float, tile + tile_off<UPR>(rowA + j, col >> 2) + ((col & 3) << 2));
      f.v[4 + j] = *LDS_PTR(const float, tile + tile_off<UPR>(rowB + j, col >> 2) + ((col & 3) << 2));
    }
  }
  return f;
}

// ---------------------------------------------------------------------------
// index loads (int32 | int64 offsets / targets)
// ---------------------------------------------------------------------------
HSTU_DEV int64_t load_index(const void* p, int64_t i, int is64) {
  return is64 ? ((const int64_t*)p)[i] : (int64_t)((const int32_t*)p)[i];
}
// The same through the CONSTANT address space: with a wave-uniform index hipcc emits a scalar load (s_load, counted by lgkmcnt) instead
// of a vector load + v_readfirstlane + s_waitcnt vmcnt(0).  The difference matters in the persistent kernels: a vmcnt(0) at the top of
// a problem waits for every LDS-DMA request and every store the previous problem's tail left in flight.  Legal for arrays the kernel
// only reads (offsets, targets, launch order): the scalar cache is not coherent with this kernel's own vector stores.
HSTU_DEV int64_t sload_index(const void* p, int64_t i, int is64) {
  typedef const __attribute__((address_space(4))) int64_t* c64;
  typedef const __attribute__((address_space(4))) int32_t* c32;
  return is64 ? ((c64)(uintptr_t)p)[i] : (int64_t)((c32)(uintptr_t)p)[i];
}

// ---------------------------------------------------------------------------
// mask algebra (reference: ops/pytorch/pt_hstu_attention.py:32-84)
// ---------------------------------------------------------------------------
struct MaskCtx {
  int len;        // L of this user
  int max_id;     // L (c==0) or L-c+1, minus num_targets if present
  int ctx;        // contextual_seq_len
  int win;        // max_attn_len (0 = none)
  int full;       // min_full_attn_seq_len
  int has_targets;
  int simple;     // plain causal: no targets, no window, no contextual rows -> valid = (j <= i)

  HSTU_DEV int id_of(int pos) const {
    int id = ctx > 0 ? max(pos - ctx + 1, 0) : pos;
    return has_targets ? min(id, max_id) : id;
  }
  // row position i (query), col position j (key); both < len
  HSTU_DEV bool valid(int i, int j) const { return valid_ids(i, j, id_of(i), id_of(j)); }
  HSTU_DEV bool valid_ids(int i, int j, int idi, int idj) const {
    if (simple) return j <= i;
    const int d = idi - idj;
    bool m = (i == j) | (d > 0);
    if (win > 0) m = m & ((d <= win) | ((full > 0) & (idi >= max_id - full)));
    if (ctx > 0) m = m | ((idi == 0) & (idj < max_id));
    return m;
  }
  // Same predicate as valid() & (i < len), as an all-ones / zero integer, for ctx == 0 (the folded backward):
  // integer arithmetic only.  Compares would go through SGPR pairs and scalar and/or chains -- a VALU -> SALU ->
  // VALU round trip per element.  `j` must be < len (the caller ANDs its own lane-constant key mask).
  HSTU_DEV int keep_bits_noctx(int i, int j, int idj) const {
    const int i_eff = i | ((len - 1 - i) >> 31);               // i, or -1 when i >= len
    const int idi = has_targets ? min(i_eff, max_id) : i_eff;
    return keep_bits_row(i_eff, idi, j, idj);
  }
  // the same with the row side (i_eff, idi as above) precomputed: the forward's lanes own one query row each
  HSTU_DEV int keep_bits_row(int i_eff, int idi, int j, int idj) const {
    const int d = idi - idj;
    const int x = i_eff ^ j;
    int m = (~(x | -x) >> 31) | ((-d) >> 31);                  // (i == j) | (d > 0)
    if (win > 0) {                                             // wave-uniform
      int in_win = (d - win - 1) >> 31;                        // d <= win
      if (full > 0) in_win |= (max_id - full - 1 - idi) >> 31; // idi >= max_id - full
      m &= in_win;
    }
    return m;
  }
  // Is EVERY (i, j) with i in [i0, i0+ni), j in [j0, j0+nj), i,j < len valid?  (sufficient
  // condition; such tiles skip the per-element mask.  Query rows >= len need no mask: their
  // q / dO rows are zero-filled by the register staging path, so silu(0) = 0 and 0 * x = 0.)
  HSTU_DEV bool pair_fully_valid(int i0, int ni, int j0, int nj) const {
    if (i0 >= len || j0 + nj > len) return false;   // key rows past len may hold garbage (LDS-DMA): mask them
    const int i1 = min(i0 + ni, len) - 1;
    const int j1 = j0 + nj - 1;
    if (id_of(i0) - id_of(j1) < 1) return false;          // every pair strictly causal
    if (win > 0 && id_of(i1) - id_of(j0) > win) return false;  // every pair inside the window
    return true;
  }
  // Conservative test: can ANY (i, j) with i in [i0, i0+ni), j in [j0, j0+nj) be valid?
  HSTU_DEV bool pair_may_be_active(int i0, int ni, int j0, int nj) const {
    if (i0 >= len || j0 >= len) return false;
    const int i1 = min(i0 + ni, len) - 1;      // last row
    const int j1 = min(j0 + nj, len) - 1;      // last col
    const bool ctx_rows = (ctx > 0) && (i0 < ctx);  // rows with id 0 see every non-target col
    if (ctx_rows) return true;
    if (i1 < j0) return false;                  // strictly above the diagonal
    if (win > 0 && full == 0) {
      // smallest distance in the block: first row vs last col
      if (id_of(i0) - id_of(j1) > win && !(i0 <= j1)) return false;
    }
    return true;
  }
};

#ifndef HSTU_TARGETS_PLAIN
#define HSTU_TARGETS_PLAIN 1   // 0: query tiles in front of the first target take the general mask path too (the comparator of tools/ab_bwd.py --targets)
#endif
// SCALAR: the user index is wave-uniform and num_targets read-only -> sload_index
template <bool SCALAR = false>
HSTU_DEV MaskCtx make_mask_ctx(const HstuAttnParams& p, int b, int len) {
  MaskCtx m;
  m.len = len;
  m.ctx = p.contextual_seq_len;
  m.win = p.max_attn_len;
  m.full = p.min_full_attn_seq_len;
  m.has_targets = p.num_targets != nullptr;
  int max_id = len;
  if (m.ctx > 0) max_id = max_id - m.ctx + 1;
  if (m.has_targets) max_id -= (int)(SCALAR ? sload_index(p.num_targets, b, p.targets_dtype) : load_index(p.num_targets, b, p.targets_dtype));
  m.max_id = max_id;
  m.simple = (!m.has_targets) && m.win == 0 && m.ctx == 0;
  return m;
}

// ---------------------------------------------------------------------------
// research-path additive bias (research/modeling/sequential/hstu.py:87-144):
//   bias[i,j] = pos_w[(N-1) + j - i] + ts_w[bucket(ts[i+1] - ts[j])],  ts[N] := ts[N-1]
//   bucket(d) = clamp((int)(log(max(|d|,1)) / bucket_div), 0, num_buckets)     (fp32 log and divide)
// ---------------------------------------------------------------------------
// The two weight tables and the user's timestamp row are staged in LDS once per workgroup (a few KB): every
// (query, key) element needs one timestamp and two table entries, and as global loads those three gathers per
// element -- L1 hits, but dependent ones -- made the bias kernels 3x (forward) / 6x (backward) slower than the plain
// ones.
struct BiasCtx {
  const char* lpos;    // LDS: pos_w, 2n-1 floats
  const char* lts;     // LDS: ts_w, nb+1 floats, or nullptr (position-only bias)
  const char* ltime;   // LDS: this user's n timestamps (int64), or nullptr
  const char* lt32;    // LDS: the same as int32 offsets from the row's first timestamp, then one "out of range" word per wave
  int n, nb, npad;     // npad = n + 32 rounded up to a multiple of 4: length of each int32 array
  bool small;          // every offset fits 30 bits: time differences are formed and converted in 32-bit arithmetic
  float div, kf;       // kf = ln 2 / div: bucket coordinate = log2(d) * kf
  HSTU_DEV int64_t ts_at(int pos) const {
    return ltime ? *LDS_PTR(const int64_t, ltime + 8 * min(max(pos, 0), n - 1)) : 0;
  }
  // positions 0 .. n + 31 are readable: entries >= n repeat the last timestamp (ts[N] := ts[N-1], and key / query
  // positions of a partial tile past the sequence end, whose elements are masked anyway)
  HSTU_DEV int t32_at(int pos) const { return *LDS_PTR(const int, lt32 + 4 * pos); }
  // four consecutive entries in one 16-byte read: offsets of positions pos .. pos+3 (pos a multiple of 4), and of the
  // positions FOLLOWING them (the query side uses the next item's timestamp) from a copy shifted by one
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  HSTU_DEV i32x4 t32x4_at(int pos) const { return *LDS_PTR(const i32x4, lt32 + 4 * pos); }
  HSTU_DEV i32x4 t32x4_next(int pos) const { return *LDS_PTR(const i32x4, lt32 + 4 * (npad + pos)); }
  // after the barrier that follows stage_bias_tables: did any wave see an offset outside 30 bits?
  HSTU_DEV void finish(int nwaves) {
    small = ltime != nullptr;
    for (int w = 0; w < nwaves; ++w) small = small && (*LDS_PTR(const int, lt32 + 4 * (2 * npad + w)) == 0);
  }
  // both positions are below n, so n - 1 + key - qi is a valid table index as it is
  HSTU_DEV int pos_index(int qi, int key) const { return n - 1 + key - qi; }
  // bucket(d) = clamp((int)(logf((float)max(|d|, 1)) / div), 0, nb), exactly: the coordinate is first formed with the
  // hardware log2 (one quarter-rate instruction instead of the ~40 of logf + an IEEE divide).  Its error is below
  // 2e-5 (1 ulp of log2(x) <= 27 is 2e-6, times kf ~ 2.3, plus the rounding of the product), and the accurate
  // expression's own is below 1e-5, so whenever the fast coordinate lies farther than 1e-4 from an integer its floor
  // IS the floor of the accurate expression; the ~2 in 10^4 elements closer than that (and d = 1, coordinate 0) take
  // the accurate expression itself.
  HSTU_DEV int bucket(int64_t t_q1, int64_t t_k) const {
    int64_t d = t_q1 - t_k;
    d = d < 0 ? -d : d;
    d = d < 1 ? 1 : d;
    return bucket_of((float)d);
  }
  // the same from 32-bit offsets (|difference| < 2^31; the int -> float conversion rounds like the int64 one)
  HSTU_DEV int bucket32(int t_q1, int t_k) const {
    int d = t_q1 - t_k;
    d = max(d, -d);
    d = max(d, 1);
    return bucket_of((float)d);
  }
  HSTU_DEV int bucket_of(float x) const {
    const float c = __builtin_amdgcn_logf(x) * kf;
    int bk = (int)c;
    const float fr = c - (float)bk;
    if (fr < 1e-4f || fr > 0.9999f) bk = (int)(logf(x) / div);
    return min(bk, nb);   // x >= 1: never negative
  }
  HSTU_DEV float value(int pidx, int bkt) const {
    return *LDS_PTR(const float, lpos + 4 * pidx) + (lts ? *LDS_PTR(const float, lts + 4 * bkt) : 0.f);
  }
};

// Time-bucket histogram of dS' (research-path backward).  A lane owns ONE key and walks its query rows in order, so the
// time difference -- and with it the (logarithmic) bucket -- changes only a handful of times per tile: the running sum of
// the current bucket stays in a register and goes to the LDS histogram when the bucket changes.  The state is the LDS
// byte offset of the current bucket's word, so one element costs a shift, two compares, one exec-masked ds_add, an add, a
// select and a move -- no multiply, no nested branches (22 -> 10 instructions).
struct TsRun {
  unsigned cur;     // byte offset of the running bucket's word inside this lane's histogram copy: bucket << shift
  float sum;
  unsigned base;    // wave-uniform: LDS byte address of the histogram
  int shift;        // wave-uniform: log2(4 * copies)
  HSTU_DEV void init(const float* hts_lds, int copies) {
    base = (unsigned)(uintptr_t)LDS_PTR(const float, hts_lds);
    shift = 2 + (31 - __builtin_clz((unsigned)copies));
    cur = 0;
    sum = 0.f;
  }
  // address of the word: the lane's copy (lane & (copies - 1)) is worked out here, on the rare path, not kept in a register
  HSTU_DEV __attribute__((address_space(3))) float* word() const {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    return (__attribute__((address_space(3))) float*)(uintptr_t)(base + cur + ((lane << 2) & ((1u << shift) - 1u)));
  }
  HSTU_DEV void flush() {
    if (sum != 0.f) __hip_atomic_fetch_add(word(), sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sum = 0.f;
  }
  HSTU_DEV void add(int bkt, float v) {
    const unsigned a = (unsigned)bkt << shift;
    const bool chg = a != cur;
    if (chg & (sum != 0.f)) __hip_atomic_fetch_add(word(), sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    sum = chg ? v : sum + v;
    cur = a;
  }
};

HSTU_DEV const int64_t* bias_ts_row(const HstuAttnParams& p, int b) {
  return (p.ts_w && p.timestamps) ? p.timestamps + (int64_t)b * p.ts_row_stride : nullptr;
}

// cooperative copy of the tables into `lds` (bias_table_bytes); the caller puts a barrier before the first use.
// `user_only`: the position / time tables are already there from an earlier call of this workgroup with the same `lds`
// (they do not depend on the user): only the user's timestamps are staged.
// the context of tables laid out at `lds` (the layout stage_bias_tables fills)
HSTU_DEV BiasCtx bias_ctx_at(const HstuAttnParams& p, bool has_ts, char* lds) {
  BiasCtx c;
  const int n = p.max_seq_len;
  char* lpos = lds + 128;
  char* lts = lpos + (2 * n * 4 + 15) / 16 * 16;
  char* ltime = lts + ((p.num_buckets + 1) * 4 + 15) / 16 * 16;
  char* lt32 = ltime + (8 * n + 15) / 16 * 16;
  c.lt32 = lt32;
  c.npad = (n + 32 + 3) / 4 * 4;
  c.small = false;
  c.lpos = lpos;
  c.lts = has_ts ? lts : nullptr;
  c.ltime = has_ts ? ltime : nullptr;
  c.n = n;
  c.nb = p.num_buckets;
  c.div = p.bucket_div;
  c.kf = 0.69314718055994530942f / p.bucket_div;
  return c;
}

HSTU_DEV BiasCtx stage_bias_tables(const HstuAttnParams& p, int b, char* lds, int tid, int nthreads, bool user_only = false) {
  const int n = p.max_seq_len;
  // The position index n - 1 + key - query is formed for EVERY element of a tile, also for the query rows of the last
  // tile that lie past max_seq_len (masked, but their value still passes through silu and a 0 x value product): down to
  // -31.  Those reads must see finite numbers whatever the previous kernel left in LDS (a NaN there survives 0 x NaN):
  // 32 zeros in front of the table, zeros behind its 2n - 1 entries.
  const int64_t* ts_row = bias_ts_row(p, b);
  const BiasCtx c = bias_ctx_at(p, ts_row != nullptr, lds);
  char* lpos = lds + 128;
  char* lts = lpos + (2 * n * 4 + 15) / 16 * 16;
  char* ltime = lts + ((p.num_buckets + 1) * 4 + 15) / 16 * 16;
  char* lt32 = ltime + (8 * n + 15) / 16 * 16;
  if (!user_only) {
    for (int i = tid; i < 32; i += nthreads) *LDS_PTR(float, lds + 4 * i) = 0.f;
    for (int i = tid; i < 2 * n - 1; i += nthreads) *LDS_PTR(float, lpos + 4 * i) = GLOBAL_PTR(const float, p.pos_w)[i];
    for (int i = 2 * n - 1 + tid; i < (int)(lts - lpos) / 4; i += nthreads) *LDS_PTR(float, lpos + 4 * i) = 0.f;
  }
  if (ts_row) {
    if (!user_only)
      for (int i = tid; i <= p.num_buckets; i += nthreads) *LDS_PTR(float, lts + 4 * i) = GLOBAL_PTR(const float, p.ts_w)[i];
    const auto* const ts_row_g = GLOBAL_PTR(const int64_t, ts_row);
    const int64_t t0 = ts_row_g[0];
    bool big = false;
    for (int i = tid; i < n; i += nthreads) {
      const int64_t t = ts_row_g[i], o = t - t0;
      *LDS_PTR(int64_t, ltime + 8 * i) = t;
      *LDS_PTR(int, lt32 + 4 * i) = (int)o;
      big = big || o >= (1LL << 30) || o <= -(1LL << 30);
    }
    const int npad = (n + 32 + 3) / 4 * 4;
    const int last = (int)(ts_row_g[n - 1] - t0);
    for (int i = n + tid; i < npad; i += nthreads) *LDS_PTR(int, lt32 + 4 * i) = last;          // padding: see t32_at
    for (int i = tid; i < npad; i += nthreads)                                                  // shifted copy: entry i = t[i+1]
      *LDS_PTR(int, lt32 + 4 * (npad + i)) = i + 1 < n ? (int)(ts_row_g[i + 1] - t0) : last;
    const bool wave_big = __builtin_amdgcn_ballot_w64(big) != 0;
    if ((tid & 63) == 0) *LDS_PTR(int, lt32 + 4 * (2 * npad + (tid >> 6))) = wave_big ? 1 : 0;
  } else if (!user_only) {
    // position-only bias with n < 32: indices up to n + 30 run past the position table into this slot
    for (int i = tid; i < (int)(ltime - lts) / 4; i += nthreads) *LDS_PTR(float, lts + 4 * i) = 0.f;
  }
  return c;
}

// user of workgroup slot `slot` (HstuAttnParams::user_order: heavy-first launch order for long-tailed batches)
HSTU_DEV int user_of_slot(const HstuAttnParams& p, int slot) { return p.user_order ? p.user_order[slot] : slot; }
HSTU_DEV int user_of_slot_s(const HstuAttnParams& p, int slot) {      // (wave-uniform slot: scalar load, see sload_index)
  return p.user_order ? (int)sload_index(p.user_order, slot, 0) : slot;
}

// 1/N, or the caller's device-side replacement for it (HstuAttnParams::attn_scale: the reference's attn_scale[0])
HSTU_DEV float attn_scale_of(const HstuAttnParams& p) { return p.attn_scale ? *p.attn_scale : p.scale; }

// silu(s) = s * sigmoid(s), fp32, hardware exp2 / rcp (1 ulp each)
HSTU_DEV float fast_sigmoid(float s) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * s));
}

// Optional cycle-counter trace (debug builds with -DHSTU_TRACE only; see tools/trace_build.sh):
// lane 0 of every wave of ONE workgroup appends s_memtime stamps to a global buffer.
#ifdef HSTU_TRACE
static __device__ unsigned long long* g_hstu_trace_fwd = nullptr;   // set by hstu_trace_set_fwd (trace builds only)
struct TraceCtx { unsigned long long* p; bool on; int i; };
#define HSTU_TRACE_DECL(ptr, on_) TraceCtx _tc{(unsigned long long*)(ptr), (on_), 0}
#define HSTU_TRACE_ARG , TraceCtx& _tc
#define HSTU_TRACE_PASS , _tc
#define HSTU_MARK(tag)                                                                       \
  do {                                                                                       \
    if (_tc.on && (threadIdx.x & 63) == 0 && _tc.i < 126) {                                  \
      _tc.p[(threadIdx.x >> 6) * 256 + 2 * _tc.i] = (unsigned long long)(tag);               \
      _tc.p[(threadIdx.x >> 6) * 256 + 2 * _tc.i + 1] = __builtin_readcyclecounter();        \
      ++_tc.i;                                                                               \
    }                                                                                        \
  } while (0)
#else
#define HSTU_TRACE_DECL(ptr, on)
#define HSTU_TRACE_ARG
#define HSTU_TRACE_PASS
#define HSTU_MARK(tag)
#endif

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains
// vmcnt, i.e. waits for every outstanding global store / atomic of the wave to be acknowledged -- in a loop whose
// iterations end with atomics (the multi-key-block backward) that is a full L2 round trip per step.
HSTU_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-byte global load / store helpers.  The pointers are cast to the GLOBAL address space: a pointer hipcc cannot trace back to a
// kernel argument (one re-read from the kernel-argument segment inside a persistent loop, see hstu_attn_bwd_fold.cuh) would otherwise
// be accessed with flat_* instructions, which also occupy the LDS counter.
HSTU_DEV u32x4 gload16(const void* p) { return *GLOBAL_PTR(const u32x4, p); }
HSTU_DEV void gstore16(void* p, u32x4 v) { *GLOBAL_PTR(u32x4, p) = v; }
// streaming variant (written once, not read again by this kernel).  Measured: forward 1.40 -> 1.38 ms; the folded
// backward gets SLOWER with it (3.08 -> 3.15 ms), so only the forward's output rows use it.
HSTU_DEV void gstore16_nt(void* p, u32x4 v) { __builtin_nontemporal_store(v, GLOBAL_PTR(u32x4, p)); }

}  // namespace hstu
