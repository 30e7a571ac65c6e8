// Second arrangement of the fused LayerNorm + projection kernel (hstu_ln_linear.cuh): TWO workgroups of four waves per CU
// instead of one of eight.  Same arithmetic, same fragments, same store path -- what changes is who waits for whom: a
// workgroup's row prologue (load + normalise 128 rows: ~10 us, four times per workgroup), its barriers and its stalls on
// the store path now pass under the other workgroup's MFMAs instead of idling the CU.  The price: each workgroup streams W
// for its own 128 rows, twice the L2 -> LDS traffic (measured +3 % with every request issued twice in the first
// arrangement), and 80 KiB of LDS per workgroup:
//   ring of 4 HALF tiles (32 columns x 256 k = 16 KiB; a chain of 32 MFMAs crosses two of them)   64 KiB
//   LayerNorm tables and bias in the I/O type (2 + 4 KiB at n = 2048), the waves' store staging 4 x 2 KiB: 78 KiB
// Barrier b_h behind the last MFMA of half tile h: every wave's pieces of half h + 2 have landed (requested two barriers
// ago), and half h is dead, so its slot is requested again for half h + 4.
#pragma once
#include "hstu_ln_linear.cuh"

namespace hstu {

constexpr int kLn2Waves = 4;
constexpr int kLn2Threads = 64 * kLn2Waves;
constexpr int kLn2BlockRows = 32 * kLn2Waves;
constexpr int kLn2HalfBytes = 32 * (kLnlK / 2) * 2;       // 16 KiB
constexpr int kLn2Slots = 4;
constexpr int kLn2MaxN = 2048;
constexpr int kLn2StageBytes = 2048;

static inline int ln2_smem_bytes(int n) { return kLn2Slots * kLn2HalfBytes + 2 * kLnlK * 2 + n * 2 + kLn2Waves * kLn2StageBytes; }

template <int N> HSTU_DEV void ln2_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
HSTU_DEV void ln2_sync(int behind) {      // `behind`: memory instructions of this wave issued after the requests that must have landed
  switch (behind) {
    case 6: ln2_wait_barrier<6>(); break;
    case 5: ln2_wait_barrier<5>(); break;
    case 4: ln2_wait_barrier<4>(); break;
    case 2: ln2_wait_barrier<2>(); break;
    case 1: ln2_wait_barrier<1>(); break;
    default: ln2_wait_barrier<0>(); break;
  }
}

template <typename T>
__global__ __launch_bounds__(kLn2Threads) __attribute__((amdgpu_waves_per_eu(2, 2)))
void hstu_ln_linear_fwd2_kernel(const LnLinearArgs g) {
  extern __shared__ __attribute__((aligned(1024))) char ln2_smem[];
  typedef Elem<T> E;
  typedef typename E::Frag Frag;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  char* ring = ln2_smem;
  T* gam = (T*)(ln2_smem + kLn2Slots * kLn2HalfBytes);
  T* bet = gam + kLnlK;
  T* bia = bet + kLnlK;
  for (int i = tid; i < kLnlK; i += kLn2Threads) {
    gam[i] = ((const T*)g.ln_w)[i];
    bet[i] = ((const T*)g.ln_b)[i];
  }
  for (int i = tid; i < g.n; i += kLn2Threads) bia[i] = g.bias ? ((const T*)g.bias)[i] : (T)0.f;

  // units here are (128-row block, column tile); g.units / g.n_tiles were set for this block size by the launcher
  const int64_t u0 = g.units * blockIdx.x / gridDim.x, u1 = g.units * (blockIdx.x + 1) / gridDim.x;
  const int nsteps = (int)(u1 - u0);
  if (nsteps <= 0) return;
  const int nhalves = 2 * nsteps;
  int64_t blk = u0 / g.n_tiles;
  int tile = (int)(u0 - blk * g.n_tiles);

  // the wave's share of a half tile: 4 of its 16 one-KiB pieces (piece i = rows 2 i, 2 i + 1 of 512 bytes); the lane that
  // fills physical unit u of row r fetches logical unit u ^ swz(r)
  // (recomputed per request -- six VALU instructions -- rather than kept: this arrangement is short of registers)
  auto uo_of = [&](int j) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int r = 2 * (4 * wave + j) + (ln >> 5);
    return (uint32_t)(r * (kLnlK * 2) + (((ln & 31) ^ swz<32>(r)) << 4));
  };
  const uint32_t ring0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ring);
  int it = tile, ihalf = 0, islot = 0, issued = 0;     // request cursor: column tile, k half, ring slot, half tiles requested
  auto req_base = [&]() { return (const char*)g.w + (int64_t)it * kLnlTileBytes + ihalf * (kLnlK / 2) * 2; };
  auto req_dst = [&]() { return ring0 + islot * kLn2HalfBytes + 4 * wave * 1024; };
  auto req_advance = [&]() {
    if (ihalf) it = it + 1 == g.n_tiles ? 0 : it + 1;
    ihalf ^= 1;
    islot = (islot + 1) & (kLn2Slots - 1);
    ++issued;
  };
  for (int i = 0; i < 3 && i < nhalves; ++i) {
    const char* b = req_base();
    const uint32_t d = req_dst();
#pragma unroll
    for (int j = 0; j < 4; ++j) lnl_dma16(uo_of(j), b, d + j * 1024);
    req_advance();
  }
  __syncthreads();   // tables

  u32x4 xf[kLnlKS];
  f32x16 acc[2];
  Frag wf[LNL_AHEAD];
  int cslot = 0;     // ring slot of the first half of the current tile
  char* stage = (char*)(bia + g.n) + wave * kLn2StageBytes;
  // fragment kk of a half tile = unit 2 kk + h of row m, at slot (2 kk + h) ^ swz(m): row base and swizzle are kept, the slot is
  // formed per read (xor, shift-add) -- 2 registers instead of 8 addresses
  const uint32_t frow = ring0 + (uint32_t)(m * (kLnlK / 2) * 2);
  const uint32_t fsw = (uint32_t)(swz<32>(m) ^ h);
  // fragment ks (0..31) of the tile whose first half sits in slot `s0`
  auto frag_at = [&](int s0, int ks) {
    Frag f;
    const int sl = (s0 + (ks >> 4)) & (kLn2Slots - 1);
    const u32x4 x = *LDS_PTR(const u32x4, (uintptr_t)(frow + (((uint32_t)(2 * (ks & 15)) ^ fsw) << 4) + sl * kLn2HalfBytes));
    f.v = __builtin_bit_cast(typename E::vec8, x);
    return f;
  };
  auto bias_into = [&](f32x16& a, int t) {
    const T* bt = bia + t * 32 + 4 * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      typedef T t4 __attribute__((ext_vector_type(4)));
      const t4 b4 = *LDS_PTR(const t4, bt + 8 * j);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[4 * j + e] = (float)b4[e];
    }
  };

  LnlPacked pk;
  char* pk_dst = nullptr;
  bool ok_lo = false, ok_hi = false;
  int n_stores = 0;
  auto step = [&](f32x16& cur, f32x16& oth, bool have_prev, bool have_packed, char* row_dst) {
    const int nslot = (cslot + 2) & (kLn2Slots - 1);
    int n_req = 0;
#pragma unroll
    for (int ks = 0; ks < kLnlKS; ++ks) {
      Frag xb;
      xb.v = __builtin_bit_cast(typename E::vec8, xf[ks]);
      cur = E::mma(wf[ks % LNL_AHEAD], xb, cur);
      if (ks == 3 && have_packed && ok_lo) lnl_gstore(pk_dst, pk.lo);
      if (ks == 9 && have_packed && ok_hi) lnl_gstore(pk_dst + 16 * g.ldy * 2, pk.hi);
      if (ks % 4 == 0) {                 // one request of the half tile due after the last barrier: 4 in each half of the chain
        if (ks % 16 == 0) n_req = 0;
        if (issued < nhalves) {
          lnl_dma16(uo_of((ks % 16) / 4), req_base(), req_dst() + ((ks % 16) / 4) * 1024);
          ++n_req;
          if (ks % 16 == 12) req_advance();
        }
      }
      if (ks == kLnlKS / 2 - 1) ln2_sync(n_req + (have_packed ? n_stores : 0));
      if (ks == kLnlKS - 1) ln2_sync(n_req);
      wf[ks % LNL_AHEAD] = ks + LNL_AHEAD < kLnlKS ? frag_at(cslot, ks + LNL_AHEAD) : frag_at(nslot, ks + LNL_AHEAD - kLnlKS);
      if (ks == kLnlKS / 2 + 1 && have_prev) {
        pk = lnl_pack_tile<T>(oth, stage, lane);
        pk_dst = row_dst + (tile - 1) * 64;
      }
      if (ks == kLnlKS / 2 + 6) bias_into(oth, tile + 1 == g.n_tiles ? 0 : tile + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    cslot = nslot;
    ++tile;
  };
  auto store_packed = [&]() {
    if (ok_lo) lnl_gstore(pk_dst, pk.lo);
    if (ok_hi) lnl_gstore(pk_dst + 16 * g.ldy * 2, pk.hi);
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();      // the first three half tiles are in the ring
  int left = nsteps;
  while (left > 0) {
    const int64_t row0 = blk * kLn2BlockRows + wave * 32;
    lnl_load_rows<T, false, true>(g, row0, (const float*)gam, (const float*)bet, lane, xf, false, stage, false);
    const int64_t row = row0 + (lane >> 2);
    ok_lo = row < g.rows;
    ok_hi = row + 16 < g.rows;
    char* row_dst = (char*)g.y + row * g.ldy * 2 + 16 * (lane & 3);
    int nt = g.n_tiles - tile;
    if (nt > left) nt = left;
    left -= nt;
    bias_into(acc[0], tile);
#pragma unroll
    for (int i = 0; i < LNL_AHEAD; ++i) wf[i] = frag_at(cslot, i);
    n_stores = (__builtin_amdgcn_ballot_w64(ok_lo) != 0) + (__builtin_amdgcn_ballot_w64(ok_hi) != 0);
    step(acc[0], acc[1], false, false, row_dst);
    int k = 1;
    for (; k + 1 < nt; k += 2) {
      step(acc[1], acc[0], true, k > 1, row_dst);
      step(acc[0], acc[1], true, true, row_dst);
    }
    if (k < nt) {
      step(acc[1], acc[0], true, k > 1, row_dst);
      ++k;
    }
    if (nt > 1) store_packed();
    pk = (k & 1) ? lnl_pack_tile<T>(acc[0], stage, lane) : lnl_pack_tile<T>(acc[1], stage, lane);
    pk_dst = row_dst + (tile - 1) * 64;
    store_packed();
    if (tile == g.n_tiles) { tile = 0; ++blk; }
  }
}

template <typename T>
static int launch_ln_linear2(LnLinearArgs g, hipStream_t st) {
  const int smem = ln2_smem_bytes(g.n);
  g.units = ((g.rows + kLn2BlockRows - 1) / kLn2BlockRows) * g.n_tiles;
  auto kern = hstu_ln_linear_fwd2_kernel<T>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "ln_linear_fwd: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
  static const int n_cu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int grid = (int)(g.units < 2 * n_cu ? g.units : 2 * n_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kLn2Threads), smem, st, g);
  return check_launch("ln_linear_fwd(split)");
}

}  // namespace hstu
