// Projection GEMM for the STU layer (MFMA-bound):
//   C (M x N) = A (M x K) . B (K x N)  [+ bias (N)]  [SiLU on columns < silu_cols]  [+ residual (M x N)]
// 16-bit I/O, fp32 accumulation.  M = all jagged rows of the batch (10^5..10^6), K and N a few hundred to a few
// thousand: `hstu_compute_uqvk` (ops/hstu_compute.py:62-89: addmm + the SiLU of the u slice) and the output projection
// of `hstu_compute_output` (ops/pytorch/pt_hstu_linear.py:85-99: x + y_cat @ W_o).
//
// One workgroup (8 waves) per 256 x 256 tile of C; a wave owns 128 rows x 64 columns = 8 accumulators of 32 x 32.
// K is walked in steps of 32: A and B panels go HBM -> LDS by LDS-DMA (inline asm: hipcc would drain vmcnt before every
// LDS read after a builtin DMA) into a ring of 4 stages, so the panels of the next 3 steps are in flight while a step is
// multiplied (counted vmcnt waits); one barrier per step.  LDS tiles are row-major with XOR-swizzled 16-byte units (hstu_common.cuh): A rows are read as 16-byte row
// fragments, B is read with the hardware transpose (ds_read_b64_tr_b16).  The product is formed TRANSPOSED
// (C^T = B^T A^T: B supplies the MFMA's A operand) so that a lane ends up with 4 consecutive columns of one row of C:
// the tile is parked in the (dead) stages as bf16 rows and leaves as whole 128-byte lines -- scattered 8-byte stores
// are store-issue bound on this part.  Bias / SiLU are applied to the fp32 accumulators, the residual is added while
// the rows are copied out.  Workgroup ids walk the N tiles of one M panel first and are dealt to the XCDs in runs, so
// the panel of A is fetched from HBM once and re-read from one L2.
//
// EXPERIMENT, not part of libhstu_hip.so (tools/gemm_exp/build.sh builds libgemm_exp.so next to this file;
// tools/bench_gemm.py measures it).  Round-1 result at the layer shape (194,560 rows, bf16): correct (bf16-rounding
// error only), uvqk forward 610 TFLOP/s with bias + SiLU fused (hipBLASLt: 740 for the bare GEMM), output forward
// 780 with the residual fused (hipBLASLt 1080 bare).  Not MFMA- or LDS-bound: with the multiplication removed the
// loads + epilogue alone take 72 % of the time, and neither a deeper ring (3 / 4 / 5 stages) nor 64- vs 32-wide K
// steps move it.  PMC: 5.2 GB of L2 requests per launch (hipBLASLt: 4.7 GB), 76 % hits, HBM fetches 1.4x ideal: the
// bound is L2 -> CU bandwidth (~10 TB/s here), and a 256 x 256 tile -- the largest the register file allows -- needs
// one byte per 128 flops: ~1.3 PFLOP/s is the ceiling at K = 512 whoever writes the kernel.
#include "hstu_common.cuh"
#include "capi_internal.h"

namespace hstu {

constexpr int kGemmThreads = 512;
constexpr int kGemmWaves = 8;
constexpr int kBM = 256, kBN = 256, kBK = 32;
#ifndef GEMM_STAGES
#define GEMM_STAGES 4
#endif
constexpr int kNS = GEMM_STAGES;               // ring of K-step stages: kNS - 1 steps of loads in flight
constexpr int kATile = kBM * kBK * 2;          // 16 KiB
constexpr int kBTile = kBK * kBN * 2;          // 16 KiB
constexpr int kStage = kATile + kBTile;
constexpr int kGemmSmem = kNS * kStage < 128 * 1024 ? 128 * 1024 : kNS * kStage;   // >= the 128 KiB C tile of the epilogue
constexpr int kDmaPerStep = 2 * (kATile / 1024) / kGemmWaves;   // LDS-DMA instructions per wave and step (A + B)

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
HSTU_DEV void gemm_dma16(const char* g, uint32_t lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory", "m0");
}
#pragma clang diagnostic pop

struct GemmArgs {
  const void* a; const void* b; void* c; const void* bias; const void* residual;
  int64_t m, lda, ldb, ldc, ldr;
  int n, k, silu_cols, n_tiles, m_tiles;
};

// stage loads of K step `kt`: A rows [m0, m0+256) x k [64 kt, +64), B rows k x cols [n0, n0+256)
template <typename T>
HSTU_DEV void gemm_stage_dma(const GemmArgs& g, char* stage, int64_t m0, int n0, int kt, int wave, int lane) {
  const uint32_t lds_a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)stage);
  const uint32_t lds_b = lds_a + kATile;
  const char* A = (const char*)g.a;
  const char* B = (const char*)g.b;
  constexpr int UA = kBK / 8;                     // 16-byte units per A row
#pragma unroll
  for (int j = 0; j < kATile / 1024 / kGemmWaves; ++j) {
    const int c = wave + kGemmWaves * j;          // 1 KiB chunk
    const int pidx = c * 64 + lane;
    {  // A: UA units per row
      const int row = pidx / UA, slot = pidx % UA;
      const int unit = slot ^ swz<UA>(row);
      int64_t grow = m0 + row;
      grow = grow < g.m ? grow : g.m - 1;
      gemm_dma16(A + (grow * g.lda + kt * kBK + unit * 8) * 2, lds_a + c * 1024);
    }
    {  // B: 32 units per row
      const int row = pidx >> 5, slot = pidx & 31;
      const int unit = slot ^ swz<32>(row);
      gemm_dma16(B + ((int64_t)(kt * kBK + row) * g.ldb + n0 + unit * 8) * 2, lds_b + c * 1024);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kGemmThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nn_kernel(const GemmArgs g) {
  using E = Elem<T>;
  using Frag = typename E::Frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31, hf = lane >> 5;
  // workgroup -> tile: runs of n_tiles consecutive tiles (one M panel) stay on one XCD (ids are dealt round-robin
  // to the 8 XCDs by the dispatcher)
  const int total = g.m_tiles * g.n_tiles;
  int id = blockIdx.x;
  {
    const int per = total / 8;                    // tiles per XCD in the remappable prefix
    if (per > 0 && id < per * 8) id = (id & 7) * per + (id >> 3);
  }
  const int mt = id / g.n_tiles, nt = id - mt * g.n_tiles;
  const int64_t m0 = (int64_t)mt * kBM;
  const int n0 = nt * kBN;
  const int wm = wave >> 2, wn = wave & 3;

  f32x16 acc[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][mb][r] = 0.f;

  const int nk = g.k / kBK;
  // ring: the loads of steps kt+1 .. kt+kNS-2 stay in flight while step kt is multiplied (one step of HBM/L2 latency
  // is ~2 us, a step of MFMA work ~0.5 us: with a single step of prefetch every step paid the difference)
#pragma unroll
  for (int s0 = 0; s0 < kNS - 1; ++s0)
    if (s0 < nk) gemm_stage_dma<T>(g, smem + s0 * kStage, m0, n0, s0, wave, lane);
  for (int kt = 0; kt < nk; ++kt) {
    // vector memory operations retire in order: allow the kNS-2 younger steps to be outstanding
    if (kt + kNS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerStep * (kNS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                // stage kt landed everywhere; stage kt-1 fully consumed
#ifndef GEMM_EXP_NO_DMA
    if (kt + kNS - 1 < nk) gemm_stage_dma<T>(g, smem + ((kt + kNS - 1) % kNS) * kStage, m0, n0, kt + kNS - 1, wave, lane);
#endif
    const char* At = smem + (kt % kNS) * kStage;
    const char* Bt = At + kATile;
#ifndef GEMM_EXP_NO_COMPUTE
#pragma unroll
    for (int kk = 0; kk < kBK / 16; ++kk) {
      Frag af[4], bf[2];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) af[mb] = lds_row_frag<T, kBK / 8>(At, 128 * wm + 32 * mb + n32, 16 * kk + 8 * hf);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        bf[nb] = lds_col_frag<T, 32>(Bt, 16 * kk + 8 * hf, 16 * kk + 8 * hf + 4, 64 * wn + 32 * nb, lane);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = E::mma(bf[nb], af[mb], acc[nb][mb]);   // C^T block: rows n, cols m
    }
#endif
  }
  lds_barrier();                                  // every wave is done with the stages: they become the C tile

  // ---- epilogue.  acc[nb][mb][r]: column m = 32 mb + n32 of C^T (= row of C), row n = 32 nb + 8 (r >> 2) + 4 hf + (r & 3)
  char* mine = smem + wave * (128 * 64 * 2);      // this wave's [128 rows][64 cols] of C, 8 units per row
  const T* bias = (const T*)g.bias;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int ncol0 = n0 + 64 * wn + 32 * nb;     // first column of the block (wave-uniform)
    const bool do_silu = ncol0 < g.silu_cols;     // silu_cols is a multiple of 32
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
        const T* bp = bias + ncol0 + 8 * rq + 4 * hf;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = (float)bp[j];
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = acc[nb][mb][4 * rq + j] + bv[j];
          if (do_silu) x = x * fast_sigmoid(x);
          v[j] = x;
        }
        const u32x2 w = {E::pk2(v[0], v[1]), E::pk2(v[2], v[3])};
        *LDS_PTR(u32x2, mine + tile_off<8>(32 * mb + n32, 4 * nb + rq) + 8 * hf) = w;
      }
    }
  }
  // copy-out: 8 lanes per row (128 bytes = the wave's 64 columns), 8 rows per pass
  const int crow = lane >> 3, cunit = lane & 7;
  char* C = (char*)g.c;
  const char* R = (const char*)g.residual;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int row = 8 * p + crow;
    const int64_t grow = m0 + 128 * wm + row;
    u32x4 v = *LDS_PTR(const u32x4, mine + tile_off<8>(row, cunit));
    if (grow < g.m) {
      const int64_t col = n0 + 64 * wn + 8 * cunit;
      if (R) {
        const u32x4 rv = gload16(R + (grow * g.ldr + col) * 2);
        typedef T t8 __attribute__((ext_vector_type(8)));
        const t8 a = __builtin_bit_cast(t8, v), b = __builtin_bit_cast(t8, rv);
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = (float)a[j] + (float)b[j];
        v = u32x4{E::pk2(s[0], s[1]), E::pk2(s[2], s[3]), E::pk2(s[4], s[5]), E::pk2(s[6], s[7])};
      }
      gstore16(C + (grow * g.ldc + col) * 2, v);
    }
  }
}

template <typename T>
static int gemm_launch(const GemmArgs& g, hipStream_t st) {
  auto kern = gemm_nn_kernel<T>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmSmem);
  if (e != hipSuccess) return set_error(HSTU_ELAUNCH, "hstu_gemm: cannot reserve %d bytes of LDS: %s", kGemmSmem, hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(g.m_tiles * g.n_tiles), dim3(kGemmThreads), kGemmSmem, st, g);
  return check_launch("hstu_gemm");
}

}  // namespace hstu

using namespace hstu;

extern "C" {

int hstu_gemm_supported(int64_t m, int32_t n, int32_t k, int dtype) {
  return m > 0 && n > 0 && k > 0 && n % kBN == 0 && k % kBK == 0 && (dtype == HSTU_DTYPE_BF16 || dtype == HSTU_DTYPE_F16) &&
         (m + kBM - 1) / kBM * (int64_t)(n / kBN) < 0x7fffffffLL;
}

int hstu_gemm(const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc, const void* bias,
              const void* residual, int64_t ldr, int64_t m, int32_t n, int32_t k, int32_t silu_cols, int dtype,
              void* stream) {
  if (m == 0) return HSTU_OK;
  if (!hstu_gemm_supported(m, n, k, dtype))
    return set_error(HSTU_EUNSUPPORTED, "hstu_gemm: needs 16-bit I/O, N %% %d == 0 and K %% %d == 0 (got M=%lld N=%d K=%d)", kBN, kBK,
                     (long long)m, n, k);
  if (!a || !b || !c) return set_error(HSTU_EINVAL, "hstu_gemm: NULL matrix");
  if ((lda | ldb | ldc | (residual ? ldr : 0)) % 8 || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)residual | (uintptr_t)bias) & 15))
    return set_error(HSTU_EINVAL, "hstu_gemm: rows must be 16-byte aligned");
  if (silu_cols < 0 || silu_cols > n || silu_cols % 32) return set_error(HSTU_EINVAL, "hstu_gemm: silu_cols must be a multiple of 32 in [0, N]");
  GemmArgs g{a, b, c, bias, residual, m, lda, ldb, ldc, ldr, n, k, silu_cols, n / kBN, (int)((m + kBM - 1) / kBM)};
  hipStream_t st = (hipStream_t)stream;
  return dtype == HSTU_DTYPE_BF16 ? gemm_launch<bf16_t>(g, st) : gemm_launch<f16_t>(g, st);
}

}  // extern "C"
