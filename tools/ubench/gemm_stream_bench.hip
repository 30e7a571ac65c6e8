// Micro-benchmark of the inner loop of the fused LayerNorm + projection kernel (csrc/hstu_ln_linear.cuh): what paces a
// stream of 32x32x16 bf16 MFMAs whose A operand comes from LDS (one ds_read_b128 each, 4 ahead) and whose B operand sits
// in registers?  Waves per SIMD, independent accumulator chains per wave, a workgroup barrier every 32 MFMAs, and how many
// MFMAs share one LDS fragment are the knobs.  Whole chip (256 workgroups) and one workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_stream_bench.hip -o gemm_stream_bench ; prints cycles / MFMA / SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

DEV int swz64(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
DEV uint32_t tile_off(int r, int u) { return (uint32_t)((r * 64 + (u ^ swz64(r))) << 4); }
DEV f32x16 mma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// WAVES: 4 or 8 per workgroup; CHAINS: 1 or 2 independent accumulators; BAR: barrier every 32 MFMAs of a wave;
// LDSA: A operand from LDS (else registers); SHARE: MFMAs per LDS fragment (1, or 2 = two row tiles per wave: B regs x 2)
template <int WAVES, int CHAINS, bool BAR, bool LDSA, int SHARE>
__global__ __launch_bounds__(64 * WAVES) void bench(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];     // 3 tiles of 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 3 * 32768 / 16; i += 64 * WAVES) *LDS_PTR(u32x4, smem + 16 * i) = src[i & 4095];
  const int m = lane & 31, h = lane >> 5;
  constexpr int NB = SHARE == 2 ? 16 : 32;      // B fragments kept (SHARE 2: 2 x 16 = the same 128 registers, half of K each)
  u32x4 xf[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) xf[i] = src[(tid + 64 * i) & 4095];
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  uint32_t fa[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) fa[kk] = (uint32_t)(uintptr_t)smem + tile_off(m, 2 * kk + h);
  __syncthreads();
  u32x4 wf[4];
  int slot = 0;
  auto frag = [&](int sl, int ks) { return *LDS_PTR(const u32x4, (uintptr_t)(fa[ks & 7] + sl * 32768 + (ks >> 3) * 256)); };
#pragma unroll
  for (int i = 0; i < 4; ++i) wf[i] = LDSA ? frag(0, i) : xf[i];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    const int ns = slot == 2 ? 0 : slot + 1;
    if constexpr (SHARE == 1) {
#pragma unroll
      for (int ks = 0; ks < 32; ++ks) {
        f32x16& c = acc[CHAINS == 2 ? (ks & 1) : 0];
        c = mma(wf[ks & 3], xf[ks], c);
        if (LDSA) wf[ks & 3] = ks + 4 < 32 ? frag(slot, ks + 4) : frag(ns, ks + 4 - 32);
        if (BAR && ks == 15) asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < NB; ++ks) {       // 16 fragments, each into two MFMAs (row tiles 0 and 1)
        acc[0] = mma(wf[ks & 3], xf[ks], acc[0]);
        acc[1] = mma(wf[ks & 3], xf[16 + ks], acc[1]);
        if (LDSA) wf[ks & 3] = ks + 4 < NB ? frag(slot, ks + 4) : frag(ns, ks + 4 - NB);
        if (BAR && ks == 7) asm volatile("s_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    slot = ns;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
  out[blockIdx.x * 64 * WAVES + tid] = s;
  if (lane == 0) cyc[blockIdx.x * WAVES + (tid >> 6)] = t1 - t0;
}

template <int WAVES, int CHAINS, bool BAR, bool LDSA, int SHARE>
static void run(const char* name, const u32x4* src, float* out, unsigned long long* cyc, int grid) {
  auto k = bench<WAVES, CHAINS, BAR, LDSA, SHARE>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
  const int reps = 2000;
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WAVES), 3 * 32768, 0, src, out, cyc, reps);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WAVES), 3 * 32768, 0, src, out, cyc, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * WAVES);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  const double mfma_per_simd = (double)reps * 32 * (WAVES / 4);
  printf("%-86s grid %3d: %6.1f cycles / MFMA / SIMD   wall %.3f ms -> %.2f GHz   %.0f TFLOP/s\n", name, grid, mx / mfma_per_simd, ms,
         mx / (ms * 1e6), grid * 4 * mfma_per_simd * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  u32x4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 4096 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  std::vector<uint32_t> h(4096 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c003c00u + (uint32_t)(i * 2654435761u >> 20 & 0x00ff00ffu);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int grid : {1, 256}) {
    run<8, 1, false, false, 1>("2 waves/SIMD, 1 chain, A in registers (MFMA only)", src, out, cyc, grid);
    run<8, 2, false, false, 1>("2 waves/SIMD, 2 chains, A in registers", src, out, cyc, grid);
    run<4, 1, false, false, 1>("1 wave/SIMD, 1 chain, A in registers", src, out, cyc, grid);
    run<4, 2, false, false, 1>("1 wave/SIMD, 2 chains, A in registers", src, out, cyc, grid);
    run<8, 1, false, true, 1>("2 waves/SIMD, 1 chain, A = ds_read_b128 (4 ahead)            [the kernel's loop]", src, out, cyc, grid);
    run<8, 1, true, true, 1>("2 waves/SIMD, 1 chain, A from LDS, barrier every 32 MFMAs    [the kernel's loop]", src, out, cyc, grid);
    run<8, 2, false, true, 1>("2 waves/SIMD, 2 chains, A from LDS", src, out, cyc, grid);
    run<8, 2, true, true, 1>("2 waves/SIMD, 2 chains, A from LDS, barrier every 32", src, out, cyc, grid);
    run<4, 1, false, true, 1>("1 wave/SIMD, 1 chain, A from LDS", src, out, cyc, grid);
    run<4, 2, false, true, 1>("1 wave/SIMD, 2 chains, A from LDS", src, out, cyc, grid);
    run<8, 2, false, true, 2>("2 waves/SIMD, 2 row tiles per fragment (half the LDS reads)", src, out, cyc, grid);
    run<8, 2, true, true, 2>("2 waves/SIMD, 2 row tiles per fragment, barrier every 32", src, out, cyc, grid);
    run<4, 2, false, true, 2>("1 wave/SIMD, 2 row tiles per fragment", src, out, cyc, grid);
  }
  return 0;
}
