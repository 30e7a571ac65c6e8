// Micro-benchmark of the MFMA streams of the wide backward (tools/ubench): what paces a 16-MFMA stream on ONE wave per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 stream_bench.hip -o stream_bench ; run on the GPU box, prints cycles / MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

DEV int swz16(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
DEV int tile_off(int r, int u) { return (r * 16 + (u ^ swz16(r))) << 4; }
DEV void mma_v(bool first, u32x4 a, u32x4 b, f32x16& c) {
  if (first) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
DEV void mma_va(bool first, u32x4 a, u32x4 b, f32x16& c) {
  if (first) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
DEV u32x4 row_frag(const char* tile, int row, int e0) { return *LDS_PTR(const u32x4, tile + tile_off(row, e0 >> 3)); }
DEV u32x4 col_frag(const char* tile, int rowA, int rowB, int colblk, int lane) {
  const int i16 = lane & 15;
  const int col = colblk + (((lane >> 4) & 1) << 4) + ((i16 & 3) << 2);
  const int sub = (col & 7) << 1;
  const int ra = rowA + (i16 >> 2), rb = rowB + (i16 >> 2);
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off(ra, col >> 3) + sub));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + tile_off(rb, col >> 3) + sub));
  s16x8 ab = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(u32x4, ab);
}

// variant ids
enum { V_MFMA_BUILTIN = 0, V_MFMA_ASM_V, V_MFMA_ASM_A, V_SDP_LDS_A, V_SDP_LDS_V, V_SDP_LDS_BOTH, V_DVDK_D2, V_DVDK_D8, V_DVDK_ASM_D2, V_DQ_CHAIN, V_SDP_NOADDR, V_EW, NV };
static const char* kNames[NV] = {"mfma only, builtin (AGPR acc), 2 alternating acc", "mfma only, asm VGPR acc, B in VGPR", "mfma only, asm VGPR acc, B in AGPR",
                                 "S/dP: A = ds_read_b128 (ahead 4), B AGPR regs", "S/dP: A = ds_read_b128 (ahead 4), B VGPR regs", "S/dP: A and B = ds_read_b128 (fold)",
                                 "dV/dK: builtin, A = 2 tr reads (ahead 4), acc dependence distance 2", "dV/dK: the same, dependence distance 8",
                                 "dV/dK: asm VGPR acc, distance 2", "dQ chain: one acc, A and B = tr reads (ahead 2 tiles)", "S/dP: ds_read_b128 at immediate offsets (no address VALU)",
                                 "element-wise block only (16 elements, packed)"};

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bench(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n32 = lane & 31, hf = lane >> 5;
  for (int i = tid; i < 4 * 16384 / 16; i += 256) *LDS_PTR(u32x4, smem + 16 * i) = src[i];
  __syncthreads();
  const char* Qs = smem;
  const char* dOs = smem + 8192;
  const char* Kt = smem + 16384;
  u32x4 kf[8], vf[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) { kf[m] = row_frag(Kt, n32, hf * 64 + m * 8); vf[m] = row_frag(Kt + 8192, n32, hf * 64 + m * 8); }
  f32x16 acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  f32x16 s, dp;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
  u32x4 pb[2] = {kf[0], kf[1]}, dsb[2] = {vf[0], vf[1]};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int n32 = ln & 31, hf = ln >> 5;
    if constexpr (V == V_MFMA_BUILTIN) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, kf[m >> 1]), __builtin_bit_cast(bf8, vf[m >> 1]), acc[m & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == V_MFMA_ASM_V) {
#pragma unroll
      for (int m = 0; m < 16; ++m) { if (m & 1) mma_v(false, kf[m >> 1], vf[m >> 1], dp); else mma_v(false, vf[m >> 1], kf[m >> 1], s); }
    } else if constexpr (V == V_MFMA_ASM_A) {
#pragma unroll
      for (int m = 0; m < 16; ++m) { if (m & 1) mma_va(false, pb[0], vf[m >> 1], dp); else mma_va(false, pb[1], kf[m >> 1], s); }
    } else if constexpr (V == V_SDP_LDS_A || V == V_SDP_LDS_V || V == V_SDP_LDS_BOTH) {
      constexpr int AH = 4;
      u32x4 fa[AH + 1], fb[AH + 1];
      auto load_item = [&](int m, u32x4& a, u32x4& b) {
        const int e0 = hf * 64 + (m >> 1) * 8;
        a = row_frag((m & 1) ? dOs : Qs, n32, e0);
        if (V == V_SDP_LDS_BOTH) b = row_frag((m & 1) ? Kt + 8192 : Kt, n32, e0);
      };
#pragma unroll
      for (int m = 0; m < AH; ++m) load_item(m, fa[m % (AH + 1)], fb[m % (AH + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m + AH < 16) load_item(m + AH, fa[(m + AH) % (AH + 1)], fb[(m + AH) % (AH + 1)]);
        if (V == V_SDP_LDS_A) { if (m & 1) mma_va(m < 2, fa[m % (AH + 1)], vf[m >> 1], dp); else mma_va(m < 2, fa[m % (AH + 1)], kf[m >> 1], s); }
        else if (V == V_SDP_LDS_V) { if (m & 1) mma_v(m < 2, fa[m % (AH + 1)], vf[m >> 1], dp); else mma_v(m < 2, fa[m % (AH + 1)], kf[m >> 1], s); }
        else { if (m & 1) mma_v(m < 2, fa[m % (AH + 1)], fb[m % (AH + 1)], dp); else mma_v(m < 2, fa[m % (AH + 1)], fb[m % (AH + 1)], s); }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == V_SDP_NOADDR) {
      // lane-constant base, immediate offsets: the swizzle term is dropped (conflicts aside, this isolates the address VALU)
      const char* qb = Qs + n32 * 256 + hf * 128;
      const char* ob = dOs + n32 * 256 + hf * 128;
      constexpr int AH = 4;
      u32x4 fa[AH + 1];
#pragma unroll
      for (int m = 0; m < AH; ++m) fa[m] = *LDS_PTR(const u32x4, ((m & 1) ? ob : qb) + (m >> 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m + AH < 16) fa[(m + AH) % (AH + 1)] = *LDS_PTR(const u32x4, (((m + AH) & 1) ? ob : qb) + ((m + AH) >> 1) * 16);
        if (m & 1) mma_va(m < 2, fa[m % (AH + 1)], vf[m >> 1], dp); else mma_va(m < 2, fa[m % (AH + 1)], kf[m >> 1], s);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == V_DVDK_D2 || V == V_DVDK_D8 || V == V_DVDK_ASM_D2) {
      constexpr int AH = 4;
      u32x4 fa[AH + 1];
      // item m: D2: (dV | dK) fastest, then k half, then d block (as the kernel); D8: (dV | dK), d block, then k half
      auto decode = [&](int m, int& which, int& ks, int& d) {
        which = m & 1;
        if (V == V_DVDK_D8) { d = (m >> 1) & 3; ks = m >> 3; } else { ks = (m >> 1) & 1; d = m >> 2; }
      };
      auto load_item = [&](int m, u32x4& a) {
        int which, ks, d;
        decode(m, which, ks, d);
        a = col_frag(which ? Qs : dOs, 16 * ks + 4 * hf, 16 * ks + 8 + 4 * hf, 32 * d, ln);
      };
#pragma unroll
      for (int m = 0; m < AH; ++m) load_item(m, fa[m % (AH + 1)]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m + AH < 16) load_item(m + AH, fa[(m + AH) % (AH + 1)]);
        int which, ks, d;
        decode(m, which, ks, d);
        if (V == V_DVDK_ASM_D2) mma_v(false, fa[m % (AH + 1)], which ? dsb[ks] : pb[ks], acc[2 * d + which]);
        else acc[2 * d + which] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fa[m % (AH + 1)]), __builtin_bit_cast(bf8, which ? dsb[ks] : pb[ks]), acc[2 * d + which], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == V_DQ_CHAIN) {
      // 8 key tiles (all aliased to the same LDS tiles), 2 MFMAs each, one accumulator, fragments of tile t + 2 in flight
      constexpr int AH = 2, N = 8;
      u32x4 a0[AH + 1], a1[AH + 1], b0[AH + 1], b1[AH + 1];
      const int ra = 8 * hf, rb = 16 + 8 * hf;
      auto load_tile = [&](int t, int sl) {
        const char* K2 = Kt + (t & 1) * 8192;
        a0[sl] = col_frag(K2, ra, ra + 4, 32 * (threadIdx.x >> 6), ln);
        a1[sl] = col_frag(K2, rb, rb + 4, 32 * (threadIdx.x >> 6), ln);
        b0[sl] = col_frag(Qs, ra, ra + 4, 0, ln);
        b1[sl] = col_frag(Qs, rb, rb + 4, 0, ln);
      };
#pragma unroll
      for (int t = 0; t < AH; ++t) load_tile(t, t);
#pragma unroll
      for (int t = 0; t < N; ++t) {
        if (t + AH < N) load_tile(t + AH, (t + AH) % (AH + 1));
        mma_v(t == 0, a0[t % (AH + 1)], b0[t % (AH + 1)], s);
        mma_v(false, a1[t % (AH + 1)], b1[t % (AH + 1)], s);
      }
    } else if constexpr (V == V_EW) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      const float alpha = 0.0883f;
      const f32x2 a2 = {alpha, alpha}, c2 = {-1.44269504f * alpha, -1.44269504f * alpha}, one2 = {1.f, 1.f};
      uint32_t w[16];
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const f32x2 sv = {s[j], s[j + 1]}, dpv = {dp[j], dp[j + 1]};
        const f32x2 x = sv * a2, t = sv * c2;
        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        const f32x2 dn = e + one2;
        const f32x2 sg = {__builtin_amdgcn_rcpf(dn[0]), __builtin_amdgcn_rcpf(dn[1])};
        const f32x2 pr = x * sg;
        const f32x2 ww = x * (one2 - sg) + one2;
        const f32x2 dsr = dpv * sg * ww;
        typedef __bf16 h2 __attribute__((ext_vector_type(2)));
        w[j / 2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, h2));
        w[8 + j / 2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(dsr, h2));
      }
      pb[0] = u32x4{w[0], w[1], w[2], w[3]}; pb[1] = u32x4{w[4], w[5], w[6], w[7]};
      dsb[0] = u32x4{w[8], w[9], w[10], w[11]}; dsb[1] = u32x4{w[12], w[13], w[14], w[15]};
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] += __builtin_bit_cast(float, pb[r & 1][r & 3] & 0x3fffffffu) * 1e-9f; dp[r] += 1e-9f; }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc[d][r];
#pragma unroll
  for (int r = 0; r < 16; ++r) sum += s[r] + dp[r];
  sum += __builtin_bit_cast(float, pb[0][0] ^ dsb[1][3]);
  out[blockIdx.x * 256 + tid] = sum;
  if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int V>
static void run(const u32x4* src, float* out, unsigned long long* cyc, int grid, int reps, int lds) {
  auto k = bench<V>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, src, out, cyc, reps);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, src, out, cyc, reps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; unsigned long long mx = 0;
  for (auto c : h) { avg += c; if (c > mx) mx = c; }
  avg /= h.size();
  const int per = (V == V_EW) ? 1 : 16;
  printf("%-80s grid %4d: %8.1f cycles / %s (max wave %8.1f)  wall %.3f ms -> %.2f GHz-equivalent\n", kNames[V], grid, avg / reps / per, V == V_EW ? "block" : "MFMA",
         (double)mx / reps / per, ms, avg / (ms * 1e6));
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 2000;
  u32x4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 65536); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
  std::vector<uint16_t> h(32768);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 22);   // bf16 values around 0.01 .. 0.03
  hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
  for (int grid : {1, 256}) {
    const int lds = 160 * 1024;
    run<V_MFMA_BUILTIN>(src, out, cyc, grid, reps, lds);
    run<V_MFMA_ASM_V>(src, out, cyc, grid, reps, lds);
    run<V_MFMA_ASM_A>(src, out, cyc, grid, reps, lds);
    run<V_SDP_LDS_A>(src, out, cyc, grid, reps, lds);
    run<V_SDP_LDS_V>(src, out, cyc, grid, reps, lds);
    run<V_SDP_LDS_BOTH>(src, out, cyc, grid, reps, lds);
    run<V_SDP_NOADDR>(src, out, cyc, grid, reps, lds);
    run<V_DVDK_D2>(src, out, cyc, grid, reps, lds);
    run<V_DVDK_D8>(src, out, cyc, grid, reps, lds);
    run<V_DVDK_ASM_D2>(src, out, cyc, grid, reps, lds);
    run<V_DQ_CHAIN>(src, out, cyc, grid, reps, lds);
    run<V_EW>(src, out, cyc, grid, reps, lds);
  }
  return 0;
}
