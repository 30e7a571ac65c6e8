// Micro-benchmark of LDS access patterns on gfx950 (tool, not product): every wave repeats the same
// 64 per-lane byte addresses; reports s_memtime cycles per wave-instruction.  Used to pick the tile
// swizzle of csrc/hstu_common.cuh (ds_read_b128 row fragments + ds_read_b64_tr_b16 transposed fragments).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void lds_bench_kernel(const int* addr, int iters, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 16384 * (int)(blockDim.x >> 6) / 4; i += blockDim.x) ((uint32_t*)lds)[i] = i;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)lds + addr[threadIdx.x & 63] + (threadIdx.x >> 6) * 16384;
  uint32_t sink = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
      u32x2 r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[j]) : "v"(a));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) sink ^= r[j][0] ^ r[j][1];
    } else if constexpr (KIND == 1) {
      u32x4 r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(r[j]) : "v"(a));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) sink ^= r[j][0] ^ r[j][3];
    } else if constexpr (KIND == 2) {
      u32x2 d = {sink, (uint32_t)it};
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(d) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      u32x2 r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64 %0, %1" : "=v"(r[j]) : "v"(a));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) sink ^= r[j][0] ^ r[j][1];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (sink == 0x12345678u) out[63] = sink;
}

extern "C" int lds_bench_run(const int* addr, int kind, int iters, int threads, unsigned long long* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int smem = 16384 * (threads / 64);
  switch (kind) {
    case 0: hipLaunchKernelGGL(lds_bench_kernel<0>, dim3(1), dim3(threads), smem, st, addr, iters, out); break;
    case 1: hipLaunchKernelGGL(lds_bench_kernel<1>, dim3(1), dim3(threads), smem, st, addr, iters, out); break;
    case 2: hipLaunchKernelGGL(lds_bench_kernel<2>, dim3(1), dim3(threads), smem, st, addr, iters, out); break;
    default: hipLaunchKernelGGL(lds_bench_kernel<3>, dim3(1), dim3(threads), smem, st, addr, iters, out); break;
  }
  return (int)hipGetLastError();
}
