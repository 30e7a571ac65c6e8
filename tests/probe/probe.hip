// Test-only probes of the gfx950 facts the attention kernels rely on (MFMA fragment
// layouts, ds_read_b64_tr_b16 semantics).  Built into tests/probe/libhstu_probe.so by
// __graft_entry__.build(); never linked into the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// C = A(32x16) * B(16x32), operands given densely in global memory (bf16 as uint16 bits).
// Lane l loads A[l&31][8*(l>>5) + j] and B[8*(l>>5) + j][l&31]; writes its 16 accumulators.
__global__ void probe_mfma_bf16(const uint16_t* A, const uint16_t* B, float* Cregs) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(__bf16, A[(l & 31) * 16 + 8 * (l >> 5) + j]);
    b[j] = __builtin_bit_cast(__bf16, B[(8 * (l >> 5) + j) * 32 + (l & 31)]);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Cregs[l * 16 + r] = c[r];
}

// 16x16x32 bf16: lane l supplies A[l&15][8*(l>>4) + j], B[8*(l>>4) + j][l&15]; writes its 4 accumulators.
typedef float f32x4p __attribute__((ext_vector_type(4)));
__global__ void probe_mfma16_bf16(const uint16_t* A, const uint16_t* B, float* Cregs) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(__bf16, A[(l & 15) * 32 + 8 * (l >> 4) + j]);
    b[j] = __builtin_bit_cast(__bf16, B[(8 * (l >> 4) + j) * 16 + (l & 15)]);
  }
  f32x4p c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Cregs[l * 4 + r] = c[r];
}

// fp32 32x32x2: lane l supplies A[l&31][l>>5], B[l>>5][l&31]
__global__ void probe_mfma_f32(const float* A, const float* B, float* Cregs) {
  const int l = threadIdx.x;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Cregs[l * 16 + r] = c[r];
}

// LDS holds lds[i] = i (int16).  Every lane reads with ds_read_b64_tr_b16 from the byte
// address addr[l]; out[l*4 + j] = the 4 values it received.
__global__ void probe_tr_read(const int* addr, int16_t* out) {
  __shared__ __attribute__((aligned(16))) int16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (int16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((char*)lds + addr[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = t[j];
}

typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
// each lane copies 16 B from src[perm[lane]] (16-byte units) into LDS via LDS-DMA; then the LDS
// image (64 units) is written back linearly to out.
__global__ void probe_glds(const pu32x4* src, const int* perm, pu32x4* out) {
  __shared__ __attribute__((aligned(16))) pu32x4 lds[256];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = pu32x4{0xdeadu, 0, 0, 0};
  __syncthreads();
  const pu32x4* g = src + perm[l] + 64 * w;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + 64 * w), 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = lds[i];
}

// Fill the whole 160 KiB LDS of every CU with a bit pattern (0xFFFFFFFF = NaN as fp32, bf16 and fp16): kernels that
// read LDS they never wrote then fail loudly instead of depending on what the previous kernel left behind.
__global__ __launch_bounds__(256) void probe_poison_lds_kernel(uint32_t pattern, int spin, uint32_t* sink) {
  extern __shared__ uint32_t poison_lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) poison_lds[i] = pattern;
  __syncthreads();
  uint32_t x = poison_lds[threadIdx.x];
  for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;    // keeps the workgroup resident: one per CU at a time
  if (x == 0x12345u) sink[0] = x;
}

extern "C" {
int probe_poison_lds(uint32_t pattern, void* sink, void* stream) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)probe_poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
    attr = true;
  }
  hipLaunchKernelGGL(probe_poison_lds_kernel, dim3(2048), dim3(256), 160 * 1024, (hipStream_t)stream, pattern, 20000, (uint32_t*)sink);
  return (int)hipGetLastError();
}
int probe_run_mfma_bf16(const void* A, const void* B, void* C, void* stream) {
  hipLaunchKernelGGL(probe_mfma_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)B, (float*)C);
  return (int)hipGetLastError();
}
int probe_run_mfma16_bf16(const void* A, const void* B, void* C, void* stream) {
  hipLaunchKernelGGL(probe_mfma16_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)A, (const uint16_t*)B, (float*)C);
  return (int)hipGetLastError();
}
int probe_run_mfma_f32(const void* A, const void* B, void* C, void* stream) {
  hipLaunchKernelGGL(probe_mfma_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)A, (const float*)B, (float*)C);
  return (int)hipGetLastError();
}
int probe_run_tr_read(const void* addr, void* out, void* stream) {
  hipLaunchKernelGGL(probe_tr_read, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)addr, (int16_t*)out);
  return (int)hipGetLastError();
}
int probe_run_glds(const void* src, const void* perm, void* out, void* stream) {
  hipLaunchKernelGGL(probe_glds, dim3(1), dim3(256), 0, (hipStream_t)stream, (const pu32x4*)src, (const int*)perm, (pu32x4*)out);
  return (int)hipGetLastError();
}
}
