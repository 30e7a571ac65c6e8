// tools/membench: what HBM bandwidth can a streaming kernel reach on this part?  (context for the roofline
// fractions: 8 TB/s is the vendor peak; this measures read-only, write-only, copy and a 3:1 read:write mix with
// 16-byte accesses, grid-stride, for a few grid sizes.)   hipcc --offload-arch=gfx950 -O3 mem_bench.hip -o mem_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const u32x4* a, size_t n, u32x4* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= __builtin_nontemporal_load(a + i);
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_write(u32x4* a, size_t n) {
  const u32x4 v = {1, 2, 3, 4};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, a + i);
}
__global__ __launch_bounds__(256) void k_copy(const u32x4* a, u32x4* b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// 3 reads : 1 write (the attention forward's mix)
__global__ __launch_bounds__(256) void k_mix31(const u32x4* a, const u32x4* b, const u32x4* c, u32x4* d, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = a[i] ^ b[i] ^ c[i];
}
// 4 reads : 3 writes (the attention backward's mix)
__global__ __launch_bounds__(256) void k_mix43(const u32x4* a, const u32x4* b, const u32x4* c, const u32x4* e, u32x4* d0,
                                               u32x4* d1, u32x4* d2, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const u32x4 x = a[i], y = b[i], z = c[i], w = e[i];
    d0[i] = x ^ y; d1[i] = z ^ w; d2[i] = x ^ w;
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename F> static double time_ms(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}
int main() {
  const size_t bytes = 1ull << 30, n = bytes / 16;   // 1 GiB per stream
  u32x4* buf[7];
  for (auto& p : buf) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
  u32x4* sink; CK(hipMalloc(&sink, 16));
  printf("{");
  const int grids[] = {1024, 2048, 4096, 16384, 65536};
  bool first = true;
  for (int g : grids) {
    double t;
    t = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, buf[0], n, sink); }, 10);
    printf("%s\"read_g%d\": %.0f", first ? "" : ", ", g, bytes / t / 1e6); first = false;
    t = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, buf[1], n); }, 10);
    printf(", \"write_g%d\": %.0f", g, bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, buf[0], buf[1], n); }, 10);
    printf(", \"copy_g%d\": %.0f", g, 2.0 * bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_mix31, dim3(g), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], n); }, 10);
    printf(", \"r3w1_g%d\": %.0f", g, 4.0 * bytes / t / 1e6);
    t = time_ms([&] { hipLaunchKernelGGL(k_mix43, dim3(g), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], n); }, 10);
    printf(", \"r4w3_g%d\": %.0f", g, 7.0 * bytes / t / 1e6);
  }
  printf(", \"unit\": \"GB/s\"}\n");
  return 0;
}
