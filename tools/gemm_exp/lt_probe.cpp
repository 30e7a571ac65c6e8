// hipBLASLt feasibility probe for the STU layer's projections (stand-alone: no torch in the process).
//   1. output stage  D = Y Wo + X with C != D (torch.addmm copies X into D first: a 210 MB pass per layer)
//   2. UVQK weight gradient with the bias gradient as an epilogue (BGRADA: the column sums of d uvqk, a 840 MB pass today)
//      as one GEMM and as 16 row slabs (the split ops/mm.py::weight_grad_mm uses)
// build: hipcc -O2 --offload-arch=gfx950 lt_probe.cpp -o lt_probe -lhipblaslt      run: ./lt_probe [rows]
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                              \
  do {                                                                     \
    auto e_ = (x);                                                         \
    if ((int)e_ != 0) {                                                    \
      printf("FAILED %s -> %d (line %d)\n", #x, (int)e_, __LINE__);        \
      return 1;                                                            \
    }                                                                      \
  } while (0)

struct Gemm {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
  hipblasLtMatmulHeuristicResult_t heur[8];
  int nheur = 0;
};

static int make(hipblasLtHandle_t h, Gemm& g, hipblasOperation_t ta, hipblasOperation_t tb, int64_t m, int64_t n, int64_t k, int64_t lda,
                int64_t ldb, int64_t ldc, hipDataType tc, hipDataType td, int batch, int64_t sa, int64_t sb, int64_t sc, uint32_t epilogue,
                void* bias, hipDataType tbias, size_t ws) {
  CK(hipblasLtMatmulDescCreate(&g.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  CK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  CK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  if (epilogue != HIPBLASLT_EPILOGUE_DEFAULT) {
    CK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epilogue, sizeof(epilogue)));
    CK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    int32_t bt = (int32_t)tbias;
    CK(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
  }
  const int64_t ar = ta == HIPBLAS_OP_N ? m : k, ac = ta == HIPBLAS_OP_N ? k : m;
  const int64_t br = tb == HIPBLAS_OP_N ? k : n, bc = tb == HIPBLAS_OP_N ? n : k;
  CK(hipblasLtMatrixLayoutCreate(&g.a, HIP_R_16BF, ar, ac, lda));
  CK(hipblasLtMatrixLayoutCreate(&g.b, HIP_R_16BF, br, bc, ldb));
  CK(hipblasLtMatrixLayoutCreate(&g.c, tc, m, n, ldc));
  CK(hipblasLtMatrixLayoutCreate(&g.d, td, m, n, ldc));
  if (batch > 1) {
    int32_t bcount = batch;
    for (auto [l, s] : {std::pair{g.a, sa}, std::pair{g.b, sb}, std::pair{g.c, sc}, std::pair{g.d, sc}}) {
      CK(hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bcount, sizeof(bcount)));
      int64_t st = s;
      CK(hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &st, sizeof(st)));
    }
  }
  hipblasLtMatmulPreference_t pref;
  CK(hipblasLtMatmulPreferenceCreate(&pref));
  uint64_t w = ws;
  CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &w, sizeof(w)));
  auto st = hipblasLtMatmulAlgoGetHeuristic(h, g.desc, g.a, g.b, g.c, g.d, pref, 8, g.heur, &g.nheur);
  if ((int)st != 0 || g.nheur == 0) {
    printf("  no algorithm (status %d, %d results)\n", (int)st, g.nheur);
    return 2;
  }
  return 0;
}

static double run(hipblasLtHandle_t h, Gemm& g, int algo, const void* A, const void* B, const void* C, void* D, float beta, void* ws,
                  size_t wsz, hipStream_t st, int iters) {
  float alpha = 1.f;
  for (int i = 0; i < 3; ++i)
    if ((int)hipblasLtMatmul(h, g.desc, &alpha, A, g.a, B, g.b, &beta, C, g.c, D, g.d, &g.heur[algo].algo, ws, wsz, st) != 0) return -1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) hipblasLtMatmul(h, g.desc, &alpha, A, g.a, B, g.b, &beta, C, g.c, D, g.d, &g.heur[algo].algo, ws, wsz, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

__global__ void fill(__hip_bfloat16* p, size_t n, float scale, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
    p[i] = __float2bfloat16(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
  }
}
__global__ void colsum_check(const __hip_bfloat16* dy, int64_t rows, int n, const float* got, float* maxerr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0;
  for (int64_t r = 0; r < rows; ++r) s += (double)__bfloat162float(dy[r * n + c]);
  const float e = fabsf((float)s - got[c]) / (fabsf((float)s) + 1.f);
  atomicMax((int*)maxerr, __float_as_int(e));
}

int main(int argc, char** argv) {
  const int64_t L = argc > 1 ? atoll(argv[1]) : 204800;
  const int D = 512;
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t wsz = 256u << 20;
  void* ws;
  CK(hipMalloc(&ws, wsz));
  __hip_bfloat16 *x, *y3, *wo, *out, *xn, *dy, *wu;
  float *dw, *dbias, *dbias16, *dw16, *err;
  CK(hipMalloc(&x, L * D * 2));
  CK(hipMalloc(&y3, L * 3 * D * 2));
  CK(hipMalloc(&wo, 3 * D * D * 2));
  CK(hipMalloc(&out, L * D * 2));
  CK(hipMalloc(&xn, L * D * 2));
  CK(hipMalloc(&dy, L * 4 * D * 2));
  CK(hipMalloc(&wu, 4 * D * D * 2));
  CK(hipMalloc(&dw, (size_t)D * 4 * D * 4));
  CK(hipMalloc(&dw16, (size_t)16 * D * 4 * D * 4));
  CK(hipMalloc(&dbias, 4 * D * 4));
  CK(hipMalloc(&dbias16, 16 * 4 * D * 4));
  CK(hipMalloc(&err, 4));
  fill<<<4096, 256, 0, st>>>(x, L * D, 1.f, 1);
  fill<<<4096, 256, 0, st>>>(y3, L * 3 * D, 1.f, 2);
  fill<<<4096, 256, 0, st>>>(wo, 3 * D * D, 0.05f, 3);
  fill<<<4096, 256, 0, st>>>(xn, L * D, 1.f, 4);
  fill<<<4096, 256, 0, st>>>(dy, L * 4 * D, 1.f, 5);
  fill<<<4096, 256, 0, st>>>(wu, 4 * D * D, 0.05f, 6);
  CK(hipStreamSynchronize(st));

  // ---- 1. out (row-major L x D) = y3 (L x 3D) Wo (3D x D) + x: column-major D^T (D x L) = Wo^T-as-stored (D x 3D) * y3^T-as-stored (3D x L)
  {
    printf("output stage  D = Y Wo + X,  %lld x %d x %d\n", (long long)L, D, 3 * D);
    const double fl = 2.0 * L * D * 3 * D;
    for (int same = 0; same < 2; ++same) {
      Gemm g;
      if (make(h, g, HIPBLAS_OP_N, HIPBLAS_OP_N, D, L, 3 * D, D, 3 * D, D, HIP_R_16BF, HIP_R_16BF, 1, 0, 0, 0, HIPBLASLT_EPILOGUE_DEFAULT, nullptr,
               HIP_R_32F, wsz))
        continue;
      for (int a = 0; a < g.nheur && a < 4; ++a) {
        if (same) hipMemcpyAsync(out, x, L * D * 2, hipMemcpyDeviceToDevice, st);
        const double ms = run(h, g, a, wo, y3, same ? out : x, out, 1.f, ws, wsz, st, 20);
        printf("  %s algo %d: %.1f us  %.0f TFLOP/s\n", same ? "C == D (in place)" : "C != D          ", a, ms * 1e3, fl / ms / 1e9);
      }
    }
  }
  // ---- 2. dW (D x 4D row-major) = xn^T dy: column-major dW^T (4D x D) = dy-as-stored (4D x L) * op_T(xn-as-stored (D x L))
  {
    printf("uvqk weight gradient  dW = Xn^T dY (+ column sums of dY), fp32 out, %d x %d x %lld\n", D, 4 * D, (long long)L);
    const double fl = 2.0 * L * D * 4 * D;
    for (int ep = 0; ep < 2; ++ep) {
      for (int slabs : {1, 16}) {
        const int64_t slab = L / slabs;
        Gemm g;
        if (make(h, g, HIPBLAS_OP_N, HIPBLAS_OP_T, 4 * D, D, slab, 4 * D, D, 4 * D, HIP_R_32F, HIP_R_32F, slabs, slab * 4 * D, slab * D,
                 (int64_t)D * 4 * D, ep ? HIPBLASLT_EPILOGUE_BGRADA : HIPBLASLT_EPILOGUE_DEFAULT, slabs == 1 ? dbias : dbias16, HIP_R_32F, wsz)) {
          printf("  epilogue %s, %2d slab(s): not available\n", ep ? "BGRADA" : "none  ", slabs);
          continue;
        }
        for (int a = 0; a < g.nheur && a < 3; ++a) {
          hipMemsetAsync(slabs == 1 ? dbias : dbias16, 0, 16 * 4 * D * 4 / (slabs == 1 ? 16 : 1), st);
          const double ms = run(h, g, a, dy, xn, slabs == 1 ? dw : dw16, slabs == 1 ? dw : dw16, 0.f, ws, wsz, st, 10);
          printf("  epilogue %s, %2d slab(s), algo %d: %.1f us  %.0f TFLOP/s", ep ? "BGRADA" : "none  ", slabs, a, ms * 1e3, fl / ms / 1e9);
          if (ep && slabs == 1 && ms > 0) {
            hipMemsetAsync(err, 0, 4, st);
            colsum_check<<<(4 * D + 63) / 64, 64, 0, st>>>(dy, L, 4 * D, dbias, err);
            float e = 0;
            hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
            printf("   bias-gradient max rel err %.2e", e);
          }
          printf("\n");
        }
      }
    }
  }
  // ---- 3. uvqk forward with the bias epilogue, weight (in, out) as stored vs a K-contiguous copy
  {
    printf("uvqk forward  Y = Xn W + b, %lld x %d x %d\n", (long long)L, 4 * D, D);
    const double fl = 2.0 * L * D * 4 * D;
    __hip_bfloat16* bias16;
    CK(hipMalloc(&bias16, 4 * D * 2));
    fill<<<8, 256, 0, st>>>(bias16, 4 * D, 1.f, 9);
    for (int kc = 0; kc < 2; ++kc) {
      Gemm g;   // D^T (4D x L) = W-as-stored: (4D x D, ld 4D) op N  |  K-contiguous copy: stored (D x 4D col-major, ld D) op T
      if (make(h, g, kc ? HIPBLAS_OP_T : HIPBLAS_OP_N, HIPBLAS_OP_N, 4 * D, L, D, kc ? D : 4 * D, D, 4 * D, HIP_R_16BF, HIP_R_16BF, 1, 0, 0, 0,
               HIPBLASLT_EPILOGUE_BIAS, bias16, HIP_R_16BF, wsz))
        continue;
      for (int a = 0; a < g.nheur && a < 3; ++a) {
        const double ms = run(h, g, a, wu, xn, dy, dy, 0.f, ws, wsz, st, 20);
        printf("  weight %s algo %d: %.1f us  %.0f TFLOP/s\n", kc ? "K-contiguous" : "(in, out)   ", a, ms * 1e3, fl / ms / 1e9);
      }
    }
  }
  return 0;
}
