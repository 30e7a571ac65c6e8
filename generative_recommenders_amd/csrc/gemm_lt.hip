// d = a . b + c through hipBLASLt with C and D as DIFFERENT buffers (ABI v13: hstu_addmm_residual).
//
// The output stage of an STU layer is out = x + y . W_o (ops/hstu_compute.py:92-136, `torch.addmm(x, y, output_weight)`;
// ops/triton/triton_addmm.py:185-340 is the reference's own kernel for it).  Through torch that is TWO passes: ATen copies x into the
// result and runs the GEMM in place with beta = 1 (hipBLASLt takes C and D separately, torch's wrapper does not) -- 3 x 40 us of
// copy kernel per step at the bench's layer shape.  This file is the one place the library calls a vendor GEMM itself: a plain
// library GEMM (no fusion to write by hand), one launch, no copy.  hipBLASLt is looked up at run time (dlopen: inside a PyTorch
// process that is the copy PyTorch has loaded; elsewhere /opt/rocm's), nothing links against it.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "capi_internal.h"

namespace hstu {
namespace {

struct LtApi {
  decltype(&hipblasLtCreate) create = nullptr;
  decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
  decltype(&hipblasLtMatmulDescDestroy) desc_destroy = nullptr;
  decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
  decltype(&hipblasLtMatrixLayoutDestroy) layout_destroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
  decltype(&hipblasLtMatmulPreferenceDestroy) pref_destroy = nullptr;
  decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
  decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
  decltype(&hipblasLtMatmul) matmul = nullptr;
  bool ok = false;
};

const LtApi& lt_api() {
  static const LtApi api = [] {
    LtApi a;
    void* h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return a;
#define LT_SYM(field, name) a.field = (decltype(a.field))dlsym(h, #name)
    LT_SYM(create, hipblasLtCreate);
    LT_SYM(desc_create, hipblasLtMatmulDescCreate);
    LT_SYM(desc_destroy, hipblasLtMatmulDescDestroy);
    LT_SYM(layout_create, hipblasLtMatrixLayoutCreate);
    LT_SYM(layout_destroy, hipblasLtMatrixLayoutDestroy);
    LT_SYM(pref_create, hipblasLtMatmulPreferenceCreate);
    LT_SYM(pref_destroy, hipblasLtMatmulPreferenceDestroy);
    LT_SYM(pref_set, hipblasLtMatmulPreferenceSetAttribute);
    LT_SYM(heuristic, hipblasLtMatmulAlgoGetHeuristic);
    LT_SYM(matmul, hipblasLtMatmul);
#undef LT_SYM
    a.ok = a.create && a.desc_create && a.desc_destroy && a.layout_create && a.layout_destroy && a.pref_create && a.pref_destroy &&
           a.pref_set && a.heuristic && a.matmul;
    return a;
  }();
  return api;
}

// one handle per device, one planned problem per (device, shape, leading dims, dtype, workspace): descriptors + the heuristic's first choice
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr, ld = nullptr;
  // (the result array is over-sized: the structure may grow between library versions)
  alignas(16) unsigned char res_bytes[256] = {0};
  hipblasLtMatmulHeuristicResult_t* res() { return reinterpret_cast<hipblasLtMatmulHeuristicResult_t*>(res_bytes); }
  bool ok = false;
};
using Key = std::tuple<int, int64_t, int, int, int64_t, int64_t, int64_t, int64_t, int, size_t>;
std::mutex g_mu;
std::map<int, hipblasLtHandle_t> g_handles;
std::map<Key, Plan> g_plans;

}  // namespace

bool addmm_lt_available() { return lt_api().ok; }

// row-major d (m, n) = a (m, k) . b (k, n) + c (m, n).  hipBLASLt is column-major: d^T (n x m) = b^T (n x k) . a^T (k x m) + c^T,
// and a row-major (r, s) array IS the column-major (s x r) one -- no transposes, A := b, B := a.
int addmm_lt(const void* c, int64_t ldc, const void* a, int64_t lda, const void* b, int64_t ldb, void* d, int64_t ldd, int64_t m, int n,
             int k, int dtype, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const LtApi& api = lt_api();
  if (!api.ok) return set_error(HSTU_EUNSUPPORTED, "hstu_addmm_residual: libhipblaslt.so.1 could not be loaded");
  int dev = 0;
  (void)hipGetDevice(&dev);
  const hipDataType ty = dtype == HSTU_DTYPE_BF16 ? HIP_R_16BF : HIP_R_16F;
  std::lock_guard<std::mutex> lock(g_mu);
  hipblasLtHandle_t& handle = g_handles[dev];
  if (!handle && api.create(&handle) != HIPBLAS_STATUS_SUCCESS) {
    handle = nullptr;
    return set_error(HSTU_ELAUNCH, "hstu_addmm_residual: hipblasLtCreate failed");
  }
  const Key key{dev, m, n, k, lda, ldb, ldc, ldd, dtype, workspace_bytes};
  // (jagged batches: the row count m changes from call to call, and so does the planned problem -- the cache is bounded: emptied, with
  // its descriptors destroyed, when it has grown to kMaxPlans; planning a problem is what torch's own wrapper does on EVERY call)
  constexpr size_t kMaxPlans = 256;
  if (g_plans.size() >= kMaxPlans && g_plans.find(key) == g_plans.end()) {
    for (auto& kv : g_plans) {
      Plan& q = kv.second;
      if (q.desc) api.desc_destroy(q.desc);
      for (hipblasLtMatrixLayout_t l : {q.la, q.lb, q.lc, q.ld})
        if (l) api.layout_destroy(l);
    }
    g_plans.clear();
  }
  Plan& pl = g_plans[key];
  if (!pl.ok) {
    if (api.desc_create(&pl.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS ||
        api.layout_create(&pl.la, ty, (uint64_t)n, (uint64_t)k, ldb) != HIPBLAS_STATUS_SUCCESS ||
        api.layout_create(&pl.lb, ty, (uint64_t)k, (uint64_t)m, lda) != HIPBLAS_STATUS_SUCCESS ||
        api.layout_create(&pl.lc, ty, (uint64_t)n, (uint64_t)m, ldc) != HIPBLAS_STATUS_SUCCESS ||
        api.layout_create(&pl.ld, ty, (uint64_t)n, (uint64_t)m, ldd) != HIPBLAS_STATUS_SUCCESS) {
      if (pl.desc) api.desc_destroy(pl.desc);
      for (hipblasLtMatrixLayout_t l : {pl.la, pl.lb, pl.lc, pl.ld})
        if (l) api.layout_destroy(l);
      g_plans.erase(key);
      return set_error(HSTU_ELAUNCH, "hstu_addmm_residual: hipBLASLt descriptors could not be created");
    }
    hipblasLtMatmulPreference_t pref = nullptr;
    int found = 0;
    hipblasStatus_t hs = api.pref_create(&pref);
    if (hs == HIPBLAS_STATUS_SUCCESS) {
      uint64_t ws = workspace_bytes;
      hs = api.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    }
    if (hs == HIPBLAS_STATUS_SUCCESS) hs = api.heuristic(handle, pl.desc, pl.la, pl.lb, pl.lc, pl.ld, pref, 1, pl.res(), &found);
    if (pref) api.pref_destroy(pref);
    if (hs != HIPBLAS_STATUS_SUCCESS || found < 1 || pl.res()->state != HIPBLAS_STATUS_SUCCESS) {
      api.desc_destroy(pl.desc);
      for (hipblasLtMatrixLayout_t l : {pl.la, pl.lb, pl.lc, pl.ld}) api.layout_destroy(l);
      g_plans.erase(key);
      return set_error(HSTU_EUNSUPPORTED, "hstu_addmm_residual: hipBLASLt has no algorithm for (%lld, %d, %d) (status %d)", (long long)m, n, k, (int)hs);
    }
    pl.ok = true;
  }
  const float alpha = 1.f, beta = 1.f;
  const hipblasStatus_t hs = api.matmul(handle, pl.desc, &alpha, b, pl.la, a, pl.lb, &beta, c, pl.lc, d, pl.ld, &pl.res()->algo, workspace,
                                        workspace_bytes, st);
  if (hs != HIPBLAS_STATUS_SUCCESS) return set_error(HSTU_ELAUNCH, "hstu_addmm_residual: hipblasLtMatmul failed (status %d)", (int)hs);
  return HSTU_OK;
}

}  // namespace hstu

extern "C" {

int hstu_addmm_residual_supported(void) { return hstu::addmm_lt_available() ? 1 : 0; }

int hstu_addmm_residual(const void* c, int64_t ldc, const void* a, int64_t lda, const void* b, int64_t ldb, void* d, int64_t ldd, int64_t m,
                        int32_t n, int32_t k, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace hstu;
  if (dtype != HSTU_DTYPE_BF16 && dtype != HSTU_DTYPE_F16) return set_error(HSTU_EUNSUPPORTED, "hstu_addmm_residual: 16-bit dtypes only");
  if (m < 0 || n <= 0 || k <= 0) return set_error(HSTU_EINVAL, "hstu_addmm_residual: bad shape (%lld, %d, %d)", (long long)m, n, k);
  if (m == 0) return HSTU_OK;
  if (!a || !b || !c || !d) return set_error(HSTU_EINVAL, "hstu_addmm_residual: NULL operand");
  if (lda < k || ldb < n || ldc < n || ldd < n) return set_error(HSTU_EINVAL, "hstu_addmm_residual: a leading dimension is shorter than its row");
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) return set_error(HSTU_EINVAL, "hstu_addmm_residual: operands must be 16-byte aligned");
  if (workspace_bytes && !workspace) return set_error(HSTU_EINVAL, "hstu_addmm_residual: workspace_bytes without a workspace");
  return addmm_lt(c, ldc, a, lda, b, ldb, d, ldd, m, n, k, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
