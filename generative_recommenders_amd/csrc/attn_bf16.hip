// bf16 instantiations of the HSTU attention kernels (one TU per dtype: parallel builds).
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_bf16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_dtype<bf16_t>(p, st); }
int launch_attn_bwd_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_dtype<bf16_t>(p, st); }
int attn_bwd_tiles_bf16(int dqk, int dv, int n) { return bwd_tiles_dtype<bf16_t>(dqk, dv, n); }
}  // namespace hstu

#ifdef HSTU_TRACE
extern "C" int hstu_trace_set_fwd(void* ptr) {   // debug builds only (tools/trace_build.sh)
  unsigned long long* p = (unsigned long long*)ptr;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(hstu::g_hstu_trace_fwd), &p, sizeof(p));
}
#endif
