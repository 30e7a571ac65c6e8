// bf16 instantiations of the HSTU attention kernels (one TU per dtype: parallel builds).
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_bf16(const HstuAttnParams& p, hipStream_t st) {
  if (attn_solo_applicable(p, false)) return launch_attn_fwd_solo_bf16(p, st);
  if (attn_solo_bias_applicable(p, false)) return launch_attn_fwd_solo_bias_bf16(p, st);
  return p.pos_w ? launch_attn_fwd_bias_bf16(p, st) : launch_fwd_dtype<bf16_t>(p, st);
}
int launch_attn_bwd_bf16(const HstuAttnBwdParams& p, hipStream_t st) {
  if (attn_solo_applicable(p.fwd, true)) return launch_attn_bwd_solo_bf16(p, st);
  if (attn_solo_bias_applicable(p.fwd, true)) return launch_attn_bwd_solo_bias_bf16(p, st);
  if (attn_bwd_fold_applicable(p)) return launch_attn_bwd_fold_bf16(p, st);
  if (attn_bwd_long_applicable(p)) return launch_attn_bwd_long_bf16(p, st);
  return p.fwd.pos_w ? launch_attn_bwd_bias_bf16(p, st) : launch_bwd_dtype<bf16_t>(p, st);
}
int attn_bwd_tiles_bf16(int dqk, int dv, int n, int extra_lds) { return bwd_tiles_dtype<bf16_t>(dqk, dv, n, extra_lds); }
}  // namespace hstu
#ifdef HSTU_TRACE
// trace builds only (tools/trace_build.sh): the trace pointer lives in this TU, next to the kernels it instruments
extern "C" int hstu_trace_set_fwd(unsigned long long* ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(hstu::g_hstu_trace_fwd), &ptr, sizeof(ptr));
}
#endif
