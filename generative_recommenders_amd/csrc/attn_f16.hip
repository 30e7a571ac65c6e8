// f16 instantiations of the HSTU attention kernels (one TU per dtype: parallel builds).
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_f16(const HstuAttnParams& p, hipStream_t st) {
  if (attn_solo_applicable(p, false)) return launch_attn_fwd_solo_f16(p, st);
  if (attn_solo_bias_applicable(p, false)) return launch_attn_fwd_solo_bias_f16(p, st);
  return p.pos_w ? launch_attn_fwd_bias_f16(p, st) : launch_fwd_dtype<f16_t>(p, st);
}
int launch_attn_bwd_f16(const HstuAttnBwdParams& p, hipStream_t st) {
  if (attn_solo_applicable(p.fwd, true)) return launch_attn_bwd_solo_f16(p, st);
  if (attn_solo_bias_applicable(p.fwd, true)) return launch_attn_bwd_solo_bias_f16(p, st);
  if (attn_bwd_fold_applicable(p)) return launch_attn_bwd_fold_f16(p, st);
  if (attn_bwd_long_applicable(p)) return launch_attn_bwd_long_f16(p, st);
  return p.fwd.pos_w ? launch_attn_bwd_bias_f16(p, st) : launch_bwd_dtype<f16_t>(p, st);
}
int attn_bwd_tiles_f16(int dqk, int dv, int n, int extra_lds) { return bwd_tiles_dtype<f16_t>(dqk, dv, n, extra_lds); }
}  // namespace hstu
