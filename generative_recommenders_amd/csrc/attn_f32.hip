// f32 instantiations of the HSTU attention kernels (one TU per dtype: parallel builds).
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_f32(const HstuAttnParams& p, hipStream_t st) {
  return p.pos_w ? launch_attn_fwd_bias_f32(p, st) : launch_fwd_dtype<float>(p, st);
}
int launch_attn_bwd_f32(const HstuAttnBwdParams& p, hipStream_t st) {
  return p.fwd.pos_w ? launch_attn_bwd_bias_f32(p, st) : launch_bwd_dtype<float>(p, st);
}
int attn_bwd_tiles_f32(int dqk, int dv, int n, int extra_lds) { return bwd_tiles_dtype<float>(dqk, dv, n, extra_lds); }
}  // namespace hstu
