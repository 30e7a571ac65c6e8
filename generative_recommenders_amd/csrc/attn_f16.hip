// f16 instantiations of the HSTU attention kernels (one TU per dtype: parallel builds).
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_f16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_dtype<f16_t>(p, st); }
int launch_attn_bwd_f16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_dtype<f16_t>(p, st); }
int attn_bwd_tiles_f16(int dqk, int dv, int n) { return bwd_tiles_dtype<f16_t>(dqk, dv, n); }
}  // namespace hstu
