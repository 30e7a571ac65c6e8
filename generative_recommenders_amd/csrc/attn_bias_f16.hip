// f16 instantiations of the research-path (relative position / time bias) attention kernels.
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_bias_f16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_bias_dtype<f16_t>(p, st); }
int launch_attn_bwd_bias_f16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_bias_dtype<f16_t>(p, st); }
}  // namespace hstu
