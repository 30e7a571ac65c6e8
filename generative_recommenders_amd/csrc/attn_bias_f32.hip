// f32 instantiations of the research-path (relative position / time bias) attention kernels.
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_bias_f32(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_bias_dtype<float>(p, st); }
int launch_attn_bwd_bias_f32(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_bias_dtype<float>(p, st); }
}  // namespace hstu
