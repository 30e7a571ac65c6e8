// bf16 instantiations of the research-path (relative position / time bias) attention kernels.
#include "attn_launch.cuh"
namespace hstu {
int launch_attn_fwd_bias_bf16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_bias_dtype<bf16_t>(p, st); }
int launch_attn_bwd_bias_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_bias_dtype<bf16_t>(p, st); }
}  // namespace hstu
