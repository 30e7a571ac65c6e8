// bf16 instantiations of the short-sequence kernels.
#include "attn_solo.cuh"
namespace hstu {
int launch_attn_fwd_solo_bf16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_solo<bf16_t>(p, st); }
int launch_attn_bwd_solo_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_solo<bf16_t>(p, st); }
int launch_attn_fwd_solo_bias_bf16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_solo_bias<bf16_t>(p, st); }
int launch_attn_bwd_solo_bias_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_solo_bias<bf16_t>(p, st); }
}  // namespace hstu
