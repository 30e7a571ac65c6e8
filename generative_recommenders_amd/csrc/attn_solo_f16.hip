// fp16 instantiations of the short-sequence kernels.
#include "attn_solo.cuh"
namespace hstu {
int launch_attn_fwd_solo_f16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_solo<f16_t>(p, st); }
int launch_attn_bwd_solo_f16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_solo<f16_t>(p, st); }
int launch_attn_fwd_solo_bias_f16(const HstuAttnParams& p, hipStream_t st) { return launch_fwd_solo_bias<f16_t>(p, st); }
int launch_attn_bwd_solo_bias_f16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_solo_bias<f16_t>(p, st); }
}  // namespace hstu
