// f16 instantiation of the sixteen-wave backward.
#include "attn_w16.cuh"
namespace hstu {
int launch_attn_bwd_w16_f16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_w16_dtype<f16_t>(p, st); }
}  // namespace hstu
