// bf16 instantiation of the sixteen-wave backward.
#include "attn_w16.cuh"
namespace hstu {
int launch_attn_bwd_w16_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_w16_dtype<bf16_t>(p, st); }
}  // namespace hstu
