// bf16 instantiation of the wide backward schedule.
#include "attn_wide.cuh"
namespace hstu {
int launch_attn_bwd_wide_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_wide_dtype<bf16_t>(p, st); }
}  // namespace hstu
