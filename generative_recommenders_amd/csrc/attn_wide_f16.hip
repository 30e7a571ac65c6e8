// fp16 instantiation of the wide backward schedule.
#include "attn_wide.cuh"
namespace hstu {
int launch_attn_bwd_wide_f16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_wide_dtype<f16_t>(p, st); }
}  // namespace hstu
