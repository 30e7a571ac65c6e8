// bf16 instantiations of the folded backward schedule.
#include "attn_fold.cuh"
namespace hstu {
int launch_attn_bwd_fold_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_fold_dtype<bf16_t>(p, st); }
}  // namespace hstu
