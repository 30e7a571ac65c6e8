// bf16 instantiations of the long-sequence backward (hstu_attn_bwd_long.cuh: dK / dV kernel + dQ kernel).
#include "capi_internal.h"
#include "hstu_attn_bwd_long.cuh"
namespace hstu {
int launch_attn_bwd_long_bf16(const HstuAttnBwdParams& p, hipStream_t st) { return launch_bwd_long_dtype<bf16_t>(p, st); }
}  // namespace hstu
