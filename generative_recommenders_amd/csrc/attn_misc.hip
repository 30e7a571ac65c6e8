// dtype-independent helpers of the attention ABI.
#include "capi_internal.h"
namespace hstu {
int attn_bwd_tiles_bf16(int, int, int);
int attn_bwd_tiles_f16(int, int, int);
int attn_bwd_tiles_f32(int, int, int);

int attn_bwd_tiles_per_block(int dtype, int dqk, int dv, int max_seq_len) {
  switch (dtype) {
    case HSTU_DTYPE_BF16: return attn_bwd_tiles_bf16(dqk, dv, max_seq_len);
    case HSTU_DTYPE_F16: return attn_bwd_tiles_f16(dqk, dv, max_seq_len);
    default: return attn_bwd_tiles_f32(dqk, dv, max_seq_len);
  }
}

size_t attn_bwd_workspace_bytes(const HstuAttnBwdParams& bp) {
  const HstuAttnParams& p = bp.fwd;
  const int nw = attn_bwd_tiles_per_block(p.dtype, p.dqk, p.dv, p.max_seq_len);
  if (nw <= 0) return 0;
  const int nkb = (p.max_seq_len + 32 * nw - 1) / (32 * nw);
  if (nkb <= 1) return 0;
  return (size_t)bp.total_rows * p.heads * p.dqk * sizeof(float);
}
}  // namespace hstu
