#!/bin/bash
# Debug build of the attention kernels with the s_memtime trace enabled -> tests/probe/libhstu_trace.so
set -e
cd "$(dirname "$0")/../generative_recommenders_amd/csrc"
mkdir -p build_trace
for f in capi attn_misc attn_bf16 attn_bias_bf16 attn_fold_bf16 attn_solo_bf16 jagged_ops norm_ops ln_linear aux_ops position_ops embedding_grad loss_ops; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DHSTU_TRACE $HSTU_EXTRA -I. -I../../include -c $f.hip -o build_trace/$f.o &
done
wait
cat > build_trace/stubs.cpp <<'EOS'
#include "../capi_internal.h"
namespace hstu {
int launch_attn_fwd_f16(const HstuAttnParams&, hipStream_t) { return -2; }
int launch_attn_fwd_f32(const HstuAttnParams&, hipStream_t) { return -2; }
int launch_attn_bwd_f16(const HstuAttnBwdParams&, hipStream_t) { return -2; }
int launch_attn_bwd_f32(const HstuAttnBwdParams&, hipStream_t) { return -2; }
int attn_bwd_tiles_f16(int, int, int, int) { return 0; }
int attn_bwd_tiles_f32(int, int, int, int) { return 0; }
int launch_attn_fwd_bias_f16(const HstuAttnParams&, hipStream_t) { return -2; }
int launch_attn_fwd_bias_f32(const HstuAttnParams&, hipStream_t) { return -2; }
int launch_attn_bwd_bias_f16(const HstuAttnBwdParams&, hipStream_t) { return -2; }
int launch_attn_bwd_bias_f32(const HstuAttnBwdParams&, hipStream_t) { return -2; }
int launch_attn_bwd_fold_f16(const HstuAttnBwdParams&, hipStream_t) { return -2; }
int launch_attn_fwd_solo_f16(const HstuAttnParams&, hipStream_t) { return -2; }
int launch_attn_bwd_solo_f16(const HstuAttnBwdParams&, hipStream_t) { return -2; }
}
EOS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -I../../include -c build_trace/stubs.cpp -o build_trace/stubs.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_trace/*.o -o ../../tests/probe/libhstu_trace.so
echo built tests/probe/libhstu_trace.so
