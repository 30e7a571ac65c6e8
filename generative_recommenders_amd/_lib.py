"""ctypes binding of libhstu_hip.so (C ABI declared in include/hstu_hip.h).

The library is the product: there is NO fallback.  If it is missing, stale or the
wrong ABI version, importing the ops raises immediately (``HstuLibraryError``).
PyTorch is used for device memory and streams only; tensors cross the boundary as
raw device pointers + strides.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime our .so binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# HSTU_HIP_LIBRARY overrides the in-tree build (A/B measurements of kernel variants, packaged installs)
LIB_PATH = os.environ.get("HSTU_HIP_LIBRARY") or os.path.join(_HERE, "libhstu_hip.so")
ABI_VERSION = 13

HSTU_DTYPE_BF16, HSTU_DTYPE_F16, HSTU_DTYPE_F32 = 0, 1, 2
HSTU_INDEX_I32, HSTU_INDEX_I64 = 0, 1


class HstuLibraryError(RuntimeError):
    pass


class HstuAttnParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
        ("seq_offsets", C.c_void_p), ("num_targets", C.c_void_p),
        ("q_row_stride", C.c_int64), ("q_head_stride", C.c_int64),
        ("k_row_stride", C.c_int64), ("k_head_stride", C.c_int64),
        ("v_row_stride", C.c_int64), ("v_head_stride", C.c_int64),
        ("o_row_stride", C.c_int64), ("o_head_stride", C.c_int64),
        ("batch", C.c_int32), ("heads", C.c_int32), ("dqk", C.c_int32), ("dv", C.c_int32),
        ("max_seq_len", C.c_int32), ("delta_q", C.c_int32),
        ("alpha", C.c_float), ("scale", C.c_float),
        ("max_attn_len", C.c_int32), ("contextual_seq_len", C.c_int32), ("min_full_attn_seq_len", C.c_int32),
        ("dtype", C.c_int32), ("offsets_dtype", C.c_int32), ("targets_dtype", C.c_int32),
        ("pos_w", C.c_void_p), ("ts_w", C.c_void_p), ("timestamps", C.c_void_p),
        ("ts_row_stride", C.c_int64), ("num_buckets", C.c_int32), ("bucket_div", C.c_float),
        ("attn_scale", C.c_void_p), ("user_order", C.c_void_p),
    ]


class HstuAttnBwdParams(C.Structure):
    _fields_ = [
        ("fwd", HstuAttnParams),
        ("dout", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
        ("do_row_stride", C.c_int64), ("do_head_stride", C.c_int64),
        ("dq_row_stride", C.c_int64), ("dq_head_stride", C.c_int64),
        ("dk_row_stride", C.c_int64), ("dk_head_stride", C.c_int64),
        ("dv_row_stride", C.c_int64), ("dv_head_stride", C.c_int64),
        ("workspace", C.c_void_p), ("total_rows", C.c_int64),
        ("dpos_w", C.c_void_p), ("dts_w", C.c_void_p),
        ("deterministic", C.c_int),
    ]


class HstuCastItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("numel", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("transpose", C.c_int32)]


CAST_MAX_ITEMS = 8

_vp, _i32, _i64, _f32, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_int
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/hstu_hip.h exactly (checked by tests/test_abi.py)
SIGNATURES = {
    "hstu_abi_version": (_int, []),
    "hstu_last_error": (C.c_char_p, []),
    "hstu_attn_fwd": (_int, [C.POINTER(HstuAttnParams), _vp]),
    "hstu_attn_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(HstuAttnBwdParams)]),
    "hstu_attn_bwd": (_int, [C.POINTER(HstuAttnBwdParams), _vp]),
    "hstu_attn_fwd_kernel_name": (_int, [C.POINTER(HstuAttnParams), C.c_char_p, C.c_size_t]),
    "hstu_attn_bwd_kernel_name": (_int, [C.POINTER(HstuAttnBwdParams), C.c_char_p, C.c_size_t]),
    "hstu_complete_cumsum": (_int, [_vp, _vp, _i64, _int, _vp]),
    "hstu_concat_2d_jagged": (_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _int, _vp]),
    "hstu_split_2d_jagged": (_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _int, _vp]),
    "hstu_jagged_to_padded_dense": (_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _int, _vp]),
    "hstu_dense_to_jagged": (_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _int, _vp]),
    "hstu_expand_1d_jagged_to_dense": (_int, [_vp, _vp, _vp, _i32, _i32, _i32, _int, _vp]),
    "hstu_concat_1d_jagged_jagged": (_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _int, _vp]),
    "hstu_layer_norm_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _int, _vp]),
    "hstu_ln_linear_fwd_supported": (_int, [_i64, _i32, _i32, _int]),
    "hstu_ln_linear_fwd": (_int, [_vp, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _int, _vp]),
    "hstu_linear_k512_supported": (_int, [_i64, _i32, _i32, _int]),
    "hstu_linear_k512": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _int, _vp]),
    "hstu_addmm_residual_supported": (_int, []),
    "hstu_addmm_residual": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _int, _vp, C.c_size_t, _vp]),
    "hstu_layer_norm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _int, _vp]),
    "hstu_swish_layer_norm_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _int, _vp]),
    "hstu_swish_layer_norm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _int, _vp]),
    "hstu_norm_bwd_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "hstu_norm_mul_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _int, _int, _int, _vp]),
    "hstu_norm_mul_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _int, _int, _int, _vp]),
    "hstu_norm_mul_dropout_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _int, _int, _f32, C.c_uint64, _int, _vp]),
    "hstu_layer_norm_bwd_residual": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _int, _vp]),
    "hstu_norm_mul_silu_fwd": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _int, _int, _f32, C.c_uint64, _int, _vp]),
    "hstu_norm_mul_silu_bwd": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _int, _int, _f32, C.c_uint64, _int, _vp]),
    "hstu_norm_mul_dropout_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _int, _int, _f32, C.c_uint64, _int, _vp]),
    "hstu_silu_fwd": (_int, [_vp, _vp, _i64, _i32, _i64, _i64, _int, _vp]),
    "hstu_silu_bwd": (_int, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _i64, _int, _vp]),
    "hstu_add_ts_pos_emb_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _int, _int, _vp]),
    "hstu_l2_norm_fwd": (_int, [_vp, _vp, _i64, _i32, _f32, _int, _vp]),
    "hstu_l2_norm_bwd": (_int, [_vp, _vp, _vp, _i64, _i32, _f32, _int, _vp]),
    "hstu_embedding_grad_segment_sum": (_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _int, _vp]),
    "hstu_embedding_grad_workspace_bytes": (_int, [_i64, _i32, _vp]),
    "hstu_embedding_grad": (_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _int, _vp]),
    "hstu_jagged_write_tail": (_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _int, _vp]),
    "hstu_cast_params": (_int, [C.POINTER(HstuCastItem), _i32, _int, _vp]),
    "hstu_column_sum_workspace_bytes": (C.c_size_t, [_i64, _i32]),
    "hstu_column_sum": (_int, [_vp, _i64, _i64, _i32, _vp, _vp, _int, _vp]),
    "hstu_calib_mfma_stream": (_int, [_i32, _vp, C.POINTER(C.c_double), _vp]),
    "hstu_calib_read_stream": (_int, [_vp, C.c_size_t, _vp, _vp]),
    "hstu_sampled_softmax_fwd": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _f32, _i32,
                                        _i32, _f32, _vp, _vp, _int, _vp]),
    "hstu_sampled_softmax_bwd": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _f32, _i32,
                                        _i32, _f32, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _int, _vp]),
}

_lib: Optional[C.CDLL] = None


def build(verbose: bool = False, torch_ops_optional: bool = False) -> str:
    """Compile libhstu_hip.so and libhstu_torch_ops.so in-tree for gfx950 (hipcc cross-compiles without a GPU).  A failure of
    either is an error; ``torch_ops_optional=True`` (C-ABI-only consumers) downgrades the second one to a warning."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(os.cpu_count() or 4)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise HstuLibraryError("building libhstu_hip.so failed (see output above)")
    try:
        build_torch_ops(verbose)
    except HstuLibraryError as e:   # the C-ABI library is usable without the torch.ops.hstu registration
        if not torch_ops_optional:
            raise
        import warnings
        warnings.warn(f"libhstu_hip.so built, but {e}; torch.ops.hstu.* will not be available")
    return LIB_PATH


def build_torch_ops(verbose: bool = False) -> str:
    """Compile libhstu_torch_ops.so: host-side C++ (no device code) that registers the ``hstu::`` torch.library schemas
    with CUDA + Meta kernels on top of the C ABI (csrc/torch_ops/hstu_torch_ops.cpp).  Skipped when up to date."""
    src = os.path.join(_HERE, "csrc", "torch_ops", "hstu_torch_ops.cpp")
    out = os.path.join(_HERE, "libhstu_torch_ops.so")
    # rebuilt when the source, the header, the core library it links (HstuAttnParams is passed by pointer: a stale pair
    # would read garbage fields) or the torch it was compiled against changes
    deps = [src, os.path.join(_HERE, "..", "include", "hstu_hip.h"), os.path.join(_HERE, "libhstu_hip.so")]
    stamp = out + ".torch_version"
    stamped = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if (os.path.exists(out) and stamped == torch.__version__
            and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps if os.path.exists(d))):
        return out
    tdir = os.path.dirname(torch.__file__)
    # (the two -D switches are what PyTorch-ROCm's OWN headers need to be compiled by a plain host compiler -- the flags
    # torch.utils.cpp_extension passes on ROCm; nothing in this repository is conditional on them)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           f"-I{tdir}/include", f"-I{tdir}/include/torch/csrc/api/include", "-I/opt/rocm/include", src, "-o", out,
           f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", f"-L{_HERE}", "-lhstu_hip",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tdir}/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise HstuLibraryError("building libhstu_torch_ops.so failed (see output above)")
    with open(stamp, "w") as f:
        f.write(torch.__version__)
    return out


def lib() -> C.CDLL:
    """Load (once) and return the library; raise loudly if it is not usable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HstuLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'from generative_recommenders_amd import _lib; _lib.build()'` "
            "(= __graft_entry__.build(): libhstu_hip.so AND libhstu_torch_ops.so).  There is no CPU / PyTorch fallback "
            "for the HSTU ops."
        )
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise HstuLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise HstuLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    got = handle.hstu_abi_version()
    if got != ABI_VERSION:
        raise HstuLibraryError(f"{LIB_PATH} has ABI version {got}, expected {ABI_VERSION}; rebuild it")
    _lib = handle
    return _lib


def check(code: int) -> None:
    """Turn a negative HSTU_E* return code into a RuntimeError carrying the library's message
    (the reference surfaces TORCH_CHECK failures as RuntimeError too)."""
    if code != 0:
        msg = lib().hstu_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libhstu_hip error {code}: {msg}")


def torch_dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.bfloat16:
        return HSTU_DTYPE_BF16
    if dtype == torch.float16:
        return HSTU_DTYPE_F16
    if dtype == torch.float32:
        return HSTU_DTYPE_F32
    raise RuntimeError(f"HSTU HIP ops support bf16 / fp16 / fp32 tensors, got {dtype}")


def index_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return HSTU_INDEX_I64
    if t.dtype == torch.int32:
        return HSTU_INDEX_I32
    raise RuntimeError(f"offsets / num_targets must be int32 or int64, got {t.dtype}")


def current_stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on the GPU: the HSTU ops are HIP kernels with no CPU fallback "
            f"(got device {t.device})"
        )
