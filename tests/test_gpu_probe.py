"""Hardware-assumption probes (GPU): the MFMA fragment layouts and the LDS transpose-read
semantics that csrc/hstu_common.cuh documents.  When an attention parity test fails these
say whether a layout assumption or the kernel logic is at fault; on failure they dump the
observed mapping to gpurun_out/."""

import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "probe", "libhstu_probe.so")
OUT = os.path.join(ROOT, "gpurun_out")


def _probe():
    if not os.path.exists(PROBE):
        pytest.skip("probe library not built")
    return C.CDLL(PROBE)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dump(name, arr):
    os.makedirs(OUT, exist_ok=True)
    np.save(os.path.join(OUT, name), arr)


def test_runtime_identity():
    props = torch.cuda.get_device_properties(0)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "device.txt"), "w") as f:
        f.write(f"{props}\n gcnArch={getattr(props, 'gcnArchName', '?')}\n")
    assert "gfx950" in getattr(props, "gcnArchName", "gfx950")


def test_mfma_bf16_32x32x16_layout():
    lib = _probe()
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-4, 5, (32, 16), generator=g).float()
    B = torch.randint(-4, 5, (16, 32), generator=g).float()
    Ab = A.to(torch.bfloat16).view(torch.int16).cuda()
    Bb = B.to(torch.bfloat16).view(torch.int16).cuda()
    out = torch.zeros(64 * 16, device="cuda")
    assert lib.probe_run_mfma_bf16(C.c_void_p(Ab.data_ptr()), C.c_void_p(Bb.data_ptr()), C.c_void_p(out.data_ptr()), _stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(64, 16).numpy()
    ref = (A @ B).numpy()
    exp = np.zeros((64, 16), dtype=np.float32)
    for l in range(64):
        for r in range(16):
            exp[l, r] = ref[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    if not np.array_equal(got, exp):
        _dump("probe_mfma_bf16_got.npy", got)
        _dump("probe_mfma_bf16_ref.npy", ref)
    assert np.array_equal(got, exp)


def test_mfma_bf16_16x16x32_layout():
    """dQ GEMM of the folded backward: A[m = l&15][k = 8 (l>>4) + j], B[k][n = l&15], C[m = 4 (l>>4) + r][n = l&15]."""
    lib = _probe()
    g = torch.Generator().manual_seed(2)
    A = torch.randint(-4, 5, (16, 32), generator=g).float()
    B = torch.randint(-4, 5, (32, 16), generator=g).float()
    Ab = A.to(torch.bfloat16).view(torch.int16).cuda()
    Bb = B.to(torch.bfloat16).view(torch.int16).cuda()
    out = torch.zeros(64 * 4, device="cuda")
    assert lib.probe_run_mfma16_bf16(C.c_void_p(Ab.data_ptr()), C.c_void_p(Bb.data_ptr()), C.c_void_p(out.data_ptr()), _stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4).numpy()
    ref = (A @ B).numpy()
    exp = np.zeros((64, 4), dtype=np.float32)
    for l in range(64):
        for r in range(4):
            exp[l, r] = ref[4 * (l >> 4) + r, l & 15]
    if not np.array_equal(got, exp):
        _dump("probe_mfma16_bf16_got.npy", got)
        _dump("probe_mfma16_bf16_ref.npy", ref)
    assert np.array_equal(got, exp)


def test_mfma_f32_32x32x2_layout():
    lib = _probe()
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-4, 5, (32, 2), generator=g).float()
    B = torch.randint(-4, 5, (2, 32), generator=g).float()
    out = torch.zeros(64 * 16, device="cuda")
    Ad, Bd = A.cuda(), B.cuda()
    assert lib.probe_run_mfma_f32(C.c_void_p(Ad.data_ptr()), C.c_void_p(Bd.data_ptr()), C.c_void_p(out.data_ptr()), _stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(64, 16).numpy()
    ref = (A @ B).numpy()
    exp = np.zeros((64, 16), dtype=np.float32)
    for l in range(64):
        for r in range(16):
            exp[l, r] = ref[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    if not np.array_equal(got, exp):
        _dump("probe_mfma_f32_got.npy", got)
    assert np.array_equal(got, exp)


def test_ds_read_tr16_b64_semantics():
    """Within a 16-lane group, lane i supplies the address of 4 consecutive 16-bit
    elements (row i//4, cols 4*(i%4)..+3 of a 4x16 block) and receives column i of the 4
    rows.  Rows may sit at any stride."""
    lib = _probe()
    row_stride_elems = 136  # deliberately not a power of two
    addr = np.zeros(64, dtype=np.int32)
    for l in range(64):
        grp, i = l >> 4, l & 15
        base = grp * 16  # each group reads a different 16-column window
        addr[l] = 2 * ((i >> 2) * row_stride_elems + base + 4 * (i & 3))
    a = torch.from_numpy(addr).cuda()
    out = torch.zeros(64 * 4, dtype=torch.int16, device="cuda")
    assert lib.probe_run_tr_read(C.c_void_p(a.data_ptr()), C.c_void_p(out.data_ptr()), _stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(64, 4).numpy()
    exp = np.zeros((64, 4), dtype=np.int16)
    for l in range(64):
        grp, i = l >> 4, l & 15
        for j in range(4):
            exp[l, j] = j * row_stride_elems + grp * 16 + i
    if not np.array_equal(got, exp):
        _dump("probe_tr_read_got.npy", got)
        _dump("probe_tr_read_addr.npy", addr)
    assert np.array_equal(got, exp)


def test_lds_dma_lane_linear_destination():
    """global_load_lds_dwordx4: lane l of a wave writes its 16 bytes to (wave-uniform LDS base)
    + 16*l, whatever global address it read from (the source address is per lane)."""
    lib = _probe()
    g = torch.Generator().manual_seed(5)
    src = torch.randint(0, 2**31 - 1, (256 + 64, 4), generator=g, dtype=torch.int32).cuda()
    perm = torch.randperm(64, generator=g).to(torch.int32).cuda()
    out = torch.zeros(256, 4, dtype=torch.int32, device="cuda")
    assert lib.probe_run_glds(C.c_void_p(src.data_ptr()), C.c_void_p(perm.data_ptr()), C.c_void_p(out.data_ptr()), _stream()) == 0
    torch.cuda.synchronize()
    exp = torch.stack([src[perm[l].item() + 64 * w] for w in range(4) for l in range(64)])
    if not torch.equal(out, exp):
        _dump("probe_glds_got.npy", out.cpu().numpy())
    assert torch.equal(out, exp)
