"""A seeded slice of the randomised sweeps (tools/fuzz_attention.py, tools/fuzz_ops.py) inside the GPU suite.

The sweeps found round 2's only real bug (research forward, 129..160 rows, more than one head) and were not part of
`pytest -m gpu`; the full runs (thousands of cases, minutes of oracle time) stay tools, this is the minute of them the driver
executes every round: one case per test, fixed seeds, max_seq_len cycling through the tile-boundary lengths that have broken
kernels before, always at least two heads.  Every case is checked against the fp64 oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# max_seq_len values on both sides of the kernels' tile / schedule boundaries (32-row tiles; 128-row forward blocks; the
# folded / 4-wave backward up to 224, several key blocks from 225 on), 129..160 being the range of the round-2 bug
BOUNDARY_N = [129, 146, 155, 160, 193, 224, 225]


@pytest.mark.parametrize("i", range(40))
def test_fuzz_hstu_mha(i):
    import fuzz_attention as F

    fails, _ = F.mha_sweep(1, seed=4000 + i, force_n=[BOUNDARY_N[i % len(BOUNDARY_N)]], exit_process=False)
    assert fails == 0


@pytest.mark.parametrize("i", range(6))
def test_fuzz_hstu_mha_free_shapes(i):
    """the sweep's own shape distribution (short one-wave shapes, several key blocks, mixed head dims), 5 cases per test"""
    import fuzz_attention as F

    fails, _ = F.mha_sweep(5, seed=4100 + i, exit_process=False)
    assert fails == 0


@pytest.mark.parametrize("i", range(30))
def test_fuzz_research_bias(i):
    import fuzz_attention as F

    fails, _ = F.bias_sweep(1, seed=4200 + i, force_n=[BOUNDARY_N[i % len(BOUNDARY_N)]], exit_process=False)
    assert fails == 0


@pytest.mark.parametrize("i", range(20))
def test_fuzz_delta_attention(i):
    import fuzz_ops as G

    assert G.run_slice(4300 + i, n_delta=1) == 0


@pytest.mark.parametrize("i", range(20))
def test_fuzz_row_ops(i):
    import fuzz_ops as G

    assert G.run_slice(4400 + i, n_row=1) == 0


@pytest.mark.parametrize("i", range(6))
def test_fuzz_jagged_movers_and_layer(i):
    import fuzz_ops as G

    assert G.run_slice(4500 + i, n_jagged=3, n_layer=1) == 0


def test_fuzz_wide_backward_opt_in():
    """round 4's four-wave backward (csrc/hstu_attn_bwd_wide.cuh; slower than the folded kernel, so opt-in) stays correct:
    the head-dim-128 backward tests and a sweep slice in a child process with HSTU_BWD_WIDE=1 (the switch is read once per
    process)."""
    env = dict(os.environ, HSTU_BWD_WIDE="1")
    code = ("import sys; sys.path.insert(0, 'tools'); import fuzz_attention as F\n"
            "from generative_recommenders_amd.ops import _launch; import torch\n"
            "assert _launch.attn_bwd_kernel_name(torch.bfloat16, 128, 128, 200).startswith('hstu_attn_bwd_wide_kernel')\n"
            "f = 0\n"
            "for n in (1, 33, 97, 160, 193, 200, 224):\n"
            "    f += F.mha_sweep(2, seed=4600 + n, force_n=[n], exit_process=False, force_d=128)[0]\n"
            "f += F.mha_sweep(2, seed=4700, big=True, force_n=[200, 185], exit_process=False, force_d=128)[0]\n"
            "sys.exit(1 if f else 0)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_attention_gpu.py", "-q", "-m", "gpu", "-x", "-k",
                        "batch_composition or strided_fused or (fold_backward_every_tile_count and 128 and dtype0)"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
