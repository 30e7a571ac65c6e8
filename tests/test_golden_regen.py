"""The golden fixtures are reproducible: when the reference tree is present (the build container; it does not exist on
the GPU box), ``tests/golden/make_golden.py`` is run into a scratch directory and every array of every fixture must
equal the committed one EXACTLY -- same seeds, same reference, same torch.  Catches both an edited fixture and a
generator that draws from an unseeded source."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REFERENCE = "/root/reference/generative_recommenders"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference tree (build container only)")
def test_golden_fixtures_regenerate_bit_identically(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--out", str(tmp_path)], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    fresh = sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz"))
    assert committed == fresh, f"fixture sets differ: committed {committed}, regenerated {fresh}"
    for f in committed:
        a, b = np.load(os.path.join(GOLDEN, f)), np.load(os.path.join(tmp_path, f))
        assert sorted(a.files) == sorted(b.files), f"{f}: array names differ"
        for key in a.files:
            assert a[key].dtype == b[key].dtype and a[key].shape == b[key].shape, f"{f}:{key} dtype / shape"
            assert np.array_equal(a[key], b[key]), f"{f}:{key} is not reproduced bit for bit"
