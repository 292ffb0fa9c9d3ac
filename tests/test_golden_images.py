"""Golden images (tests/golden/oracle_*.npz, written by tools/gen_golden_oracle.py): the oracle must reproduce them bit for bit
on the CPU, and the HIP path must match them on the GPU -- a drift guard for both between rounds."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_oracle as gg


def load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "oracle_%s.npz" % name))
    return z["hdr"], z["result"]


@pytest.mark.parametrize("name", sorted(gg.CASES))
def test_oracle_reproduces_golden(name):
    hdr, result = gg.render(name)
    g_hdr, g_result = load(name)
    assert np.array_equal(hdr.view(np.uint32), g_hdr.view(np.uint32)) and np.array_equal(result.view(np.uint32), g_result.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gg.CASES))
def test_hip_matches_golden(name, instance):
    import glaze_amd
    c = gg.CASES[name]
    scene = glaze_amd.RayTraceScene.from_desc(instance, c["scene"]())
    r = glaze_amd.RayTraceRenderer.new(instance, scene, c["w"], c["h"])
    r.set_depth(c["depth"])
    r.set_seed(c["seed"])
    r.step(c["launches"])
    g_hdr, g_result = load(name)
    hdr, result = r.read_hdr(), r.read_result()
    same = (hdr.view(np.uint32) == g_hdr.view(np.uint32)) | (np.isnan(hdr) & np.isnan(g_hdr))
    assert same.all(), "%d of %d values differ" % ((~same).sum(), same.size)
    same = (result.view(np.uint32) == g_result.view(np.uint32)) | (np.isnan(result) & np.isnan(g_result))
    assert same.all()
