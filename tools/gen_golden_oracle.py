#!/usr/bin/env python3
"""Writes tests/golden/oracle_*.npz: cumulative RGBA32F images computed by the CPU oracle for two small fixed workloads.
They pin the oracle (and with it the HIP path) against drift between rounds: tests/test_golden_images.py requires the
oracle to reproduce them bit for bit and the GPU to match them.  Regenerate ONLY when a deliberate semantic change is made
(and say so in DESIGN.md)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from glaze_amd.scenes import cube_scene
from helpers import desc_from_oracle_parse
from oracle.pyoracle import OracleRenderer, OracleScene

CASES = {
    "cube_lambert": dict(scene=lambda: cube_scene(), w=64, h=64, depth=2, seed=0, launches=16),
    "mattest": dict(scene=lambda: desc_from_oracle_parse(os.path.join(ROOT, "tests", "golden", "mattest.glaze")), w=48, h=48, depth=4, seed=3, launches=8),
}


def render(case):
    c = CASES[case]
    r = OracleRenderer(OracleScene(c["scene"]()), c["w"], c["h"])
    r.set_depth(c["depth"])
    r.set_seed(c["seed"])
    r.step(c["launches"])
    return r.read_hdr(), r.read_result()


if __name__ == "__main__":
    for name in CASES:
        hdr, result = render(name)
        path = os.path.join(ROOT, "tests", "golden", "oracle_%s.npz" % name)
        np.savez_compressed(path, hdr=hdr, result=result)
        print("wrote", path, hdr.shape, float(hdr[..., :3].mean()))
