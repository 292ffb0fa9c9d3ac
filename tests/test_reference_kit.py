"""The reference-comparison kit (kit/README.md, tools/reference_compare.py): the acceptance rule on synthetic images (CPU), the scene
files of the kit read back by the oracle's independent reader (CPU), and the end-to-end self-test on the GPU."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("reference_compare", os.path.join(ROOT, "tools", "reference_compare.py"))
rc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rc)


def noisy(rng, base, sigma):
    """an 8-bit sRGB "render": a smooth linear image plus per-pixel Monte Carlo noise, through the OETF"""
    lin = np.clip(base + rng.normal(0.0, sigma, base.shape), 0.0, 1.0)
    enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055)
    return np.round(enc * 255.0).astype(np.uint8)


def scene_image(h=256, w=320):
    y, x = np.mgrid[0:h, 0:w]
    g = 0.15 + 0.5 * (x / w) * (0.4 + 0.6 * y / h)
    return np.stack([g, 0.8 * g, 0.6 * g + 0.05], -1)


def test_the_rule_accepts_the_same_renderer_and_rejects_a_biased_one():
    rng = np.random.default_rng(1)
    base = scene_image()
    ours = [noisy(rng, base, 0.08) for _ in range(8)]
    accepted = 0
    for _ in range(20):
        accepted += rc.compare(noisy(rng, base, 0.08), ours)["pass"]
    assert accepted >= 19                                                     # false alarms are rare
    for bias in (0.97, 1.03, 0.9):
        out = rc.compare(noisy(rng, base * bias, 0.08), ours)
        assert not out["pass"], out
    # a local defect: one material (a region) 8 % darker
    local = base.copy()
    local[64:192, 64:192] *= 0.92
    out = rc.compare(noisy(rng, local, 0.08), ours)
    assert not out["pass"] and out["fraction_outside"] > 0.02
    # a different noise level alone (another sampler, same expectation) is not a difference of the mean image
    assert rc.compare(noisy(rng, base, 0.05), ours)["pass"]


def test_saturated_and_black_tiles_are_left_out():
    rng = np.random.default_rng(2)
    base = scene_image()
    base[:64] = 2.0                         # a blown-out band: 255 everywhere
    base[-64:] = 0.0
    ours = [noisy(rng, base, 0.05) for _ in range(6)]
    out = rc.compare(noisy(rng, base, 0.05), ours)
    assert out["pass"] and out["cells_used"] < out["cells_total"]


def test_tile_means_of_a_ragged_image():
    img = np.zeros((70, 130, 4), np.uint8)
    img[..., :3] = 128
    img[64:, 128:, :3] = 255
    m, sat = rc.tile_means(img)
    assert m.shape == (2, 3, 3) and sat.shape == (2, 3)
    assert abs(m[0, 0, 0] - rc.srgb8_to_linear(np.uint8(128))) < 1e-12 and sat[1, 2] == 1.0 and sat[0, 0] == 0.0


def test_kit_scenes_are_valid_glaze_files_and_reproducible(tmp_path):
    """every chunk of the kit's scene files read back by the oracle's reader (liblzma, xxhash, PIL); regenerating them gives the same bytes"""
    from oracle import glaze_v1
    cube = glaze_v1.parse(os.path.join(ROOT, "kit", "scenes", "cube.glaze"))
    assert cube.vertices().shape[0] == 24 and len(cube.meshes()) == 1 and len(cube.materials()) == 3 and len(cube.lights()) == 1 and len(cube.cameras()) == 1
    atrium = glaze_v1.parse(os.path.join(ROOT, "kit", "scenes", "atrium.glaze"))
    assert sum(m["indices"].size for m in atrium.meshes()) // 3 == 262267 and len(atrium.materials()) == 26 and len(atrium.lights()) == 2 and atrium.meta() is not None
    import hashlib
    from glaze_amd.scene_desc import save_scene
    from glaze_amd.scenes import cube_scene
    save_scene(cube_scene(), str(tmp_path / "cube.glaze"))
    assert hashlib.md5(open(tmp_path / "cube.glaze", "rb").read()).hexdigest() == hashlib.md5(open(os.path.join(ROOT, "kit", "scenes", "cube.glaze"), "rb").read()).hexdigest()


@pytest.mark.gpu
def test_self_test_on_the_gpu():
    """two renders of this build with different seeds pass, a deliberately broken BSDF (every albedo x 0.9) fails"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reference_compare.py"), "--scene", os.path.join(ROOT, "kit", "scenes", "cube.glaze"), "--self-test",
                        "--res", "256x256", "--spp", "32"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "self-test PASSED" in p.stdout
