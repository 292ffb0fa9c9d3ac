"""glaze-cli (glaze_amd/csrc/cli/main.cpp): the reference's command line (cli/src/main.rs:24-135).

Argument handling is checked without a GPU (the reference validates the output name, its writability and the resolution
before it creates the instance, cli/src/main.rs:46-75); rendering through the binary is a GPU test and must produce
the very image the library API produces for the same seed.
"""
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import glaze_amd

from conftest import MATTEST

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "glaze_amd", "csrc")
CLI = os.path.join(CSRC, "glaze-cli")

needs_cli = pytest.mark.skipif(not os.path.exists(CLI), reason="glaze-cli is not built")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True)


@needs_cli
def test_cli_argument_errors(tmp_path):
    out = str(tmp_path / "o.png")
    assert run("--help").returncode == 0 and "Usage" in run("--help").stdout + run("--help").stderr
    assert run().returncode == 2 and run(MATTEST).returncode == 2                       # missing positional arguments
    r = run(MATTEST, str(tmp_path / "o.bmp"))
    assert r.returncode == 1 and "must end with .jpg or .png" in r.stderr              # cli/src/main.rs:46-49
    r = run(MATTEST, str(tmp_path / "nodir" / "o.png"))
    assert r.returncode == 1 and "can not be written" in r.stderr                      # :50-55
    for bad in ("1920", "0x1080", "axb", "1920x", "70000x10"):
        r = run(MATTEST, out, "-r", bad)
        assert r.returncode == 1 and ("resolution" in r.stderr or "width" in r.stderr or "height" in r.stderr), bad
    r = run(MATTEST, out, "-i", "bidir")
    assert r.returncode == 2 and "possible values: direct, pt" in r.stderr
    r = run(MATTEST, out, "--frobnicate")
    assert r.returncode == 2 and "unexpected argument" in r.stderr
    r = run(MATTEST, out, "--spp")
    assert r.returncode == 2 and "value is required" in r.stderr
    r = run(MATTEST, out, "--texture-lod", "trilinear")
    assert r.returncode == 2 and "possible values: off, cones, aniso" in r.stderr


@pytest.mark.gpu
@needs_cli
def test_cli_renders_what_the_library_renders(tmp_path, instance):
    png, jpg, pfm = str(tmp_path / "o.png"), str(tmp_path / "o.jpg"), str(tmp_path / "o.pfm")
    r = run(MATTEST, png, "-r", "96x64", "-s", "3", "--seed", "11", "--depth", "4", "--hdr-out", pfm, "--report")
    assert r.returncode == 0 and "All done :)" in r.stderr, r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["width"] == 96 and rep["height"] == 64 and rep["spp"] == 3 and rep["steps_per_sample"] == 4 and rep["launches"] == 12
    assert rep["triangles"] == 138480 and rep["msamples_per_s"] > 0
    scene = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))
    ren = glaze_amd.RayTraceRenderer.new(instance, scene, 96, 64)
    ren.set_seed(11)
    ren.set_depth(4)
    want = ren.draw(3)
    assert np.array_equal(np.asarray(Image.open(png)), want)                                  # same RGBA8 image, losslessly stored
    hdr = ren.read_hdr()
    with open(pfm, "rb") as f:
        assert f.readline() == b"PF\n" and f.readline() == b"96 64\n" and float(f.readline()) < 0   # little-endian, bottom-up rows
        data = np.frombuffer(f.read(), "<f4").reshape(64, 96, 3)[::-1]
    ok = np.isfinite(hdr[..., :3]).all(-1)
    assert np.allclose(data[ok], (hdr[..., :3] / hdr[..., 3:4])[ok], rtol=1e-6, atol=0)
    r = run(MATTEST, jpg, "-r", "96x64", "-s", "3", "--seed", "11", "--depth", "4", "-i", "direct")
    assert r.returncode == 0
    im = Image.open(jpg)
    assert im.size == (96, 64) and im.mode == "RGB" and np.asarray(im).std() > 5
    r = run(MATTEST, png, "-r", "96x64", "-s", "3", "--seed", "11", "--depth", "4", "--texture-lod", "aniso")
    assert r.returncode == 0, r.stderr
    ren.set_texture_lod(2)
    assert np.array_equal(np.asarray(Image.open(png)), ren.draw(3)) and not np.array_equal(ren.read_rgba8(), want)
