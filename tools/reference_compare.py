#!/usr/bin/env python3
"""Reference-comparison kit: does an image written by the REAL glaze-cli (Vulkan ray tracing) agree with this build's render of the same
`.glaze` file?

Nothing in this repository's own tests can answer that -- the reference needs rustc, shaderc and a Vulkan-RT device, none of which the
authoring image has -- so DESIGN.md calls the render parity "unpinned".  This script is what a maintainer with such a machine runs:

    glaze-cli kit/scenes/cube.glaze ref_cube.png --res 512x512 --spp 64 -i pt               (the reference, cli/src/main.rs:24-39)
    python tools/reference_compare.py --scene kit/scenes/cube.glaze --reference ref_cube.png --spp 64

The reference seeds its launches from OS entropy (lib/src/vulkan/raytracer.rs:779), so two renders never share a sample: the test is
statistical.  This build renders the scene K times with K different seeds (default 8), same resolution, same samples per pixel, same
depth (the reference's PT_STEPS = 6), exports each through the same 8-bit sRGB rule the reference's blit applies, and compares, per
64 x 64 tile and colour channel, the reference's mean with the K means:

    z = (ref - mean_K) / (std_K * sqrt(1 + 1 / K))              Student t with K - 1 degrees of freedom under "same renderer"

Accepted when (a) no more than 2 % of the (tile, channel) cells lie outside the two-sided 99.9 % interval of that t distribution,
(b) the mean of z over all cells is within +- 4 / sqrt(cells) of zero (no global brightness shift), and (c) the frame means differ by
less than 1 %.  Tiles that are more than 1 % saturated (a value of 255 hides the mean) or nearly black are left out.  `--self-test`
checks the rule on this build alone: a render with a ninth seed passes, the same scene with one BSDF changed (every Lambert albedo
x 0.9) fails.  The pure-numpy core (tile_means / compare) is unit-tested on the CPU with synthetic images (tests/test_reference_kit.py).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# two-sided 99.9 % points of Student's t for K - 1 = 1 .. 30 degrees of freedom (then the normal's 3.29)
T999 = [636.62, 31.599, 12.924, 8.610, 6.869, 5.959, 5.408, 5.041, 4.781, 4.587, 4.437, 4.318, 4.221, 4.140, 4.073, 4.015, 3.965, 3.922, 3.883, 3.850,
        3.819, 3.792, 3.768, 3.745, 3.725, 3.707, 3.690, 3.674, 3.659, 3.646]


def srgb8_to_linear(img8):
    c = img8.astype(np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def tile_means(img8, tile=64):
    """(tiles_y, tiles_x, 3) means of the linearised image, and the fraction of saturated samples per tile"""
    h, w = img8.shape[:2]
    ty, tx = (h + tile - 1) // tile, (w + tile - 1) // tile
    lin = srgb8_to_linear(img8[..., :3])
    means, sat = np.zeros((ty, tx, 3)), np.zeros((ty, tx))
    for j in range(ty):
        for i in range(tx):
            blk = lin[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile]
            means[j, i] = blk.reshape(-1, 3).mean(0)
            sat[j, i] = (img8[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile, :3] == 255).mean()
    return means, sat


def compare(ref8, ours8, tile=64):
    """ref8: H x W x (3|4) uint8; ours8: K such images of this build (K >= 3).  Returns a dict with the verdict and the numbers behind it."""
    k = len(ours8)
    assert k >= 3 and all(o.shape[:2] == ref8.shape[:2] for o in ours8), "same resolution, at least three renders of this build"
    ref_m, ref_sat = tile_means(ref8, tile)
    ours = [tile_means(o, tile) for o in ours8]
    m = np.stack([o[0] for o in ours])
    sat = np.maximum(ref_sat, np.max(np.stack([o[1] for o in ours]), axis=0))
    mu, sd = m.mean(0), m.std(0, ddof=1)
    used = (sat < 0.01)[..., None] & (mu > 1e-4) & (sd > 0)
    z = np.where(used, (ref_m - mu) / np.where(sd > 0, sd, 1.0) / np.sqrt(1.0 + 1.0 / k), 0.0)
    cells = int(used.sum())
    crit = T999[min(k - 1, len(T999)) - 1] if k - 1 <= len(T999) else 3.29
    outside = float((np.abs(z[used]) > crit).mean()) if cells else 1.0
    # the mean of z: t-distributed cells have variance (K - 1) / (K - 3) for K > 3 (taken as 3 for K = 3)
    var_t = (k - 1.0) / (k - 3.0) if k > 3 else 3.0
    shift = float(z[used].mean()) if cells else 0.0
    shift_limit = 4.0 * np.sqrt(var_t / max(cells, 1))
    frame_rel = float(abs(ref_m[used].mean() - mu[used].mean()) / mu[used].mean()) if cells else 1.0
    ok = bool(cells >= 12 and outside <= 0.02 and abs(shift) <= shift_limit and frame_rel < 0.01)
    return {"pass": ok, "renders_of_this_build": k, "cells_used": cells, "cells_total": int(used.size), "t_999": crit, "fraction_outside": round(outside, 4),
            "mean_z": round(shift, 4), "mean_z_limit": round(float(shift_limit), 4), "frame_mean_relative_difference": round(frame_rel, 5),
            "worst_cells": [{"tile": [int(a), int(b)], "channel": int(c), "z": round(float(z[a, b, c]), 2)} for a, b, c in
                            sorted(zip(*np.nonzero(used)), key=lambda q: -abs(z[q]))[:5]]}


def load_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGBA"))


def render(scene_path, width, height, spp, depth, seed, integrator="pt", tweak=None):
    import glaze_amd
    inst = glaze_amd.RayTraceInstance.new()
    if inst is None:
        raise SystemExit("no gfx950 device: " + glaze_amd.abi.last_error())
    scene = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(scene_path))
    r = glaze_amd.RayTraceRenderer.new(inst, scene, width, height)
    if tweak is not None:
        p = glaze_amd.parse(scene_path)
        mats, lights = p.materials(), p.lights()
        tweak(mats)
        r.update_materials_and_lights(mats, lights)
    r.set_integrator(glaze_amd.Integrator.DIRECT if integrator == "direct" else glaze_amd.Integrator.PATH_TRACE)
    r.set_depth(depth)
    r.set_seed(seed)
    return r.draw(spp)


def darker_lamberts(mats):
    """the deliberately broken BSDF of the self-test: every material's diffuse multiplier x 0.9"""
    for m in mats:
        for c in range(3):
            m.diffuse_mul[c] = int(m.diffuse_mul[c] * 0.9)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scene", required=True, help="the .glaze file both renderers were given")
    ap.add_argument("--reference", help="PNG written by the reference's glaze-cli (its --res and --spp must be given here too)")
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=6, help="path segments per sample: the reference's PT_STEPS (raytracer.rs:29)")
    ap.add_argument("--integrator", choices=("pt", "direct"), default="pt")
    ap.add_argument("--res", default=None, help="WxH (default: the reference image's)")
    ap.add_argument("--renders", type=int, default=8, help="independent renders of this build")
    ap.add_argument("--self-test", action="store_true")
    a = ap.parse_args()
    if a.self_test:
        w, h = (int(x) for x in (a.res or "256x256").split("x"))
        ours = [render(a.scene, w, h, a.spp, a.depth, 1000 + k, a.integrator) for k in range(a.renders)]
        same = compare(render(a.scene, w, h, a.spp, a.depth, 77, a.integrator), ours)
        broken = compare(render(a.scene, w, h, a.spp, a.depth, 78, a.integrator, tweak=darker_lamberts), ours)
        print(json.dumps({"another_seed": same, "albedo_x_0.9": broken}, indent=1))
        ok = same["pass"] and not broken["pass"]
        print("self-test %s: another seed %s, a 10 %% darker albedo %s" % ("PASSED" if ok else "FAILED", "accepted" if same["pass"] else "REJECTED", "rejected" if not broken["pass"] else "ACCEPTED"))
        sys.exit(0 if ok else 1)
    if not a.reference:
        raise SystemExit("--reference ref.png (or --self-test)")
    ref = load_png(a.reference)
    h, w = ref.shape[:2]
    if a.res:
        assert (w, h) == tuple(int(x) for x in a.res.split("x")), "the reference image is %dx%d" % (w, h)
    ours = [render(a.scene, w, h, a.spp, a.depth, 1000 + k, a.integrator) for k in range(a.renders)]
    out = compare(ref, ours)
    print(json.dumps(out, indent=1))
    print("PASS: the reference image is statistically indistinguishable from this build's renders" if out["pass"] else "FAIL: the images differ beyond Monte Carlo noise")
    sys.exit(0 if out["pass"] else 1)


if __name__ == "__main__":
    main()
