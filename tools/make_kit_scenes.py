#!/usr/bin/env python3
"""Writes the scenes of the reference-comparison kit as `.glaze` V1 files (kit/scenes/): the cube of BASELINE config 2 and the synthetic
atrium of config 4, through this build's Serializer -- files the reference's `parse()` must accept (every chunk is read back by the
oracle's independent reader in tests/test_reference_kit.py).  Deterministic: re-running it reproduces the committed bytes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from glaze_amd.scene_desc import save_scene
from glaze_amd.scenes import atrium_scene, cube_scene
out = os.path.join(ROOT, "kit", "scenes")
os.makedirs(out, exist_ok=True)
save_scene(cube_scene(), os.path.join(out, "cube.glaze"))
# (the bench renders this scene with 1024^2 textures -- an 11 MB file; the kit carries the same geometry, materials and lights with 256^2 ones)
save_scene(atrium_scene(texture_size=256), os.path.join(out, "atrium.glaze"))
for f in sorted(os.listdir(out)):
    print(f, os.path.getsize(os.path.join(out, f)))
