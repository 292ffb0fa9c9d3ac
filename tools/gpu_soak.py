"""Determinism soak: the same render repeated (and with other chain counts) must give bit-identical float images; run after kernel changes.
Usage: python tools/gpu_soak.py [launches] [repeats]"""
import sys, time, hashlib
sys.path.insert(0, ".")
import numpy as np
import glaze_amd
from glaze_amd.scenes import atrium_scene
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 512
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
r.set_depth(8); r.set_seed(5)
ref = None
for rep in range(repeats):
    for chains in (1, 3):
        r.set_chains(chains)
        r.restart()
        t = time.time(); r.step(launches); r.wait_idle(); dt = time.time() - t
        h = hashlib.sha256(r.read_hdr().tobytes()).hexdigest()[:16]
        ok = ref is None or h == ref
        ref = ref or h
        print("repeat %d chains %d: %s %s  (%.1f Msamples/s)" % (rep, chains, h, "ok" if ok else "DIFFERENT", 1920 * 1080 * launches / dt / 1e6), flush=True)
        assert ok
print("soak ok")
