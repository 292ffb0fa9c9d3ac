"""Determinism soak: the same render repeated (and with other chain counts) must give bit-identical float images; run after kernel changes.
Usage: python tools/gpu_soak.py [launches] [repeats]          env: SOAK_SCENE=sponza_like (the atrium with opacity / normal / roughness maps: the
alpha phase decides candidates in whatever order its quorum fills, which must change no pixel) SOAK_PATH=<renders of the k_path part>"""
import sys, time, hashlib
sys.path.insert(0, ".")
import numpy as np
import glaze_amd
from glaze_amd.scenes import atrium_scene
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 512
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inst = glaze_amd.RayTraceInstance.new()
import os
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene(sponza_like=os.environ.get("SOAK_SCENE") == "sponza_like"))
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

# The per-wave launch loop (k_path) hands data from lane to lane THROUGH GLOBAL MEMORY inside one wave -- shadow-queue entries written by
# the shading lanes and read by whichever lane traces them, accumulators written by a tracing lane and read by the pixel's own lane --
# and relies on a wave's vector memory operations staying in order.  A violation would show as a rare, unrepeatable difference:
# one GPU's share of an 8-way partition, rendered over and over in both launch modes, batch lengths varied.
import os
r.set_chains(0)
r.set_depth(8)
r.set_partition(3, 8)
r.set_launch_mode("two_kernels")
r.restart(); r.step(77)
ref = r.read_hdr().view(np.uint32).copy()
r.set_launch_mode("path")
runs = int(os.environ.get("SOAK_PATH", "60"))
t0 = time.time()
for i in range(runs):
    r.restart()
    left = 77
    for k in ((77,), (1, 76), (13, 64), (40, 37), (5, 5, 67))[i % 5]:
        r.step(k); left -= k
    assert left == 0
    img = r.read_hdr().view(np.uint32)
    same = np.array_equal(np.nan_to_num(img.view(np.float32), nan=-1.0).view(np.uint32), np.nan_to_num(ref.view(np.float32), nan=-1.0).view(np.uint32))
    assert same, "k_path run %d differs from the two-kernel image" % i
print("k_path: %d renders of a 1/8 share (77 launches each, five batch patterns) bit-identical to the two-kernel image, %.1f s" % (runs, time.time() - t0))
