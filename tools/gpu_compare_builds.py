"""Two builds of the library, the same renders, compared bit for bit (a NaN's sign and payload are not part of the contract, every NaN is made
the same one): python tools/gpu_compare_builds.py variants/libglaze_hip_OLD.so [launches]   -- each build renders in a process of its own."""
import os
import subprocess
import sys
import tempfile

import numpy as np

code = r'''
import sys
sys.path.insert(0, ".")
import numpy as np
import glaze_amd
from glaze_amd.scenes import atrium_scene
out, launches = sys.argv[1], int(sys.argv[2])
inst = glaze_amd.RayTraceInstance.new()
res = {}
for name, like in (("atrium", False), ("sponza_like", True)):
    scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene(sponza_like=like))
    r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
    r.set_depth(8); r.set_seed(5)
    for chains in (1, 3):
        r.set_chains(chains); r.restart(); r.step(launches); r.wait_idle()
        res["%s_chains%d_hdr" % (name, chains)] = r.read_hdr()
        res["%s_chains%d_out" % (name, chains)] = r.read_result()
np.savez(out, **res)
'''
old = os.path.abspath(sys.argv[1])
launches = sys.argv[2] if len(sys.argv) > 2 else "128"
tmp = tempfile.mkdtemp()
for tag, lib in (("new", None), ("old", old)):
    env = dict(os.environ)
    if lib:
        env["GLAZE_HIP_LIB"] = lib
    subprocess.run([sys.executable, "-c", code, os.path.join(tmp, tag + ".npz"), launches], env=env, check=True)
a, b = np.load(os.path.join(tmp, "new.npz")), np.load(os.path.join(tmp, "old.npz"))
bad = 0
for k in a.files:
    x, y = np.nan_to_num(a[k], nan=-1.0).view(np.uint32), np.nan_to_num(b[k], nan=-1.0).view(np.uint32)
    raw = int((a[k].view(np.uint32) != b[k].view(np.uint32)).any(-1).sum())
    d = int((x != y).any(-1).sum())
    bad += d
    print("%-28s pixels that differ: %d (raw words, NaN payloads included: %d; NaN pixels %d / %d)" % (k, d, raw, int(np.isnan(a[k]).any(-1).sum()), int(np.isnan(b[k]).any(-1).sum())))
print("identical" if bad == 0 else "DIFFERENT")
sys.exit(1 if bad else 0)
