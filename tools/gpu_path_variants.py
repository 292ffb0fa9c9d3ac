"""Times the per-wave launch loop (k_path) of every variants/libglaze_hip_*.so and of the in-tree build on rank 0's share of the 1080p
atrium at world sizes 4 / 8 / 16, next to the two-kernel mode -- each library in its own process."""
import glob, os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8)
out = []
import os
cases = [(c.split("/")[0], int(c.split("/")[1])) for c in os.environ.get("GLZ_VARIANT_CASES", "two_kernels/1,two_kernels/8,path/4,path/8,path/16").split(",")]
for mode, world in cases:
    r.set_partition(0, world); r.set_launch_mode(mode)
    r.restart(); r.step(16); r.wait_idle()
    n = 128
    t = time.time(); r.step(n); r.wait_idle(); dt = (time.time() - t) / n * 1e3
    out.append("%s/%d %.4f" % (mode[:4], world, dt))
# the image does not depend on the mode, whatever the variant does to the pixel -> wave mapping
import numpy as np
r.set_partition(3, 8)
imgs = []
for mode in ("two_kernels", "path"):
    r.set_launch_mode(mode); r.restart(); r.step(19); imgs.append(r.read_hdr())
same = np.array_equal(np.nan_to_num(imgs[0], nan=-1).view(np.uint32), np.nan_to_num(imgs[1], nan=-1).view(np.uint32))
print(" | ".join(out) + (" | images identical" if same else " | IMAGES DIFFER"))
'''
libs = [None] + sorted(glob.glob("variants/libglaze_hip_*.so"))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["GLAZE_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-36s %s" % (os.path.basename(lib) if lib else "in-tree", (out.stdout.strip() or out.stderr.strip()[-300:])), flush=True)
