"""k_trace / k_shade per launch of the Sponza-like atrium for every variants/libglaze_hip_*.so and the in-tree build (1080p, depth 8)."""
import glob, os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
out = []
for classes in (False, {"opacity"}):
    r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene(sponza_like=classes, texture_size=256)), 1920, 1080)
    r.set_depth(8); r.step(16); r.wait_idle(); r.stats()
    s0 = r.stats(); n = 96
    r.step(n); r.wait_idle(); s = r.stats()
    out.append("%s: trace %.3f shade %.3f" % ("opacity" if classes else "plain", (s.trace_closest_ms - s0.trace_closest_ms) / n, (s.shade_ms - s0.shade_ms) / n))
    del r
print(" | ".join(out))
'''
for lib in [None] + sorted(glob.glob("variants/libglaze_hip_*.so")):
    env = dict(os.environ)
    if lib:
        env["GLAZE_HIP_LIB"] = os.path.abspath(lib)
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-36s %s" % (os.path.basename(lib) if lib else "in-tree", (o.stdout.strip() or o.stderr.strip()[-300:])), flush=True)
