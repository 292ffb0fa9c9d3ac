"""Times every variants/libglaze_hip_*.so (plus the in-tree build) on the bench workload, each in its own process."""
import glob, os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8); r.step(16); r.wait_idle(); r.stats()
s0 = r.stats(); n = 64
t = time.time(); r.step(n); r.wait_idle(); dt = time.time() - t
s = r.stats()
print("%8.1f Msamples/s | trace %.3f shade %.3f shadow-flush %.3f ms" % (1920*1080*n/dt/1e6, (s.trace_closest_ms-s0.trace_closest_ms)/n, (s.shade_ms-s0.shade_ms)/n, (s.trace_shadow_ms-s0.trace_shadow_ms)/n))
'''
libs = [None] + sorted(glob.glob("variants/libglaze_hip_*.so"))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["GLAZE_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-40s %s" % (os.path.basename(lib) if lib else "in-tree", (out.stdout.strip() or out.stderr.strip()[-300:])))
