"""Cost of texture level of detail: the 1080p atrium with level 0 everywhere (the default) with ray-cone LOD, and with the anisotropic footprint."""
import sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
r.set_depth(8)
for name, mode in (("base", 0), ("ray_cones", 1), ("aniso", 2), ("base", 0)):
    r.set_texture_lod(mode)
    r.restart(); r.step(24); r.wait_idle(); r.stats()
    s0 = r.stats()
    t = time.time(); r.step(64); r.wait_idle(); dt = (time.time() - t) / 64 * 1e3
    s = r.stats()
    print("%-9s %.3f ms/launch (k_trace %.3f, k_shade %.3f)" % (name, dt, (s.trace_closest_ms - s0.trace_closest_ms) / 64, (s.shade_ms - s0.shade_ms) / 64), flush=True)
