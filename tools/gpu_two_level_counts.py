"""Counted work of the flattened and the two-level structure on the instanced forest (enable_counters): node visits, triangle tests per
closest-hit / shadow ray.  With GLAZE_HIP_LIB naming a -DGLZ_WAVE_TIMES variant also instance entries and the split of the node visits."""
import os, sys
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import forest_scene
inst = glaze_amd.RayTraceInstance.new()
for n in (200, 2000):
    desc = forest_scene(n)
    for mode in ("flat", "two_level"):
        inst.set_as_levels(mode)
        r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, desc), 1920, 1080)
        r.set_depth(8)
        r.enable_counters(True, True)
        r.step(16); r.wait_idle()
        s = r.stats()
        print("forest x%-5d %-9s per closest-hit ray: %.2f node visits, %.2f triangle tests, hit %.3f | per shadow ray: %.2f node visits, %.2f triangle tests (%.3f shadow rays per sample)" % (
            n, mode, s.closest_nodes / s.closest_rays, s.closest_tris / s.closest_rays, s.hits / s.closest_rays, s.shadow_nodes / max(s.shadow_rays, 1),
            s.shadow_tris / max(s.shadow_rays, 1), s.shadow_rays / s.closest_rays), flush=True)
inst.set_as_levels("auto")
