"""What each content class of the Sponza-like atrium costs (glaze_amd.scenes.atrium_scene(sponza_like=...)): per-kernel time per launch at 1080p depth 8 for
the plain atrium, each class alone (the geometry of the cards and vines is in all of them) and all together.    python tools/gpu_sponza_like.py"""
import sys
import time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
for name, classes in (("plain atrium", False), ("cards + vines, no maps", {"none"}), ("opacity maps", {"opacity"}), ("normal maps", {"normal"}), ("roughness maps", {"roughness"}), ("all three", True)):
    r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene(sponza_like=classes)), 1920, 1080)
    r.set_depth(8); r.step(16); r.wait_idle(); r.stats()
    s0 = r.stats(); n = 128
    t = time.time(); r.step(n); r.wait_idle(); dt = (time.time() - t) / n * 1e3
    s = r.stats()
    r.enable_counters(True, False); r.restart(); r.step(16); a = r.stats(); r.step(32); b = r.stats()
    rays = max(1, b.closest_rays - a.closest_rays)
    print("%-24s %.4f ms/launch (k_trace %.3f + k_shade %.3f) | per sample: nodes %.2f + %.2f, tris %.2f + %.2f, alpha texel bytes %.2f, shade texel bytes %.1f" % (
        name, dt, (s.trace_closest_ms - s0.trace_closest_ms) / n, (s.shade_ms - s0.shade_ms) / n, (b.closest_nodes - a.closest_nodes) / rays, (b.shadow_nodes - a.shadow_nodes) / rays,
        (b.closest_tris - a.closest_tris) / rays, (b.shadow_tris - a.shadow_tris) / rays, (b.alpha_tex_bytes - a.alpha_tex_bytes) / rays,
        ((b.tex_bytes - a.tex_bytes) - (b.alpha_tex_bytes - a.alpha_tex_bytes)) / rays), flush=True)
    del r
