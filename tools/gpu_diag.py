"""Ad-hoc GPU diagnostics (not part of the test-suite): full-size parity + first timings."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene, cube_scene
from oracle.pyoracle import OracleScene, OracleRenderer

import os
inst = glaze_amd.RayTraceInstance.new()
if os.environ.get("GLZ_DIAG_BUILDER"):
    inst.set_bvh_builder(os.environ["GLZ_DIAG_BUILDER"])
desc = atrium_scene()
t = time.time(); scene = glaze_amd.RayTraceScene.from_desc(inst, desc); print("scene create %.3fs" % (time.time() - t))
i = scene.info(); print("tris", i.n_world_triangles, "bvh depth", i.bvh_depth, "sah", i.bvh_sah_cost, "build ms", i.build_ms)
W, H, L = 1920, 1080, 16
r = glaze_amd.RayTraceRenderer.new(inst, scene, W, H)
r.set_depth(8)
r.step(L); r.wait_idle()
a = r.read_hdr()
bad = ~np.isfinite(a[..., :3]).all(-1)
print("gpu nonfinite", int(bad.sum()), "nan", int(np.isnan(a[..., :3]).any(-1).sum()))
if "--oracle" in sys.argv:
    o = OracleRenderer(OracleScene(desc), W, H); o.set_depth(8)
    t = time.time(); o.step(L); print("oracle %d launches %.1fs" % (L, time.time() - t))
    c = o.read_hdr()
    badc = ~np.isfinite(c[..., :3]).all(-1)
    print("oracle nonfinite", int(badc.sum()), "same set", bool(np.array_equal(bad, badc)))
    ok = ~(bad | badc)
    exact = (a.view(np.uint32) == c.view(np.uint32)).all(-1)
    print("bit-exact pixels %.6f%%" % (100 * exact.mean()), "max rel err", float(np.max(np.abs(a[ok] - c[ok]) / np.maximum(np.abs(c[ok]), 1e-6))))
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:8]:
        print("  bad px", y, x, a[y, x], c[y, x])
# timing
for (name, d, w, h, depth) in (("atrium1080p", None, 1920, 1080, 8), ("cube512", cube_scene(), 512, 512, 2)):
    if d is not None:
        sc = glaze_amd.RayTraceScene.from_desc(inst, d)
        rr = glaze_amd.RayTraceRenderer.new(inst, sc, w, h)
    else:
        rr = r
    rr.set_depth(depth)
    rr.restart(); rr.step(8); rr.wait_idle()
    rr.restart()
    n = 64
    t = time.time(); rr.step(n); rr.wait_idle(); dt = time.time() - t
    s = rr.stats()
    print("%s: %d launches %.3fs wall -> %.1f Msamples/s | kernels ms: trace %.2f shade %.2f shadow-flush %.2f (per launch %.3f/%.3f/%.3f)" % (
        name, n, dt, w * h * n / dt / 1e6, s.trace_closest_ms, s.shade_ms, s.trace_shadow_ms, s.trace_closest_ms / n, s.shade_ms / n, s.trace_shadow_ms / n))
    rr.enable_counters(True, True); rr.restart(); rr.step(n); rr.wait_idle(); s = rr.stats()
    smp = w * h * n
    print("   per sample: closest rays %.3f shadow rays %.3f hits %.3f | nodes c %.1f s %.1f | tris c %.2f s %.2f" % (
        s.closest_rays / smp, s.shadow_rays / smp, s.hits / smp, s.closest_nodes / smp, s.shadow_nodes / smp, s.closest_tris / smp, s.shadow_tris / smp))
    ph = list(s.phase)
    for nm, o in (("closest", 0), ("shadow", 6)):
        print("   %s phases: node rounds/ray %.1f lanes %.1f | leaf rounds/ray %.2f lanes %.1f | refills lanes %.1f" % (
            nm, 64.0 * ph[o] / max(1, s.closest_rays if o == 0 else s.shadow_rays), ph[o + 1] / max(1, ph[o]),
            64.0 * ph[o + 2] / max(1, s.closest_rays if o == 0 else s.shadow_rays), ph[o + 3] / max(1, ph[o + 2]), ph[o + 5] / max(1, ph[o + 4])))
    rr.enable_counters(False, True)
