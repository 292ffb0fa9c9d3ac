"""Cost of the two-level traversal, ms per launch at 1080p, depth 8:
  forest : one fluted column (6 144 triangles) instanced N times under random placements on a ground grid -- what two levels are for
  atrium : the 1080p atrium forced through a top level over its 230 per-mesh hierarchies (every mesh instanced once; wall-sized
           boxes overlap everything) -- the worst case, which the automatic choice never takes."""
import os, sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene, forest_scene


inst = glaze_amd.RayTraceInstance.new()
cases = [("forest x%d" % n, forest_scene(n)) for n in [int(x) for x in os.environ.get("FOREST", "200,2000").split(",")]]
if os.environ.get("ATRIUM", "1") != "0":
    cases.append(("atrium", atrium_scene()))
for name, desc in cases:
    for mode in ("flat", "two_level"):
        inst.set_as_levels(mode)
        t0 = time.time()
        scene = glaze_amd.RayTraceScene.from_desc(inst, desc)
        build = time.time() - t0
        i = scene.info()
        r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
        r.set_depth(8)
        r.enable_counters(False, False)
        r.restart(); r.step(8); r.wait_idle()
        best = 1e9
        for rep in range(3):
            t = time.time(); r.step(16); r.wait_idle(); best = min(best, (time.time() - t) / 16 * 1e3)
        lib = os.environ.get("GLAZE_HIP_LIB")
        if lib and mode == "two_level":
            import ctypes
            dl = ctypes.CDLL(lib)
            if hasattr(dl, "glz_debug_tl_stats"):
                st = (ctypes.c_ulonglong * 8)()
                dl.glz_debug_tl_stats(st, 1)
                r.step(4); r.wait_idle()
                dl.glz_debug_tl_stats(st, 1)
                rays = max(st[0], 1)
                print("   per closest-hit ray: %.1f top-level nodes, %.1f mesh nodes, %.2f instances entered, %.2f leaf visits; per wave-iteration %.1f lanes at nodes, %.1f at leaves" % (
                    st[1] / rays, st[2] / rays, st[3] / rays, st[4] / rays, (st[1] + st[2]) / max(st[5], 1), (st[3] + st[4]) / max(st[6], 1)), flush=True)
        print("%-12s %-9s levels %d, %8d world / %8d structure triangles, %7.1f MB, built in %.2f s: %.3f ms/launch" % (
            name, mode, i.as_levels, i.n_world_triangles, i.n_as_triangles, i.as_bytes / 1e6, build, best), flush=True)
        del r, scene
inst.set_as_levels("auto")
