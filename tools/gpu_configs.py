"""Throughput of the BASELINE.json configurations that are not the bench line (for DESIGN.md section 5): one GPU, Msamples/s."""
import os, sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene, cube_scene
inst = glaze_amd.RayTraceInstance.new()
MATTEST = os.path.join("tests", "golden", "mattest.glaze")
cases = [("config 2: cube 512x512 depth 2", lambda: glaze_amd.RayTraceScene.from_desc(inst, cube_scene()), 512, 512, 2),
         ("config 3: mattest.glaze 1024x1024 depth 8", lambda: glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(MATTEST)), 1024, 1024, 8),
         ("config 4: atrium 1920x1080 depth 8", lambda: glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080, 8),
         ("config 5: atrium 3840x2160 depth 12 (whole frame on one GPU)", lambda: glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 3840, 2160, 12)]
for name, make, w, h, depth in cases:
    t = time.time(); scene = make(); setup = time.time() - t
    r = glaze_amd.RayTraceRenderer.new(inst, scene, w, h)
    r.set_depth(depth)
    for mode in ("auto", "two_kernels", "path"):
        if mode == "path" and w * h > 600000:
            continue          # the per-wave launch loop with several groups per wave: of no interest
        r.set_launch_mode(mode)
        r.restart(); r.step(2 * depth); r.wait_idle()
        n = 64 * depth
        t = time.time(); r.step(n); r.wait_idle(); dt = time.time() - t
        print("%-62s %-11s (%s) %8.1f Msamples/s  %.4f ms/launch  (scene setup %.3f s, %d tris)" % (name, mode, r.launch_mode(), w * h * n / dt / 1e6, dt / n * 1e3, setup, scene.info().n_world_triangles), flush=True)
