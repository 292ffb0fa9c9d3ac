"""ms per launch of rank 0's 1/8 share of the 1080p atrium when the launches come in calls of K (bench.py --steps K: one timed region = one call + the
exchange), k_path against the two-kernel mode: k_path ends a call when its slowest wave does, and over few launches the slowest wave's sum is
further above the mean."""
import os, sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8)
world = int(os.environ.get("WORLD", "8"))
r.set_partition(0, world)
for mode in ("two_kernels", "path"):
    r.set_launch_mode(mode)
    out = []
    for k in (5, 10, 20, 40, 80, 192):
        r.restart(); r.step(32); r.wait_idle()
        calls = max(3, 384 // k)
        t = time.time()
        for _ in range(calls):
            r.step(k); r.wait_idle()
        out.append("%d: %.4f" % (k, (time.time() - t) / (calls * k) * 1e3))
    print("1/%d share, %-11s ms per launch by launches per call  %s" % (world, mode, "  ".join(out)), flush=True)
