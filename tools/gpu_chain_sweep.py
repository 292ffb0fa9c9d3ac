"""1/WORLD share of the 1080p atrium frame on one GPU: ms per launch against the number of concurrent launch chains
(run once per GPU_MAX_HW_QUEUES setting: beyond the hardware queues HIP maps streams onto, chains serialise)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
world = int(os.environ.get("WORLD", "8"))
r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
r.set_depth(8)
r.enable_counters(False, False)
for chains in [int(c) for c in os.environ.get("CHAINS", "1,2,3,4,6,8").split(",")]:
    r.set_partition(0, world)
    r.set_chains(chains)
    r.restart(); r.step(32); r.wait_idle()
    best = 1e9
    for rep in range(3):
        n = 256
        t = time.time(); r.step(n); r.wait_idle(); best = min(best, (time.time() - t) / n * 1e3)
    print("world %d chains %d: %.4f ms/launch" % (world, chains, best), flush=True)
