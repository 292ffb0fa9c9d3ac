"""Host cost of enqueueing launches: wall time of step(n) WITHOUT waiting for the GPU, per launch, for several chain counts
(1/8 share of the 1080p atrium frame).  If this approaches the per-launch GPU time the renderer is bound by the host."""
import os, sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import cube_scene, atrium_scene
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
r.set_depth(8)
r.set_partition(0, 8)
r.enable_counters(False, False)
for chains in (1, 2, 3, 4, 6, 8):
    r.set_chains(chains)
    r.restart(); r.step(16); r.wait_idle()
    n = 256
    t = time.perf_counter(); r.step(n); t_enq = time.perf_counter() - t
    r.wait_idle(); t_all = time.perf_counter() - t
    print("chains %d: enqueue %.1f us/launch (%.1f us per kernel), with GPU %.1f us/launch" % (chains, t_enq / n * 1e6, t_enq / n / (2 * chains) * 1e6, t_all / n * 1e6))
