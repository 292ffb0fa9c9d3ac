"""Per-launch cost of one rank's share of the 1080p frame at world sizes 1/2/4/8, measured on ONE GPU (rank 0's tiles).
Predicts the strong-scaling efficiency of bench.py --gpus N before the 8-GPU run: speed-up(N) ~ t(1) / t(N)."""
import sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
base = None
import os
chains = int(os.environ.get('CHAINS', '0'))
for world in (1, 2, 3, 4, 6, 8, 16):
    r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080) if world == 1 else r
    r.set_depth(8)
    r.set_partition(0, world)
    r.set_chains(chains)
    if os.environ.get('NOPROFILE'):
        r.enable_counters(False, False)
    r.restart(); r.step(16); r.wait_idle(); r.stats()
    s0 = r.stats(); n = 128
    t = time.time(); r.step(n); r.wait_idle(); dt = (time.time() - t) / n * 1e3
    s = r.stats()
    k = [(s.trace_closest_ms - s0.trace_closest_ms) / n, (s.shade_ms - s0.shade_ms) / n, (s.trace_shadow_ms - s0.trace_shadow_ms) / n]
    base = base or dt
    print("world %d: %.3f ms/launch wall (kernels %.3f + %.3f + %.3f = %.3f) -> predicted speed-up %.2fx, efficiency %.0f%%" % (
        world, dt, k[0], k[1], k[2], sum(k), base / dt, 100 * base / dt / world))
