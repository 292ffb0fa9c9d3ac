"""Per-launch cost of one rank's share of the 1080p frame at world sizes 1..16, measured on ONE GPU (rank 0's tiles), for both
launch modes (two kernels per launch / the per-wave launch loop k_path).  Predicts the strong scaling of bench.py --gpus N before
the 8-GPU run: speed-up(N) ~ t(1) / t(N), against the best mode at world 1.

    python tools/gpu_partition_timing.py [two_kernels|path|auto ...]        env: CHAINS=<n> NOPROFILE=1 STEPS=<n>
                                                                            WIDTH HEIGHT DEPTH WORLDS=1,2,4,8 (BASELINE configs[4]: 3840 2160 12) NODE_WIDTH=0|4|8
"""
import os
import sys
import time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
chains = int(os.environ.get('CHAINS', '0'))
n = int(os.environ.get('STEPS', '128'))
modes = sys.argv[1:] or ["two_kernels", "path", "auto"]
W, H, D = int(os.environ.get('WIDTH', '1920')), int(os.environ.get('HEIGHT', '1080')), int(os.environ.get('DEPTH', '8'))
worlds = [int(w) for w in os.environ.get('WORLDS', '1,2,3,4,6,8,16').split(',')]
print("frame %d x %d, depth %d, %d launches per call" % (W, H, D, n), flush=True)
r = glaze_amd.RayTraceRenderer.new(inst, scene, W, H)
r.set_depth(D)
base = None
for mode in modes:
    for world in worlds:
        if mode == "path" and world == 1 and os.environ.get("SKIP_PATH_1", "1") == "1":
            continue            # the full frame through k_path: 8 groups per wave one after the other, of no interest
        r.set_partition(0, world)
        r.set_launch_mode(mode)
        r.set_chains(chains)
        r.set_node_width(int(os.environ.get('NODE_WIDTH', '0')))
        if os.environ.get('NOPROFILE'):
            r.enable_counters(False, False)
        r.restart(); r.step(16); r.wait_idle(); r.stats()
        s0 = r.stats()
        t = time.time(); r.step(n); r.wait_idle(); dt = (time.time() - t) / n * 1e3
        s = r.stats()
        k = [(s.trace_closest_ms - s0.trace_closest_ms) / n, (s.shade_ms - s0.shade_ms) / n, (s.trace_shadow_ms - s0.trace_shadow_ms) / n, (s.other_ms - s0.other_ms) / n]
        base = base or dt
        print("%-11s world %2d (%s, %d-wide nodes): %.4f ms/launch wall (k_trace %.3f + k_shade %.3f + shadow pass %.3f + k_path %.3f = %.3f) -> speed-up %.2fx, efficiency %.0f%%" % (
            mode, world, r.launch_mode(), r.node_width(), dt, k[0], k[1], k[2], k[3], sum(k), base / dt, 100 * base / dt / world), flush=True)
