"""Experiment: two launch chains at N = 1 with k_trace's persistent grid capped (GLAZE_TRACE_BLOCKS_PER_CU) so that the other chain's
k_shade blocks fit next to it -- the VALU-bound and the memory-bound kernel of different half frames running side by side."""
import os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8)
out = []
for chains in (1, 2, 3, 4):
    r.set_chains(chains); r.restart(); r.step(16); r.wait_idle()
    n = 128
    t = time.time(); r.step(n); r.wait_idle(); dt = (time.time() - t) / n * 1e3
    out.append("%d chains %.4f" % (chains, dt))
print(" | ".join(out))
'''
for cap in ("", "5", "4", "3", "2"):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    if cap:
        env["GLAZE_TRACE_BLOCKS_PER_CU"] = cap
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("k_trace blocks per CU %-8s %s" % (cap or "6 (all)", out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
