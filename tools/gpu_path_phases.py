"""Where a wave of k_path spends its time: tracing (closest hits + the previous launch's shadow rays) against shading, per launch, on
rank 0's share of the 1080p atrium.  Needs a library built with -DGLZ_PATH_TIMES (tools/build_variant.sh times -DGLZ_PATH_TIMES):
    GLAZE_HIP_LIB=variants/libglaze_hip_times.so python tools/gpu_path_phases.py"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd import abi
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8)
for world in (8, 16, 32):
    r.set_partition(0, world); r.set_launch_mode("path")
    r.restart(); r.step(32); r.wait_idle()
    r.step(16); r.wait_idle()                       # one batch of 16 launches: what the counters hold
    n_waves = min(8192, (r.packed_pixels(0, world) + 63) // 64)
    buf = np.zeros((8192, 3), np.uint64)
    assert abi.lib().glz_debug_path_times(buf.ctypes.data_as(C.c_void_p), 8192) == 0
    t = buf[:n_waves].astype(np.float64) * 0.01 / 16      # us per launch
    q = lambda a: "mean %.1f p50 %.1f p90 %.1f max %.1f" % (a.mean(), np.median(a), np.percentile(a, 90), a.max())
    print("world %2d, %d waves, us per launch and wave: trace %s | shade %s | whole kernel / 16: %s" % (world, n_waves, q(t[:, 0]), q(t[:, 1]), q(t[:, 2])), flush=True)
    # where does the spread between waves live?  waves 4b .. 4b+3 are one block (one CU, four SIMDs); blocks b, b + 8, .. share an XCD
    tot = t[:n_waves - n_waves % 4, 2].reshape(-1, 4)
    blk = tot.mean(axis=1)
    xcd = [blk[x::8].mean() for x in range(8)]
    print("          std over waves %.1f us; std of block means %.1f, mean std inside a block %.1f; XCD means %s" % (
        tot.std(), blk.std(), tot.std(axis=1).mean(), " ".join("%.0f" % x for x in xcd)), flush=True)
