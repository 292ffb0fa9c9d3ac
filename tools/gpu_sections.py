"""Where a wave's time goes inside trace_wave's round -- refill, hand-overs to idle lanes (share), the node visit, loop control (ballots,
quorum checks), the leaf phase, merge + retire -- in shader clocks, for k_trace at N = 1 and for k_path / k_trace on a 1/8 share.
Needs a library built with -DGLZ_SECTION_TIMES (tools/build_variant.sh sections -DGLZ_SECTION_TIMES):
    GLAZE_HIP_LIB=variants/libglaze_hip_sections.so python tools/gpu_sections.py"""
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
L = abi.lib()
NAMES = ["refill", "share", "node visit", "control", "leaf", "merge+retire"]


def read(fn):
    buf = np.zeros((8192, 16), np.uint64)
    assert fn(buf.ctypes.data_as(C.c_void_p), 1) == 0
    return buf.astype(np.float64)


for world, mode in ((1, "two_kernels"), (8, "two_kernels"), (8, "path"), (32, "path")):
    r.set_partition(0, world); r.set_launch_mode(mode)
    r.restart(); r.step(32); r.wait_idle()
    fn = L.glz_debug_sections_path if mode == "path" else L.glz_debug_sections_trace
    read(fn)                                        # reset
    n = 16
    r.step(n); r.wait_idle()
    b = read(fn)
    used = b[:, 10] > 0
    b = b[used]
    merges, merge_clk = b[:, 11].sum(), b[:, 12].sum()
    fine = b[:, 13:16].sum(axis=0)
    if fine.sum() > 0:     # -DGLZ_SECTION_TIMES=2: the node visit in pieces (their clocks are not in `node visit` then)
        print("      node visit in pieces, clocks per iteration: wait for the node %.0f, box tests + sort %.0f, links through LDS %.0f, pushes / pop + the rest %.0f" % (
            fine[0] / b[:, 7].sum(), fine[1] / b[:, 7].sum(), fine[2] / b[:, 7].sum(), b[:, 2].sum() / b[:, 7].sum()))
        b[:, 2] += b[:, 13:16].sum(axis=1)
    b[:, 5] += b[:, 12]          # (merge is stamped separately from retire)
    clocks = b[:, :6].sum()
    per = b[:, :6].sum(axis=0)
    rounds, iters, leaves, takes = b[:, 6].sum(), b[:, 7].sum(), b[:, 8].sum(), b[:, 9].sum()
    wave_total = b[:, :6].sum(axis=1) / n
    print("world %2d %-11s: %d waves, %.0f clocks per wave and launch (p90 %.0f, max %.0f); per wave and launch: %.1f rounds, %.1f node iterations, %.1f leaf phases, %.1f hand-over steps" % (
        world, mode, used.sum(), wave_total.mean(), np.percentile(wave_total, 90), wave_total.max(), rounds / used.sum() / n, iters / used.sum() / n, leaves / used.sum() / n, takes / used.sum() / n))
    print("      share of the time: " + ", ".join("%s %.1f %%" % (NAMES[k], 100 * per[k] / clocks) for k in range(6)))
    print("      merge: %.1f helper results per wave and launch, %.0f clocks each, %.1f %% of the time" % (merges / used.sum() / n, merge_clk / max(merges, 1), 100 * merge_clk / clocks))
    print("      clocks per occurrence: node visit %.0f, share step %.0f (per node iteration), control %.0f (per node iteration), leaf phase %.0f, refill %.0f and merge+retire %.0f per round" % (
        per[2] / iters, per[1] / iters, per[3] / iters, per[4] / max(leaves, 1), per[0] / rounds, per[5] / rounds), flush=True)
