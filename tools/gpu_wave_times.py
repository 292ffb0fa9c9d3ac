"""When the waves of one k_trace finish (variant built with -DGLZ_WAVE_TIMES): share of the frame WORLD, one chain."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd.scenes import atrium_scene
lib = ctypes.CDLL(os.environ["GLAZE_HIP_LIB"])
inst = glaze_amd.RayTraceInstance.new()
scene = glaze_amd.RayTraceScene.from_desc(inst, atrium_scene())
r = glaze_amd.RayTraceRenderer.new(inst, scene, 1920, 1080)
r.set_depth(8)
r.enable_counters(False, False)
for world in [int(w) for w in os.environ.get("WORLDS", "8,1").split(",")]:
    r.set_partition(0, world)
    r.set_chains(1)
    r.restart(); r.step(24); r.wait_idle()
    for rep in range(3):
        r.step(int(os.environ.get("LAUNCHES", "1"))); r.wait_idle()   # LAUNCHES=2: the last k_trace also traces the shadow rays of the launch before
        t = np.zeros(3 * 8192, dtype=np.uint64)
        assert lib.glz_debug_wave_times(t.ctypes.data_as(ctypes.c_void_p), 8192) == 0
        t = t.reshape(-1, 3).astype(np.int64)
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0   # wall_clock64: 100 MHz
        end = np.sort(us[:, 2])
        print("world %d rep %d: %d waves, start spread %.1f us, closest done p50/p90/p99/max %.1f/%.1f/%.1f/%.1f us, end p10/p50/p75/p90/p95/p99/max %.1f/%.1f/%.1f/%.1f/%.1f/%.1f/%.1f us" % (
            world, rep, len(t), us[:, 0].max(), *np.percentile(us[:, 1], [50, 90, 99, 100]), *np.percentile(end, [10, 50, 75, 90, 95, 99, 100])), flush=True)
        # what sharing work inside a block (4 consecutive waves) could reach at best: the slowest block's mean against the slowest wave
        work = (us[:, 2] - us[:, 0])
        nb = len(work) // 4
        blk = work[:nb * 4].reshape(nb, 4)
        print("   slowest wave %.1f us; slowest block mean %.1f us (its waves %s); blocks of 8 waves: %.1f us" % (
            work.max(), blk.mean(axis=1).max(), np.round(blk[blk.mean(axis=1).argmax()], 1), work[:(len(work) // 8) * 8].reshape(-1, 8).mean(axis=1).max()), flush=True)
        st = np.zeros(8 * 8192, dtype=np.uint32)
        if hasattr(lib, "glz_debug_wave_stats") and lib.glz_debug_wave_stats(st.ctypes.data_as(ctypes.c_void_p), 8192) == 0:
            st = st.reshape(-1, 8).astype(np.float64)[: len(us)]
            dur = us[:, 1] - us[:, 0]
            busy = dur > 5.0
            order = np.argsort(dur)
            def row(name, sel):
                x = st[sel]
                print("   %-14s %5d waves: %.1f us, rounds %.0f, node iterations %.0f with %.1f lanes, leaf iterations %.0f with %.1f lanes, rounds with helpers %.0f, rounds only waiting for helpers %.0f, us per node iteration %.3f" % (
                    name, len(x), dur[sel].mean(), x[:, 0].mean(), x[:, 1].mean(), x[:, 2].sum() / max(x[:, 1].sum(), 1), x[:, 3].mean(), x[:, 4].sum() / max(x[:, 3].sum(), 1),
                    x[:, 5].mean(), x[:, 6].mean(), dur[sel].sum() / max(x[:, 1].sum(), 1)), flush=True)
            nb = int(busy.sum())
            ob = order[-nb:]
            row("all", ob)
            row("fastest tenth", ob[: nb // 10])
            row("median tenth", ob[nb // 2 - nb // 20: nb // 2 + nb // 20])
            row("slowest tenth", ob[-(nb // 10):])
            row("slowest 1 %", ob[-max(nb // 100, 1):])
        # where the slow waves are: by XCD (consecutive blocks go to consecutive XCDs) and by position in the grid (= in the image: wave w
        # traces the groups w, w + n_waves, ...)
        dur = us[:, 1] - us[:, 0]
        blk_id = np.arange(len(dur)) // 4
        print("   closest phase by XCD (block %% 8): %s us" % " ".join("%.0f" % dur[blk_id % 8 == x].mean() for x in range(8)), flush=True)
        print("   closest phase by sixteenth of the grid: %s us" % " ".join("%.0f" % c.mean() for c in np.array_split(dur, 16)), flush=True)
        # waves still running as a function of time
        grid = np.linspace(0, end[-1], 11)
        print("   running at", " ".join("%.0fus:%d" % (g, int(((us[:, 0] <= g) & (us[:, 2] > g)).sum())) for g in grid), flush=True)
