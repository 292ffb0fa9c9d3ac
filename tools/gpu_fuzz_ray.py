"""One ray of a fuzz scene (tests/fuzz_scenes.py) through the HIP tracer's debug hooks and through the oracle:
    python tools/gpu_fuzz_ray.py SEED ox oy oz dx dy dz tmax      (floats as decimal or C99 hexadecimal literals, e.g. what ORC_DEBUG_PIXEL prints)
"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
import fuzz_scenes as fz
from oracle.pyoracle import OracleScene


def num(s):
    return float.fromhex(s) if "x" in s.lower() else float(s)


seed = int(sys.argv[1])
v = [num(a) for a in sys.argv[2:9]]
o, d, tmax = np.array([v[0:3]], np.float32), np.array([v[3:6]], np.float32), np.array([v[6]], np.float32)
desc, run = fz.random_scene(seed)
osc = OracleScene(desc)
for tmin in (1e-3, 1e-4):
    t, tri, inst, u, vv = osc.trace_closest(o, d, tmin=tmin)
    print("oracle     tmin %g: closest t %r (%s) tri %d inst %d u %r v %r; any-hit(tmax %r) %d" % (tmin, float(t[0]), float(t[0]).hex(), tri[0], inst[0], float(u[0]), float(vv[0]), float(tmax[0]),
                                                                                   osc.trace_any(o, d, tmax, tmin=tmin)[0]))
    for levels in ("flat", "two_level"):
        inst_ = glaze_amd.RayTraceInstance.new()
        inst_.set_as_levels(levels)
        g = glaze_amd.RayTraceScene.from_desc(inst_, desc)
        t, tri, inst, u, vv = g.debug_trace_closest(o, d, tmin=tmin)
        print("hip %-9s tmin %g: closest t %r (%s) tri %d inst %d u %r v %r; any-hit(tmax %r) %d" % (levels, tmin, float(t[0]), float(t[0]).hex(), tri[0], inst[0], float(u[0]), float(vv[0]), float(tmax[0]),
                                                                                     g.debug_trace_any(o, d, tmax, tmin=tmin)[0]))
