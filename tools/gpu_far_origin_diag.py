import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import glaze_amd
from test_gpu_two_level import instanced_cubes, scenes
from oracle.pyoracle import OracleScene
inst = glaze_amd.RayTraceInstance.new()
desc = instanced_cubes(120, seed=5, scale=(0.01, 0.05))
flat, two = scenes(inst, desc)
osc = OracleScene(desc)
rng = np.random.default_rng(8)
n = 20000
for far in (50.0, 200.0, 2000.0, 100000.0):
    target = rng.uniform(-0.9, 0.9, (n, 3)); dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o = (target - far * dirs).astype(np.float32); d = dirs.astype(np.float32)
    a, b = flat.debug_trace_closest(o, d), two.debug_trace_closest(o, d)
    t, tri = osc.trace_closest(o, d, brute=True)
    print("far", far, "flat!=brute", int((a[1] != tri).sum()), "two!=brute", int((b[1] != tri).sum()), "flat!=two", int((a[1] != b[1]).sum()), "t-bits flat!=two", int((a[0].view(np.uint32) != b[0].view(np.uint32)).sum()))
