"""Which company does a ray need to come out wrong?  python tools/gpu_fuzz_subset.py SEED FILE RAY  (FILE: "orc closest" lines, RAY: index in it)"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
import fuzz_scenes as fz

seed, ray = int(sys.argv[1]), int(sys.argv[3])
desc, run = fz.random_scene(seed)
inst = glaze_amd.RayTraceInstance.new()
inst.set_as_levels("flat")
sc = glaze_amd.RayTraceScene.from_desc(inst, desc)
fl = float.fromhex
lines = [l.split() for l in open(sys.argv[2]) if l.startswith("orc closest")]
o = np.array([[fl(x) for x in w[3:6]] for w in lines], np.float32)
d = np.array([[fl(x) for x in w[7:10]] for w in lines], np.float32)
wid = np.array([int(w[22]) for w in lines])
print("oracle: id", wid[ray], "t", lines[ray][16])
g0 = ray // 64 * 64
for lo, hi in ((ray, ray + 1), (g0, g0 + 64), (g0 - 64, g0 + 128), (0, len(o)), (0, g0 + 64), (g0, len(o))):
    lo, hi = max(lo, 0), min(hi, len(o))
    t, tri, _, _, _ = sc.debug_trace_closest(o[lo:hi], d[lo:hi], tmin=1e-4)
    bad = np.nonzero((tri != wid[lo:hi]) & np.isfinite(t))[0] + lo
    print("rays %5d .. %5d: ray %d -> id %d; wrong in this batch: %s" % (lo, hi, ray, tri[ray - lo], bad.tolist()[:12]))
# the ray's group of 64, 10 times: is it deterministic?
outs = []
for _ in range(10):
    t, tri, _, _, _ = sc.debug_trace_closest(o[g0:g0 + 64], d[g0:g0 + 64], tmin=1e-4)
    outs.append(int(tri[ray - g0]))
print("the ray's group of 64 alone, 10 times:", outs)
# greedy reduction: drop rays of the group while the ray still comes out wrong
keep = list(range(g0, min(g0 + 64, len(o))))
changed = True
while changed:
    changed = False
    for r in list(keep):
        if r == ray:
            continue
        trial = [k for k in keep if k != r]
        t, tri, _, _, _ = sc.debug_trace_closest(o[trial], d[trial], tmin=1e-4)
        if tri[trial.index(ray)] != wid[ray]:
            keep = trial
            changed = True
print("smallest company found (%d rays):" % len(keep), keep)
t, tri, _, _, _ = sc.debug_trace_closest(o[keep], d[keep], tmin=1e-4)
for k, i in enumerate(keep):
    print("   ray %d (pixel %s %s): o %s d %s -> hip t %r id %d | oracle id %d %s" % (i, lines[i][-2], lines[i][-1], o[i].tolist(), d[i].tolist(), float(t[k]), tri[k], wid[i], lines[i][16]))
