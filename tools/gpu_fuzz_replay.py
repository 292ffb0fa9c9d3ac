"""The rays the oracle traced for one pixel (ORC_DEBUG_PIXEL=x,y ... 2> file, lines "orc closest ..." / "orc shadow ...") through the HIP
tracer's debug hooks, both structures:   python tools/gpu_fuzz_replay.py SEED FILE"""
import re
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
import fuzz_scenes as fz

seed = int(sys.argv[1])
desc, run = fz.random_scene(seed)
scenes = {}
for levels in ("flat", "two_level"):
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_as_levels(levels)
    scenes[levels] = glaze_amd.RayTraceScene.from_desc(inst, desc)
fl = float.fromhex
for line in open(sys.argv[2]):
    w = line.split()
    if w[:2] == ["orc", "closest"]:
        o, d = np.array([[fl(x) for x in w[3:6]]], np.float32), np.array([[fl(x) for x in w[7:10]]], np.float32)
        want = (int(w[14]), fl(w[16]), int(w[22]))     # valid, t, world id
        for levels, sc in scenes.items():
            t, tri, inst, u, v = sc.debug_trace_closest(o, d, tmin=1e-4)
            got = (int(np.isfinite(t[0])), float(t[0]), int(tri[0]))
            same = got[0] == want[0] and (not want[0] or (got[1] == np.float32(want[1]) and got[2] == want[2]))
            if not same:
                print("CLOSEST %-9s differs: oracle valid %d t %r id %d | hip valid %d t %r id %d inst %d   ray o %s d %s" % (levels, want[0], want[1], want[2], got[0], got[1], got[2], inst[0], w[3:6], w[7:10]))
    elif w[:2] == ["orc", "shadow"]:
        o, d = np.array([[fl(x) for x in w[3:6]]], np.float32), np.array([[fl(x) for x in w[7:10]]], np.float32)
        tmax, want = np.array([fl(w[11])], np.float32), int(w[14])
        for levels, sc in scenes.items():
            got = int(sc.debug_trace_any(o, d, tmax, tmin=1e-3)[0])
            if got != want:
                t, tri, inst, u, v = sc.debug_trace_closest(o, d, tmin=1e-3)
                print("SHADOW  %-9s differs: oracle occluded %d | hip %d (hip closest along it: t %r id %d; tmax %r)   ray o %s d %s tmax %s" % (levels, want, got, float(t[0]), tri[0], float(tmax[0]), w[3:6], w[7:10], w[11]))
# all closest-hit rays of the file again, in ONE call per structure: the lanes of a wave then work on neighbouring rays, as in a render
lines = [l.split() for l in open(sys.argv[2]) if l.startswith("orc closest")]
if len(lines) > 1:
    o = np.array([[fl(x) for x in w[3:6]] for w in lines], np.float32)
    d = np.array([[fl(x) for x in w[7:10]] for w in lines], np.float32)
    valid = np.array([int(w[14]) for w in lines])
    wt = np.array([fl(w[16]) for w in lines], np.float32)
    wid = np.array([int(w[22]) for w in lines])
    for levels, sc in scenes.items():
        t, tri, inst, u, v = sc.debug_trace_closest(o, d, tmin=1e-4)
        bad = (np.isfinite(t) != (valid == 1)) | ((valid == 1) & ((t.view(np.uint32) != wt.view(np.uint32)) | (tri != wid)))
        print("batch of %d closest rays, %-9s: %d differ" % (len(lines), levels, int(bad.sum())))
        for i in np.nonzero(bad)[0][:8]:
            print("   ray %d %s: oracle valid %d t %r id %d | hip t %r id %d" % (i, " ".join(lines[i][-3:]), valid[i], float(wt[i]), wid[i], float(t[i]), tri[i]))
print("replayed", sys.argv[2])
