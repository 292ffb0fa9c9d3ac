"""The GPU SAH builder against its host reference on random scenes (tests/fuzz_scenes.py): the node arrays must be identical, and every
builder's structure must hold every triangle inside the boxes on the way down to it.
    python tools/gpu_fuzz_builders.py [first_seed] [count]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
from fuzz_scenes import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = []
for seed in range(first, first + count):
    desc, run = random_scene(seed)
    out = {}
    for builder in ("sah", "sah_host", "lbvh", "ploc"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_as_levels("flat")
        inst.set_bvh_builder(builder)
        sc = glaze_amd.RayTraceScene.from_desc(inst, desc)
        nodes, tris = sc.debug_bvh()
        info = sc.info()
        out[builder] = (nodes.copy(), tris.copy(), np.array(info.bvh_grid_lo, np.float32), np.array(info.bvh_grid_cell, np.float32))
    ok = np.array_equal(out["sah"][0], out["sah_host"][0]) and np.array_equal(out["sah"][1].view(np.uint32), out["sah_host"][1].view(np.uint32))
    # containment: every triangle's vertices inside every box on its path (grid coordinates), for each builder
    contained = True
    for builder, (nodes, tris, glo, gcell) in out.items():
        if len(nodes) == 0:
            continue
        stack = [(0, np.zeros(3), np.full(3, 32767.0))]
        seen = 0
        while stack:
            i, plo, phi = stack.pop()
            nd = nodes[i]
            for k in range(4):
                link = int(np.int32(nd[12 + k]))
                if link == 0x7FFFFFFF:
                    continue
                w = nd[3 * k:3 * k + 3]
                lo, hi = (w & 0xFFFF).astype(np.float64), (w >> 16).astype(np.float64)
                if link >= 0:
                    stack.append((link, lo, hi))
                else:
                    slot = ~link
                    n_in_leaf = 2 if (int(tris[slot, 11:12].view(np.uint32)[0]) & 0x40000000) else 1
                    for s in range(slot, slot + n_in_leaf):
                        seen += 1
                        for v in (tris[s, 0:3], tris[s, 4:7], tris[s, 8:11]):
                            g = (v.astype(np.float64) - glo) / gcell
                            if not (np.all(g >= lo - 1e-3) and np.all(g <= hi + 1e-3)):
                                contained = False
        if seen != len(tris):
            contained = False
    print("seed %d: %d triangles, sah == sah_host %s, containment %s" % (seed, len(out["sah"][1]), ok, contained), flush=True)
    if not (ok and contained):
        bad.append(seed)
print("%d scenes, %d bad%s" % (count, len(bad), (": " + " ".join(map(str, bad))) if bad else ""))
sys.exit(1 if bad else 0)
