import sys, time
sys.path.insert(0, ".")
import numpy as np
import glaze_amd
from glaze_amd.scenes import atrium_scene
for detail in [float(a) for a in sys.argv[1:]] or (0.2, 0.564, 16.0):
    desc = atrium_scene(detail=detail, texture_size=64, sky_size=(256, 128))
    out = {}
    for b in ("sah", "sah_host"):
        inst = glaze_amd.RayTraceInstance.new(); inst.set_bvh_builder(b)
        sc = glaze_amd.RayTraceScene.from_desc(inst, desc); del sc
        t = time.time(); sc = glaze_amd.RayTraceScene.from_desc(inst, desc); dt = time.time() - t
        i = sc.info()
        out[b] = sc.debug_bvh()
        print(detail, b, "nodes %d depth %d sah %.2f build %.1f ms scene %.3f s" % (i.bvh_nodes, i.bvh_depth, i.bvh_sah_cost, i.build_ms, dt), flush=True)
    same = out["sah"][0].shape == out["sah_host"][0].shape and np.array_equal(out["sah"][0], out["sah_host"][0]) and np.array_equal(out["sah"][1].view(np.uint32), out["sah_host"][1].view(np.uint32))
    print("   identical trees:", same, flush=True)
