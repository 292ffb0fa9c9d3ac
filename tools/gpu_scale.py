"""Scene-size sweep on one MI355X: atrium at growing tessellation -> build time, BVH shape, throughput, and a builder cross-check
(LBVH and PLOC must give bit-identical images: hits never depend on the hierarchy).  Usage: python tools/gpu_scale.py [details...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import glaze_amd
from glaze_amd.scenes import atrium_scene

details = [float(a) for a in sys.argv[1:]] or [0.564, 4.0, 16.0, 48.0]
W, H, DEPTH, N = 1920, 1080, 8, 32
for det in details:
    desc = atrium_scene(detail=det, texture_size=256)
    ntri = sum(int(x) for x in desc.meshes["index_count"]) // 3
    imgs = {}
    for builder in ("lbvh", "ploc", "sah"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_bvh_builder(builder)
        scene = glaze_amd.RayTraceScene.from_desc(inst, desc); del scene   # first build of a process pays module load
        t = time.time(); scene = glaze_amd.RayTraceScene.from_desc(inst, desc)
        t_scene = time.time() - t
        i = scene.info()
        info = "nodes %d depth %d sah %.1f build %.1f ms" % (i.bvh_nodes, i.bvh_depth, i.bvh_sah_cost, i.build_ms)
        r = glaze_amd.RayTraceRenderer.new(inst, scene, W, H)
        r.set_depth(DEPTH); r.set_seed(7)
        r.step(8); r.wait_idle()
        t = time.time(); r.step(N); r.wait_idle(); dt = time.time() - t
        img = r.read_result()
        imgs[builder] = img
        print("detail %6.2f %9d tris %-4s | scene %.3f s | %s | %7.1f Msamples/s" % (det, ntri, builder, t_scene, info, W * H * N / dt / 1e6), flush=True)
        del r, scene, inst
    a, b = imgs["lbvh"], imgs["ploc"]
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    print("   builders agree bit for bit:", bool(same.all()), flush=True)
