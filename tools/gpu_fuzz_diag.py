"""Where does a scene of tools/gpu_fuzz_parity.py differ?  For every seed given: closest-hit and any-hit traces of the camera rays and of
random rays, HIP (flattened and two-level structure) against the oracle (its own hierarchy and brute force), then the rendered image
launch by launch -- the first launch and pixel at which the accumulators part.

    python tools/gpu_fuzz_diag.py 118,520,564
"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
from oracle.pyoracle import OracleRenderer, OracleScene
from helpers import camera_rays

import fuzz_scenes as fz


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


for seed in [int(v) for v in sys.argv[1].split(",")]:
    desc, run = fz.random_scene(seed)
    print("== seed %d: %dx%d spp %d depth %d %s, %d triangles, %d instances, materials %s, lights %s, camera %s" % (
        seed, run["w"], run["h"], run["spp"], run["depth"], run["integrator"].name, desc.n_triangles, len(desc.instances),
        [(m.mtype, m.opacity, m.normal) for m in desc.materials], [l.ltype for l in desc.lights], "ortho" if desc.camera.type else "persp"))
    osc = OracleScene(desc)
    rng = np.random.default_rng(1000 + seed)
    for levels in ("flat", "two_level"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_as_levels(levels)
        gsc = glaze_amd.RayTraceScene.from_desc(inst, desc)
        r = glaze_amd.RayTraceRenderer.new(inst, gsc, run["w"], run["h"])
        if desc.camera.type == 0:
            o, d = camera_rays(r.push_constants(), run["w"], run["h"])
        else:
            o, d = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
        ro = rng.uniform(-2.8, 2.8, (20000, 3)).astype(np.float32)
        rd = rng.standard_normal((20000, 3)).astype(np.float32)
        rd /= np.linalg.norm(rd, axis=1, keepdims=True)
        o, d = np.concatenate([o, ro]), np.concatenate([d, rd])
        gt, gtri, ginst, gu, gv = gsc.debug_trace_closest(o, d)
        ct, ctri, cinst, cu, cv = osc.trace_closest(o, d)
        bt, btri = osc.trace_closest(o, d, brute=True)
        bad = (bits(gt) != bits(ct)) | (gtri != ctri) | (bits(gu) != bits(cu)) | (bits(gv) != bits(cv))
        bad_b = (bits(ct) != bits(bt)) | (ctri != btri)
        bad_h = (bits(gt) != bits(bt)) | (gtri != btri)
        print("   %-9s closest: %d of %d rays differ from the oracle, %d from its brute force (oracle's hierarchy vs its brute force: %d)" % (
            levels, int(bad.sum()), len(o), int(bad_h.sum()), int(bad_b.sum())))
        bad = bad | bad_h
        for i in np.nonzero(bad)[0][:5]:
            print("      ray %d o %s d %s: hip t %r tri %d inst %d u %r v %r | oracle t %r tri %d inst %d u %r v %r | brute t %r tri %d" % (
                i, o[i].tolist(), d[i].tolist(), float(gt[i]), gtri[i], ginst[i], float(gu[i]), float(gv[i]), float(ct[i]), ctri[i], cinst[i], float(cu[i]), float(cv[i]), float(bt[i]), btri[i]))
        tmax = rng.uniform(0.1, 6.0, len(o)).astype(np.float32)
        ga, ca = gsc.debug_trace_any(o, d, tmax), osc.trace_any(o, d, tmax)
        print("   %-9s any-hit: %d of %d rays differ" % (levels, int((ga != ca).sum()), len(o)))
        for i in np.nonzero(ga != ca)[0][:5]:
            print("      ray %d o %s d %s tmax %r: hip %d oracle %d (closest t %r tri %d)" % (i, o[i].tolist(), d[i].tolist(), float(tmax[i]), ga[i], ca[i], float(ct[i]), ctri[i]))
    # launch by launch, both structures
    for levels in ("flat", "two_level"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_as_levels(levels)
        r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, desc), run["w"], run["h"])
        orc = OracleRenderer(osc, run["w"], run["h"])
        for x in (r, orc):
            x.set_integrator(run["integrator"] if x is r else run["integrator"].value)
            x.set_depth(run["depth"])
            x.set_seed(run["seed"])
        r.set_launch_mode("two_kernels")
        r.restart()
        orc.restart()
        n_launches = run["spp"] * r.steps_per_sample()
        for launch in range(1, n_launches + 1):
            r.step(1)
            orc.step(1)
            g, c = r.read_hdr(), orc.read_hdr()
            gr, cr = r.read_result(), orc.read_result()
            dh, dr = (bits(g) != bits(c)).any(-1), (bits(gr) != bits(cr)).any(-1)
            if dh.any() or dr.any():
                ys, xs = np.nonzero(dh | dr)
                print("   %-9s first difference after launch %d of %d: %d pixels (hdr %d, result %d)" % (levels, launch, n_launches, int((dh | dr).sum()), int(dh.sum()), int(dr.sum())))
                st = orc.read_state()
                for y, x in list(zip(ys, xs))[:6]:
                    print("      (x %d, y %d) hip %s -> %s | oracle %s -> %s\n         oracle state %s" % (x, y, g[y, x].tolist(), gr[y, x].tolist(), c[y, x].tolist(), cr[y, x].tolist(),
                                                                                          np.round(st[y, x], 5).tolist()))
                break
        else:
            print("   %-9s no difference launch by launch" % levels)
