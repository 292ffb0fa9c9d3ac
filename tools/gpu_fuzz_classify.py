"""What kind of difference is it?  For each seed: render both sides, take the first pixel that differs, have a child process render the
oracle again with ORC_DEBUG_PIXEL set to it, replay the oracle's rays of that pixel through the HIP tracer's debug hooks (the structure
the run used), and for every ray whose outcome differs look at the triangle involved: the angle between ray and plane, how far the
origin is off the plane, whether the reported point lies inside the triangle's bounds.  "phantom" = the class of DESIGN.md section 3.
    python tools/gpu_fuzz_classify.py 223644,230234,..."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
from fuzz_scenes import random_scene, render_both
from oracle.pyoracle import OracleScene

fl = float.fromhex


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


def triangle_of(desc, world_id):
    bases = np.cumsum([0] + [int(desc.meshes[x["mesh_id"]]["index_count"]) // 3 for x in desc.instances])
    i = int(np.searchsorted(bases, world_id, side="right") - 1)
    m = desc.meshes[desc.instances[i]["mesh_id"]]
    p = world_id - bases[i]
    ix = desc.indices[int(m["index_offset"]) + 3 * p:int(m["index_offset"]) + 3 * p + 3]
    M = desc.transforms[desc.instances[i]["transform_id"]].reshape(4, 4).T.astype(np.float64)
    return (M @ np.concatenate([desc.vertices["vv"][ix].astype(np.float64), np.ones((3, 1))], 1).T).T[:, :3]


def describe(desc, o, d, t, world_id):
    V = triangle_of(desc, world_id)
    n = np.cross(V[1] - V[0], V[2] - V[0])
    ln = np.linalg.norm(n)
    if ln == 0:
        return "degenerate triangle", True
    n /= ln
    p = o.astype(np.float64) + d.astype(np.float64) * t
    ext = float(np.max(V.max(0) - V.min(0)))
    outside = float(np.max(np.maximum(V.min(0) - p, p - V.max(0))))
    sin_angle, off = abs(float(d @ n)), abs(float((o - V[0]) @ n))
    phantom = sin_angle < 3e-3 and outside > 1e-4 * ext
    return "ray-to-plane sine %.1e, origin %.1e off the plane, reported point %.1e outside the bounds (extent %.2g)" % (sin_angle, off, max(outside, 0.0), ext), phantom


verdicts = []
for seed in [int(v) for v in sys.argv[1].split(",")]:
    desc, run = random_scene(seed)
    r, o = render_both(desc, run)
    g, c = r.read_hdr(), o.read_hdr()
    diff = (bits(g) != bits(c)).any(-1) | (bits(r.read_result()) != bits(o.read_result())).any(-1)
    if not diff.any():
        print("seed %d: identical" % seed)
        continue
    ys, xs = np.nonzero(diff)
    x, y = int(xs[0]), int(ys[0])
    code = "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); from fuzz_scenes import random_scene, render_oracle; d, r = random_scene(%d); render_oracle(d, r)" % seed
    child = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORC_DEBUG_PIXEL="%d,%d" % (x, y)), capture_output=True, text=True)
    lines = [l.split() for l in child.stderr.splitlines() if l.startswith("orc closest") or l.startswith("orc shadow")]
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_as_levels(run["levels"])
    inst.set_bvh_builder(run["builder"])
    sc = glaze_amd.RayTraceScene.from_desc(inst, desc)
    osc = OracleScene(desc)
    found, all_phantom = [], True
    for w in lines:
        oo, dd = np.array([[fl(v) for v in w[3:6]]], np.float32), np.array([[fl(v) for v in w[7:10]]], np.float32)
        if w[1] == "closest":
            t, tri, _, _, _ = sc.debug_trace_closest(oo, dd, tmin=1e-4)
            want_valid, want_t, want_id = int(w[14]), np.float32(fl(w[16])), int(w[22])
            same = (int(np.isfinite(t[0])) == want_valid) and (not want_valid or (t[0] == want_t and int(tri[0]) == want_id))
            if not same:
                for who, tt, tid in (("oracle", want_t, want_id), ("hip", t[0], int(tri[0]))):
                    if np.isfinite(tt) and tid != 0xFFFFFFFF:
                        text, ph = describe(desc, oo[0], dd[0], float(tt), tid)
                        found.append("closest ray: %s hits triangle %d at t %.6g -- %s" % (who, tid, float(tt), text))
                        all_phantom &= ph
        else:
            tmax, want = np.array([fl(w[11])], np.float32), int(w[14])
            got = int(sc.debug_trace_any(oo, dd, tmax, tmin=1e-3)[0])
            if got != want:
                t, tri = osc.trace_closest(oo, dd, tmin=1e-3, brute=True) if want else sc.debug_trace_closest(oo, dd, tmin=1e-3)[:2]
                text, ph = describe(desc, oo[0], dd[0], float(t[0]), int(tri[0])) if np.isfinite(t[0]) else ("nothing along the ray?", False)
                found.append("shadow ray (tmax %.4g): oracle occluded %d, hip %d; the occluder, triangle %d at t %.6g -- %s" % (float(tmax[0]), want, got, int(tri[0]), float(t[0]), text))
                all_phantom &= ph
    verdict = "phantom hit of a triangle seen edge-on" if found and all_phantom else ("NO differing ray among the oracle's %d rays of the pixel: the paths part elsewhere" % len(lines) if not found else "NOT (only) phantoms")
    verdicts.append((seed, verdict))
    print("seed %d (%s / %s / %s), pixel (%d, %d): %s" % (seed, run["mode"], run["levels"], run["builder"], x, y, verdict))
    for f in found[:4]:
        print("     " + f)
print("summary:", ", ".join("%d: %s" % v for v in verdicts))
