"""Why does the HIP tracer not find a triangle?  For one ray of a fuzz scene: every triangle of the flattened structure through a numpy
restatement of the triangle test, then the boxes on the way from the root to the leaf that holds the accepted one through a numpy
restatement of the slab test (device/wavefront.h box_key).
    python tools/gpu_fuzz_walk.py SEED ox oy oz dx dy dz tmax [tmin]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import glaze_amd
import fuzz_scenes as fz

f32 = np.float32


def num(s):
    return float.fromhex(s) if "x" in s.lower() else float(s)


seed = int(sys.argv[1])
v = [num(a) for a in sys.argv[2:9]]
tmin = f32(num(sys.argv[9])) if len(sys.argv) > 9 else f32(1e-3)
o, d, tmax = np.array(v[0:3], f32), np.array(v[3:6], f32), f32(v[6])
desc, run = fz.random_scene(seed)
inst = glaze_amd.RayTraceInstance.new()
inst.set_as_levels("flat")
scene = glaze_amd.RayTraceScene.from_desc(inst, desc)
nodes, tris = scene.debug_bvh()
info = scene.info()
glo, gcell = np.array(info.bvh_grid_lo, f32), np.array(info.bvh_grid_cell, f32)
print("nodes %d, triangle slots %d, grid lo %s cell %s" % (len(nodes), len(tris), glo, gcell))


def fma(a, b, c):
    return f32(np.float64(a) * np.float64(b) + np.float64(c))


def edge(a, b, c, d_):
    p, q = f32(a * b), f32(c * d_)
    e = f32(p - q)
    return e if e != 0 else f32(fma(a, b, -p) - fma(c, d_, -q))


kz = int(np.argmax(np.abs(d)))       # (ties: the first of the largest, as ray_shear has it)
kx, ky = (kz + 1) % 3, (kz + 2) % 3
sz = f32(1.0) / d[kz]
sx, sy = f32(d[kx] * sz), f32(d[ky] * sz)


def shear(p):
    a = (p - o).astype(f32)
    return np.array([fma(-sx, a[kz], a[kx]), fma(-sy, a[kz], a[ky]), f32(sz * a[kz])], f32)


accepted = []
for slot in range(len(tris)):
    t_ = tris[slot]
    A, B, C = shear(t_[0:3]), shear(t_[4:7]), shear(t_[8:11])
    U, V, W = edge(C[0], B[1], C[1], B[0]), edge(A[0], C[1], A[1], C[0]), edge(B[0], A[1], B[1], A[0])
    if (min(U, V, W) < 0) and (max(U, V, W) > 0):
        continue
    det = f32(f32(U + V) + W)
    if det == 0:
        continue
    inv = f32(1.0) / det
    t = f32(fma(W, C[2], fma(V, B[2], f32(U * A[2]))) * inv)
    if t > tmin and t < tmax:
        accepted.append((float(t), slot, int(t_[3:4].view(np.uint32)[0]), int(t_[7:8].view(np.uint32)[0])))
print("triangle test accepts (t, slot, world id, instance):", sorted(accepted))

# the grid-space ray
inv_cell = (f32(1.0) / gcell).astype(f32)
og = ((o - glo) * inv_cell).astype(f32)
idir = np.array([np.clip(f32(1.0) / x if x != 0 else np.inf * np.sign(1 / x), -1e30, 1e30) for x in d], f32)
ig = (idir * gcell).astype(f32)
cg = np.array([fma(f32(-32768.0), ig[k], -f32(og[k] * ig[k])) for k in range(3)], f32)
print("og", og, "ig", ig, "cg", cg)


def child_boxes(node):
    out = []
    for k in range(4):
        link = int(np.int32(node[12 + k]))
        w = node[3 * k:3 * k + 3]
        lo, hi = [int(x & 0xFFFF) for x in w], [int(x >> 16) for x in w]
        out.append((link, lo, hi))
    return out


def slab(lo, hi, bound):
    t0, t1 = tmin, bound
    detail = []
    for k in range(3):
        a, b = fma(f32(32768 + lo[k]), ig[k], cg[k]), fma(f32(32768 + hi[k]), ig[k], cg[k])
        n, f = (a, b) if ig[k] >= 0 else (b, a)
        detail.append((float(n), float(f)))
        t0, t1 = max(t0, n), min(t1, f)
    return t0 <= t1, float(t0), float(t1), detail


# parents
parent = {}
for i, nd in enumerate(nodes):
    for k, (link, lo, hi) in enumerate(child_boxes(nd)):
        if link != 0x7FFFFFFF:
            parent[link] = (i, k)
for t, slot, wid, ins in sorted(accepted):
    # the leaf link is ~(first slot of the leaf): this slot or the one before it
    link = ~slot if ~slot in parent else ~(slot - 1)
    chain = []
    while link in parent:
        chain.append(parent[link])
        link = parent[link][0]
        if link == 0:
            break
    print("triangle slot %d (world %d, t %r): boxes from the root down" % (slot, wid, t))
    for node, k in reversed(chain):
        lnk, lo, hi = child_boxes(nodes[node])[k]
        ok, t0, t1, detail = slab(lo, hi, tmax)
        wl = [float(glo[a] + gcell[a] * lo[a]) for a in range(3)]
        wh = [float(glo[a] + gcell[a] * hi[a]) for a in range(3)]
        print("   node %d child %d link %d: grid box %s .. %s = world %s .. %s -> %s (entry %r exit %r; per axis %s)" % (node, k, lnk, lo, hi, np.round(wl, 5).tolist(), np.round(wh, 5).tolist(), "pass" if ok else "FAIL", t0, t1, detail))
    tr = tris[slot]
    print("   vertices", tr[0:3].tolist(), tr[4:7].tolist(), tr[8:11].tolist(), "hit point", (o + d * f32(t)).tolist())
