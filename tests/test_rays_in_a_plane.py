"""A ray that lies in the plane of a triangle hits nothing (oracle.cpp ray_tri, device/wavefront.h triangle_finish: a candidate whose
det is not above 2^-19 of its products is rounding noise, not a hit) -- stated on a designed scene: a tessellated floor with a raised
step on it, (a) in an axis-aligned plane near coordinate 0 (translated and scaled only), (b) rotated.
  * shadow rays from a point of the floor to other points of the floor (what sampling a light that lies in the floor's plane
    produces): on the axis-aligned floor they are not occluded by the floor's own triangles, whatever walks the scene -- the case the
    randomised runs found, where every box is flat and whether such a candidate was even offered depended on the structure.  On the
    ROTATED floor origin and direction carry their own rounding (the ray is 1e-8 ... 1e-6 out of the plane), some of those rays do
    report a coplanar triangle, and what is required is that everybody reports the same: hierarchy, walk over all triangles, HIP;
  * rays that graze the floor at 1e-3 rad towards the step still hit the step at the distance a float64 computation gives;
  * rays that come down onto the floor at 1e-3 rad still hit it.
CPU: the oracle's hierarchy and its walk over all triangles; GPU: the HIP tracer, both structures, against them."""
import numpy as np
import pytest

import glaze_amd
from fuzz_scenes import col_major, rot
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc
from oracle.pyoracle import OracleScene


def scene(rotated):
    n = 6
    s, t = np.meshgrid(np.linspace(-1, 1, n + 1), np.linspace(-1, 1, n + 1), indexing="ij")
    floor = np.stack([s, np.zeros_like(s), t], -1).reshape(-1, 3)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    ftri = np.stack([a, b, c, a, c, d], -1).reshape(-1)
    step = np.array([[0.5, 0.0, -0.3], [0.5, 0.0, 0.3], [0.5, 0.25, 0.3], [0.5, 0.25, -0.3]])      # a wall standing on the floor, facing -x
    stri = np.array([0, 1, 2, 0, 2, 3]) + len(floor)
    pos = np.concatenate([floor, step])
    v = np.zeros(len(pos), VERTEX_DTYPE)
    v["vv"], v["vn"], v["vt"] = pos, (0, 1, 0), pos[:, [0, 2]]
    meshes = np.array([(0, 0, 0, len(ftri)), (1, 0, len(ftri), len(stri))], MESH_DTYPE)
    xf = np.eye(4)
    xf[:3, 3] = (0.31, -0.002, -0.17)
    xf = xf @ (rot(0, 17.0) @ rot(2, -8.0) if rotated else np.eye(4)) @ np.diag([0.9, 1.3, 1.1, 1.0])
    desc = SceneDesc(v, np.concatenate([ftri, stri]).astype(np.uint32), meshes, np.stack([col_major(np.eye(4)), col_major(xf)]),
                     np.array([(0, 1), (1, 1)], INSTANCE_DTYPE))
    return desc, xf


def to_world(xf, p):
    return (xf @ np.concatenate([p, np.ones((len(p), 1))], 1).T).T[:, :3]


def rays(rotated):
    desc, xf = scene(rotated)
    rng = np.random.default_rng(3)
    n = 20000
    # (a) in the plane: from a point of the floor to another one, away from the step's foot
    p0 = np.stack([rng.uniform(-0.95, 0.4, n), np.zeros(n), rng.uniform(-0.95, 0.95, n)], 1)
    p1 = np.stack([rng.uniform(-0.95, 0.4, n), np.zeros(n), rng.uniform(-0.95, 0.95, n)], 1)
    o_a, e_a = to_world(xf, p0).astype(np.float32), to_world(xf, p1)
    d_a = e_a - o_a
    len_a = np.linalg.norm(d_a, axis=1)
    d_a = (d_a / len_a[:, None]).astype(np.float32)
    # (b) grazing towards the step: start 1e-3 rad above the floor line that ends at 40 % of the step's height
    q0 = np.stack([rng.uniform(-0.9, 0.2, n), np.zeros(n), rng.uniform(-0.25, 0.25, n)], 1)
    q1 = np.stack([np.full(n, 0.5), np.full(n, 0.1), rng.uniform(-0.25, 0.25, n)], 1)
    q0[:, 1] = q1[:, 1] - 1e-3 * np.linalg.norm(q1 - q0, axis=1)
    o_b = to_world(xf, q0).astype(np.float32)
    d_b = to_world(xf, q1) - o_b
    t_b = np.linalg.norm(d_b, axis=1)
    d_b = (d_b / t_b[:, None]).astype(np.float32)
    # (c) down onto the floor at 1e-3 rad
    r1 = np.stack([rng.uniform(-0.9, 0.3, n), np.zeros(n), rng.uniform(-0.9, 0.9, n)], 1)
    r0 = r1 + np.stack([-rng.uniform(0.3, 0.6, n), np.zeros(n), rng.uniform(-0.1, 0.1, n)], 1)
    r0[:, 1] = 1e-3 * np.linalg.norm(r1 - r0, axis=1)
    o_c = to_world(xf, r0).astype(np.float32)
    d_c = to_world(xf, r1) - o_c
    t_c = np.linalg.norm(d_c, axis=1)
    d_c = (d_c / t_c[:, None]).astype(np.float32)
    return desc, (o_a, d_a, (0.98 * len_a).astype(np.float32)), (o_b, d_b, t_b), (o_c, d_c, t_c)


def check(trace_any, trace_closest, who, rotated):
    desc, (o_a, d_a, tmax_a), (o_b, d_b, t_b), (o_c, d_c, t_c) = rays(rotated)
    occluded = trace_any(o_a, d_a, tmax_a)
    if not rotated:
        assert not occluded.any(), who + ": a ray in the floor's plane was occluded by the floor"
    t, tri = trace_closest(o_b, d_b)
    assert (tri >= 72).all() and np.allclose(t, t_b, rtol=2e-4), who + ": grazing rays towards the step"
    t, tri = trace_closest(o_c, d_c)
    hit = np.isfinite(t)
    assert hit.mean() > 0.999 and (tri[hit] < 72).all() and np.allclose(t[hit], t_c[hit], rtol=2e-2), who + ": rays coming down onto the floor at 1e-3 rad"
    return occluded


@pytest.mark.parametrize("rotated", [False, True])
def test_oracle_rays_in_a_plane(rotated):
    osc = OracleScene(rays(rotated)[0])
    a = check(lambda o, d, tm: osc.trace_any(o, d, tm), lambda o, d: osc.trace_closest(o, d)[:2], "oracle hierarchy", rotated)
    b = check(lambda o, d, tm: osc.trace_any(o, d, tm), lambda o, d: osc.trace_closest(o, d, brute=True), "oracle brute force", rotated)
    assert np.array_equal(a, b)
    for o, d, _ in rays(rotated)[1:]:
        t, tri = osc.trace_closest(o, d)[:2]
        bt, btri = osc.trace_closest(o, d, brute=True)
        assert np.array_equal(t.view(np.uint32), bt.view(np.uint32)) and np.array_equal(tri[np.isfinite(bt)], btri[np.isfinite(bt)])


@pytest.mark.gpu
@pytest.mark.parametrize("rotated", [False, True])
@pytest.mark.parametrize("levels", ["flat", "two_level"])
def test_hip_rays_in_a_plane(levels, rotated):
    desc, a, b, c = rays(rotated)
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_as_levels(levels)
    sc = glaze_amd.RayTraceScene.from_desc(inst, desc)
    occluded = check(lambda o, d, tm: sc.debug_trace_any(o, d, tm), lambda o, d: sc.debug_trace_closest(o, d)[:2], "hip " + levels, rotated)
    osc = OracleScene(desc)
    assert np.array_equal(occluded, osc.trace_any(a[0], a[1], a[2]))
    for o, d, _ in (a, b, c):
        t, tri = sc.debug_trace_closest(o, d)[:2]
        bt, btri = osc.trace_closest(o, d, brute=True)
        assert np.array_equal(t.view(np.uint32), bt.view(np.uint32)) and np.array_equal(tri[np.isfinite(bt)], btri[np.isfinite(bt)])
