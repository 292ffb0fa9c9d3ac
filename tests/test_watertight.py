"""The ray / triangle test is watertight: no ray passes between two triangles that share an edge.

The reference's hits come from traceRayEXT (lib/src/shaders/path_trace.rgen:169, :106-109) on the driver's acceleration
structure (lib/src/vulkan/acceleration.rs:319-345), and the Vulkan specification requires that intersector to be watertight.
The build's intersector (oracle.cpp ray_tri = kernels_render.hip ray_triangle) gets there by construction -- the edge function
of a shared edge is the same number with the opposite sign in the two triangles -- and these tests look for the leak that
Moeller-Trumbore, which rounds 1-2 ran, shows: rays aimed at vertices, at points of edges and along the diagonals of
pixel-aligned quads.  Oracle on the CPU; the HIP path on the GPU must give the same hits bit for bit.
"""
import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_camera, make_light, make_material, make_meta
from oracle.pyoracle import OracleRenderer, OracleScene


def scene_of(pos, tri, lights=None, camera=None):
    vertices = np.zeros(len(pos), VERTEX_DTYPE)
    vertices["vv"] = pos
    vertices["vn"] = (0.0, 0.0, -1.0)
    vertices["vt"] = 0.5
    tri = np.asarray(tri, np.uint32).reshape(-1)
    meshes = np.array([(0, 1, 0, tri.size)], MESH_DTYPE)
    instances = np.array([(0, 0)], INSTANCE_DTYPE)
    materials = [make_material("default"), make_material("wall", diffuse_mul=(200, 200, 200))]
    textures = [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default")]
    lights = lights if lights is not None else [make_light(abi.LIGHT_OMNI, "lamp", position=(0.25, 0.125, 0.0), intensity=1.0)]
    camera = camera or make_camera(position=(0, 0, 0), target=(0, 0, 100), up=(0, 1, 0), near=1e-3, far=100.0)
    return SceneDesc(vertices, tri, meshes, None, instances, materials, lights, textures, camera, make_meta((0, 0, 0), 4.0, 1.0))


def blob(n=20, wobble=0.25):
    """A closed, non-convex surface: the lattice points on the surface of the cube {0..n}^3 pushed onto a wobbly sphere.  A
    position is a function of its integer lattice point alone, so the faces of the cube meet in the SAME floats; every triangle
    edge is shared by exactly two triangles."""
    index, pos, tris = {}, [], []

    def vertex(p):
        if p not in index:
            q = np.array(p, np.float64) / n * 2.0 - 1.0
            q /= np.linalg.norm(q)
            r = 1.0 + wobble * np.sin(5.0 * q[0] + 1.0) * np.cos(4.0 * q[1] - 0.5) * np.sin(3.0 * q[2] + 2.0)
            index[p] = len(pos)
            pos.append((q * r * 0.75 + np.array([0.03, -0.02, 0.05])).astype(np.float32))
        return index[p]

    for axis in range(3):
        for side in (0, n):
            for i in range(n):
                for j in range(n):
                    corner = []
                    for di, dj in ((0, 0), (1, 0), (1, 1), (0, 1)):
                        p = [0, 0, 0]
                        p[axis] = side
                        p[(axis + 1) % 3] = i + di
                        p[(axis + 2) % 3] = j + dj
                        corner.append(vertex(tuple(p)))
                    a, b, c, d = corner
                    tris += [(a, b, c), (a, c, d)] if (i + j) % 2 else [(a, b, d), (b, c, d)]       # both diagonals occur
    return np.array(pos, np.float32), np.array(tris, np.uint32)


def edge_rays(pos, tris, origins, per_edge=6, seed=0):
    """rays from each origin through every vertex and through points of every edge (ends, middle, random places), targets rounded to float32"""
    rng = np.random.default_rng(seed)
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    e = np.unique(np.sort(e, axis=1), axis=0)
    assert len(e) * 2 == len(tris) * 3                                   # closed: every edge belongs to two triangles
    s = np.concatenate([np.array([0.0, 0.5, 1.0]), rng.random(per_edge - 3)])
    p, q = pos[e[:, 0]].astype(np.float64), pos[e[:, 1]].astype(np.float64)
    targets = (p[:, None, :] + s[None, :, None] * (q - p)[:, None, :]).reshape(-1, 3).astype(np.float32)
    o = np.repeat(np.asarray(origins, np.float32), len(targets), axis=0)
    d = np.tile(targets, (len(origins), 1)) - o
    return o, d.astype(np.float32)


INSIDE = [(0.0, 0.0, 0.0), (0.2, -0.15, 0.1), (-0.25, 0.1, -0.2), (0.05, 0.3, 0.25)]


def test_no_ray_leaves_a_closed_surface_through_its_edges():
    """From inside a closed mesh every ray hits it -- also the rays aimed exactly at vertices and at points of edges, where an
    intersector that evaluates each triangle's barycentrics independently falls between two triangles."""
    pos, tris = blob()
    sc = OracleScene(scene_of(pos, tris))
    o, d = edge_rays(pos, tris, INSIDE)
    assert len(o) > 100000
    t, tri, _, u, v = sc.trace_closest(o, d, tmin=1e-4)
    assert np.isfinite(t).all(), "%d rays leaked" % int((~np.isfinite(t)).sum())
    assert (t > 0.05).all() and (t < 2.5).all()
    assert (u >= 0).all() and (v >= 0).all() and (u + v <= 1.0 + 1e-6).all()
    # the BVH walk and the brute force over all triangles agree (boxes are conservative for the new test too)
    tb, trib = sc.trace_closest(o[::7], d[::7], tmin=1e-4, brute=True)
    assert np.array_equal(tb.view(np.uint32), t[::7].view(np.uint32)) and np.array_equal(trib, tri[::7])
    # occlusion rays from inside to far outside are all blocked
    far = np.full(len(o), 50.0, np.float32)
    assert sc.trace_any(o[::3], d[::3], far[::3]).all()
    # from outside, aimed at the same places on the side that faces the origin of the rays (a ray aimed at the silhouette may
    # rightly pass by): a hit, never a miss
    out = np.array([(2.0, 1.5, -1.0), (-1.5, 2.0, 1.0)], np.float32)
    o2, d2 = edge_rays(pos, tris, out, per_edge=4, seed=1)
    target = (o2 + d2).astype(np.float64)
    radial = target / np.linalg.norm(target, axis=1, keepdims=True)                  # the blob is star-shaped around the origin
    back = -d2 / np.linalg.norm(d2, axis=1, keepdims=True)
    facing = (radial * back).sum(-1) > 0.8                                         # the wobble tilts the surface against the radius by less than that
    assert facing.sum() > 1000
    assert np.isfinite(sc.trace_closest(o2[facing], d2[facing])[0]).all()


def tilted_wall(k=32):
    """k x k quads over [-1, 1]^2 on the plane z = 1 + x/4 + y/8: every coordinate is a dyadic rational, exact in float32"""
    g = np.arange(k + 1, dtype=np.float64) / k * 2.0 - 1.0
    x, y = np.meshgrid(g, g, indexing="ij")
    pos = np.stack([x, y, 1.0 + x / 4 + y / 8], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange((k + 1) * (k + 1), dtype=np.uint32).reshape(k + 1, k + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    alt = ((np.arange(k)[:, None] + np.arange(k)[None, :]) % 2).astype(bool)[..., None]
    tri = np.where(alt, np.stack([a, b, c, a, c, d], -1), np.stack([a, b, d, b, c, d], -1)).reshape(-1, 3)
    return pos, tri


def lattice_rays(k=32, sub=4):
    """parallel rays along +z through every point of a lattice `sub` times finer than the wall's: vertices, edges and both diagonals are met exactly"""
    g = np.arange(1, k * sub, dtype=np.float64) / (k * sub) * 2.0 - 1.0          # strictly inside the wall
    x, y = np.meshgrid(g, g, indexing="ij")
    o = np.stack([x, y, np.full_like(x, -1.0)], -1).reshape(-1, 3).astype(np.float32)
    d = np.broadcast_to(np.array([0.0, 0.0, 1.0], np.float32), o.shape).copy()
    return o, d


def test_parallel_rays_through_a_lattice_of_shared_edges_all_hit():
    pos, tri = tilted_wall()
    sc = OracleScene(scene_of(pos, tri))
    o, d = lattice_rays()
    t, _, _, u, v = sc.trace_closest(o, d)
    assert np.isfinite(t).all(), "%d of %d lattice rays leaked" % (int((~np.isfinite(t)).sum()), len(t))
    want = 2.0 + o[:, 0].astype(np.float64) / 4 + o[:, 1].astype(np.float64) / 8                                # distance from z = -1 to the plane
    assert np.abs(t - want).max() < 1e-5
    # a good share of them do meet an edge or a vertex exactly (a barycentric coordinate is exactly zero)
    on_edge = (u == 0) | (v == 0) | (u + v == 1)
    assert on_edge.mean() > 0.3


def ortho_wall():
    pos, tri = tilted_wall(64)
    cam = make_camera(position=(0, 0, 0), target=(0, 0, 100), up=(0, 1, 0), orthographic=True, scale=1.0, near=1e-3, far=100.0)
    return scene_of(pos, tri, camera=cam)


def test_oracle_orthographic_wall_has_no_leaked_pixel():
    """The verdict's scene: a pixel-aligned tessellated wall under the orthographic camera.  128 x 128 pixels over 64 x 64 quads: every
    pixel centre of the first launches (jitter 1/2, 1/4, 3/4) lies on a diagonal or an edge.  A leaked camera ray misses everything and
    its pixel stays exactly black; a leaked shadow ray cannot happen here (nothing between the wall and the lamp)."""
    o = OracleRenderer(OracleScene(ortho_wall()), 128, 128)
    o.set_integrator(abi.DIRECT)
    for launches in (1, 4):
        o.restart()
        o.step(launches)
        img = o.read_hdr()
        assert (img[..., 3] == launches).all()
        assert (img[..., :3].min(axis=-1) > 0).all(), "%d leaked pixels" % int((img[..., :3].min(axis=-1) <= 0).sum())


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_hits_on_shared_edges_equal_the_oracle(instance):
    pos, tris = blob()
    desc = scene_of(pos, tris)
    o, d = edge_rays(pos, tris, INSIDE)
    want = OracleScene(desc).trace_closest(o, d)
    for builder in ("sah", "lbvh"):
        instance.set_bvh_builder(builder)
        try:
            scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
        finally:
            instance.set_bvh_builder("auto")
        got = scene.debug_trace_closest(o, d)
        assert np.isfinite(got[0]).all(), builder
        for g, w in zip(got, want):
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), builder
        far = np.full(len(o), 50.0, np.float32)
        assert scene.debug_trace_any(o, d, far).all()
    pos, tri = tilted_wall()
    desc = scene_of(pos, tri)
    o, d = lattice_rays()
    got, want = glaze_amd.RayTraceScene.from_desc(instance, desc).debug_trace_closest(o, d), OracleScene(desc).trace_closest(o, d)
    assert np.isfinite(got[0]).all()
    for g, w in zip(got, want):
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


@pytest.mark.gpu
def test_hip_orthographic_wall_has_no_leaked_pixel(instance):
    desc = ortho_wall()
    for n in (128, 512):
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), n, n)
        r.set_integrator(glaze_amd.Integrator.DIRECT)
        r.step(4)
        img = r.read_hdr()
        assert (img[..., 3] == 4).all()
        assert (img[..., :3].min(axis=-1) > 0).all(), "%d leaked pixels at %d x %d" % (int((img[..., :3].min(axis=-1) <= 0).sum()), n, n)
        if n == 128:
            o = OracleRenderer(OracleScene(desc), n, n)
            o.set_integrator(abi.DIRECT)
            o.step(4)
            assert np.array_equal(img.view(np.uint32), o.read_hdr().view(np.uint32))


# ---------------------------------------------------------------------------------------------------------------------
# the intersector against mathematics (CPU): hit / miss, distance and barycentrics of random triangle soups against an independent
# float64 statement (the plane equation and edge functions of the exact triangle, nothing shared with oracle.cpp)
# ---------------------------------------------------------------------------------------------------------------------
def exact_hits(pos, tris, o, d):
    """closest intersection of each ray with the soup in float64: (t, triangle, u, v, margin) -- margin = how far inside the triangle the hit
    is (smallest barycentric), so that rays within rounding of an edge can be left out of the hit / miss comparison"""
    a, b, c = (pos[tris[:, k]].astype(np.float64) for k in range(3))
    n = np.cross(b - a, c - a)                                                    # (T, 3)
    o64, d64 = o.astype(np.float64), d.astype(np.float64)
    best_t = np.full(len(o), np.inf)
    best = np.full(len(o), -1)
    best_uv = np.zeros((len(o), 2))
    best_margin = np.full(len(o), np.inf)
    closest_edge = np.full(len(o), np.inf)
    for k in range(len(tris)):
        den = d64 @ n[k]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((a[k] - o64) @ n[k]) / den
        p = o64 + t[:, None] * d64
        nn = n[k] @ n[k]
        u = np.cross(p - a[k], c[k] - a[k]) @ n[k] / nn                           # weight of b: (p - a) x (c - a) = u n
        v = np.cross(b[k] - a[k], p - a[k]) @ n[k] / nn                           # weight of c
        w = 1.0 - u - v
        m = np.minimum(np.minimum(u, v), w)
        inside = (den != 0) & (m >= 0) & (t > 1e-4)
        closest_edge = np.where((den != 0) & (t > 1e-4), np.minimum(closest_edge, np.abs(m)), closest_edge)
        take = inside & (t < best_t)
        best_t = np.where(take, t, best_t)
        best = np.where(take, k, best)
        best_uv[take] = np.stack([u, v], -1)[take]
        best_margin = np.where(take, m, best_margin)
    return best_t, best, best_uv, best_margin, closest_edge


def test_intersector_matches_an_independent_float64_statement():
    rng = np.random.default_rng(11)
    nt = 60
    centre = rng.uniform(-1, 1, (nt, 3))
    pos = (centre[:, None, :] + rng.normal(0, 0.35, (nt, 3, 3))).reshape(-1, 3).astype(np.float32)      # a soup: nothing is shared
    tris = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
    sc = OracleScene(scene_of(pos, tris))
    n = 40_000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)   # not normalised: t is in units of d
    t, tri, _, u, v = sc.trace_closest(o, d)
    et, etri, euv, margin, edge = exact_hits(pos, tris, o, d)
    clear = edge > 1e-5                                                            # no triangle's edge within rounding of the ray
    hit, ehit = np.isfinite(t), np.isfinite(et)
    assert clear.mean() > 0.98 and ehit[clear].mean() > 0.15
    assert np.array_equal(hit[clear], ehit[clear])
    both = clear & hit
    assert np.array_equal(tri[both], etri[both].astype(np.uint32))
    assert np.abs(t[both] - et[both]).max() < 2e-5 * np.maximum(1.0, et[both]).max()
    assert np.abs(u[both] - euv[both, 0]).max() < 2e-5 and np.abs(v[both] - euv[both, 1]).max() < 2e-5
    # and the brute-force walk agrees with the hierarchy (boxes are conservative for soups too)
    tb, trib = sc.trace_closest(o[::5], d[::5], brute=True)
    assert np.array_equal(tb.view(np.uint32), t[::5].view(np.uint32)) and np.array_equal(trib, tri[::5])
