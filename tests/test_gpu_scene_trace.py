"""GPU parity, scene build + ray tracing level: HIP path (through the C ABI) vs the CPU oracle.

Bit-exact unless stated: derivatives, RT material/light/sky tables, push constants, closest-hit
records (t, triangle, u, v) and any-hit results.
"""
import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scenes import atrium_scene, cube_scene
from oracle.pyoracle import OracleRenderer, OracleScene

from conftest import MATTEST
from helpers import camera_rays, desc_from_oracle_parse

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cube(instance):
    desc = cube_scene()
    return desc, glaze_amd.RayTraceScene.from_desc(instance, desc), OracleScene(desc)


@pytest.fixture(scope="module")
def mattest(instance):
    desc = desc_from_oracle_parse(MATTEST)                       # oracle reader -> oracle scene
    gpu = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))   # product reader -> product scene
    return desc, gpu, OracleScene(desc)


def test_scene_info(cube, mattest):
    i = cube[1].info()
    assert (i.n_vertices, i.n_triangles, i.n_world_triangles, i.n_instances, i.n_materials, i.n_lights) == (24, 12, 12, 1, 3, 1)
    i = mattest[1].info()
    # SURVEY F10 / BASELINE config 1
    assert (i.n_vertices, i.n_triangles, i.n_world_triangles) == (70876, 138480, 138480)
    assert (i.n_instances, i.n_materials, i.n_textures, i.n_lights, i.n_rt_lights) == (3, 5, 3, 1, 1)
    assert 138480 // 6 <= i.bvh_nodes <= 138479 and 8 <= i.bvh_depth <= 48      # leaves hold one or two triangles


@pytest.mark.parametrize("which", ["cube", "mattest"])
def test_derivatives_bit_exact(which, cube, mattest):
    _, gpu, orc = cube if which == "cube" else mattest
    a, b = gpu.debug_derivatives(), orc.derivatives()
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("which", ["cube", "mattest"])
def test_rt_tables_bit_exact(which, cube, mattest):
    _, gpu, orc = cube if which == "cube" else mattest
    assert np.array_equal(gpu.debug_rt_materials(), orc.rt_materials())
    assert np.array_equal(gpu.debug_rt_lights(), orc.rt_lights())
    a, b = gpu.debug_sky(), orc.sky()
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def mattest_by_builder(mattest):
    """The same scene built by every acceleration-structure builder (glz_instance_set_bvh_builder), each on its own instance."""
    out = {}
    for name in ("lbvh", "ploc", "sah", "sah_host"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_bvh_builder(name)
        out[name] = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(MATTEST))
    return out


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah"])
def test_bvh_structure(builder, mattest_by_builder):
    gpu = mattest_by_builder[builder]
    nodes, tris = gpu.debug_bvh()
    info = gpu.info()
    n, n4 = tris.shape[0], nodes.shape[0]
    assert nodes.shape == (info.bvh_nodes, 16) and nodes.dtype == np.uint32                                  # 64-byte 4-wide nodes
    ids = tris.view(np.uint32)[:, 3]
    assert np.array_equal(np.sort(ids), np.arange(n, dtype=np.uint32))         # every world triangle is in the leaf array exactly once
    EMPTY = 0x7FFFFFFF
    links = nodes[:, 12:16].view(np.int32).astype(np.int64)                    # (n4, 4)
    empty = links == EMPTY
    leaf = links < 0
    inner = ~empty & ~leaf
    assert (~empty[:, :2]).all()                                               # at least two children, packed to the front
    assert ((~empty).astype(int)[:, :-1] >= (~empty).astype(int)[:, 1:]).all()
    assert np.array_equal(np.sort(links[inner]), np.arange(1, n4))             # every inner node but the root has one parent
    # a leaf is one triangle or two adjacent ones (bit 30 of the first one's flags); leaf links point at the first:
    # the linked slots and their partners cover the triangle array exactly once
    first = np.sort(~links[leaf])
    partner = (tris.view(np.uint32)[:, 11] & 0x40000000) != 0
    covered = np.sort(np.concatenate([first, first[partner[first]] + 1]))
    assert np.array_equal(covered, np.arange(n)) and not partner[first[partner[first]] + 1].any()
    n_leaves = first.size
    assert n_leaves // 3 <= n4 <= max(1, n_leaves - 1) and n_leaves >= n // 2
    # both triangles of a pair belong to one instance, follow each other in its index buffer and share an edge
    p0 = first[partner[first]]
    assert (tris.view(np.uint32)[p0, 7] == tris.view(np.uint32)[p0 + 1, 7]).all()
    assert ((tris.view(np.uint32)[p0 + 1, 11] & 0x3FFFFFFF) == (tris.view(np.uint32)[p0, 11] & 0x3FFFFFFF) + 1).all()

    u = lambda x, hi: ((x >> 16) if hi else (x & 0xFFFF)).astype(np.int64)
    w = nodes[:, :12].reshape(n4, 4, 3)
    lo = np.stack([u(w[..., 0], 0), u(w[..., 1], 0), u(w[..., 2], 0)], -1)     # (n4, 4, 3) grid units; one word per axis: lo | hi << 16
    hi = np.stack([u(w[..., 0], 1), u(w[..., 1], 1), u(w[..., 2], 1)], -1)
    assert (lo[~empty] <= hi[~empty]).all()
    # the box stored for an inner child contains every box stored in that child (quantisation only grows boxes, and
    # the same world box is quantised to the same grid cell everywhere)
    big = np.iinfo(np.int64).max
    node_lo = np.where(empty[..., None], big, lo).min(1)
    node_hi = np.where(empty[..., None], -1, hi).max(1)
    ch = links[inner]
    assert (lo[inner] <= node_lo[ch]).all() and (hi[inner] >= node_hi[ch]).all()
    # a leaf's quantised box contains the triangle's vertices
    glo, cell = np.array(info.bvh_grid_lo, np.float64), np.array(info.bvh_grid_cell, np.float64)
    v0 = tris[:, 0:3].astype(np.float64)
    t = ~links[leaf]
    for slot, sel in ((t, np.ones(t.size, bool)), (np.minimum(t + 1, n - 1), partner[t])):   # the leaf's first triangle, then its partner
        for v in (v0, tris[:, 4:7].astype(np.float64), tris[:, 8:11].astype(np.float64)):      # the records hold the three vertices
            q = (v[slot[sel]] - glo) / cell
            assert (lo[leaf][sel] <= q + 1e-6).all() and (hi[leaf][sel] >= q - 1e-6).all()
    assert 2 <= info.bvh_depth <= 48


def _check_closest(gpu, orc, o, d, tmin=1e-4):
    t, tri, inst, u, v = gpu.debug_trace_closest(o, d, tmin)
    t2, tri2, inst2, u2, v2 = orc.trace_closest(o, d, tmin)
    same = (t.view(np.uint32) == t2.view(np.uint32)) & (tri == tri2) & (inst == inst2) & \
           (u.view(np.uint32) == u2.view(np.uint32)) & (v.view(np.uint32) == v2.view(np.uint32))
    return same, (t, tri), (t2, tri2)


def test_closest_hit_cube(cube):
    _, gpu, orc = cube
    rng = np.random.default_rng(1)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = (rng.random((20000, 3)).astype(np.float32) - 0.5) * 1.5
    same, (t, tri), _ = _check_closest(gpu, orc, o, d)
    assert same.all()
    assert np.isfinite(t).all()            # origin inside a closed box: every ray hits


def test_closest_hit_mattest_camera_and_random(mattest, instance):
    desc, gpu, orc = mattest
    push = np.zeros(32, np.float32)
    abi.check(abi.lib().glz_host_push_constants(__import__("ctypes").byref(desc.camera), 256, 256, push.ctypes.data))
    o, d = camera_rays(push, 256, 256)
    same, (t, tri), (t2, tri2) = _check_closest(gpu, orc, o, d)
    assert same.all(), "mismatching rays: %d" % (~same).sum()
    assert np.isfinite(t).mean() > 0.5
    # incoherent rays from points inside the scene bounds
    rng = np.random.default_rng(2)
    i = gpu.info()
    lo, hi = np.array(i.bounds_min), np.array(i.bounds_max)
    o = (lo + rng.random((50000, 3)) * (hi - lo)).astype(np.float32)
    d = rng.normal(size=(50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    same, _, _ = _check_closest(gpu, orc, o, d)
    assert same.all(), "mismatching rays: %d" % (~same).sum()


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah"])
def test_builders_same_hits(builder, mattest, mattest_by_builder):
    """Hits must not depend on the acceleration structure: every builder's scene vs the oracle."""
    _, _, orc = mattest
    gpu = mattest_by_builder[builder]
    rng = np.random.default_rng(7)
    i = gpu.info()
    assert 138480 // 6 <= i.bvh_nodes <= 138479 and 8 <= i.bvh_depth <= 48      # leaves hold one or two triangles
    lo, hi = np.array(i.bounds_min), np.array(i.bounds_max)
    o = (lo + rng.random((50000, 3)) * (hi - lo)).astype(np.float32)
    d = rng.normal(size=(50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    same, _, _ = _check_closest(gpu, orc, o, d)
    assert same.all(), "mismatching rays: %d" % (~same).sum()
    tmax = (rng.random(50000) * 3.0).astype(np.float32)
    assert np.array_equal(gpu.debug_trace_any(o, d, tmax), orc.trace_any(o, d, tmax))


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah"])
def test_tiny_scenes(builder):
    """1, 2 and 3 triangles: the builders' smallest cases."""
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_bvh_builder(builder)
    for ntri in (1, 2, 3):
        desc = cube_scene()
        desc.indices = desc.indices[: 3 * ntri].copy()
        desc.meshes["index_count"][0] = 3 * ntri
        gpu, orc = glaze_amd.RayTraceScene.from_desc(inst, desc), OracleScene(desc)
        rng = np.random.default_rng(ntri)
        d = rng.normal(size=(4000, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = (rng.random((4000, 3)).astype(np.float32) - 0.5) * 1.5
        same, (t, _), _ = _check_closest(gpu, orc, o, d)
        assert same.all() and np.isfinite(t).any()


@pytest.mark.parametrize("builder", ["lbvh", "ploc", "sah"])
def test_triangle_soup_with_duplicates_and_degenerates(builder):
    """Random soup: exact duplicates (equal Morton codes -> the index tie-break of the hierarchy), zero-area and needle triangles,
    one triangle spanning the whole scene, clusters far apart (deep, unbalanced tree).  Hits must equal the oracle's."""
    from glaze_amd.scene_desc import MESH_DTYPE, VERTEX_DTYPE, SceneDesc
    rng = np.random.default_rng(11)
    n = 60000
    centres = np.concatenate([rng.normal(size=(n // 2, 3)) * 0.3, rng.normal(size=(n // 2, 3)) * 0.02 + [40.0, -25.0, 10.0]]).astype(np.float32)
    tri = centres[:, None, :] + (rng.normal(size=(n, 3, 3)) * 0.05).astype(np.float32)
    tri[100:2100] = tri[100]                                     # 2000 exact copies of one triangle
    tri[3000:3500, 2] = tri[3000:3500, 1]                        # zero area (two equal vertices)
    tri[4000:4200, 1] = tri[4000:4200, 0] + np.float32(1e-7)     # needles
    tri[5000] = [[-60, -60, -3], [60, -60, -3], [0, 90, -3]]     # one huge triangle under everything
    verts = np.zeros((n * 3, 8), np.float32)
    verts[:, :3] = tri.reshape(-1, 3)
    verts[:, 3:6] = [0, 0, 1]
    verts[:, 6:8] = rng.random((n * 3, 2))
    desc = cube_scene()
    desc = SceneDesc(verts.view(VERTEX_DTYPE).reshape(-1), np.arange(n * 3, dtype=np.uint32), np.array([(0, 1, 0, n * 3)], MESH_DTYPE), None, desc.instances, desc.materials,
                     desc.lights, desc.textures, desc.camera, desc.meta)
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_bvh_builder(builder)
    gpu, orc = glaze_amd.RayTraceScene.from_desc(inst, desc), OracleScene(desc)
    i = gpu.info()
    assert i.n_world_triangles == n and i.bvh_nodes >= n // 6
    m = 30000
    o = np.concatenate([rng.normal(size=(m // 2, 3)) * 0.5, rng.normal(size=(m // 2, 3)) * 0.1 + [40.0, -25.0, 10.0]]).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50] = [0, 0, -1]                                           # straight down onto the big triangle
    same, (t, _), _ = _check_closest(gpu, orc, o, d)
    assert same.all(), "mismatching rays: %d" % (~same).sum()
    assert np.isfinite(t[:50]).all() and 0.01 < np.isfinite(t).mean() < 1.0      # the straight-down rays all land on the big triangle
    tmax = (rng.random(m) * 5.0).astype(np.float32)
    assert np.array_equal(gpu.debug_trace_any(o, d, tmax), orc.trace_any(o, d, tmax))


def test_sah_build_is_deterministic_and_default(mattest, mattest_by_builder):
    """The GPU SAH builder (one launch per level, node ids fixed by the ranges) gives the same array every time, it is what
    'auto' (the default) builds, and it is the tree of the host reference implementation node for node."""
    inst = glaze_amd.RayTraceInstance.new()
    inst.set_bvh_builder("sah")
    again = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(MATTEST))
    n0, t0 = mattest_by_builder["sah"].debug_bvh()
    n1, t1 = again.debug_bvh()
    assert np.array_equal(n0, n1) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    n2, _ = mattest[1].debug_bvh()                              # built by the default instance
    assert np.array_equal(n0, n2)
    n3, t3 = mattest_by_builder["sah_host"].debug_bvh()
    assert np.array_equal(n0, n3) and np.array_equal(t0.view(np.uint32), t3.view(np.uint32))
    assert mattest_by_builder["sah"].info().bvh_sah_cost < mattest_by_builder["lbvh"].info().bvh_sah_cost


@pytest.mark.parametrize("case", ["atrium", "equal centroids", "tiny"])
def test_gpu_sah_builder_equals_its_host_reference(case):
    """k_sah_level restates bvh_sah.cpp statement for statement (bins, candidate order, tie-breaks, stable partition): the two
    builders must emit identical node and triangle arrays -- also where binning finds no split and ranges are halved."""
    from glaze_amd.scene_desc import MESH_DTYPE, VERTEX_DTYPE, SceneDesc
    if case == "atrium":
        from glaze_amd.scenes import atrium_scene
        descs = [atrium_scene(detail=0.2, texture_size=16, sky_size=(64, 32))]
    elif case == "equal centroids":
        rng = np.random.default_rng(5)
        n = 3000
        tri = np.tile((rng.normal(size=(1, 3, 3)) * 0.3).astype(np.float32), (n, 1, 1))      # n copies of one triangle ...
        tri[n // 2:] += (rng.normal(size=(n - n // 2, 1, 3)) * 2.0).astype(np.float32)        # ... and n/2 scattered ones
        verts = np.zeros((n * 3, 8), np.float32)
        verts[:, :3] = tri.reshape(-1, 3)
        verts[:, 3:6] = [0, 0, 1]
        base = cube_scene()
        descs = [SceneDesc(verts.view(VERTEX_DTYPE).reshape(-1), np.arange(n * 3, dtype=np.uint32), np.array([(0, 1, 0, n * 3)], MESH_DTYPE), None,
                           base.instances, base.materials, base.lights, base.textures, base.camera, base.meta)]
    else:
        descs = []
        for ntri in (1, 2, 3, 5, 12):
            d = cube_scene()
            d.indices = d.indices[: 3 * ntri].copy()
            d.meshes["index_count"][0] = 3 * ntri
            descs.append(d)
    for desc in descs:
        out = []
        for b in ("sah", "sah_host"):
            inst = glaze_amd.RayTraceInstance.new()
            inst.set_bvh_builder(b)
            out.append(glaze_amd.RayTraceScene.from_desc(inst, desc).debug_bvh())
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))


def test_non_finite_vertices_do_not_break_the_builders():
    """NaN and infinite vertices (a broken export): every builder must finish, and the triangles that are finite must be hit
    exactly as with any other builder (triangles with a non-finite vertex can never pass the intersection test)."""
    from glaze_amd.scene_desc import MESH_DTYPE, VERTEX_DTYPE, SceneDesc
    rng = np.random.default_rng(3)
    n = 20000
    tri = (rng.normal(size=(n, 1, 3)) * 2 + rng.normal(size=(n, 3, 3)) * 0.05).astype(np.float32)
    tri[::97, 0, 0] = np.nan
    tri[5::131, 1] = np.inf
    tri[7::211] = -np.inf
    verts = np.zeros((n * 3, 8), np.float32)
    verts[:, :3] = tri.reshape(-1, 3)
    verts[:, 3:6] = [0, 0, 1]
    base = cube_scene()
    desc = SceneDesc(verts.view(VERTEX_DTYPE).reshape(-1), np.arange(n * 3, dtype=np.uint32), np.array([(0, 1, 0, n * 3)], MESH_DTYPE), None,
                     base.instances, base.materials, base.lights, base.textures, base.camera, base.meta)
    o = (rng.normal(size=(20000, 3)) * 2).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ref = None
    for b in ("lbvh", "ploc", "sah", "sah_host"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_bvh_builder(b)
        t, tri_id, _, _, _ = glaze_amd.RayTraceScene.from_desc(inst, desc).debug_trace_closest(o, d)
        assert 1000 < np.isfinite(t).sum() < 20000
        if ref is None:
            ref = (t, tri_id)
        assert np.array_equal(t.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(tri_id, ref[1]), b


def test_transform_memory_layout_kat_on_device(instance):
    """geometry/mesh.rs:110-119 (column-major Transform = row-major 3x4 Vulkan transform): world-space leaf of the flattened instance."""
    from test_oracle_kats import LAYOUT_KAT_WORLD, _layout_kat_scene
    gpu = glaze_amd.RayTraceScene.from_desc(instance, _layout_kat_scene())
    _, tris = gpu.debug_bvh()
    w = LAYOUT_KAT_WORLD
    assert tris.shape[0] == 1
    assert np.array_equal(tris[0, 0:3], w[0]) and np.array_equal(tris[0, 4:7], w[1]) and np.array_equal(tris[0, 8:11], w[2])


def test_oracle_bvh_against_brute_force(mattest):
    """The oracle's own BVH must agree with testing every triangle (checks the checker)."""
    _, _, orc = mattest
    rng = np.random.default_rng(3)
    o = (rng.random((300, 3)).astype(np.float32) - 0.5) * 2.0 + np.array([0, 1, 1], np.float32)
    d = rng.normal(size=(300, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t, tri, _, _, _ = orc.trace_closest(o, d)
    tb, trib = orc.trace_closest(o, d, brute=True)
    assert np.array_equal(t.view(np.uint32), tb.view(np.uint32)) and np.array_equal(tri, trib)


def test_any_hit_mattest(mattest):
    _, gpu, orc = mattest
    rng = np.random.default_rng(4)
    i = gpu.info()
    lo, hi = np.array(i.bounds_min), np.array(i.bounds_max)
    o = (lo + rng.random((50000, 3)) * (hi - lo)).astype(np.float32)
    d = rng.normal(size=(50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = (rng.random(50000) * 3.0).astype(np.float32)
    a = gpu.debug_trace_any(o, d, tmax)
    b = orc.trace_any(o, d, tmax)
    assert np.array_equal(a, b)
    assert 0.05 < a.mean() < 0.95


def test_axis_aligned_and_degenerate_rays(cube):
    """Slab-parallel rays (0 * inf in the box test), rays along edges and zero directions must not diverge."""
    _, gpu, orc = cube
    o = np.array([[0, 0, 0]] * 6 + [[1, 1, 0], [0.999999, 0.999999, 0], [0, 0, 0], [0, 0, 0]], np.float32)
    d = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [0, 0, 1], [0, 0, 1], [0, 0, 0],
                  [1, 1, 1]], np.float32)
    same, _, _ = _check_closest(gpu, orc, o, d)
    assert same.all()


def test_empty_scene_traces_nothing(instance):
    desc = cube_scene()
    desc.instances = desc.instances[:0]
    gpu = glaze_amd.RayTraceScene.from_desc(instance, desc)
    t, tri, _, _, _ = gpu.debug_trace_closest(np.zeros((4, 3), np.float32), np.array([[0, 0, 1]] * 4, np.float32))
    assert np.isinf(t).all() and (tri == 0xFFFFFFFF).all()
