"""GPU parity, render level: the HIP wavefront path tracer vs the CPU oracle on the same scene, seed,
depth and launch sequence.

Tolerance (stated per BASELINE north_star "pixels within 1e-4 rel"): the oracle and the kernels use
the same deterministic arithmetic (-ffp-contract=off, include/glz_detmath.h), so the cumulative
RGBA32F image is expected to be BIT-EXACT; the assertions allow rel-err <= 1e-4 on >= 99.9 % of the
pixels and require the launch counter (alpha) to match exactly.
"""
import ctypes

import numpy as np
import pytest
from PIL import Image as PILImage

import glaze_amd
import glaze_amd.distributed
from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, make_camera, make_light, make_material
from glaze_amd.scenes import atrium_scene, cube_scene
from oracle.pyoracle import OracleRenderer, OracleScene

from conftest import MATTEST
from helpers import desc_from_oracle_parse, rel_err

pytestmark = pytest.mark.gpu


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)

REL_TOL = 1e-4


def render_both(instance, desc, w, h, spp, depth=6, seed=0, integrator=glaze_amd.Integrator.PATH_TRACE, gpu_scene=None):
    scene = gpu_scene if gpu_scene is not None else glaze_amd.RayTraceScene.from_desc(instance, desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, w, h)
    r.set_integrator(integrator)
    r.set_depth(depth)
    r.set_seed(seed)
    img = r.draw(spp)
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_integrator(integrator.value)
    o.set_depth(depth)
    o.set_seed(seed)
    o.draw(spp)
    return r, o, img


def assert_parity(r, o, name=""):
    g, c = r.read_hdr(), o.read_hdr()
    assert np.array_equal(g[..., 3], c[..., 3]), name + ": launch counters differ"
    nan_g, nan_c = np.isnan(g[..., :3]).any(-1), np.isnan(c[..., :3]).any(-1)
    assert np.array_equal(nan_g, nan_c), name + ": NaN pixels differ"
    ok = ~nan_c
    err = rel_err(g[..., :3][ok], c[..., :3][ok])
    frac = float((err.max(-1) <= REL_TOL).mean()) if err.size else 1.0
    exact = float(np.mean(g.view(np.uint32) == c.view(np.uint32)))
    assert frac >= 0.999, "%s: only %.4f%% of pixels within %g (bit-exact %.4f%%)" % (name, 100 * frac, REL_TOL, 100 * exact)
    mean_rel = abs(float(g[..., :3][ok].mean()) - float(c[..., :3][ok].mean())) / max(1e-12, abs(float(c[..., :3][ok].mean())))
    assert mean_rel <= 1e-5, name + ": mean image differs by %g" % mean_rel
    gr, cr = r.read_result(), o.read_result()
    err = rel_err(gr[..., :3][ok], cr[..., :3][ok])
    assert float((err.max(-1) <= REL_TOL).mean()) >= 0.999, name + ": result image (out32) differs"
    return exact


def test_cube_lambert_config2(instance):
    """BASELINE config 2 at reduced size: cube, Lambert, omni light, depth 2."""
    r, o, img = render_both(instance, cube_scene(), 128, 128, spp=8, depth=2)
    exact = assert_parity(r, o, "cube")
    assert exact > 0.999
    assert img.shape == (128, 128, 4) and img[..., 3].min() == 255 and img[..., :3].max() > 10
    assert np.array_equal(img, o.read_rgba8())                            # 8-bit sRGB export: byte work is bit-exact


def test_cube_direct_integrator(instance):
    r, o, _ = render_both(instance, cube_scene(), 96, 64, spp=4, integrator=glaze_amd.Integrator.DIRECT)
    assert r.steps_per_sample() == 1
    assert_parity(r, o, "cube direct")
    assert r.read_hdr()[..., 3].max() == 4.0


@pytest.mark.parametrize("mtype", [abi.MAT_LAMBERT, abi.MAT_MIRROR, abi.MAT_GLASS, abi.MAT_METAL, abi.MAT_FROSTED, abi.MAT_UBER])
def test_cube_every_bsdf(instance, mtype):
    desc = cube_scene(material_type=mtype)
    desc.materials[2].roughness_mul = 0.35
    desc.materials[2].metalness_mul = 0.6
    desc.materials[2].metal = 2
    desc.materials[2].anisotropy = 0.2
    r, o, _ = render_both(instance, desc, 64, 64, spp=6, depth=6, seed=mtype)
    assert_parity(r, o, "cube mtype %d" % mtype)


@pytest.mark.parametrize("mtype", [abi.MAT_METAL, abi.MAT_FROSTED, abi.MAT_UBER])
def test_roughness_and_metalness_textures(instance, mtype):
    """Gray roughness / metalness maps next to the sRGB diffuse map (material.rs:17-342: every scalar is texture x multiplier):
    the three material textures are fetched once per hit and shared by the NEE evaluation and the BSDF sample."""
    desc = cube_scene(material_type=mtype)
    rng = np.random.default_rng(mtype)
    desc.textures.append((abi.TEX_GRAY, rng.integers(20, 256, (32, 48), dtype=np.uint8), "rough"))     # not square, not a power of two
    desc.textures.append((abi.TEX_GRAY, rng.integers(0, 256, (64, 64), dtype=np.uint8), "metal"))
    m = desc.materials[2]
    m.roughness, m.metalness, m.roughness_mul, m.metalness_mul, m.metal, m.anisotropy = 2, 3, 0.8, 0.9, 1, 0.1
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.3, -0.6, 0.5), intensity=0.6))
    r, o, _ = render_both(instance, desc, 64, 64, spp=6, depth=5, seed=20 + mtype)
    assert_parity(r, o, "rough/metal maps mtype %d" % mtype)


@pytest.mark.parametrize("shape", [(1, 7), (5, 1), (3, 129), (37, 5), (130, 66)])
def test_odd_texture_sizes(instance, shape):
    """Textures live in 128-byte tiles (8 x 4 RGBA / 16 x 8 gray texels) on the device: sizes that end inside a tile, single rows
    and columns, and REPEAT wrapping across the ragged edge must sample exactly like the oracle's row-major images."""
    desc = cube_scene(material_type=abi.MAT_UBER)
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    desc.textures[1] = (abi.TEX_RGBA_SRGB, rng.integers(0, 256, shape + (4,), dtype=np.uint8), "odd rgba")
    desc.textures.append((abi.TEX_GRAY, rng.integers(30, 256, shape[::-1], dtype=np.uint8), "odd gray"))
    desc.materials[2].roughness = 2
    r, o, _ = render_both(instance, desc, 48, 48, spp=4, depth=3, seed=shape[0])
    assert_parity(r, o, "texture %dx%d" % shape)


def test_cube_all_light_types(instance):
    desc = cube_scene()
    desc.materials.append(make_material("emitter", diffuse_mul=(255, 200, 150)))
    desc.lights = [make_light(abi.LIGHT_OMNI, "omni", position=(0.3, 0.2, -0.4), intensity=0.7),
                   make_light(abi.LIGHT_SUN, "sun", direction=(0.2, -0.9, 0.3), intensity=0.5),
                   make_light(abi.LIGHT_AREA, "area", resource_id=2, intensity=0.8),
                   make_light(abi.LIGHT_SKY, "sky", resource_id=1, intensity=0.3, yaw=20, pitch=75, roll=10)]
    r, o, _ = render_both(instance, desc, 64, 64, spp=8, depth=4, seed=7)
    assert_parity(r, o, "cube lights")


def test_cube_orthographic_camera(instance):
    desc = cube_scene()
    desc.camera = make_camera(position=(0.1, 0.0, -0.5), target=(0, 0.2, 10), orthographic=True, scale=0.8, near=1e-3, far=50.0)
    r, o, _ = render_both(instance, desc, 64, 48, spp=4, depth=3)
    assert_parity(r, o, "cube ortho")


def _col_major(m):
    return np.asarray(m, np.float32).T.reshape(16)


def test_instance_transforms(instance):
    """Instances with real object-to-world transforms (translation, rotation with non-uniform scale, a mirror): world
    triangles, instance-space shading (normals through the inverse transpose, raytrace_hit.rchit:41-60) and the
    identity-transform shortcut must all agree with the oracle.  mattest.glaze and the cube only use the identity."""
    from glaze_amd.scene_desc import INSTANCE_DTYPE
    desc = cube_scene(material_type=abi.MAT_UBER)
    def T(x, y, z):
        m = np.eye(4); m[:3, 3] = (x, y, z); return m
    def S(x, y, z):
        return np.diag([x, y, z, 1.0])
    def R(axis, deg):
        a = np.radians(deg); c, s_ = np.cos(a), np.sin(a)
        m = np.eye(4)
        i, j = [(1, 2), (2, 0), (0, 1)][axis]
        m[i, i], m[i, j], m[j, i], m[j, j] = c, -s_, s_, c
        return m
    mats = [np.eye(4), T(0.3, -0.2, 0.6) @ R(2, 30) @ R(1, 20) @ S(0.15, 0.25, 0.1), T(-0.4, 0.1, 0.5) @ S(-0.2, 0.2, 0.2)]
    desc.transforms = np.stack([_col_major(m) for m in mats])
    desc.instances = np.array([(0, 0), (0, 1), (0, 2)], INSTANCE_DTYPE)
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.2, -0.7, 0.4), intensity=0.5))
    desc.lights.append(make_light(abi.LIGHT_AREA, "area", resource_id=2, intensity=0.6))   # one RTLight per instance, sampled through its transform
    r, o, _ = render_both(instance, desc, 96, 96, spp=6, depth=5, seed=11)
    assert_parity(r, o, "instance transforms")
    hdr = r.read_hdr()
    assert np.isfinite(hdr[..., :3]).all() and hdr[..., :3].std() > 0


def test_alpha_and_normal_maps(instance):
    """Non-opaque geometry (any-hit alpha test, raytrace_hit.rahit) and normal mapping (raytrace_hit.rchit:62-70)."""
    desc = cube_scene()
    rng = np.random.default_rng(0)
    nmap = np.concatenate([rng.integers(96, 160, (64, 64, 2), dtype=np.uint8), np.full((64, 64, 1), 255, np.uint8),
                           np.full((64, 64, 1), 255, np.uint8)], -1)
    y, x = np.mgrid[0:64, 0:64]
    alpha = np.where(((x // 8 + y // 8) % 2) == 0, 255, 0).astype(np.uint8)
    desc.textures.append((abi.TEX_RGBA_NORM, nmap, "normals"))
    desc.textures.append((abi.TEX_GRAY, alpha, "alpha"))
    desc.materials[2].normal = 2
    desc.materials[2].opacity = 3
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.1, -0.8, 0.5), intensity=1.0))
    r, o, _ = render_both(instance, desc, 64, 64, spp=6, depth=4, seed=3)
    assert_parity(r, o, "alpha+normal")
    assert (r.read_hdr()[..., :3].sum(-1) == 0).any()      # some primary rays leave through the holes


def test_opacity_map_switched_on_and_off_on_a_live_renderer(instance):
    """The traversal kernel is chosen by what the scene holds: without an opacity map k_trace carries no alpha code, with one its candidates
    on non-opaque geometry wait for the alpha phase (launch_trace, DeviceScene::has_non_opaque).  update_materials_and_lights switches a
    live renderer from the one to the other and back (the leaf records are rebuilt: the flag lives in them), in both launch modes, and
    every state must match the oracle -- alpha-tested candidates decided in a phase of their own change no pixel."""
    y, x = np.mgrid[0:64, 0:64]
    alpha = np.where(((x // 4 + y // 4) % 2) == 0, 255, 0).astype(np.uint8)
    desc = cube_scene()
    desc.textures.append((abi.TEX_GRAY, alpha, "alpha"))
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.1, -0.8, 0.5), intensity=1.0))
    for mode in ("two_kernels", "path"):
        cur = desc.copy()
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, cur), 200, 120)
        r.set_launch_mode(mode)
        r.set_depth(4)
        r.set_seed(5)
        holes = []
        for opacity in (0, 2, 0, 2):
            cur = cur.copy()
            cur.materials[2].opacity = opacity
            r.update_materials_and_lights(cur.materials, cur.lights)
            r.step(9)
            o = OracleRenderer(OracleScene(cur), 200, 120)
            o.set_depth(4)
            o.set_seed(5)
            o.step(9)
            assert_parity(r, o, "%s, opacity map %s" % (mode, "on" if opacity else "off"))
            holes.append(int((r.read_hdr()[..., :3].sum(-1) == 0).sum()))
        assert holes[0] == holes[2] and holes[1] == holes[3] and holes[1] > holes[0] + 500, holes   # primary rays leave through the holes exactly while the map is bound


def test_alpha_records_follow_the_opacity_map_a_material_names_and_its_texels(instance):
    """The alpha test reads one record per triangle slot (DeviceScene::alpha_recs: texture coordinates + the opacity map's descriptor,
    Scene::build_alpha_records).  The records have to follow every change that does NOT rebuild the hierarchy: a material pointed at
    ANOTHER opacity map (still non-opaque: the leaf flags stay), the texture array replaced by maps of another size and format (the
    descriptors move), and refresh_binded_textures on a running accumulation -- in both launch modes, against the oracle."""
    y, x = np.mgrid[0:64, 0:64]
    checker = np.where(((x // 4 + y // 4) % 2) == 0, 255, 0).astype(np.uint8)
    yy, xx = np.mgrid[0:40, 0:24]
    discs = np.where(((xx % 12 - 6) ** 2 + (yy % 10 - 5) ** 2) < 14, 0, 255).astype(np.uint8)
    bars = np.zeros((16, 48, 4), np.uint8)
    bars[..., 0] = np.where((np.arange(48) // 3) % 2 == 0, 255, 30)[None, :]          # an RGBA map: the test reads its red channel
    bars[..., 3] = 255
    desc = cube_scene()
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.1, -0.8, 0.5), intensity=1.0))
    desc.textures.append((abi.TEX_GRAY, checker, "checker"))      # 2
    desc.textures.append((abi.TEX_GRAY, discs, "discs"))          # 3
    for mode in ("two_kernels", "path"):
        cur = desc.copy()
        cur.materials[2].opacity = 2
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, cur), 160, 96)
        r.set_launch_mode(mode)
        r.set_depth(4)
        r.set_seed(9)

        def check(what, launches=7):
            r.step(launches)
            o = OracleRenderer(OracleScene(cur), 160, 96)
            o.set_depth(4)
            o.set_seed(9)
            o.step(launches)
            assert np.array_equal(bits(r.read_hdr()), bits(o.read_hdr())), "%s, %s" % (mode, what)
            return int((r.read_hdr()[..., :3].sum(-1) == 0).sum())

        holes = [check("the checker as opacity map")]
        cur = cur.copy()
        cur.materials[2].opacity = 3                                                     # another map, no flip: nothing rebuilds the hierarchy
        r.update_materials_and_lights(cur.materials, cur.lights)
        holes.append(check("the discs as opacity map"))
        cur = cur.copy()
        cur.textures = [cur.textures[0], cur.textures[1], (abi.TEX_RGBA_NORM, bars, "bars"), (abi.TEX_GRAY, checker[:32, :16].copy(), "small checker")]
        r.update_materials_and_lights(cur.materials, cur.lights, cur.textures)           # other sizes, another format: every descriptor moves
        holes.append(check("the small checker after the texture array was replaced"))
        cur = cur.copy()
        cur.materials[2].opacity = 2
        r.update_materials_and_lights(cur.materials, cur.lights)
        holes.append(check("the red channel of an RGBA map"))
        assert len(set(holes)) == len(holes), holes                                      # four different sets of holes
        # refresh_binded_textures keeps accumulating: compare a fresh frame on the refreshed textures
        cur = cur.copy()
        cur.textures = [cur.textures[0], cur.textures[1], (abi.TEX_GRAY, discs, "discs again"), cur.textures[3]]
        r.refresh_binded_textures(cur.textures)
        r.restart()
        check("after refresh_binded_textures")


def test_tables_too_large_for_lds_and_a_sky_taller_than_its_lds_copy(instance):
    """k_shade stages the material / light / descriptor tables (8 KB) and the sky's marginal cdf (1 087 rows) in LDS when they fit; a
    scene with 64 materials (13 KB of RTMaterial) and a 16 x 1500 sky reads both from memory instead -- same pixels as the oracle, in both
    launch modes (k_path sizes its table copy by itself)."""
    desc = cube_scene()
    base = desc.materials[2]
    rng = np.random.default_rng(4)
    for i in range(61):
        desc.materials.append(make_material("m%d" % i, mtype=abi.MAT_UBER if i % 3 == 0 else abi.MAT_LAMBERT, diffuse=1,
                                            diffuse_mul=tuple(int(v) for v in rng.integers(60, 255, 3)), roughness_mul=0.3 + 0.01 * i))
    assert len(desc.materials) == 64
    # six instances of the cube's mesh... the cube is one mesh of 12 triangles: give every face pair its own mesh / material
    idx = desc.indices.reshape(-1, 3)
    desc.meshes = np.array([(k, 3 + 10 * k, 6 * k, 6) for k in range(6)], MESH_DTYPE)     # materials 3, 13, 23, 33, 43, 53
    desc.instances = np.array([(k, 0) for k in range(6)], INSTANCE_DTYPE)
    sky = np.zeros((1500, 16, 4), np.uint8)
    sky[..., :3] = rng.integers(0, 256, (1500, 1, 3))
    sky[::37, :, :3] = 255
    sky[..., 3] = 255
    desc.textures.append((abi.TEX_RGBA_SRGB, sky, "tall sky"))
    desc.lights.append(make_light(abi.LIGHT_SKY, "sky", resource_id=2, intensity=0.4, yaw=10, pitch=80, roll=0))
    assert idx.shape[0] == 12
    for mode in ("two_kernels", "path"):
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 96, 80)
        r.set_launch_mode(mode)
        r.set_depth(5)
        r.set_seed(2)
        r.step(12)
        o = OracleRenderer(OracleScene(desc), 96, 80)
        o.set_depth(5)
        o.set_seed(2)
        o.step(12)
        assert assert_parity(r, o, "64 materials, tall sky, " + mode) == 1.0


def test_no_lights_renders_black_and_counts_nothing(instance):
    """lights_no == 0: the raygen shader returns before touching anything (path_trace.rgen:137-141)."""
    r, o, img = render_both(instance, cube_scene(light=False), 32, 32, spp=2)
    assert not r.read_hdr().any() and not o.read_hdr().any()
    assert not img[..., :3].any()


@pytest.fixture(scope="module")
def mattest_desc():
    return desc_from_oracle_parse(MATTEST)


def test_mattest_as_is(instance, mattest_desc):
    """BASELINE config 3 (as-is: Lambert / Metal / Glass + sky light) at reduced size, product reader on the GPU side."""
    gpu_scene = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))
    r, o, _ = render_both(instance, mattest_desc, 96, 96, spp=2, depth=6, gpu_scene=gpu_scene)
    assert_parity(r, o, "mattest")


@pytest.mark.parametrize("mtype", [abi.MAT_LAMBERT, abi.MAT_GLASS, abi.MAT_MIRROR, abi.MAT_METAL, abi.MAT_UBER, abi.MAT_FROSTED])
def test_mattest_outer_material_variants(instance, mattest_desc, mtype):
    """BASELINE config 3 variants: OuterMat (material 4) overridden, as glaze-app does interactively."""
    desc = mattest_desc.copy()
    desc.materials[4].mtype = mtype
    r, o, _ = render_both(instance, desc, 64, 64, spp=2, depth=8, seed=11)
    assert_parity(r, o, "mattest OuterMat=%d" % mtype)


def test_update_materials_and_lights_restarts(instance, mattest_desc):
    desc = mattest_desc.copy()
    scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 48, 48)
    r.set_depth(4)
    r.draw(1, want_image=False)
    desc.materials[4].mtype = abi.MAT_METAL
    r.update_materials_and_lights(desc.materials, desc.lights)
    r.step(8)                                   # request_new_frame: accumulation restarted
    assert r.read_hdr()[..., 3].max() == 8.0
    o = OracleRenderer(OracleScene(desc), 48, 48)
    o.set_depth(4)
    o.step(8)
    assert_parity(r, o, "after material update")


def _sky_cube(textures=None):
    desc = cube_scene()
    desc.lights = [make_light(abi.LIGHT_OMNI, "omni", position=(0.0, 0.5, 0.0), intensity=1.0),
                   make_light(abi.LIGHT_SKY, "sky", resource_id=1, intensity=0.3, yaw=20, pitch=75, roll=10)]
    if textures is not None:
        desc.textures = textures
    return desc


def _stripes(seed, size=64):
    rng = np.random.default_rng(seed)
    px = np.zeros((size, size, 4), np.uint8)
    px[..., :3] = rng.integers(0, 256, (size, 1, 3))          # horizontal bands: a different sky distribution per seed
    px[::7, :, :3] = 255
    px[..., 3] = 255
    return px


def test_update_textures_through_update_materials_and_lights(instance):
    """raytracer.rs:311-326 with Some textures: new texels for the materials AND a rebuilt sky distribution (the sky light samples texture 1)."""
    desc = _sky_cube()
    scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 64, 64)
    r.set_depth(3)
    r.draw(2, want_image=False)
    new_textures = [desc.textures[0], (abi.TEX_RGBA_SRGB, _stripes(1), "stripes")]
    r.update_materials_and_lights(desc.materials, desc.lights, new_textures)
    assert scene.info().n_textures == 2
    want = _sky_cube(new_textures)
    orc = OracleScene(want)
    assert np.array_equal(scene.debug_sky().view(np.uint32), orc.sky().view(np.uint32))          # marginal / conditional tables follow the new texels
    r.step(9)                                                                                     # request_new_frame: restarted
    assert r.read_hdr()[..., 3].max() == 9.0
    o = OracleRenderer(orc, 64, 64)
    o.set_depth(3)
    o.step(9)
    assert_parity(r, o, "after texture replacement")
    # a longer texture list, the cube's material switched to the new entry
    more = new_textures + [(abi.TEX_RGBA_SRGB, _stripes(2, 32), "more")]
    mats = [m for m in desc.materials]
    mats[2] = make_material("Material", mtype=abi.MAT_LAMBERT, diffuse=2, diffuse_mul=(204, 204, 204), ior=1.45)   # the cube's material
    r.update_materials_and_lights(mats, desc.lights, more)
    want = _sky_cube(more)
    want.materials = mats
    r.step(6)
    o = OracleRenderer(OracleScene(want), 64, 64)
    o.set_depth(3)
    o.step(6)
    assert_parity(r, o, "after texture list growth")
    # errors: a material or the sky pointing past the new list, an empty list
    with pytest.raises(abi.GlazeError):
        r.update_materials_and_lights(mats, desc.lights, more[:2])
    with pytest.raises(abi.GlazeError):
        r.update_materials_and_lights(desc.materials, desc.lights, more[:1])
    r.step(1)                                                                                     # failed updates left the scene usable


def test_refresh_binded_textures_keeps_accumulating(instance):
    """raytracer.rs:328-356: texture array swapped under the same materials; no request_new_frame."""
    desc = _sky_cube()
    scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 48, 48)
    r.set_depth(2)
    r.step(4)
    before = r.read_hdr()
    new_textures = [desc.textures[0], (abi.TEX_RGBA_SRGB, _stripes(5), "stripes")]
    r.refresh_binded_textures(new_textures)
    r.step(4)
    after = r.read_hdr()
    assert after[..., 3].min() == 8.0 and before[..., 3].max() == 4.0                            # same accumulation, continued
    assert not np.array_equal(after[..., :3], before[..., :3])
    r.restart()
    r.step(8)
    o = OracleRenderer(OracleScene(_sky_cube(new_textures)), 48, 48)
    o.set_depth(2)
    o.step(8)
    assert_parity(r, o, "fresh frame on the refreshed textures")


def test_reference_raytracer_unit_tests(instance, tmp_path):
    """lib/src/vulkan/raytracer.rs:1230-1290: load_raytrace (no scene), draw_outlive, save_to_disk, change_resolution."""
    r = glaze_amd.RayTraceRenderer.new(instance, None, 2, 2)                   # RayTraceRenderer::new(instance, None, 2, 2)
    img = r.draw(1)
    assert img.shape == (2, 2, 4)
    scene = glaze_amd.RayTraceScene.from_desc(instance, cube_scene())
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 2, 2)
    ticks = []
    img = r.draw(1, callback=lambda: ticks.append(1))                          # draw_outlive
    assert img.shape == (2, 2, 4) and len(ticks) == 1
    glaze_amd.save_image(tmp_path / "save.png", img)                           # save_to_disk
    assert (tmp_path / "save.png").exists() and np.array_equal(np.asarray(PILImage.open(tmp_path / "save.png")), img)
    r.change_resolution(4, 4)                                                  # change_resolution
    img = r.draw(1)
    assert img.shape == (4, 4, 4)
    o = OracleRenderer(OracleScene(cube_scene()), 4, 4)
    o.draw(1)
    assert_parity(r, o, "cube 4x4 after change_resolution")


def test_progressive_step_equals_draw(instance):
    desc = cube_scene()
    scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 64, 64)
    r.set_depth(3)
    ticks = []
    r.draw(5, callback=lambda: ticks.append(1), want_image=False)
    assert len(ticks) == 5                      # callback once per spp on the caller's thread (raytracer.rs:651-653)
    a = r.read_hdr()
    r.restart()
    for _ in range(5):
        r.step(3)
    assert np.array_equal(a.view(np.uint32), r.read_hdr().view(np.uint32))


def test_camera_rays_made_by_the_shading_code_in_every_launch_shape(instance):
    """A path that ends gets the NEXT launch's camera ray from the shading code (shade_pixel's reset site, FrameData::next_pixel_offset,
    ray_o.w = -0.0) instead of from the traversal kernel's refill.  The marker has to mean the same to everybody who reads the state: the
    two kernels and the per-wave launch loop, any number of launch chains, calls of any length (the last launch of a call peeks at the
    pixel offset of a launch that a later call makes), an exposure changed in between (no restart), a restart in the middle (camera rays
    that nobody has made yet), an image whose edge tiles reach past it (those pixels never get a ray) -- against the oracle's plain
    sequence of launches, bit for bit after every call."""
    desc = cube_scene(material_type=abi.MAT_UBER)
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.2, -0.7, 0.4), intensity=1.5))
    w, h = 150, 83                                # 3 x 2 tiles, the right and the bottom ones partly outside
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    o = OracleRenderer(OracleScene(desc), w, h)
    for x in (r, o):
        x.set_depth(4)                            # short paths: every pixel starts several new ones
        x.set_seed(11)
    for mode, chains in (("two_kernels", 1), ("path", 0), ("two_kernels", 2), ("two_kernels", 3)):
        r.set_launch_mode(mode)                   # (a change of the launch shape restarts the frame: the state arrays are laid out anew)
        r.set_chains(chains)
        o.restart()
        total = 0
        for n, exposure in ((1, 1.0), (3, 1.0), (2, 0.5), (7, 2.0), (1, 2.0), (4, 1.0)):
            r.set_exposure(exposure)
            o.set_exposure(exposure)
            r.step(n)
            o.step(n)
            total += n
            g, c = r.read_hdr(), o.read_hdr()
            assert g[..., 3].max() == float(total)
            assert np.array_equal(bits(g), bits(c)), "%s, %d chains, after %d launches: %d pixels differ" % (mode, chains, total, int((bits(g) != bits(c)).any(-1).sum()))
        assert np.array_equal(bits(r.read_result()), bits(o.read_result()))
        r.restart()
        o.restart()
        r.step(3)
        o.step(3)
        assert np.array_equal(bits(r.read_hdr()), bits(o.read_hdr())), "%s, %d chains: after a restart" % (mode, chains)


def test_exposure_applies_without_restart_and_resolution_change(instance):
    desc = cube_scene()
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 32, 32)
    r.set_depth(2)
    r.draw(2, want_image=False)
    before = r.read_hdr()
    r.set_exposure(2.0)                         # raytracer.rs:186-193: no restart
    r.step(2)
    after = r.read_hdr()
    assert after[..., 3].max() == 6.0 and (after[..., :3] >= before[..., :3]).all()
    res, cum = r.read_result(), r.read_hdr()
    lit = res[..., 3] > 0
    assert np.allclose(res[..., :3][lit], (cum[..., :3] * 2.0 / cum[..., 3:4])[lit], rtol=1e-6)
    r.change_resolution(4, 4)                   # raytracer.rs:1259-1284 test: image is 4x4 afterwards
    img = r.draw(1)
    assert img.shape == (4, 4, 4)


def test_tile_partition_sums_to_single_gpu(instance):
    """Multi-GPU sharding: every rank renders its 64x64 tiles; the sum over ranks is bit-identical."""
    desc = cube_scene()
    w, h = 200, 136                              # 4 x 3 tiles, ragged right/bottom edges
    full = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    full.set_depth(3)
    full.draw(2, want_image=False)
    ref = full.read_hdr()
    owner = np.zeros((h, w), np.uint16)
    abi.check(abi.lib().glz_host_tile_owner(w, h, 3, owner.ctypes.data))
    total = np.zeros_like(ref)
    for rank in range(3):
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
        r.set_partition(rank, 3)
        r.set_depth(3)
        r.draw(2, want_image=False)
        part = r.read_hdr()
        assert not part[owner != rank].any()     # zero outside the owned tiles
        assert part[owner == rank][:, 3].min() == 6.0
        total += part
    assert np.array_equal(total.view(np.uint32), ref.view(np.uint32))


def test_concurrent_chains_do_not_change_the_image(instance, mattest_desc):
    """glz_renderer_set_chains: the rank's tiles advance as n independent launch sequences on n streams; bit-identical for every n,
    also under a tile partition (chain s of S is the finer partition (rank + s * world, world * S))."""
    scene = glaze_amd.RayTraceScene.from_desc(instance, mattest_desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 200, 136)             # 4 x 3 tiles, ragged right and bottom edges
    r.set_launch_mode("two_kernels")                                          # chains belong to this mode (a frame this small would run as k_path)
    r.set_depth(5)
    r.set_chains(1)
    r.step(23)
    ref, ref_result = r.read_hdr(), r.read_result()
    for n in (2, 3, 5, 12, 16, 0):
        r.set_chains(n)
        r.step(10)
        r.set_exposure(2.0 * mattest_desc.meta.exposure)                       # queued shadow rays keep the exposure of their launch
        r.set_exposure(mattest_desc.meta.exposure)
        r.step(13)
        assert np.array_equal(r.read_hdr().view(np.uint32), ref.view(np.uint32)), "chains=%d" % n
        assert np.array_equal(r.read_result().view(np.uint32), ref_result.view(np.uint32)), "chains=%d" % n
    total = np.zeros_like(ref)
    for rank in range(3):
        r.set_partition(rank, 3)
        r.set_chains(2)
        r.step(23)
        total += r.read_hdr()
    assert np.array_equal(total.view(np.uint32), ref.view(np.uint32))
    s = r.stats()
    assert s.launches == 23 and s.trace_closest_ms > 0 and s.shade_ms > 0


@pytest.fixture(scope="module")
def atrium_file(tmp_path_factory):
    """The atrium the way bench.py and glaze-cli hand a scene to the product (cli/src/main.rs:78-91): Serializer -> .glaze V1 file ->
    parse -> RayTraceScene::new.  Returns the path and the scene as the ORACLE's own reader (oracle/glaze_v1.py: liblzma, PIL)
    gets it out of the same file, so the full-size tests compare two independent readings of what the bench measures."""
    from glaze_amd.scene_desc import save_scene
    path = str(tmp_path_factory.mktemp("atrium") / "atrium.glaze")
    save_scene(atrium_scene(), path)
    return path, desc_from_oracle_parse(path)


def file_scene(instance, atrium_file):
    return glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(atrium_file[0]))


def test_full_size_properties_atrium(instance, atrium_file):
    """BASELINE config 4 shape (1920x1080, synthetic atrium, through the file format): size-independent properties at full size."""
    desc = atrium_file[1]
    scene = file_scene(instance, atrium_file)
    info = scene.info()
    assert 200_000 <= info.n_world_triangles <= 330_000
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 1920, 1080)
    r.set_depth(8)
    r.step(16)
    a = r.read_hdr()
    assert (a[..., 3] == 16.0).all()                         # update_count on every pixel, every launch
    # a handful of pixels may legitimately go NaN (0/0 in the reference's GGX terms at exactly-normal half
    # vectors, faithfully kept); the oracle produces the same NaN set (checked at 1080p by tools/gpu_diag.py)
    finite = np.isfinite(a).all(-1)
    assert finite.mean() > 0.99999 and a[finite][:, :3].mean() > 0
    r.restart()
    r.step(16)
    assert np.array_equal(a.view(np.uint32), r.read_hdr().view(np.uint32))   # same seed stream -> bit-identical
    # the oracle on a crop-sized version of the same scene/camera agrees (same launch sequence)
    small = glaze_amd.RayTraceRenderer.new(instance, file_scene(instance, atrium_file), 96, 54)
    small.set_depth(8)
    small.step(8)
    o = OracleRenderer(OracleScene(desc), 96, 54)
    o.set_depth(8)
    o.step(8)
    assert_parity(small, o, "atrium 96x54")


def test_full_size_properties_4k_depth12(instance, atrium_file):
    """BASELINE config 5 shape (3840x2160, depth 12, tiles sharded over 8 GPUs) on one GPU: every rank's share of the 8-way
    partition rendered in turn (automatic chain count) sums to the whole frame bit for bit; counters and restart determinism."""
    scene = file_scene(instance, atrium_file)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 3840, 2160)
    r.set_depth(12)
    n = 13                                                   # one full path of depth 12 + the first segment of the next
    r.step(n)
    whole = r.read_hdr()
    assert (whole[..., 3] == float(n)).all()
    finite = np.isfinite(whole).all(-1)
    assert finite.mean() > 0.99999 and whole[finite][:, :3].mean() > 0
    total = np.zeros_like(whole)
    owners = glaze_amd.distributed.tile_owner(3840, 2160, 8)
    for rank in range(8):
        r.set_partition(rank, 8)
        r.step(n)
        part = r.read_hdr()
        assert (part[owners != rank] == 0).all()             # a rank only writes its own tiles
        total += part
    same = (total.view(np.uint32) == whole.view(np.uint32)) | (np.isnan(total) & np.isnan(whole))
    assert same.all()


def test_full_size_mattest_1024(instance, mattest_desc):
    """BASELINE config 3 shape (mattest.glaze, 1024x1024, depth 8): properties at full size + DIRECT integrator = depth-independent."""
    scene = glaze_amd.RayTraceScene.from_desc(instance, mattest_desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 1024, 1024)
    r.set_depth(8)
    r.step(24)
    a = r.read_hdr()
    assert (a[..., 3] == 24.0).all() and np.isfinite(a).all(-1).mean() > 0.9999 and a[..., :3][np.isfinite(a).all(-1)].mean() > 0
    img = r.read_rgba8()
    assert img.shape == (1024, 1024, 4) and (img[..., 3] == 255).mean() > 0.9 and img[..., :3].std() > 5   # alpha is 1 once update_result ran
    # DIRECT: one launch per sample whatever the depth setting
    r.set_integrator(glaze_amd.Integrator.DIRECT)
    r.set_depth(3)
    r.draw(2, want_image=False)
    d3 = r.read_hdr()
    r.set_depth(11)
    r.draw(2, want_image=False)
    assert (d3[..., 3] == 2.0).all() and np.array_equal(d3.view(np.uint32), r.read_hdr().view(np.uint32))


def test_traversal_work_counters_match_the_oracle(instance, mattest_desc):
    """node visits / triangle tests per sample are COUNTED, not estimated.  The instrumented kernels'
    counters must equal what the oracle counts when it walks the very same LBVH (downloaded from the device) with the
    documented visit rule (near child first, ties -> child0, prune with the current best t)."""
    for desc, w, h, depth, launches in ((cube_scene(), 64, 64, 2, 4), (mattest_desc, 48, 48, 6, 6)):
        scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
        nodes, tris = scene.debug_bvh()
        r = glaze_amd.RayTraceRenderer.new(instance, scene, w, h)
        r.set_depth(depth)
        r.enable_counters(True, True)
        r.step(launches)
        r.wait_idle()
        s = r.stats()
        osc = OracleScene(desc)
        info = scene.info()
        osc.set_ext_bvh(nodes, tris, list(info.bvh_grid_lo), list(info.bvh_grid_cell))
        o = OracleRenderer(osc, w, h)
        o.set_depth(depth)
        o.set_counting(True)
        o.step(launches)
        c = o.counters()
        assert s.closest_rays == c["closest_rays"] == w * h * launches
        assert s.shadow_rays == c["shadow_rays"] and s.hits == c["hits"]
        assert (s.closest_nodes, s.closest_tris) == (c["closest_nodes"], c["closest_tris"])
        assert (s.shadow_nodes, s.shadow_tris) == (c["shadow_nodes"], c["shadow_tris"])
        assert_parity(r, o, "counting build")


# ---------------------------------------------------------------------------------------------------------------------
# RGBA8 export (raytracer.rs:576-584 blit + memory.rs:269-483 export): bit-exact
# ---------------------------------------------------------------------------------------------------------------------
def test_rgba8_export_is_bit_exact(instance, mattest_desc):
    """The sRGB8 quantiser is stated as a threshold rule (no pow() per pixel), so the device, the host table behind the C ABI and
    the oracle's count over the rule agree on every byte: rendered frames, and a frame of adversarial floats pushed through
    glz_debug_tonemap (values at, just below and just above every threshold, denormals, negatives, NaN, inf)."""
    from oracle.pyoracle import lib as orc
    for desc, w, h, depth in ((cube_scene(material_type=abi.MAT_UBER), 160, 96, 4), (mattest_desc, 128, 128, 6)):
        r, o, img = render_both(instance, desc, w, h, spp=3, depth=depth, seed=5)
        assert np.array_equal(img, o.read_rgba8())
        assert np.array_equal(r.read_rgba8(), img)
    thr = np.zeros(256, np.float32)
    abi.check(abi.lib().glz_host_srgb8_thresholds(thr.ctypes.data))
    vals = [thr, np.nextafter(thr, np.float32(-1)), np.nextafter(thr, np.float32(2)),
            np.array([0.0, -0.0, -1.0, 1.0, 1.5, np.inf, -np.inf, np.nan, 1e-45, 1e-38, 0.0031308, 0.00313081, 0.5, 0.999999], np.float32),
            np.random.default_rng(3).random(4096, dtype=np.float32), np.random.default_rng(4).random(4096, dtype=np.float32) * 0.01]
    v = np.concatenate(vals).astype(np.float32)
    w, h = 64, (v.size + 63) // 64
    frame = np.zeros((h * w, 4), np.float32)
    frame[:v.size, 0] = v
    frame[:v.size, 1] = v[::-1]
    frame[:v.size, 2] = np.roll(v, 17)
    frame[:, 3] = 1.0
    got = np.zeros((h * w, 4), np.uint8)
    abi.check(abi.lib().glz_debug_tonemap(instance._h, frame.ctypes.data, h * w, got.ctypes.data))
    L = orc()
    want = np.array([[L.orc_to_srgb8(float(c)) for c in px[:3]] for px in frame], np.uint8)
    assert np.array_equal(got[:, :3], want)
    assert (got[:, 3] == 255).all()


# ---------------------------------------------------------------------------------------------------------------------
# update_camera (raytracer.rs:300-309) and change_scene (raytracer.rs:234-248) on a live renderer
# ---------------------------------------------------------------------------------------------------------------------
def test_update_camera_restarts_and_matches_the_oracle(instance):
    """update_camera replaces the push constants and requests a new frame: perspective -> orthographic -> another perspective
    on a live renderer, each compared with an oracle that was given the same camera."""
    desc = cube_scene(material_type=abi.MAT_UBER)
    desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.2, -0.7, 0.4), intensity=0.5))
    scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 80, 48)
    r.set_depth(4)
    r.set_seed(9)
    o = OracleRenderer(OracleScene(desc), 80, 48)
    o.set_depth(4)
    o.set_seed(9)
    r.step(5)
    o.step(5)
    assert_parity(r, o, "before update_camera")
    cams = [make_camera(position=(0.2, -0.1, -0.6), target=(0.0, 0.1, 5.0), orthographic=True, scale=0.7, near=1e-3, far=50.0),
            make_camera(position=(-0.3, 0.3, -0.2), target=(0.6, -0.2, 0.9), up=(0.1, 1.0, 0.0), fovx=1.2, near=1e-2, far=20.0),
            make_camera(position=(0.0, 0.0, 0.0), target=(0.0, 0.0, 100.0), fovx=np.pi / 2)]
    for i, cam in enumerate(cams):
        r.update_camera(cam)
        o.update_camera(cam)
        assert np.array_equal(r.push_constants().view(np.uint32), o.push_constants().view(np.uint32))
        r.step(6)                                                                       # request_new_frame: the count restarts at 0
        o.step(6)
        assert r.read_hdr()[..., 3].max() == 6.0
        assert_parity(r, o, "after update_camera %d" % i)
    # the scene's own camera is untouched (the renderer keeps its copy, raytracer.rs:300-309)
    assert tuple(scene.camera().position) == tuple(desc.camera.position)
    with pytest.raises(abi.GlazeError):
        bad = make_camera()
        bad.type = 7
        r.update_camera(bad)
    r.step(1)                                                                           # a rejected camera leaves the renderer usable


def test_change_scene_on_a_live_renderer(instance, mattest_desc):
    """change_scene (raytracer.rs:234-248): cube -> mattest -> cube on one renderer.  The new scene's camera and exposure are
    taken over (:246-247), accumulation restarts, the traversal spill area follows the new BVH; the old scene handle stays
    valid for the info / debug hooks after the renderer let go of it (shared ownership behind the C ABI)."""
    cube = cube_scene()
    s_cube = glaze_amd.RayTraceScene.from_desc(instance, cube)
    r = glaze_amd.RayTraceRenderer.new(instance, s_cube, 72, 72)
    r.set_depth(5)
    r.set_seed(2)
    r.step(7)
    s_mat = glaze_amd.RayTraceScene.from_desc(instance, mattest_desc)
    r.change_scene(s_mat)
    r.step(10)
    assert r.read_hdr()[..., 3].max() == 10.0
    o = OracleRenderer(OracleScene(mattest_desc), 72, 72)
    o.set_depth(5)
    o.set_seed(2)
    o.set_exposure(mattest_desc.meta.exposure)
    o.step(10)
    assert_parity(r, o, "after change_scene(mattest)")
    assert np.array_equal(r.push_constants().view(np.uint32), o.push_constants().view(np.uint32))   # the new scene's camera
    assert np.array_equal(r.read_rgba8(), o.read_rgba8())                                            # and its exposure
    # the scene the renderer dropped is still a valid handle
    info = s_cube.info()
    assert info.n_world_triangles == 12
    t, tri, _, _, _ = s_cube.debug_trace_closest([[0, 0, 0]], [[0, 0, 1]])
    assert np.isfinite(t[0]) and tri[0] < 12
    with pytest.raises(abi.GlazeError):
        r.change_scene(s_mat)                                                                         # already owned by a renderer
    s_cube2 = glaze_amd.RayTraceScene.from_desc(instance, cube)
    r.change_scene(s_cube2)
    r.step(4)
    o2 = OracleRenderer(OracleScene(cube), 72, 72)
    o2.set_depth(5)
    o2.set_seed(2)
    o2.step(4)
    assert_parity(r, o2, "after change_scene(cube)")
    del r
    assert s_mat.info().n_world_triangles == 138480                                                   # outlives the renderer too


# ---------------------------------------------------------------------------------------------------------------------
# Full-size frames against the oracle on a sample of tiles (BASELINE configs 3, 4, 5 at their real sizes)
# ---------------------------------------------------------------------------------------------------------------------
def _sample_tiles(w, h, n_random, seed):
    """Row-major ids of 64x64 tiles: the four corners (ragged right / bottom edges included), the centre, and n_random more."""
    tx, ty = (w + 63) // 64, (h + 63) // 64
    fixed = [0, tx - 1, (ty - 1) * tx, ty * tx - 1, (ty // 2) * tx + tx // 2, (ty // 3) * tx + (2 * tx) // 3]
    rng = np.random.default_rng(seed)
    rest = [int(t) for t in rng.permutation(tx * ty) if int(t) not in fixed][:n_random]
    return sorted(fixed + rest)


def _tile_mask(w, h, tiles):
    tx = (w + 63) // 64
    m = np.zeros((h, w), bool)
    for t in tiles:
        y0, x0 = (t // tx) * 64, (t % tx) * 64
        m[y0:y0 + 64, x0:x0 + 64] = True
    return m


def _assert_tiles_bit_equal(r, o, w, h, tiles, name):
    g, c = r.read_hdr(), o.read_hdr()
    m = _tile_mask(w, h, tiles)
    assert m.sum() >= 16 * 64 * 32
    gb, cb = g[m].view(np.uint32), c[m].view(np.uint32)
    same = (gb == cb) | (np.isnan(g[m]) & np.isnan(c[m]))
    assert same.all(), "%s: %d of %d sampled pixels differ from the oracle" % (name, int((~same.all(-1)).sum()), int(m.sum()))
    assert np.array_equal(np.isnan(g[m]), np.isnan(c[m])), name + ": NaN sets differ"
    gr, cr = r.read_result()[m], o.read_result()[m]
    assert ((gr.view(np.uint32) == cr.view(np.uint32)) | (np.isnan(gr) & np.isnan(cr))).all(), name + ": result image differs"
    assert np.array_equal(r.read_rgba8()[m], o.read_rgba8()[m]), name + ": RGBA8 export differs"


def test_full_size_mattest_1024_tiles_vs_oracle(instance, mattest_desc):
    """Config 3: mattest.glaze, 1024 x 1024, depth 8 -- the whole frame on the GPU, 18 of its 256 tiles on the oracle (pixels are
    independent: absolute-pixel RNG, path_trace.rgen:143-147), bit for bit, NaN sets and the 8-bit export included."""
    w = h = 1024
    launches = 17                                                                 # two full paths + the first segment of a third
    gpu_scene = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))   # product reader on the GPU side
    r = glaze_amd.RayTraceRenderer.new(instance, gpu_scene, w, h)
    r.set_depth(8)
    r.step(launches)
    tiles = _sample_tiles(w, h, 12, 1)
    o = OracleRenderer(OracleScene(mattest_desc), w, h)
    o.set_depth(8)
    o.set_exposure(mattest_desc.meta.exposure)
    o.set_tiles(tiles)
    o.step(launches)
    _assert_tiles_bit_equal(r, o, w, h, tiles, "mattest 1024^2")


@pytest.mark.parametrize("mtype", [abi.MAT_LAMBERT, abi.MAT_GLASS, abi.MAT_MIRROR, abi.MAT_METAL, abi.MAT_UBER])
def test_full_size_mattest_1024_outer_material_overrides(instance, mattest_desc, mtype):
    """Config 3 AT ITS SIZE for each of the five BSDF overrides BASELINE names (OuterMat = material 4, as glaze-app changes it
    interactively -- here through update_materials_and_lights on the live renderer of the parsed file): 1024 x 1024, depth 8, the whole
    frame on the GPU, eight sampled tiles per material on the oracle, bit for bit (accumulator, out32, NaN sets, RGBA8)."""
    w = h = 1024
    launches = 17
    desc = mattest_desc.copy()
    desc.materials[4].mtype = mtype
    gpu_scene = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))
    r = glaze_amd.RayTraceRenderer.new(instance, gpu_scene, w, h)
    r.set_depth(8)
    r.step(3)
    r.update_materials_and_lights(desc.materials, desc.lights)      # restarts the accumulation (raytracer.rs:268-326)
    r.step(launches)
    tiles = _sample_tiles(w, h, 4, 10 + mtype)
    assert len(tiles) >= 6
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(8)
    o.set_exposure(desc.meta.exposure)
    o.set_tiles(tiles)
    o.step(launches)
    m = _tile_mask(w, h, tiles)
    g, c = r.read_hdr(), o.read_hdr()
    assert ((g[m].view(np.uint32) == c[m].view(np.uint32)) | (np.isnan(g[m]) & np.isnan(c[m]))).all(), "mattest 1024^2 OuterMat=%d differs from the oracle" % mtype
    gr, cr = r.read_result()[m], o.read_result()[m]
    assert ((gr.view(np.uint32) == cr.view(np.uint32)) | (np.isnan(gr) & np.isnan(cr))).all()
    assert np.array_equal(r.read_rgba8()[m], o.read_rgba8()[m])


@pytest.mark.parametrize("size", [(512, 512), (512, 513), (513, 512)])
def test_full_size_cube_config2_on_both_sides_of_the_residency_boundary(instance, size):
    """Config 2 AT ITS SIZE: the cube, Lambert, omni light, 512 x 512, depth 2.  262 144 pixels are exactly the 4 096 groups of 64 that
    get a resident wave each, the boundary at which the automatic mode takes the per-wave launch loop (k_path); one more row or column of
    pixels (512 x 513: a ninth, 1-pixel-high row of tiles; 513 x 512: a ninth column) is on the other side of it and runs as two
    kernels per launch.  Both against the oracle on sampled tiles, 24 launches = 12 spp, and against each other where they overlap."""
    w, h = size
    launches = 24
    desc = cube_scene()
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    r.set_depth(2)
    assert r.launch_mode() == ("path" if (w, h) == (512, 512) else "two_kernels")
    r.step(launches)
    n_tiles = ((w + 63) // 64) * ((h + 63) // 64)
    tiles = sorted(set(_sample_tiles(w, h, 6, w + h)) | {n_tiles - 1, ((w + 63) // 64) - 1})        # the ragged corner tiles among them
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(2)
    o.set_tiles(tiles)
    o.step(launches)
    m = _tile_mask(w, h, tiles)
    g, c = r.read_hdr(), o.read_hdr()
    assert m.sum() >= 6 * 4096 // 2
    assert ((g[m].view(np.uint32) == c[m].view(np.uint32)) | (np.isnan(g[m]) & np.isnan(c[m]))).all(), "cube %dx%d differs from the oracle" % size
    assert (g[..., 3] == launches).all()
    assert np.array_equal(r.read_rgba8()[m], o.read_rgba8()[m])
    # the other launch mode renders the same frame bit for bit
    other = "two_kernels" if r.launch_mode() == "path" else "path"
    r2 = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    r2.set_launch_mode(other)
    r2.set_depth(2)
    r2.step(launches)
    assert np.array_equal(r2.read_hdr().view(np.uint32), g.view(np.uint32))


def test_full_size_atrium_1080p_tiles_vs_oracle(instance, atrium_file):
    """Config 4, what bench.py measures: the Sponza-class atrium as a .glaze file through the product's parse, 1920 x 1080 (30 x 17
    tiles, the bottom row 56 pixels high), depth 8; the oracle renders the sampled tiles from its own reading of the same file."""
    w, h = 1920, 1080
    launches = 17
    desc = atrium_file[1]
    r = glaze_amd.RayTraceRenderer.new(instance, file_scene(instance, atrium_file), w, h)
    r.set_depth(8)
    r.step(launches)
    tiles = sorted(set(_sample_tiles(w, h, 12, 2)) | {3 + 8 * k for k in (0, 17, 40)})         # some of rank 3's of an 8-way partition among them
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(8)
    o.set_tiles(tiles)
    o.step(launches)
    _assert_tiles_bit_equal(r, o, w, h, tiles, "atrium 1080p")
    assert r.launch_mode() == "two_kernels"
    # one GPU's share of the 8-way partition of this frame -- what bench.py --gpus 8 gives every device: 259 k pixels, rendered by the
    # per-wave launch loop (k_path), all 17 launches in one kernel
    r.set_partition(3, 8)
    assert r.launch_mode() == "two_kernels"      # a share that fills every wave slot with a scene of this size: two kernels since round 4
    r.set_launch_mode("path")
    assert r.launch_mode() == "path"
    r.step(launches)
    mine = [t for t in tiles if t % 8 == 3]
    part, c = r.read_hdr(), o.read_hdr()
    m = _tile_mask(w, h, mine)
    assert m.any() and ((part[m].view(np.uint32) == c[m].view(np.uint32)) | (np.isnan(part[m]) & np.isnan(c[m]))).all()
    assert (part[~_tile_mask(w, h, list(range(3, 30 * 17, 8)))] == 0).all()                   # nothing outside its own tiles


def test_full_size_sponza_like_atrium_1080p_tiles_vs_oracle(instance, tmp_path):
    """The atrium with the content classes real Sponza has (bench.py's extra.atrium_sponza_like): opacity-mapped lace cloth, foliage cards
    and vines -- candidates on them go through the any-hit alpha test INSIDE the traversal (raytrace_hit.rahit:24-39) -- normal maps on
    stone and brick (raytrace_hit.rchit:53-60), roughness maps on the Uber materials; 1920 x 1080, depth 8, through the file format, the
    sampled tiles (the planters' and the cloths' rows among them) bit for bit against the oracle's own reading of the same file."""
    from glaze_amd.scene_desc import save_scene
    w, h, launches = 1920, 1080, 17
    path = str(tmp_path / "atrium_sponza_like.glaze")
    save_scene(atrium_scene(sponza_like=True), path)
    desc = desc_from_oracle_parse(path)
    assert sum(1 for m in desc.materials if m.opacity) >= 4 and sum(1 for m in desc.materials if m.normal) >= 10 and sum(1 for m in desc.materials if m.roughness) >= 4
    scene = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(path))
    assert scene.info().n_world_triangles == 262487
    r = glaze_amd.RayTraceRenderer.new(instance, scene, w, h)
    r.set_depth(8)
    r.step(launches)
    tiles = sorted(set(_sample_tiles(w, h, 8, 5)) | {row * 30 + col for row in (7, 9, 11, 13) for col in (3, 12, 20, 27)})
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(8)
    o.set_tiles(tiles)
    o.step(launches)
    _assert_tiles_bit_equal(r, o, w, h, tiles, "Sponza-like atrium 1080p")
    # the alpha test's texture fetches happened, inside k_trace (the counting build books them apart from k_shade's)
    r.enable_counters(True, False)
    r.restart()
    r.step(9)
    st = r.stats()
    assert st.alpha_tex_bytes > 0 and st.tex_bytes > st.alpha_tex_bytes
    # and the other launch mode / the 8-wide walk see the same image on a tile share
    r.enable_counters(False, False)
    r.set_partition(3, 8)
    r.restart()
    r.step(launches)
    a = r.read_hdr()
    r.set_node_width(8)
    r.step(launches)
    b = r.read_hdr()
    r.set_node_width(0)
    r.set_launch_mode("path")
    r.step(launches)
    c = r.read_hdr()
    bits = lambda x: np.nan_to_num(x, nan=-1.0).view(np.uint32)
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(c))
    mine = [t for t in tiles if t % 8 == 3]
    m = _tile_mask(w, h, mine)
    ref = o.read_hdr()
    assert m.any() and ((a[m].view(np.uint32) == ref[m].view(np.uint32)) | (np.isnan(a[m]) & np.isnan(ref[m]))).all()


def test_full_size_atrium_4k_depth12_tiles_vs_oracle(instance, atrium_file):
    """Config 5: the same file, 3840 x 2160 (60 x 34 tiles, ragged bottom row), depth 12, rendered as rank 3's share of the 8-way
    partition plus the whole frame: the sampled tiles that rank 3 owns must match the oracle in both, the others in the whole frame."""
    w, h = 3840, 2160
    launches = 14
    desc = atrium_file[1]
    r = glaze_amd.RayTraceRenderer.new(instance, file_scene(instance, atrium_file), w, h)
    r.set_depth(12)
    r.step(launches)
    tiles = _sample_tiles(w, h, 10, 3)
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(12)
    o.set_tiles(tiles)
    o.step(launches)
    _assert_tiles_bit_equal(r, o, w, h, tiles, "atrium 4K")
    r.set_partition(3, 8)
    r.step(launches)
    mine = [t for t in tiles if t % 8 == 3]
    part, c = r.read_hdr(), o.read_hdr()
    m = _tile_mask(w, h, mine)
    if m.any():
        assert ((part[m].view(np.uint32) == c[m].view(np.uint32)) | (np.isnan(part[m]) & np.isnan(c[m]))).all()


# ---------------------------------------------------------------------------------------------------------------------
# launch modes: two kernels per launch (throughput) and the per-wave launch loop k_path (a small tile share per GPU)
# ---------------------------------------------------------------------------------------------------------------------
def _mode_pair(instance, desc, w, h, depth, seed=7, **kw):
    out = []
    for mode in ("two_kernels", "path"):
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
        r.set_launch_mode(mode)
        assert r.launch_mode() == mode
        r.set_depth(depth)
        r.set_seed(seed)
        out.append(r)
    return out


@pytest.mark.parametrize("mtype", [abi.MAT_LAMBERT, abi.MAT_GLASS, abi.MAT_METAL, abi.MAT_UBER])
def test_path_mode_equals_two_kernels_and_the_oracle(instance, mtype):
    """k_path runs every wave's 64 pixels through the launches of a batch with no grid-wide boundary; per pixel the operations and
    their order are those of k_trace / k_shade: cumulative image, result image and RGBA8 bit for bit, for every BSDF family."""
    desc = cube_scene(material_type=mtype)
    w, h, depth = 136, 72, 5                                                                   # ragged tiles
    two, path = _mode_pair(instance, desc, w, h, depth)
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(depth)
    o.set_seed(7)
    for k in (1, 4, 17, 70, 200):                                                              # batches of 1, 4, 17, 70 and 192 + 8 launches
        two.step(k); path.step(k); o.step(k)
        a, b = two.read_hdr(), path.read_hdr()
        assert np.array_equal(bits(a), bits(b)), k
        assert np.array_equal(bits(two.read_result()), bits(path.read_result())), k
        assert np.array_equal(bits(b), bits(o.read_hdr())), k
    assert np.array_equal(two.read_rgba8(), path.read_rgba8())
    assert path.stats().launches == two.stats().launches == 292
    assert path.stats().other_ms > 0 and path.stats().trace_closest_ms == 0                    # it really was k_path
    assert two.stats().other_ms == 0 and two.stats().trace_closest_ms > 0


def test_path_mode_mattest_partition_exposure_and_counters(instance):
    """mattest (sky light, glass, metal) on a tile partition; exposure changed between batches; the work counters switched on in
    the middle (launches then run as two kernels) and off again; draw() with its per-sample callback."""
    desc = desc_from_oracle_parse(MATTEST)
    w, h, depth = 200, 136, 6
    two, path = _mode_pair(instance, desc, w, h, depth, seed=3)
    for r in (two, path):
        r.set_partition(1, 3)
    assert path.launch_mode() == "path"
    two.step(7); path.step(7)
    for r in (two, path):
        r.set_exposure(0.4)
    two.step(9); path.step(9)
    path.enable_counters(True, True)
    two.step(5); path.step(5)
    path.wait_idle()
    assert path.stats().closest_rays > 0                                                       # counted: these five ran as k_trace / k_shade
    path.enable_counters(False, True)
    two.step(20); path.step(20)
    assert np.array_equal(bits(two.read_hdr()), bits(path.read_hdr()))
    assert np.array_equal(bits(two.read_result()), bits(path.read_result()))
    ticks = []
    img = path.draw(3, callback=lambda: ticks.append(1))
    assert len(ticks) == 3 and np.array_equal(img, two.draw(3))
    # DIRECT integrator: one launch per sample
    for r in (two, path):
        r.set_integrator(glaze_amd.Integrator.DIRECT)
        r.step(6)
    assert np.array_equal(bits(two.read_hdr()), bits(path.read_hdr()))


def test_auto_launch_mode_goes_by_the_pixels_a_device_owns(instance):
    desc = cube_scene()
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 1920, 1080)
    assert r.launch_mode() == "two_kernels"
    r.set_partition(0, 8)
    assert r.launch_mode() == "path"
    r.set_partition(0, 2)
    assert r.launch_mode() == "two_kernels"
    r.set_launch_mode("path")
    assert r.launch_mode() == "path"
    r.set_launch_mode("auto")
    r.set_partition(0, 1)
    r.change_resolution(320, 200)
    assert r.launch_mode() == "path"
    # a scene of some size: the launch loop only where the chip is less than four fifths full
    big = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, atrium_scene()), 1920, 1080)
    for world, mode in ((1, "two_kernels"), (8, "two_kernels"), (10, "path"), (16, "path")):
        big.set_partition(0, world)
        assert big.launch_mode() == mode, (world, big.launch_mode())


def test_shading_side_work_counters(instance, mattest_desc):
    """The counted quantities bench.py books for k_shade besides state and hit records: texture fetches that read memory, their texel
    bytes, light samples.  Cube: every hit is Lambertian with the 512 x 512 checker as its diffuse texture (the 1 x 1 textures live in
    their descriptors) and samples the one omni light -- one RGBA fetch and one light sample per hit, no sky."""
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, cube_scene()), 200, 136)
    r.set_depth(3)
    r.enable_counters(True, True)
    r.step(7)
    r.wait_idle()
    s = r.stats()
    assert s.hits > 0 and s.tex_fetches == s.hits and s.tex_bytes == 16 * s.hits and s.alpha_tex_bytes == 0
    assert s.light_samples == s.hits and s.sky_samples == 0
    # mattest: a sky light (sampled by every non-specular hit; its texel fetch is counted) and sky texels on misses
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, mattest_desc), 200, 136)
    r.set_depth(4)
    r.enable_counters(True, True)
    r.step(9)
    r.wait_idle()
    s = r.stats()
    assert 0 < s.sky_samples <= s.hits and s.light_samples == 0 and s.tex_fetches >= s.sky_samples and s.tex_bytes <= 16 * s.tex_fetches
    r.enable_counters(False, True)
    r.restart()
    r.step(3)
    r.wait_idle()
    assert r.stats().tex_fetches == 0                                           # nothing is counted (or paid for) outside counting passes
