"""Anisotropic texture footprint of the ray-cone level of detail (lod mode 2): what the oracle's rule does, on the
CPU.  The reference's sampler enables anisotropy at the device's maximum (scene.rs:716-749) but only its raster viewer samples with
derivatives; mode 2 is the build's opt-in next to GLZ_LOD_RAY_CONES, so there is no reference image to compare with: the checks
are against a converged level-0 render (the ground truth any texture filter approximates) and against closed forms."""
import numpy as np

from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_camera, make_light, make_material, make_meta
from oracle.pyoracle import OracleRenderer, OracleScene


def floor_scene(texture, tiles=6.0, height=0.12, depth=24.0, half_width=8.0):
    """A floor seen at a grazing angle: y = -height, z in [0.2, depth], the texture repeated `tiles` times per unit length."""
    p = [(-half_width, -height, 0.2), (half_width, -height, 0.2), (half_width, -height, depth), (-half_width, -height, depth)]
    vertices = np.zeros(4, VERTEX_DTYPE)
    for i, q in enumerate(p):
        vertices[i] = (q, (0.0, 1.0, 0.0), (q[0] * tiles, q[2] * tiles))
    indices = np.array([0, 2, 1, 0, 3, 2], np.uint32)
    materials = [make_material("default"), make_material("floor", mtype=abi.MAT_LAMBERT, diffuse=1)]
    textures = [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default"), (abi.TEX_RGBA_SRGB, texture, "floor")]
    lights = [make_light(abi.LIGHT_SUN, "sun", direction=(0.0, -1.0, 0.0), intensity=1.0)]
    camera = make_camera(position=(0, 0, 0), target=(0, -0.02, 1), up=(0, 1, 0), fovx=np.float32(np.radians(60.0)), near=1e-3, far=100.0)
    return SceneDesc(vertices, indices, np.array([(0, 1, 0, 6)], MESH_DTYPE), None, np.array([(0, 0)], INSTANCE_DTYPE), materials, lights,
                     textures, camera, make_meta(centre=(0, 0, depth / 2), radius=depth, exposure=1.0))


def stripes(n=64, along_u=True):
    """Black / white lines one texel wide, varying along u (columns) or along v (rows)."""
    line = np.where(np.arange(n) % 2 == 0, 255, 0).astype(np.uint8)
    c = np.tile(line[None, :], (n, 1)) if along_u else np.tile(line[:, None], (1, n))
    return np.stack([c, c, c, np.full_like(c, 255)], -1)


def render(desc, w, h, lod, launches):
    o = OracleRenderer(OracleScene(desc), w, h)
    o.set_depth(1)
    o.set_integrator(0)          # DIRECT: one textured bounce lit by the sun
    o.set_seed(3)
    o.set_texture_lod(lod)
    o.step(launches)
    return o.read_hdr()[..., :3] / float(launches)


def test_probes_keep_detail_across_the_footprints_short_axis():
    """Stripes that vary ACROSS the view direction (along x = u): the footprint's long axis lies along z = v, so the probes of mode 2
    all see the same stripe and the pattern survives where the footprint is narrower than a stripe; the isotropic cone takes the long
    axis as the footprint's diameter and blurs the same pixels to grey.  Ground truth: level 0, many jittered launches."""
    desc = floor_scene(stripes(64, along_u=True), tiles=1.0 / 64.0 * 24.0)      # 24 texels per unit length: stripes 1/24 wide
    w, h = 160, 60
    truth = render(desc, w, h, 0, 192)
    iso = render(desc, w, h, 1, 8)
    aniso = render(desc, w, h, 2, 8)
    floor = truth.sum(-1) > 0
    rows = np.where(floor.all(1))[0]
    assert rows.size > 20
    # the band where a pixel is 0.15 .. 0.6 stripes wide: the pattern is resolvable, and the floor is seen at 1 / cos = 3 .. 12
    dist = 0.12 / ((rows - h / 2 + 0.5) / (w / 2) * np.tan(np.radians(30.0)) + 0.02)          # z of the row (pinhole, tilt 0.02)
    pixel = dist * 2 * np.tan(np.radians(30.0)) / w * 24.0                                      # texels per pixel across
    band = rows[(pixel > 0.15) & (pixel < 0.6)]
    assert band.size >= 5
    contrast = lambda im: float(np.mean([im[r, :, 0].std() for r in band]))
    err = lambda im: float(np.abs(im[band] - truth[band]).mean())
    assert contrast(aniso) > 0.7 * contrast(truth)
    assert contrast(iso) < 0.5 * contrast(truth)
    assert err(aniso) < 0.5 * err(iso)


def test_probes_average_along_the_footprints_long_axis():
    """Stripes that vary ALONG the view direction: far away many stripes fall into one pixel and every filter must return their mean;
    mode 2 does (as the isotropic cone does), level 0 with few samples does not."""
    desc = floor_scene(stripes(64, along_u=False), tiles=1.0 / 64.0 * 24.0)
    w, h = 160, 60
    truth = render(desc, w, h, 0, 192)
    one = render(desc, w, h, 0, 4)
    aniso = render(desc, w, h, 2, 4)
    far = slice(h // 2 + 3, h // 2 + 9)                                                          # rows just below the horizon
    rough = lambda im: float(np.abs(np.diff(im[far, :, 0], axis=0)).mean())                     # row to row: the floor is evenly lit
    assert rough(aniso) < 0.25 * rough(one)
    assert abs(aniso[far].mean() - truth[far].mean()) < 0.05 * truth[far].mean()


def test_head_on_surfaces_take_one_probe():
    """1 / |cos| <= 1 + rounding: one probe, the isotropic result bit for bit (wall facing an orthographic camera)."""
    n = 64
    tex = np.random.default_rng(5).integers(0, 256, (n, n, 4), dtype=np.uint8)
    vertices = np.zeros(4, VERTEX_DTYPE)
    for i, (x, y) in enumerate([(-1, -1), (1, -1), (1, 1), (-1, 1)]):
        vertices[i] = ((x, y, 2.0), (0.0, 0.0, -1.0), (x * 3.0, y * 3.0))
    desc = SceneDesc(vertices, np.array([0, 1, 2, 0, 2, 3], np.uint32), np.array([(0, 1, 0, 6)], MESH_DTYPE), None, np.array([(0, 0)], INSTANCE_DTYPE),
                     [make_material("default"), make_material("wall", diffuse=1)], [make_light(abi.LIGHT_SUN, "sun", direction=(0.0, 0.0, 1.0))],
                     [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default"), (abi.TEX_RGBA_SRGB, tex, "noise")],
                     make_camera(position=(0, 0, 0), target=(0, 0, 1), orthographic=True, scale=1.0, near=1e-3, far=50.0), make_meta(centre=(0, 0, 2), radius=2.0))
    a, b = render(desc, 48, 48, 1, 3), render(desc, 48, 48, 2, 3)
    assert a.max() > 0 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
