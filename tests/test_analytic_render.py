"""A whole render against a closed form.

HIP == oracle bit for bit says nothing about a misreading both share, and the oracle-against-mathematics tests look at one
routine at a time.  This one closes the loop at the other end: a scene whose converged image can be written down -- the unit cube
seen from its centre, grey Lambertian walls, one point light AT the camera, direct lighting -- so that camera rays, intersection,
shading frame, the light's inverse-square law, the cosine, the division by pi, the jitter average and the accumulation are all
compared with numpy in float64:

    pixel(u, v) = rgb(albedo / pi * light colour) * intensity / r^3,     r = |(u, v, 1)| / max(|u|, |v|, 1)

(the point on the cube along direction (u, v, 1) lies at distance r, the light is r away from it, and the wall it is on is seen
and lit under the cosine 1 / r).  The scene is mirror-symmetric in x and y, so the formula does not depend on how the image axes
are oriented.  Only the spectral step -- an RGB albedo and an RGB light colour to a radiance in RGB -- is taken from the oracle's
own routines, which have their own tests (test_oracle_math.py: grey stays grey, luminance is the Y row).
"""
import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, make_camera, make_light
from glaze_amd.scenes import cube_scene
from oracle import pyoracle
from oracle.pyoracle import OracleRenderer, OracleScene

CUBE_MAT = 2
ALBEDO = (150, 150, 150)
INTENSITY = 0.8


def room():
    desc = cube_scene(material_type=abi.MAT_LAMBERT)
    m = desc.materials[CUBE_MAT]
    m.diffuse = 0                      # the 1 x 1 white texture: the walls' colour is diffuse_mul
    m.diffuse_mul[:3] = ALBEDO
    desc.lights = [make_light(abi.LIGHT_OMNI, "at the camera", position=(0.0, 0.0, 0.0), intensity=INTENSITY)]
    return desc


def colour_factor(desc, light_pos=(0.0, 0.0, 0.0)):
    """rgb of (Lambert value of the walls) x (emission of the light at unit distance): the oracle's spectral routines"""
    o = OracleScene(desc)
    up = np.array([[0.0, 0.0, 1.0]], np.float32)
    value, pdf = o.bsdf_value(CUBE_MAT, up, up)                         # albedo spectrum / pi
    assert pdf[0] > 0
    at = np.asarray(light_pos, np.float32) + np.array([1.0, 0.0, 0.0], np.float32)
    _, dist, lpdf, em = o.light_sample(0, at[None, :], np.zeros((1, 3), np.float32))
    assert dist[0] == 1.0 and lpdf[0] == 1.0                            # emission at distance 1 = colour x intensity
    sp = np.ascontiguousarray(value[0] * em[0], np.float32)
    rgb = np.zeros(3, np.float32)
    pyoracle.lib().orc_dev_rgb(sp.ctypes.data, rgb.ctypes.data)
    # the Lambert value is albedo / pi: a grey of 150 / 255 (sRGB-decoded by the material upload) under a white light
    assert rgb.min() > 0
    return rgb.astype(np.float64)


def closed_form(n, factor, sub=8):
    """mean over each pixel (sub x sub positions) of factor / r^3 for a square image of n pixels, horizontal field of view 90 degrees"""
    c = (np.arange(n * sub) + 0.5) / (n * sub) * 2.0 - 1.0              # tan(45 deg) = 1: image plane coordinates at distance 1
    u, v = np.meshgrid(c, c, indexing="xy")
    r = np.sqrt(u * u + v * v + 1.0) / np.maximum(np.maximum(np.abs(u), np.abs(v)), 1.0)
    e = (1.0 / r ** 3).reshape(n, sub, n, sub).mean(axis=(1, 3))
    return e[..., None] * factor[None, None, :]


def check(img, n, factor):
    assert (img[..., 3] > 0).all()
    got = img[..., :3].astype(np.float64) / img[..., 3:4]
    want = closed_form(n, factor)
    assert np.isfinite(got).all()
    rel = np.abs(got - want) / want
    # the jitter sequence visits a handful of positions per pixel, the closed form averages 64: both sample a smooth function
    assert rel.max() < 0.01 and rel.mean() < 0.004, (rel.max(), rel.mean())
    # the brightest pixels look straight at a wall (r = 1), the darkest into a corner (r = sqrt 3): a ratio of 3^1.5
    assert abs(got[..., 1].max() / got[..., 1].min() / 3.0 ** 1.5 - 1.0) < 0.1


def test_oracle_render_matches_the_closed_form():
    desc = room()
    n = 48
    o = OracleRenderer(OracleScene(desc), n, n)
    o.set_integrator(abi.DIRECT)
    o.set_seed(5)
    o.step(24)
    check(o.read_hdr(), n, colour_factor(desc))


@pytest.mark.gpu
def test_hip_render_matches_the_closed_form(instance):
    desc = room()
    n = 256
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), n, n)
    r.set_integrator(glaze_amd.Integrator.DIRECT)
    r.set_seed(5)
    r.step(32)
    check(r.read_hdr(), n, colour_factor(desc))


# ---------------------------------------------------------------------------------------------------------------------
# a hard shadow: a small square between an off-axis light and the far wall
# ---------------------------------------------------------------------------------------------------------------------
LIGHT2 = np.array([0.5, 0.0, -0.5])
OCC_Z, OCC_H = 0.5, 0.1          # the occluder: |x|, |y| <= OCC_H in the plane z = OCC_Z, facing the camera


def room_with_occluder():
    desc = room()
    desc.lights = [make_light(abi.LIGHT_OMNI, "off axis", position=tuple(LIGHT2), intensity=INTENSITY)]
    quad = np.zeros(4, VERTEX_DTYPE)
    for i, (x, y) in enumerate(((-OCC_H, -OCC_H), (OCC_H, -OCC_H), (OCC_H, OCC_H), (-OCC_H, OCC_H))):
        quad[i] = ((x, y, OCC_Z), (0.0, 0.0, -1.0), (0.5 + x, 0.5 + y))
    base = len(desc.vertices)
    desc.vertices = np.concatenate([desc.vertices, quad])
    first = len(desc.indices)
    desc.indices = np.concatenate([desc.indices, np.array([0, 1, 2, 0, 2, 3], np.uint32) + base])
    desc.meshes = np.concatenate([desc.meshes, np.array([(1, CUBE_MAT, first, 6)], MESH_DTYPE)])
    desc.instances = np.concatenate([desc.instances, np.array([(1, 0)], INSTANCE_DTYPE)])
    return desc


def closed_form_with_occluder(n, factor, flip_x, ortho=False):
    """per pixel centre: expected value, and a mask of the pixels that are safely inside one region (away from the edges of the
    occluder and of its shadow, where a pixel mixes two values).  ortho: parallel rays from (u, v, 0) instead of rays through
    (u, v, 1) from the origin."""
    c = (np.arange(n) + 0.5) / n * 2.0 - 1.0
    u, v = np.meshgrid(-c if flip_x else c, c, indexing="xy")
    zs = 1.0 if ortho else OCC_Z                                         # what a pixel coordinate is multiplied by on the plane z = OCC_Z
    on_occ = (np.abs(u) * zs < OCC_H) & (np.abs(v) * zs < OCC_H)
    z = np.where(on_occ, OCC_Z, 1.0)
    p = np.stack([u, v, z], -1) if ortho else np.stack([u * z, v * z, z], -1)   # the surface point: on the occluder or on the wall z = 1
    wi = LIGHT2 - p
    d2 = (wi ** 2).sum(-1)
    cos = np.abs(wi[..., 2]) / np.sqrt(d2)                               # both surfaces have the normal (0, 0, -1)
    e = cos / d2
    # a wall point is shadowed when the segment to the light passes through the square
    s = (OCC_Z - LIGHT2[2]) / (p[..., 2] - LIGHT2[2])                    # parameter of the plane z = OCC_Z on the segment light -> p
    q = LIGHT2 + s[..., None] * (p - LIGHT2)
    shadow = ~on_occ & (np.abs(q[..., 0]) < OCC_H) & (np.abs(q[..., 1]) < OCC_H)
    e = np.where(shadow, 0.0, e)
    m = 3.0 / n                                                          # margin: one and a half pixels
    near_occ_edge = (np.abs(np.abs(u) * zs - OCC_H) < m) & (np.abs(v) * zs < OCC_H + m) | (np.abs(np.abs(v) * zs - OCC_H) < m) & (np.abs(u) * zs < OCC_H + m)
    near_shadow_edge = (np.abs(np.abs(q[..., 0]) - OCC_H) < m) & (np.abs(q[..., 1]) < OCC_H + m) | (np.abs(np.abs(q[..., 1]) - OCC_H) < m) & (np.abs(q[..., 0]) < OCC_H + m)
    # The square is two triangles and parallel rays on a pixel grid aligned with it meet their shared edge (the diagonal x = y)
    # exactly: the intersector is watertight (oracle.cpp ray_tri), so the diagonal is NOT left out -- neither for the camera rays
    # that see it nor for the shadow rays that cross it.  (Rounds 1-2 ran Moeller-Trumbore and had to mask it.)
    safe = ~near_occ_edge & ~(near_shadow_edge & ~on_occ)
    return e[..., None] * factor[None, None, :], safe, shadow, on_occ


def check_shadow(img, n, factor, ortho=False):
    got = img[..., :3].astype(np.float64) / img[..., 3:4]
    best = None
    for flip in (False, True):     # the closed form is stated without knowing which way the image's x axis runs; exactly one way fits
        want, safe, shadow, on_occ = closed_form_with_occluder(n, factor, flip, ortho)
        lit = safe & ~shadow
        rel = np.abs(got[lit] - want[lit]) / want[lit]
        dark = got[safe & shadow]
        ok = rel.max() < 0.02 and (dark == 0.0).all()
        if ok:
            assert best is None, "both orientations fit: the scene is not asymmetric enough"
            best = (flip, rel.max(), int((safe & shadow).sum()), int((safe & on_occ).sum()))
    assert best is not None, "neither orientation of the x axis reproduces the closed form"
    assert best[2] >= max(6, 0.0005 * n * n) and best[3] >= max(6, 0.0005 * n * n)      # the shadow and the occluder are really in view
    return best[0]


def test_oracle_shadow_matches_the_closed_form():
    desc = room_with_occluder()
    n = 96
    o = OracleRenderer(OracleScene(desc), n, n)
    o.set_integrator(abi.DIRECT)
    o.set_seed(9)
    o.step(16)
    check_shadow(o.read_hdr(), n, colour_factor(desc, LIGHT2))


@pytest.mark.gpu
def test_hip_shadow_matches_the_closed_form(instance):
    desc = room_with_occluder()
    n = 256
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), n, n)
    r.set_integrator(glaze_amd.Integrator.DIRECT)
    r.set_seed(9)
    r.step(24)
    check_shadow(r.read_hdr(), n, colour_factor(desc, LIGHT2))


# ---------------------------------------------------------------------------------------------------------------------
# the orthographic camera
# ---------------------------------------------------------------------------------------------------------------------
def ortho_room(scale):
    desc = room_with_occluder()
    desc.camera = make_camera(position=(0, 0, 0), target=(0, 0, 100), up=(0, 1, 0), orthographic=True, scale=scale, near=1e-3, far=100.0)
    return desc


def test_oracle_orthographic_render_matches_the_closed_form_whatever_the_scale():
    """path_trace.rgen:47-56: the orthographic ray starts at camera2world * (ndc, 0, 1) and runs along the view direction, so the image
    always covers [-1, 1]^2 of the camera plane -- `scale` (geometry/camera.rs: the orthographic projection's half extent) only enters
    screen2camera, which an orthographic ray never multiplies a pixel coordinate with (Q18).  Kept: the images of two scales are
    bit-identical, and both are the closed form for parallel rays from (u, v, 0)."""
    n = 96
    imgs = []
    for scale in (0.35, 3.0):
        desc = ortho_room(scale)
        o = OracleRenderer(OracleScene(desc), n, n)
        o.set_integrator(abi.DIRECT)
        o.set_seed(9)
        o.step(16)
        imgs.append(o.read_hdr())
        flip_ortho = check_shadow(imgs[-1], n, colour_factor(desc, LIGHT2), ortho=True)
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    # both cameras put world + x on the same side of the image
    o = OracleRenderer(OracleScene(room_with_occluder()), n, n)
    o.set_integrator(abi.DIRECT)
    o.set_seed(9)
    o.step(16)
    assert check_shadow(o.read_hdr(), n, colour_factor(room_with_occluder(), LIGHT2)) == flip_ortho


@pytest.mark.gpu
def test_hip_orthographic_render_matches_the_closed_form(instance):
    desc = ortho_room(0.35)
    n = 256
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), n, n)
    r.set_integrator(glaze_amd.Integrator.DIRECT)
    r.set_seed(9)
    r.step(24)
    check_shadow(r.read_hdr(), n, colour_factor(desc, LIGHT2), ortho=True)
