"""More whole renders against closed forms (see test_analytic_render.py for why): the light callables and specular paths that the
point-light room does not touch.  Every scene is OPEN -- a wall that fills the view, a few small quads -- so that the converged
image can be written down in float64 numpy from the reference's shaders alone:

  * sun light (light_sun_sample_visible.rcall): no distance falloff, pdf 1, a parallel shadow
  * area light as the reference has it (light_area_sample_visible.rcall:32-63): uniform point on a uniformly picked triangle,
    "area" 0.5 * vec3.length() = 1.5 whatever the triangle (Q1), direction FROM the light THROUGH the shaded point (Q2: the shadow ray
    leaves the point away from the light, so an occluder between light and wall casts no shadow and is itself black when something
    lies behind it)
  * mirror under a graded sky (mat_mirror_sample_value.rcall + path_trace.rgen:170-179): a specular bounce takes a launch of its
    own, carries importance F(cos) and collects the sky in the reflected direction on the next launch -- half the launches contribute
  * glass quad under a constant sky (mat_glass_sample_value.rcall): reflection with probability F and weight 1, transmission with
    probability 1 - F and weight etai^2 / etat^2 -- a "furnace" whose value is 1/2 L (F + (1 - F) / ior^2) whatever the directions are

Only spectral steps (RGB -> spectrum -> RGB, the conductor's Fresnel spectrum) are taken from the oracle's own routines, which
tests/test_oracle_math.py checks against the formulas; geometry, cosines, densities, probabilities and the launch accounting are
restated here.  The oracle renders on the CPU, the HIP path on the GPU; both must meet the same numbers.
"""
import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, make_light, make_material
from glaze_amd.scenes import cube_scene
from oracle import pyoracle
from oracle.pyoracle import OracleRenderer, OracleScene

WALL_MAT = 2
ALBEDO = (150, 150, 150)
WALL_Z, WALL_H = 1.0, 1.6          # the wall z = 1, |x|, |y| <= 1.6: more than the 90 degree view sees
OCC_Z, OCC_H = 0.5, 0.1            # a small square in front of it


def quad_scene(quads, lights, materials=None, textures=None):
    """cube_scene()'s camera (at the origin, looking down + z, 90 degrees), meta and texture 0 around a list of axis-parallel squares
    (centre, half size, z-normal sign, material): two triangles each, fan order, like every quad of the fixtures."""
    desc = cube_scene(material_type=abi.MAT_LAMBERT)
    m = desc.materials[WALL_MAT]
    m.diffuse = 0
    m.diffuse_mul[:3] = ALBEDO
    if materials:
        desc.materials = desc.materials + materials
    if textures:
        desc.textures = desc.textures + textures
    verts, indices, meshes, instances = [], [], [], []
    for i, (centre, half, nz, material) in enumerate(quads):
        cx, cy, cz = centre
        base = len(verts)
        for x, y in ((-half, -half), (half, -half), (half, half), (-half, half)):
            verts.append(((cx + x, cy + y, cz), (0.0, 0.0, float(nz)), (0.5 + x, 0.5 + y)))
        meshes.append((i, material, len(indices), 6))
        indices += [base, base + 1, base + 2, base, base + 2, base + 3]
        instances.append((i, 0))
    desc.vertices = np.array(verts, VERTEX_DTYPE)
    desc.indices = np.array(indices, np.uint32)
    desc.meshes = np.array(meshes, MESH_DTYPE)
    desc.instances = np.array(instances, INSTANCE_DTYPE)
    desc.lights = lights
    return desc


def spectrum_rgb(sp):
    out = np.zeros(3, np.float32)
    sp = np.ascontiguousarray(sp, np.float32)
    pyoracle.lib().orc_dev_rgb(sp.ctypes.data, out.ctypes.data)
    return out.astype(np.float64)


def lambert_times_emission(desc, light_index, at, rand3=(0.0, 0.0, 0.0)):
    """rgb of (Lambert value of the wall) x (the light's emission as sampled from `at`), the sample's distance and density"""
    o = OracleScene(desc)
    up = np.array([[0.0, 0.0, 1.0]], np.float32)
    value, pdf = o.bsdf_value(WALL_MAT, up, up)
    assert pdf[0] > 0
    _, dist, lpdf, em = o.light_sample(light_index, np.asarray(at, np.float32)[None, :], np.asarray(rand3, np.float32)[None, :], scene_radius=float(desc.meta.scene_radius))
    return spectrum_rgb(value[0] * em[0]), float(dist[0]), float(lpdf[0])


def pixel_grid(n, sub=1):
    c = (np.arange(n * sub) + 0.5) / (n * sub) * 2.0 - 1.0
    return np.meshgrid(c, c, indexing="xy")


def mean_image(img):
    assert (img[..., 3] > 0).all()
    return img[..., :3].astype(np.float64) / img[..., 3:4]


def render_oracle(desc, n, launches, integrator, depth=None, seed=3):
    o = OracleRenderer(OracleScene(desc), n, n)
    o.set_integrator(integrator)
    if depth:
        o.set_depth(depth)
    o.set_seed(seed)
    o.step(launches)
    return o.read_hdr()


def render_hip(instance, desc, n, launches, integrator, depth=None, seed=3):
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), n, n)
    r.set_integrator(glaze_amd.Integrator.DIRECT if integrator == abi.DIRECT else glaze_amd.Integrator.PATH_TRACE)
    if depth:
        r.set_depth(depth)
    r.set_seed(seed)
    r.step(launches)
    return r.read_hdr()


# The closed forms are stated without knowing which way the image's axes run: every scene here is asymmetric in x AND y, and exactly one
# of the four orientations must fit.  It is the one the point-light room found for x (world + x to the LEFT of a camera that looks down
# + z with + y up: a right-handed look-at) with row 0 at the TOP (world + y up: the projection's [1][1] is negated, raytracer.rs:1100-1103).
ORIENTATIONS = [(False, False), (True, False), (False, True), (True, True)]
EXPECTED_ORIENTATION = ORIENTATIONS.index((True, True))


def fits_one_orientation(got, forms, tol, what):
    """`forms`: the closed form for each orientation: (expected, mask of pixels safely inside one region, mask of the pixels that must be exactly zero)."""
    fit = []
    for want, safe, zero in forms:
        lit = safe & ~zero
        rel = np.abs(got[lit] - want[lit]) / want[lit]
        fit.append(bool(rel.max() < tol and (got[safe & zero] == 0.0).all()))
    assert sum(fit) == 1, "%s: %d orientations of the image axes fit the closed form (expected exactly one)" % (what, sum(fit))
    assert fit.index(True) == EXPECTED_ORIENTATION, "%s: the image is mirrored (orientation %s fits)" % (what, ORIENTATIONS[fit.index(True)])
    return fit.index(True)


# ---------------------------------------------------------------------------------------------------------------------
# the sun
# ---------------------------------------------------------------------------------------------------------------------
SUN_DIR = np.array([0.6, -0.4, 0.7]) / np.linalg.norm([0.6, -0.4, 0.7])      # RTLight::dir: the way the light travels
SUN_I = 0.7


def sun_scene():
    return quad_scene([((0, 0, WALL_Z), WALL_H, -1, WALL_MAT), ((0, 0, OCC_Z), OCC_H, -1, WALL_MAT)],
                      [make_light(abi.LIGHT_SUN, "sun", direction=tuple(float(x) for x in SUN_DIR), intensity=SUN_I)])


def sun_forms(n, factor):
    out = []
    wi = -SUN_DIR                                                 # sam.wiW = -light.dir
    for flip_x, flip_y in ORIENTATIONS:
        u, v = pixel_grid(n)
        u, v = (-u if flip_x else u), (-v if flip_y else v)
        on_occ = (np.abs(u) * OCC_Z < OCC_H) & (np.abs(v) * OCC_Z < OCC_H)
        z = np.where(on_occ, OCC_Z, WALL_Z)
        p = np.stack([u * z, v * z, z], -1)
        e = np.full(u.shape, abs(wi[2]))                          # |dot(wiW, n)| / pdf, pdf = 1, both normals (0, 0, -1); no falloff
        # the shadow: the point's ray towards the sun crosses the plane of the square inside it
        s = (OCC_Z - p[..., 2]) / wi[2]
        q = p + s[..., None] * wi
        shadow = ~on_occ & (s > 0) & (np.abs(q[..., 0]) < OCC_H) & (np.abs(q[..., 1]) < OCC_H)
        m = 3.0 / n
        near_edge = ((np.abs(np.abs(u) * OCC_Z - OCC_H) < m) & (np.abs(v) * OCC_Z < OCC_H + m)) | ((np.abs(np.abs(v) * OCC_Z - OCC_H) < m) & (np.abs(u) * OCC_Z < OCC_H + m))
        near_shadow = ~on_occ & (((np.abs(np.abs(q[..., 0]) - OCC_H) < m) & (np.abs(q[..., 1]) < OCC_H + m)) | ((np.abs(np.abs(q[..., 1]) - OCC_H) < m) & (np.abs(q[..., 0]) < OCC_H + m)))
        out.append((e[..., None] * factor[None, None, :], ~near_edge & ~near_shadow, shadow))
    return out


def check_sun(img, n):
    desc = sun_scene()
    factor, dist, pdf = lambert_times_emission(desc, 0, (0.0, 0.0, 1.0))
    assert pdf == 1.0 and abs(dist - (2.0 * float(desc.meta.scene_radius) + 1.0)) < 1e-5      # light_sun_sample_visible.rcall: pdf 1, distance 2 r + 1
    got = mean_image(img)
    forms = sun_forms(n, factor)
    assert all(int((s & z).sum()) >= 6 for _, s, z in forms)                               # the shadow is in view
    fits_one_orientation(got, forms, 0.005, "sun")                                         # constant irradiance x cosine: nothing to average


def test_oracle_sun_matches_the_closed_form():
    check_sun(render_oracle(sun_scene(), 96, 12, abi.DIRECT), 96)


@pytest.mark.gpu
def test_hip_sun_matches_the_closed_form(instance):
    check_sun(render_hip(instance, sun_scene(), 256, 16, abi.DIRECT), 256)


# ---------------------------------------------------------------------------------------------------------------------
# the area light, as the reference has it
# ---------------------------------------------------------------------------------------------------------------------
EMIT_MAT = 3
AREA_C, AREA_H, AREA_I = (1.0, -0.7, 1.6), 0.25, 0.9          # a square BEHIND the wall, off axis


def area_scene():
    return quad_scene([((0, 0, WALL_Z), WALL_H, -1, WALL_MAT), ((0, 0, OCC_Z), OCC_H, -1, WALL_MAT), (AREA_C, AREA_H, -1, EMIT_MAT)],
                      [make_light(abi.LIGHT_AREA, "area", resource_id=EMIT_MAT, intensity=AREA_I)],
                      materials=[make_material("emitter", diffuse_mul=(255, 200, 150))])


def area_forms(n, factor_unit, sub=20):
    """What the reference's area light does (kept, Q1 / Q2): wiW = normalize(position - rand_point) runs from the light THROUGH the
    shaded point, and the Lambert density is zero unless wiW and the viewer are on the same side of the surface
    (mat_lambert_value.rcall: same_hemi) -- so a light in front of a wall gives it nothing, and a light BEHIND the wall lights the side the
    camera sees.  The shadow ray leaves the point along wiW, towards the camera's side, for the distance to the light: the small square
    between camera and wall casts a (blurred) shadow on the wall from a light that is behind the wall, and is itself lit through it.
    Expected value: mean over the light's points q (uniform on the square: two congruent triangles picked with equal probability,
    uniform inside) of visible x |cos| / d^2, times 1 / pdf = triangles x 1.5 (Q1: 0.5 * vec3.length())."""
    g = (np.arange(sub) + 0.5) / sub * 2.0 - 1.0
    qx, qy = np.meshgrid(AREA_C[0] + AREA_H * g, AREA_C[1] + AREA_H * g, indexing="xy")
    q = np.stack([qx.ravel(), qy.ravel(), np.full(qx.size, AREA_C[2])], -1)
    out = []
    for flip_x, flip_y in ORIENTATIONS:
        u, v = pixel_grid(n)
        u, v = (-u if flip_x else u), (-v if flip_y else v)
        on_occ = (np.abs(u) * OCC_Z < OCC_H) & (np.abs(v) * OCC_Z < OCC_H)
        z = np.where(on_occ, OCC_Z, WALL_Z)
        p = np.stack([u * z, v * z, z], -1)
        w = p[:, :, None, :] - q[None, None, :, :]                # position - rand_point
        d2 = (w ** 2).sum(-1)
        cos = np.abs(w[..., 2]) / np.sqrt(d2)
        # the shadow ray p + t w, 0 < t < 1 (it is as long as the distance to the light), meets the plane of the small square at t = s
        s = (OCC_Z - p[:, :, None, 2]) / w[..., 2]
        r = p[:, :, None, :] + s[..., None] * w
        blocked = ~on_occ[:, :, None] & (s > 0) & (s < 1) & (np.abs(r[..., 0]) < OCC_H) & (np.abs(r[..., 1]) < OCC_H)
        e = (np.where(blocked, 0.0, cos / d2)).mean(-1) * 2.0 * 1.5   # 1 / (select_pdf * area_pdf) = triangles x "area"
        m = 3.0 / n
        near_edge = ((np.abs(np.abs(u) * OCC_Z - OCC_H) < m) & (np.abs(v) * OCC_Z < OCC_H + m)) | ((np.abs(np.abs(v) * OCC_Z - OCC_H) < m) & (np.abs(u) * OCC_Z < OCC_H + m))
        out.append((e[..., None] * factor_unit[None, None, :], ~near_edge, np.zeros_like(on_occ), blocked.mean(-1)))
    return out


def check_area(img, n):
    desc = area_scene()
    v0 = np.array([AREA_C[0] - AREA_H, AREA_C[1] - AREA_H, AREA_C[2]])                    # rand = (0, 0, 0): the first vertex of the first triangle
    factor_unit, dist, pdf = lambert_times_emission(desc, 0, v0 + np.array([0.0, 0.0, -1.0]))
    assert abs(dist - 1.0) < 1e-6 and abs(pdf - 1.0 / (2 * 1.5)) < 1e-6                    # Q1: the density of a point is 1 / (triangles x 1.5)
    got = mean_image(img)
    forms = area_forms(n, factor_unit)
    shade = forms[EXPECTED_ORIENTATION][3]
    assert shade.max() > 0.15 and (shade > 0.08).sum() >= 20                                # the (blurred) shadow is in view: up to a fifth of the light is hidden
    # Monte Carlo over the light's surface: compared on the mean over 8 x 8 blocks of pixels and all launches
    fits_blocks(got, [f[:3] for f in forms], 8, 0.03)


def fits_blocks(got, forms, b, tol):
    fit = []
    n = got.shape[0]
    for want, safe, zero in forms:
        lit = safe & ~zero
        blocks_ok = lit.reshape(n // b, b, n // b, b).all(axis=(1, 3))
        gm = got[..., 1].reshape(n // b, b, n // b, b).mean(axis=(1, 3))
        wm = want[..., 1].reshape(n // b, b, n // b, b).mean(axis=(1, 3))
        rel = np.abs(gm[blocks_ok] - wm[blocks_ok]) / wm[blocks_ok]
        fit.append(bool(rel.max() < tol and (got[safe & zero] == 0.0).all()))
    assert sum(fit) == 1, "%d orientations fit" % sum(fit)
    assert fit.index(True) == EXPECTED_ORIENTATION
    return fit.index(True)


def test_oracle_area_light_matches_the_closed_form():
    check_area(render_oracle(area_scene(), 64, 400, abi.DIRECT), 64)


@pytest.mark.gpu
def test_hip_area_light_matches_the_closed_form(instance):
    check_area(render_hip(instance, area_scene(), 128, 600, abi.DIRECT), 128)


# ---------------------------------------------------------------------------------------------------------------------
# a mirror under a graded sky
# ---------------------------------------------------------------------------------------------------------------------
MIRROR_MAT = 3
SKY_W, SKY_H = 64, 128


def sky_rows():
    return np.round(60.0 + 150.0 * np.arange(SKY_H) / (SKY_H - 1)).astype(np.uint8)       # darker at the pole theta = 0, linear in the row


def mirror_scene(mtype=abi.MAT_MIRROR, sky=None):
    tex = np.repeat(sky_rows()[:, None, None], SKY_W, axis=1).repeat(4, axis=2) if sky is None else np.full((SKY_H, SKY_W, 4), sky, np.uint8)
    tex = np.ascontiguousarray(tex)
    tex[..., 3] = 255
    return quad_scene([((0, 0, WALL_Z), WALL_H, -1, MIRROR_MAT)],
                      [make_light(abi.LIGHT_SKY, "sky", resource_id=2, intensity=1.0)],
                      materials=[make_material("specular", mtype=mtype, metal=2, ior=1.46)],
                      textures=[(abi.TEX_RGBA_NORM, tex, "graded sky")])


def sky_grey(direction):
    """sky_radiance (path_trace.rgen:75-82) for a texture that only depends on the row: bilinear between row centres, REPEAT addressing"""
    w = direction / np.linalg.norm(direction, axis=-1, keepdims=True)
    theta = np.arccos(np.clip(w[..., 2], -1.0, 1.0))
    y = theta / np.pi * SKY_H - 0.5
    y0 = np.floor(y)
    f = y - y0
    rows = sky_rows().astype(np.float64) / 255.0
    return rows[(y0.astype(int)) % SKY_H] * (1.0 - f) + rows[(y0.astype(int) + 1) % SKY_H] * f


def check_mirror(img, n, launches):
    desc = mirror_scene()
    o = OracleScene(desc)
    # the conductor's Fresnel spectrum at a ladder of cosines, from the oracle's own routine: value = F / |cos|, pdf 1 (mat_mirror_sample_value.rcall)
    cosines = np.linspace(0.3, 1.0, 141)
    wo = np.stack([np.sqrt(1.0 - cosines ** 2), np.zeros_like(cosines), cosines], -1)
    wi, value, pdf = o.bsdf_sample(MIRROR_MAT, wo, np.zeros((len(cosines), 3)))
    assert (pdf == 1.0).all() and np.allclose(wi, wo * np.array([-1.0, -1.0, 1.0]), atol=1e-6)
    white = np.zeros(16, np.float32)
    one = np.ones(3, np.float32)
    pyoracle.lib().orc_dev_from_illuminant_color(one.ctypes.data, white.ctypes.data)
    ladder = np.stack([spectrum_rgb(value[i] * cosines[i] * white) for i in range(len(cosines))])      # rgb(F(cos) x the sky's spectrum at grey 1)
    u, v = pixel_grid(n)
    d = np.stack([u, v, np.ones_like(u)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cos = d[..., 2]                                                # the wall's normal is (0, 0, -1)
    refl = d * np.array([1.0, 1.0, -1.0])
    f_rgb = np.stack([np.interp(cos, cosines, ladder[:, c]) for c in range(3)], -1)
    want = 0.5 * f_rgb * sky_grey(refl)[..., None]                 # one launch hits the mirror, the next one collects the sky: half of them contribute
    assert launches % 2 == 0
    got = mean_image(img)
    assert (img[..., 3] == launches).all()
    rel = np.abs(got - want) / want
    # (the largest deviations sit at the image centre: the ray reflected straight back looks at the sky's pole, where the texture's last
    # row blends into its first -- REPEAT addressing -- and a pixel's jittered launches average a kink the pixel centre does not see)
    away = (u * u + v * v) > 0.02
    assert rel[away].max() < 0.01 and rel.max() < 0.15 and rel.mean() < 0.003, (rel[away].max(), rel.max(), rel.mean())
    assert got[..., 1].max() / got[..., 1].min() > 1.15            # the gradient of the sky is really in the picture


def test_oracle_mirror_under_a_graded_sky_matches_the_closed_form():
    check_mirror(render_oracle(mirror_scene(), 64, 16, abi.PATH_TRACE, depth=6), 64, 16)


@pytest.mark.gpu
def test_hip_mirror_under_a_graded_sky_matches_the_closed_form(instance):
    check_mirror(render_hip(instance, mirror_scene(), 256, 24, abi.PATH_TRACE, depth=6), 256, 24)


# ---------------------------------------------------------------------------------------------------------------------
# a glass quad under a constant sky: a furnace
# ---------------------------------------------------------------------------------------------------------------------
GLASS_IOR, SKY_GREY = 1.46, 200


def fresnel_dielectric(cos_i, eta_i, eta_t):
    """the unpolarised Fresnel reflectance of a dielectric interface (textbook form; fresnel.glsl:19-35 is checked against it in test_oracle_math.py)"""
    sin_t2 = (eta_i / eta_t) ** 2 * (1.0 - cos_i ** 2)
    cos_t = np.sqrt(np.maximum(0.0, 1.0 - sin_t2))
    r_par = (eta_t * cos_i - eta_i * cos_t) / (eta_t * cos_i + eta_i * cos_t)
    r_perp = (eta_i * cos_i - eta_t * cos_t) / (eta_i * cos_i + eta_t * cos_t)
    return 0.5 * (r_par ** 2 + r_perp ** 2)


def check_glass(img, n, launches):
    sky = np.zeros(16, np.float32)
    g = np.full(3, SKY_GREY / 255.0, np.float32)
    pyoracle.lib().orc_dev_from_illuminant_color(g.ctypes.data, sky.ctypes.data)
    l_rgb = spectrum_rgb(sky)
    u, v = pixel_grid(n)
    cos = 1.0 / np.sqrt(u * u + v * v + 1.0)
    f = fresnel_dielectric(cos, 1.0, GLASS_IOR)
    # reflected with probability F and weight (F / |cos|) |cos| / F = 1; transmitted with probability 1 - F and weight
    # (1 - F) etai^2 / (etat^2 |cos'|) x |cos'| / (1 - F) = 1 / ior^2; either way the next launch misses and collects the constant sky
    want = 0.5 * (f + (1.0 - f) / GLASS_IOR ** 2)[..., None] * l_rgb[None, None, :]
    got = mean_image(img)
    assert (img[..., 3] == launches).all() and np.isfinite(got).all()
    # a pixel's paths take one of two values: the mean over 8 x 8 pixels and all its launches is what converges
    b = 8
    gm = got.reshape(n // b, b, n // b, b, 3).mean(axis=(1, 3))
    wm = want.reshape(n // b, b, n // b, b, 3).mean(axis=(1, 3))
    rel = np.abs(gm - wm) / wm
    assert rel.max() < 0.03 and rel.mean() < 0.008, (rel.max(), rel.mean())
    # and nothing is ever brighter than an all-reflected or darker than an all-transmitted pixel
    assert (got <= 0.5 * l_rgb * 1.0001).all() and (got >= 0.5 * l_rgb / GLASS_IOR ** 2 * 0.9999).all()


def test_oracle_glass_furnace_matches_the_closed_form():
    check_glass(render_oracle(mirror_scene(abi.MAT_GLASS, sky=SKY_GREY), 64, 200, abi.PATH_TRACE, depth=6), 64, 200)


@pytest.mark.gpu
def test_hip_glass_furnace_matches_the_closed_form(instance):
    check_glass(render_hip(instance, mirror_scene(abi.MAT_GLASS, sky=SKY_GREY), 128, 400, abi.PATH_TRACE, depth=6), 128, 400)
