"""The oracle against MATHEMATICS (CPU only).

Render parity of the HIP path is defined against oracle/oracle.cpp, and no reference test pins a BSDF value, a light sample or a
pixel: a misreading of the GLSL on the oracle's side would be invisible to every HIP-vs-oracle test.  These tests
look at the oracle's shading routines from the other side: independent float64 numpy statements of the published formulas
(GGX D / Lambda / VNDF pdf, Fresnel, cosine sampling, piecewise-constant 2D distributions), integral identities (a pdf
integrates to one, a white Lambertian furnace returns its albedo), and agreement between each BSDF's `sample` and `value`
callables at the sampled direction.  The reference's quirks are the allow-list: each one that is
observable at this level has a test that FAILS if the quirk is "fixed", next to the statement of what the correct formula
would give.
"""
import numpy as np
import pytest

from glaze_amd import abi
from glaze_amd.scene_desc import make_camera, make_light, make_material
from glaze_amd.scenes import cube_scene
from oracle import pyoracle
from oracle.pyoracle import OracleRenderer, OracleScene

CUBE_MAT = 2          # the cube's own material in cube_scene()


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def hemisphere_dirs(n, rng, side=+1.0, min_cos=0.05):
    """uniformly distributed unit vectors with |z| >= min_cos on the given side"""
    z = rng.uniform(min_cos, 1.0, n)
    phi = rng.uniform(0, 2 * np.pi, n)
    r = np.sqrt(1 - z * z)
    return np.stack([r * np.cos(phi), r * np.sin(phi), side * z], -1).astype(np.float32)


def scene_with(mtype, **kw):
    desc = cube_scene(material_type=mtype)
    m = desc.materials[CUBE_MAT]
    m.diffuse = 0                                 # the 1x1 white default texture: colour = diffuse_mul
    for k, v in kw.items():
        if k == "diffuse_mul":
            m.diffuse_mul[:] = v
        else:
            setattr(m, k, v)
    return desc, OracleScene(desc)


# ---------------------------------------------------------------------------------------------------------------------
# independent statements of the microfacet formulas (float64)
# ---------------------------------------------------------------------------------------------------------------------
def ggx_terms(v, ax, ay):
    c2 = v[..., 2] ** 2
    s2 = np.maximum(0.0, 1.0 - c2)
    t2 = s2 / c2
    with np.errstate(invalid="ignore", divide="ignore"):
        cos2p = np.where(s2 > 0, v[..., 0] ** 2 / s2, 1.0)
        sin2p = np.where(s2 > 0, v[..., 1] ** 2 / s2, 0.0)
    return c2, t2, cos2p, sin2p


def ggx_d(wh, ax, ay):
    c2, t2, cos2p, sin2p = ggx_terms(wh, ax, ay)
    e = 1.0 + (cos2p / ax ** 2 + sin2p / ay ** 2) * t2
    return 1.0 / (np.pi * ax * ay * c2 * c2 * e * e)


def ggx_lambda(v, ax, ay):
    _, t2, cos2p, sin2p = ggx_terms(v, ax, ay)
    return (-1.0 + np.sqrt(1.0 + t2 * (cos2p * ax ** 2 + sin2p * ay ** 2))) / 2.0


def vndf_pdf(wo, wi, ax, ay, quirk_q6):
    """pdf of wi for visible-normal sampling of GGX; quirk_q6: G1 evaluated at wh instead of wo (microfacets.glsl:94-99)"""
    wh = unit(wo + wi)
    d = ggx_d(wh, ax, ay)
    dot = np.sum(wo * wh, -1)
    if quirk_q6:
        g1 = 1.0 / (1.0 + ggx_lambda(wh, ax, ay))
        pdf_h = d * g1 * np.abs(dot) / np.abs(wh[..., 2])
    else:
        g1 = 1.0 / (1.0 + ggx_lambda(wo, ax, ay))
        pdf_h = d * g1 * np.abs(dot) / np.abs(wo[..., 2])
    return pdf_h / (4.0 * dot)


def hemisphere_quadrature(f, n_theta=600, n_phi=1200):
    """integral of f(w) over the upper hemisphere, midpoint rule in (cos theta, phi)"""
    cz = (np.arange(n_theta) + 0.5) / n_theta
    phi = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    CZ, PH = np.meshgrid(cz, phi, indexing="ij")
    r = np.sqrt(1 - CZ * CZ)
    w = np.stack([r * np.cos(PH), r * np.sin(PH), CZ], -1)
    return float(f(w).sum() * (1.0 / n_theta) * (2 * np.pi / n_phi))


def fresnel_dielectric(c, eta_i, eta_t):
    s = eta_i ** 2 / eta_t ** 2 * np.maximum(0.0, 1 - c * c)
    ct = np.sqrt(np.maximum(0.0, 1 - s))
    rpar = (eta_t * c - eta_i * ct) / (eta_t * c + eta_i * ct)
    rper = (eta_i * c - eta_t * ct) / (eta_i * c + eta_t * ct)
    return np.where(s >= 1.0, 1.0, (rpar ** 2 + rper ** 2) / 2)


# ---------------------------------------------------------------------------------------------------------------------
# Lambert
# ---------------------------------------------------------------------------------------------------------------------
def test_lambert_sample_matches_value_and_cosine_law():
    desc, o = scene_with(abi.MAT_LAMBERT, diffuse_mul=(204, 102, 51))
    rng = np.random.default_rng(1)
    for side in (+1.0, -1.0):
        wo = hemisphere_dirs(2000, rng, side)
        r3 = rng.random((2000, 3)).astype(np.float32)
        wi, val_s, pdf_s = o.bsdf_sample(CUBE_MAT, wo, r3)
        val_e, pdf_e = o.bsdf_value(CUBE_MAT, wo, wi)
        assert np.allclose(np.linalg.norm(wi, axis=1), 1.0, atol=2e-6)
        assert (np.sign(wi[:, 2]) == side).all()                                  # sampled on wo's side (mat_lambert_sample_value.rcall:19-29)
        assert np.allclose(pdf_s, np.abs(wi[:, 2]) / np.pi, rtol=2e-6)            # cosine law
        assert np.array_equal(val_s, val_e)                                       # the two callables state the same value
        assert np.allclose(pdf_e, pdf_s, rtol=3e-6, atol=1e-9)                    # (eval re-derives |wi.z| from the normalised world vector)
    # malley's method: the disk point is (sqrt(r.y) cos(2 pi r.x), sqrt(r.y) sin(2 pi r.x))
    r3 = np.array([[0.25, 0.49, 0.0]], np.float32)
    wi, _, _ = o.bsdf_sample(CUBE_MAT, [[0, 0, 1]], r3)
    assert np.allclose(wi[0], [0.0, 0.7, np.sqrt(1 - 0.49)], atol=2e-6)


def test_lambert_white_furnace_and_pdf_normalisation():
    """Each cosine-weighted sample returns value * |cos| / pdf = pi * S(albedo / pi): the furnace estimate has zero variance, equals the
    upsampled albedo (Smits-style basis x 0.94, spectrum.glsl:202-242) and never exceeds one for a white surface."""
    desc, o = scene_with(abi.MAT_LAMBERT, diffuse_mul=(255, 255, 255))
    rng = np.random.default_rng(2)
    wo = hemisphere_dirs(512, rng)
    wi, val, pdf = o.bsdf_sample(CUBE_MAT, wo, rng.random((512, 3)).astype(np.float32))
    est = val * (np.abs(wi[:, 2]) / pdf)[:, None]
    albedo = np.zeros(16, np.float32)
    pyoracle.lib().orc_dev_from_surface_color(np.array([1, 1, 1], np.float32).ctypes.data, albedo.ctypes.data)
    assert np.allclose(est, albedo[None, :], rtol=3e-6)
    # the white basis is ~1.062 in every bin and the result is scaled by 0.94 (spectrum.glsl:204-208, :241): 0.9965 ... 0.9988 <= 1
    assert est.max() <= 1.0 and np.allclose(albedo, 1.0615 * 0.94, atol=2.5e-3)    # energy conserving, spectrally flat
    # the eval pdf integrates to one over the hemisphere of wo (quadrature over the oracle's own values)
    cz = (np.arange(256) + 0.5) / 256
    phi = (np.arange(64) + 0.5) / 64 * 2 * np.pi
    CZ, PH = np.meshgrid(cz, phi, indexing="ij")
    w = np.stack([np.sqrt(1 - CZ ** 2) * np.cos(PH), np.sqrt(1 - CZ ** 2) * np.sin(PH), CZ], -1).reshape(-1, 3).astype(np.float32)
    _, pdf_e = o.bsdf_value(CUBE_MAT, np.tile([[0.3, 0.1, 0.948]], (w.shape[0], 1)).astype(np.float32), w)
    assert abs(pdf_e.astype(np.float64).sum() * (1 / 256) * (2 * np.pi / 64) - 1.0) < 2e-3


def test_q9_lambert_value_ignores_the_hemisphere():
    """Q9 (mat_lambert_value.rcall:27-33): only the pdf is masked when wi is on the other side; the value is not."""
    desc, o = scene_with(abi.MAT_LAMBERT)
    val_same, pdf_same = o.bsdf_value(CUBE_MAT, [[0.2, 0.1, 0.97]], [[0.3, -0.2, 0.93]])
    val_opp, pdf_opp = o.bsdf_value(CUBE_MAT, [[0.2, 0.1, 0.97]], [[0.3, -0.2, -0.93]])
    assert pdf_same[0] > 0 and pdf_opp[0] == 0.0
    assert val_opp[0].min() > 0 and np.array_equal(val_opp, val_same)            # a "fixed" shader would return zero here


# ---------------------------------------------------------------------------------------------------------------------
# GGX: metal (conductor) and the Q6 pdf
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rough,aniso", [(0.35, 0.0), (0.6, 0.4), (0.15, -0.3)])
def test_metal_sample_matches_value_and_the_vndf_formulas(rough, aniso):
    desc, o = scene_with(abi.MAT_METAL, roughness_mul=rough, anisotropy=aniso, metal=3)
    ax, ay = rough * (1 + aniso), rough * (1 - aniso)                             # microfacets.glsl:122-125
    rng = np.random.default_rng(3)
    wo = hemisphere_dirs(4000, rng, min_cos=0.1)
    wi, val_s, pdf_s = o.bsdf_sample(CUBE_MAT, wo, rng.random((4000, 3)).astype(np.float32))
    ok = (pdf_s > 0) & (wi[:, 2] * wo[:, 2] > 0)
    assert ok.mean() > 0.4                                                         # rough lobes at grazing wo reflect part of their mass below the horizon
    val_e, pdf_e = o.bsdf_value(CUBE_MAT, wo[ok], wi[ok])
    assert np.allclose(pdf_e, pdf_s[ok], rtol=2e-3)                               # wh is re-derived from wo + wi in the value callable
    assert np.allclose(val_e, val_s[ok], rtol=4e-3, atol=1e-7)
    # the pdf the reference computes is the VNDF pdf with G1 taken at wh (Q6); independent float64 statement
    want = vndf_pdf(wo[ok].astype(np.float64), wi[ok].astype(np.float64), ax, ay, quirk_q6=True)
    assert np.allclose(pdf_e, want, rtol=2e-3)
    # the sampled half vectors follow the TRUE visible-normal distribution: E_wi[ 1 / pdf_true ] over samples = solid angle reached,
    # checked through the identity  mean(pdf_quirk / pdf_true) = integral of pdf_quirk  (importance sampling with pdf_true)
    true = vndf_pdf(wo[ok].astype(np.float64), wi[ok].astype(np.float64), ax, ay, quirk_q6=False)
    assert np.isfinite(true).all() and (true > 0).all()


def test_q6_ggx_pdf_uses_g1_of_the_half_vector():
    """Q6 (microfacets.glsl:94-99, the author's own TODO): with G1(wo) the density of the reflected direction integrates to one over
    the sphere of directions (GGX tails send part of it below the horizon); with G1(wh), what the reference computes, it does not.
    The oracle keeps the quirk."""
    rough = 0.4
    ax = ay = rough
    wo = unit([0.3, 0.0, 0.954])
    def integral(quirk):
        total = 0.0
        for flip in (1.0, -1.0):
            def f(w):
                w = w * np.array([1.0, 1.0, flip])
                p = vndf_pdf(np.broadcast_to(wo, w.shape), w, ax, ay, quirk)
                return np.where(np.isfinite(p) & (p > 0), p, 0.0)
            total += hemisphere_quadrature(f)
        return total
    i_true, i_quirk = integral(False), integral(True)
    assert abs(i_true - 1.0) < 1e-2                                               # a proper density
    assert abs(i_quirk - 1.0) > 0.1                                               # the quirk's "pdf" is not normalised (1.167 here)
    desc, o = scene_with(abi.MAT_METAL, roughness_mul=rough, anisotropy=0.0)
    rng = np.random.default_rng(4)
    wi = hemisphere_dirs(3000, rng, min_cos=0.05)
    _, pdf_e = o.bsdf_value(CUBE_MAT, np.tile(wo.astype(np.float32), (3000, 1)), wi)
    q = vndf_pdf(np.broadcast_to(wo, (3000, 3)), wi.astype(np.float64), ax, ay, True)
    t = vndf_pdf(np.broadcast_to(wo, (3000, 3)), wi.astype(np.float64), ax, ay, False)
    assert np.allclose(pdf_e, q, rtol=2e-3)
    assert not np.allclose(pdf_e, t, rtol=2e-2)                                   # fails the day somebody "fixes" Q6 in the oracle only


def test_ggx_normal_distribution_is_normalised():
    """integral of D(wh) cos(theta_h) over the hemisphere = 1 for the D of microfacets.glsl:57-69 (independent statement used above)"""
    for ax, ay in ((0.3, 0.3), (0.7, 0.2)):
        assert abs(hemisphere_quadrature(lambda w: ggx_d(w, ax, ay) * w[..., 2]) - 1.0) < 3e-3


def test_q5_uber_applies_roughness_mul_twice():
    """Q5 (mat_uber_value.rcall:26,31): roughness = tex * mul, then to_anisotropic(roughness * mul): alpha = mul^2 for a white texture."""
    mul = 0.7
    desc, o = scene_with(abi.MAT_UBER, roughness_mul=mul, anisotropy=0.0, metalness_mul=1.0)
    rng = np.random.default_rng(5)
    wo = hemisphere_dirs(2000, rng, min_cos=0.2)
    wi = hemisphere_dirs(2000, rng, min_cos=0.2)
    _, pdf = o.bsdf_value(CUBE_MAT, wo, wi, rand=np.zeros(2000, np.float32))     # rand < 0.5: the specular lobe, pdf x 0.5
    a2, a1 = mul * mul, mul
    p2 = 0.5 * vndf_pdf(wo.astype(np.float64), wi.astype(np.float64), a2, a2, True)
    p1 = 0.5 * vndf_pdf(wo.astype(np.float64), wi.astype(np.float64), a1, a1, True)
    assert np.allclose(pdf, p2, rtol=3e-3)
    assert not np.allclose(pdf, p1, rtol=3e-2)


# ---------------------------------------------------------------------------------------------------------------------
# specular BSDFs
# ---------------------------------------------------------------------------------------------------------------------
def _rt_material_spectra(o, material_id):
    """metal_ior (n) and metal_fresnel (n^2 + k^2) of an RTMaterial record (raytrace_structures.rs:44-64: offsets 32 and 96)"""
    raw = o.rt_materials().reshape(-1, 208)[material_id]
    return raw[32:96].view(np.float32).astype(np.float64), raw[96:160].view(np.float32).astype(np.float64)


def test_mirror_reflects_about_the_normal_with_unit_pdf_and_q15_conductor_fresnel():
    """mat_mirror_sample_value.rcall:16-34.  The value is fresnel_conductor(cos) / |cos| with the formula AS WRITTEN in
    fresnel.glsl:7-17, which groups r_perp^2 = (a - (2 n c + c^2)) / (a + (2 n c + c^2)) and r_par^2 = (a c^2 - (2 n c + 1)) / (a c^2 + (2 n c + 1)),
    a = n^2 + k^2 -- the textbook approximation has a - 2 n c + c^2 and a c^2 - 2 n c + 1 in the numerators ("Q15": silver reflects
    1 ... 86 % instead of 95+ %).  The oracle states the shader's grouping; a physically "corrected" Fresnel would fail here."""
    desc, o = scene_with(abi.MAT_MIRROR, metal=0)
    rng = np.random.default_rng(6)
    wo = hemisphere_dirs(256, rng)
    wi, val, pdf = o.bsdf_sample(CUBE_MAT, wo, rng.random((256, 3)).astype(np.float32))
    assert np.allclose(wi, wo * np.array([-1, -1, 1], np.float32), atol=3e-7)
    assert (pdf == 1.0).all()
    n, a = _rt_material_spectra(o, CUBE_MAT)
    c = wo[:, 2:3].astype(np.float64)
    e = 2 * n[None, :] * c
    as_written = ((a - (e + c * c)) / (a + (e + c * c)) + (a * c * c - (e + 1)) / (a * c * c + (e + 1))) / 2
    textbook = ((a - e + c * c) / (a + e + c * c) + (a * c * c - e + 1) / (a * c * c + e + 1)) / 2
    refl = val.astype(np.float64) * np.abs(c)
    assert np.allclose(refl, as_written, rtol=2e-5, atol=1e-7)
    assert textbook.min() > 0.9 and np.abs(refl - textbook).max() > 0.3          # silver: what physics says, and how far the shader is from it
    _, pdf_e = o.bsdf_value(CUBE_MAT, wo, wi)
    assert (pdf_e == 0.0).all()                                                   # a delta lobe evaluates to nothing (mat_mirror_value.rcall:8-11)


def test_glass_branch_probabilities_and_q7_refraction():
    """Fresnel-weighted choice between reflection and transmission (mat_glass_sample_value.rcall:18-56), and Q7: refract() is
    called with the OUTWARD wo, so the transmitted direction keeps the sign of wo's tangential components (physics flips it)."""
    ior = 1.46
    desc, o = scene_with(abi.MAT_GLASS, ior=ior)
    eta_air = 1.000293
    rng = np.random.default_rng(7)
    for wo in (unit([0.5, 0.2, 0.84]).astype(np.float32), unit([0.1, -0.7, -0.7]).astype(np.float32)):
        outside = wo[2] >= 0
        ei, et = (eta_air, ior) if outside else (ior, eta_air)
        F = float(fresnel_dielectric(abs(float(wo[2])), ei, et))
        n = 4000
        r3 = rng.random((n, 3)).astype(np.float32)
        wi, val, pdf = o.bsdf_sample(CUBE_MAT, np.tile(wo, (n, 1)), r3)
        reflected = r3[:, 2] < np.float32(F)
        borderline = np.abs(r3[:, 2] - F) < 1e-5
        same_side = np.sign(wi[:, 2]) == np.sign(wo[2])
        assert (same_side == reflected)[~borderline].all()                         # rand.z < F reflects
        assert np.allclose(pdf[reflected & ~borderline], F, rtol=1e-4) and np.allclose(pdf[~reflected & ~borderline], 1 - F, rtol=1e-4)
        t = wi[~reflected & ~borderline]
        if F < 1.0 and t.size:
            eta = ei / et
            # GLSL refract(I = wo, N = (0, 0, sign wo.z), eta): tangential part = eta * wo.xy -- same sign as wo (Q7)
            assert np.allclose(t[:, :2], eta * wo[None, :2], atol=2e-6)
            assert (np.sign(t[:, 2]) == -np.sign(wo[2])).all()
            assert np.allclose(np.linalg.norm(t, axis=1), 1.0, atol=1e-5)
        # radiance scaling of the transmitted lobe: (1 - F) eta_i^2 / eta_t^2 / |cos_t|
        if t.size:
            want = (1 - F) * ei ** 2 / et ** 2 / np.abs(t[:, 2])
            assert np.allclose(val[~reflected & ~borderline][:, 0], want, rtol=2e-4)
            assert np.allclose(val[~reflected & ~borderline], val[~reflected & ~borderline][:, :1], rtol=0, atol=0)   # spectrally flat


# ---------------------------------------------------------------------------------------------------------------------
# lights
# ---------------------------------------------------------------------------------------------------------------------
def _light_scene(lights, extra_material=None):
    desc = cube_scene()
    if extra_material is not None:
        desc.materials.append(extra_material)
    desc.lights = lights
    return desc, OracleScene(desc)


def test_omni_light_inverse_square_law():
    desc, o = _light_scene([make_light(abi.LIGHT_OMNI, "omni", position=(0.3, 0.5, -0.2), intensity=2.0)])
    p = np.array([[0.0, 0.0, 0.0], [0.3, -0.5, -0.2], [1.3, 0.5, -0.2]], np.float32)
    wi, dist, pdf, em = o.light_sample(0, p, np.zeros((3, 3), np.float32))
    d = np.array([0.3, 0.5, -0.2]) - p
    assert np.allclose(wi, unit(d), atol=2e-7) and np.allclose(dist, np.linalg.norm(d, axis=1), rtol=2e-7) and (pdf == 1.0).all()
    assert np.allclose(em[:, 0] * dist ** 2, em[0, 0] * dist[0] ** 2, rtol=3e-6)      # emission x d^2 = colour x intensity, whatever the distance


def test_q14_sun_direction_is_not_normalised():
    """Q14 (scene.rs:1878-1887: `dir.normalize();` discards its result): wi = -dir as given, of any length."""
    desc, o = _light_scene([make_light(abi.LIGHT_SUN, "sun", direction=(0.0, -2.0, 0.0), intensity=0.5)])
    wi, dist, pdf, em = o.light_sample(0, [[0, 0, 0]], [[0.1, 0.2, 0.3]], scene_radius=3.0)
    assert np.allclose(wi[0], [0.0, 2.0, 0.0]) and pdf[0] == 1.0 and dist[0] == 7.0   # distance = 2 R + 1 (light_sun_sample_visible.rcall:22-29)


def test_q1_q2_area_light_pdf_and_direction():
    """Q1: `cross(..).length()` is the component count 3, so the triangle area is 1.5 whatever the triangle; the triangle is picked
    uniformly by index.  Q2: wi = normalize(position - point) points AWAY from the sampled point."""
    desc, o = _light_scene([make_light(abi.LIGHT_AREA, "area", resource_id=2, intensity=1.0)])
    assert o.n_rt_lights == 1                                                      # one RTLight per instance using the material
    rng = np.random.default_rng(8)
    r3 = rng.random((500, 3)).astype(np.float32)
    p = np.tile([[0.1, 0.05, -0.2]], (500, 1)).astype(np.float32)                  # inside the +-1 cube
    wi, dist, pdf, em = o.light_sample(0, p, r3)
    assert np.allclose(pdf, (1.0 / 12.0) * (1.0 / 1.5), rtol=1e-6)                 # 12 triangles; the cube's faces have area 2, not 1.5
    point = p - wi * dist[:, None]                                                 # a direction towards the light would give p + wi d
    on_cube = np.isclose(np.abs(point).max(axis=1), 1.0, atol=1e-4)
    assert on_cube.all()
    towards = p + wi * dist[:, None]
    assert not np.isclose(np.abs(towards).max(axis=1), 1.0, atol=1e-4).all()


def _sky_desc(w=48, h=24, seed=0, intensity=0.3):
    desc = cube_scene()
    rng = np.random.default_rng(seed)
    tex = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    tex[h // 3, w // 4] = (255, 255, 255, 255)
    tex[..., 3] = 255
    desc.textures[1] = (abi.TEX_RGBA_SRGB, tex, "sky")
    desc.lights = [make_light(abi.LIGHT_SKY, "sky", resource_id=1, intensity=intensity, yaw=20, pitch=75, roll=10)]
    return desc, tex


def test_sky_distribution_tables_against_an_independent_computation():
    """calculate_skymap_distributions (scene.rs:2191-2313) + Distribution1D::new (distribution.rs:14-38): per texel
    luminance(from_rgb(rgb8 / 255, illuminant)) * sin(pi (y + 0.5) / H); per row cdf_i = cdf_{i-1} + f_{i-1} / W normalised by the
    row integral; the marginal over the row integrals.  Recomputed here in float64 from the texels, with the colour conversion taken
    from the reference's own pinned host routines (Spectrum::from_rgb / luminance, KATs in test_oracle_kats.py)."""
    desc, tex = _sky_desc()
    o = OracleScene(desc)
    h, w = tex.shape[:2]
    L = pyoracle.lib()
    f = np.zeros((h, w))
    sp = np.zeros(16, np.float32)
    for y in range(h):
        for x in range(w):
            r, g, b = (tex[y, x, :3].astype(np.float32) / np.float32(255.0))
            L.orc_spectrum_from_rgb(float(r), float(g), float(b), 1, sp.ctypes.data)
            f[y, x] = L.orc_spectrum_luminance(sp.ctypes.data) * np.sin(np.pi * (y + 0.5) / h)
    row_int = f.sum(1) / w
    cdf = np.concatenate([np.zeros((h, 1)), np.cumsum(f, 1) / w], 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        cdf_n = np.where(row_int[:, None] > 0, cdf / row_int[:, None], np.arange(w + 1)[None, :] / w)
    marg_int = row_int.sum() / h
    marg_cdf = np.concatenate([[0.0], np.cumsum(row_int) / h]) / marg_int
    values, ccdf = o.sky_cond()
    sky = o.sky()
    header = sky[36:40]
    assert tuple(header[:3].view(np.uint32)) == (h + 1, 2 * h + 1, w + 1)            # light_sky_sample_visible.rcall:19-26
    assert np.isclose(header[3], marg_int, rtol=2e-5)
    m = sky[40:]
    assert np.allclose(m[:h + 1], marg_cdf, atol=3e-6) and np.allclose(m[h + 1:2 * h + 1], row_int, rtol=3e-5) and np.allclose(m[2 * h + 1:3 * h + 1], row_int, rtol=3e-5)
    assert np.allclose(values, f, rtol=3e-5, atol=1e-9) and np.allclose(ccdf, cdf_n, atol=5e-6)
    assert m[h] == 1.0 and (ccdf[:, -1] == 1.0).all() and (np.diff(ccdf, axis=1) >= 0).all()


def test_q3_sky_conditional_lookup_always_hits_texel_zero():
    """Q3 (light_sky_sample_visible.rcall:67-98): integer texel coordinates go to a normalised REPEAT / NEAREST sampler, so every cdf
    lookup returns cdf[0][0] = 0: the search ends at the last cell, u = (W - 1 + rand.x) / (W + 1) whatever the row's distribution,
    and pdf_u = value[0][0] / integral[row].  A correct lookup would spread u over [0, 1) following the texels."""
    desc, tex = _sky_desc(seed=3)
    o = OracleScene(desc)
    h, w = tex.shape[:2]
    rng = np.random.default_rng(9)
    r3 = rng.random((2000, 3)).astype(np.float32)
    wi, dist, pdf, em = o.light_sample(0, np.zeros((2000, 3), np.float32), r3, scene_radius=2.0)
    ok = pdf > 0
    assert ok.mean() > 0.9 and np.allclose(dist[ok], 5.0)
    # undo the sky rotation: obj2world = R_y(yaw) R_z(pitch) R_x(roll) is the first 16 floats of the RTSky block (column-major)
    o2w = o.sky()[:16].reshape(4, 4).T.astype(np.float64)[:3, :3]
    local = wi[ok].astype(np.float64) @ o2w                                        # inverse of a rotation = transpose
    phi = np.arctan2(local[:, 1], local[:, 0]) % (2 * np.pi)
    u = phi / (2 * np.pi)
    lo, hi = (w - 1) / (w + 1), w / (w + 1)
    assert (u > lo - 1e-4).all() and (u < hi + 1e-4).all()                         # every sample in the one cell the broken search ends in
    assert np.allclose(u, (w - 1 + r3[ok, 0]) / (w + 1), atol=2e-4)


def test_q4_sky_on_a_miss_is_not_scaled_by_the_light_intensity():
    """Q4 (path_trace.rgen:75-82 against light_sky_sample_visible.rcall:127-129): the texel a missing ray looks up is used as is; only
    sampled sky light carries `intensity`.  With no geometry every camera ray misses: the image does not depend on the intensity."""
    from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_meta
    images = []
    for intensity in (0.25, 4.0):
        _, tex = _sky_desc(seed=5)
        desc = SceneDesc(np.zeros(0, VERTEX_DTYPE), np.zeros(0, np.uint32), np.zeros(0, MESH_DTYPE), None, np.zeros(0, INSTANCE_DTYPE),
                         [make_material("m")], [make_light(abi.LIGHT_SKY, "sky", resource_id=1, intensity=intensity)],
                         [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default"), (abi.TEX_RGBA_SRGB, tex, "sky")],
                         make_camera(position=(0, 0, 0), target=(0.3, 0.2, 1.0)), make_meta((0, 0, 0), 1.0, 1.0))
        r = OracleRenderer(OracleScene(desc), 24, 16, threads=1)
        r.set_depth(3)
        r.step(3)
        images.append(r.read_hdr())
    assert images[0][..., :3].max() > 0 and np.array_equal(images[0], images[1])


# ---------------------------------------------------------------------------------------------------------------------
# RNG and host-side quirks
# ---------------------------------------------------------------------------------------------------------------------
def test_q11_seed_goes_through_a_float():
    """Q11 (path_trace.rgen:143, random.glsl:36-42): srand(vec3(seed, x, y)) converts the uint seed to float: seeds that differ below
    the 24-bit mantissa give the same stream."""
    def stream(seed, x=7, y=9, n=8):
        out = np.zeros(n, np.float32)
        pyoracle.lib().orc_rand_stream(seed, x, y, out.ctypes.data, n)
        return out
    assert np.array_equal(stream(1 << 24), stream((1 << 24) + 1))                 # 16777217 is not a float
    assert not np.array_equal(stream(1 << 24), stream((1 << 24) + 2))
    assert not np.array_equal(stream(5), stream(6))
    s = stream(123, n=4096)
    assert 0.0 <= s.min() and s.max() < 1.0 and abs(s.mean() - 0.5) < 0.02           # floats in [0, 1) from 23 mantissa bits
    assert np.all(np.modf(s.astype(np.float64) * (1 << 23))[0] == 0)


def test_q10_device_colour_upsampling_is_unclamped_and_host_is_clamped():
    """Q10 (spectrum.rs:135-137 against spectrum.glsl:202-284): Spectrum::from_rgb clamps to [0, 1], the shader versions do not."""
    L = pyoracle.lib()
    host1, host2, dev1, dev2 = (np.zeros(16, np.float32) for _ in range(4))
    L.orc_spectrum_from_rgb(1.0, 1.0, 1.0, 0, host1.ctypes.data)
    L.orc_spectrum_from_rgb(2.0, 2.0, 2.0, 0, host2.ctypes.data)
    L.orc_dev_from_surface_color(np.array([1, 1, 1], np.float32).ctypes.data, dev1.ctypes.data)
    L.orc_dev_from_surface_color(np.array([2, 2, 2], np.float32).ctypes.data, dev2.ctypes.data)
    assert host1.max() < 1.0 and (host2 == 1.0).all()                              # the host clamps every bin to [0, 1]
    assert np.allclose(dev2, 2 * dev1, rtol=1e-6) and dev2.min() > 1.9            # the device does not


def test_q12_a_jitter_offset_is_consumed_by_every_launch():
    """Q12 (raytracer.rs:489): WorkScheduler::next() runs once per launch although only fresh paths use the offset: launch i of a
    restart gets the i-th element of the hierarchical midpoint sequence (raytracer.rs:1168-1206), not the (i / depth)-th."""
    offs = [pyoracle.launch_constants(0, i)[1] for i in range(10)]
    assert offs[:5] == [(0.5, 0.5), (0.25, 0.75), (0.75, 0.25), (0.75, 0.75), (0.25, 0.25)]
    assert len(set(offs)) == 10


# ---------------------------------------------------------------------------------------------------------------------
# Frosted glass and the Uber material: sample / value agreement lobe by lobe, Oren-Nayar against the formula
# ---------------------------------------------------------------------------------------------------------------------
def test_frosted_sample_matches_value_in_both_branches():
    """mat_frosted_{value,sample_value}.rcall: rand.z < 0.5 reflects off a sampled microfacet, else refracts through it; the value
    callable recognises the branch from the hemispheres of wo and wi and must return the same value and pdf."""
    desc, o = scene_with(abi.MAT_FROSTED, roughness_mul=0.4, anisotropy=0.0, ior=1.5)
    rng = np.random.default_rng(12)
    wo = hemisphere_dirs(6000, rng, min_cos=0.2)
    r3 = rng.random((6000, 3)).astype(np.float32)
    wi, val_s, pdf_s = o.bsdf_sample(CUBE_MAT, wo, r3)
    ok = pdf_s > 0
    same = wi[:, 2] * wo[:, 2] > 0
    # the value callable tells the branch from the hemispheres (reflection: same side, transmission: opposite sides); a rough lobe
    # also reflects below the horizon and the shader leaves those samples their pdf (no hemisphere test in the reflect branch)
    reflect = ok & (r3[:, 2] < 0.5) & same
    refract = ok & (r3[:, 2] >= 0.5) & ~same
    assert reflect.sum() > 1500 and refract.sum() > 1000
    # reflection: the sampling callable halves its pdf for the coin flip between the branches (mat_frosted_sample_value.rcall:47); the
    # value callable returns the branch's own density (mat_frosted_value.rcall:46): a factor of two, kept
    val_e, pdf_e = o.bsdf_value(CUBE_MAT, wo[reflect], wi[reflect])
    close = np.isclose(pdf_e, 2.0 * pdf_s[reflect], rtol=5e-3) & np.isclose(val_e[:, 0], val_s[reflect][:, 0], rtol=2e-2, atol=1e-6)
    assert close.mean() > 0.99                                                     # wh is re-derived from (wo, wi): a few grazing cases drift
    # transmission ("Q17", the microfacet sibling of Q7): refract(wo, wh, eta) is called with the OUTWARD wo (:56), so the sampled
    # direction is not the refraction of wo about wh, and the half vector the value callable reconstructs from (wo, wi)
    # (mat_frosted_value.rcall:50-51) is not the sampled one: the two callables disagree by orders of magnitude.  A shader
    # with the incident direction negated would make them agree; the oracle states the shader.
    val_t, pdf_t = o.bsdf_value(CUBE_MAT, wo[refract], wi[refract])
    ratio = pdf_t / (2.0 * pdf_s[refract])
    assert np.median(np.abs(np.log10(ratio))) > 1.0
    assert np.allclose(val_s[ok], val_s[ok][:, :1])                               # dielectric: spectrally flat
    assert np.allclose(np.linalg.norm(wi[ok], axis=1), 1.0, atol=1e-5)


def test_uber_lobes_and_oren_nayar():
    """mat_uber_{value,sample_value}.rcall: rand < 0.5 picks the GGX specular lobe (pdf x 0.5), else the Oren-Nayar diffuse lobe with
    sigma = roughness / 2 (cosine sampling, pdf = 0.5 |cos| / pi).  The diffuse value against an independent statement of the formula."""
    rough_mul = 0.8
    desc, o = scene_with(abi.MAT_UBER, roughness_mul=rough_mul, metalness_mul=0.3, anisotropy=0.0, diffuse_mul=(200, 150, 100))
    rng = np.random.default_rng(13)
    wo = hemisphere_dirs(5000, rng, min_cos=0.2)
    r3 = rng.random((5000, 3)).astype(np.float32)
    wi, val_s, pdf_s = o.bsdf_sample(CUBE_MAT, wo, r3)
    diffuse = (r3[:, 2] >= 0.5) & (pdf_s > 0)
    spec = (r3[:, 2] < 0.5) & (pdf_s > 0) & (wi[:, 2] * wo[:, 2] > 0)
    assert diffuse.sum() > 2000 and spec.sum() > 1000
    assert np.allclose(pdf_s[diffuse], 0.5 * np.abs(wi[diffuse][:, 2]) / np.pi, rtol=3e-6)
    # the value callable with the matching lobe choice (its own scalar rand)
    vd, pd = o.bsdf_value(CUBE_MAT, wo[diffuse], wi[diffuse], rand=np.full(diffuse.sum(), 0.75, np.float32))
    assert np.allclose(pd, pdf_s[diffuse], rtol=5e-6) and np.allclose(vd, val_s[diffuse], rtol=2e-5, atol=1e-8)
    vs, ps = o.bsdf_value(CUBE_MAT, wo[spec], wi[spec], rand=np.full(spec.sum(), 0.25, np.float32))
    assert np.isclose(ps, pdf_s[spec], rtol=5e-3).mean() > 0.995 and np.isclose(vs[:, 5], val_s[spec][:, 5], rtol=2e-2, atol=1e-7).mean() > 0.99
    # Oren-Nayar: value = S(albedo * (A + B max(0, cos dphi) sin(alpha) tan(beta)) / pi); S is linear in a common scale factor
    a = wo[diffuse].astype(np.float64); b = wi[diffuse].astype(np.float64)
    sigma = (1.0 * rough_mul) / 2.0                                               # roughness = texture (1.0) x multiplier; sigma = roughness / 2
    A = 1 - sigma ** 2 / (2 * (sigma ** 2 + 0.33)); B = 0.45 * sigma ** 2 / (sigma ** 2 + 0.09)
    sin_o, sin_i = np.sqrt(np.maximum(0, 1 - a[:, 2] ** 2)), np.sqrt(np.maximum(0, 1 - b[:, 2] ** 2))
    with np.errstate(invalid="ignore", divide="ignore"):
        cosd = np.maximum(0.0, (b[:, 0] / sin_i) * (a[:, 0] / sin_o) + (b[:, 1] / sin_i) * (a[:, 1] / sin_o))
    co, ci = np.abs(a[:, 2]), np.abs(b[:, 2])
    # "Q16" (mat_uber_value.rcall:70-72): step(|wo.z|, |wi.z|) is 1 when wi is the steeper direction, and mix() then takes
    # sin(theta_i) and tan(theta_o): the sine of the SMALLER angle and the tangent of the LARGER one -- Oren-Nayar's alpha and beta
    # are the other way round.  The oracle states the shader.
    steeper_i = ci >= co
    as_written = (A + B * cosd * np.where(steeper_i, sin_i, sin_o) * np.where(steeper_i, sin_o / co, sin_i / ci)) / np.pi
    textbook = (A + B * cosd * np.where(steeper_i, sin_o, sin_i) * np.where(steeper_i, sin_i / ci, sin_o / co)) / np.pi
    ref = np.zeros(16, np.float32)
    pyoracle.lib().orc_dev_from_surface_color(np.array([200 / 255, 150 / 255, 100 / 255], np.float32).ctypes.data, ref.ctypes.data)
    good = np.isfinite(as_written)
    assert np.allclose(vd[good], ref[None, :] * as_written[good, None], rtol=3e-4, atol=1e-8)
    assert np.abs(as_written[good] - textbook[good]).max() > 0.02                  # a corrected shader would fail the line above


def test_spectrum_to_rgb_round_trip_and_luminance():
    """spectrum.glsl:39-86: XYZ by the CIE bin weights x 0.17557178, linear sRGB by the 3 x 3; a grey surface colour upsampled to a
    spectrum (flat white basis) comes back grey, and the device luminance is the Y row."""
    L = pyoracle.lib()
    sp, rgb = np.zeros(16, np.float32), np.zeros(3, np.float32)
    # a grey surface upsamples to a flat spectrum (x 0.94 x 1.062), i.e. illuminant E, whose linear-sRGB (D65) coordinates are
    # M . (1, 1, 1) = (1.2048, 0.9484, 0.9087) with the XYZ -> sRGB matrix of spectrum.glsl:74-81 -- not grey: the shader has no
    # chromatic adaptation.  Checks the bin weights' normalisation and the matrix against the published sRGB one.
    M = np.array([[3.2406, -1.5372, -0.4986], [-0.9689, 1.8758, 0.0415], [0.0557, -0.2040, 1.0570]])
    e_white = M @ np.ones(3)
    for g in (0.1, 0.5, 1.0):
        L.orc_dev_from_surface_color(np.array([g, g, g], np.float32).ctypes.data, sp.ctypes.data)
        L.orc_dev_rgb(sp.ctypes.data, rgb.ctypes.data)
        assert np.allclose(rgb / (0.94 * 1.0617 * g), e_white, rtol=0.02)
        assert abs(L.orc_dev_luminance(sp.ctypes.data) / (0.94 * 1.0617 * g) - 1.0) < 0.01     # Y of the flat spectrum = its level
    # primaries keep their hue: the largest channel of the round trip is the one that went in
    for i, c in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
        L.orc_dev_from_surface_color(np.array(c, np.float32).ctypes.data, sp.ctypes.data)
        L.orc_dev_rgb(sp.ctypes.data, rgb.ctypes.data)
        assert int(np.argmax(rgb)) == i and rgb[i] > 0.6
