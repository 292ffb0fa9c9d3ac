"""Pins the oracle against every golden vector / known-answer test the reference's own tests hold for
this path.  None of them pins a ray hit, a BSDF value or a pixel ("render parity
unpinned"); what they do pin is the CPU-side colour math, the camera helper, the pixel-jitter
schedule and (through resources/mattest.glaze, see test_parser.py) the byte format.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle


@pytest.fixture(scope="module")
def L():
    return pyoracle.lib()


def _sp(L, r, g, b, light=False):
    out = np.zeros(16, np.float32)
    L.orc_spectrum_from_rgb(r, g, b, int(light), out.ctypes.data)
    return out


def _xyz(L, sp):
    out = np.zeros(3, np.float32)
    L.orc_spectrum_to_xyz(sp.ctypes.data, out.ctypes.data)
    return out


def _rgb_of_xyz(L, xyz):
    out = np.zeros(3, np.float32)
    xyz = np.asarray(xyz, np.float32)
    L.orc_xyz_to_rgb(xyz.ctypes.data, out.ctypes.data)
    return out


# ---- lib/src/geometry/spectrum.rs:734-794 ---------------------------------------------------------
def test_spectrum_to_xyz_kat(L):
    """spectrum.rs:762-770: from_rgb(1,0,1,false).to_xyz() = (63.915726, 31.252344, 86.988340) +-1e-5"""
    xyz = _xyz(L, _sp(L, 1.0, 0.0, 1.0))
    assert abs(xyz[0] - 63.915726) < 2e-5 and abs(xyz[1] - 31.252344) < 2e-5 and abs(xyz[2] - 86.988340) < 2e-5


def test_spectrum_luminance_kat(L):
    """spectrum.rs:772-777: luminance = 0.31252345 +-1e-5"""
    assert abs(L.orc_spectrum_luminance(_sp(L, 1.0, 0.0, 1.0).ctypes.data) - 0.31252345) < 1e-5


def test_spectrum_constructors(L):
    """spectrum.rs:734-760: black < 0.05, white and from_rgb(1,1,1) > 0.95 after to_xyz().to_rgb()"""
    assert (_rgb_of_xyz(L, _xyz(L, np.zeros(16, np.float32))) < 0.05).all()
    white = np.zeros(16, np.float32)
    L.orc_spectrum_white(white.ctypes.data)
    assert (_rgb_of_xyz(L, _xyz(L, white)) > 0.95).all()
    assert (_rgb_of_xyz(L, _xyz(L, _sp(L, 1.0, 1.0, 1.0))) > 0.95).all()


def test_spectrum_blackbody_kats(L):
    """spectrum.rs:779-805: blackbody(0) is black; blackbody(800) -> r = 0.4153538, g = b = 0; blackbody(6500) is white-ish"""
    sp = np.ones(16, np.float32)
    L.orc_spectrum_from_blackbody(0.0, sp.ctypes.data)
    assert not sp.any()
    L.orc_spectrum_from_blackbody(800.0, sp.ctypes.data)
    rgb = _rgb_of_xyz(L, _xyz(L, sp))
    assert abs(rgb[0] - 0.4153538) < 1e-5 and rgb[1] == 0.0 and rgb[2] == 0.0
    L.orc_spectrum_from_blackbody(6500.0, sp.ctypes.data)
    assert _rgb_of_xyz(L, _xyz(L, sp))[0] > 0.9


# ---- lib/src/geometry/color.rs:329-345 ------------------------------------------------------------
def test_xyz_rgb_kats(L):
    rgb = _rgb_of_xyz(L, (23.954, 19.020, 13.234))
    assert np.allclose(rgb, (0.67843, 0.39608, 0.37255), atol=1e-5)
    xyz = np.zeros(3, np.float32)
    src = np.array((0.67843, 0.39608, 0.37255), np.float32)
    L.orc_rgb_to_xyz(src.ctypes.data, xyz.ctypes.data)
    assert np.allclose(xyz, (23.954, 19.020, 13.234), atol=1e-3)


# ---- lib/src/geometry/camera.rs:296-307 -----------------------------------------------------------
def test_fovx_to_fovy_kat(L):
    fovy = L.orc_fovy(np.float32(np.radians(91.0)), np.float32(1.453))
    assert abs(np.degrees(fovy) - 70.0) < 1e-3 * 180 / np.pi + 1e-3


# ---- lib/src/vulkan/raytracer.rs:1168-1206 (sequence listed in SURVEY 8 a8) ------------------------
def test_work_scheduler_sequence():
    expected = [(.5, .5), (.25, .75), (.75, .25), (.75, .75), (.25, .25), (.125, .375), (.375, .125), (.375, .375),
                (.125, .125), (.625, .875)]
    got = [pyoracle.launch_constants(0, i)[1] for i in range(len(expected))]
    assert got == expected
    offs = {pyoracle.launch_constants(0, i)[1] for i in range(85)}
    assert len(offs) == 85 and all(0 < x < 1 and 0 < y < 1 for x, y in offs)      # 1 + 4 + 16 + 64 distinct midpoints


# ---- lib/src/shaders/random.glsl --------------------------------------------------------------------
def test_pcg_hash_and_float_construction(L):
    # PCG-RXS-M-XS-32 of random.glsl:7-12, restated in python integers
    def pcg(x):
        state = (x * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        return ((word >> 22) ^ word) & 0xFFFFFFFF
    for x in (0, 1, 2, 12345, 0xDEADBEEF, 0xFFFFFFFF):
        assert L.orc_pcg_hash(x) == pcg(x)
    out = np.zeros(64, np.float32)
    L.orc_rand_stream(123456789, 17, 33, out.ctypes.data, 64)
    assert (out >= 0).all() and (out < 1).all() and len(set(out.tolist())) > 60
    # seed goes through float(uint) (Q11): seeds that round to the same float give the same stream
    out2 = np.zeros(64, np.float32)
    L.orc_rand_stream(123456791, 17, 33, out2.ctypes.data, 64)     # 123456789 and ..91 both round to 123456792.0f
    assert np.array_equal(out, out2)


# ---- device-flavour colour math consistency ---------------------------------------------------------
def test_device_spectrum_matches_host_tables_closely(L):
    """spectrum.glsl carries 7-digit copies of the spectrum.rs tables: the two must agree to ~1e-6 (unclamped inputs in range)."""
    rng = np.random.default_rng(0)
    for _ in range(50):
        rgb = rng.random(3).astype(np.float32)
        dev = np.zeros(16, np.float32)
        L.orc_dev_from_surface_color(rgb.ctypes.data, dev.ctypes.data)
        host = _sp(L, *rgb.tolist())
        assert np.allclose(np.clip(dev, 0, 1), host, atol=2e-6)
        dev_l = np.zeros(16, np.float32)
        L.orc_dev_from_illuminant_color(rgb.ctypes.data, dev_l.ctypes.data)
        assert np.allclose(np.clip(dev_l, 0, 1), _sp(L, *rgb.tolist(), light=True), atol=2e-6)
        assert abs(L.orc_dev_luminance(np.clip(dev, 0, 1).astype(np.float32).ctypes.data) - L.orc_spectrum_luminance(host.ctypes.data)) < 1e-5
    white = np.ones(16, np.float32)
    out = np.zeros(3, np.float32)
    L.orc_dev_rgb(white.ctypes.data, out.ctypes.data)
    # the Y bins times 0.17557178 integrate to 1 for an equal-energy spectrum (spectrum.glsl:39-48); rgb() is linear
    assert abs(L.orc_dev_luminance(white.ctypes.data) - 1.0) < 1e-5
    half = (white * 0.5).astype(np.float32)
    out_half = np.zeros(3, np.float32)
    L.orc_dev_rgb(half.ctypes.data, out_half.ctypes.data)
    assert np.allclose(out_half * 2, out, rtol=1e-6)


# ---- include/glz_detmath.h --------------------------------------------------------------------------
def _ulp_err(a, ref):
    ref32 = ref.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(a.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)


def test_detmath_accuracy():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-2 * np.pi, 4 * np.pi, 200000), np.linspace(0, 2 * np.pi, 1001)]).astype(np.float32)
    x64 = x.astype(np.float64)
    assert np.abs(pyoracle.detmath("sin", x).astype(np.float64) - np.sin(x64)).max() < 2e-7
    assert np.abs(pyoracle.detmath("cos", x).astype(np.float64) - np.cos(x64)).max() < 2e-7
    c = np.concatenate([rng.uniform(-1, 1, 200000), [-1, -0.5, 0, 0.5, 1, 1.5, -1.5]]).astype(np.float32)
    assert np.abs(pyoracle.detmath("acos", c).astype(np.float64) - np.arccos(np.clip(c.astype(np.float64), -1, 1))).max() < 5e-7
    yy = rng.normal(size=200000).astype(np.float32)
    xx = rng.normal(size=200000).astype(np.float32)
    assert np.abs(pyoracle.detmath("atan2", xx, yy).astype(np.float64) - np.arctan2(yy.astype(np.float64), xx.astype(np.float64))).max() < 5e-7
    # exact special values the shaders rely on
    assert pyoracle.detmath("acos", np.array([1.0], np.float32))[0] == 0.0
    assert pyoracle.detmath("atan2", np.array([1.0], np.float32), np.array([0.0], np.float32))[0] == 0.0
    assert abs(pyoracle.detmath("atan2", np.array([-1.0], np.float32), np.array([0.0], np.float32))[0] - np.pi) < 1e-6


# ---- oracle self-consistency on the reference's own cube (counts pinned by converter/src/main.rs:740-746) ----
def test_cube_scene_counts_and_render_sanity():
    from glaze_amd.scenes import cube_scene
    from oracle.pyoracle import OracleRenderer, OracleScene
    desc = cube_scene()
    assert desc.vertices.shape[0] == 24 and desc.meshes.shape[0] == 1 and desc.transforms.shape[0] == 1
    assert desc.instances.shape[0] == 1 and len(desc.materials) == 3 and len(desc.textures) == 2
    sc = OracleScene(desc)
    assert sc.lights_no == 1 and sc.n_world_triangles == 12
    r = OracleRenderer(sc, 48, 48, threads=4)
    r.set_depth(2)
    r.draw(4)
    hdr = r.read_hdr()
    assert (hdr[..., 3] == 8.0).all() and np.isfinite(hdr).all() and hdr[..., :3].mean() > 0
    # thread count must not change the result (pixels are independent)
    r1 = OracleRenderer(sc, 48, 48, threads=1)
    r1.set_depth(2)
    r1.draw(4)
    assert np.array_equal(hdr.view(np.uint32), r1.read_hdr().view(np.uint32))
    # F11: without the build-added light the reference renders black and does not even count launches
    r0 = OracleRenderer(OracleScene(cube_scene(light=False)), 16, 16, threads=1)
    r0.draw(1)
    assert not r0.read_hdr().any()


def _layout_kat_scene():
    """One triangle under the matrix of the reference's `cgmath_vktransform_memory_layout` test (geometry/mesh.rs:110-119):
    Matrix4::new takes COLUMNS, so the 16 stored floats 0,4,8,12, 1,5,9,13, ... are the row-major 3x4 Vulkan transform
    [0,1,2,3; 4,5,6,7; 8,9,10,11]: world = M * (x, y, z, 1)."""
    from glaze_amd.scene_desc import MESH_DTYPE, VERTEX_DTYPE, SceneDesc
    from glaze_amd.scenes import cube_scene
    base = cube_scene()
    verts = np.zeros(3, VERTEX_DTYPE)
    verts["vv"] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    verts["vn"] = [0, 0, 1]
    m = np.array([0.0, 4.0, 8.0, 12.0, 1.0, 5.0, 9.0, 13.0, 2.0, 6.0, 10.0, 14.0, 3.0, 7.0, 11.0, 15.0], np.float32)
    return SceneDesc(verts, np.array([0, 1, 2], np.uint32), np.array([(0, 1, 0, 3)], MESH_DTYPE), m.reshape(1, 16), base.instances, base.materials,
                     base.lights, base.textures, base.camera, base.meta)


LAYOUT_KAT_WORLD = np.array([[3, 7, 11], [3 + 0, 7 + 4, 11 + 8], [3 + 1, 7 + 5, 11 + 9]], np.float32)     # rows of [0..11] applied to (x, y, z, 1)


def test_transform_memory_layout_kat():
    from oracle.pyoracle import OracleScene
    sc = OracleScene(_layout_kat_scene())
    w = LAYOUT_KAT_WORLD
    centre = w.mean(0)
    n = np.cross(w[1] - w[0], w[2] - w[0])
    n /= np.linalg.norm(n)
    o = (centre + 2.0 * n).astype(np.float32)[None]
    t, tri, *_ = sc.trace_closest(o, (-n).astype(np.float32)[None])
    assert tri[0] == 0 and abs(float(t[0]) - 2.0) < 1e-4            # the world triangle is where the row-major reading puts it
    t, tri, *_ = sc.trace_closest(np.array([[0.25, 0.25, 1.0]], np.float32), np.array([[0, 0, -1]], np.float32))
    assert not np.isfinite(t[0])                                    # and not where the object-space triangle was
