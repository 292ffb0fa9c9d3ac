"""OBJ converter, JPEG codec and image output, CPU only.

The counts are the ones the reference's converter tests pin for resources/cube.obj (converter/src/main.rs:713-800:
1 mesh, 1 transform, 1 instance, 1 camera, 3 materials, 2 textures, 24 vertices; 10 mip levels for the 512x512
checker with --gen-mipmaps, 1 without).  cube.obj, cube.mtl and checker.jpg under tests/golden are the reference's own
fixture files.  PIL (libjpeg) is the independent JPEG implementation the codec is checked against.
"""
import io
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import glaze_amd
from glaze_amd import abi
from oracle import glaze_v1

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CUBE = os.path.join(GOLDEN, "cube.obj")
CSRC = os.path.join(os.path.dirname(GOLDEN), "..", "glaze_amd", "csrc")


def test_working_conversion(tmp_path):
    out = str(tmp_path / "cube.glaze")
    counts = glaze_amd.convert_obj(CUBE, out)
    assert counts == dict(vertices=24, triangles=12, meshes=1, materials=3, textures=2, lights=0)
    for p in (glaze_amd.parse(out), glaze_v1.parse(out)):
        assert len(p.meshes()) == 1 and len(p.transforms()) == 1 and len(p.instances()) == 1 and len(p.cameras()) == 1
        assert len(p.materials()) == 3 and len(p.textures()) == 2 and len(p.vertices()) == 24


def test_converted_cube_contents(tmp_path):
    """What SURVEY §8(d) config 2 says the converter's output for cube.obj is."""
    out = str(tmp_path / "cube.glaze")
    glaze_amd.convert_obj(CUBE, out)
    p = glaze_amd.parse(out)
    mats = p.materials()
    assert [m.name for m in mats] == [b"default", b"DefaultMaterial", b"Material"]
    assert tuple(mats[2].diffuse_mul) == (204, 204, 204) and mats[2].diffuse == 1 and mats[2].mtype == abi.MAT_LAMBERT   # (0.8*255) as u8
    assert tuple(mats[1].diffuse_mul) == (153, 153, 153) and not mats[2].has_emissive
    mesh = p.meshes()[0]
    assert mesh["material"] == 2 and mesh["indices"].size == 36 and mesh["indices"].max() == 23
    assert np.array_equal(p.transforms()[0].reshape(4, 4), np.eye(4, dtype=np.float32)) and p.instances().tolist() == [[0, 0]]
    v = p.vertices()
    assert np.array_equal(np.unique(np.abs(v[:, :3])), [1.0]) and np.allclose(np.linalg.norm(v[:, 3:6], axis=1), 1.0)
    assert len({tuple(r) for r in v.view(np.uint32).tolist()}) == 24                         # de-duplicated
    tris = v[mesh["indices"].reshape(-1, 3)]
    n = np.cross(tris[:, 1, :3] - tris[:, 0, :3], tris[:, 2, :3] - tris[:, 0, :3])
    assert (np.einsum("ij,ij->i", n, tris[:, 0, 3:6]) > 0).all()                                # winding agrees with the stored normals
    # first face of the file: `f 1/1/1 ...` = position (1,1,-1), uv (0.625, 0.5) -> v flipped, normal +Y
    assert v[0].tolist() == [1.0, 1.0, -1.0, 0.0, 1.0, 0.0, 0.625, 0.5]
    cam = p.cameras()[0]
    assert cam.type == abi.CAMERA_PERSPECTIVE and tuple(cam.position) == (0, 0, 0) and tuple(cam.target) == (0, 0, 100) and tuple(cam.up) == (0, 1, 0)
    assert cam.fovx_or_scale == np.float32(np.pi / 2) and cam.near_plane == np.float32(1e-3) and cam.far_plane == 100.0
    m = p.meta()
    assert tuple(m.scene_centre) == (0, 0, 0) and m.scene_radius == np.float32(np.sqrt(np.float32(12.0)) / 2) and m.exposure == 1.0
    tex = p.textures()
    assert (tex[0][0], tex[0][1].shape, tex[0][2]) == (abi.TEX_RGBA_SRGB, (1, 1, 4), "default") and (tex[0][1] == 255).all()
    assert (tex[1][0], tex[1][1].shape, tex[1][2], tex[1][3]) == (abi.TEX_RGBA_SRGB, (512, 512, 4), "checker.jpg", 1)
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "checker.jpg")).convert("RGBA"))
    d = np.abs(tex[1][1].astype(int) - ref.astype(int))
    assert d.max() <= 4 and d.mean() < 0.1 and (tex[1][1][..., 3] == 255).all()               # vs libjpeg: IDCT / upsampling rounding only


def test_mipmap_generation_and_skip(tmp_path):
    out = str(tmp_path / "mm.glaze")
    glaze_amd.convert_obj(CUBE, out, gen_mipmaps=True)
    assert [t[3] for t in glaze_amd.parse(out).textures()] == [1, 10]
    levels = glaze_v1.parse(out).textures()[1]["levels"]
    assert [l.shape[:2] for l in levels] == [(512 >> k, 512 >> k) for k in range(10)]
    assert abs(float(levels[9].mean()) - float(levels[0].mean())) < 2.0                          # the 1x1 level is the image mean
    glaze_amd.convert_obj(CUBE, out, gen_mipmaps=False)
    assert [t[3] for t in glaze_amd.parse(out).textures()] == [1, 1]


def test_converted_cube_renders_like_the_builtin_cube(tmp_path):
    """The converted file and glaze_amd.scenes.cube_scene() describe the same geometry (same triangles, any order)."""
    from glaze_amd.scenes import cube_scene
    out = str(tmp_path / "cube.glaze")
    glaze_amd.convert_obj(CUBE, out)
    p = glaze_amd.parse(out)
    v, idx = p.vertices(), p.meshes()[0]["indices"]
    d = cube_scene()
    dv = d.vertices.view(np.float32).reshape(-1, 8)
    a = {tuple(sorted(map(tuple, v[t, :3].tolist()))) for t in idx.reshape(-1, 3)}
    b = {tuple(sorted(map(tuple, dv[t, :3].tolist()))) for t in d.indices.reshape(-1, 3)}
    assert a == b and len(a) == 12


def test_mipmaps_nonuniform(tmp_path):
    """materials/texture.rs:329-360: checker_nu.jpg is 64 x 512 -> 10 levels, the width stays 1 once it got there."""
    shutil = __import__("shutil")
    for f in ("checker_nu.jpg",):
        shutil.copy(os.path.join(GOLDEN, f), tmp_path / f)
    _write(tmp_path / "nu.mtl", "newmtl m\nmap_Kd checker_nu.jpg\n")
    _write(tmp_path / "nu.obj", "mtllib nu.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    out = str(tmp_path / "nu.glaze")
    glaze_amd.convert_obj(str(tmp_path / "nu.obj"), out, gen_mipmaps=True)
    levels = glaze_v1.parse(out).textures()[1]["levels"]
    assert [l.shape[1] for l in levels] == [64, 32, 16, 8, 4, 2, 1, 1, 1, 1]
    assert [l.shape[0] for l in levels] == [512, 256, 128, 64, 32, 16, 8, 4, 2, 1]
    ref = np.asarray(Image.open(os.path.join(GOLDEN, "checker_nu.jpg")).convert("RGBA"))
    assert np.abs(levels[0].astype(int) - ref.astype(int)).max() <= 4          # 4:2:0 baseline JPEG, vs libjpeg


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)


def test_obj_features(tmp_path):
    """Polygons, negative indices, missing normals / uvs, several materials, emissive -> AREA light, PNG + shared textures."""
    Image.fromarray(np.full((4, 4, 3), 128, np.uint8)).save(tmp_path / "tex a.png")
    Image.fromarray(np.arange(64, dtype=np.uint8).reshape(8, 8)).save(tmp_path / "alpha.png")
    _write(tmp_path / "s.mtl", "newmtl lamp\nKd 1 0.5 0\nKe 2.0 0.5 0\n\nnewmtl wall\nKd 0.25 0.25 0.25\nmap_Kd tex a.png\nmap_d alpha.png\nnorm tex a.png\n"
                               "newmtl other\nmap_Kd tex a.png\n")
    _write(tmp_path / "s.obj", "mtllib s.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                               "usemtl wall\nf 1/1 2/2 3/3 4/4\n"            # quad with uvs, no normals
                               "usemtl lamp\nf -1 -4 -3\n"                    # negative indices, no uvs, no normals
                               "usemtl missing\nf 1 2 5\n"                    # unknown material -> DefaultMaterial
                               "usemtl wall\nf 1/1 3/3 5/2 \\\n 4/4 2/2\n")   # pentagon, continuation line
    out = str(tmp_path / "s.glaze")
    counts = glaze_amd.convert_obj(str(tmp_path / "s.obj"), out)
    assert counts["meshes"] == 3 and counts["triangles"] == 2 + 1 + 1 + 3 and counts["materials"] == 5 and counts["lights"] == 1
    p = glaze_amd.parse(out)
    mats, meshes, tex = p.materials(), p.meshes(), p.textures()
    assert [m.name for m in mats] == [b"default", b"DefaultMaterial", b"lamp", b"wall", b"other"]
    assert [m["material"] for m in meshes] == [3, 2, 1] and [m["indices"].size for m in meshes] == [15, 3, 3]   # first-use order; assimp index + 1
    assert mats[2].has_emissive and tuple(mats[2].emissive_col) == (255, 127, 0) and tuple(mats[2].diffuse_mul) == (255, 127, 0)
    assert [(t[2], t[0]) for t in tex] == [("default", 2), ("tex a.png", 2), ("tex a.png", 3), ("alpha.png", 1)]      # same file, two formats
    assert (mats[3].diffuse, mats[3].normal, mats[3].opacity, mats[4].diffuse) == (1, 2, 3, 1)                          # "other" shares texture 1
    assert tex[3][1].shape == (8, 8) and np.array_equal(tex[3][1], np.arange(64, dtype=np.uint8).reshape(8, 8))
    light = p.lights()[0]
    assert light.ltype == abi.LIGHT_AREA and light.resource_id == 2 and light.name == b"lamp" and light.intensity == 1.0
    v = p.vertices()
    lamp = v[meshes[1]["indices"]]
    assert lamp[:, 6:8].tolist() == [[0.0, 1.0], [1.0, 1.0], [1.0, 0.0]]                       # default corner uvs, v flipped
    assert np.allclose(np.abs(lamp[:, 3:6]), np.abs(np.cross(lamp[1, :3] - lamp[0, :3], lamp[2, :3] - lamp[0, :3]) /
                                                    np.linalg.norm(np.cross(lamp[1, :3] - lamp[0, :3], lamp[2, :3] - lamp[0, :3]))))
    assert p.meta().scene_radius == np.float32(np.sqrt(3.0) / 2)


def test_converter_errors(tmp_path):
    with pytest.raises(abi.GlazeError):
        glaze_amd.convert_obj(str(tmp_path / "missing.obj"), str(tmp_path / "o.glaze"))
    _write(tmp_path / "bad.obj", "v 0 0 0\nf 1 2 3\n")
    with pytest.raises(abi.GlazeError):
        glaze_amd.convert_obj(str(tmp_path / "bad.obj"), str(tmp_path / "o.glaze"))
    _write(tmp_path / "t.mtl", "newmtl m\nmap_Kd nothere.png\n")
    _write(tmp_path / "t.obj", "mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    with pytest.raises(abi.GlazeError):
        glaze_amd.convert_obj(str(tmp_path / "t.obj"), str(tmp_path / "o.glaze"))


def test_converter_cli(tmp_path):
    exe = os.path.join(CSRC, "glaze-converter")
    if not os.path.exists(exe):
        pytest.skip("glaze-converter is not built")
    out = str(tmp_path / "cli.glaze")
    r = subprocess.run([exe, CUBE, out, "--gen-mipmaps"], capture_output=True, text=True)
    assert r.returncode == 0 and "Done!" in r.stdout and glaze_amd.converted_file(out)
    r = subprocess.run([exe, "--benchmark", out], capture_output=True, text=True)
    assert r.returncode == 0 and "Total vertices: 24" in r.stdout and "Total materials: 3" in r.stdout
    assert subprocess.run([exe, CUBE], capture_output=True).returncode == 2
    assert subprocess.run([exe, str(tmp_path / "nope.obj"), out], capture_output=True).returncode == 1


# ---- JPEG decoder through the converter (texture input), against libjpeg ----------------------------------------
def _jpeg_as_texture(tmp_path, jpeg_bytes, tag):
    with open(tmp_path / ("%s.jpg" % tag), "wb") as f:
        f.write(jpeg_bytes)
    _write(tmp_path / ("%s.mtl" % tag), "newmtl m\nmap_Kd %s.jpg\nmap_d %s.jpg\n" % (tag, tag))
    _write(tmp_path / ("%s.obj" % tag), "mtllib %s.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n" % tag)
    out = str(tmp_path / ("%s.glaze" % tag))
    glaze_amd.convert_obj(str(tmp_path / ("%s.obj" % tag)), out)
    tex = glaze_amd.parse(out).textures()
    return tex[1][1], tex[2][1]          # RGBA (sRGB diffuse), gray (opacity)


@pytest.mark.parametrize("subsampling,quality,restart", [(0, 90, 0), (1, 75, 0), (2, 30, 0), (2, 85, 3), (1, 95, 1)])
def test_jpeg_decoder_matches_libjpeg(tmp_path, subsampling, quality, restart):
    rng = np.random.default_rng(subsampling * 100 + quality)
    yy, xx = np.mgrid[0:131, 0:77]
    img = np.stack([(xx * 3) % 256, (yy * 2) % 256, ((xx + yy) * 5) % 256], -1).astype(np.float64) * 0.7 + rng.integers(0, 80, (131, 77, 3))
    img = img.clip(0, 255).astype(np.uint8)
    buf = io.BytesIO()
    kw = dict(restart_marker_blocks=restart) if restart else {}
    Image.fromarray(img).save(buf, format="JPEG", quality=quality, subsampling=subsampling, **kw)
    rgba, gray = _jpeg_as_texture(tmp_path, buf.getvalue(), "t")
    ref = Image.open(io.BytesIO(buf.getvalue()))
    d = np.abs(rgba[..., :3].astype(int) - np.asarray(ref.convert("RGB")).astype(int))
    assert rgba.shape == (131, 77, 4) and d.max() <= 4 and d.mean() < 0.15
    luma = Image.open(io.BytesIO(buf.getvalue()))
    luma.draft("L", luma.size)                                       # libjpeg's own grayscale output = the Y plane, which is what ours takes
    assert gray.shape == (131, 77) and np.abs(gray.astype(int) - np.asarray(luma).astype(int)).max() <= 2


def test_jpeg_gray_and_unsupported(tmp_path):
    g = (np.add.outer(np.arange(40), np.arange(57)) * 3 % 256).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(g).save(buf, format="JPEG", quality=85)
    rgba, gray = _jpeg_as_texture(tmp_path, buf.getvalue(), "g")
    ref = np.asarray(Image.open(io.BytesIO(buf.getvalue())))
    assert np.abs(gray.astype(int) - ref.astype(int)).max() <= 2 and (rgba[..., 0] == rgba[..., 2]).all() and (rgba[..., 3] == 255).all()
    buf = io.BytesIO()
    Image.fromarray(np.dstack([g, g, g])).save(buf, format="JPEG", progressive=True)
    with pytest.raises(abi.GlazeError, match="progressive"):
        _jpeg_as_texture(tmp_path, buf.getvalue(), "p")
    with pytest.raises(abi.GlazeError):
        _jpeg_as_texture(tmp_path, buf.getvalue()[:200], "trunc")


# ---- image.save(): PNG and JPEG writers ------------------------------------------------------------------------------
def test_save_image_png_and_jpeg(tmp_path):
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:90, 0:123]
    img = np.stack([(xx * 2) % 256, (yy * 3) % 256, (xx + yy) % 256, np.full_like(xx, 255)], -1).astype(np.uint8)
    img[..., :3] = (img[..., :3] * 0.8 + rng.integers(0, 40, (90, 123, 3))).clip(0, 255)
    glaze_amd.save_image(tmp_path / "o.png", img)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "o.png")), img)                       # lossless, RGBA
    glaze_amd.save_image(tmp_path / "o.jpg", img)
    back = Image.open(tmp_path / "o.jpg")
    assert back.size == (123, 90) and back.mode == "RGB"
    ours = np.asarray(back).astype(float)
    buf = io.BytesIO()
    Image.fromarray(img[..., :3]).save(buf, format="JPEG", quality=75, subsampling=0)
    theirs = np.asarray(Image.open(io.BytesIO(buf.getvalue()))).astype(float)
    psnr = lambda a: 10 * np.log10(255 ** 2 / ((a - img[..., :3].astype(float)) ** 2).mean())
    assert psnr(ours) > psnr(theirs) - 0.25 and os.path.getsize(tmp_path / "o.jpg") < len(buf.getvalue()) * 1.05   # same tables, same quality scaling
    with pytest.raises(abi.GlazeError):
        glaze_amd.save_image(tmp_path / "o.bmp", img)
    with pytest.raises(abi.GlazeError):
        glaze_amd.save_image(tmp_path / "nodir" / "o.png", img)
