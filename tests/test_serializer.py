"""`.glaze` write side: `Serializer`, `ParsedScene::update`, the xz and PNG encoders.

Mirrors the reference's own parser tests (lib/src/parser/v1.rs:1400-2140): write_and_read_only_<chunk>,
write_and_read_everything, corrupted_offset / corrupted_<chunk>, update_reopen / update_some / update_all.
Every file the product writes is read back by BOTH readers: the product's C++ parser and the oracle's independent
Python reader (liblzma + xxhash + PIL), so a stream only our own decoder accepts cannot pass.
"""
import ctypes as C
import lzma
import os
import struct

import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from oracle import glaze_v1


# ---- generators (the reference's gen_* helpers, v1.rs:1100-1398, draw random content the same way) -------------
def gen_vertices(n, seed):
    return np.random.default_rng(seed).random((n, 8), dtype=np.float32) * 2 - 1


def gen_meshes(n, seed):
    rng = np.random.default_rng(seed)
    return [dict(id=i, material=int(rng.integers(0, 65536)), indices=rng.integers(0, 2 ** 32, size=3 * int(rng.integers(1, 100)), dtype=np.uint32))
            for i in range(n)]


def gen_cameras(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        c = abi.Camera()
        c.type = int(rng.integers(0, 2))
        c.position[:], c.target[:], c.up[:] = [rng.random(3, dtype=np.float32).tolist() for _ in range(3)]
        c.fovx_or_scale, c.near_plane, c.far_plane = [float(x) for x in rng.random(3, dtype=np.float32)]
        out.append(c)
    return out


def gen_materials(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        m = abi.Material()
        m.mtype, m.metal = int(rng.integers(0, 7)), int(rng.integers(0, 29))
        m.diffuse_mul[:] = rng.integers(0, 256, 3).tolist()
        if i % 3:
            m.has_emissive = 1
            m.emissive_col[:] = rng.integers(1, 256, 3).tolist()
        m.ior, m.roughness_mul, m.metalness_mul, m.anisotropy = [float(x) for x in rng.random(4, dtype=np.float32)]
        m.diffuse, m.roughness, m.metalness, m.normal, m.opacity = [int(x) for x in rng.integers(0, 65536, 5)]
        m.name = ("material %d é中 " % i + "x" * int(rng.integers(0, 40))).encode("utf8")
        out.append(m)
    return out


def gen_lights(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        l = abi.Light()
        l.ltype = int(rng.integers(0, 4))
        l.position[:], l.direction[:] = rng.normal(size=3).astype(np.float32).tolist(), rng.normal(size=3).astype(np.float32).tolist()
        l.resource_id = int(rng.integers(0, 2 ** 32))
        l.intensity, l.yaw_deg, l.pitch_deg, l.roll_deg = [float(x) for x in rng.random(4, dtype=np.float32) * 360]
        l.color[:] = rng.random(16, dtype=np.float32).tolist()
        l.name = ("light %d" % i).encode()
        out.append(l)
    return out


def gen_textures(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        fmt = [abi.TEX_GRAY, abi.TEX_RGBA_SRGB, abi.TEX_RGBA_NORM][i % 3]
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        smooth = (np.add.outer(np.arange(h), np.arange(w)) * 3 % 256).astype(np.uint8)     # filters other than "none" get chosen
        px = smooth if fmt == abi.TEX_GRAY else np.stack([smooth, rng.integers(0, 256, (h, w), dtype=np.uint8), smooth[::-1], smooth.T[:h, :w] if h == w else smooth], -1)
        out.append((fmt, np.ascontiguousarray(px), "texture %d ü" % i))
    return out


def gen_transforms(n, seed):
    return np.random.default_rng(seed).normal(size=(n, 16)).astype(np.float32)


def gen_instances(n, seed):
    return np.random.default_rng(seed).integers(0, 65536, (n, 2)).astype(np.uint16)


def gen_meta(seed):
    rng = np.random.default_rng(seed)
    m = abi.Meta()
    m.scene_centre[:] = rng.normal(size=3).astype(np.float32).tolist()
    m.scene_radius, m.exposure = float(np.float32(rng.random() * 100)), float(np.float32(rng.random() * 4))
    return m


def bytes_of(s):
    return bytes(memoryview(s))


def both(path):
    return glaze_amd.parse(path), glaze_v1.parse(path)


def check_cameras(p, o, cams):
    got = p.cameras()
    assert len(got) == len(cams) == len(o.cameras())
    for a, b, c in zip(got, cams, o.cameras()):
        assert a.type == b.type and bytes_of(a) == bytes_of(b)
        assert tuple(np.float32(x) for x in c["position"]) == tuple(b.position) and np.float32(c["far"]) == b.far_plane


def check_materials(p, o, mats):
    got, ref = p.materials(), o.materials()
    assert len(got) == len(mats) == len(ref)
    for a, b, c in zip(got, mats, ref):
        assert (a.mtype, a.metal, tuple(a.diffuse_mul), bool(a.has_emissive)) == (b.mtype, b.metal, tuple(b.diffuse_mul), bool(b.has_emissive))
        assert (not b.has_emissive) or tuple(a.emissive_col) == tuple(b.emissive_col)
        assert (a.ior, a.roughness_mul, a.metalness_mul, a.anisotropy) == (b.ior, b.roughness_mul, b.metalness_mul, b.anisotropy)
        assert (a.diffuse, a.roughness, a.metalness, a.normal, a.opacity) == (b.diffuse, b.roughness, b.metalness, b.normal, b.opacity)
        assert a.name == b.name and c["name"] == b.name.decode("utf8") and c["mtype"] == b.mtype and c["opacity"] == b.opacity


def check_lights(p, o, lights):
    got, ref = p.lights(), o.lights()
    assert len(got) == len(lights) == len(ref)
    for a, b, c in zip(got, lights, ref):
        assert bytes_of(a) == bytes_of(b)
        assert c["ltype"] == b.ltype and np.array_equal(c["color"], np.array(b.color[:], np.float32)) and c["name"] == b.name.decode()


def check_textures(p, o, textures):
    got, ref = p.textures(), o.textures()
    assert len(got) == len(textures) == len(ref)
    for a, b, c in zip(got, textures, ref):
        assert a[0] == b[0] == c["format"] and a[2] == b[2] == c["name"]
        assert np.array_equal(a[1], b[1]) and np.array_equal(c["levels"][0], b[1])      # PNG round trip is lossless, both decoders


def check_meta(p, o, meta):
    assert bytes_of(p.meta()) == bytes_of(meta)
    m = o.meta()
    assert tuple(np.float32(x) for x in m["scene_centre"]) == tuple(meta.scene_centre) and np.float32(m["exposure"]) == meta.exposure


# ---- write_and_read_only_<chunk> (v1.rs:1489-1662) ------------------------------------------------------------
def test_write_and_read_only_vert(tmp_path):
    v = gen_vertices(1000, 0xC2B4D5A5A9E49945)
    path = str(tmp_path / "only_vert.bin")
    glaze_amd.Serializer(path).with_vertices(v).serialize()
    p, o = both(path)
    assert np.array_equal(p.vertices().view(np.uint32), v.view(np.uint32)) and np.array_equal(o.vertices().view(np.uint32), v.view(np.uint32))
    assert len(p.meshes()) == 0 and len(p.cameras()) == 0 and len(p.textures()) == 0 and len(p.materials()) == 0 and len(p.lights()) == 0
    assert p.transforms().shape[0] == 0 and p.instances().shape[0] == 0


def test_write_and_read_only_meshes(tmp_path):
    meshes = gen_meshes(128, 0x2FD1D7B5A9D4E5E7)
    path = str(tmp_path / "only_meshes.bin")
    glaze_amd.Serializer(path).with_meshes(meshes).serialize()
    p, o = both(path)
    for got in (p.meshes(), o.meshes()):
        assert len(got) == len(meshes)
        for a, b in zip(got, meshes):
            assert a["id"] == b["id"] and a["material"] == b["material"] and np.array_equal(a["indices"], b["indices"])
    assert p.vertices().shape[0] == 0


def test_write_and_read_only_cameras(tmp_path):
    cams = gen_cameras(32, 0xCC6AD9820F396116)
    path = str(tmp_path / "only_cameras.bin")
    glaze_amd.Serializer(path).with_cameras(cams).serialize()
    check_cameras(*both(path), cams)


def test_write_and_read_only_textures(tmp_path):
    textures = gen_textures(7, 0x50DFC0EA9BF6E9BE)
    path = str(tmp_path / "only_textures.bin")
    glaze_amd.Serializer(path).with_textures(textures).serialize()
    check_textures(*both(path), textures)


def test_write_and_read_only_materials(tmp_path):
    mats = gen_materials(100, 0xE1BBF0A4B7C6B7B5)
    path = str(tmp_path / "only_materials.bin")
    glaze_amd.Serializer(path).with_materials(mats).serialize()
    check_materials(*both(path), mats)


def test_write_and_read_only_transforms(tmp_path):
    t = gen_transforms(64, 0x5D1D4E5E3A6D1B6D)
    path = str(tmp_path / "only_transforms.bin")
    glaze_amd.Serializer(path).with_transforms(t).serialize()
    p, o = both(path)
    assert np.array_equal(p.transforms().view(np.uint32), t.view(np.uint32)) and np.array_equal(o.transforms().view(np.uint32), t.view(np.uint32))


def test_write_and_read_only_instances(tmp_path):
    i = gen_instances(500, 0x9D0A5B3C1A8E7F11)
    path = str(tmp_path / "only_instances.bin")
    glaze_amd.Serializer(path).with_instances(i).serialize()
    p, o = both(path)
    assert np.array_equal(p.instances(), i) and np.array_equal(o.instances(), i)


def test_write_and_read_only_lights(tmp_path):
    lights = gen_lights(50, 0x1B0E8D6F3A1C9B2D)
    path = str(tmp_path / "only_lights.bin")
    glaze_amd.Serializer(path).with_lights(lights).serialize()
    check_lights(*both(path), lights)


def test_write_and_read_only_meta(tmp_path):
    meta = gen_meta(0x7A1C3E5B9D2F4A6C)
    path = str(tmp_path / "only_meta.bin")
    glaze_amd.Serializer(path).with_metadata(meta).serialize()
    check_meta(*both(path), meta)


def test_write_empty_scene(tmp_path):
    """Serializer without any with_*(): header + an offsets table of zero chunks."""
    path = str(tmp_path / "empty.bin")
    glaze_amd.Serializer(path).serialize()
    assert os.path.getsize(path) == 16 + 8 + 1
    p, o = both(path)
    assert p.vertices().shape[0] == 0 and len(p.lights()) == 0 and abi.lib().glz_parsed_meta(p._h, C.byref(abi.Meta())) == 1
    assert glaze_amd.converted_file(path)


def _everything(seed):
    return dict(vertices=gen_vertices(1000, seed), meshes=gen_meshes(100, seed + 1), cameras=gen_cameras(25, seed + 2),
                textures=gen_textures(4, seed + 3), materials=gen_materials(25, seed + 4), transforms=gen_transforms(25, seed + 5),
                instances=gen_instances(100, seed + 6), lights=gen_lights(50, seed + 7), meta=gen_meta(seed + 8))


def _write_everything(path, s):
    (glaze_amd.Serializer(path).with_vertices(s["vertices"]).with_meshes(s["meshes"]).with_instances(s["instances"])
     .with_transforms(s["transforms"]).with_textures(s["textures"]).with_materials(s["materials"]).with_lights(s["lights"])
     .with_cameras(s["cameras"]).with_metadata(s["meta"]).serialize())


def _check_everything(path, s):
    p, o = both(path)
    assert np.array_equal(p.vertices().view(np.uint32), s["vertices"].view(np.uint32))
    assert np.array_equal(o.vertices().view(np.uint32), s["vertices"].view(np.uint32))
    for got in (p.meshes(), o.meshes()):
        assert len(got) == len(s["meshes"])
        for a, b in zip(got, s["meshes"]):
            assert a["id"] == b["id"] and a["material"] == b["material"] and np.array_equal(a["indices"], b["indices"])
    assert np.array_equal(p.transforms().view(np.uint32), s["transforms"].view(np.uint32)) and np.array_equal(p.instances(), s["instances"])
    assert np.array_equal(o.transforms().view(np.uint32), s["transforms"].view(np.uint32)) and np.array_equal(o.instances(), s["instances"])
    check_cameras(p, o, s["cameras"])
    check_textures(p, o, s["textures"])
    check_materials(p, o, s["materials"])
    check_lights(p, o, s["lights"])
    check_meta(p, o, s["meta"])


def test_write_and_read_everything(tmp_path):
    s = _everything(0x4D595DF4D0F33173)
    path = str(tmp_path / "everything.bin")
    _write_everything(path, s)
    _check_everything(path, s)


# ---- corruption (v1.rs:1750-1974): the writer's hashes must make every flipped byte visible ---------------------
def _flip(path, offset, payload=b"\xFF\xFF\xFF\xFF"):
    with open(path, "r+b") as f:
        f.seek(offset)
        old = f.read(len(payload))
        f.seek(offset)
        f.write(bytes(a ^ 0xFF for a in old) if old == payload else payload)


def test_corrupted_offset(tmp_path):
    path = str(tmp_path / "corrupted_off.bin")
    glaze_amd.Serializer(path).with_vertices(gen_vertices(100, 0x8794A1E593281F2F)).serialize()
    glaze_amd.parse(path).close()
    _flip(path, 16 + 8 + 10)
    with pytest.raises(abi.GlazeError):
        glaze_amd.parse(path)
    with pytest.raises(IOError):
        glaze_v1.parse(path)


@pytest.mark.parametrize("chunk", ["vertices", "meshes", "cameras", "textures", "materials", "transforms", "instances", "lights"])
def test_corrupted_chunk(tmp_path, chunk):
    s = _everything(0x62A9F273AF56253C)
    path = str(tmp_path / "corrupted.bin")
    ser = glaze_amd.Serializer(path)
    getattr(ser, "with_" + chunk)(s[chunk]).serialize()
    p = glaze_amd.parse(path)
    getattr(p, chunk)()
    p.close()
    _flip(path, os.path.getsize(path) - 40)            # inside the only chunk of the file
    p = glaze_amd.parse(path)                          # the offsets table is intact ...
    with pytest.raises(abi.GlazeError):
        getattr(p, chunk)()                            # ... the chunk hash is not


# ---- the encoders on their own ------------------------------------------------------------------------------------
def _vertex_chunk_xz(path):
    data = open(path, "rb").read()
    n = data[24]
    for i in range(n):
        cid, off, ln = struct.unpack_from("<BQQ", data, 25 + 17 * i)
        if cid == 0:
            return data[off + 8: off + ln]
    raise AssertionError("no vertex chunk")


@pytest.mark.parametrize("kind", ["zeros", "ramp", "random", "repeats", "long"])
def test_compress_decompress(tmp_path, kind):
    """compress_decompress (v1.rs:1976-1984) through the file: liblzma must decode what xz_enc.cpp wrote."""
    rng = np.random.default_rng(5)
    n = {"long": 300_000}.get(kind, 20_000)
    if kind == "zeros":
        v = np.zeros((n, 8), np.float32)
    elif kind == "ramp":
        v = np.arange(n * 8, dtype=np.float32).reshape(n, 8)
    elif kind == "random":
        v = rng.integers(0, 2 ** 32, (n, 8), dtype=np.uint32).view(np.float32)          # incompressible: includes NaN bit patterns
    elif kind == "repeats":
        v = np.tile(rng.random((37, 8), dtype=np.float32), (n // 37 + 1, 1))[:n]
    else:
        v = np.cumsum(rng.normal(size=(n, 8)), axis=0).astype(np.float32)              # > 2 MiB: several LZMA2 chunks
    path = str(tmp_path / "xz.bin")
    glaze_amd.Serializer(path).with_vertices(v).serialize()
    xz = _vertex_chunk_xz(path)
    assert xz[:6] == b"\xfd7zXZ\x00" and xz[-2:] == b"YZ"
    assert lzma.decompress(xz, format=lzma.FORMAT_XZ) == v.tobytes()
    assert np.array_equal(glaze_amd.parse(path).vertices().view(np.uint32), v.view(np.uint32))
    if kind in ("zeros", "ramp", "repeats"):
        assert len(xz) < v.nbytes // 20                                               # it does compress
    if kind == "random":
        assert len(xz) < v.nbytes * 1.03                                              # and does not blow up


def test_png_encoder_variants(tmp_path):
    """1x1, single row / column, odd sizes, gray and RGBA, requested mip chains: PIL must decode every level."""
    rng = np.random.default_rng(9)
    textures = [(abi.TEX_GRAY, np.array([[7]], np.uint8), "1x1"),
                (abi.TEX_RGBA_SRGB, rng.integers(0, 256, (1, 33, 4), dtype=np.uint8), "row"),
                (abi.TEX_RGBA_NORM, rng.integers(0, 256, (29, 1, 4), dtype=np.uint8), "col"),
                (abi.TEX_GRAY, rng.integers(0, 256, (64, 48), dtype=np.uint8), "mips", 4),
                (abi.TEX_RGBA_SRGB, np.full((16, 16, 4), 200, np.uint8), "flat-mips", 99)]
    path = str(tmp_path / "png.bin")
    glaze_amd.Serializer(path).with_textures(textures).serialize()
    p, o = both(path)
    got, ref = p.textures(), o.textures()
    for a, b, c in zip(got, textures, ref):
        assert np.array_equal(a[1], b[1]) and np.array_equal(c["levels"][0], b[1])
    assert [t[3] for t in got] == [1, 1, 1, 4, 5]                                      # 16x16 has 5 levels down to 1x1
    lv = ref[3]["levels"]
    assert [l.shape for l in lv] == [(64, 48), (32, 24), (16, 12), (8, 6)]
    # every level is the previous one halved with the Catmull-Rom filter (Texture::gen_mipmaps, texture.rs:256-277, through
    # image::imageops::resize); PIL's BICUBIC is the same kernel (a = -0.5) with 8-bit fixed-point coefficients
    from PIL import Image
    prev = textures[3][1]
    for level in lv[1:]:
        want = np.asarray(Image.fromarray(prev).resize((prev.shape[1] // 2, prev.shape[0] // 2), Image.BICUBIC))
        assert np.abs(level.astype(int) - want.astype(int)).max() <= 2 and np.abs(level.astype(float) - want).mean() < 0.3
        prev = level
    assert all((l == 200).all() for l in ref[4]["levels"])                            # weights are normalised: a flat image stays flat
    # ringing is clamped, a hard edge overshoots neither way
    edge = np.zeros((32, 32, 4), np.uint8); edge[:, 16:] = 255
    path2 = str(tmp_path / "edge.bin")
    glaze_amd.Serializer(path2).with_textures([textures[0], (abi.TEX_RGBA_SRGB, edge, "edge", 6)]).serialize()
    lv = both(path2)[1].textures()[1]["levels"]
    assert [l.shape[:2] for l in lv] == [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    assert (lv[1][:, :7] == 0).all() and (lv[1][:, 9:] == 255).all() and 120 <= int(lv[5][0, 0, 0]) <= 135


# ---- ParsedScene::update (v1.rs:1986-2140) -------------------------------------------------------------------------
def test_update_reopen(tmp_path):
    v = gen_vertices(100, 0xBD4D59BF04981A1A)
    path = str(tmp_path / "update_reopen.bin")
    glaze_amd.Serializer(path).with_vertices(v).serialize()
    p = glaze_amd.parse(path)
    assert p.vertices().shape[0] == 100
    p.update()
    assert p.vertices().shape[0] == 100
    p.close()
    p, o = both(path)
    assert np.array_equal(p.vertices(), v) and np.array_equal(o.vertices(), v)


def test_update_some(tmp_path):
    v = gen_vertices(100, 0xBD4D59BF04981A1A)
    path = str(tmp_path / "update_some.bin")
    glaze_amd.Serializer(path).with_vertices(v).serialize()
    p = glaze_amd.parse(path)
    cams, mats, lights = gen_cameras(100, 0xECD7D80A8A4C4C95), gen_materials(100, 0xAA9475DE05B6CE41), gen_lights(100, 0xEF2F6EF8FD11E92E)
    textures, meta = gen_textures(4, 0xFFB764694A84BEDA), gen_meta(0x3A77182EE1A0747E)
    p.update(cameras=cams, materials=mats, lights=lights, textures=textures, meta=meta)
    assert np.array_equal(p.vertices(), v)
    o = glaze_v1.parse(path)
    check_cameras(p, o, cams)
    check_materials(p, o, mats)
    check_lights(p, o, lights)
    check_textures(p, o, textures)
    check_meta(p, o, meta)


def test_update_all(tmp_path):
    s = _everything(0x7CE285088B15CD6C)
    path = str(tmp_path / "update_all.bin")
    _write_everything(path, s)
    p = glaze_amd.parse(path)
    stored = {name: glaze_v1.parse(path)._raw(name) for name in ("vertex", "mesh", "transform", "instance")}
    new = dict(s, cameras=gen_cameras(100, 0x056F0B996A248BC4), materials=gen_materials(100, 0x3ABE1A9BEB00DA7B),
               lights=gen_lights(100, 0x5871F342932A7B6A), textures=gen_textures(4, 0x05E96CDC62E9A586), meta=gen_meta(0xF4AF4CA42889AAD0))
    p.update(cameras=new["cameras"], materials=new["materials"], lights=new["lights"], textures=new["textures"], meta=new["meta"])
    p.close()
    _check_everything(path, new)
    o = glaze_v1.parse(path)
    for name, raw in stored.items():
        assert o._raw(name) == raw                       # kept chunks are copied byte for byte, not re-encoded


def test_update_keeps_and_empties_chunks(tmp_path):
    """None keeps a chunk; an empty list removes it (Some(&[]) encodes to an empty chunk, which set_offset skips)."""
    s = _everything(77)
    path = str(tmp_path / "update_partial.bin")
    _write_everything(path, s)
    p = glaze_amd.parse(path)
    p.update(lights=[], meta=None)
    assert len(p.lights()) == 0 and len(p.cameras()) == len(s["cameras"]) and bytes_of(p.meta()) == bytes_of(s["meta"])
    o = glaze_v1.parse(path)
    assert "light" not in o.chunks and "camera" in o.chunks
    check_textures(p, o, s["textures"])


def test_mattest_survives_a_rewrite(tmp_path):
    """The reference fixture, rewritten by the product (update with nothing replaced, then a full re-serialisation),
    still reads as SURVEY F10 says through both readers."""
    import shutil
    from conftest import MATTEST
    path = str(tmp_path / "mattest_copy.glaze")
    shutil.copy(MATTEST, path)
    p = glaze_amd.parse(path)
    p.update()
    assert p.vertices().shape[0] == 70876 and sum(m["indices"].size for m in p.meshes()) == 3 * 138480 and len(p.textures()) == 3
    path2 = str(tmp_path / "mattest_rewritten.glaze")
    (glaze_amd.Serializer(path2).with_vertices(p.vertices()).with_meshes(p.meshes()).with_transforms(p.transforms())
     .with_instances(p.instances()).with_cameras(p.cameras()).with_textures([t[:3] for t in p.textures()])
     .with_materials(p.materials()).with_lights(p.lights()).with_metadata(p.meta()).serialize())
    q, o = both(path2)
    assert np.array_equal(q.vertices().view(np.uint32), p.vertices().view(np.uint32))
    assert all(np.array_equal(a["indices"], b["indices"]) for a, b in zip(q.meshes(), p.meshes()))
    assert [bytes_of(m) for m in q.materials()] == [bytes_of(m) for m in p.materials()]
    assert all(np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(q.textures(), p.textures()))
    assert np.array_equal(o.vertices().view(np.uint32), p.vertices().view(np.uint32)) and len(o.textures()) == 3


def test_serialize_errors(tmp_path):
    with pytest.raises(abi.GlazeError):
        glaze_amd.Serializer(str(tmp_path / "no_such_dir" / "x.bin")).with_vertices(gen_vertices(3, 1)).serialize()
    d = abi.SerializeDescC()
    mesh = abi.Mesh(0, 0, 10, 5)                                # index range outside the (empty) index array
    d.meshes, d.n_meshes = C.cast(C.pointer(mesh), C.c_void_p), 1
    assert abi.lib().glz_serialize(str(tmp_path / "bad.bin").encode(), C.byref(d)) == abi.E_INVALID_INPUT
    assert abi.lib().glz_serialize(None, C.byref(d)) == abi.E_ARG
