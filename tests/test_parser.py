"""`.glaze` V1 read side: the C++ reader behind the C ABI (own XXH64 / XZ-LZMA2 / PNG decoders) against
 (1) the reference's fixture resources/mattest.glaze (tests/golden/mattest.glaze, contents SURVEY F10),
 (2) the oracle's independent python reader (oracle/glaze_v1.py: liblzma, xxhash, PIL),
 (3) files written by tests/glaze_writer.py -- the properties of the reference's parser tests
     (write -> parse -> compare per chunk and for all chunks, lib/src/parser/v1.rs:1489-1748;
      corruption -> Err, :1750-1973; missing chunk -> empty, :294-323).
CPU only: BASELINE config 1.
"""
import lzma
import os
import struct

import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from oracle import glaze_v1

from conftest import MATTEST
from glaze_writer import write_glaze


# ---------------------------------------------------------------------------------------------
# the reference fixture
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def mattest():
    return glaze_amd.parse(MATTEST), glaze_v1.parse(MATTEST)


def test_mattest_counts_f10(mattest):
    p, _ = mattest
    assert p.vertices().shape == (70876, 8)
    meshes = p.meshes()
    assert [len(m["indices"]) for m in meshes] == [9600, 191232, 214608] and sum(len(m["indices"]) for m in meshes) // 3 == 138480
    assert [m["material"] for m in meshes] == [2, 4, 3] and [m["id"] for m in meshes] == [0, 1, 2]
    assert p.instances().tolist() == [[0, 0], [1, 0], [2, 0]]
    assert np.array_equal(p.transforms(), np.eye(4, dtype=np.float32).reshape(1, 16))
    mats = p.materials()
    assert [m.name.decode() for m in mats] == ["default", "DefaultMaterial", "BackGroundMat", "InnerMat", "OuterMat"]
    assert [m.mtype for m in mats] == [abi.MAT_LAMBERT] * 3 + [abi.MAT_METAL, abi.MAT_GLASS]
    assert abs(mats[3].roughness_mul - 0.2209) < 1e-6 and abs(mats[4].ior - 1.46) < 1e-6 and mats[2].diffuse == 2
    lights = p.lights()
    assert len(lights) == 1 and lights[0].ltype == abi.LIGHT_SKY and lights[0].resource_id == 1
    assert abs(lights[0].intensity - 0.019) < 1e-7 and abs(lights[0].yaw_deg - 26.341) < 1e-4
    tex = p.textures()
    assert [(t[2], t[1].shape, t[0], t[3]) for t in tex] == [("default", (1, 1, 4), 2, 1), ("dome_e", (1024, 1024, 4), 2, 1),
                                                             ("Checker", (1024, 1024, 4), 2, 1)]
    cams = p.cameras()
    assert len(cams) == 1 and cams[0].type == abi.CAMERA_PERSPECTIVE and abs(cams[0].fovx_or_scale - 0.87266) < 1e-5
    meta = p.meta()
    assert abs(meta.scene_radius - 3.27297) < 1e-5 and abs(meta.exposure - 0.157) < 1e-7
    assert glaze_amd.converted_file(MATTEST) and not glaze_amd.converted_file(__file__)


def test_mattest_bytes_match_the_oracle_reader(mattest):
    p, o = mattest
    assert np.array_equal(p.vertices().view(np.uint32), o.vertices().view(np.uint32))
    for a, b in zip(p.meshes(), o.meshes()):
        assert a["id"] == b["id"] and a["material"] == b["material"] and np.array_equal(a["indices"], b["indices"])
    assert np.array_equal(p.transforms(), o.transforms()) and np.array_equal(p.instances(), o.instances())
    for a, b in zip(p.textures(), o.textures()):
        assert a[0] == b["format"] and a[2] == b["name"] and np.array_equal(a[1], b["levels"][0])
    for a, b in zip(p.materials(), o.materials()):
        assert (a.mtype, a.metal, tuple(a.diffuse_mul), a.diffuse, a.roughness, a.metalness, a.normal, a.opacity) == \
               (b["mtype"], b["metal"], b["diffuse_mul"], b["diffuse"], b["roughness"], b["metalness"], b["normal"], b["opacity"])
        assert (a.ior, a.roughness_mul, a.metalness_mul, a.anisotropy) == tuple(np.float32(b[k]) for k in ("ior", "roughness_mul", "metalness_mul", "anisotropy"))
        assert a.name.decode() == b["name"] and bool(a.has_emissive) == (b["emissive"] is not None)
    for a, b in zip(p.lights(), o.lights()):
        assert a.ltype == b["ltype"] and np.array_equal(np.array(a.color[:], np.float32), b["color"]) and a.name.decode() == b["name"]
        assert tuple(a.position) == tuple(np.float32(x) for x in b["position"])
    oc = o.cameras()[0]
    pc = p.cameras()[0]
    assert tuple(pc.position) == tuple(np.float32(x) for x in oc["position"]) and pc.far_plane == np.float32(oc["far"])


# ---------------------------------------------------------------------------------------------
# round trips through the test writer
# ---------------------------------------------------------------------------------------------
def _random_scene(rng, n_vert=1000, n_mesh=3):
    v = rng.normal(size=(n_vert, 8)).astype(np.float32)
    meshes = [dict(id=i, material=int(rng.integers(0, 4)), indices=rng.integers(0, n_vert, size=3 * int(rng.integers(1, 200))).astype(np.uint32))
              for i in range(n_mesh)]
    mats = [dict(mtype=int(rng.integers(0, 7)), metal=int(rng.integers(0, 29)), diffuse_mul=tuple(int(x) for x in rng.integers(0, 256, 3)),
                 emissive=None if i % 2 else tuple(int(x) for x in rng.integers(1, 256, 3)), ior=float(np.float32(rng.random() + 1)),
                 roughness_mul=float(np.float32(rng.random())), metalness_mul=float(np.float32(rng.random())),
                 anisotropy=float(np.float32(rng.random() * 2 - 1)), diffuse=1, roughness=0, metalness=1, normal=0, opacity=0,
                 name="mat%d é中" % i) for i in range(4)]
    lights = [dict(ltype=i % 4, position=tuple(float(np.float32(x)) for x in rng.normal(size=3)),
                   direction=tuple(float(np.float32(x)) for x in rng.normal(size=3)), resource_id=int(rng.integers(0, 2)),
                   intensity=float(np.float32(rng.random())), yaw=10.0, pitch=20.0, roll=30.0, color=rng.random(16).astype(np.float32), name="L%d" % i)
              for i in range(5)]
    cams = [dict(type=0, position=(1.0, 2.0, 3.0), target=(0.0, 0.5, -1.0), up=(0.0, 1.0, 0.0), fovx_or_scale=1.25, near=0.5, far=50.0),
            dict(type=1, position=(4.0, 5.0, 6.0), target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), fovx_or_scale=2.0, near=0.25, far=75.0)]
    textures = [dict(format=2, name="default", levels=[np.full((1, 1, 4), 255, np.uint8)]),
                dict(format=1, name="gray ü", levels=[rng.integers(0, 256, (37, 19), dtype=np.uint8)]),
                dict(format=3, name="rgba", levels=[rng.integers(0, 256, (16, 32, 4), dtype=np.uint8), rng.integers(0, 256, (8, 16, 4), dtype=np.uint8)])]
    transforms = rng.normal(size=(3, 16)).astype(np.float32)
    instances = np.array([[0, 0], [1, 2], [2, 1], [0, 1]], np.uint16)
    meta = dict(scene_centre=(1.0, -2.0, 3.5), scene_radius=12.5, exposure=0.75)
    return dict(vertices=v, meshes=meshes, cameras=cams, textures=textures, materials=mats, transforms=transforms, instances=instances,
                lights=lights, meta=meta)


def _compare(p, s):
    assert np.array_equal(p.vertices().view(np.uint32), s["vertices"].view(np.uint32))
    for a, b in zip(p.meshes(), s["meshes"]):
        assert a["id"] == b["id"] and a["material"] == b["material"] and np.array_equal(a["indices"], b["indices"])
    assert np.array_equal(p.transforms(), s["transforms"]) and np.array_equal(p.instances(), s["instances"])
    mats = p.materials()
    assert len(mats) == len(s["materials"])
    for a, b in zip(mats, s["materials"]):
        assert a.mtype == b["mtype"] and a.metal == b["metal"] and tuple(a.diffuse_mul) == b["diffuse_mul"]
        assert bool(a.has_emissive) == (b["emissive"] is not None) and (not a.has_emissive or tuple(a.emissive_col) == b["emissive"])
        assert a.name.decode("utf8") == b["name"] and a.ior == np.float32(b["ior"]) and a.anisotropy == np.float32(b["anisotropy"])
    lights = p.lights()
    assert len(lights) == len(s["lights"])
    for a, b in zip(lights, s["lights"]):
        assert a.ltype == b["ltype"] and np.array_equal(np.array(a.color[:], np.float32), b["color"]) and a.resource_id == b["resource_id"]
    cams = p.cameras()
    assert [c.type for c in cams] == [c["type"] for c in s["cameras"]] and cams[1].fovx_or_scale == 2.0 and tuple(cams[0].target) == (0.0, 0.5, -1.0)
    tex = p.textures()
    for a, b in zip(tex, s["textures"]):
        assert a[0] == b["format"] and a[2] == b["name"] and np.array_equal(a[1], b["levels"][0]) and a[3] == len(b["levels"])
    m = p.meta()
    assert tuple(m.scene_centre) == s["meta"]["scene_centre"] and m.scene_radius == 12.5 and m.exposure == 0.75


@pytest.mark.parametrize("preset,check", [(0, lzma.CHECK_CRC64), (6, lzma.CHECK_CRC32), (9, lzma.CHECK_CRC64), (9 | lzma.PRESET_EXTREME, lzma.CHECK_NONE),
                                          (3, lzma.CHECK_SHA256)])
def test_write_parse_all_chunks(tmp_path, preset, check):
    rng = np.random.default_rng(preset)
    s = _random_scene(rng)
    path = str(tmp_path / "all.glaze")
    write_glaze(path, preset=preset, check=check, **s)
    _compare(glaze_amd.parse(path), s)


def test_xz_decoder_on_hard_inputs(tmp_path):
    """LZMA2 corner cases: incompressible data (uncompressed chunks), long runs (rep matches, 273-byte lengths), > 2 MiB
    (several LZMA2 chunks with state carried over), tiny inputs."""
    rng = np.random.default_rng(7)
    cases = [rng.integers(0, 256, 32 * 100003, dtype=np.uint8).tobytes(),          # incompressible, 3.2 MB
             np.zeros(32 * 200000, np.uint8).tobytes(),                           # 6.4 MB of zeros
             np.tile(rng.integers(0, 256, 32 * 7, dtype=np.uint8), 30000).tobytes(),   # periodic
             rng.integers(0, 4, 32 * 50000, dtype=np.uint8).tobytes(),            # low entropy
             bytes(range(32))]
    for i, raw in enumerate(cases):
        v = np.frombuffer(raw, "<f4").reshape(-1, 8)
        for preset in (1, 9):
            path = str(tmp_path / ("xz%d_%d.glaze" % (i, preset)))
            write_glaze(path, vertices=v, preset=preset)
            got = glaze_amd.parse(path).vertices()
            assert got.tobytes() == raw


def test_single_chunks_and_missing_chunks(tmp_path):
    """write one chunk type at a time (v1.rs:1489-1700); absent chunks read back empty, absent meta = Meta::default()."""
    s = _random_scene(np.random.default_rng(3), n_vert=50)
    for key in ("vertices", "meshes", "cameras", "textures", "materials", "transforms", "instances", "lights", "meta"):
        path = str(tmp_path / (key + ".glaze"))
        write_glaze(path, **{key: s[key]})
        p = glaze_amd.parse(path)
        counts = dict(vertices=len(p.vertices()), meshes=len(p.meshes()), cameras=len(p.cameras()), textures=len(p.textures()),
                      materials=len(p.materials()), transforms=len(p.transforms()), instances=len(p.instances()), lights=len(p.lights()))
        for k, n in counts.items():
            assert (n > 0) == (k == key), (key, k, n)
        m = p.meta()
        if key != "meta":
            assert (m.scene_radius, m.exposure, tuple(m.scene_centre)) == (100.0, 1.0, (0.0, 0.0, 0.0))     # parser/mod.rs:280-288
    path = str(tmp_path / "empty.glaze")
    write_glaze(path)
    p = glaze_amd.parse(path)
    assert len(p.vertices()) == 0 and p.meshes() == [] and p.lights() == []


def test_unknown_chunks_are_ignored_and_defaults_applied(tmp_path):
    s = _random_scene(np.random.default_rng(4), n_vert=20)
    path = str(tmp_path / "unk.glaze")
    write_glaze(path, vertices=s["vertices"], extra_chunks=[(99, b"future chunk"), (200, b"x" * 100)])      # v1.rs:160-162
    assert len(glaze_amd.parse(path).vertices()) == 20
    # unknown material type / metal ids fall back to LAMBERT / SILVER (material.rs:290-298, metal.rs:418-451)
    mats = [dict(s["materials"][0], mtype=77, metal=200)]
    write_glaze(path, materials=mats)
    m = glaze_amd.parse(path).materials()[0]
    assert m.mtype == abi.MAT_LAMBERT and m.metal == 0


# ---------------------------------------------------------------------------------------------
# corruption (v1.rs:1750-1973)
# ---------------------------------------------------------------------------------------------
def _status(fn):
    with pytest.raises(glaze_amd.GlazeError) as e:
        fn()
    return e.value.status


def test_header_errors(tmp_path):
    s = _random_scene(np.random.default_rng(5), n_vert=20)
    path = str(tmp_path / "h.glaze")
    data = bytearray(write_glaze(path, **s))
    assert _status(lambda: glaze_amd.parse(str(tmp_path / "does_not_exist.glaze"))) == abi.E_IO
    open(path, "wb").write(b"")
    assert _status(lambda: glaze_amd.parse(path)) == abi.E_INVALID_INPUT          # "Wrong or empty input file"
    open(path, "wb").write(b"glazy" + bytes(data[5:]))
    assert _status(lambda: glaze_amd.parse(path)) == abi.E_INVALID_INPUT
    open(path, "wb").write(bytes(data[:5]) + b"\x02" + bytes(data[6:]))
    assert _status(lambda: glaze_amd.parse(path)) == abi.E_INVALID_INPUT          # "Unsupported file version"
    open(path, "wb").write(bytes(data[:20]))
    assert _status(lambda: glaze_amd.parse(path)) in (abi.E_IO, abi.E_INVALID_DATA)
    bad = bytearray(data)
    bad[16 + 8 + 3] ^= 0xFF                                                        # inside the offsets table
    open(path, "wb").write(bytes(bad))
    assert _status(lambda: glaze_amd.parse(path)) == abi.E_INVALID_DATA           # "Corrupted file structure"
    bad = bytearray(data)
    bad[16] ^= 0x01                                                                # the table hash itself
    open(path, "wb").write(bytes(bad))
    assert _status(lambda: glaze_amd.parse(path)) == abi.E_INVALID_DATA


def test_chunk_corruption_is_detected_per_chunk(tmp_path):
    s = _random_scene(np.random.default_rng(6), n_vert=200)
    path = str(tmp_path / "c.glaze")
    data = write_glaze(path, **s)
    ref = glaze_v1.parse(path)
    getters = dict(vertex="vertices", mesh="meshes", camera="cameras", texture="textures", material="materials", transform="transforms",
                   instance="instances", light="lights", meta="meta")
    for name, (off, ln) in ref.chunks.items():
        for pos in (off + 3, off + 8 + ln // 2, off + ln - 1):                     # hash bytes, payload middle, payload end
            bad = bytearray(data)
            bad[pos] ^= 0x40
            open(path, "wb").write(bytes(bad))
            p = glaze_amd.parse(path)                                              # header + table are still fine
            assert _status(getattr(p, getters[name])) == abi.E_INVALID_DATA, (name, pos)
            for other, g in getters.items():
                if other != name:
                    getattr(p, g)()                                                # every other chunk still reads (v1.rs:1760-1768)


def test_truncated_file_and_bad_payloads(tmp_path):
    s = _random_scene(np.random.default_rng(8), n_vert=300)
    path = str(tmp_path / "t.glaze")
    data = write_glaze(path, **s)
    ref = glaze_v1.parse(path)
    last = max(ref.chunks.items(), key=lambda kv: kv[1][0])
    open(path, "wb").write(data[:-5])
    p = glaze_amd.parse(path)
    getters = dict(vertex="vertices", mesh="meshes", camera="cameras", texture="textures", material="materials", transform="transforms",
                   instance="instances", light="lights", meta="meta")
    assert _status(getattr(p, getters[last[0]])) in (abi.E_IO, abi.E_INVALID_DATA)
    # a chunk whose hash is right but whose payload is not xz
    write_glaze(path, extra_chunks=[(0, b"this is not an xz stream")])
    assert _status(glaze_amd.parse(path).vertices) == abi.E_INVALID_DATA
    # xz stream with a flipped byte but a recomputed chunk hash: the xz CRC / structure check must catch it
    import xxhash
    body = bytearray(lzma.compress(np.arange(8000, dtype="<f4").tobytes(), preset=6))
    body[len(body) // 2] ^= 0x10
    write_glaze(path, extra_chunks=[(0, bytes(body))])
    assert _status(glaze_amd.parse(path).vertices) == abi.E_INVALID_DATA
    # invalid enum values that make the reference panic -> InvalidData here
    cam = dict(s["cameras"][0], type=7)
    write_glaze(path, cameras=[cam])
    assert _status(glaze_amd.parse(path).cameras) == abi.E_INVALID_DATA
    light = dict(s["lights"][0], ltype=9)
    write_glaze(path, lights=[light])
    assert _status(glaze_amd.parse(path).lights) == abi.E_INVALID_DATA
    # PNG with a broken IDAT
    t = s["textures"][2]
    write_glaze(path, textures=[t])
    raw = bytearray(open(path, "rb").read())
    idat = raw.find(b"IDAT")
    raw[idat + 10] ^= 0xFF
    # recompute the chunk hash so only the PNG layer can object
    off, ln = glaze_v1.parse(path).chunks["texture"]
    raw[off:off + 8] = struct.pack("<Q", xxhash.xxh64(bytes(raw[off + 8:off + ln]), seed=0x368262AAA1DEB64D).intdigest())
    open(path, "wb").write(bytes(raw))
    assert _status(glaze_amd.parse(path).textures) == abi.E_INVALID_DATA


def test_parse_time_reported(mattest):
    """BASELINE config 1 reports parse ms: all nine getters on mattest.glaze."""
    import time
    t = time.perf_counter()
    p = glaze_amd.parse(MATTEST)
    for g in ("vertices", "meshes", "transforms", "instances", "cameras", "textures", "materials", "lights", "meta"):
        getattr(p, g)()
    ms = (time.perf_counter() - t) * 1e3
    print("mattest.glaze full parse: %.1f ms" % ms)
    assert ms < 5000
