"""C-ABI surface and device-independent host logic (CPU only; no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import make_camera
from oracle import pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "glaze_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glz_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = abi.lib()
    names = declared_symbols()
    assert len(names) >= 55
    for n in names:
        assert hasattr(lib, n), "libglaze_hip.so does not export " + n
    assert sorted(abi.PROTOTYPES) == names, "glaze_amd/abi.py and include/glaze_abi.h disagree: %s" % (
        set(abi.PROTOTYPES) ^ set(names))
    assert lib.glz_version().startswith(b"glaze-hip")


def test_struct_sizes_match_the_header():
    assert C.sizeof(abi.Vertex) == 32 and C.sizeof(abi.Mesh) == 12 and C.sizeof(abi.Transform) == 64 and C.sizeof(abi.MeshInstance) == 4
    assert C.sizeof(abi.Camera) == 52 and C.sizeof(abi.Meta) == 20
    assert C.sizeof(abi.Material) == 12 + 16 + 12 + 256 and C.sizeof(abi.Light) == 4 + 24 + 4 + 16 + 64 + 256


def test_no_device_means_none_not_a_fallback():
    """RayTraceInstance::new() -> None without a usable device (instance.rs:376-427); nothing renders on the CPU."""
    import torch
    inst = glaze_amd.RayTraceInstance.new()
    if not torch.cuda.is_available():
        assert inst is None
        assert abi.lib().glz_last_status() == abi.E_DEVICE
    with pytest.raises(glaze_amd.GlazeError):
        abi.check(abi.lib().glz_renderer_step(None, 1))
    assert abi.lib().glz_renderer_create(None, None, 4, 4) is None


def test_launch_constants_match_the_oracle():
    for seed in (0, 1, 0xDEADBEEFCAFE):
        for launch in (0, 1, 5, 21, 100, 341):
            s = C.c_uint32()
            off = (C.c_float * 2)()
            abi.check(abi.lib().glz_host_launch_constants(seed, launch, C.byref(s), off))
            so, oo = pyoracle.launch_constants(seed, launch)
            assert s.value == so and (off[0], off[1]) == oo
    seeds = set()
    for launch in range(64):
        s = C.c_uint32()
        off = (C.c_float * 2)()
        abi.lib().glz_host_launch_constants(7, launch, C.byref(s), off)
        seeds.add(s.value)
    assert len(seeds) == 64


@pytest.mark.parametrize("cam", [
    make_camera(),                                                                  # PerspectiveCam::default()
    make_camera(position=(-0.1331, 0.2942, -0.3347), target=(32.937, -47.133, 81.256), fovx=np.float32(0.87266), near=1e-3, far=100.0),
    make_camera(position=(1, 2, 3), target=(0, 0, 0), up=(0, 0, 1), orthographic=True, scale=2.5, near=0.1, far=40.0),
])
@pytest.mark.parametrize("res", [(1920, 1080), (512, 512), (33, 77)])
def test_push_constants_match_the_oracle_and_invert_the_view(cam, res):
    from glaze_amd.scene_desc import SceneDesc
    out = np.zeros(32, np.float32)
    abi.check(abi.lib().glz_host_push_constants(C.byref(cam), res[0], res[1], out.ctypes.data))
    # oracle: needs a renderer object; a scene with no geometry is enough
    from glaze_amd.scenes import cube_scene
    desc = cube_scene()
    desc.camera = cam
    o = pyoracle.OracleRenderer(pyoracle.OracleScene(desc), res[0], res[1], threads=1).push_constants()
    assert np.array_equal(out.view(np.uint32), o.view(np.uint32))
    c2w = out[:16].reshape(4, 4).T.astype(np.float64)
    assert np.allclose(c2w[3], (0, 0, 0, 1), atol=1e-6)
    assert np.allclose(c2w[:3, 3], tuple(cam.position), atol=1e-5)                # camera2world maps the origin to the eye
    fwd = np.array(tuple(cam.target)) - np.array(tuple(cam.position))
    fwd /= np.linalg.norm(fwd)
    assert np.allclose(c2w[:3, :3] @ np.array([0, 0, -1.0]), fwd, atol=1e-5)      # right-handed: the camera looks down -z
    assert abs(np.linalg.det(c2w[:3, :3]) - 1.0) < 1e-5


def test_tile_owner_partition():
    from glaze_amd.distributed import tile_owner
    for (w, h, world) in ((1920, 1080, 8), (200, 136, 3), (64, 64, 2), (1, 1, 1), (130, 70, 5)):
        own = tile_owner(w, h, world)
        assert own.shape == (h, w) and own.max() < world
        tiles_x = (w + 63) // 64
        y, x = np.mgrid[0:h, 0:w]
        assert np.array_equal(own, (((y // 64) * tiles_x + x // 64) % world).astype(np.uint16))
        counts = np.bincount(own.ravel(), minlength=world)
        if w * h >= 64 * 64 * world * 4:
            assert counts.min() > 0.5 * counts.max()                                  # interleaved tiles balance the load


def test_scene_desc_round_trips_through_ctypes():
    from glaze_amd.scenes import cube_scene
    d = cube_scene()
    c = d.as_c()
    assert c.n_vertices == 24 and c.n_indices == 36 and c.n_meshes == 1 and c.n_materials == 3 and c.n_textures == 2 and c.n_lights == 1
    mats = C.cast(c.materials, C.POINTER(abi.Material))
    assert mats[2].name == b"Material" and tuple(mats[2].diffuse_mul) == (204, 204, 204) and mats[2].diffuse == 1
    d2 = d.copy()
    d2.materials[2].mtype = abi.MAT_GLASS
    assert d.materials[2].mtype == abi.MAT_LAMBERT


def test_launch_chains_partition_a_ranks_tiles():
    """glz_host_chain_owner: chain s of S renders the finer partition (rank + s * world, world * S); together the chains cover the
    rank's tiles exactly once, and the automatic count follows the pixels per rank (1 chain from a million pixels, 2 from 400 k, else 3)."""
    from glaze_amd.distributed import chain_owner, tile_owner
    W, H = 1920, 1080
    expect = {1: 1, 2: 1, 4: 2, 8: 3}
    for world, chains in expect.items():
        ranks = tile_owner(W, H, world)
        seen = np.zeros((H, W), np.int32)
        for rank in range(world):
            n, own = chain_owner(W, H, rank, world)
            assert n == chains, (world, n)
            mine = own != 0xFFFF
            assert np.array_equal(mine, ranks == rank)
            assert set(np.unique(own[mine]).tolist()) == set(range(n))
            # a chain's tiles are whole 64x64 tiles
            t = own[::64, ::64]
            assert np.array_equal(np.repeat(np.repeat(t, 64, 0), 64, 1)[:H, :W], own)
            seen += mine
        assert (seen == 1).all()
    assert chain_owner(3840, 2160, 3, 8)[0] == 1                      # 4K over 8 GPUs: a million pixels per rank
    assert chain_owner(W, H, 0, 8, chains=5)[0] == 5 and chain_owner(64, 64, 0, 1, chains=4)[0] == 1    # never more chains than tiles
    n, own = chain_owner(200, 136, 1, 3, chains=2)                     # ragged edges
    assert n == 2 and (own[tile_owner(200, 136, 3) == 1] != 0xFFFF).all()


def _sah_tree(lo, hi):
    import ctypes as C
    n = lo.shape[0]
    lo4 = np.zeros((n, 4), np.float32); lo4[:, :3] = lo
    hi4 = np.zeros((n, 4), np.float32); hi4[:, :3] = hi
    children = np.full((n - 1, 2), 0x7FFFFFF0, np.int32)
    parent = np.full(2 * n - 1, 0x7FFFFFF0, np.int32)
    abi.check(abi.lib().glz_host_build_sah(n, lo4.ctypes.data, hi4.ctypes.data, children.ctypes.data, parent.ctypes.data))
    return children, parent


def _check_tree(children, parent, n):
    """every leaf and every inner node but the root has exactly one parent, links and parents agree, ids are the pre-order"""
    assert parent[0] == -1
    links = children.reshape(-1).astype(np.int64)
    leaf, inner = ~links[links < 0], links[links >= 0]
    assert np.array_equal(np.sort(leaf), np.arange(n)) and np.array_equal(np.sort(inner), np.arange(1, n - 1))
    for side in (0, 1):
        l = children[:, side].astype(np.int64)
        slot = np.where(l >= 0, l, (n - 1) + ~l)
        assert np.array_equal(parent[slot], np.arange(n - 1))
    # a subtree over c leaves owns c - 1 consecutive ids: the left child of node i is i + 1, the right one follows the left subtree
    left = children[:, 0]
    assert (left[left >= 0] == np.nonzero(left >= 0)[0] + 1).all()


def test_host_sah_builder_structure_and_determinism():
    """bvh_sah.cpp without a GPU: random boxes, clustered boxes, all boxes equal (no split by binning: halving), two leaves."""
    rng = np.random.default_rng(3)
    n = 70000                                                      # above the threshold where the top ranges are binned in parallel
    c = np.concatenate([rng.random((n // 2, 3)) * 10, rng.normal(size=(n - n // 2, 3)) * 0.01 + [50, -20, 5]]).astype(np.float32)
    ext = (rng.random((n, 3)) * 0.05).astype(np.float32)
    ch, pa = _sah_tree(c - ext, c + ext)
    _check_tree(ch, pa, n)
    ch2, pa2 = _sah_tree(c - ext, c + ext)
    assert np.array_equal(ch, ch2) and np.array_equal(pa, pa2)     # threads do not change the tree
    # the split of the root separates the two clusters
    def leaves_under(link):
        stack, out = [link], []
        while stack:
            l = stack.pop()
            if l < 0: out.append(~l)
            else: stack.extend(ch[l].tolist())
        return np.array(out)
    a, b = leaves_under(int(ch[0, 0])), leaves_under(int(ch[0, 1]))
    first, second = (a, b) if a.min() < n // 2 else (b, a)
    assert len(a) + len(b) == n and (first < n // 2).all() and (second >= n // 2).all()
    same = np.tile(np.array([[1.0, 2.0, 3.0]], np.float32), (257, 1))
    ch, pa = _sah_tree(same, same + 1)
    _check_tree(ch, pa, 257)
    ch, pa = _sah_tree(np.zeros((2, 3), np.float32), np.ones((2, 3), np.float32))
    assert sorted((~ch[0]).tolist()) == [0, 1] and pa.tolist() == [-1, 0, 0]
    nan = c[:1000].copy(); nan[::7] = np.nan                      # degenerate input must still give a tree
    ch, pa = _sah_tree(nan, nan)
    _check_tree(ch, pa, 1000)


def test_srgb8_quantiser_rule_host_table_equals_the_oracle_count():
    """The 8-bit export (raytracer.rs:576-584 blit) is stated as thresholds, not as pow() per pixel: the table behind the C ABI
    (what k_tonemap searches) and the oracle's count over the rule give the same byte for every float tried, including the
    thresholds themselves and their neighbours; away from the thresholds the rule is round(255 * OETF(c)) in double precision."""
    thr = np.zeros(256, np.float32)
    abi.check(abi.lib().glz_host_srgb8_thresholds(thr.ctypes.data))
    assert thr[0] == 0.0 and (np.diff(thr) > 0).all() and thr[255] < 1.0
    L = pyoracle.lib()
    rng = np.random.default_rng(0)
    v = np.concatenate([thr, np.nextafter(thr, np.float32(-1)), np.nextafter(thr, np.float32(2)), rng.random(20000, dtype=np.float32),
                        rng.random(5000, dtype=np.float32) * 0.01, np.array([0, -1, 1, 2, np.inf, 1e-45], np.float32)])
    got = np.searchsorted(thr[1:], v, side="right")
    want = np.array([L.orc_to_srgb8(float(c)) for c in v])
    assert np.array_equal(got, want)
    assert L.orc_to_srgb8(float("nan")) == 0 and L.orc_to_srgb8(-0.0) == 0 and L.orc_to_srgb8(float("inf")) == 255
    x = rng.random(20000).astype(np.float32)
    enc = np.where(x <= 0.0031308, 12.92 * x.astype(np.float64), 1.055 * np.power(x.astype(np.float64), 1 / 2.4) - 0.055) * 255.0
    clear = np.abs(enc - np.floor(enc) - 0.5) > 1e-4                       # not within rounding noise of a threshold
    assert np.array_equal(np.searchsorted(thr[1:], x[clear], side="right"), np.floor(enc[clear] + 0.5).astype(int))


def test_oracle_tile_subset_equals_the_full_frame():
    """Pixels are independent (absolute-pixel RNG, path_trace.rgen:143-147): the oracle rendering only some 64x64 tiles gives those
    tiles of the full render bit for bit and touches nothing else -- what the full-size GPU spot checks rely on."""
    from glaze_amd.scenes import cube_scene
    from oracle.pyoracle import OracleRenderer, OracleScene
    d = cube_scene()
    w, h, tiles = 200, 136, [0, 3, 5, 11]                                   # 4 x 3 tiles, ragged right and bottom edges
    full = OracleRenderer(OracleScene(d), w, h)
    full.set_depth(3)
    full.step(5)
    part = OracleRenderer(OracleScene(d), w, h)
    part.set_depth(3)
    part.set_tiles(tiles)
    part.step(5)
    m = np.zeros((h, w), bool)
    for t in tiles:
        m[(t // 4) * 64:(t // 4) * 64 + 64, (t % 4) * 64:(t % 4) * 64 + 64] = True
    a, b = full.read_hdr(), part.read_hdr()
    assert np.array_equal(a[m].view(np.uint32), b[m].view(np.uint32)) and not b[~m].any()
    assert np.array_equal(full.read_rgba8()[m], part.read_rgba8()[m])
    part.set_tiles([])
    part.step(5)
    assert np.array_equal(part.read_hdr().view(np.uint32), a.view(np.uint32))


def _host_mip_chain(fmt, px):
    tex = abi.Texture()
    tex.format, tex.width, tex.height, tex.mip_levels = fmt, px.shape[1], px.shape[0], 1
    keep = np.ascontiguousarray(px)
    tex.pixels = keep.ctypes.data
    levels = []
    for level in range(32):
        w, h = C.c_uint32(), C.c_uint32()
        n = abi.check(abi.lib().glz_host_mip_level(C.byref(tex), level, None, 0, C.byref(w), C.byref(h)))
        if n == 0:
            break
        out = np.zeros(n, np.uint8)
        abi.check(abi.lib().glz_host_mip_level(C.byref(tex), level, out.ctypes.data, n, C.byref(w), C.byref(h)))
        levels.append(out.reshape(h.value, w.value) if fmt == abi.TEX_GRAY else out.reshape(h.value, w.value, 4))
    return levels


@pytest.mark.parametrize("fmt,shape", [(abi.TEX_RGBA_SRGB, (64, 64)), (abi.TEX_RGBA_NORM, (32, 128)), (abi.TEX_GRAY, (16, 16)),
                                       (abi.TEX_RGBA_SRGB, (5, 13)), (abi.TEX_GRAY, (1, 7)), (abi.TEX_RGBA_NORM, (3, 1))])
def test_generated_mip_chain_matches_the_oracle_and_the_blit_rule(fmt, shape):
    """Textures that bring no mip levels get them the way load_texture_to_gpu generates them (scene.rs:1012-1263): level l from
    level l - 1 by a LINEAR blit, max(1, w >> l) x max(1, h >> l), 1 + floor(log2(max(w, h))) levels (texture.rs:200-207).  The
    library's chain (mipchain.h, what Scene::ensure_mips uploads) equals the oracle's restatement byte for byte; for 2:1 steps a
    texel is the average of 2 x 2 -- in linear light for sRGB colour channels."""
    from glaze_amd.scenes import cube_scene
    from oracle.pyoracle import OracleScene
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    px = rng.integers(0, 256, shape if fmt == abi.TEX_GRAY else shape + (4,), dtype=np.uint8)
    mine = _host_mip_chain(fmt, px)
    assert len(mine) == 1 + int(np.floor(np.log2(max(shape))))
    desc = cube_scene()
    desc.textures.append((fmt, px, "t"))
    o = OracleScene(desc)
    tid = len(desc.textures) - 1
    for l, m in enumerate(mine):
        assert m.shape[:2] == (max(1, shape[0] >> l), max(1, shape[1] >> l))
        assert np.array_equal(m, o.texture_level(tid, l)), "level %d" % l
    assert o.texture_level(tid, len(mine)) is None
    if shape == (64, 64):                                                   # exact halving: plain 2 x 2 averages
        lut = np.where(np.arange(256) / 255.0 <= 0.04045, np.arange(256) / 255.0 / 12.92, ((np.arange(256) / 255.0 + 0.055) / 1.055) ** 2.4)
        lin = lut[px[..., :3]].reshape(32, 2, 32, 2, 3).mean((1, 3))
        enc = np.where(lin <= 0.0031308, 12.92 * lin, 1.055 * lin ** (1 / 2.4) - 0.055) * 255.0
        assert np.abs(mine[1][..., :3].astype(np.float64) - enc).max() <= 0.5 + 1e-6          # round to nearest code
        alpha = px[..., 3].astype(np.float64).reshape(32, 2, 32, 2).mean((1, 3))
        assert np.abs(mine[1][..., 3] - alpha).max() <= 0.5 + 1e-9
