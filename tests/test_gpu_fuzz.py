"""Randomised parity (a short version of tools/gpu_fuzz_parity.py, which ran thousands of scenes clean at the end of round 4): the scenes
of tests/fuzz_scenes.py -- random meshes, instances, materials, textures, lights, cameras, and a random way to render each (launch
mode, acceleration-structure levels, chains, integrator, depth, size) -- rendered by the HIP path and by the oracle, bit for bit.

The seeds listed by name are the ones that found something:
  * 297, 515, 520, 564: the ORACLE's hierarchy lost exact ties between coincident instances (tests/test_oracle_fuzz.py);
  * 118, 476, 653, 884, 986, 1340, 1401: the TWO-LEVEL tracer padded the object-space boxes by a number of cells that was capped at
    twice the grid's span -- across the thin side of a flat mesh (cells of 1e-13 units) far less than the rounding of the
    transformed ray: coplanar meshes under one transform lost near ties, thin meshes hits (trace_wave_tl, set_grid_ray);
  * 61907: a FLAT SCENE (every instance a grid in the plane y = 0): the flattened structure's grid had cells of 1e-13 units across
    the thin side, a box's entry distance -- the plane's, by an fma in grid space -- could round one ulp past the distance the
    triangle test gives, and with a hit in hand the boxes of the coincident triangles of other instances were left out: the tie
    went to whichever was met first.  The grid now reaches 2^-19 of the largest coordinate past the bounds and every box grows by
    as much (kernels_build.hip k_grid_params, grid_margin);
  * 3257, 11494, 50759, 68482, 73991, 100817, 102401, 102759, 103497, 107249, 223644, 230234, 234947, 235554, 236777, 244722: a shadow ray
    IN THE PLANE of a triangle (towards a light coplanar with the surface it leaves): det is rounding noise, the distance anything, and
    whether the candidate was even offered depended on the structure that walked the scene -- the oracle's own hierarchy included.
    Both sides now leave out a candidate whose |det| is not above 2^-19 of its products (oracle.cpp ray_tri, triangle_finish).
"""
import numpy as np
import pytest

import glaze_amd
from fuzz_scenes import LARGE, col_major, random_scene, render_both, rot
from oracle.pyoracle import OracleScene

pytestmark = pytest.mark.gpu

FOUND_SOMETHING = [118, 297, 476, 515, 520, 564, 653, 884, 986, 1340, 1401, 61907]
RAYS_IN_A_PLANE = [3257, 11494, 50759, 68482, 73991, 100817, 102401, 102759, 103497, 107249, 223644, 230234, 234947, 235554, 236777, 244722]


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


render_pair = render_both


def test_random_scenes_bit_identical_to_the_oracle():
    lit = 0
    for seed in list(range(150)) + FOUND_SOMETHING + RAYS_IN_A_PLANE + list(range(LARGE, LARGE + 12)):
        desc, run = random_scene(seed)
        r, o = render_pair(desc, run)
        g, c = r.read_hdr(), o.read_hdr()
        assert np.array_equal(bits(g), bits(c)), "seed %d (%s): accumulators differ in %d pixels" % (seed, run, int((bits(g) != bits(c)).any(-1).sum()))
        assert np.array_equal(bits(r.read_result()), bits(o.read_result())), "seed %d (%s): result images differ" % (seed, run)
        assert np.array_equal(r.read_rgba8(), o.read_rgba8()), "seed %d (%s): 8-bit images differ" % (seed, run)
        lit += bool(np.nan_to_num(c[..., :3]).any())
    assert lit > 80          # most of the scenes show something


@pytest.mark.parametrize("seed", FOUND_SOMETHING)
def test_the_seeds_that_found_something_in_every_configuration(seed):
    desc, run = random_scene(seed)
    for levels in ("flat", "two_level"):
        for mode in ("two_kernels", "path"):
            r, o = render_pair(desc, run, levels, mode)
            assert np.array_equal(bits(r.read_hdr()), bits(o.read_hdr())), (seed, levels, mode)
            assert np.array_equal(bits(r.read_result()), bits(o.read_result())), (seed, levels, mode)


def test_coplanar_flat_meshes_under_one_transform_trace_like_brute_force():
    """Seed 884's pattern stated on purpose: flat grids of different tessellation in the same plane, instanced with the same rotated,
    non-uniformly scaled transform -- their hits are 1-2 ulps apart along most rays, the nearer must win whatever the hierarchy,
    and the object-space boxes of a mesh without thickness must not be missed."""
    from glaze_amd.scene_desc import INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc

    def grid(n):
        s, t = np.meshgrid(np.linspace(-0.5, 0.5, n + 1), np.linspace(-0.5, 0.5, n + 1), indexing="ij")
        pos = np.stack([s, np.zeros_like(s), t], -1).reshape(-1, 3)
        idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        return pos, np.stack([a, b, c, a, c, d], -1).reshape(-1)
    parts, meshes, nv, ni = [], [], 0, 0
    for n in (2, 3, 5):
        pos, tri = grid(n)
        block = np.zeros(len(pos), VERTEX_DTYPE)
        block["vv"], block["vn"], block["vt"] = pos, (0, 1, 0), pos[:, [0, 2]]
        parts.append((block, tri.astype(np.uint32) + nv))
        meshes.append((len(meshes), 0, ni, len(tri)))
        nv += len(pos)
        ni += len(tri)
    t = np.eye(4)
    t[:3, 3] = (-1.08, 0.56, 0.30)
    xf = t @ rot(2, 214.6) @ np.diag([1.06, 0.57, 0.55, 1.0])
    desc = SceneDesc(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), np.array(meshes, MESH_DTYPE),
                     np.stack([col_major(np.eye(4)), col_major(xf)]), np.array([(0, 1), (1, 1), (2, 1), (1, 0)], INSTANCE_DTYPE))
    osc = OracleScene(desc)
    rng = np.random.default_rng(5)
    n = 60000
    o = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    on_plane = np.concatenate([rng.uniform(-0.5, 0.5, (n, 1)), np.zeros((n, 1)), rng.uniform(-0.5, 0.5, (n, 1)), np.ones((n, 1))], 1)
    d = ((xf @ on_plane.T).T[:, :3] - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    bt, btri = osc.trace_closest(o, d, brute=True)
    hit = np.isfinite(bt)
    assert hit.mean() > 0.95
    for levels in ("flat", "two_level"):
        inst = glaze_amd.RayTraceInstance.new()
        inst.set_as_levels(levels)
        gt, gtri, _, _, _ = glaze_amd.RayTraceScene.from_desc(inst, desc).debug_trace_closest(o, d)
        assert np.array_equal(gt.view(np.uint32), bt.view(np.uint32)) and np.array_equal(gtri[hit], btri[hit]), levels
