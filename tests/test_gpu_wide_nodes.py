"""The 8-wide nodes of the hierarchy (types.h BvhNode8) and the tracer that walks them (k_trace8, what a small tile share runs).

The hierarchy is the driver's business in the reference (acceleration.rs:319-345): any tree is legal as long as the hits are the hits.
So: the 8-wide nodes must be a hierarchy over exactly the leaves the 4-wide nodes hold, with boxes that contain them, and every image
must be bit for bit the image of the 4-wide walk (which the other GPU tests hold to the oracle) -- whole frames, tile partitions, chains,
both integrators, alpha-tested geometry, camera updates.
"""
import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scenes import atrium_scene, cube_scene

from conftest import MATTEST

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


@pytest.fixture(scope="module")
def atrium(instance):
    desc = atrium_scene(detail=0.05, texture_size=64, sky_size=(256, 128))
    return desc, glaze_amd.RayTraceScene.from_desc(instance, desc)


def _walk(nodes, width, first_link):
    """(leaf slot -> box as (lo, hi) grid coordinates of the tightest child box naming it), inner-node count, max children"""
    leaves = {}
    seen = set()
    stack = [0]
    while stack:
        n = stack.pop()
        assert n not in seen, "node %d reached twice" % n
        seen.add(n)
        w = nodes[n]
        for k in range(width):
            link = int(np.int32(w[first_link + k]))
            if link == 0x7FFFFFFF:
                assert w[3 * k] & 0xFFFF == 32767 and w[3 * k] >> 16 == 0, "an unused slot carries an inverted box"
                continue
            lo = np.array([w[3 * k + a] & 0xFFFF for a in range(3)])
            hi = np.array([w[3 * k + a] >> 16 for a in range(3)])
            assert (lo <= hi).all()
            if link < 0:
                assert ~link not in leaves, "leaf %d linked twice" % ~link
                leaves[~link] = (lo, hi)
            else:
                stack.append(link)
    return leaves, len(seen)


@pytest.mark.parametrize("which", ["cube", "atrium", "mattest"])
def test_wide_nodes_hold_the_same_leaves_in_boxes_that_contain_them(which, instance, atrium):
    scene = {"cube": lambda: glaze_amd.RayTraceScene.from_desc(instance, cube_scene()), "atrium": lambda: atrium[1],
             "mattest": lambda: glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))}[which]()
    info = scene.info()
    n4, tris = scene.debug_bvh()
    n8 = scene.debug_bvh8()
    assert info.bvh_nodes8 == n8.shape[0] and 0 < n8.shape[0] <= n4.shape[0]
    leaves4, count4 = _walk(n4, 4, 12)
    leaves8, count8 = _walk(n8, 8, 24)
    assert count4 == n4.shape[0] and count8 == n8.shape[0], "every node is reachable from the root"
    assert set(leaves4) == set(leaves8), "the two collapses hold the same leaves"
    # a leaf's box is the same box in both (it is the leaf's own, quantised by the same rule) and contains the leaf's triangles
    lo_g, cell = np.array(info.bvh_grid_lo[:]), np.array(info.bvh_grid_cell[:])
    flags = tris.view(np.uint32)[:, 11]
    for slot in list(leaves8)[:: max(1, len(leaves8) // 2000)]:
        lo4, hi4 = leaves4[slot]
        lo8, hi8 = leaves8[slot]
        assert (lo4 == lo8).all() and (hi4 == hi8).all()
        count = 2 if flags[slot] & 0x40000000 else 1
        v = tris[slot:slot + count].reshape(-1, 4)[:, :3].astype(np.float64)
        g = (v - lo_g) / cell
        assert (g >= lo8 - 1e-3).all() and (g <= hi8 + 1e-3).all()
    # inner links: a child's box contains the boxes of everything below it
    def subtree_box(n):
        lo, hi = np.full(3, 1 << 20), np.zeros(3, np.int64)
        for k in range(8):
            link = int(np.int32(n8[n][24 + k]))
            if link == 0x7FFFFFFF:
                continue
            blo = np.array([n8[n][3 * k + a] & 0xFFFF for a in range(3)]); bhi = np.array([n8[n][3 * k + a] >> 16 for a in range(3)])
            if link >= 0:
                slo, shi = subtree_box(link)
                assert (blo <= slo).all() and (bhi >= shi).all(), "child %d of node %d does not contain its subtree" % (k, n)
            lo, hi = np.minimum(lo, blo), np.maximum(hi, bhi)
        return lo, hi
    import sys
    sys.setrecursionlimit(10000)
    subtree_box(0)
    assert count8 < count4 or count4 <= 2, "eight wide needs fewer nodes"


def _render(r, width, launches, **kw):
    r.set_node_width(width)
    for k, v in kw.items():
        getattr(r, "set_" + k)(v)
    r.restart()
    r.step(launches)
    return r.read_hdr(), r.read_result()


@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_cube_images_do_not_depend_on_the_node_width(integrator, instance):
    for mtype in (abi.MAT_LAMBERT, abi.MAT_GLASS, abi.MAT_UBER):
        desc = cube_scene(material_type=mtype)
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 200, 136)
        r.set_launch_mode("two_kernels")
        r.set_depth(6)
        r.set_integrator(glaze_amd.Integrator.DIRECT if integrator == "direct" else glaze_amd.Integrator.PATH_TRACE)
        a = _render(r, 4, 21)
        assert r.node_width() == 4
        b = _render(r, 8, 21)
        assert r.node_width() == 8
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1]))


@pytest.mark.parametrize("partition", [(0, 1), (3, 8), (1, 3)])
@pytest.mark.parametrize("chains", [1, 3])
def test_atrium_images_do_not_depend_on_the_node_width(partition, chains, instance, atrium):
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, atrium[0]), 640, 360)   # (a renderer owns its scene)
    r.set_launch_mode("two_kernels")
    r.set_depth(8)
    r.set_partition(*partition)
    r.set_chains(chains)
    a = _render(r, 4, 27)
    b = _render(r, 8, 27)
    assert r.node_width() == 8
    assert np.isfinite(a[0]).all() and a[0][..., :3].max() > 0
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1]))


def test_mattest_with_alpha_tested_and_every_bsdf(instance):
    """mattest.glaze through the file reader: every BSDF family by override, and its materials as they are"""
    scene = glaze_amd.RayTraceScene.new(instance, glaze_amd.parse(MATTEST))
    r = glaze_amd.RayTraceRenderer.new(instance, scene, 256, 256)
    r.set_launch_mode("two_kernels")
    r.set_depth(8)
    a = _render(r, 4, 33)
    b = _render(r, 8, 33)
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1]))


def test_the_width_in_force(instance, atrium):
    """automatic is the 4-wide walk (the 8-wide one measured slower at every share); on request 8, except while the work counters run"""
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, atrium[0]), 512, 512)
    r.set_launch_mode("two_kernels")
    assert r.node_width() == 4
    r.set_node_width(8)
    assert r.node_width() == 8
    r.enable_counters(True, False)
    assert r.node_width() == 4
    r.step(3)
    assert r.stats().closest_nodes > 0
    r.enable_counters(False, False)
    assert r.node_width() == 8
    r.set_node_width(0)
    assert r.node_width() == 4
    with pytest.raises(Exception):
        r.set_node_width(5)


def test_camera_update_and_restart_under_the_wide_walk(instance, atrium):
    from glaze_amd.scene_desc import make_camera
    images = []
    for w in (4, 8):
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, atrium[0]), 320, 200)
        r.set_launch_mode("two_kernels")
        r.set_node_width(w)
        r.set_depth(5)
        r.step(7)
        r.update_camera(make_camera(position=(-10.0, 3.0, 1.0), target=(15.0, 5.0, -2.0), up=(0, 1, 0), fovx=1.1, near=1e-2, far=200.0))
        r.step(11)
        assert r.node_width() == w
        images.append(r.read_hdr())
    assert images[0][..., 3].max() == 11.0
    assert np.array_equal(_bits(images[0]), _bits(images[1]))
