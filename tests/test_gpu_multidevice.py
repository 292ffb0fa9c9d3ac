"""Several GPUs inside ONE process behind the C ABI (glz_renderer_set_devices) and first contact with RCCL.

The lease is one GPU, so: (1) the sharding / replica / host-thread / forwarding machinery runs in loop-back mode
(GLAZE_MULTI_LOOPBACK=1: the device list names the one GPU n times; the tiles then meet without RCCL, which cannot put two
ranks on one device) and must reproduce the one-device image bit for bit through every kind of update; (2) RCCL itself is
exercised with one-rank communicators: from C (ncclCommInitAll + ncclReduce on the instance stream) and through
torch.distributed's "nccl" backend with world size 1 (init, reduce of the 33 MB 1080p frame, barrier, ordering between the
instance stream and torch's stream) -- the code bench.py --gpus N runs, so the scaling run is not RCCL's first execution.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import glaze_amd
from glaze_amd import abi
from glaze_amd.scene_desc import make_camera, make_light
from glaze_amd.scenes import cube_scene

from conftest import MATTEST, ROOT
from helpers import desc_from_oracle_parse

pytestmark = pytest.mark.gpu


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


@pytest.fixture()
def loopback(monkeypatch):
    monkeypatch.setenv("GLAZE_MULTI_LOOPBACK", "1")


def test_rccl_selftest_one_rank_communicator(instance):
    v = C.c_int(0)
    abi.check(abi.lib().glz_debug_rccl_selftest(instance._h, 1920 * 1080 * 4, C.byref(v)))     # the 33 MB frame of the bench
    assert v.value >= 20000                                                                    # ncclGetVersion(): 2.x.y -> 2xxyy
    abi.check(abi.lib().glz_debug_rccl_selftest(instance._h, 7, None))                         # a ragged count


def test_set_devices_argument_checks(instance):
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, cube_scene()), 64, 64)
    dev = instance.device
    r.set_devices([dev])                                                                       # one device: nothing changes
    r.set_depth(2)
    r.step(3)
    a = r.read_hdr()
    for bad in ([], [dev + 1], [dev, 4096], [dev, dev]):                                       # empty, not ours first, out of range, twice without loop-back
        with pytest.raises(abi.GlazeError):
            r.set_devices(bad)
    r.restart()
    r.step(3)
    assert np.array_equal(bits(r.read_hdr()), bits(a))                                         # rejected lists leave the renderer as it was
    r.set_partition(0, 2)
    with pytest.raises(abi.GlazeError):
        os.environ["GLAZE_MULTI_LOOPBACK"] = "1"
        try:
            r.set_devices([dev, dev])                                                          # already one rank of a process partition
        finally:
            del os.environ["GLAZE_MULTI_LOOPBACK"]


@pytest.mark.parametrize("n", [2, 3, 8])
def test_loopback_devices_reproduce_the_one_device_image(instance, loopback, n):
    """n renderers (own stream, scene replica, host thread each) for the tiles t % n == i; every read-back gathers them."""
    desc = desc_from_oracle_parse(MATTEST)
    w, h = 200, 136                                                                            # 4 x 3 tiles, ragged edges; fewer tiles than 8 x 2 chains
    one = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    one.set_depth(5)
    one.set_seed(4)
    img1 = one.draw(3)
    hdr1, res1 = one.read_hdr(), one.read_result()
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    r.set_depth(5)
    r.set_seed(4)
    r.set_devices([instance.device] * n)
    ticks = []
    img = r.draw(3, callback=lambda: ticks.append(1))
    assert len(ticks) == 3                                                                     # once per sample, on the caller's thread
    assert np.array_equal(bits(r.read_hdr()), bits(hdr1)) and np.array_equal(bits(r.read_result()), bits(res1))
    assert np.array_equal(img, img1)
    s, s1 = r.stats(), one.stats()
    assert s.launches == s1.launches == 15 and s.samples == s1.samples == w * h * 15           # every pixel belongs to exactly one device
    # progressive stepping, exposure without restart, and back to one device
    r.restart(); one.restart()
    for k in (4, 1, 6):
        r.step(k); one.step(k)
    r.set_exposure(0.5); one.set_exposure(0.5)
    r.step(3); one.step(3)
    assert np.array_equal(bits(r.read_hdr()), bits(one.read_hdr())) and np.array_equal(bits(r.read_result()), bits(one.read_result()))
    r.set_devices([instance.device])
    r.restart(); one.restart()
    r.step(7); one.step(7)
    assert np.array_equal(bits(r.read_hdr()), bits(one.read_hdr()))


def test_loopback_devices_follow_every_update(instance, loopback):
    """update_camera, change_resolution, set_integrator, update_materials_and_lights (+ textures), refresh_binded_textures,
    change_scene, set_chains and the traversal counters reach every device."""
    desc = cube_scene(material_type=abi.MAT_UBER)
    desc.lights.append(make_light(abi.LIGHT_SKY, "sky", resource_id=1, intensity=0.3, yaw=20, pitch=75, roll=10))
    one = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 136, 200)
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), 136, 200)
    r.set_devices([instance.device] * 3)

    def same(tag, launches=5):
        r.step(launches); one.step(launches)
        assert np.array_equal(bits(r.read_hdr()), bits(one.read_hdr())), tag
        assert np.array_equal(r.read_rgba8(), one.read_rgba8()), tag

    for x in (r, one):
        x.set_depth(4)
        x.set_seed(21)
    same("initial")
    cam = make_camera(position=(0.2, -0.1, -0.6), target=(0.0, 0.1, 5.0), orthographic=True, scale=0.7, near=1e-3, far=50.0)
    for x in (r, one):
        x.update_camera(cam)
    same("update_camera")
    for x in (r, one):
        x.change_resolution(150, 70)
    same("change_resolution")
    for x in (r, one):
        x.set_integrator(glaze_amd.Integrator.DIRECT)
    same("direct integrator", 3)
    for x in (r, one):
        x.set_integrator(glaze_amd.Integrator.PATH_TRACE)
    rng = np.random.default_rng(1)
    tex = [desc.textures[0], (abi.TEX_RGBA_SRGB, rng.integers(0, 256, (32, 48, 4), dtype=np.uint8), "noise")]
    mats = list(desc.materials)
    mats[2].mtype = abi.MAT_METAL
    for x in (r, one):
        x.update_materials_and_lights(mats, desc.lights, tex)
    same("update_materials_and_lights + textures")
    tex2 = [desc.textures[0], (abi.TEX_RGBA_SRGB, rng.integers(0, 256, (16, 16, 4), dtype=np.uint8), "noise2")]
    for x in (r, one):
        x.refresh_binded_textures(tex2)
    same("refresh_binded_textures (accumulation continues)")
    mat_desc = desc_from_oracle_parse(MATTEST)
    for x in (r, one):
        x.change_scene(glaze_amd.RayTraceScene.from_desc(instance, mat_desc))
    same("change_scene")
    r.set_chains(2)
    r.restart(); one.restart()
    same("two chains per device", 6)
    for x in (r, one):
        x.enable_counters(True, True)
        x.restart()
        x.step(4)
        x.wait_idle()
    s, s1 = r.stats(), one.stats()
    for f in ("closest_rays", "shadow_rays", "closest_nodes", "closest_tris", "shadow_nodes", "shadow_tris", "hits", "fresh_paths"):
        assert getattr(s, f) == getattr(s1, f), f                                              # the work counters add up over the devices
    with pytest.raises(abi.GlazeError):
        r.set_partition(0, 2)                                                                  # not while it spans several devices


def test_cli_devices_flag(tmp_path, instance):
    """glaze-cli --devices: one device = the default path; a loop-back list of three = the same image."""
    cli = os.path.join(ROOT, "glaze_amd", "csrc", "glaze-cli")
    if not os.path.exists(cli):
        pytest.skip("glaze-cli is not built")
    d = str(instance.device)
    outs = {}
    for tag, extra, env in (("plain", [], {}), ("one", ["--devices", d], {}), ("three", ["--devices", ",".join([d] * 3)], {"GLAZE_MULTI_LOOPBACK": "1"})):
        png = str(tmp_path / (tag + ".png"))
        p = subprocess.run([cli, MATTEST, png, "-r", "200x136", "-s", "2", "--seed", "3", "--depth", "4", "--report"] + extra,
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert p.returncode == 0 and "All done :)" in p.stderr, p.stderr
        rep = json.loads(p.stdout.strip().splitlines()[-1])
        assert rep["launches"] == 8 and rep["devices"] == (3 if tag == "three" else 1)
        from PIL import Image
        outs[tag] = np.asarray(Image.open(png)).copy()
    assert np.array_equal(outs["plain"], outs["one"]) and np.array_equal(outs["plain"], outs["three"])
    p = subprocess.run([cli, MATTEST, str(tmp_path / "x.png"), "--devices", d + "," + d], capture_output=True, text=True)
    assert p.returncode == 1 and "listed twice" in p.stderr                                    # without the loop-back switch
    p = subprocess.run([cli, MATTEST, str(tmp_path / "x.png"), "--devices", "0,x"], capture_output=True, text=True)
    assert p.returncode == 2


NCCL_WORLD1 = r'''
import os, sys, json
import numpy as np
import torch                      # before libglaze_hip.so: the process then carries ONE HIP runtime (torch's)
import torch.distributed as dist
sys.path.insert(0, os.environ["GLAZE_ROOT"])
import glaze_amd
from glaze_amd.distributed import reduce_frame
from glaze_amd.scenes import cube_scene
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%s" % os.environ["GLAZE_PORT"], rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
inst = glaze_amd.RayTraceInstance.new(0)
W, H = 1920, 1080
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, cube_scene()), W, H)
r.set_depth(2)
r.set_partition(0, 1)
r.step(4)
frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
r.export_device(0, frame.data_ptr())          # instance stream, synchronised before it returns
reduce_frame(frame, force=True)               # ncclReduce(sum) on torch's stream: a one-rank communicator, 33 MB
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)      # the timing reduction of bench.py
same = bool(np.array_equal(frame.cpu().numpy().view(np.uint32), r.read_hdr().view(np.uint32)))
r.step(3)                                     # rendering continues after the collective (stream ordering both ways)
r.export_device(0, frame.data_ptr())
reduce_frame(frame, force=True)
torch.cuda.synchronize()
same2 = bool(np.array_equal(frame.cpu().numpy().view(np.uint32), r.read_hdr().view(np.uint32)))
dist.destroy_process_group()
print(json.dumps({"same": same, "same_after_more_steps": same2, "t": float(t.item()), "backend": "nccl"}))
'''


def test_torch_nccl_backend_world_size_one():
    """bench.py's N > 1 path on the real "nccl" (= RCCL) backend with a one-rank group, in its own process (torch first)."""
    env = dict(os.environ, GLAZE_ROOT=ROOT, GLAZE_PORT=str(29500 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", NCCL_WORLD1], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out == {"same": True, "same_after_more_steps": True, "t": 1.5, "backend": "nccl"}
