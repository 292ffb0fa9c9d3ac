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
from helpers import DeviceArray, desc_from_oracle_parse

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
    r.set_launch_mode("two_kernels")                                                           # chains belong to this mode; frames this small run as k_path otherwise
    r.set_chains(2)
    r.restart(); one.restart()
    same("two chains per device", 6)
    r.set_launch_mode("path")                                                                  # (a change of mode starts a new frame, like set_chains)
    r.restart(); one.restart()
    same("k_path on every device", 6)
    r.set_launch_mode("auto")
    r.restart(); one.restart()
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
from glaze_amd.distributed import gather_frame, reduce_frame
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
# the packed-tile exchange bench.py uses by default: export_packed -> dist.gather ("nccl": send / receive pairs) -> scatter on rank 0
frame.fill_(-3.0)
gather_frame(r, frame)
torch.cuda.synchronize()
same3 = bool(np.array_equal(frame.cpu().numpy().view(np.uint32), r.read_hdr().view(np.uint32)))
dist.destroy_process_group()
print(json.dumps({"same": same, "same_after_more_steps": same2, "same_gathered": same3, "t": float(t.item()), "backend": "nccl"}))
'''


def test_torch_nccl_backend_world_size_one():
    """bench.py's N > 1 path on the real "nccl" (= RCCL) backend with a one-rank group, in its own process (torch first)."""
    env = dict(os.environ, GLAZE_ROOT=ROOT, GLAZE_PORT=str(29500 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", NCCL_WORLD1], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out == {"same": True, "same_after_more_steps": True, "same_gathered": True, "t": 1.5, "backend": "nccl"}


def test_replicas_take_the_shape_of_the_root_scene(instance, loopback):
    """A shape forced on the root (two levels for memory, or flattened) is what every peer builds, whatever AUTO would pick."""
    desc = desc_from_oracle_parse(MATTEST)
    for mode, want in (("two_level", 2), ("flat", 1)):
        instance.set_as_levels(mode)
        try:
            scene = glaze_amd.RayTraceScene.from_desc(instance, desc)
        finally:
            instance.set_as_levels("auto")                     # the peers must not depend on the option still being set
        assert scene.info().as_levels == want
        r = glaze_amd.RayTraceRenderer.new(instance, scene, 136, 72)
        r.set_devices([instance.device] * 3)
        assert [r.device_scene_info(i).as_levels for i in range(3)] == [want] * 3
        with pytest.raises(abi.GlazeError):
            r.device_scene_info(3)
        # ... also after change_scene to a scene of the other shape
        instance.set_as_levels("flat" if want == 2 else "two_level")
        try:
            other = glaze_amd.RayTraceScene.from_desc(instance, desc)
        finally:
            instance.set_as_levels("auto")
        r.change_scene(other)
        assert [r.device_scene_info(i).as_levels for i in range(3)] == [3 - want] * 3


def test_packed_tiles_export_and_scatter(instance):
    """The 1/world-sized exchange of a one-process-per-GPU job: every rank exports its tiles only, rank 0 scatters them; equals
    the sum of the zero-padded frames (three sequential partitions on the one GPU, with one and with several chains)."""
    desc = cube_scene(material_type=abi.MAT_UBER)
    w, h, world = 200, 136, 3
    one = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    one.set_depth(3)
    one.step(5)
    want = one.read_hdr()
    for chains in (1, 2):
        frame = DeviceArray((h, w, 4))
        for rank in range(world):
            r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
            r.set_depth(3)
            r.set_partition(rank, world)
            r.set_launch_mode("two_kernels" if chains > 1 else "auto")                        # chains are the two-kernel mode's
            r.set_chains(chains)
            r.step(5)
            n = r.packed_pixels(rank, world)
            assert n == len(range(rank, 4 * 3, world)) * 4096
            if rank == 0:
                r.export_device(0, frame.ptr)
            else:
                packed = DeviceArray((r.packed_pixels(0, world) + 4096, 4), fill=-7.0)      # room beyond its own tiles
                r.export_packed(0, packed.ptr)
                assert float(packed.numpy()[n:].max()) == -7.0                              # nothing past its own tiles is written
                one.scatter_packed(rank, world, packed.ptr, frame.ptr)
        assert np.array_equal(bits(frame.numpy()), bits(want)), chains
    with pytest.raises(abi.GlazeError):
        one.scatter_packed(3, 3, frame.ptr, frame.ptr)
    # the receiving side of a gather in one call (what glaze_amd.distributed.gather_frame does on rank 0): every rank's part in ONE buffer,
    # n_max pixels apart, rank 0's own among them -- the frame needs nothing else (it starts as garbage here)
    n_max = one.packed_pixels(0, world)
    whole = np.full((world, n_max, 4), -7.0, np.float32)
    for rank in range(world):
        r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
        r.set_depth(3)
        r.set_partition(rank, world)
        r.step(5)
        packed = DeviceArray((n_max, 4), fill=-7.0)
        r.export_packed(0, packed.ptr)
        whole[rank] = packed.numpy()
    gathered = DeviceArray((world, n_max, 4))
    gathered.upload(whole)
    frame = DeviceArray((h, w, 4), fill=123.0)
    one.scatter_packed_all(world, gathered.ptr, n_max, frame.ptr)
    assert np.array_equal(bits(frame.numpy()), bits(want))
    with pytest.raises(abi.GlazeError):
        one.scatter_packed_all(world, gathered.ptr, n_max - 1, frame.ptr)


FAKE_RCCL = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["GLAZE_ROOT"])
import glaze_amd
from glaze_amd import abi
from glaze_amd.scenes import cube_scene
n = int(os.environ["GLAZE_TEST_N"])
inst = glaze_amd.RayTraceInstance.new(0)
w, h = 520, 200                                     # 9 x 4 = 36 tiles: every one of 8 devices owns some, ragged edges
desc = cube_scene(material_type=abi.MAT_UBER)
def fresh():
    r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, desc), w, h)
    r.set_depth(3); r.set_seed(9)
    return r
one = fresh()
one.step(7)
want_hdr, want_res = one.read_hdr(), one.read_result()
out = {"version": int(abi.lib().glz_rccl_version())}
r = fresh()
try:
    r.set_devices([0] * n)
    out["set_devices"] = "ok"
except abi.GlazeError as e:
    out["set_devices"] = str(e)
out["devices"] = r.device_count()
chains = int(os.environ.get("GLAZE_TEST_CHAINS", "0"))
if chains:
    r.set_chains(chains)
r.step(7)
def read(fn, want):
    try:
        return bool(np.array_equal(np.nan_to_num(fn(), nan=-1).view(np.uint32), np.nan_to_num(want, nan=-1).view(np.uint32)))
    except abi.GlazeError as e:
        return str(e)
out["hdr"] = read(r.read_hdr, want_hdr)
out["hdr_again"] = read(r.read_hdr, want_hdr)       # after a failed exchange the next one works: nothing is left half done
out["result"] = read(r.read_result, want_res)
r.step(2); one.step(2)                              # rendering continues after an exchange
out["hdr_after_more"] = read(r.read_hdr, one.read_hdr())
del r
import gc; gc.collect()
print(json.dumps(out))
'''


def run_fake(tmp_path, n, exchange="gather", fail=None, chains=0):
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        pytest.skip("tests/fake_rccl/libfake_rccl.so is not built (__graft_entry__.build())")
    log = str(tmp_path / ("rccl_%s_%d_%s_%d.log" % (exchange, n, fail or "ok", chains)).replace(":", "_"))
    env = dict(os.environ, GLAZE_ROOT=ROOT, GLAZE_RCCL_LIBRARY=lib, GLAZE_MULTI_LOOPBACK="rccl", GLAZE_MULTI_EXCHANGE=exchange,
               GLAZE_FAKE_RCCL_LOG=log, GLAZE_TEST_N=str(n), GLAZE_TEST_CHAINS=str(chains))
    if fail:
        env["GLAZE_FAKE_RCCL_FAIL"] = fail
    p = subprocess.run([sys.executable, "-c", FAKE_RCCL], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    calls = [json.loads(l) for l in open(log)] if os.path.exists(log) else []
    return json.loads(p.stdout.strip().splitlines()[-1]), calls


def exchanges_of(calls):
    """the calls between each outermost ncclGroupStart and its ncclGroupEnd"""
    groups, cur = [], None
    for c in calls:
        if c["fn"] == "ncclGroupStart":
            cur = []
        elif c["fn"] == "ncclGroupEnd":
            groups.append((cur, c["result"]))
            cur = None
        elif cur is not None:
            cur.append(c)
        else:
            assert c["fn"] in ("ncclCommInitAll", "ncclCommDestroy"), "a transfer outside a group: %r" % (c,)
    assert cur is None, "a group was left open"
    return groups


@pytest.mark.parametrize("n,chains", [(2, 0), (8, 0), (3, 2)])
def test_group_construction_packed_gather(tmp_path, n, chains):
    """The n >= 2 exchange against the recording stand-in for librccl (tests/fake_rccl): n "ranks" on the one GPU, real
    communicator / group / stream semantics.  Default shape: every peer SENDS its packed tiles once, device 0 receives."""
    out, calls = run_fake(tmp_path, n, chains=chains)
    assert out == {"version": 99999, "set_devices": "ok", "devices": n, "hdr": True, "hdr_again": True, "result": True, "hdr_after_more": True}
    init = [c for c in calls if c["fn"] == "ncclCommInitAll"]
    assert len(init) == 1 and init[0]["count"] == n
    groups = exchanges_of(calls)
    assert len(groups) == 4 and all(res == 0 for _, res in groups)                               # four read-backs, four groups
    for ops, _ in groups:
        sends = [c for c in ops if c["fn"] == "ncclSend"]
        recvs = [c for c in ops if c["fn"] == "ncclRecv"]
        assert len(sends) == len(recvs) == n - 1 and len(ops) == 2 * (n - 1)
        assert sorted(c["rank"] for c in sends) == list(range(1, n)) and all(c["peer"] == 0 for c in sends)     # one send per peer, to device 0
        assert all(c["rank"] == 0 for c in recvs) and sorted(c["peer"] for c in recvs) == list(range(1, n))
        assert len({c["stream"] for c in sends}) == n - 1 and len({c["stream"] for c in recvs}) == 1            # each peer's own stream; the root's
        assert not ({c["stream"] for c in sends} & {c["stream"] for c in recvs})
        by_peer = {c["rank"]: c["count"] for c in sends}
        assert all(by_peer[c["peer"]] == c["count"] for c in recvs)
        assert sum(by_peer.values()) == sum(len(range(i, 36, n)) for i in range(1, n)) * 4096 * 4                # 1/n-sized: the peers' tiles only
        spans = sorted((c["recv"], c["count"] * 4) for c in recvs)
        assert all(a + b <= c for (a, b), (c, _) in zip(spans, spans[1:]))                                       # disjoint staging areas
    destroyed = [c for c in calls if c["fn"] == "ncclCommDestroy"]
    assert len(destroyed) == n and all(c["result"] == 0 for c in destroyed) and sorted(c["rank"] for c in destroyed) == list(range(n))
    assert calls.index(destroyed[0]) > max(i for i, c in enumerate(calls) if c["fn"] == "ncclGroupEnd")         # after the last exchange


def test_group_construction_reduce(tmp_path):
    """GLAZE_MULTI_EXCHANGE=reduce: one ncclReduce(sum) per device in one group, in place on the root, no receive buffer on the peers."""
    n = 8
    out, calls = run_fake(tmp_path, n, exchange="reduce")
    assert out == {"version": 99999, "set_devices": "ok", "devices": n, "hdr": True, "hdr_again": True, "result": True, "hdr_after_more": True}
    groups = exchanges_of(calls)
    assert len(groups) == 4
    for ops, res in groups:
        assert res == 0 and [c["fn"] for c in ops] == ["ncclReduce"] * n
        assert sorted(c["rank"] for c in ops) == list(range(n)) and all(c["peer"] == 0 for c in ops)            # root 0
        assert all(c["count"] == 520 * 200 * 4 for c in ops)
        root = [c for c in ops if c["rank"] == 0][0]
        assert root["send"] == root["recv"] != 0                                                                 # in place on the root
        assert all(c["recv"] == 0 and c["send"] not in (0, root["send"]) for c in ops if c["rank"] != 0)
        assert len({c["stream"] for c in ops}) == n and len({c["send"] for c in ops}) == n
    assert len([c for c in calls if c["fn"] == "ncclCommDestroy"]) == n


@pytest.mark.parametrize("exchange,fail", [("gather", "ncclCommInitAll:1"), ("gather", "ncclSend:3"), ("gather", "ncclRecv:1"),
                                           ("gather", "ncclGroupEnd:1"), ("gather", "ncclGroupStart:1"), ("reduce", "ncclReduce:2")])
def test_rccl_failures_leave_a_usable_renderer(tmp_path, exchange, fail):
    """An RCCL call that fails: the error reaches the caller, an opened group is always closed, the communicators are destroyed
    exactly once, and the renderer keeps working (one device after a failed set_devices; the next exchange after a failed one)."""
    n = 4
    out, calls = run_fake(tmp_path, n, exchange=exchange, fail=fail)
    groups = exchanges_of(calls)                                                                                 # asserts no group is left open
    created = sum(c["count"] for c in calls if c["fn"] == "ncclCommInitAll" and c["result"] == 0)
    destroyed = [c for c in calls if c["fn"] == "ncclCommDestroy"]
    assert len(destroyed) == created and all(c["result"] == 0 for c in destroyed)
    if fail.startswith("ncclCommInitAll"):
        assert "ncclCommInitAll" in out["set_devices"] and out["devices"] == 1 and created == 0
        assert out["hdr"] is True and out["result"] is True and out["hdr_after_more"] is True                    # one device renders the whole frame
        assert groups == []
    else:
        assert out["set_devices"] == "ok" and out["devices"] == n
        assert isinstance(out["hdr"], str) and fail.split(":")[0] in out["hdr"]                                  # the first exchange reports the failure
        assert out["hdr_again"] is True and out["result"] is True and out["hdr_after_more"] is True


@pytest.mark.parametrize("n,chains", [(2, 0), (3, 2), (8, 0)])
def test_peer_copy_exchange_reproduces_the_one_device_image(instance, monkeypatch, n, chains):
    """GLAZE_MULTI_EXCHANGE=peer: the packed tiles travel by hipMemcpyPeerAsync on the peers' streams, ordered into device 0's stream by
    events -- no RCCL anywhere.  n "devices" on the one GPU (GLAZE_MULTI_LOOPBACK=peer keeps the copies and the staging area)."""
    monkeypatch.setenv("GLAZE_MULTI_LOOPBACK", "peer")
    monkeypatch.setenv("GLAZE_RCCL_LIBRARY", "/nonexistent/librccl.so")                       # proves that nothing asks for it
    desc = cube_scene(material_type=abi.MAT_UBER)
    w, h = 520, 200
    one = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    r = glaze_amd.RayTraceRenderer.new(instance, glaze_amd.RayTraceScene.from_desc(instance, desc), w, h)
    for x in (one, r):
        x.set_depth(3)
        x.set_seed(9)
    r.set_devices([instance.device] * n)
    if chains:
        r.set_chains(chains)
    assert r.device_count() == n
    for k in (7, 2):
        one.step(k)
        r.step(k)
        assert np.array_equal(bits(r.read_hdr()), bits(one.read_hdr())) and np.array_equal(bits(r.read_result()), bits(one.read_result()))
        assert np.array_equal(r.read_rgba8(), one.read_rgba8())
    for x in (one, r):
        x.change_resolution(200, 136)                                                          # fewer tiles than devices x chains; staging areas shrink / stay
        x.step(4)
    assert np.array_equal(bits(r.read_hdr()), bits(one.read_hdr()))
    r.set_devices([instance.device])
    r.restart(); one.restart()
    r.step(3); one.step(3)
    assert np.array_equal(bits(r.read_hdr()), bits(one.read_hdr()))


BENCH_LOOPBACK_ENV = {"GLAZE_MULTI_LOOPBACK": "1"}


def instance_count():
    n = 0
    while glaze_amd.RayTraceInstance.new(n) is not None:
        n += 1
    return n


def test_bench_in_process_multi_gpu(tmp_path):
    """`python bench.py --gpus N` without a launcher: the in-process set_devices path (loop-back on the one GPU), verified bit for bit;
    a plain --gpus 2 on a one-GPU box fails with a clear message instead of hanging."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2", "--width", "640", "--height", "360", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, **BENCH_LOOPBACK_ENV), timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 8 and out["multi_gpu"]["gpus_seen"] == 8 and out["multi_gpu"]["measurement"] is False
    assert out["verify"]["bit_identical_to_one_gpu"] is True and out["verify"]["launches"] == 2 + 4 * len(out["regions_ms"])
    assert out["multi_gpu"]["exchange_ms"] > 0 and out["value"] > 0
    # BASELINE configs[4] (3840 x 2160, depth 12) is timed next to the 1080p frame in the same invocation, under its own name
    c5 = out["multi_gpu"]["config5"]
    assert c5["value"] > 0 and c5["ms_per_step"] > 0 and "3840x2160" in c5["workload"] and "depth 12" in c5["workload"] and len(c5["regions_ms"]) == 3
    if instance_count() < 2:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode != 0 and "this machine has 1 GPU" in p.stderr


def test_bench_falls_back_to_peer_copies_when_rccl_will_not_start(tmp_path):
    """ncclCommInitAll fails (injected into the stand-in library): bench.py says why on stderr, switches to the peer-copy exchange, still
    verifies the frame bit for bit and names the exchange and the reason in its line."""
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        pytest.skip("tests/fake_rccl/libfake_rccl.so is not built (__graft_entry__.build())")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "2", "--width", "640", "--height", "360", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GLAZE_MULTI_EXCHANGE")}
    for fail in ("ncclCommInitAll", "ncclSend"):
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                           env=dict(env, GLAZE_MULTI_LOOPBACK="rccl", GLAZE_RCCL_LIBRARY=lib, GLAZE_FAKE_RCCL_FAIL=fail + ":1", GLAZE_FAKE_RCCL_LOG=str(tmp_path / ("fb_%s.log" % fail))))
        assert p.returncode == 0, p.stderr[-3000:]
        assert "falling back to peer copies" in p.stderr
        out = json.loads(p.stdout.strip().splitlines()[-1])
        assert out["n_gpus"] == 4 and out["multi_gpu"]["gpus_seen"] == 4
        assert "hipMemcpyPeerAsync" in out["multi_gpu"]["exchange"] and fail in out["multi_gpu"]["rccl_fallback"] and out["multi_gpu"]["rccl_version"] is None
        assert out["verify"]["bit_identical_to_one_gpu"] is True


def test_bench_watchdog_ends_a_hung_first_exchange_with_diagnostics(tmp_path):
    """The first contact between ranks goes wrong in the worst way -- ncclGroupEnd never returns (injected into the stand-in library):
    bench.py's watchdog prints multi_gpu.diagnostics (which call, RCCL version, peer-access matrix, bytes per peer) to stderr and ends
    the process with status 3 within its time-out, instead of burning the caller's."""
    import time
    lib = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        pytest.skip("tests/fake_rccl/libfake_rccl.so is not built (__graft_entry__.build())")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "2", "--width", "640", "--height", "360", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GLAZE_MULTI_EXCHANGE")}
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(env, GLAZE_MULTI_LOOPBACK="rccl", GLAZE_RCCL_LIBRARY=lib, GLAZE_FAKE_RCCL_HANG="ncclGroupEnd:1", GLAZE_BENCH_WATCHDOG_S="6",
                                GLAZE_FAKE_RCCL_LOG=str(tmp_path / "hang.log")))
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert time.time() - t0 < 240
    tail = p.stderr.strip().splitlines()[-1]
    diag = json.loads(tail)["multi_gpu"]["diagnostics"]
    assert diag["stuck_in"].startswith("exchange #1") and diag["timeout_s"] == 6.0 and diag["elapsed_s"] >= 6.0
    assert diag["peers"] == 3 and diag["bytes_per_peer"] == 640 * 360 * 16 // 4 and diag["devices_in_process"] == 4
    assert isinstance(diag["peer_access"], list) and "rccl_version" in diag and "GLAZE_RCCL_LIBRARY" in diag["env"]
    assert not p.stdout.strip(), "no result line from a run that did not finish"
