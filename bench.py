#!/usr/bin/env python3
"""bench.py -- Msamples/s + achieved HBM GB/s of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]                      (N > 1: ONE process drives the N GPUs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                               (one process per GPU)

Workload (BASELINE.json configs[3]): the Sponza-class frame, 1920x1080, path tracer, depth 8.  Sponza
itself is not in the reference repository, so the scene is the deterministic synthetic
atrium of glaze_amd/scenes.py (262 140 triangles, 25 materials, sun + sky), written as a `.glaze` V1 file
(glz_serialize) and loaded the way glaze-cli loads a scene: parse -> RayTraceScene::new.  Inputs (scene, BVH,
path state) are resident in HBM before the timed region starts.

A "step" is ONE launch of the hot path = one path segment for every pixel of the frame
(one vkCmdTraceRaysKHR in the reference, raytracer.rs:553-562): W*H samples.  Steps continue the
same accumulation (draw_frame semantics), K steps = K/depth samples per pixel.

N > 1: the frame's 64x64 tiles are sharded over the GPUs (tile t -> GPU t % N); after the K steps the float HDR
accumulator of every GPU is brought onto GPU 0 over RCCL / xGMI (inside the timed region; its share of the time is also
printed as `exchange_ms`).  Total work is fixed -> "scaling": "strong".  GPU 0 then checks the assembled frame bit for bit
against the frame it renders alone (--no-verify skips it).  Two ways to get there, same kernels and same partition:
  * started plainly (`python bench.py --gpus N`, no WORLD_SIZE in the environment): the product's own in-process path,
    glz_renderer_set_devices -- what `glaze-cli --devices 0,1,..` runs: one host thread, one scene replica and one RCCL
    communicator (ncclCommInitAll) per GPU, packed tiles sent to GPU 0 (ncclSend / ncclRecv in one group);
  * started by torch.distributed.run: one process per GPU, glz_renderer_set_partition(rank, N), and the same packed-tile
    exchange through torch.distributed's "nccl" backend (= RCCL).
GLAZE_MULTI_EXCHANGE=reduce switches both to one ncclReduce(sum) of the zero-padded W*H*4 frame; =peer (in-process form only) sends
the packed tiles by hipMemcpyPeerAsync instead of RCCL -- also what the in-process form falls back to, saying so, when RCCL cannot be
loaded or its communicators / first exchange fail on this machine.

The timed region is EXACTLY K steps between barrier + synchronize pairs, MAX over ranks.  A region shorter than 0.5 s is
repeated (five regions in all, the accumulation simply continues) and `value` comes from the MEDIAN region; every
region's time is listed under "regions_ms".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# a rank with a small tile share renders with up to three concurrent launch chains (HIP streams) next to torch's and RCCL's own
# streams; the runtime's default of four hardware queues would make some of them share one
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-launches", type=int, default=16, help="launches of the bounded CPU-baseline sample")
    ap.add_argument("--verify", dest="verify", action="store_true", default=True,
                    help="N > 1: rank 0 re-renders the whole frame alone and compares it bit for bit (default)")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of K steps each (0 = five when a region is shorter than 0.5 s, else one)")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure roofline.traffic (N = 1 only)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)     # this process IS one of those passes: render only
    return ap.parse_args()


def measure_hbm_traffic(args):
    """roofline.traffic, measured by THIS invocation: two short runs of this very workload under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE in separate passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes), per-kernel averages over their
    launches.  HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1 KiB: the counters are in KiB and on gfx950 FETCH_SIZE reports half
    of the bytes of wide reads (the 2 x is calibrated for coalesced streams: an upper estimate for scattered 16-byte loads).
    Returns ({kernel: bytes per launch}, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    # not from under a profiler: its preloaded library initialises the GPU in every child before the child's own exec
    if any("rocprof" in v.lower() for v in (os.environ.get("LD_PRELOAD", ""), os.environ.get("ROCP_TOOL_LIBRARIES", ""), os.environ.get("HSA_TOOLS_LIB", ""))):
        return None, "this process runs under a profiler"
    acc = {}
    tmp = tempfile.mkdtemp(prefix="glaze_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--steps", "16", "--warmup", "8", "--width", str(args.width), "--height", str(args.height), "--depth", str(args.depth),
                   "--seed", str(args.seed)]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s did not finish within 150 s" % counter
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s: rc %d: %s" % (counter, p.returncode, (p.stderr or p.stdout)[-200:].replace("\n", " "))
            files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            if not files:
                return None, "rocprofv3 --pmc %s left no counter_collection.csv" % counter
            per = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].replace(" ", "")
                    if row["Counter_Name"] != counter or "<true" in name:
                        continue
                    for k in ("k_trace", "k_shade"):
                        if "::" + k + "<" in name or "::" + k + "(" in name:
                            per.setdefault(k, []).append(float(row["Counter_Value"]))
            for k, v in per.items():
                acc.setdefault(k, {})[counter] = (sum(v) / len(v), len(v))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k, v in acc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            res[k] = {"hbm_bytes_per_launch": int((2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024),
                      "uncorrected": int((v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024), "launches": min(v["FETCH_SIZE"][1], v["WRITE_SIZE"][1])}
    if not res:
        return None, "no k_trace / k_shade rows in the counter files"
    return res, "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (with --kernel-trace only) of this workload, run by this invocation after its timed regions: 24 launches each, (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB per launch"


def algorithmic_bytes(c):
    """Algorithmic bytes per SAMPLE of each kernel, from counted per-sample work.

    node visit = 64 B (one quantised BVH4 node: four child boxes + four links), triangle test = 48 B (BvhTri: the three vertices
    and the ids the tie-break / alpha test need), hit-attribute fetch = 192 B (128-B shading record + the 64 B of RTMaterial
    scalars k_shade loads), path state = 96 B (ray 32 + importance 64), hit record 16 B, accumulator = 32 B r/w + 16 B result,
    shadow-queue entry = 48 B.  Counted too since round 3 (they were the unbooked part of k_shade's traffic): texels (16 B per
    bilinear fetch of an RGBA texture, 4 B per gray one: SURVEY 8(d)'s "16 B/texture tap"), 112 B of RTLight per light sample, and
    per sky-light sample the binary search of the marginal table (4 B a step), one marginal and two conditional values.
    """
    closest = 32 + 16 + 32 * c["f_fresh"] + 64 * c["nodes_closest"] + 48 * c["tris_closest"]
    shade = (16 + 32 + 64 * (1 - c["f_fresh"]) + 192 * c["f_hit"] + 48 * c["f_shadow"] + 48 * (1 - c["f_shadow"])
             + 96 * c["f_hit"])
    shade += c.get("tex_bytes_shade", 0.0) + 112 * (c.get("light_samples", 0.0) + c.get("sky_samples", 0.0)) + c.get("sky_bytes_per_sample", 0.0) * c.get("sky_samples", 0.0)
    shadow = 48 * c["f_shadow"] + 48 * c["f_shadow"] + 64 * c["nodes_shadow"] + 48 * c["tris_shadow"]
    # k_trace traverses the closest-hit rays of a launch and the shadow rays of the launch before it in one kernel
    return {"k_trace": closest + shadow + c.get("tex_bytes_trace", 0.0), "k_shade": shade}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # n_dev > 1: ONE process spans the GPUs (glz_renderer_set_devices); world > 1: one process per GPU (torch.distributed)
    n_dev = 1
    if world == 1 and args.gpus > 1:
        n_dev = args.gpus
    elif world != args.gpus:
        args.gpus = world
    n_gpus = max(world, n_dev)
    import numpy as np
    import torch
    import torch.distributed as dist

    import glaze_amd
    from glaze_amd.distributed import gather_frame, reduce_frame
    from glaze_amd.scenes import atrium_scene

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda is not available (there is no CPU fallback)")
    # GLAZE_BENCH_REHEARSAL=1: every rank shares GPU 0 and the reduce goes through gloo on host tensors -- lets the N > 1
    # code path (partition, chains, reduce, max-over-ranks timing) be run on a one-GPU box.  Not a measurement.
    rehearsal = os.environ.get("GLAZE_BENCH_REHEARSAL") == "1"
    # GLAZE_MULTI_LOOPBACK=1 (in-process path only): the N "devices" are all GPU 0 -- same threads, replicas and tile sharding,
    # the tiles meet without RCCL.  Not a measurement either.
    loopback = n_dev > 1 and os.environ.get("GLAZE_MULTI_LOOPBACK") is not None
    exchange = os.environ.get("GLAZE_MULTI_EXCHANGE", "gather")
    if exchange not in ("gather", "reduce", "peer"):
        raise SystemExit("GLAZE_MULTI_EXCHANGE must be `gather`, `reduce` or `peer`")
    if exchange == "peer" and world > 1:
        raise SystemExit("GLAZE_MULTI_EXCHANGE=peer is the in-process exchange without RCCL (python bench.py --gpus N)")
    if loopback and os.environ.get("GLAZE_MULTI_LOOPBACK") == "peer":
        exchange = "peer"
    if n_dev > 1 and not loopback and torch.cuda.device_count() < n_dev:
        raise SystemExit("--gpus %d: this machine has %d GPU(s) (GLAZE_MULTI_LOOPBACK=1 rehearses the %d-device path on one)"
                         % (n_dev, torch.cuda.device_count(), n_dev))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    inst = glaze_amd.RayTraceInstance.new(local_rank)
    if inst is None:
        raise SystemExit("no gfx950 device for rank %d: %s" % (rank, glaze_amd.abi.last_error()))
    W, H = args.width, args.height
    desc = atrium_scene()
    # the scene goes through the file format end to end: Serializer -> .glaze V1 -> parse -> RayTraceScene::new (cli/src/main.rs:78-91)
    import tempfile
    from glaze_amd.scene_desc import save_scene
    with tempfile.TemporaryDirectory(prefix="glaze_bench_") as tmp:
        path = os.path.join(tmp, "atrium_rank%d.glaze" % rank)
        t0 = time.time()
        save_scene(desc, path)
        serialize_s = time.time() - t0
        glaze_bytes = os.path.getsize(path)
        t0 = time.time()
        scene = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(path))
        setup_s = time.time() - t0
    info = scene.info()
    sky_rows = max([desc.textures[l.resource_id][1].shape[0] for l in desc.lights if l.ltype == glaze_amd.abi.LIGHT_SKY] + [0])
    renderer = glaze_amd.RayTraceRenderer.new(inst, scene, W, H)
    renderer.set_depth(args.depth)
    renderer.set_seed(args.seed)
    devices_s = None
    rccl_fallback = None        # why the RCCL exchange was given up for peer copies, if it was

    def span_devices():
        """replicas + BVH builds + ncclCommInitAll.  RCCL that cannot be loaded or will not initialise on this machine does not
        cost the measurement: the same packed tiles then travel by hipMemcpyPeerAsync (GLAZE_MULTI_EXCHANGE=peer), and the line
        says so (`multi_gpu.exchange`, `multi_gpu.rccl_fallback`)."""
        nonlocal exchange, rccl_fallback
        ids = [0] * n_dev if loopback else list(range(n_dev))
        try:
            renderer.set_devices(ids)
        except glaze_amd.abi.GlazeError as e:
            if exchange == "peer" or (loopback and os.environ.get("GLAZE_MULTI_LOOPBACK") != "rccl"):
                raise
            rccl_fallback = str(e)
            print("bench.py: %s -- falling back to peer copies (GLAZE_MULTI_EXCHANGE=peer)" % rccl_fallback, file=sys.stderr, flush=True)
            exchange = "peer"
            os.environ["GLAZE_MULTI_EXCHANGE"] = "peer"
            if loopback:
                os.environ["GLAZE_MULTI_LOOPBACK"] = "peer"     # (tests: the stand-in RCCL refused; n "devices" on the one GPU)
            renderer.set_devices(ids)

    if world > 1:
        renderer.set_partition(rank, world)
    elif n_dev > 1:
        t0 = time.time()
        span_devices()
        devices_s = time.time() - t0
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")

    def sync_all():
        renderer.wait_idle()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def exchange_to_gpu0():
        """every GPU's tiles onto GPU 0 (RCCL over xGMI; gloo on host tensors in a rehearsal)"""
        if n_dev > 1:
            renderer.export_device(0, frame.data_ptr())     # set_devices: the library's own exchange, synchronised on return
        elif rehearsal:
            renderer.export_device(0, frame.data_ptr())
            host = frame.cpu()
            reduce_frame(host)
            frame.copy_(host)
        elif exchange == "gather":
            gather_frame(renderer, frame)
        else:
            renderer.export_device(0, frame.data_ptr())
            reduce_frame(frame)

    if args.pmc_child:
        # one of the rocprofv3 --pmc passes of measure_hbm_traffic(): the same launches as the timed regions, nothing else
        renderer.restart()
        renderer.step(args.warmup + args.steps)
        renderer.wait_idle()
        return
    # ---- warmup (untimed): same accumulation continues afterwards, like the interactive draw_frame loop
    renderer.restart()
    renderer.step(args.warmup)
    if n_gpus > 1:
        try:
            exchange_to_gpu0()      # first use of the communicators / the process group belongs to the warm-up
        except glaze_amd.abi.GlazeError as e:
            if n_dev == 1 or exchange == "peer" or (loopback and os.environ.get("GLAZE_MULTI_LOOPBACK") != "rccl"):
                raise
            # the communicators came up but the first send / receive did not work: same fallback as in span_devices()
            rccl_fallback = str(e)
            print("bench.py: %s -- falling back to peer copies (GLAZE_MULTI_EXCHANGE=peer)" % rccl_fallback, file=sys.stderr, flush=True)
            exchange = "peer"
            os.environ["GLAZE_MULTI_EXCHANGE"] = "peer"
            if loopback:
                os.environ["GLAZE_MULTI_LOOPBACK"] = "peer"
            renderer.set_devices([0])
            renderer.set_devices([0] * n_dev if loopback else list(range(n_dev)))
            renderer.restart()
            renderer.step(args.warmup)
            exchange_to_gpu0()
    sync_all()
    renderer.stats()            # drains the warmup's kernel events
    s0 = renderer.stats()

    # ---- timed regions: exactly K steps each, barrier + synchronize on both sides, MAX over ranks ----
    def timed_region():
        sync_all()
        t_start = time.perf_counter()
        renderer.step(args.steps)
        if n_gpus > 1:
            exchange_to_gpu0()
        sync_all()
        t = torch.tensor([time.perf_counter() - t_start], dtype=torch.float64, device="cpu" if rehearsal or world == 1 else "cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    regions = [timed_region()]
    # every rank sees the same (max-reduced) first time, so all of them take the same number of regions
    n_regions = args.repeats if args.repeats > 0 else (5 if regions[0] < 0.5 else 1)
    while len(regions) < n_regions:
        regions.append(timed_region())
    s1 = renderer.stats()
    # the exchange alone (it is INSIDE every timed region above; this is only to show its share): nothing new was rendered,
    # so it moves the same frame again
    exchange_ms = None
    if n_gpus > 1:
        ex = []
        for _ in range(3):
            sync_all()
            t_start = time.perf_counter()
            exchange_to_gpu0()
            sync_all()
            ex.append((time.perf_counter() - t_start) * 1e3)
        exchange_ms = sorted(ex)[1]
    # what GPU 0 holds now is the frame of warmup + K * regions launches, assembled from all GPUs: kept for the check at the end
    verify_frame = frame.cpu().numpy().copy() if (n_gpus > 1 and args.verify and rank == 0) else None
    elapsed = sorted(regions)[len(regions) // 2]      # median region
    total_steps = args.steps * len(regions)
    samples = W * H * args.steps                     # whole frame, all ranks together, per region
    value = samples / elapsed / 1e6

    # per-kernel device time inside the timed regions (hipEvents on the instance stream), all regions together
    # (trace_shadow_ms is the stand-alone shadow pass that closes a region: one launch's worth, inside its time)
    # (k_path: the per-wave launch loop a device with a small tile share runs instead of the two -- both kernels' work in one)
    kern_ms = {"k_trace": s1.trace_closest_ms - s0.trace_closest_ms, "k_shade": s1.shade_ms - s0.shade_ms, "k_path": s1.other_ms - s0.other_ms}

    out = None
    if rank == 0:
        # counted per-sample work of the same workload: a separate, untimed pass with the instrumented kernels over the SAME
        # launch indices as the timed window (launches [warmup, warmup + K * regions) after a restart -- the mix of path ages,
        # and with it the work per sample, depends on the launch index), capped at 1024 launches
        renderer.enable_counters(True, True)
        renderer.restart()
        renderer.step(args.warmup)
        renderer.wait_idle()
        sa = renderer.stats()
        n_count = min(total_steps, 1024)
        renderer.step(n_count)
        renderer.wait_idle()
        sc = renderer.stats()
        renderer.enable_counters(False, True)
        rays = max(1, sc.closest_rays - sa.closest_rays)
        counted = {
            "nodes_closest": (sc.closest_nodes - sa.closest_nodes) / rays, "tris_closest": (sc.closest_tris - sa.closest_tris) / rays,
            "nodes_shadow": (sc.shadow_nodes - sa.shadow_nodes) / rays, "tris_shadow": (sc.shadow_tris - sa.shadow_tris) / rays,
            "f_hit": (sc.hits - sa.hits) / rays, "f_shadow": (sc.shadow_rays - sa.shadow_rays) / rays,
            "f_fresh": (sc.fresh_paths - sa.fresh_paths) / rays,
            "tex_fetches": (sc.tex_fetches - sa.tex_fetches) / rays,
            "tex_bytes_shade": ((sc.tex_bytes - sa.tex_bytes) - (sc.alpha_tex_bytes - sa.alpha_tex_bytes)) / rays,
            "tex_bytes_trace": (sc.alpha_tex_bytes - sa.alpha_tex_bytes) / rays,
            "light_samples": (sc.light_samples - sa.light_samples) / rays, "sky_samples": (sc.sky_samples - sa.sky_samples) / rays,
            # per sky-light sample: ceil(log2(H + 1)) steps of the marginal search + one marginal value + two conditional ones
            "sky_bytes_per_sample": 4.0 * (int(np.ceil(np.log2(sky_rows + 1))) + 3) if sky_rows else 0.0,
        }
        bytes_per_sample = algorithmic_bytes(counted)
        bytes_per_sample["k_path"] = bytes_per_sample["k_trace"] + bytes_per_sample["k_shade"]
        dominant = max(kern_ms, key=kern_ms.get)
        owned = W * H / n_gpus      # per GPU: kernel times are per device (the slowest one when one process spans several)
        avg_ms = kern_ms[dominant] / total_steps
        achieved = bytes_per_sample[dominant] * owned / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic from the PMC counters: measured by this invocation (two rocprofv3 --pmc passes of the same workload as child
        # processes, N = 1 only); if that is not possible, the per-launch average of the committed passes (profiles/pmc_summary.json)
        traffic, traffic_source, traffic_all = None, None, None
        if n_gpus == 1 and not args.no_pmc:
            measured, note = measure_hbm_traffic(args)
            if measured is not None and dominant in measured:
                traffic, traffic_source, traffic_all = measured[dominant]["hbm_bytes_per_launch"], note, measured
            else:
                traffic_source = "not measured in this run (%s); " % note
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if traffic is None and os.path.exists(prof) and n_gpus == 1:      # the committed PMC passes profiled the N = 1 command
            try:
                traffic = json.load(open(prof)).get(dominant, {}).get("hbm_bytes_per_launch")
                traffic_source = (traffic_source or "") + "profiles/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py`, not this run)"
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_per_kernel": traffic_all,
                    "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_sample": {k: round(v, 1) for k, v in bytes_per_sample.items()},
                    "whole_job_achieved": round((bytes_per_sample["k_trace"] + bytes_per_sample["k_shade"]) * samples / elapsed / 1e9, 1),
                    "counted_per_sample": {k: round(v, 3) for k, v in counted.items()},
                    "counted_over_launches": [args.warmup, args.warmup + n_count],
                    "kernel_ms_per_step": {k: round(v / total_steps, 4) for k, v in kern_ms.items() if v > 0 or k != "k_path"},
                    "launch_mode": renderer.launch_mode()}
        cpu = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            # CPU baseline: the oracle (scalar C++ restatement; the reference has no CPU tracer) on a
            # bounded sample of the SAME workload: same scene/frame/depth/seed, fewer launches.
            from oracle.pyoracle import OracleRenderer, OracleScene
            # a 1-GPU box grants 16 host cores of the machine's CPUs; never size the pool beyond that share
            cores = max(1, min(len(os.sched_getaffinity(0)), 16))
            o = OracleRenderer(OracleScene(desc), W, H, threads=cores)
            o.set_depth(args.depth)
            o.set_seed(args.seed)
            o.restart()
            tc = time.perf_counter()
            o.step(args.cpu_launches)
            dt = time.perf_counter() - tc
            cpu = {"value": round(W * H * args.cpu_launches / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
                   "sample": "%dx%d atrium, depth %d, %d launches (%.1f s)" % (W, H, args.depth, args.cpu_launches, dt)}
            # the same port on ONE core (SURVEY 8d asks for both): one launch of the same frame
            o.set_threads(1)
            o.restart()
            tc = time.perf_counter()
            o.step(1)
            dt1 = time.perf_counter() - tc
            cpu["single_thread"] = {"value": round(W * H / dt1 / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                    "sample": "1 launch (%.1f s)" % dt1}
        multi = None
        how = ""
        if n_gpus > 1:
            if n_dev > 1:
                mode = "one process, glz_renderer_set_devices (a host thread, a scene replica and an RCCL communicator per GPU)"
                stand_in = loopback and os.environ.get("GLAZE_MULTI_LOOPBACK") == "rccl"      # the RCCL entry points of GLAZE_RCCL_LIBRARY, n "ranks" on one GPU
                how = "loop-back on ONE GPU, no RCCL (rehearsal)" if loopback and exchange != "peer" and not stand_in else {
                    "gather": "ncclSend/ncclRecv of packed tiles in one group", "reduce": "ncclReduce(sum) of the zero-padded frame",
                    "peer": "hipMemcpyPeerAsync of packed tiles, one per peer (no RCCL)"}[exchange]
                rccl = None if loopback or exchange == "peer" else int(glaze_amd.abi.lib().glz_rccl_version())
                seen = renderer.device_count()
            else:
                mode = "one process per GPU (torch.distributed), glz_renderer_set_partition"
                how = "gloo reduce of host tensors on ONE GPU (rehearsal)" if rehearsal else (
                    "dist.gather of packed tiles (RCCL send/recv)" if exchange == "gather" else "dist.reduce(sum) of the zero-padded frame (RCCL)")
                try:
                    rccl = None if rehearsal else list(torch.cuda.nccl.version())
                except Exception:       # noqa: BLE001
                    rccl = None
                seen = dist.get_world_size()
            multi = {"mode": mode, "exchange": how, "exchange_ms": round(exchange_ms, 4), "exchange_bytes_to_gpu0": int(W * H * 16 * (n_gpus - 1) / n_gpus) if exchange != "reduce" or loopback else W * H * 16,
                     "rccl_version": rccl, "gpus_seen": seen, "measurement": not (loopback or rehearsal), "set_devices_s": None if devices_s is None else round(devices_s, 3), "rccl_fallback": rccl_fallback}
        out = {
            "metric": "Msamples/s + achieved HBM GB/s, Sponza 1080p, 1/2/4/8xMI355X",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "regions_ms": [round(t * 1e3, 3) for t in regions],
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Sponza-class synthetic atrium (%d tris) as a .glaze V1 file (%d bytes) through parse, %dx%d, path tracer depth %d, %d steps = %.1f spp"
                                   % (int(info.n_world_triangles), glaze_bytes, W, H, args.depth, args.steps, args.steps / args.depth),
                       "width": W, "height": H, "depth": args.depth, "triangles": int(info.n_world_triangles),
                       "sharding": "64x64 tiles round-robin over %d GPU(s)%s" % (n_gpus, "" if n_gpus == 1 else ", RGBA32F accumulator onto GPU 0: " + how),
                       "bvh": {"builder": "binned SAH on the GPU, leaves of 1-2 triangles, 4-wide quantised nodes", "nodes": int(info.bvh_nodes),
                               "depth": int(info.bvh_depth), "sah_cost": round(float(info.bvh_sah_cost), 2), "build_ms": round(float(info.build_ms), 3)},
                       "setup_s": round(setup_s, 3), "serialize_s": round(serialize_s, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "multi_gpu": multi,
            "mpaths_per_s": round(value / args.depth, 2),
            "grays_per_s": round(value * (1 + counted["f_shadow"]) / 1e3, 3),
        }
    failure = None
    if n_gpus > 1 and args.verify and rank == 0:
        # the assembled frame against the whole frame rendered by GPU 0 alone, same seed and launch count; a failure
        # (or an exception) is only recorded here: every rank must still reach the barrier below, or the others hang in it
        try:
            launches = args.warmup + total_steps
            if world == 1:
                renderer.set_devices([0])       # back to one device (the peers' replicas and communicators are released)
            else:
                renderer.set_partition(0, 1)
            renderer.restart()
            renderer.step(launches)
            alone = renderer.read_hdr()
            same = bool(np.array_equal(np.nan_to_num(verify_frame, nan=-1.0).view(np.uint32), np.nan_to_num(alone, nan=-1.0).view(np.uint32)))
            out["verify"] = {"bit_identical_to_one_gpu": same, "launches": launches, "nonzero_pixels": int((alone[..., 3] > 0).sum())}
            if not same:
                failure = "multi-GPU frame differs from the single-GPU frame"
        except Exception as ex:     # noqa: BLE001 -- reported after the collective teardown
            failure = "verify failed: %r" % (ex,)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))
    if failure:
        raise SystemExit(failure)


if __name__ == "__main__":
    main()
