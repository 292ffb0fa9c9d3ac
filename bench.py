#!/usr/bin/env python3
"""bench.py -- Msamples/s + achieved HBM GB/s of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]                      (N > 1: ONE process drives the N GPUs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                               (one process per GPU)

Workload (BASELINE.json configs[3]): the Sponza-class frame, 1920x1080, path tracer, depth 8.  Sponza
itself is not in the reference repository, so the scene is the deterministic synthetic
atrium of glaze_amd/scenes.py (262 267 triangles, 25 Lambert / Uber materials each with its own 1024^2 sRGB texture, sun + 2048x1024 sky), written as a `.glaze` V1 file
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

After the timed regions (and outside them) the same invocation also times, for at most half a second each, the other BASELINE
configurations on one GPU (`extra.configs`: 2 = cube 512^2 depth 2, 3 = mattest.glaze 1024^2 depth 8, 5 = the atrium at 3840x2160 depth 12)
and the atrium with the content classes real Sponza has and the stand-in lacks (`extra.atrium_sponza_like`: opacity-mapped cloth, foliage
cards and vines, normal maps on stone, roughness maps on the Uber materials); `--no-extras` skips them.  N > 1: every step that can block
on another GPU -- spanning the devices, every exchange, the process group's barrier -- runs under a watchdog: after its time-out
(GLAZE_BENCH_WATCHDOG_SPAN_S, default 180 s, to span the devices; GLAZE_BENCH_WATCHDOG_S, default 90 s, per exchange or barrier) it prints `multi_gpu.diagnostics` (which call, how long, RCCL
version, the peer-access matrix, bytes per peer, the RCCL / HSA environment) to stderr and ends the process with status 3 through os._exit.

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
    ap.add_argument("--scene", default=os.environ.get("GLAZE_BENCH_SCENE") or None,
                    help="a .glaze V1 file (or an .obj, converted with glz_convert_obj first) to measure instead of the synthetic atrium, "
                         "e.g. the Sponza the reference's README links (also: GLAZE_BENCH_SCENE)")
    ap.add_argument("--config5", dest="config5", action="store_true", default=None,
                    help="also time BASELINE configs[4] (3840x2160, depth 12, same scene) as a second, separately named measurement "
                         "(default: when N > 1)")
    ap.add_argument("--no-config5", dest="config5", action="store_false")
    ap.add_argument("--no-extras", action="store_true", help="N = 1: do not time the other BASELINE configurations and the Sponza-like atrium after the timed regions")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: do not time BASELINE configs 2, 3 and 5 (extra.configs)")
    return ap.parse_args()


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES"))
PMC_CHILD_STEPS, PMC_CHILD_WARMUP = 16, 8


def run_bounded(cmd, timeout, **kw):
    """subprocess.run with a timeout that takes the child's whole process GROUP down: rocprofv3 starts the workload as a child of its
    own, and killing only the launcher would leave a render running on the GPU next to the measurement that follows."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        p.communicate()
        return None, "", ""
    return p.returncode, out, err


class Watchdog:
    """Ends the process when a step that can block on another GPU does not come back: `with dog.phase(name, seconds, **facts)` arms it, leaving
    the block disarms it.  A hang inside ncclGroupEnd (or a barrier) on the first contact between ranks would otherwise burn the caller's
    whole time-out and leave nothing to look at; this prints what is known -- as one JSON object, `multi_gpu.diagnostics` -- and exits
    with status 3 through os._exit (no re-exec, no retry in the same process: the GPU state of a wedged collective is not worth keeping)."""

    def __init__(self, facts):
        import threading
        self.facts = facts              # callable -> dict of what the diagnostics always carry
        self.lock = threading.Lock()
        self.current = None             # (name, deadline, started, extra)
        self.history = []
        t = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        t.start()

    def phase(self, name, seconds, **extra):
        dog = self

        class _Phase:
            def __enter__(self_inner):
                with dog.lock:
                    dog.current = (name, time.monotonic() + seconds, time.monotonic(), extra, seconds)
                return self_inner

            def __exit__(self_inner, *exc):
                with dog.lock:
                    nm, _, started, _, _ = dog.current
                    dog.history.append((nm, round((time.monotonic() - started) * 1e3, 3)))
                    dog.current = None
                return False
        return _Phase()

    def _run(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                cur = self.current
            if cur is None or time.monotonic() < cur[1]:
                continue
            name, _, started, extra, seconds = cur
            diag = {"stuck_in": name, "timeout_s": seconds, "elapsed_s": round(time.monotonic() - started, 1), "completed_before": self.history[-8:]}
            diag.update(extra)
            try:
                diag.update(self.facts())
            except Exception as e:      # noqa: BLE001 -- the diagnostics must come out whatever else is broken
                diag["facts_error"] = repr(e)
            sys.stderr.write("bench.py: watchdog: `%s` did not return within %g s -- giving up\n" % (name, seconds))
            sys.stderr.write(json.dumps({"multi_gpu": {"diagnostics": diag}}) + "\n")
            sys.stderr.flush()
            os._exit(3)


def measure_pmc(args, scene_file=None):
    """roofline.traffic and the dominant kernel's real bound, measured by THIS invocation: three short runs of this very workload under
    `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE and the SQ issue counters in separate passes with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes), per-kernel averages over the steady-state launches: the first PMC_CHILD_WARMUP launches of every
    kernel (fresh camera rays, by Dispatch_Id order) are dropped, like the warm-up of the timed regions.
    HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1 KiB: the counters are in KiB and on gfx950 FETCH_SIZE reports half of the bytes of
    wide reads (the 2 x is calibrated for coalesced streams: an upper estimate for scattered 16-byte loads, so both readings are kept).
    Returns ({kernel: {...}}, note) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    # not from under a profiler: its preloaded library initialises the GPU in every child before the child's own exec
    if any("rocprof" in v.lower() or "rocprofiler" in v.lower() for v in (os.environ.get("LD_PRELOAD", ""), os.environ.get("ROCP_TOOL_LIBRARIES", ""), os.environ.get("HSA_TOOLS_LIB", ""), os.environ.get("ROCPROFILER_LIBRARY", ""))):
        return None, "this process runs under a profiler"
    acc = {}
    failed = []
    tmp = tempfile.mkdtemp(prefix="glaze_pmc_")
    try:
        for counters in PMC_PASSES:
            out = os.path.join(tmp, counters[0])
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--no-pmc", "--steps", str(PMC_CHILD_STEPS), "--warmup", str(PMC_CHILD_WARMUP), "--width", str(args.width), "--height", str(args.height),
                   "--depth", str(args.depth), "--seed", str(args.seed)] + (["--scene", os.path.abspath(args.scene or scene_file)] if (args.scene or scene_file) else [])
            rc, so, se = run_bounded(cmd, 150, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            why = None
            if rc is None:
                why = "did not finish within 150 s (its process group was killed)"
            elif rc != 0:
                why = "rc %d: %s" % (rc, (se or so)[-200:].replace("\n", " "))
            files = [] if why else glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            if not why and not files:
                why = "left no counter_collection.csv"
            if why:
                if counters[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    return None, "rocprofv3 --pmc %s %s" % (counters[0], why)
                failed.append("--pmc %s %s" % (" ".join(counters), why))
                continue
            rows = {}       # (kernel, counter) -> [(dispatch id, value)]
            for f in files:
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].replace(" ", "")
                    if row["Counter_Name"] not in counters or "<true" in name:
                        continue
                    for k in ("k_trace", "k_shade", "k_path"):
                        if "::" + k + "<" in name or "::" + k + "(" in name:
                            rows.setdefault((k, row["Counter_Name"]), []).append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
            for (k, c), v in rows.items():
                v.sort()
                # by LAUNCH, not by row: a kernel may be dispatched more than once per launch (several chains) or once more at the end (the
                # stand-alone shadow pass), and the warm-up's launches must go whole
                per_launch = max(1, round(len(v) / float(PMC_CHILD_WARMUP + PMC_CHILD_STEPS)))
                steady = [x for _, x in v[PMC_CHILD_WARMUP * per_launch:]] or [x for _, x in v]
                acc.setdefault(k, {})[c] = (sum(steady) / len(steady), len(steady))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    res = {}
    for k, v in acc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            res[k] = {"hbm_bytes_per_launch": int((2.0 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024),
                      "uncorrected": int((v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024), "launches": min(v["FETCH_SIZE"][1], v["WRITE_SIZE"][1])}
            if all(c in v for c in PMC_PASSES[2]) and v["SQ_BUSY_CYCLES"][0] > 0 and v["SQ_ACTIVE_INST_VALU"][0] > 0:
                res[k]["valu_busy"] = round(v["SQ_ACTIVE_INST_VALU"][0] / 8.0 / v["SQ_BUSY_CYCLES"][0], 3)
                res[k]["lane_utilisation"] = round(v["SQ_THREAD_CYCLES_VALU"][0] / 64.0 / v["SQ_ACTIVE_INST_VALU"][0], 3)
                res[k]["valu_insts_per_launch"] = int(v["SQ_INSTS_VALU"][0])
    if not res:
        return None, "no k_trace / k_shade rows in the counter files"
    note = ("rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ issue counters, each with --kernel-trace only) of this workload, run by this invocation after "
            "its timed regions: launches %d..%d of each, (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB per launch" % (PMC_CHILD_WARMUP, PMC_CHILD_WARMUP + PMC_CHILD_STEPS))
    if failed:
        note += "; not collected: " + "; ".join(failed)
    return res, note


def kernel_bound(pmc, avg_ms):
    """What the counters say bounds a kernel: VALU issue when the SIMDs' vector ALUs are busy most of the kernel; HBM when the bytes that
    cross the fabric come close to what the chip streams (MI355X_MICROARCH.md: 6.29 TB/s measured achievable of the 8 TB/s specification);
    otherwise the memory system's random-access rate / latency (neither the ALUs nor the pins are busy)."""
    if pmc is None:
        return None
    if pmc.get("valu_busy") is not None and pmc["valu_busy"] >= 0.75:
        return "valu-issue"
    if avg_ms > 0 and pmc["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 >= 0.7 * 6290.0:
        return "hbm"
    return "memory-latency" if pmc.get("valu_busy") is not None else None


def algorithmic_bytes(c):
    """Algorithmic bytes per SAMPLE of each kernel, from counted per-sample work.

    node visit = 64 B (one quantised BVH4 node: four child boxes + four links), triangle test = 36 B (the three vertices;
    the tracer reads one 64-byte BvhQuad per leaf -- four vertices and the ids for two triangles that share an edge, 32 B a test, the
    full 64 for a single triangle -- so 36 B is within 12 % of what crosses the wire for the atrium's leaves; rounds 1-3 booked the
    48-byte BvhTri records the tracer read then), hit-attribute fetch = 192 B (128-B shading record + the 64 B of RTMaterial
    scalars k_shade loads), path state = 96 B (ray 32 + importance 64), hit record 16 B, accumulator = 32 B r/w + 16 B result,
    shadow-queue entry = 48 B.  Counted too since round 3 (they were the unbooked part of k_shade's traffic): texels (16 B per
    bilinear fetch of an RGBA texture, 4 B per gray one: SURVEY 8(d)'s "16 B/texture tap"), 112 B of RTLight per light sample, and
    per sky-light sample the binary search of the marginal table (4 B a step), one marginal and two conditional values.
    """
    # (the camera ray of a new path: written by k_trace's refill until round 5 -- 32 B x f_fresh there -- and since then by the shading
    # code where the old path ends: the pixels that hit write their 96 B of state either way, the ones that missed 32 B they did not)
    closest = 32 + 16 + 64 * c["nodes_closest"] + 36 * c["tris_closest"]
    shade = (16 + 32 + 64 * (1 - c["f_fresh"]) + 192 * c["f_hit"] + 48 * c["f_shadow"] + 48 * (1 - c["f_shadow"])
             + 96 * c["f_hit"] + 32 * (1 - c["f_hit"]))
    shade += c.get("tex_bytes_shade", 0.0) + 112 * (c.get("light_samples", 0.0) + c.get("sky_samples", 0.0)) + c.get("sky_bytes_per_sample", 0.0) * c.get("sky_samples", 0.0)
    shadow = 48 * c["f_shadow"] + 48 * c["f_shadow"] + 64 * c["nodes_shadow"] + 36 * c["tris_shadow"]
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

    def diagnostics_facts():
        """what the watchdog's report always carries (nothing here may block on another GPU)"""
        n_seen = torch.cuda.device_count()
        facts = {"rank": rank, "world_size": world, "devices_in_process": n_dev, "gpus_visible": n_seen, "exchange": exchange, "loopback": bool(loopback), "rehearsal": rehearsal,
                 "env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_", "GLAZE_", "GPU_MAX", "ROCR_")) or k in ("MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK")}}
        try:
            facts["rccl_version"] = int(glaze_amd.abi.lib().glz_rccl_version()) if world == 1 else list(torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001
            facts["rccl_version"] = "unavailable: %r" % (e,)
        try:
            facts["peer_access"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n_seen)] for i in range(n_seen)]
        except Exception as e:      # noqa: BLE001
            facts["peer_access"] = "unavailable: %r" % (e,)
        return facts

    dog = Watchdog(diagnostics_facts) if n_gpus > 1 else None
    span_timeout = float(os.environ.get("GLAZE_BENCH_WATCHDOG_SPAN_S", "180"))      # spanning the devices / the process group's init
    exchange_timeout = float(os.environ.get("GLAZE_BENCH_WATCHDOG_S", "90"))        # every exchange, every barrier

    def guarded(name, seconds, fn, **extra):
        if dog is None:
            return fn()
        with dog.phase(name, seconds, **extra):
            return fn()

    if world > 1:
        if rehearsal:
            guarded("dist.init_process_group(gloo)", span_timeout, lambda: dist.init_process_group(backend="gloo"))
        else:
            guarded("dist.init_process_group(nccl = RCCL)", span_timeout, lambda: dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank)))

    inst = glaze_amd.RayTraceInstance.new(local_rank)
    if inst is None:
        raise SystemExit("no gfx950 device for rank %d: %s" % (rank, glaze_amd.abi.last_error()))
    W, H = args.width, args.height
    import tempfile
    from glaze_amd.scene_desc import save_scene
    serialize_s = 0.0
    desc = None                 # what the CPU baseline renders: the generator's description, or the oracle's own reading of the file
    import atexit
    import shutil
    tmp = tempfile.mkdtemp(prefix="glaze_bench_")        # kept until the process ends: the PMC passes' children read the scene file again
    atexit.register(shutil.rmtree, tmp, ignore_errors=True)
    scene_file = None
    if True:
        if args.scene:
            # a supplied scene (BASELINE configs[3] reads "Sponza .glaze"; the reference's README links one): the file as it is, or an
            # .obj through the converter (glz_convert_obj = glaze-converter, converter/src/main.rs)
            if not os.path.exists(args.scene):
                raise SystemExit("--scene %s: no such file" % args.scene)
            path = args.scene
            if path.lower().endswith(".obj"):
                path = os.path.join(tmp, os.path.splitext(os.path.basename(args.scene))[0] + "_rank%d.glaze" % rank)
                t0 = time.time()
                glaze_amd.convert_obj(args.scene, path)
                serialize_s = time.time() - t0
            scene_name = os.path.basename(args.scene)
            data_tag = "file:" + scene_name
        else:
            # the scene goes through the file format end to end: Serializer -> .glaze V1 -> parse -> RayTraceScene::new (cli/src/main.rs:78-91)
            desc = atrium_scene()
            path = os.path.join(tmp, "atrium_rank%d.glaze" % rank)
            t0 = time.time()
            save_scene(desc, path)
            serialize_s = time.time() - t0
            scene_file = path
            scene_name = "Sponza-class synthetic atrium"
            data_tag = "synthetic"
        glaze_bytes = os.path.getsize(path)
        # what the bookkeeping needs to know of the file (the sky's table height): read through the same parser, before the scene consumes it
        meta_parse = glaze_amd.parse(path)
        sky_ids = [int(l.resource_id) for l in meta_parse.lights() if int(l.ltype) == glaze_amd.abi.LIGHT_SKY]
        tex = meta_parse.textures() if sky_ids else []
        sky_rows = max([int(tex[i][1].shape[0]) for i in sky_ids if i < len(tex)] + [0])
        n_lights = len(meta_parse.lights())
        meta_parse.close()
        t0 = time.time()
        scene = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(path))
        setup_s = time.time() - t0
        if desc is None and rank == 0 and not args.no_cpu_baseline and not args.pmc_child and n_gpus == 1:
            from oracle.pyoracle import desc_from_file        # (the checker's own reader: only the cpu_baseline leg uses it)
            desc = desc_from_file(path)
    info = scene.info()
    if n_lights == 0:
        print("bench.py: %s has no lights -- the reference's raygen returns at once for such a scene (path_trace.rgen:137-141) and so does this "
              "renderer: the numbers below time empty launches" % scene_name, file=sys.stderr, flush=True)
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
            guarded("glz_renderer_set_devices (scene replicas, BVH builds, ncclCommInitAll)", span_timeout, lambda: renderer.set_devices(ids), devices=ids)
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
            guarded("dist.barrier", exchange_timeout, dist.barrier)
            torch.cuda.synchronize()

    n_exchanges = [0]

    def exchange_to_gpu0():
        """every GPU's tiles onto GPU 0 (RCCL over xGMI; gloo on host tensors in a rehearsal), under the watchdog"""
        n_exchanges[0] += 1
        per_peer = int(renderer_size[0] * renderer_size[1] * 16 / n_gpus) if exchange != "reduce" else renderer_size[0] * renderer_size[1] * 16
        guarded("exchange #%d onto GPU 0 (%s)" % (n_exchanges[0], exchange), exchange_timeout, exchange_unguarded, bytes_per_peer=per_peer, peers=n_gpus - 1,
                frame="%dx%d" % (renderer_size[0], renderer_size[1]))

    renderer_size = [W, H]

    def exchange_unguarded():
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
        # one of the rocprofv3 --pmc passes of measure_pmc(): the same launches as the timed regions, nothing else
        renderer.restart()
        renderer.step(args.warmup + args.steps)
        renderer.wait_idle()
        return
    # ---- warmup (untimed): same accumulation continues afterwards, like the interactive draw_frame loop
    renderer.restart()
    renderer.step(args.warmup)
    first_exchange_ms = None
    if n_gpus > 1:
        try:
            t_first = time.perf_counter()
            exchange_to_gpu0()      # first use of the communicators / the process group belongs to the warm-up
            first_exchange_ms = (time.perf_counter() - t_first) * 1e3
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
    # ---- BASELINE configs[4] next to it (N > 1 by default): the same scene at 3840x2160, depth 12, tiles over the same GPUs, the same
    # exchange inside the timed regions -- a second, separately named measurement; `value` stays the 1080p frame.
    config5 = None
    if args.config5 if args.config5 is not None else n_gpus > 1:
        W5, H5, D5 = 3840, 2160, 12
        renderer.change_resolution(W5, H5)
        renderer_size[:] = [W5, H5]
        renderer.set_depth(D5)
        frame_1080 = frame
        frame = torch.zeros((H5, W5, 4), dtype=torch.float32, device="cuda")
        renderer.restart()
        renderer.step(args.warmup)
        if n_gpus > 1:
            exchange_to_gpu0()
        r5 = [timed_region() for _ in range(3)]
        e5 = sorted(r5)[1]
        config5 = {"workload": "%s, %dx%d, path tracer depth %d, %d steps per region" % (scene_name, W5, H5, D5, args.steps), "ms_per_step": round(e5 / args.steps * 1e3, 4),
                   "value": round(W5 * H5 * args.steps / e5 / 1e6, 2), "unit": "Msamples/s", "regions_ms": [round(t * 1e3, 3) for t in r5], "launch_mode": renderer.launch_mode()}
        frame = frame_1080
        renderer.change_resolution(W, H)
        renderer_size[:] = [W, H]
        renderer.set_depth(args.depth)
        renderer.restart()
        sync_all()
        renderer.stats()
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
            measured, note = measure_pmc(args, scene_file)
            if measured is not None and dominant in measured:
                traffic, traffic_source, traffic_all = measured[dominant]["hbm_bytes_per_launch"], note, measured
            else:
                traffic_source = "not measured in this run (%s); " % note
        prof = os.path.join(ROOT, "profiles", "pmc_summary.json")
        committed = None
        if os.path.exists(prof) and n_gpus == 1 and not args.scene:      # the committed PMC passes profiled the default N = 1 command
            try:
                committed = json.load(open(prof))
            except Exception:       # noqa: BLE001
                committed = None
        if committed:       # (the summary's own key for the uncorrected reading, under the name this line uses)
            for v in committed.values():
                if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
                    v.setdefault("uncorrected", v.get("hbm_bytes_per_launch_uncorrected", v["hbm_bytes_per_launch"]))
        if traffic is None and committed:
            traffic = committed.get(dominant, {}).get("hbm_bytes_per_launch")
            traffic_all = {k: v for k, v in committed.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v}
            traffic_source = (traffic_source or "") + "profiles/pmc_summary.json (rocprofv3 --pmc passes of `python bench.py`, not this run)"
        elif traffic_all and committed:
            # the SQ pass alone failed: the committed pass stands in for the issue counters only (and the line says so)
            for k, v in traffic_all.items():
                if "valu_busy" not in v and "valu_busy" in committed.get(k, {}):
                    v.update({c: committed[k][c] for c in ("valu_busy", "lane_utilisation", "valu_insts_per_launch") if c in committed[k]})
                    v["issue_counters_from"] = "profiles/pmc_summary.json"
        # What the COUNTERS say next to the algorithmic fraction SURVEY 8(d) defines: the bytes that really crossed the fabric per launch
        # over the launch time, against the 8 TB/s specification and the 6.29 TB/s MI355X_MICROARCH.md measures as achievable -- with the
        # guide's 2 x FETCH_SIZE correction (an upper estimate for scattered 16-byte loads) and without it -- for the dominant kernel and
        # for the whole step; and what bounds the dominant kernel by its issue counters.
        step_ms = elapsed / args.steps * 1e3
        counter_frac, bound, kernels_pmc = None, None, {}
        if traffic_all:
            def frac(nbytes, ms):
                return None if not ms else round(nbytes / (ms * 1e-3) / 1e9 / 8000.0, 4)
            dom = traffic_all.get(dominant)
            whole_c = sum(v["hbm_bytes_per_launch"] for k, v in traffic_all.items() if k in kern_ms and kern_ms[k] > 0)
            whole_u = sum(v["uncorrected"] for k, v in traffic_all.items() if k in kern_ms and kern_ms[k] > 0)
            counter_frac = {"kernel": None if dom is None else {"corrected": frac(dom["hbm_bytes_per_launch"], avg_ms), "uncorrected": frac(dom["uncorrected"], avg_ms)},
                            "whole_step": {"corrected": frac(whole_c, step_ms), "uncorrected": frac(whole_u, step_ms)},
                            "peak": 8000.0, "achievable_peak": 6290.0, "unit": "GB/s",
                            "note": "traffic / launch time / 8 TB/s; `corrected` = (2 x FETCH_SIZE + WRITE_SIZE), the guide's gfx950 correction for coalesced streams, an upper bound for "
                                    "scattered 16-byte loads; FETCH_SIZE includes Infinity-Cache hits"}
            bound = kernel_bound(dom, avg_ms)
            for k, v in traffic_all.items():
                if k in kern_ms and kern_ms[k] > 0:
                    kernels_pmc[k] = {"bound": kernel_bound(v, kern_ms[k] / total_steps), "valu_busy": v.get("valu_busy"), "lane_utilisation": v.get("lane_utilisation"),
                                      "valu_insts_per_launch": v.get("valu_insts_per_launch"),
                                      "counter_frac": frac(v["hbm_bytes_per_launch"], kern_ms[k] / total_steps), "algorithmic_frac": round(bytes_per_sample[k] * owned / (kern_ms[k] / total_steps * 1e-3) / 1e9 / 8000.0, 4)}
        # `bound`: what limits the dominant kernel by its own counters ("valu-issue": the vector ALUs issue most of the kernel's cycles;
        # "hbm": the fabric-side bytes are near what the chip streams; "memory-latency": neither).  `roofline_class` is the roof `frac` is
        # priced against (HBM bandwidth; no MFMA on this path).
        roofline = {"bound": bound or "hbm", "roofline_class": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_per_kernel": traffic_all,
                    "counter_frac": counter_frac, "valu_busy": None if not traffic_all or dominant not in traffic_all else traffic_all[dominant].get("valu_busy"),
                    "lane_utilisation": None if not traffic_all or dominant not in traffic_all else traffic_all[dominant].get("lane_utilisation"),
                    "per_kernel": kernels_pmc,
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
                   "sample": "%dx%d %s, depth %d, %d launches (%.1f s)" % (W, H, "atrium" if not args.scene else scene_name, args.depth, args.cpu_launches, dt)}
            # the same port on ONE core (SURVEY 8d asks for both): one launch of the same frame
            o.set_threads(1)
            o.restart()
            tc = time.perf_counter()
            o.step(1)
            dt1 = time.perf_counter() - tc
            cpu["single_thread"] = {"value": round(W * H / dt1 / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                    "sample": "1 launch (%.1f s)" % dt1}
        extra = None
        if n_gpus == 1 and not args.no_extras and not args.scene:
            extra = {}

            def quick_time(r, w, h, budget_s=0.4, warm=16):
                """<= ~0.5 s of launches of a renderer after a warm-up: ms per launch, Msamples/s"""
                r.restart()
                r.step(warm)
                r.wait_idle()
                tq = time.perf_counter()
                r.step(16)
                r.wait_idle()
                est = (time.perf_counter() - tq) / 16
                n = int(max(16, min(2048, budget_s / max(est, 1e-6))))
                tq = time.perf_counter()
                r.step(n)
                r.wait_idle()
                dtq = time.perf_counter() - tq
                return {"ms_per_step": round(dtq / n * 1e3, 4), "value": round(w * h * n / dtq / 1e6, 2), "unit": "Msamples/s", "steps": n, "launch_mode": r.launch_mode()}

            if not args.no_configs:
                # the other BASELINE configurations on this GPU (parity-test cases, not bench lines: timed here so that every one of them has a
                # number the driver's own run produced)
                from glaze_amd.scenes import cube_scene
                cfgs = {}
                r2 = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, cube_scene()), 512, 512)
                r2.set_depth(2)
                r2.set_seed(args.seed)
                cfgs["2"] = dict(quick_time(r2, 512, 512), workload="cube (12 tris, Lambert + omni light), 512x512, depth 2")
                del r2
                mt = os.path.join(ROOT, "tests", "golden", "mattest.glaze")
                if os.path.exists(mt):
                    r3 = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(mt)), 1024, 1024)
                    r3.set_depth(8)
                    r3.set_seed(args.seed)
                    cfgs["3"] = dict(quick_time(r3, 1024, 1024), workload="mattest.glaze as it is (138 480 tris), 1024x1024, depth 8")
                    del r3
                else:
                    cfgs["3"] = {"skipped": "tests/golden/mattest.glaze is not here"}
                renderer.change_resolution(3840, 2160)
                renderer.set_depth(12)
                cfgs["5"] = dict(quick_time(renderer, 3840, 2160), workload="%s, 3840x2160, depth 12, the whole frame on ONE GPU" % scene_name)
                renderer.change_resolution(W, H)
                renderer.set_depth(args.depth)
                extra["configs"] = cfgs
            # the atrium with what real Sponza has and the stand-in lacks: alpha-tested cloth / foliage / vines (the any-hit texture fetch
            # inside the traversal), normal maps on stone, roughness maps on the Uber materials -- through the file format like the headline
            tq = time.time()
            sp_path = os.path.join(tmp, "atrium_sponza_like.glaze")
            save_scene(atrium_scene(sponza_like=True), sp_path)
            sp_scene = glaze_amd.RayTraceScene.new(inst, glaze_amd.parse(sp_path))
            sp_info = sp_scene.info()
            rs = glaze_amd.RayTraceRenderer.new(inst, sp_scene, W, H)
            rs.set_depth(args.depth)
            rs.set_seed(args.seed)
            sp = quick_time(rs, W, H)
            rs.enable_counters(True, True)
            rs.restart()
            rs.step(args.warmup)
            rs.wait_idle()
            qa = rs.stats()
            rs.step(64)
            rs.wait_idle()
            qc = rs.stats()
            qr = max(1, qc.closest_rays - qa.closest_rays)
            sp.update({"workload": "atrium + lace cloth, foliage cards, vines (opacity maps), normal maps on stone / brick, roughness maps on Uber (%d tris), %dx%d, depth %d"
                                   % (int(sp_info.n_world_triangles), W, H, args.depth),
                       "tex_bytes_trace": round((qc.alpha_tex_bytes - qa.alpha_tex_bytes) / qr, 3),
                       "counted_per_sample": {"nodes_closest": round((qc.closest_nodes - qa.closest_nodes) / qr, 3), "tris_closest": round((qc.closest_tris - qa.closest_tris) / qr, 3),
                                              "nodes_shadow": round((qc.shadow_nodes - qa.shadow_nodes) / qr, 3), "tris_shadow": round((qc.shadow_tris - qa.shadow_tris) / qr, 3),
                                              "f_hit": round((qc.hits - qa.hits) / qr, 3), "f_shadow": round((qc.shadow_rays - qa.shadow_rays) / qr, 3),
                                              "tex_fetches": round((qc.tex_fetches - qa.tex_fetches) / qr, 3),
                                              "tex_bytes_shade": round(((qc.tex_bytes - qa.tex_bytes) - (qc.alpha_tex_bytes - qa.alpha_tex_bytes)) / qr, 3)},
                       "setup_s": round(time.time() - tq, 2)})
            extra["atrium_sponza_like"] = sp
            del rs
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
                     "rccl_version": rccl, "gpus_seen": seen, "measurement": not (loopback or rehearsal), "set_devices_s": None if devices_s is None else round(devices_s, 3), "rccl_fallback": rccl_fallback,
                     "first_exchange_ms": None if first_exchange_ms is None else round(first_exchange_ms, 3),
                     "exchange_bytes_per_peer": int(W * H * 16 / n_gpus) if exchange != "reduce" else W * H * 16,
                     "watchdog": {"span_timeout_s": span_timeout, "exchange_timeout_s": exchange_timeout, "phases_ms": dog.history[:6] + dog.history[-4:] if dog else None},
                     "config5": config5}
        out = {
            "metric": "Msamples/s + achieved HBM GB/s, Sponza 1080p, 1/2/4/8xMI355X",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "regions_ms": [round(t * 1e3, 3) for t in regions],
            "vs_baseline": None, "dtype": "f32", "data": data_tag,
            "config": {"workload": "%s (%d tris, %d materials, %d textures) as a .glaze V1 file (%d bytes) through parse, %dx%d, path tracer depth %d, %d steps = %.1f spp"
                                   % (scene_name, int(info.n_world_triangles), int(info.n_materials), int(info.n_textures), glaze_bytes, W, H, args.depth, args.steps, args.steps / args.depth),
                       "materials": int(info.n_materials), "textures": int(info.n_textures),
                       "scene": scene_name,
                       "width": W, "height": H, "depth": args.depth, "triangles": int(info.n_world_triangles),
                       "sharding": "64x64 tiles round-robin over %d GPU(s)%s" % (n_gpus, "" if n_gpus == 1 else ", RGBA32F accumulator onto GPU 0: " + how),
                       "bvh": {"builder": "binned SAH on the GPU, leaves of 1-2 triangles, 4-wide quantised nodes", "nodes": int(info.bvh_nodes),
                               "depth": int(info.bvh_depth), "sah_cost": round(float(info.bvh_sah_cost), 2), "build_ms": round(float(info.build_ms), 3)},
                       "setup_s": round(setup_s, 3), "serialize_s": round(serialize_s, 3)},
            "roofline": roofline, "cpu_baseline": cpu, "multi_gpu": multi, "config5": config5 if multi is None else None, "extra": extra,
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
