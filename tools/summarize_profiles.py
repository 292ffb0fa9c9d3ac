#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that tools/profile_gpu.sh left under gpurun_out/ into the committed summaries:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python3 bench.py ...`
  profiles/<tag>_pmc.json           per-kernel FETCH_SIZE / WRITE_SIZE averages (separate --pmc passes)
  profiles/pmc_summary.json         what bench.py reads for roofline.traffic (latest round)

HBM bytes per launch follow MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide (16 B/lane) reads, so hbm = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
The 2x is calibrated for coalesced streams; the traversal's scattered 16-byte loads are uncalibrated,
so the figure is an upper-bound style estimate (ratios between variants are unaffected).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) < 2:
    sys.exit("usage: summarize_profiles.py <round tag, e.g. r03>")
tag = sys.argv[1]
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)

KERNELS = ("k_trace", "k_shade")


def short(name):
    if "<true" in name.replace(" ", ""):
        return None          # instrumented (counting) variants are not the measured kernels
    for k in KERNELS:
        if "::" + k + "<" in name or "::" + k + "(" in name:
            return k
    return None


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/ and never deletes: only the latest run of a directory counts"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


stats = newest(os.path.join(ROOT, "gpurun_out", "prof_trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out_dir, tag + "_kernel_stats.csv"))
summary = defaultdict(dict)
for counter, sub in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    files = newest(os.path.join(ROOT, "gpurun_out", sub, "*", "*_counter_collection.csv"))
    acc = defaultdict(list)
    dur = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k and row["Counter_Name"] == counter:
                acc[k].append(float(row["Counter_Value"]))
                dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items():
        summary[k][counter + "_KiB_avg"] = sum(v) / len(v)
        summary[k][counter + "_launches"] = len(v)
        summary[k][counter + "_pass_avg_ns"] = sum(dur[k]) / len(dur[k])
for k, v in summary.items():
    if "FETCH_SIZE_KiB_avg" in v and "WRITE_SIZE_KiB_avg" in v:
        v["hbm_bytes_per_launch"] = int((2.0 * v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024)
        v["hbm_bytes_per_launch_uncorrected"] = int((v["FETCH_SIZE_KiB_avg"] + v["WRITE_SIZE_KiB_avg"]) * 1024)
if stats:
    for row in csv.DictReader(open(stats[0])):
        k = short(row["Name"])
        if k:
            summary[k]["kernel_trace_avg_ns"] = float(row["AverageNs"])
            summary[k]["kernel_trace_calls"] = int(row["Calls"])
bench_line = None
bj = os.path.join(ROOT, "gpurun_out", "prof_trace.json")
if os.path.exists(bj):
    try:
        bench_line = json.loads(open(bj).read().strip().splitlines()[-1])
    except Exception:
        bench_line = None
doc = dict(summary)
doc["_command"] = "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py (tools/profile_gpu.sh)"
doc["_bench_line_under_profiler"] = bench_line
json.dump(doc, open(os.path.join(out_dir, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
json.dump(doc, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(summary.items()):
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
