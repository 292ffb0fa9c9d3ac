#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs that tools/profile_gpu.sh left under gpurun_out/ into the committed summaries:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python3 bench.py ...`
  profiles/<tag>_pmc.json           per-kernel FETCH_SIZE / WRITE_SIZE averages (separate --pmc passes)
  profiles/<tag>_sq_counters.txt    per-kernel SQ issue and cache / texture-unit counters (their own --pmc passes) with the derived figures
  profiles/pmc_summary.json         what bench.py falls back to for roofline.traffic and the issue counters (latest round)
  profiles/<tag>_kernel_table.md    the "what bounds the kernels" table of DESIGN.md section 4, regenerated from the above

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
# issue and cache counters (their own passes): per-kernel averages, and what DESIGN.md section 4 derives from them
extra = defaultdict(lambda: defaultdict(list))
for sub in ("prof_sq", "prof_cache"):
    for f in newest(os.path.join(ROOT, "gpurun_out", sub, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                extra[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for k in sorted(extra):
    c = {n: sum(v) / len(v) for n, v in extra[k].items()}
    lines.append(k)
    for n in sorted(c):
        lines.append("   %-30s %16.1f  (n=%d)" % (n, c[n], len(extra[k][n])))
    d = {}
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        d["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] / 8.0 / c["SQ_BUSY_CYCLES"]
        d["lane_utilisation"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / 64.0 / c["SQ_ACTIVE_INST_VALU"]
        d["valu_insts_per_launch"] = c.get("SQ_INSTS_VALU", 0.0)
    if c.get("SQ_WAVE_CYCLES"):
        d["wait_share_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("GRBM_GUI_ACTIVE"):
        for n, key in (("TA_TA_BUSY_sum", "ta_busy"), ("TD_TD_BUSY_sum", "td_busy")):
            if n in c:
                d[key] = c[n] / (c["GRBM_GUI_ACTIVE"] / 8.0) / 256.0      # busy cycles per CU over the kernel's cycles (GRBM_GUI_ACTIVE counts per XCD)
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
        d["l1_accesses_per_launch"] = c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    for a, b in d.items():
        summary[k][a] = round(b, 4) if b < 100 else int(b)
    lines.append("   -> " + ", ".join("%s %s" % (a, summary[k][a]) for a in d))
if lines:
    open(os.path.join(out_dir, tag + "_sq_counters.txt"), "w").write("\n".join(lines) + "\n")
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
# DESIGN.md section 4's table, from the numbers above and the bench line taken under the kernel trace
rows = [("duration, rocprofv3 --kernel-trace --stats average", lambda v: "%.3f ms (%d calls)" % (v.get("kernel_trace_avg_ns", 0) / 1e6, v.get("kernel_trace_calls", 0))),
        ("L2-miss traffic (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB per launch", lambda v: "%.2f GB (%.2f uncorrected)" % (v.get("hbm_bytes_per_launch", 0) / 1e9, v.get("hbm_bytes_per_launch_uncorrected", 0) / 1e9)),
        ("... over the launch time", lambda v: "%.2f TB/s = %.2f of 8 TB/s (%.2f uncorrected)" % (v.get("hbm_bytes_per_launch", 0) / max(v.get("kernel_trace_avg_ns", 1), 1) / 1e3, v.get("hbm_bytes_per_launch", 0) / max(v.get("kernel_trace_avg_ns", 1), 1) / 8e3, v.get("hbm_bytes_per_launch_uncorrected", 0) / max(v.get("kernel_trace_avg_ns", 1), 1) / 8e3)),
        ("VALU busy, lane utilisation, wave cycles waiting", lambda v: "%.0f %%, %.0f %%, %.0f %%" % (100 * v.get("valu_busy", 0), 100 * v.get("lane_utilisation", 0), 100 * v.get("wait_share_of_wave_cycles", 0))),
        ("VALU instructions per launch", lambda v: "%.0f M" % (v.get("valu_insts_per_launch", 0) / 1e6)),
        ("texture-address / data-return units busy, L1 accesses per launch, L2 hit rate", lambda v: "%.0f %%, %.0f %%, %.0f M, %.0f %%" % (100 * v.get("ta_busy", 0), 100 * v.get("td_busy", 0), v.get("l1_accesses_per_launch", 0) / 1e6, 100 * v.get("l2_hit_rate", 0)))]
ks = [k for k in KERNELS if k in summary]
with open(os.path.join(out_dir, tag + "_kernel_table.md"), "w") as f:
    f.write("| per launch | " + " | ".join("`%s`" % k for k in ks) + " |\n|---|" + "---|" * len(ks) + "\n")
    for name, fmt in rows:
        f.write("| %s | " % name + " | ".join(fmt(summary[k]) for k in ks) + " |\n")
print(open(os.path.join(out_dir, tag + "_kernel_table.md")).read())
