#!/usr/bin/env python3
"""Prints per-kernel averages of every counter found under gpurun_out/pmc_*/ (written by tools/pmc_gpu.sh)."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "glz::k_" not in k or "<true" in k:          # counting builds are not the measured kernels
            continue
        k = k.split("glz::")[1].split("(")[0].split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(x in k for x in ("trace", "shade", "shadow", "path")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
