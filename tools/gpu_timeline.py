"""Summarises a rocprofv3 --kernel-trace CSV of a small-share run: per stream, the durations of k_trace / k_shade and the gaps
between consecutive kernels.  Usage: python tools/gpu_timeline.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
by = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "glz::k_trace" not in name and "glz::k_shade" not in name:
        continue
    by[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "trace" if "k_trace" in name else "shade"))
for q, ks in sorted(by.items()):
    ks.sort()
    ks = ks[len(ks) // 2:]            # the second half: steady state
    dur = defaultdict(list)
    gaps = defaultdict(list)
    for a, b in zip(ks, ks[1:]):
        dur[a[2]].append(a[1] - a[0])
        gaps[a[2] + "->" + b[2]].append(b[0] - a[1])
    span = (ks[-1][1] - ks[0][0]) / max(1, sum(1 for k in ks if k[2] == "trace"))
    print("queue %s: %d kernels; per launch %.1f us; " % (q, len(ks), span / 1e3) +
          "; ".join("%s %.1f us" % (k, sum(v) / len(v) / 1e3) for k, v in sorted(dur.items())) + "; gaps " +
          "; ".join("%s %.1f us" % (k, sum(v) / len(v) / 1e3) for k, v in sorted(gaps.items())))
