#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of k_trace and k_shade per launch for several checkouts of this repository (git worktrees under variants/wt_*, each
# with its library built) and for the tree itself: one rocprofv3 --pmc pass each, the same render loop (tools/pmc_render.py), launches 16..40.
#   tools/pmc_bisect.sh [counter ...]      (run through gpurun; output: gpurun_out/r05/pmc_bisect.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
COUNTERS=${@:-WRITE_SIZE FETCH_SIZE}
: > $OUT/pmc_bisect.txt
for tree in $REPO/variants/wt_* $REPO; do
  [ -f $tree/glaze_amd/csrc/libglaze_hip.so ] || continue
  for c in $COUNTERS; do
    d=/tmp/pmc_$(basename $tree)_$c
    rm -rf $d
    (cd $tree && timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $REPO/tools/pmc_render.py > /dev/null 2> $d.err) || echo "$(basename $tree) $c: failed" >> $OUT/pmc_bisect.txt
    python3 - "$d" "$(basename $tree)" "$c" >> $OUT/pmc_bisect.txt <<'PY'
import csv, glob, os, sys
d, name, counter = sys.argv[1:4]
rows = {}
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace(" ", "")
        if r["Counter_Name"] != counter:
            continue
        for kn in ("k_trace", "k_shade"):
            if "::" + kn + "<" in k and "<true" not in k:
                rows.setdefault(kn, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for kn, v in sorted(rows.items()):
    v.sort()
    steady = [x for _, x in v[16:]]
    print("%-10s %-10s %-8s %8.1f MiB per launch (launches 16..%d, %d dispatches)" % (name, counter, kn, sum(steady) / max(1, len(steady)) / 1024.0, len(v), len(steady)))
PY
  done
done
cat $OUT/pmc_bisect.txt
