#!/bin/bash
# VALU instructions, busy cycles and fetched bytes per launch of the traversal kernel over the 4-wide (k_trace) and the 8-wide nodes (k_trace8), full frame and a
# 1/8 share (one chain): one rocprofv3 --pmc pass each.   tools/pmc_node_width.sh   (through gpurun; output gpurun_out/r05/pmc_node_width.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/pmc_node_width.txt
for width in 4 8; do
  for world in 1 8; do
    for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "FETCH_SIZE"; do
      d=/tmp/pmc_w${width}_n${world}_$(echo $c | cut -d' ' -f1)
      rm -rf $d
      (cd $REPO && NODE_WIDTH=$width WORLD=$world timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $REPO/tools/pmc_render.py > /dev/null 2> $d.err) || echo "width $width world $world: failed" >> $OUT/pmc_node_width.txt
      python3 - "$d" "$width" "$world" >> $OUT/pmc_node_width.txt <<'PY'
import csv, glob, os, sys
d, width, world = sys.argv[1:4]
rows = {}
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace(" ", "")
        if "::k_trace" in k and "<true" not in k:
            rows.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for c, v in sorted(rows.items()):
    v.sort()
    steady = [x for _, x in v[16:]]
    print("%s-wide nodes, 1/%s of the frame: %-22s %14.0f per launch (%d dispatches)" % (width, world, c, sum(steady) / max(1, len(steady)), len(steady)))
PY
    done
  done
done
cat $OUT/pmc_node_width.txt
