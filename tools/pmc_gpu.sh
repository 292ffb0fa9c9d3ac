#!/bin/bash
# PMC deep-dive passes (run through gpurun).  Usage: bash tools/pmc_gpu.sh "<counters pass 1>" "<counters pass 2>" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  # (a counter set the hardware cannot collect at once makes rocprofv3 abort and then hang: bounded)
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$i -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/pmc_$i.json 2> $OUT/pmc_$i.err || echo "pass $i failed"
done
