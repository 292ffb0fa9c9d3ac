#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + two PMC passes of the bench command.
# Outputs under gpurun_out/prof_*/ ; summarise with tools/summarize_profiles.py and commit under profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
STEPS=${STEPS:-64}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 $REPO/bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-pmc > $OUT/prof_trace.json 2> $OUT/prof_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmc > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmc > $OUT/prof_write.json 2> $OUT/prof_write.err || exit 3
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -30
