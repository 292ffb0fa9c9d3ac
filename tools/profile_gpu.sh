#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + four PMC passes of the bench command (HBM reads, HBM writes, SQ issue
# counters, cache / texture-unit counters: separate passes, --kernel-trace only).
# Outputs under gpurun_out/prof_*/ ; summarise with tools/summarize_profiles.py and commit under profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
STEPS=${STEPS:-64}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 $REPO/bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-pmc --no-extras > $OUT/prof_trace.json 2> $OUT/prof_trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmc --no-extras > $OUT/prof_fetch.json 2> $OUT/prof_fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmc --no-extras > $OUT/prof_write.json 2> $OUT/prof_write.err || exit 3
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/prof_sq -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmc --no-extras > $OUT/prof_sq.json 2> $OUT/prof_sq.err || exit 4
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/prof_cache -- python3 $REPO/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pmc --no-extras > $OUT/prof_cache.json 2> $OUT/prof_cache.err || exit 5
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq $OUT/prof_cache -name "*.csv" | head -30
