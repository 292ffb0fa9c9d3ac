#!/bin/bash
# The 48-byte node experiment next to the default build (run through gpurun): counted work per sample and the cache / issue counters of
# k_trace for both.  variants/libglaze_hip_node48.so from tools/build_variant_full.sh node48 -DGLZ_NODE48
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for lib in default node48; do
  if [ $lib = node48 ]; then export GLAZE_HIP_LIB=$REPO/variants/libglaze_hip_node48.so; else unset GLAZE_HIP_LIB; fi
  echo "== $lib"
  python bench.py --steps 64 --no-pmc --no-cpu-baseline | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('ms per step', d['ms_per_step'], 'kernels', r['kernel_ms_per_step']); print('counted per sample', {k: r['counted_per_sample'][k] for k in ('nodes_closest', 'tris_closest', 'nodes_shadow', 'tris_shadow')})"
  rm -rf gpurun_out/pmc_1 gpurun_out/pmc_2
  bash tools/pmc_gpu.sh "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES" > /dev/null 2>&1
  python tools/pmc_table.py | sed -n '/^k_trace$/,/^k_[a-z_]*$/p' | grep -v "^k_shade\|^k_tri"
done
