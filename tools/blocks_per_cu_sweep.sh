#!/bin/bash
# Run on the GPU box: k_trace's persistent grid capped at 1..6 blocks per CU (GLAZE_TRACE_BLOCKS_PER_CU) x 1..3 chains, two-kernel mode,
# rank 0's share of the 1080p atrium at every world size (tools/gpu_partition_timing.py).
for c in 1 2 3; do
  for b in 1 2 3 4 6; do
    echo "== chains $c, blocks per CU <= $b"
    CHAINS=$c GLAZE_TRACE_BLOCKS_PER_CU=$b python tools/gpu_partition_timing.py two_kernels | sed 's/ wall.*-> / -> /'
  done
done
