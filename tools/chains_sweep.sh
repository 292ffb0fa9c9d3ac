for c in 2 3 4 5; do echo "CHAINS=$c"; CHAINS=$c GPU_MAX_HW_QUEUES=8 python tools/gpu_partition_timing.py two_kernels 2>&1 | grep -E "world +(4|6|8|16) "; done
