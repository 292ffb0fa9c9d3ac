#!/bin/bash
# kernel trace of rank 0's 1/8 share in the two-kernel mode (three chains) -> per-queue durations and gaps (tools/gpu_timeline.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
rm -rf $REPO/gpurun_out/prof_share
cat > /tmp/share_run.py <<PY
import sys
sys.path.insert(0, "$REPO")
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8); r.set_partition(0, 8); r.set_launch_mode("two_kernels"); r.enable_counters(False, False)
r.restart(); r.step(32); r.wait_idle(); r.step(128); r.wait_idle()
PY
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_share -- python3 /tmp/share_run.py > /dev/null 2>&1
python3 $REPO/tools/gpu_timeline.py $(find $REPO/gpurun_out/prof_share -name "*_kernel_trace.csv" | head -1)
