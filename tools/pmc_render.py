"""The render loop a `rocprofv3 --pmc` pass profiles when the tree it runs in may be an OLDER checkout (write-traffic bisection, tools/pmc_bisect.sh):
the atrium as that tree builds it, 1080p, depth 8, LAUNCHES launches after a restart.  Imports glaze_amd from the CURRENT DIRECTORY."""
import os
import sys
sys.path.insert(0, os.getcwd())
import glaze_amd
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
kw = {"texture_size": 512} if "texture_size" in atrium_scene.__code__.co_varnames else {}
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene(**kw)), 1920, 1080)
r.set_depth(8)
if os.environ.get("WORLD"):                     # rank 0's share of an N-way partition, one chain, two kernels (tools/pmc_node_width.sh)
    r.set_partition(0, int(os.environ["WORLD"]))
    r.set_launch_mode("two_kernels")
    r.set_chains(1)
if os.environ.get("NODE_WIDTH") and hasattr(r, "set_node_width"):
    r.set_node_width(int(os.environ["NODE_WIDTH"]))
r.restart()
r.step(int(os.environ.get("LAUNCHES", "40")))
r.wait_idle()
