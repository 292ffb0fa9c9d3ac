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
r.restart()
r.step(int(os.environ.get("LAUNCHES", "40")))
r.wait_idle()
