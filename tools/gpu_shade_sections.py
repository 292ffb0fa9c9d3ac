"""Where a wave of k_shade spends its time (shader clocks, summed over the waves of the launches taken): staging, the regrouping key and
sort, the stations of shade_pixel, the epilogue -- at N = 1 and on a 1/8 share.  Needs -DGLZ_SECTION_TIMES:
    tools/build_variant.sh sections -DGLZ_SECTION_TIMES;  GLAZE_HIP_LIB=variants/libglaze_hip_sections.so python tools/gpu_shade_sections.py"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import glaze_amd
from glaze_amd import abi
from glaze_amd.scenes import atrium_scene
inst = glaze_amd.RayTraceInstance.new()
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, atrium_scene()), 1920, 1080)
r.set_depth(8)
NAMES = ["stage tables", "key (hit -> material)", "regroup", "record + material", "textures + frame", "light sample", "BSDF eval + radiance", "queue / accumulate",
         "roulette + BSDF sample", "epilogue (state out)"]
for world in (1, 8):
    r.set_partition(0, world); r.set_launch_mode("two_kernels")
    r.restart(); r.step(32); r.wait_idle()
    buf = np.zeros((4096, 16), np.uint64)
    assert abi.lib().glz_debug_shade_sections(buf.ctypes.data_as(C.c_void_p), 1) == 0
    r.step(16); r.wait_idle()
    assert abi.lib().glz_debug_shade_sections(buf.ctypes.data_as(C.c_void_p), 1) == 0
    b = buf.astype(np.float64)
    b = b[b[:, 15] > 0]
    per = (b[:, :10] / b[:, 15:16]).mean(axis=0)                 # clocks per wave and launch
    tot = (b[:, :10] / b[:, 15:16]).sum(axis=1)
    print("world %d: %d waves sampled, %.0f clocks per wave and launch (p90 %.0f, max %.0f): " % (world, len(b), per.sum(), np.percentile(tot, 90), tot.max())
          + ", ".join("%s %.0f (%.0f %%)" % (NAMES[k], per[k], 100 * per[k] / per.sum()) for k in range(10)), flush=True)
