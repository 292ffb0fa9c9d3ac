import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import glaze_amd
from glaze_amd import abi
from glaze_amd.scenes import cube_scene
from glaze_amd.scene_desc import make_light
from oracle.pyoracle import OracleRenderer, OracleScene
inst = glaze_amd.RayTraceInstance.new()
desc = cube_scene(material_type=abi.MAT_UBER)
desc.lights.append(make_light(abi.LIGHT_SUN, "sun", direction=(0.2, -0.7, 0.4), intensity=1.5))
w, h = 150, 83
r = glaze_amd.RayTraceRenderer.new(inst, glaze_amd.RayTraceScene.from_desc(inst, desc), w, h)
o = OracleRenderer(OracleScene(desc), w, h)
for x in (r, o):
    x.set_depth(4); x.set_seed(11)
r.set_launch_mode("two_kernels"); r.set_chains(1); o.restart()
tot = 0
out = []
for n, e in ((1, 1.0), (3, 1.0), (2, 0.5), (7, 2.0)):
    r.set_exposure(e); o.set_exposure(e)
    r.step(n); o.step(n); tot += n
    g, c = r.read_hdr(), o.read_hdr()
    d = (np.nan_to_num(g, nan=-1).view(np.uint32) != np.nan_to_num(c, nan=-1).view(np.uint32)).any(-1)
    out.append(int(d.sum()))
print(out)
