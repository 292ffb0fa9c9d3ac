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
for n in (1, 3, 2, 2):
    r.step(n); o.step(n)
    g, c = r.read_hdr(), o.read_hdr()
    d = (np.nan_to_num(g, nan=-1).view(np.uint32) != np.nan_to_num(c, nan=-1).view(np.uint32)).any(-1)
    diff = (g[..., :3].sum(-1) - c[..., :3].sum(-1))[d]
    print("launches so far %d: %d differ; brighter %d, darker %d; count channel equal: %s; mean diff %.4g" % (int(g[..., 3].max()), int(d.sum()), int((diff > 0).sum()), int((diff < 0).sum()), bool(np.array_equal(g[..., 3], c[..., 3])), float(diff.mean()) if diff.size else 0.0))
