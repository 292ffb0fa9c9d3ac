"""One fuzz seed with one setting of its run varied: python tools/gpu_fuzz_vary.py SEED key=v1,v2,... [key2=...]   (values through eval)"""
import itertools
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from fuzz_scenes import random_scene, render_both


def bits(a):
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


seed = int(sys.argv[1])
keys = [a.split("=")[0] for a in sys.argv[2:]]
values = [[eval(v) if not v.isalpha() and "_" not in v else v for v in a.split("=")[1].split(",")] for a in sys.argv[2:]]
desc, run = random_scene(seed)
for combo in itertools.product(*values):
    rn = dict(run, **dict(zip(keys, combo)))
    r, o = render_both(desc, rn)
    g, c = r.read_hdr(), o.read_hdr()
    d = (bits(g) != bits(c)).any(-1)
    ys, xs = np.nonzero(d)
    print(dict(zip(keys, combo)), "differs in %d pixels" % int(d.sum()), [(int(x), int(y), g[y, x].tolist(), c[y, x].tolist()) for y, x in list(zip(ys, xs))[:2]])
