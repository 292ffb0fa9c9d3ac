"""Randomised parity run (GPU box; the test suite runs a short version, tests/test_gpu_fuzz.py): the scenes of tests/fuzz_scenes.py
rendered by the HIP path (random launch mode, acceleration-structure levels, chains) and by the CPU oracle with the same seed and
launch sequence, compared bit for bit.

    python tools/gpu_fuzz_parity.py [first_seed] [count]        -> one line per scene, a summary, exit code 1 on any difference
    python tools/gpu_fuzz_parity.py 118,188,297                 -> those seeds again
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")


from fuzz_scenes import random_scene, render_both


def bits(a):
    """the image's words with every NaN made the same one (a NaN's sign and payload are not part of the contract)"""
    return np.nan_to_num(a, nan=-1.0).view(np.uint32)


def matrix(seeds):
    """every launch mode x acceleration-structure level x chain count for the given seeds, against the oracle: where does a difference live?"""
    for seed in seeds:
        desc, run = random_scene(seed)
        print("seed %d (%s)" % (seed, run))
        for levels in ("flat", "two_level"):
            for mode in ("two_kernels", "path"):
                for chains in (1, 2, 3):
                    r, o = render_both(desc, dict(run, chains=chains), levels, mode)
                    g, gr, c, cr = r.read_hdr(), r.read_result(), o.read_hdr(), o.read_result()
                    print("   %-9s %-11s chains %d: hdr differs in %4d pixels, result in %4d (of them only alpha: %d)" % (
                        levels, mode, chains, int((bits(g) != bits(c)).any(-1).sum()), int((bits(gr) != bits(cr)).any(-1).sum()),
                        int(((bits(gr) != bits(cr)).any(-1) & ~(bits(gr)[..., :3] != bits(cr)[..., :3]).any(-1)).sum())), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--matrix":
        return matrix([int(v) for v in sys.argv[2].split(",")])
    if len(sys.argv) > 1 and "," in sys.argv[1] or len(sys.argv) == 2:
        seeds = [int(v) for v in sys.argv[1].split(",")]          # a list of seeds to look at again
    else:
        first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
        seeds = list(range(first, first + (int(sys.argv[2]) if len(sys.argv) > 2 else 100)))
    count = len(seeds)
    bad = []
    t0 = time.time()
    for seed in seeds:
        desc, run = random_scene(seed)
        try:
            r, o = render_both(desc, run)
            scene = None
            g, gr = r.read_hdr(), r.read_result()
            c, cr = o.read_hdr(), o.read_result()
            same = np.array_equal(bits(g), bits(c)) and np.array_equal(bits(gr), bits(cr)) and np.array_equal(r.read_rgba8(), o.read_rgba8())
            verdict = "identical"
            if not same:
                d = (bits(g) != bits(c)).any(-1) | (bits(gr) != bits(cr)).any(-1)
                ys, xs = np.nonzero(d)
                verdict = "DIFFERENT in %d of %d pixels" % (int(d.sum()), d.size)
                for y, x in list(zip(ys, xs))[:4]:
                    verdict += "\n      (x %d, y %d): hip %s -> %s | oracle %s -> %s" % (x, y, g[y, x].tolist(), gr[y, x, :3].tolist(), c[y, x].tolist(), cr[y, x, :3].tolist())
        except Exception as e:   # an error on one side only is a finding as well
            same, verdict = False, "ERROR %s: %s" % (type(e).__name__, str(e)[:200])
        print("seed %4d  %3dx%-3d spp %d depth %d %-11s %-11s levels %-9s chains %d %-8s lod %d exp %d  tris %5d inst %2d lights %d mats %d tex %d  nan %5.1f %%  %s" % (
            seed, run["w"], run["h"], run["spp"], run["depth"], run["integrator"].name, run["mode"], run["levels"], run["chains"], run["builder"], run["lod"], run["exposure_after"], desc.n_triangles,
            len(desc.instances), len(desc.lights), len(desc.materials), len(desc.textures), 100 * float(np.isnan(g).any(-1).mean()) if same or "g" in dir() else -1, verdict), flush=True)
        if not same:
            bad.append(seed)
    print("%d scenes in %.0f s: %d identical, %d not%s" % (count, time.time() - t0, count - len(bad), len(bad), (" -> seeds " + " ".join(map(str, bad))) if bad else ""))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
