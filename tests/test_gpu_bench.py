"""bench.py end to end on a supplied scene file (`--scene` / GLAZE_BENCH_SCENE): BASELINE configs[3] reads "Sponza .glaze" and the reference's
README links one (cli/src/main.rs:76-121 loads whatever file it is given); the metric must be measurable on it without editing the bench."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MATTEST = os.path.join(ROOT, "tests", "golden", "mattest.glaze")
CUBE_OBJ = os.path.join(ROOT, "tests", "golden", "cube.obj")
pytestmark = pytest.mark.gpu


def run_bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--width", "256", "--height", "192", "--no-pmc", "--cpu-launches", "2", *extra]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GLAZE_BENCH_SCENE")}
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(e, **(env or {})), timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1]), p.stderr


def check_line(out, name, triangles):
    assert out["data"] == "file:" + name and out["config"]["scene"] == name and name in out["config"]["workload"]
    assert out["config"]["triangles"] == triangles and out["config"]["bvh"]["nodes"] > 0
    assert out["value"] > 0 and out["unit"] == "Msamples/s" and out["n_gpus"] == 1 and out["vs_baseline"] is None
    r = out["roofline"]
    assert r["achieved"] > 0 and r["peak"] == 8000.0 and 0 < r["frac"] and r["kernel"] in ("k_trace", "k_shade", "k_path")
    assert r["roofline_class"] == "hbm" and r["bound"] in ("hbm", "valu-issue", "memory-latency")
    c = out["cpu_baseline"]
    assert c["value"] > 0 and c["kind"] == "port" and c["cores"] >= 1 and name in c["sample"]


def test_bench_on_a_supplied_glaze_file():
    """mattest.glaze (the reference's own fixture) instead of the atrium: a full line -- roofline and cpu_baseline included, the oracle
    reading the same file with its own reader -- whose data / workload fields name the file; triangle count and BVH figures come from
    scene.info()."""
    out, _ = run_bench("--scene", MATTEST)
    check_line(out, "mattest.glaze", 138480)
    assert out["roofline"]["counted_per_sample"]["f_hit"] > 0.1      # the camera of the file looks at the scene
    # the environment variable does the same
    out2, _ = run_bench("--no-cpu-baseline", env={"GLAZE_BENCH_SCENE": MATTEST})
    assert out2["data"] == "file:mattest.glaze" and out2["cpu_baseline"] is None


def test_bench_on_an_obj_through_the_converter():
    """an .obj goes through glz_convert_obj first (what `glaze-converter` does for the reference, converter/src/main.rs).  cube.obj carries
    no light, so the launches are empty -- the reference's raygen returns at once for such a scene, path_trace.rgen:137-141 -- and
    bench.py says so instead of printing a silent number."""
    out, err = run_bench("--scene", CUBE_OBJ, "--no-cpu-baseline")
    assert out["data"] == "file:cube.obj" and out["config"]["triangles"] == 12
    assert "has no lights" in err
