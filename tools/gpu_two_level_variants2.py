"""tools/gpu_two_level_timing.py (two-level rows only) for the in-tree library and every variants/libglaze_hip_*.so, one process each."""
import glob, os, subprocess, sys
for lib in [None] + sorted(glob.glob("variants/libglaze_hip_*.so")):
    env = dict(os.environ, ATRIUM=os.environ.get("ATRIUM", "1"))
    if lib:
        env["GLAZE_HIP_LIB"] = os.path.abspath(lib)
    o = subprocess.run([sys.executable, "tools/gpu_two_level_timing.py"], env=env, capture_output=True, text=True)
    rows = [l for l in o.stdout.splitlines() if " ms/launch" in l]
    print("%-32s %s" % (os.path.basename(lib) if lib else "in-tree", " | ".join("%s %s %s" % (l.split()[0] + l.split()[1], l.split()[2], l.split()[-2]) for l in rows) or o.stderr[-300:]), flush=True)
