"""tools/gpu_two_level_timing.py (forest only) for the in-tree build and every variants/libglaze_hip_*.so, each in its own process."""
import glob, os, subprocess, sys
libs = [None] + sorted(glob.glob("variants/libglaze_hip_*.so"))
for lib in libs:
    env = dict(os.environ, ATRIUM="0")
    if lib:
        env["GLAZE_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "tools/gpu_two_level_timing.py"], env=env, capture_output=True, text=True)
    print("== %s" % (os.path.basename(lib) if lib else "in-tree"))
    print(out.stdout.strip() or out.stderr.strip()[-400:], flush=True)
