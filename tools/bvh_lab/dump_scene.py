"""Writes the bench scene's world-space triangles for tools/bvh_lab/bvh_lab.cpp:  python tools/bvh_lab/dump_scene.py out.bin [detail]
Layout: uint32 n_vertices, uint32 n_triangles, float32 positions[n_vertices][3], uint32 indices[n_triangles][3]  (every instance of the
atrium is the identity, the camera is given to the lab on its command line)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from glaze_amd.scenes import atrium_scene

d = atrium_scene() if len(sys.argv) < 3 else atrium_scene(detail=float(sys.argv[2]))
pos = np.ascontiguousarray(d.vertices["vv"], np.float32)
idx = np.ascontiguousarray(d.indices.reshape(-1, 3), np.uint32)
with open(sys.argv[1], "wb") as f:
    np.array([pos.shape[0], idx.shape[0]], np.uint32).tofile(f)
    pos.tofile(f)
    idx.tofile(f)
print("%d vertices, %d triangles -> %s" % (pos.shape[0], idx.shape[0], sys.argv[1]))
