"""Shared helpers for the tests (scene conversion, comparison metrics)."""
import numpy as np

from glaze_amd import abi
from glaze_amd.scene_desc import (INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_camera, make_light, make_material,
                                  make_meta)


def desc_from_oracle_parse(path):
    """SceneDesc built from the ORACLE's python reader (oracle/glaze_v1.py) -- independent of the C++ reader."""
    from oracle.pyoracle import desc_from_file
    return desc_from_file(path)


def camera_rays(push, width, height, offset=(0.5, 0.5)):
    """Perspective camera rays as the raygen stage builds them (numpy float32, not bit-exact; for hit tests only)."""
    c2w = push[:16].reshape(4, 4).T.astype(np.float64)
    s2c = push[16:].reshape(4, 4).T.astype(np.float64)
    y, x = np.mgrid[0:height, 0:width]
    ndc = np.stack([-1 + 2 * (x + offset[0]) / width, -1 + 2 * (y + offset[1]) / height], -1)
    tgt = np.einsum("ij,hwj->hwi", s2c, np.concatenate([ndc, np.ones_like(ndc)], -1))[..., :3]
    tgt /= np.linalg.norm(tgt, axis=-1, keepdims=True)
    d = np.einsum("ij,hwj->hwi", c2w[:3, :3], tgt)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    return o.reshape(-1, 3).astype(np.float32), d.reshape(-1, 3).astype(np.float32)


def rel_err(a, b):
    """|a-b| / max(|b|, floor) per element, with a floor tied to the image scale."""
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    floor = max(1e-12, 1e-3 * float(np.abs(b).mean()))
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


class DeviceArray:
    """A float32 array in device memory for tests that hand device pointers through the C ABI, allocated with the HIP runtime
    libglaze_hip.so itself is linked to (a test process must not load torch next to it: torch brings its own copy of the runtime,
    and two of them in one process end in a double free at exit)."""
    _hip = None

    def __init__(self, shape, fill=0.0):
        import ctypes as C
        if DeviceArray._hip is None:
            abi.lib()                                                    # makes sure the runtime is the library's
            DeviceArray._hip = C.CDLL("libamdhip64.so.7")
        self.shape = tuple(shape)
        self.nbytes = int(np.prod(self.shape)) * 4
        p = C.c_void_p()
        assert DeviceArray._hip.hipMalloc(C.byref(p), C.c_size_t(self.nbytes)) == 0
        self.ptr = p.value
        self.upload(np.full(self.shape, fill, np.float32))

    def upload(self, a):
        import ctypes as C
        a = np.ascontiguousarray(a, np.float32)
        assert a.nbytes == self.nbytes
        assert DeviceArray._hip.hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes), 1) == 0

    def numpy(self):
        import ctypes as C
        out = np.empty(self.shape, np.float32)
        assert DeviceArray._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(self.nbytes), 2) == 0
        return out

    def __del__(self):
        import ctypes as C
        if getattr(self, "ptr", None) and DeviceArray._hip is not None:
            DeviceArray._hip.hipFree(C.c_void_p(self.ptr))
            self.ptr = None
