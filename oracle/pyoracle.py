"""ctypes front-end of oracle/_build/liboracle.so (ORACLE: test infrastructure, not product code)."""
import ctypes as C
import os
import subprocess

import numpy as np

from glaze_amd import abi  # interface PODs only (ctypes mirrors of include/glaze_abi.h)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_P = C.c_void_p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        sig = {
            "orc_scene_create": (_P, [_P]), "orc_scene_destroy": (None, [_P]),
            "orc_scene_lights_no": (C.c_uint32, [_P]), "orc_scene_world_tris": (C.c_uint64, [_P]),
            "orc_read_derivatives": (C.c_int64, [_P, _P, C.c_int64]),
            "orc_read_rt_materials": (C.c_int64, [_P, _P, C.c_int64]),
            "orc_read_rt_lights": (C.c_int64, [_P, _P, C.c_int64]),
            "orc_read_sky": (C.c_int64, [_P, _P, C.c_int64]),
            "orc_read_sky_cond": (C.c_int64, [_P, _P, _P]),
            "orc_scene_set_ext_bvh": (None, [_P, _P, C.c_uint64, _P, C.c_uint64, _P, _P]),
            "orc_trace_closest": (None, [_P, _P, _P, C.c_uint64, C.c_float, _P, _P, _P, _P, _P]),
            "orc_trace_any": (None, [_P, _P, _P, _P, C.c_uint64, C.c_float, _P]),
            "orc_trace_closest_brute": (None, [_P, _P, _P, C.c_uint64, C.c_float, _P, _P]),
            "orc_renderer_create": (_P, [_P, C.c_uint32, C.c_uint32]), "orc_renderer_destroy": (None, [_P]),
            "orc_renderer_set_integrator": (None, [_P, C.c_int]), "orc_renderer_set_depth": (None, [_P, C.c_uint32]),
            "orc_renderer_set_seed": (None, [_P, C.c_uint64]), "orc_renderer_set_exposure": (None, [_P, C.c_float]),
            "orc_renderer_set_threads": (None, [_P, C.c_int]), "orc_renderer_set_counting": (None, [_P, C.c_int]),
            "orc_renderer_update_camera": (None, [_P, _P]),
            "orc_renderer_steps_per_sample": (C.c_uint32, [_P]),
            "orc_renderer_restart": (None, [_P]), "orc_renderer_step": (None, [_P, C.c_uint32]),
            "orc_renderer_draw": (None, [_P, C.c_uint64]),
            "orc_renderer_read_hdr": (None, [_P, _P]), "orc_renderer_read_result": (None, [_P, _P]),
            "orc_renderer_read_rgba8": (None, [_P, _P]), "orc_renderer_read_state": (None, [_P, _P]),
            "orc_renderer_push_constants": (None, [_P, _P]),
            "orc_renderer_set_tiles": (None, [_P, _P, C.c_uint32]), "orc_to_srgb8": (C.c_uint8, [C.c_float]),
            "orc_bsdf_value": (None, [_P, C.c_uint32, _P, _P, _P, _P, C.c_uint64, _P, _P]),
            "orc_bsdf_sample": (None, [_P, C.c_uint32, _P, _P, _P, C.c_uint64, _P, _P, _P]),
            "orc_light_sample": (None, [_P, C.c_uint32, _P, _P, C.c_uint64, C.c_float, _P, _P, _P, _P]),
            "orc_scene_rt_light_count": (C.c_uint32, [_P]),
            "orc_renderer_set_texture_lod": (None, [_P, C.c_int]),
            "orc_texture_level": (C.c_int64, [_P, C.c_uint32, C.c_uint32, _P, _P, _P]),
            "orc_launch_constants": (None, [C.c_uint64, C.c_uint32, _P, _P]),
            "orc_renderer_counters": (None, [_P, _P]),
            "orc_spectrum_from_rgb": (None, [C.c_float, C.c_float, C.c_float, C.c_int, _P]),
            "orc_spectrum_to_xyz": (None, [_P, _P]), "orc_spectrum_luminance": (C.c_float, [_P]),
            "orc_spectrum_from_blackbody": (None, [C.c_float, _P]),
            "orc_xyz_to_rgb": (None, [_P, _P]), "orc_rgb_to_xyz": (None, [_P, _P]),
            "orc_fovy": (C.c_float, [C.c_float, C.c_float]), "orc_spectrum_white": (None, [_P]),
            "orc_dev_from_surface_color": (None, [_P, _P]), "orc_dev_from_illuminant_color": (None, [_P, _P]),
            "orc_dev_rgb": (None, [_P, _P]), "orc_dev_luminance": (C.c_float, [_P]),
            "orc_detmath": (None, [C.c_int, _P, _P, _P, C.c_uint64]),
            "orc_pcg_hash": (C.c_uint32, [C.c_uint32]),
            "orc_rand_stream": (None, [C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_uint32]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(_P)


class OracleScene:
    def __init__(self, desc):
        """desc: glaze_amd.scene_desc.SceneDesc"""
        self.desc = desc
        self._c = desc.as_c()
        self.handle = lib().orc_scene_create(C.byref(self._c))

    def __del__(self):
        if getattr(self, "handle", None):
            lib().orc_scene_destroy(self.handle)
            self.handle = None

    @property
    def lights_no(self):
        return lib().orc_scene_lights_no(self.handle)

    @property
    def n_world_triangles(self):
        return lib().orc_scene_world_tris(self.handle)

    def derivatives(self):
        n = lib().orc_read_derivatives(self.handle, None, 0)
        out = np.zeros((n, 12), np.float32)
        lib().orc_read_derivatives(self.handle, _ptr(out), n)
        return out

    def rt_materials(self):
        n = lib().orc_read_rt_materials(self.handle, None, 0)
        out = np.zeros(n, np.uint8)
        lib().orc_read_rt_materials(self.handle, _ptr(out), n)
        return out

    def rt_lights(self):
        n = lib().orc_read_rt_lights(self.handle, None, 0)
        out = np.zeros(n, np.uint8)
        lib().orc_read_rt_lights(self.handle, _ptr(out), n)
        return out

    def sky(self):
        n = lib().orc_read_sky(self.handle, None, 0)
        out = np.zeros(n, np.float32)
        lib().orc_read_sky(self.handle, _ptr(out), n)
        return out

    def set_ext_bvh(self, nodes, tris, grid_lo, grid_cell):
        nodes = np.ascontiguousarray(nodes).view(np.uint32).reshape(-1, 16)
        tris = np.ascontiguousarray(tris).view(np.float32).reshape(-1, 12)
        glo = np.ascontiguousarray(grid_lo, np.float32)
        gcell = np.ascontiguousarray(grid_cell, np.float32)
        lib().orc_scene_set_ext_bvh(self.handle, _ptr(nodes), nodes.shape[0], _ptr(tris), tris.shape[0], _ptr(glo), _ptr(gcell))

    def trace_closest(self, origins, dirs, tmin=1e-4, brute=False):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        tri = np.zeros(n, np.uint32)
        if brute:
            lib().orc_trace_closest_brute(self.handle, _ptr(o), _ptr(d), n, tmin, _ptr(t), _ptr(tri))
            return t, tri
        inst = np.zeros(n, np.uint32)
        u = np.zeros(n, np.float32)
        v = np.zeros(n, np.float32)
        lib().orc_trace_closest(self.handle, _ptr(o), _ptr(d), n, tmin, _ptr(t), _ptr(tri), _ptr(inst), _ptr(u), _ptr(v))
        return t, tri, inst, u, v

    # ---- shading routines on their own (tests/test_oracle_math.py); directions are in the canonical shading frame ----
    def bsdf_value(self, material_id, wo, wi, uv=(0.5, 0.5), rand=None):
        wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1, 3)
        n = wo.shape[0]
        rnd = np.ascontiguousarray(rand if rand is not None else np.zeros(n), np.float32)
        uvv = np.ascontiguousarray(uv, np.float32)
        value, pdf = np.zeros((n, 16), np.float32), np.zeros(n, np.float32)
        lib().orc_bsdf_value(self.handle, material_id, _ptr(wo), _ptr(wi), _ptr(uvv), _ptr(rnd), n, _ptr(value), _ptr(pdf))
        return value, pdf

    def bsdf_sample(self, material_id, wo, rand3, uv=(0.5, 0.5)):
        wo = np.ascontiguousarray(wo, np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(rand3, np.float32).reshape(-1, 3)
        n = wo.shape[0]
        uvv = np.ascontiguousarray(uv, np.float32)
        wi, value, pdf = np.zeros((n, 3), np.float32), np.zeros((n, 16), np.float32), np.zeros(n, np.float32)
        lib().orc_bsdf_sample(self.handle, material_id, _ptr(wo), _ptr(uvv), _ptr(r), n, _ptr(wi), _ptr(value), _ptr(pdf))
        return wi, value, pdf

    def light_sample(self, light_index, positions, rand3, scene_radius=1.0):
        p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        r = np.ascontiguousarray(rand3, np.float32).reshape(-1, 3)
        n = p.shape[0]
        wi, dist, pdf, em = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 16), np.float32)
        lib().orc_light_sample(self.handle, light_index, _ptr(p), _ptr(r), n, scene_radius, _ptr(wi), _ptr(dist), _ptr(pdf), _ptr(em))
        return wi, dist, pdf, em

    @property
    def n_rt_lights(self):
        return lib().orc_scene_rt_light_count(self.handle)

    def texture_level(self, texture, level):
        """pixels of one level of the generated mip chain (level 0 = the texture), or None past the last level"""
        w, h = C.c_uint32(), C.c_uint32()
        n = lib().orc_texture_level(self.handle, texture, level, None, C.byref(w), C.byref(h))
        if n == 0:
            return None
        out = np.zeros(n, np.uint8)
        lib().orc_texture_level(self.handle, texture, level, _ptr(out), C.byref(w), C.byref(h))
        return out.reshape(h.value, w.value) if n == w.value * h.value else out.reshape(h.value, w.value, 4)

    def sky_cond(self):
        """(conditional values H x W, conditional cdf H x (W+1)) of the sky distribution"""
        n = lib().orc_read_sky_cond(self.handle, None, None)
        hdr = self.sky()
        h = int(hdr[36:40].view(np.uint32)[0]) - 1
        w = n // h if h > 0 else 0
        values, cdf = np.zeros((h, w), np.float32), np.zeros((h, w + 1), np.float32)
        if n:
            lib().orc_read_sky_cond(self.handle, _ptr(values), _ptr(cdf))
        return values, cdf

    def trace_any(self, origins, dirs, tmax, tmin=1e-3):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        tm = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros(o.shape[0], np.uint8)
        lib().orc_trace_any(self.handle, _ptr(o), _ptr(d), _ptr(tm), o.shape[0], tmin, _ptr(out))
        return out


class OracleRenderer:
    def __init__(self, scene, width, height, threads=None):
        self.scene = scene
        self.w, self.h = width, height
        self.handle = lib().orc_renderer_create(scene.handle, width, height)
        lib().orc_renderer_set_threads(self.handle, threads or max(1, min(len(os.sched_getaffinity(0)), 16)))

    def __del__(self):
        if getattr(self, "handle", None):
            lib().orc_renderer_destroy(self.handle)
            self.handle = None

    def set_threads(self, n):
        lib().orc_renderer_set_threads(self.handle, n)

    def set_integrator(self, i):
        lib().orc_renderer_set_integrator(self.handle, i)

    def set_depth(self, d):
        lib().orc_renderer_set_depth(self.handle, d)

    def set_seed(self, s):
        lib().orc_renderer_set_seed(self.handle, s)

    def set_exposure(self, e):
        lib().orc_renderer_set_exposure(self.handle, e)

    def set_counting(self, on):
        lib().orc_renderer_set_counting(self.handle, int(on))

    def update_camera(self, cam):
        lib().orc_renderer_update_camera(self.handle, C.byref(cam))

    def set_texture_lod(self, mode):
        lib().orc_renderer_set_texture_lod(self.handle, int(mode))

    def set_tiles(self, tiles):
        """Render only these 64x64 tiles (row-major ids); [] = the whole frame.  Pixels outside stay zero."""
        t = np.ascontiguousarray(tiles, np.uint32)
        lib().orc_renderer_set_tiles(self.handle, _ptr(t), t.size)

    def steps_per_sample(self):
        return lib().orc_renderer_steps_per_sample(self.handle)

    def restart(self):
        lib().orc_renderer_restart(self.handle)

    def step(self, n=1):
        lib().orc_renderer_step(self.handle, n)

    def draw(self, spp):
        lib().orc_renderer_draw(self.handle, spp)

    def read_hdr(self):
        out = np.zeros((self.h, self.w, 4), np.float32)
        lib().orc_renderer_read_hdr(self.handle, _ptr(out))
        return out

    def read_result(self):
        out = np.zeros((self.h, self.w, 4), np.float32)
        lib().orc_renderer_read_result(self.handle, _ptr(out))
        return out

    def read_rgba8(self):
        out = np.zeros((self.h, self.w, 4), np.uint8)
        lib().orc_renderer_read_rgba8(self.handle, _ptr(out))
        return out

    def read_state(self):
        out = np.zeros((self.h, self.w, 24), np.float32)
        lib().orc_renderer_read_state(self.handle, _ptr(out))
        return out

    def push_constants(self):
        out = np.zeros(32, np.float32)
        lib().orc_renderer_push_constants(self.handle, _ptr(out))
        return out

    def counters(self):
        out = np.zeros(7, np.uint64)
        lib().orc_renderer_counters(self.handle, _ptr(out))
        return dict(zip(["closest_rays", "shadow_rays", "closest_nodes", "closest_tris", "shadow_nodes", "shadow_tris", "hits"],
                        (int(x) for x in out)))


def launch_constants(seed, launch):
    s = C.c_uint32()
    off = (C.c_float * 2)()
    lib().orc_launch_constants(seed, launch, C.byref(s), off)
    return s.value, (off[0], off[1])


def detmath(fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else x, np.float32)
    out = np.zeros_like(x)
    lib().orc_detmath({"sin": 0, "cos": 1, "acos": 2, "atan2": 3, "log2": 4}[fn], _ptr(x), _ptr(y), _ptr(out), x.size)
    return out


def desc_from_file(path):
    """The scene description the oracle renders, read from a `.glaze` file by the ORACLE's own python reader (oracle/glaze_v1.py) --
    independent of the product's C++ reader.  Used by the tests and by bench.py's cpu_baseline leg for a supplied scene file."""
    from glaze_amd import abi  # noqa: F401  (POD layouts only)
    from glaze_amd.scene_desc import (INSTANCE_DTYPE, MESH_DTYPE, VERTEX_DTYPE, SceneDesc, make_camera, make_light, make_material, make_meta)
    from oracle.glaze_v1 import parse
    p = parse(path)
    v = p.vertices()
    vertices = np.zeros(v.shape[0], VERTEX_DTYPE)
    vertices["vv"], vertices["vn"], vertices["vt"] = v[:, 0:3], v[:, 3:6], v[:, 6:8]
    meshes, indices, off = [], [], 0
    for m in p.meshes():
        meshes.append((m["id"], m["material"], off, m["indices"].size))
        indices.append(m["indices"])
        off += m["indices"].size
    mats = [make_material(m["name"], m["mtype"], m["metal"], m["diffuse_mul"], m["emissive"], m["ior"], m["roughness_mul"],
                          m["metalness_mul"], m["anisotropy"], m["diffuse"], m["roughness"], m["metalness"], m["normal"], m["opacity"])
            for m in p.materials()]
    lights = [make_light(l["ltype"], l["name"], l["color"], l["position"], l["direction"], l["intensity"], l["resource_id"],
                         l["yaw"], l["pitch"], l["roll"]) for l in p.lights()]
    textures = [(t["format"], t["levels"][0], t["name"]) for t in p.textures()]
    cams = p.cameras()
    cam = None
    if cams:
        c = cams[-1]
        cam = make_camera(c["position"], c["target"], c["up"], c["fovx_or_scale"], c["near"], c["far"],
                          orthographic=c["type"] == 1, scale=c["fovx_or_scale"])
    meta = p.meta()
    meta = make_meta(meta["scene_centre"], meta["scene_radius"], meta["exposure"]) if meta else None
    inst = np.array([tuple(x) for x in p.instances()], INSTANCE_DTYPE)
    return SceneDesc(vertices, np.concatenate(indices) if indices else np.zeros(0, np.uint32), np.array(meshes, MESH_DTYPE),
                     p.transforms(), inst, mats, lights, textures, cam, meta)
