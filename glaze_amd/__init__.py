"""glaze_amd -- MI355X-native render path behind glaze's Scene / parse / Renderer interface.

Python mirror of the slice of the reference's public Rust API that `glaze-cli` drives
(lib/src/lib.rs:10-24, cli/src/main.rs:76-121), implemented over the C ABI of libglaze_hip.so
(include/glaze_abi.h).  Names, argument meaning and error behaviour follow the reference:

    parsed   = glaze_amd.parse("scene.glaze")                       # glaze::parse            parser/mod.rs:93
    instance = glaze_amd.RayTraceInstance.new()                     # Option<RayTraceInstance> vulkan/instance.rs:376
    scene    = glaze_amd.RayTraceScene.new(instance, parsed)        # vulkan/scene.rs:1414
    renderer = glaze_amd.RayTraceRenderer.new(instance, scene, w, h)# vulkan/raytracer.rs:164
    renderer.set_integrator(glaze_amd.Integrator.PATH_TRACE)
    image    = renderer.draw(spp, callback)                         # HxWx4 uint8 sRGB, raytracer.rs:615

The render path has no CPU fallback: without libglaze_hip.so / a gfx950 device, RayTraceInstance.new()
returns None exactly like the reference does without a ray-tracing capable Vulkan device.
"""
import ctypes as C
import enum

import numpy as np

from . import abi
from .abi import GlazeError
from .scene_desc import SceneDesc, make_camera, make_light, make_material, make_meta  # noqa: F401

__all__ = ["parse", "converted_file", "ParsedScene", "RayTraceInstance", "RayTraceScene", "RayTraceRenderer", "Integrator",
           "GlazeError", "SceneDesc"]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Integrator(enum.Enum):
    """lib/src/vulkan/raytracer.rs:30-86"""
    DIRECT = abi.DIRECT
    PATH_TRACE = abi.PATH_TRACE

    @staticmethod
    def values():
        return [Integrator.DIRECT, Integrator.PATH_TRACE]

    def name_str(self):
        return "Direct light only" if self is Integrator.DIRECT else "Path tracing"

    def steps_per_sample(self, pt_steps=6):
        return 1 if self is Integrator.DIRECT else pt_steps


class ParsedScene:
    """`Box<dyn ParsedScene>` (lib/src/parser/mod.rs:294-323): lazy, per-chunk getters."""

    def __init__(self, handle, path):
        self._h = handle
        self.path = path

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            abi.lib().glz_parsed_free(self._h)
            self._h = None

    def _take(self):
        h, self._h = self._h, None
        if not h:
            raise ValueError("parsed scene was already consumed by RayTraceScene.new")
        return h

    def _array(self, fn, ctype):
        L = abi.lib()
        n = abi.check(fn(self._h, None, 0))
        arr = (ctype * max(1, n))()
        abi.check(fn(self._h, C.cast(arr, C.c_void_p), n))
        return arr, n

    def vertices(self):
        arr, n = self._array(abi.lib().glz_parsed_vertices, abi.Vertex)
        return np.frombuffer(arr, dtype=np.float32, count=n * 8).reshape(n, 8).copy()

    def meshes(self):
        """[(id, material, indices uint32[])]"""
        arr, n = self._array(abi.lib().glz_parsed_meshes, abi.Mesh)
        idx_arr, ni = self._array(abi.lib().glz_parsed_indices, C.c_uint32)
        idx = np.frombuffer(idx_arr, dtype=np.uint32, count=ni)
        return [dict(id=arr[i].id, material=arr[i].material,
                     indices=idx[arr[i].index_offset:arr[i].index_offset + arr[i].index_count].copy()) for i in range(n)]

    def transforms(self):
        arr, n = self._array(abi.lib().glz_parsed_transforms, abi.Transform)
        return np.frombuffer(arr, dtype=np.float32, count=n * 16).reshape(n, 16).copy()

    def instances(self):
        arr, n = self._array(abi.lib().glz_parsed_instances, abi.MeshInstance)
        return np.frombuffer(arr, dtype=np.uint16, count=n * 2).reshape(n, 2).copy()

    def cameras(self):
        arr, n = self._array(abi.lib().glz_parsed_cameras, abi.Camera)
        return [arr[i] for i in range(n)]

    def materials(self):
        arr, n = self._array(abi.lib().glz_parsed_materials, abi.Material)
        return [arr[i] for i in range(n)]

    def lights(self):
        arr, n = self._array(abi.lib().glz_parsed_lights, abi.Light)
        return [arr[i] for i in range(n)]

    def textures(self):
        """[(format, pixels HxW or HxWx4 uint8, name, mip_levels)]"""
        arr, n = self._array(abi.lib().glz_parsed_textures, abi.Texture)
        out = []
        for i in range(n):
            t = arr[i]
            ch = 1 if t.format == abi.TEX_GRAY else 4
            buf = (C.c_uint8 * (t.width * t.height * ch)).from_address(t.pixels)
            px = np.frombuffer(buf, dtype=np.uint8).copy()
            px = px.reshape(t.height, t.width) if ch == 1 else px.reshape(t.height, t.width, 4)
            out.append((t.format, px, t.name.decode("utf8", "replace"), t.mip_levels))
        return out

    def meta(self):
        m = abi.Meta()
        abi.check(abi.lib().glz_parsed_meta(self._h, C.byref(m)))
        return m

    def update(self, cameras=None, materials=None, lights=None, textures=None, meta=None):
        """ParsedScene::update (lib/src/parser/v1.rs:364-422): None keeps the chunk as stored; the file is rewritten.

        cameras / materials / lights: lists of abi.Camera / abi.Material / abi.Light;
        textures: [(format, pixels, name)] or [(format, pixels, name, mip_levels)]; meta: abi.Meta."""
        keep = []

        def arr(items, ctype):
            if items is None:
                return None, -1
            a = (ctype * max(1, len(items)))(*items)
            keep.append(a)
            return C.cast(a, C.c_void_p), len(items)

        cam, ncam = arr(cameras, abi.Camera)
        mat, nmat = arr(materials, abi.Material)
        lig, nlig = arr(lights, abi.Light)
        tex, ntex = (None, -1) if textures is None else _texture_array(textures, keep)
        abi.check(abi.lib().glz_parsed_update(self._h, cam, ncam, mat, nmat, lig, nlig, tex, ntex,
                                              C.cast(C.pointer(meta), C.c_void_p) if meta is not None else None))


def _texture_array(textures, keep):
    texs = (abi.Texture * max(1, len(textures)))()
    for i, t in enumerate(textures):
        fmt, px, name = t[0], np.ascontiguousarray(t[1], np.uint8), t[2]
        keep.append(px)
        texs[i].format, texs[i].height, texs[i].width = fmt, px.shape[0], px.shape[1]
        texs[i].mip_levels = t[3] if len(t) > 3 else 1
        texs[i].pixels = px.ctypes.data
        texs[i].name = name.encode("utf8")[:abi.NAME_MAX - 1]
    keep.append(texs)
    return C.cast(texs, C.c_void_p), len(textures)


class Serializer:
    """glaze::Serializer (lib/src/parser/mod.rs:130-233): `Serializer(path).with_vertices(v)....serialize()`."""

    def __init__(self, path):
        self.path = str(path)
        self._v = np.zeros((0, 8), np.float32)
        self._meshes, self._transforms, self._instances = [], np.zeros((0, 16), np.float32), np.zeros((0, 2), np.uint16)
        self._cameras, self._textures, self._materials, self._lights, self._meta = [], [], [], [], None

    def with_vertices(self, v):
        v = np.ascontiguousarray(v)
        self._v = (v.view(np.float32) if v.dtype.names else v.astype(np.float32, copy=False)).reshape(-1, 8)
        return self

    def with_meshes(self, meshes):
        """[dict(id, material, indices)]"""
        self._meshes = list(meshes)
        return self

    def with_transforms(self, t):
        t = np.ascontiguousarray(t)
        self._transforms = (t.view(np.float32) if t.dtype.names else t.astype(np.float32, copy=False)).reshape(-1, 16)
        return self

    def with_instances(self, i):
        i = np.ascontiguousarray(i)
        self._instances = (i.view(np.uint16) if i.dtype.names else i.astype(np.uint16, copy=False)).reshape(-1, 2)
        return self

    def with_cameras(self, c):
        self._cameras = list(c)
        return self

    def with_textures(self, t):
        self._textures = list(t)
        return self

    def with_materials(self, m):
        self._materials = list(m)
        return self

    def with_lights(self, l):
        self._lights = list(l)
        return self

    def with_metadata(self, m):
        self._meta = m
        return self

    def serialize(self):
        keep = []
        d = abi.SerializeDescC()
        idx = np.concatenate([np.asarray(m["indices"], np.uint32).ravel() for m in self._meshes]) if self._meshes else np.zeros(0, np.uint32)
        meshes = (abi.Mesh * max(1, len(self._meshes)))()
        off = 0
        for i, m in enumerate(self._meshes):
            n = int(np.asarray(m["indices"]).size)
            meshes[i].id, meshes[i].material, meshes[i].index_offset, meshes[i].index_count = m["id"], m["material"], off, n
            off += n
        d.vertices, d.n_vertices = (self._v.ctypes.data if self._v.size else None), self._v.shape[0]
        d.indices, d.n_indices = (idx.ctypes.data if idx.size else None), idx.size
        d.meshes, d.n_meshes = C.cast(meshes, C.c_void_p), len(self._meshes)
        d.transforms, d.n_transforms = (self._transforms.ctypes.data if self._transforms.size else None), self._transforms.shape[0]
        d.instances, d.n_instances = (self._instances.ctypes.data if self._instances.size else None), self._instances.shape[0]
        for name, items, ctype in (("cameras", self._cameras, abi.Camera), ("materials", self._materials, abi.Material),
                                   ("lights", self._lights, abi.Light)):
            a = (ctype * max(1, len(items)))(*items)
            keep.append(a)
            setattr(d, name, C.cast(a, C.c_void_p))
            setattr(d, "n_" + name, len(items))
        d.textures, d.n_textures = _texture_array(self._textures, keep)
        if self._meta is not None:
            d.meta = C.cast(C.pointer(self._meta), C.c_void_p)
        keep += [idx, meshes]
        abi.check(abi.lib().glz_serialize(self.path.encode(), C.byref(d)))


def parse(path):
    """glaze::parse (lib/src/parser/mod.rs:93-116).  Raises GlazeError (io::Error) on a bad file."""
    h = abi.lib().glz_parse(str(path).encode())
    if not h:
        raise abi.last_error()
    return ParsedScene(h, str(path))


def save_image(path, rgba8):
    """image.save() of the CLI (cli/src/main.rs:121): HxWx4 uint8 -> .png or .jpg by extension."""
    a = np.ascontiguousarray(rgba8, np.uint8)
    abi.check(abi.lib().glz_save_image(str(path).encode(), a.ctypes.data, a.shape[1], a.shape[0]))


def convert_obj(obj_path, out_path, gen_mipmaps=False):
    """glaze-converter for OBJ input (converter/src/main.rs:116-637, assimp-free subset).  Returns the element counts."""
    counts = (C.c_uint64 * 6)()
    abi.check(abi.lib().glz_convert_obj(str(obj_path).encode(), str(out_path).encode(), int(gen_mipmaps), counts))
    return dict(zip(("vertices", "triangles", "meshes", "materials", "textures", "lights"), (int(c) for c in counts)))


def converted_file(path):
    """lib/src/parser/mod.rs:259-271"""
    return bool(abi.lib().glz_converted_file(str(path).encode()))


class RayTraceInstance:
    """lib/src/vulkan/instance.rs:376-427"""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def new(hip_device=-1):
        """Returns None when no gfx950 device (or no libglaze_hip.so code object) is usable."""
        h = abi.lib().glz_instance_create(hip_device)
        return RayTraceInstance(h) if h else None

    @property
    def device(self):
        return abi.lib().glz_instance_device(self._h)

    @property
    def stream(self):
        """hipStream_t (as int) all kernels of this instance run on."""
        return abi.lib().glz_instance_stream(self._h)

    def set_bvh_builder(self, name):
        """'auto' (default, = 'sah': binned SAH on the GPU), 'lbvh', 'ploc' or 'sah_host' (the SAH builder's host reference) for scenes created afterwards."""
        abi.check(abi.lib().glz_instance_set_bvh_builder(self._h, {"lbvh": 0, "ploc": 1, "sah": 2, "auto": 3, "sah_host": 4}[name]))

    def set_as_levels(self, mode):
        """'auto' (two levels when instancing multiplies the triangles more than four times), 'flat' or 'two_level' for scenes created afterwards."""
        abi.check(abi.lib().glz_instance_set_as_levels(self._h, {"auto": 0, "flat": 1, "two_level": 2}[mode]))

    def __del__(self):
        if getattr(self, "_h", None):
            abi.lib().glz_instance_destroy(self._h)
            self._h = None


class RayTraceScene:
    """lib/src/vulkan/scene.rs:1352-1556"""

    def __init__(self, handle, instance, keep=None):
        self._h = handle
        self.instance = instance
        self._keep = keep
        self._owned = True

    @staticmethod
    def new(instance, parsed):
        if isinstance(parsed, SceneDesc):
            return RayTraceScene.from_desc(instance, parsed)
        h = abi.lib().glz_scene_create(instance._h, parsed._take())
        if not h:
            raise abi.last_error()
        return RayTraceScene(h, instance)

    @staticmethod
    def from_desc(instance, desc):
        c = desc.as_c()
        h = abi.lib().glz_scene_create_from_desc(instance._h, C.byref(c))
        if not h:
            raise abi.last_error()
        return RayTraceScene(h, instance)

    def info(self):
        i = abi.SceneInfo()
        abi.check(abi.lib().glz_scene_get_info(self._h, C.byref(i)))
        return i

    def camera(self):
        c = abi.Camera()
        abi.check(abi.lib().glz_scene_camera(self._h, C.byref(c)))
        return c

    # ---- parity hooks (tests) ----
    def debug_trace_closest(self, origins, dirs, tmin=1e-4):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        tri, inst = (np.zeros(n, np.uint32) for _ in range(2))
        abi.check(abi.lib().glz_debug_trace_closest(self._h, _ptr(o), _ptr(d), n, tmin, _ptr(t), _ptr(tri), _ptr(inst), _ptr(u), _ptr(v)))
        return t, tri, inst, u, v

    def debug_trace_any(self, origins, dirs, tmax, tmin=1e-3):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        tm = np.ascontiguousarray(tmax, np.float32)
        out = np.zeros(o.shape[0], np.uint8)
        abi.check(abi.lib().glz_debug_trace_any(self._h, _ptr(o), _ptr(d), _ptr(tm), o.shape[0], tmin, _ptr(out)))
        return out

    def debug_derivatives(self):
        n = abi.check(abi.lib().glz_debug_read_derivatives(self._h, None, 0))
        out = np.zeros((n, 12), np.float32)
        abi.check(abi.lib().glz_debug_read_derivatives(self._h, _ptr(out), n))
        return out

    def debug_rt_materials(self):
        n = abi.check(abi.lib().glz_debug_read_rt_materials(self._h, None, 0))
        out = np.zeros(n, np.uint8)
        abi.check(abi.lib().glz_debug_read_rt_materials(self._h, _ptr(out), n))
        return out

    def debug_rt_lights(self):
        n = abi.check(abi.lib().glz_debug_read_rt_lights(self._h, None, 0))
        out = np.zeros(n, np.uint8)
        abi.check(abi.lib().glz_debug_read_rt_lights(self._h, _ptr(out), n))
        return out

    def debug_sky(self):
        n = abi.check(abi.lib().glz_debug_read_sky(self._h, None, 0))
        out = np.zeros(n, np.float32)
        abi.check(abi.lib().glz_debug_read_sky(self._h, _ptr(out), n))
        return out

    def debug_texture_level(self, texture, level):
        """HxW[x4] uint8 pixels of one mip level as the device holds it (level 0 = the texture), or None past the last level."""
        w, h = C.c_uint32(), C.c_uint32()
        n = abi.check(abi.lib().glz_debug_read_texture_level(self._h, texture, level, None, 0, C.byref(w), C.byref(h)))
        if n == 0:
            return None
        out = np.zeros(n, np.uint8)
        abi.check(abi.lib().glz_debug_read_texture_level(self._h, texture, level, _ptr(out), n, C.byref(w), C.byref(h)))
        return out.reshape(h.value, w.value) if n == w.value * h.value else out.reshape(h.value, w.value, 4)

    def debug_bvh(self):
        i = self.info()
        nodes = np.zeros((max(1, i.bvh_nodes), 16), np.uint32)
        tris = np.zeros((max(1, i.n_as_triangles), 12), np.float32)
        abi.check(abi.lib().glz_debug_read_bvh(self._h, _ptr(nodes), i.bvh_nodes, _ptr(tris), i.n_as_triangles))
        return nodes[:i.bvh_nodes], tris[:i.n_as_triangles]

    def debug_bvh8(self):
        """The 8-wide nodes of the same hierarchy ((n, 32) uint32; empty for two-level scenes and scenes of one leaf)."""
        n = self.info().bvh_nodes8
        nodes = np.zeros((max(1, n), 32), np.uint32)
        abi.check(abi.lib().glz_debug_read_bvh8(self._h, _ptr(nodes), n))
        return nodes[:n]

    def __del__(self):
        if getattr(self, "_h", None):
            abi.lib().glz_scene_destroy(self._h)
            self._h = None


class RayTraceRenderer:
    """lib/src/vulkan/raytracer.rs:109-687"""

    def __init__(self, handle, instance, scene, width, height):
        self._h = handle
        self.instance = instance
        self.scene = scene
        self.width, self.height = width, height

    @staticmethod
    def new(instance, scene, width, height):
        h = abi.lib().glz_renderer_create(instance._h, scene._h if scene is not None else None, width, height)
        if not h:
            raise abi.last_error()
        return RayTraceRenderer(h, instance, scene, width, height)

    def __del__(self):
        if getattr(self, "_h", None):
            abi.lib().glz_renderer_destroy(self._h)
            self._h = None

    # ---- reference API ----
    def set_integrator(self, integrator):
        abi.check(abi.lib().glz_renderer_set_integrator(self._h, integrator.value if isinstance(integrator, Integrator) else int(integrator)))

    def set_exposure(self, exposure):
        abi.check(abi.lib().glz_renderer_set_exposure(self._h, exposure))

    def update_camera(self, camera):
        abi.check(abi.lib().glz_renderer_update_camera(self._h, C.byref(camera)))

    def change_resolution(self, width, height):
        abi.check(abi.lib().glz_renderer_change_resolution(self._h, width, height))
        self.width, self.height = width, height

    def change_scene(self, scene):
        abi.check(abi.lib().glz_renderer_change_scene(self._h, scene._h))
        self.scene = scene

    def update_materials_and_lights(self, materials, lights, textures=None):
        """raytracer.rs:311-326.  textures: None (keep) or [(format, pixels, name)] replacing the scene's texture array."""
        m = (abi.Material * max(1, len(materials)))(*materials)
        l = (abi.Light * max(1, len(lights)))(*lights)
        keep = []
        t, nt = (None, 0) if textures is None else _texture_array(textures, keep)
        abi.check(abi.lib().glz_renderer_update_materials_and_lights(self._h, C.cast(m, C.c_void_p), len(materials), C.cast(l, C.c_void_p),
                                                                     len(lights), t, nt))

    def refresh_binded_textures(self, textures):
        """raytracer.rs:328-356: new texture array under the same materials and lights; accumulation continues."""
        keep = []
        t, nt = _texture_array(textures, keep)
        abi.check(abi.lib().glz_renderer_refresh_binded_textures(self._h, t, nt))

    def set_texture_lod(self, mode):
        """0 = level 0 always (the reference's ray-tracing stages), 1 = ray-cone level of detail over the mip chain, 2 = ray cones
        with an anisotropic footprint (up to 16 probes along the footprint's long axis).  Restarts."""
        abi.check(abi.lib().glz_renderer_set_texture_lod(self._h, int(mode)))

    def set_devices(self, devices):
        """Render on several GPUs of this process (tiles t % n == i on devices[i], RCCL exchange onto devices[0] at every read-back);
        devices[0] must be the instance's device.  [d] returns to one device."""
        arr = (C.c_int * len(devices))(*devices)
        abi.check(abi.lib().glz_renderer_set_devices(self._h, arr, len(devices)))

    def set_launch_mode(self, mode):
        """'auto' (by the pixels this device owns), 'two_kernels' (k_trace + k_shade per launch) or 'path' (the per-wave launch loop, k_path);
        the image does not depend on it"""
        abi.check(abi.lib().glz_renderer_set_launch_mode(self._h, {"auto": 0, "two_kernels": 1, "path": 2}[mode]))

    def launch_mode(self):
        return {1: "two_kernels", 2: "path"}[int(abi.lib().glz_renderer_launch_mode(self._h))]

    def set_node_width(self, width):
        """0 (by the pixels this device owns), 4 (k_trace over the 4-wide nodes) or 8 (k_trace8 over the 8-wide nodes: small tile shares);
        the image does not depend on it"""
        abi.check(abi.lib().glz_renderer_set_node_width(self._h, int(width)))

    def node_width(self):
        return int(abi.lib().glz_renderer_node_width(self._h))

    def device_count(self):
        return int(abi.lib().glz_renderer_device_count(self._h))

    def device_scene_info(self, i):
        """scene info of the replica device i of set_devices renders (0 = this renderer's own scene)"""
        info = abi.SceneInfo()
        abi.check(abi.lib().glz_renderer_device_scene_info(self._h, i, C.byref(info)))
        return info

    def set_chains(self, n):
        """Concurrent launch chains over this rank's tiles (0 = automatic); the image does not depend on it."""
        abi.check(abi.lib().glz_renderer_set_chains(self._h, n))

    def wait_idle(self):
        abi.check(abi.lib().glz_renderer_wait_idle(self._h))

    def steps_per_sample(self):
        return abi.lib().glz_renderer_steps_per_sample(self._h)

    def draw(self, spp, callback=None, want_image=True):
        """draw(spp, callback) -> HxWx4 uint8 sRGB image (raytracer.rs:615-687)."""
        out = np.zeros((self.height, self.width, 4), np.uint8) if want_image else None
        cb = abi.DRAW_CALLBACK((lambda _u: callback()) if callback else (lambda _u: None))
        abi.check(abi.lib().glz_renderer_draw(self._h, spp, C.cast(cb, C.c_void_p) if callback else None, None, _ptr(out) if want_image else None))
        return out

    # ---- progressive API (draw_frame) ----
    def restart(self):
        abi.check(abi.lib().glz_renderer_restart(self._h))

    def step(self, n=1):
        abi.check(abi.lib().glz_renderer_step(self._h, n))

    def read_rgba8(self):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        abi.check(abi.lib().glz_renderer_read_rgba8(self._h, _ptr(out)))
        return out

    # ---- build-defined extensions ----
    def set_seed(self, seed):
        abi.check(abi.lib().glz_renderer_set_seed(self._h, seed))

    def set_depth(self, pt_steps):
        abi.check(abi.lib().glz_renderer_set_depth(self._h, pt_steps))

    def read_hdr(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        abi.check(abi.lib().glz_renderer_read_hdr(self._h, _ptr(out)))
        return out

    def read_result(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        abi.check(abi.lib().glz_renderer_read_result(self._h, _ptr(out)))
        return out

    def launch_constants(self, launch):
        s = C.c_uint32()
        off = (C.c_float * 2)()
        abi.check(abi.lib().glz_renderer_launch_constants(self._h, launch, C.byref(s), off))
        return s.value, (off[0], off[1])

    def push_constants(self):
        out = np.zeros(32, np.float32)
        abi.check(abi.lib().glz_renderer_push_constants(self._h, _ptr(out)))
        return out

    def set_partition(self, rank, world):
        abi.check(abi.lib().glz_renderer_set_partition(self._h, rank, world))

    def export_device(self, which, device_ptr):
        abi.check(abi.lib().glz_renderer_export_device(self._h, which, C.c_void_p(device_ptr)))

    def packed_pixels(self, rank, world):
        """pixels (of 4 floats) in the packed tiles of partition (rank, world) of this renderer's frame"""
        return int(abi.lib().glz_renderer_packed_pixels(self._h, rank, world))

    def export_packed(self, which, device_ptr):
        """this rank's tiles only, tile-major, into a device buffer of packed_pixels(rank, world) x 4 floats"""
        abi.check(abi.lib().glz_renderer_export_packed(self._h, which, C.c_void_p(device_ptr)))

    def scatter_packed(self, rank, world, packed_ptr, frame_ptr):
        """rank 0: puts the packed tiles received from `rank` into their place of a full-frame device buffer"""
        abi.check(abi.lib().glz_renderer_scatter_packed(self._h, rank, world, C.c_void_p(packed_ptr), C.c_void_p(frame_ptr)))

    def scatter_packed_all(self, world, packed_ptr, stride_pixels, frame_ptr):
        """the receiving side of a gather in one call: the parts of ranks 0 .. world - 1 lie `stride_pixels` pixels apart in one device buffer"""
        abi.check(abi.lib().glz_renderer_scatter_packed_all(self._h, world, C.c_void_p(packed_ptr), stride_pixels, C.c_void_p(frame_ptr)))

    def tonemap_device(self, device_ptr):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        abi.check(abi.lib().glz_renderer_tonemap_device(self._h, C.c_void_p(device_ptr), _ptr(out)))
        return out

    def enable_counters(self, counters=False, kernel_timing=True):
        abi.check(abi.lib().glz_renderer_enable_counters(self._h, (1 if counters else 0) | (2 if kernel_timing else 0)))

    def stats(self):
        s = abi.RenderStats()
        abi.check(abi.lib().glz_renderer_get_stats(self._h, C.byref(s)))
        return s
