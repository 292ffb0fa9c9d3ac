"""In-memory scene description (numpy side of `glz_scene_desc`).

The reference hands a `Box<dyn ParsedScene>` to `RayTraceScene::new` (lib/src/vulkan/scene.rs:1414);
this is the same information as plain arrays, for scenes that do not come from a `.glaze` file
(the cube of BASELINE config 2, the synthetic atrium of config 4) and for tests that override
materials the way `glaze-app` does interactively.
"""
import ctypes as C

import numpy as np

from . import abi

VERTEX_DTYPE = np.dtype([("vv", "<f4", 3), ("vn", "<f4", 3), ("vt", "<f4", 2)])
MESH_DTYPE = np.dtype([("id", "<u2"), ("material", "<u2"), ("index_offset", "<u4"), ("index_count", "<u4")])
INSTANCE_DTYPE = np.dtype([("mesh_id", "<u2"), ("transform_id", "<u2")])

SPECTRUM_WHITE = np.array([1.0619347266616228, 1.0623373513955183, 1.0624330274817486, 1.0624850787200137,
                           1.0622213950288308, 1.0613081599651542, 1.0613058645182336, 1.0618168659745209,
                           1.0624642293010491, 1.0624838864140043, 1.0624682453762331, 1.0625355983287506,
                           1.0624016329348598, 1.0622653248789862, 1.060266533148627, 1.0600420908765831], np.float32)


def make_material(name="default", mtype=abi.MAT_LAMBERT, metal=0, diffuse_mul=(255, 255, 255), emissive=None, ior=1.46,
                  roughness_mul=1.0, metalness_mul=0.0, anisotropy=0.0, diffuse=0, roughness=0, metalness=0, normal=0,
                  opacity=0):
    """Material::default() (lib/src/materials/material.rs:327-345) with overrides."""
    m = abi.Material()
    m.mtype, m.metal = mtype, metal
    m.diffuse_mul[:] = diffuse_mul
    if emissive is not None and tuple(emissive) != (0, 0, 0):
        m.emissive_col[:] = emissive
        m.has_emissive = 1
    m.ior, m.roughness_mul, m.metalness_mul, m.anisotropy = ior, roughness_mul, metalness_mul, anisotropy
    m.diffuse, m.roughness, m.metalness, m.normal, m.opacity = diffuse, roughness, metalness, normal, opacity
    m.name = name.encode("utf8")[:abi.NAME_MAX - 1]
    return m


def make_light(ltype=abi.LIGHT_OMNI, name="", color=None, position=(0, 0, 0), direction=(0, -1, 0), intensity=1.0,
               resource_id=0, yaw=0.0, pitch=0.0, roll=0.0):
    """Light::default() (lib/src/geometry/light.rs:172-187) with overrides."""
    l = abi.Light()
    l.ltype = ltype
    l.position[:] = position
    l.direction[:] = direction
    l.resource_id, l.intensity = resource_id, intensity
    l.yaw_deg, l.pitch_deg, l.roll_deg = yaw, pitch, roll
    l.color[:] = (SPECTRUM_WHITE if color is None else np.asarray(color, np.float32)).tolist()
    l.name = name.encode("utf8")[:abi.NAME_MAX - 1]
    return l


def make_camera(position=(0, 0, 0), target=(0, 0, 100), up=(0, 1, 0), fovx=np.float32(np.pi / 2), near=1e-3, far=1e3,
                orthographic=False, scale=1.0):
    """PerspectiveCam::default() / OrthographicCam::default() (lib/src/geometry/camera.rs:31-63)."""
    c = abi.Camera()
    c.type = abi.CAMERA_ORTHOGRAPHIC if orthographic else abi.CAMERA_PERSPECTIVE
    c.position[:] = position
    c.target[:] = target
    c.up[:] = up
    c.fovx_or_scale = scale if orthographic else fovx
    c.near_plane, c.far_plane = near, far
    return c


def make_meta(centre=(0, 0, 0), radius=100.0, exposure=1.0):
    m = abi.Meta()
    m.scene_centre[:] = centre
    m.scene_radius, m.exposure = radius, exposure
    return m


class SceneDesc:
    """Owns the numpy / ctypes storage behind a `glz_scene_desc` and keeps it alive."""

    def __init__(self, vertices, indices, meshes, transforms=None, instances=None, materials=None, lights=None,
                 textures=None, camera=None, meta=None):
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.meshes = np.ascontiguousarray(meshes, dtype=MESH_DTYPE)
        self.transforms = np.ascontiguousarray(
            np.eye(4, dtype=np.float32).reshape(1, 16) if transforms is None else transforms, dtype=np.float32).reshape(-1, 16)
        self.instances = np.ascontiguousarray(instances if instances is not None else [], dtype=INSTANCE_DTYPE)
        self.materials = list(materials) if materials else [make_material()]
        self.lights = list(lights) if lights else []
        # textures: list of (format, HxW or HxWx4 uint8 array, name); id 0 must be the 1x1 white default
        self.textures = list(textures) if textures else [(abi.TEX_RGBA_SRGB, np.full((1, 1, 4), 255, np.uint8), "default")]
        self.camera = camera if camera is not None else make_camera()
        self.meta = meta if meta is not None else make_meta()
        self._keep = []

    def copy(self):
        return SceneDesc(self.vertices.copy(), self.indices.copy(), self.meshes.copy(), self.transforms.copy(),
                         self.instances.copy(), [_clone(m) for m in self.materials], [_clone(l) for l in self.lights],
                         [(f, a.copy(), n) for f, a, n in self.textures], _clone(self.camera), _clone(self.meta))

    @property
    def n_triangles(self):
        return int(self.indices.size // 3)

    def as_c(self):
        d = abi.SceneDescC()
        keep = []

        def arr(a):
            keep.append(a)
            return a.ctypes.data if a.size else None

        d.vertices, d.n_vertices = arr(self.vertices), self.vertices.shape[0]
        d.indices, d.n_indices = arr(self.indices), self.indices.size
        d.meshes, d.n_meshes = arr(self.meshes), self.meshes.shape[0]
        d.transforms, d.n_transforms = arr(self.transforms), self.transforms.shape[0]
        d.instances, d.n_instances = arr(self.instances), self.instances.shape[0]
        mats = (abi.Material * len(self.materials))(*self.materials)
        lights = (abi.Light * max(1, len(self.lights)))(*self.lights)
        texs = (abi.Texture * len(self.textures))()
        for i, t in enumerate(self.textures):     # (format, pixels, name[, mip levels to store when serialised])
            fmt, px, name = t[0], t[1], t[2]
            px = np.ascontiguousarray(px, np.uint8)
            keep.append(px)
            texs[i].format, texs[i].height, texs[i].width, texs[i].mip_levels = fmt, px.shape[0], px.shape[1], 1
            texs[i].pixels = px.ctypes.data
            texs[i].name = name.encode("utf8")[:abi.NAME_MAX - 1]
        keep += [mats, lights, texs]
        d.materials, d.n_materials = C.cast(mats, C.c_void_p), len(self.materials)
        d.lights, d.n_lights = C.cast(lights, C.c_void_p), len(self.lights)
        d.textures, d.n_textures = C.cast(texs, C.c_void_p), len(self.textures)
        d.camera = C.cast(C.pointer(self.camera), C.c_void_p)
        d.meta = C.cast(C.pointer(self.meta), C.c_void_p)
        self._keep = keep
        return d


def save_scene(desc, path, cameras=None):
    """Writes a SceneDesc as a `.glaze` file through glaze_amd.Serializer (lets the synthetic benchmark scenes be real files)."""
    import glaze_amd
    meshes = [dict(id=int(m["id"]), material=int(m["material"]),
                   indices=desc.indices[int(m["index_offset"]):int(m["index_offset"]) + int(m["index_count"])]) for m in desc.meshes]
    (glaze_amd.Serializer(path).with_vertices(desc.vertices).with_meshes(meshes).with_transforms(desc.transforms)
     .with_instances(desc.instances).with_cameras(cameras if cameras is not None else [desc.camera])
     .with_textures(desc.textures).with_materials(desc.materials).with_lights(desc.lights).with_metadata(desc.meta).serialize())


def _clone(s):
    c = type(s)()
    C.memmove(C.byref(c), C.byref(s), C.sizeof(s))
    return c
