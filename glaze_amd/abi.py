"""ctypes mirror of include/glaze_abi.h and loader for libglaze_hip.so.

This is the Python-side stub of the C ABI (the Rust `-sys` binding a glaze maintainer would write
is shown in INTEGRATION.md).  There is NO fallback: if the shared library (and with it the HIP
code object) is missing, importing the render path raises.
"""
import ctypes as C
import os

NAME_MAX = 256

# status codes
OK, E_IO, E_INVALID_INPUT, E_INVALID_DATA, E_ARG, E_DEVICE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6

CAMERA_PERSPECTIVE, CAMERA_ORTHOGRAPHIC = 0, 1
MAT_FLAT, MAT_LAMBERT, MAT_MIRROR, MAT_GLASS, MAT_METAL, MAT_FROSTED, MAT_UBER = range(7)
LIGHT_OMNI, LIGHT_SUN, LIGHT_AREA, LIGHT_SKY = range(4)
TEX_GRAY, TEX_RGBA_SRGB, TEX_RGBA_NORM = 1, 2, 3
DIRECT, PATH_TRACE = 0, 1


class Vertex(C.Structure):
    _fields_ = [("vv", C.c_float * 3), ("vn", C.c_float * 3), ("vt", C.c_float * 2)]


class Mesh(C.Structure):
    _fields_ = [("id", C.c_uint16), ("material", C.c_uint16), ("index_offset", C.c_uint32), ("index_count", C.c_uint32)]


class Transform(C.Structure):
    _fields_ = [("m", C.c_float * 16)]


class MeshInstance(C.Structure):
    _fields_ = [("mesh_id", C.c_uint16), ("transform_id", C.c_uint16)]


class Camera(C.Structure):
    _fields_ = [("type", C.c_uint32), ("position", C.c_float * 3), ("target", C.c_float * 3), ("up", C.c_float * 3),
                ("fovx_or_scale", C.c_float), ("near_plane", C.c_float), ("far_plane", C.c_float)]


class Material(C.Structure):
    _fields_ = [("mtype", C.c_uint8), ("metal", C.c_uint8), ("diffuse_mul", C.c_uint8 * 3), ("emissive_col", C.c_uint8 * 3),
                ("has_emissive", C.c_uint8), ("_pad", C.c_uint8 * 3),
                ("ior", C.c_float), ("roughness_mul", C.c_float), ("metalness_mul", C.c_float), ("anisotropy", C.c_float),
                ("diffuse", C.c_uint16), ("roughness", C.c_uint16), ("metalness", C.c_uint16), ("normal", C.c_uint16),
                ("opacity", C.c_uint16), ("_pad2", C.c_uint16), ("name", C.c_char * NAME_MAX)]


class Light(C.Structure):
    _fields_ = [("ltype", C.c_uint32), ("position", C.c_float * 3), ("direction", C.c_float * 3), ("resource_id", C.c_uint32),
                ("intensity", C.c_float), ("yaw_deg", C.c_float), ("pitch_deg", C.c_float), ("roll_deg", C.c_float),
                ("color", C.c_float * 16), ("name", C.c_char * NAME_MAX)]


class Texture(C.Structure):
    _fields_ = [("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("mip_levels", C.c_uint32),
                ("pixels", C.c_void_p), ("name", C.c_char * NAME_MAX)]


class Meta(C.Structure):
    _fields_ = [("scene_centre", C.c_float * 3), ("scene_radius", C.c_float), ("exposure", C.c_float)]


class SceneDescC(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint64),
                ("indices", C.c_void_p), ("n_indices", C.c_uint64),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint32),
                ("transforms", C.c_void_p), ("n_transforms", C.c_uint32),
                ("instances", C.c_void_p), ("n_instances", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32),
                ("textures", C.c_void_p), ("n_textures", C.c_uint32),
                ("camera", C.c_void_p), ("meta", C.c_void_p)]


class SerializeDescC(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint64), ("indices", C.c_void_p), ("n_indices", C.c_uint64),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint64), ("transforms", C.c_void_p), ("n_transforms", C.c_uint64),
                ("instances", C.c_void_p), ("n_instances", C.c_uint64), ("cameras", C.c_void_p), ("n_cameras", C.c_uint64),
                ("textures", C.c_void_p), ("n_textures", C.c_uint64), ("materials", C.c_void_p), ("n_materials", C.c_uint64),
                ("lights", C.c_void_p), ("n_lights", C.c_uint64), ("meta", C.c_void_p)]


class SceneInfo(C.Structure):
    _fields_ = [("n_vertices", C.c_uint64), ("n_triangles", C.c_uint64), ("n_world_triangles", C.c_uint64),
                ("n_instances", C.c_uint32), ("n_materials", C.c_uint32), ("n_lights", C.c_uint32), ("n_rt_lights", C.c_uint32),
                ("n_textures", C.c_uint32), ("bvh_nodes", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("bvh_sah_cost", C.c_float), ("build_ms", C.c_float), ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("bvh_grid_lo", C.c_float * 3), ("bvh_grid_cell", C.c_float * 3),
                ("as_levels", C.c_uint32), ("n_as_triangles", C.c_uint32), ("as_bytes", C.c_uint64),
                ("bvh_nodes8", C.c_uint32), ("_reserved", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("samples", C.c_uint64), ("render_ms", C.c_double),
                ("trace_closest_ms", C.c_double), ("shade_ms", C.c_double), ("trace_shadow_ms", C.c_double), ("other_ms", C.c_double),
                ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("closest_nodes", C.c_uint64), ("closest_tris", C.c_uint64), ("shadow_nodes", C.c_uint64), ("shadow_tris", C.c_uint64),
                ("hits", C.c_uint64), ("fresh_paths", C.c_uint64), ("phase", C.c_uint64 * 12),
                ("tex_fetches", C.c_uint64), ("tex_bytes", C.c_uint64), ("alpha_tex_bytes", C.c_uint64), ("light_samples", C.c_uint64),
                ("sky_samples", C.c_uint64)]


DRAW_CALLBACK = C.CFUNCTYPE(None, C.c_void_p)

# every symbol include/glaze_abi.h declares: name -> (restype, argtypes)
_P = C.c_void_p
PROTOTYPES = {
    "glz_last_error": (C.c_char_p, []),
    "glz_version": (C.c_char_p, []),
    "glz_last_status": (C.c_int, []),
    "glz_parse": (_P, [C.c_char_p]),
    "glz_parsed_free": (None, [_P]),
    "glz_parsed_vertices": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_meshes": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_indices": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_transforms": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_instances": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_cameras": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_materials": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_lights": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_parsed_textures": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_serialize": (C.c_int, [C.c_char_p, _P]),
    "glz_convert_obj": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, _P]),
    "glz_save_image": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_uint32]),
    "glz_parsed_update": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64, _P, C.c_int64, _P]),
    "glz_parsed_meta": (C.c_int, [_P, _P]),
    "glz_converted_file": (C.c_int, [C.c_char_p]),
    "glz_instance_create": (_P, [C.c_int]),
    "glz_instance_destroy": (None, [_P]),
    "glz_instance_device": (C.c_int, [_P]),
    "glz_instance_stream": (_P, [_P]),
    "glz_instance_set_bvh_builder": (C.c_int, [_P, C.c_int]),
    "glz_scene_create": (_P, [_P, _P]),
    "glz_scene_create_from_desc": (_P, [_P, _P]),
    "glz_scene_destroy": (None, [_P]),
    "glz_scene_get_info": (C.c_int, [_P, _P]),
    "glz_scene_camera": (C.c_int, [_P, _P]),
    "glz_renderer_create": (_P, [_P, _P, C.c_uint32, C.c_uint32]),
    "glz_renderer_destroy": (None, [_P]),
    "glz_renderer_set_integrator": (C.c_int, [_P, C.c_int]),
    "glz_renderer_set_exposure": (C.c_int, [_P, C.c_float]),
    "glz_renderer_update_camera": (C.c_int, [_P, _P]),
    "glz_renderer_change_resolution": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "glz_renderer_change_scene": (C.c_int, [_P, _P]),
    "glz_renderer_update_materials_and_lights": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    "glz_renderer_refresh_binded_textures": (C.c_int, [_P, _P, C.c_uint32]),
    "glz_renderer_wait_idle": (C.c_int, [_P]),
    "glz_renderer_steps_per_sample": (C.c_uint32, [_P]),
    "glz_renderer_draw": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "glz_renderer_restart": (C.c_int, [_P]),
    "glz_renderer_step": (C.c_int, [_P, C.c_uint32]),
    "glz_renderer_read_rgba8": (C.c_int, [_P, _P]),
    "glz_renderer_set_seed": (C.c_int, [_P, C.c_uint64]),
    "glz_renderer_set_depth": (C.c_int, [_P, C.c_uint32]),
    "glz_renderer_read_hdr": (C.c_int, [_P, _P]),
    "glz_renderer_read_result": (C.c_int, [_P, _P]),
    "glz_renderer_launch_constants": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "glz_renderer_push_constants": (C.c_int, [_P, _P]),
    "glz_renderer_set_partition": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "glz_renderer_set_chains": (C.c_int, [_P, C.c_uint32]),
    "glz_renderer_export_device": (C.c_int, [_P, C.c_int, _P]),
    "glz_renderer_packed_pixels": (C.c_uint64, [_P, C.c_uint32, C.c_uint32]),
    "glz_renderer_export_packed": (C.c_int, [_P, C.c_int, _P]),
    "glz_renderer_scatter_packed": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P]),
    "glz_renderer_scatter_packed_all": (C.c_int, [_P, C.c_uint32, _P, C.c_uint64, _P]),
    "glz_renderer_tonemap_device": (C.c_int, [_P, _P, _P]),
    "glz_renderer_enable_counters": (C.c_int, [_P, C.c_int]),
    "glz_renderer_get_stats": (C.c_int, [_P, _P]),
    "glz_debug_trace_closest": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_float, _P, _P, _P, _P, _P]),
    "glz_debug_trace_any": (C.c_int, [_P, _P, _P, _P, C.c_uint64, C.c_float, _P]),
    "glz_debug_read_derivatives": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_debug_read_rt_materials": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_debug_read_rt_lights": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_debug_read_sky": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_debug_read_bvh": (C.c_int64, [_P, _P, C.c_int64, _P, C.c_int64]),
    "glz_debug_read_bvh8": (C.c_int64, [_P, _P, C.c_int64]),
    "glz_host_launch_constants": (C.c_int, [C.c_uint64, C.c_uint32, _P, _P]),
    "glz_host_push_constants": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    "glz_host_chain_owner": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    "glz_host_build_sah": (C.c_int, [C.c_uint32, _P, _P, _P, _P]),
    "glz_host_srgb8_thresholds": (C.c_int, [_P]),
    "glz_debug_tonemap": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "glz_debug_read_texture_level": (C.c_int64, [_P, C.c_uint32, C.c_uint32, _P, C.c_int64, _P, _P]),
    "glz_host_mip_level": (C.c_int64, [_P, C.c_uint32, _P, C.c_int64, _P, _P]),
    "glz_renderer_set_texture_lod": (C.c_int, [_P, C.c_int]),
    "glz_debug_rccl_selftest": (C.c_int, [_P, C.c_uint64, _P]),
    "glz_instance_set_as_levels": (C.c_int, [_P, C.c_int]),
    "glz_renderer_set_devices": (C.c_int, [_P, _P, C.c_int]),
    "glz_renderer_set_launch_mode": (C.c_int, [_P, C.c_int]),
    "glz_renderer_set_node_width": (C.c_int, [_P, C.c_int]),
    "glz_renderer_node_width": (C.c_int, [_P]),
    "glz_renderer_launch_mode": (C.c_int, [_P]),
    "glz_renderer_device_count": (C.c_int, [_P]),
    "glz_renderer_device_scene_info": (C.c_int, [_P, C.c_int, _P]),
    "glz_rccl_version": (C.c_int, []),
    "glz_host_tile_owner": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _P]),
}

_LIB = None


def library_path():
    # GLAZE_HIP_LIB selects another build of the same library (kernel tuning experiments); never a different backend
    return os.environ.get("GLAZE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libglaze_hip.so")


class GlazeLibraryMissing(ImportError):
    pass


def lib():
    """Loads libglaze_hip.so (built in-tree by `__graft_entry__.build()` / glaze_amd/csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise GlazeLibraryMissing(
                "%s not found: build it with `make -C glaze_amd/csrc` (there is no CPU fallback for the render path)" % path)
        handle = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


class GlazeError(IOError):
    def __init__(self, status, message):
        super().__init__("%s (status %d)" % (message, status))
        self.status = status


def last_error():
    l = lib()
    msg = l.glz_last_error()
    return GlazeError(l.glz_last_status(), msg.decode("utf8", "replace") if msg else "unknown error")


def check(status):
    if status < 0:
        raise last_error()
    return status
