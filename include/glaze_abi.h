/* glaze_abi.h -- C ABI of libglaze_hip.so, the MI355X-native replacement for glaze's render path.
 *
 * The reference (davidepi/glaze, Rust + Vulkan-RT) has no FFI seam of its own; the seam this
 * library replaces is the public Rust API that `glaze-cli` drives (cli/src/main.rs:76-121,
 * re-exported at lib/src/lib.rs:18-22).  Every entry point below cites the reference item it
 * stands in for.  A Rust `glaze-hip-sys` binding maps
 *   status < 0  ->  io::Error / panic (the reference `expect()`s on every GPU failure),
 *   NULL handle ->  Option::None (RayTraceInstance::new, lib/src/vulkan/instance.rs:376).
 *
 * Conventions: plain pointers and sizes only; all handles opaque; every function returning
 * `int` returns 0 on success or a negative GLZ_E_* code and stores a message retrievable with
 * glz_last_error() (thread-local).  No exceptions cross the boundary.  A renderer is not
 * thread-safe (the reference takes `&mut self` on every mutator, raytracer.rs:180-326).
 */
#ifndef GLAZE_ABI_H
#define GLAZE_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLZ_OK 0
#define GLZ_E_IO (-1)          /* io::ErrorKind::NotFound / UnexpectedEof */
#define GLZ_E_INVALID_INPUT (-2) /* io::ErrorKind::InvalidInput  (bad magic / version) */
#define GLZ_E_INVALID_DATA (-3)  /* io::ErrorKind::InvalidData   (hash mismatch, corrupt xz/png) */
#define GLZ_E_ARG (-4)         /* bad argument to the C ABI itself */
#define GLZ_E_DEVICE (-5)      /* HIP failure (the reference panics here) */
#define GLZ_E_UNSUPPORTED (-6)

/* ------------------------------------------------------------------------------------------
 * Scene model PODs (host side).  Field meaning follows the reference types; layouts are this
 * ABI's own (the .glaze byte format is decoded by the library, lib/src/parser/v1.rs:631-1080).
 * ---------------------------------------------------------------------------------------- */

/* geometry/vertex.rs:6-15 -- 32 bytes, identical to the reference's #[repr(C)] Vertex. */
typedef struct glz_vertex {
  float vv[3]; /* position */
  float vn[3]; /* normal   */
  float vt[2]; /* texcoord */
} glz_vertex;

/* geometry/mesh.rs:5-16 -- indices of all meshes are concatenated in one u32 array; a mesh owns
 * [index_offset, index_offset+index_count) of it (what load_indices_to_gpu builds,
 * vulkan/scene.rs:834-850). */
typedef struct glz_mesh {
  uint16_t id;
  uint16_t material;
  uint32_t index_offset;
  uint32_t index_count;
} glz_mesh;

/* geometry/mesh.rs:29-32 -- column-major 4x4, as cgmath::Matrix4<f32>. */
typedef struct glz_transform {
  float m[16];
} glz_transform;

/* geometry/mesh.rs:23-27 */
typedef struct glz_mesh_instance {
  uint16_t mesh_id;
  uint16_t transform_id;
} glz_mesh_instance;

/* geometry/camera.rs:7-83 */
#define GLZ_CAMERA_PERSPECTIVE 0
#define GLZ_CAMERA_ORTHOGRAPHIC 1
typedef struct glz_camera {
  uint32_t type;
  float position[3];
  float target[3];
  float up[3];
  float fovx_or_scale; /* fovx (radians) for perspective, scale for orthographic */
  float near_plane;
  float far_plane;
} glz_camera;

/* materials/material.rs:17-42, ids :62-73 */
#define GLZ_MAT_FLAT 0
#define GLZ_MAT_LAMBERT 1
#define GLZ_MAT_MIRROR 2
#define GLZ_MAT_GLASS 3
#define GLZ_MAT_METAL 4
#define GLZ_MAT_FROSTED 5
#define GLZ_MAT_UBER 6
#define GLZ_NAME_MAX 256
/* materials/material.rs:290-325 */
typedef struct glz_material {
  uint8_t mtype;  /* GLZ_MAT_* */
  uint8_t metal;  /* index into materials/metal.rs:6-37 (0 = SILVER) */
  uint8_t diffuse_mul[3];
  uint8_t emissive_col[3];
  uint8_t has_emissive; /* Option<[u8;3]>::is_some() */
  uint8_t _pad[3];
  float ior;
  float roughness_mul;
  float metalness_mul;
  float anisotropy;
  uint16_t diffuse;   /* texture ids; 0 = none (white 1x1 default texture) */
  uint16_t roughness;
  uint16_t metalness;
  uint16_t normal;
  uint16_t opacity;
  uint16_t _pad2;
  char name[GLZ_NAME_MAX];
} glz_material;

/* geometry/light.rs:12-22, :146-170 */
#define GLZ_LIGHT_OMNI 0
#define GLZ_LIGHT_SUN 1
#define GLZ_LIGHT_AREA 2
#define GLZ_LIGHT_SKY 3
typedef struct glz_light {
  uint32_t ltype;
  float position[3];
  float direction[3];
  uint32_t resource_id; /* material id (AREA) or texture id (SKY) */
  float intensity;
  float yaw_deg, pitch_deg, roll_deg;
  float color[16]; /* Spectrum: 16 bins, 400..700 nm (geometry/spectrum.rs:11-24) */
  char name[GLZ_NAME_MAX];
} glz_light;

/* materials/texture.rs:94-130; only mip level 0 is used by the ray-tracing stages. */
#define GLZ_TEX_GRAY 1      /* R8_UNORM  */
#define GLZ_TEX_RGBA_SRGB 2 /* R8G8B8A8_SRGB */
#define GLZ_TEX_RGBA_NORM 3 /* R8G8B8A8_UNORM */
typedef struct glz_texture {
  uint32_t format; /* GLZ_TEX_* */
  uint32_t width, height;
  uint32_t mip_levels;   /* as stored in the file; informational */
  const uint8_t* pixels; /* level 0, tightly packed rows, 1 or 4 bytes per pixel */
  char name[GLZ_NAME_MAX];
} glz_texture;

/* parser/mod.rs:274-288 */
typedef struct glz_meta {
  float scene_centre[3];
  float scene_radius;
  float exposure;
} glz_meta;

/* In-memory scene (no reference counterpart as a type: it is what `Box<dyn ParsedScene>` hands
 * to RayTraceScene::new, vulkan/scene.rs:1414-1434).  NULL / 0 members take the reference's
 * defaults: one identity transform, one default material, one default 1x1 white texture,
 * default camera, Meta::default(). */
typedef struct glz_scene_desc {
  const glz_vertex* vertices;        uint64_t n_vertices;
  const uint32_t* indices;           uint64_t n_indices;
  const glz_mesh* meshes;            uint32_t n_meshes;
  const glz_transform* transforms;   uint32_t n_transforms;
  const glz_mesh_instance* instances;uint32_t n_instances;
  const glz_material* materials;     uint32_t n_materials;
  const glz_light* lights;           uint32_t n_lights;
  const glz_texture* textures;       uint32_t n_textures;
  const glz_camera* camera;          /* the LAST camera of the file (scene.rs:1487-1491) */
  const glz_meta* meta;
} glz_scene_desc;

/* ------------------------------------------------------------------------------------------
 * Errors
 * ---------------------------------------------------------------------------------------- */
const char* glz_last_error(void);
const char* glz_version(void);

/* ------------------------------------------------------------------------------------------
 * parse()  --  lib/src/parser/mod.rs:93-116 + ContentV1 read side (parser/v1.rs:135-175,
 * :298-362, :476-609).  Header and offset table are checked at open; chunks are decoded lazily
 * by the getters (hash mismatch -> GLZ_E_INVALID_DATA, missing chunk -> count 0).
 * Getter protocol: pass out == NULL to obtain the element count; otherwise up to `cap` elements
 * are written and the total count is returned (negative = error).
 * ---------------------------------------------------------------------------------------- */
typedef struct glz_parsed glz_parsed;
glz_parsed* glz_parse(const char* path);            /* NULL on error, see glz_last_error() */
int glz_last_status(void);                           /* status code of the last failing call on this thread */
void glz_parsed_free(glz_parsed*);
int64_t glz_parsed_vertices(glz_parsed*, glz_vertex* out, int64_t cap);           /* ParsedScene::vertices   */
int64_t glz_parsed_meshes(glz_parsed*, glz_mesh* out, int64_t cap);               /* ParsedScene::meshes     */
int64_t glz_parsed_indices(glz_parsed*, uint32_t* out, int64_t cap);              /*   (their index arrays)  */
int64_t glz_parsed_transforms(glz_parsed*, glz_transform* out, int64_t cap);      /* ParsedScene::transforms */
int64_t glz_parsed_instances(glz_parsed*, glz_mesh_instance* out, int64_t cap);   /* ParsedScene::instances  */
int64_t glz_parsed_cameras(glz_parsed*, glz_camera* out, int64_t cap);            /* ParsedScene::cameras    */
int64_t glz_parsed_materials(glz_parsed*, glz_material* out, int64_t cap);        /* ParsedScene::materials  */
int64_t glz_parsed_lights(glz_parsed*, glz_light* out, int64_t cap);              /* ParsedScene::lights     */
/* textures: `pixels` of each returned glz_texture points into memory owned by the glz_parsed. */
int64_t glz_parsed_textures(glz_parsed*, glz_texture* out, int64_t cap);          /* ParsedScene::textures   */
int glz_parsed_meta(glz_parsed*, glz_meta* out);  /* ParsedScene::meta; returns 1 if the chunk is absent (out = default) */
int glz_converted_file(const char* path);         /* parser/mod.rs:259-271: 1 if the magic matches */

/* ------------------------------------------------------------------------------------------
 * Write side.  Serializer::new(file, ParserVersion::V1).with_*(..).serialize()
 * (lib/src/parser/mod.rs:130-233; ContentV1::serialize / write_chunks, parser/v1.rs:230-295; record encoders :613-1061).
 * Every array may be empty (no chunk is written for it); `meta` NULL = Serializer without with_metadata().
 * Chunks are xz streams (LZMA2, CRC64) with an XXH64 prefix, textures are PNGs: the output is read back by glz_parse and
 * by the reference's parser alike.  glz_texture.mip_levels > 1 asks for that many levels in all, each the one before halved with the
 * Catmull-Rom filter as Texture::gen_mipmaps does (texture.rs:256-277).
 * ---------------------------------------------------------------------------------------- */
typedef struct glz_serialize_desc {
  const glz_vertex* vertices;         uint64_t n_vertices;
  const uint32_t* indices;            uint64_t n_indices;    /* index arrays of all meshes (glz_mesh.index_offset/count) */
  const glz_mesh* meshes;             uint64_t n_meshes;
  const glz_transform* transforms;    uint64_t n_transforms;
  const glz_mesh_instance* instances; uint64_t n_instances;
  const glz_camera* cameras;          uint64_t n_cameras;
  const glz_texture* textures;        uint64_t n_textures;
  const glz_material* materials;      uint64_t n_materials;
  const glz_light* lights;            uint64_t n_lights;
  const glz_meta* meta;
} glz_serialize_desc;
int glz_serialize(const char* path, const glz_serialize_desc* desc);
/* ParsedScene::update (parser/v1.rs:364-422): rewrites the parsed file in place with the given chunks replaced and
 * re-reads it.  A count of -1 (or meta == NULL) keeps that chunk exactly as stored; geometry is always kept.  Pointers
 * obtained earlier from this handle's getters are invalidated. */
int glz_parsed_update(glz_parsed*, const glz_camera* cameras, int64_t n_cameras, const glz_material* materials, int64_t n_materials,
                      const glz_light* lights, int64_t n_lights, const glz_texture* textures, int64_t n_textures, const glz_meta* meta);

/* image.save(path) of the CLI (cli/src/main.rs:121): RGBA8 rows top-down -> PNG (RGBA) or baseline JPEG (alpha dropped,
 * quality 75) chosen by the extension (.png / .jpg / .jpeg). */
int glz_save_image(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
/* glaze-converter for Wavefront OBJ input (converter/src/main.rs:116-637 without assimp): OBJ + MTL + PNG / baseline-JPEG
 * textures -> .glaze V1 with the reference converter's layout (default material/texture first, "DefaultMaterial" second,
 * de-duplicated flipped-v vertices, one identity transform, AREA lights for emissive materials, default camera, Meta
 * from the bounds).  gen_mipmaps != 0 stores the full mip chain (--gen-mipmaps).  counts (may be NULL) receives
 * {vertices, triangles, meshes, materials, textures, lights}. */
int glz_convert_obj(const char* input_obj, const char* output_glaze, int gen_mipmaps, uint64_t counts[6]);

/* ------------------------------------------------------------------------------------------
 * RayTraceInstance::new() -> Option<Self>   (lib/src/vulkan/instance.rs:376-427)
 * hip_device = -1 picks the first gfx950 device (LOCAL_RANK-agnostic; pass the ordinal for
 * one-process-per-GPU jobs).  NULL <=> None: no usable device or the HIP code object is missing.
 * ---------------------------------------------------------------------------------------- */
typedef struct glz_instance glz_instance;
glz_instance* glz_instance_create(int hip_device);
void glz_instance_destroy(glz_instance*);
int glz_instance_device(const glz_instance*);
/* stream all of this instance's kernels run on (a hipStream_t), for event timing by the caller */
void* glz_instance_stream(const glz_instance*);
/* [extension] acceleration-structure builder for scenes created afterwards (the reference leaves the choice to the
 * Vulkan driver, acceleration.rs:253-257 asks for PREFER_FAST_TRACE).  Hits do not depend on it.
 *   GLZ_BVH_SAH      (default, = GLZ_BVH_AUTO) top-down binned SAH on the GPU, level by level: 7 ms for 262 k
 *                    triangles, 14 ms for 1.8 M, 34 ms for 7 M, 86 ms for 21 M; SAH cost -11 % and 5-13 % more samples
 *                    per second than the LBVH
 *   GLZ_BVH_LBVH     Karras 2012 on the GPU: fastest build (3 ms for 262 k triangles, 33 ms for 21 M)
 *   GLZ_BVH_PLOC     parallel locally-ordered clustering on the GPU (Meister & Bittner 2018): the LBVH's quality here
 *   GLZ_BVH_SAH_HOST the SAH builder's reference implementation on the host cores: the same tree node for node, 3-24 x
 *                    slower (tests compare the two) */
#define GLZ_BVH_LBVH 0
#define GLZ_BVH_PLOC 1
#define GLZ_BVH_SAH 2
#define GLZ_BVH_AUTO 3
#define GLZ_BVH_SAH_HOST 4
int glz_instance_set_bvh_builder(glz_instance*, int builder);
/* [extension] shape of the acceleration structure of scenes created afterwards.  The reference keeps one BLAS per mesh and a TLAS
 * over the instances (acceleration.rs:319-345).  GLZ_AS_AUTO (default): two levels when the instances hold more than four
 * times the triangles of the meshes they share, else ONE hierarchy over world-space triangles (fastest to trace, memory
 * grows with instances x triangles); GLZ_AS_FLAT / GLZ_AS_TWO_LEVEL force either.  Hits are bit-identical both ways: inside an
 * instance only the box tests run in object space, the triangle test stays in world space. */
#define GLZ_AS_AUTO 0
#define GLZ_AS_FLAT 1
#define GLZ_AS_TWO_LEVEL 2
int glz_instance_set_as_levels(glz_instance*, int mode);

/* ------------------------------------------------------------------------------------------
 * RayTraceScene::new(instance, parsed)   (lib/src/vulkan/scene.rs:1414-1556)
 * Uploads geometry, runs the derivative kernel (scene.rs:2113-2188), builds the LBVH that
 * replaces SceneASBuilder (vulkan/acceleration.rs:89-494), builds RTInstance/RTMaterial/RTLight
 * arrays (scene.rs:1784-1927) and the sky sampling tables (scene.rs:2191-2313).
 * glz_scene_create consumes `parsed` (it is freed, as the Box is moved in the reference).
 * ---------------------------------------------------------------------------------------- */
typedef struct glz_scene glz_scene;
glz_scene* glz_scene_create(glz_instance*, glz_parsed* parsed);
glz_scene* glz_scene_create_from_desc(glz_instance*, const glz_scene_desc*);
void glz_scene_destroy(glz_scene*);

typedef struct glz_scene_info {
  uint64_t n_vertices, n_triangles;          /* object-space triangles over all meshes */
  uint64_t n_world_triangles;                /* after instancing (BVH primitives)       */
  uint32_t n_instances, n_materials, n_lights /* lights_no, scene.rs:1549 */, n_rt_lights, n_textures;
  uint32_t bvh_nodes, bvh_depth;
  float bvh_sah_cost;
  float build_ms;                            /* LBVH build time on the device */
  float bounds_min[3], bounds_max[3];
  float bvh_grid_lo[3], bvh_grid_cell[3];    /* quantisation grid of the BVH node boxes: world = lo + q * cell */
  uint32_t as_levels;                        /* 1 = one hierarchy over all instanced triangles, 2 = TLAS over per-mesh BLAS */
  uint32_t n_as_triangles;                   /* triangle records the structure holds (= n_world_triangles when flattened) */
  uint64_t as_bytes;                         /* device bytes of nodes + triangle / leaf records + shading records + instance records */
  uint32_t bvh_nodes8, _reserved;            /* 8-wide nodes of the same hierarchy (the tracer of small tile shares; flattened builds), 0 = none */
} glz_scene_info;
int glz_scene_get_info(const glz_scene*, glz_scene_info* out);
int glz_scene_camera(const glz_scene*, glz_camera* out);

/* ------------------------------------------------------------------------------------------
 * RayTraceRenderer   (lib/src/vulkan/raytracer.rs:109-687)
 * ---------------------------------------------------------------------------------------- */
typedef struct glz_renderer glz_renderer;
#define GLZ_DIRECT 0      /* Integrator::DIRECT      raytracer.rs:36-52 */
#define GLZ_PATH_TRACE 1  /* Integrator::PATH_TRACE */

/* RayTraceRenderer::new(instance, Some(scene), w, h)  raytracer.rs:164-177; the renderer owns
 * the scene from here on (raytracer.rs:109-111): do not destroy it yourself. */
glz_renderer* glz_renderer_create(glz_instance*, glz_scene* scene, uint32_t width, uint32_t height);
void glz_renderer_destroy(glz_renderer*);
int glz_renderer_set_integrator(glz_renderer*, int integrator);          /* raytracer.rs:196-231 */
int glz_renderer_set_exposure(glz_renderer*, float exposure);            /* raytracer.rs:180-194 */
int glz_renderer_update_camera(glz_renderer*, const glz_camera*);        /* raytracer.rs:300-309 */
int glz_renderer_change_resolution(glz_renderer*, uint32_t w, uint32_t h);/* raytracer.rs:250-298 */
int glz_renderer_change_scene(glz_renderer*, glz_scene* scene);          /* raytracer.rs:233-248 */
/* raytracer.rs:311-326 (&[Material], &[Light], &[Texture]): rebuilds RTMaterial / RTLight / sky tables; restarts
 * accumulation.  The material count must not change.  `textures` NULL keeps the scene's texture array; otherwise it
 * replaces it (level 0 is uploaded; the reference needs the raw textures for the sky distributions, scene.rs:1598-1615). */
int glz_renderer_update_materials_and_lights(glz_renderer*, const glz_material* mats, uint32_t n_mats,
                                             const glz_light* lights, uint32_t n_lights,
                                             const glz_texture* textures, uint32_t n_textures);
/* raytracer.rs:328-356: the texture array changed under the same materials and lights (the reference re-binds the
 * textures it shares with the realtime viewer); accumulation is NOT restarted, like the reference. */
int glz_renderer_refresh_binded_textures(glz_renderer*, const glz_texture* textures, uint32_t n_textures);
int glz_renderer_wait_idle(glz_renderer*);                                /* raytracer.rs:328-340 */
/* Integrator::steps_per_sample()  raytracer.rs:78-85 (PATH_TRACE -> the configured depth) */
uint32_t glz_renderer_steps_per_sample(const glz_renderer*);

/* draw(spp, callback) -> RgbaImage   raytracer.rs:615-687.  Resets accumulation, runs
 * spp * steps_per_sample launches, calls `cb(user)` once per spp on the caller's thread
 * (raytracer.rs:651-653), writes W*H RGBA8 sRGB pixels (alpha 255) to rgba8_out (may be NULL). */
int glz_renderer_draw(glz_renderer*, size_t spp, void (*cb)(void*), void* user, uint8_t* rgba8_out);

/* draw_frame(): ONE launch = one path segment per pixel  (raytracer.rs:369-613), the unit the
 * interactive viewer drives.  `n` launches are enqueued; accumulation continues unless
 * glz_renderer_restart() was called (request_new_frame). */
int glz_renderer_restart(glz_renderer*);
int glz_renderer_step(glz_renderer*, uint32_t n_launches);
int glz_renderer_read_rgba8(glz_renderer*, uint8_t* rgba8_out); /* blit out32->out8 + export, memory.rs:269-483 */

/* ---- build-defined extensions (no reference counterpart) ---------------- */
int glz_renderer_set_seed(glz_renderer*, uint64_t seed);    /* replaces Xoshiro128PlusPlus::from_entropy, raytracer.rs:779 */
int glz_renderer_set_depth(glz_renderer*, uint32_t pt_steps);/* replaces const PT_STEPS = 6, raytrace_structures.rs:87 */
/* Texture level of detail.  The reference's samplers have LINEAR mip filtering and mip chains (generated by
 * LINEAR blits at upload when the file brings none, vulkan/scene.rs:1012-1263), but its ray-tracing stages call texture() with
 * the implicit level, which outside fragment shaders is level 0: GLZ_LOD_BASE (the default) reproduces that bit for bit.
 * GLZ_LOD_RAY_CONES picks the level per hit from the ray's footprint (cone width carried along the path, triangle texture /
 * world area ratio, angle of incidence) and blends the two nearest levels; the chain is built on the first launch that needs it.
 * GLZ_LOD_RAY_CONES_ANISO adds what the sampler's anisotropy (scene.rs:716-749: enabled at the device's maximum) does in the raster
 * viewer: the footprint is the cone's width across and width / |cos| along the ray's projection onto the surface; up to 16
 * trilinear probes along that axis, each at the level of its share of it, averaged.
 * Restarts accumulation. */
#define GLZ_LOD_BASE 0
#define GLZ_LOD_RAY_CONES 1
#define GLZ_LOD_RAY_CONES_ANISO 2
int glz_renderer_set_texture_lod(glz_renderer*, int mode);
/* cumulative image (xyz = sum of rgb radiance, w = launch count; path_trace.rgen:119-133),
 * W*H*4 floats, row-major, to host memory. */
int glz_renderer_read_hdr(glz_renderer*, float* rgba32f_out);
/* result image (`out32`: cumulative.xyz*exposure/w at the pixel's last update, path_trace.rgen:131-132) */
int glz_renderer_read_result(glz_renderer*, float* rgba32f_out);
/* per-launch host constants the draw loop will use for launch `i` after a restart:
 * seed (u32) and WorkScheduler pixel offset (raytracer.rs:486-489, :1168-1206). */
int glz_renderer_launch_constants(glz_renderer*, uint32_t launch, uint32_t* seed, float offset[2]);
/* camera push constants: camera2world then screen2camera, column-major (raytracer.rs:1098-1120) */
int glz_renderer_push_constants(glz_renderer*, float out32[32]);

/* Multi-GPU (one process per GPU): this renderer owns the 64x64-pixel tiles t with
 * t % world == rank; other pixels stay zero.  The exchange of the HDR accumulator itself is done
 * by the caller (RCCL through torch.distributed or rccl directly) on the device buffers below. */
int glz_renderer_set_partition(glz_renderer*, uint32_t rank, uint32_t world);
/* Scatters the owned tiles of the cumulative image (which = 0) or of the result image `out32`
 * (which = 1) into a caller-provided DEVICE buffer of W*H*4 floats, zero elsewhere, on the instance
 * stream (synchronised before returning): the operand of ncclReduce(sum).  Tiles are disjoint, so
 * the sum over ranks is bit-identical to a single-GPU render. */
int glz_renderer_export_device(glz_renderer*, int which, void* dev_rgba32f);
/* The same exchange with 1 / world of the bytes: a rank's tiles ONLY, tile-major (64x64 pixels of 4 floats per tile, local
 * tile j = global tile rank + j * world), glz_renderer_packed_pixels(r, rank, world) pixels -- the operand of an ncclSend /
 * gather to rank 0, which puts every received buffer in its place of a full frame with glz_renderer_scatter_packed (nothing
 * but those tiles is written; rank 0's own tiles come from glz_renderer_export_device).  Device pointers; synchronised
 * before returning. */
uint64_t glz_renderer_packed_pixels(glz_renderer*, uint32_t rank, uint32_t world);
int glz_renderer_export_packed(glz_renderer*, int which, void* dev_packed_rgba32f);
int glz_renderer_scatter_packed(glz_renderer*, uint32_t rank, uint32_t world, const void* dev_packed_rgba32f, void* dev_frame_rgba32f);
/* The receiving side of a gather in ONE call: `dev_packed_rgba32f` holds the packed tiles of ranks 0 .. world - 1 one after the other,
 * `stride_pixels` pixels apart (what a gather into one buffer leaves; >= glz_renderer_packed_pixels(r, 0, world)); every rank's tiles go
 * to their place in the frame -- world small kernels on the instance stream, ONE synchronisation (rank 0's own tiles included: the
 * frame needs no glz_renderer_export_device first, the tiles of all ranks cover it). */
int glz_renderer_scatter_packed_all(glz_renderer*, uint32_t world, const void* dev_packed_rgba32f, uint64_t stride_pixels, void* dev_frame_rgba32f);
/* Concurrent chains: the tiles of this process advance as `n` independent launch sequences on `n` HIP streams (0 = automatic:
 * one chain while the rank owns a million pixels, two down to 400 k, three below).  Pixels never interact, so the image does not depend on n; with a small
 * tile share per GPU the chains fill the machine while the longest rays of a launch finish. */
int glz_renderer_set_chains(glz_renderer*, uint32_t n);
/* How a launch (one path segment per pixel, draw_frame) reaches the device.  GLZ_LAUNCH_TWO_KERNELS: k_trace + k_shade per launch over
 * all pixels -- throughput, the full frame.  GLZ_LAUNCH_PATH: every wave carries its 64 pixels through the launches of a batch
 * (closest hits -> shadow rays of the launch before -> shading) inside one persistent kernel, no grid-wide boundary between launches
 * -- latency, a small tile share per GPU (a 1080p frame over 8 GPUs); flattened scenes, and only while the work counters are off.
 * GLZ_LAUNCH_AUTO (default): by the pixels this device owns.  The image does not depend on the mode.  glz_renderer_launch_mode
 * returns the mode in force (never AUTO). */
#define GLZ_LAUNCH_AUTO 0
#define GLZ_LAUNCH_TWO_KERNELS 1
#define GLZ_LAUNCH_PATH 2
int glz_renderer_set_launch_mode(glz_renderer*, int mode);
int glz_renderer_launch_mode(glz_renderer*);
/* Which nodes the two-kernel mode's traversal walks (build-defined; the reference's structure is the driver's, acceleration.rs:319-345):
 * 4 = the 64-byte 4-wide nodes (k_trace), 8 = the 128-byte 8-wide nodes of the same hierarchy (k_trace8; flattened scenes, and only
 * while the work counters are off -- the counted walk is the 4-wide one), 0 (default) = automatic, which is 4: the 8-wide walk visits
 * 31 % fewer nodes and measured slower at every tile share, it is kept as a tested option.  The image does not
 * depend on it.  glz_renderer_node_width returns the width in force. */
int glz_renderer_set_node_width(glz_renderer*, int width);
int glz_renderer_node_width(glz_renderer*);
/* Multi-GPU inside ONE process (what a Rust glaze-cli bound to this library uses for `--devices`; the reference drives exactly one
 * VkPhysicalDevice, lib/src/vulkan/device.rs:252-321): hip_devices[0] must be the renderer's own device (glz_instance_device);
 * every further device gets a stream, a replica of the scene (upload + BVH build on that device), a renderer for the 64x64 tiles
 * t % n == i and a host thread that enqueues its launches.  All setters, step / draw and scene updates apply to every device;
 * every read-back (read_hdr / read_result / read_rgba8 / draw's image / export_device) first brings the other devices' tiles
 * onto hip_devices[0] over RCCL / xGMI (one communicator per device from ncclCommInitAll, all calls of one exchange in one
 * ncclGroup): by default every device ncclSends its packed tiles (1 / n of the frame) and device 0 ncclRecvs and scatters
 * them; with GLAZE_MULTI_EXCHANGE=reduce in the environment at this call, one ncclReduce(sum, float) of the zero-padded
 * RGBA32F frame per device; with GLAZE_MULTI_EXCHANGE=peer no RCCL at all: one hipMemcpyPeerAsync of the packed tiles per device,
 * ordered into device 0's stream by events (never chosen by the library itself).  The tiles are disjoint, so the image is
 * bit-identical to a one-device render.  n = 1 returns to one
 * device; after a failure the renderer is a one-device renderer again.  Not combinable with glz_renderer_set_partition (one process per GPU).  RCCL is loaded on first use
 * (librccl.so.1); GLZ_E_DEVICE when it is missing.  With GLAZE_MULTI_LOOPBACK=1 in the environment the list may name ONE
 * device n times (tests on a one-GPU machine: same sharding, threads and replicas, the tiles meet without RCCL). */
int glz_renderer_set_devices(glz_renderer*, const int* hip_devices, int n);
int glz_renderer_device_count(glz_renderer*);   /* devices this renderer spans (1 unless set_devices succeeded with more) */
int glz_renderer_device_scene_info(glz_renderer*, int i, glz_scene_info* out);   /* the scene replica device i renders (0 = the renderer's own) */
int glz_rccl_version(void);                     /* ncclGetVersion() of the RCCL this library loads (> 0), or a negative status */
/* Tonemaps a full-frame DEVICE result image (e.g. the reduced one on rank 0) to RGBA8 sRGB host memory. */
int glz_renderer_tonemap_device(glz_renderer*, const void* dev_result_rgba32f, uint8_t* rgba8_out);

/* ---- measurement ------------------------------------------------------------------------ */
typedef struct glz_render_stats {
  uint64_t launches;        /* launches since the last restart */
  uint64_t samples;         /* owned pixels x launches */
  double   render_ms;       /* device time of those launches (hipEvent, instance stream) */
  double   trace_closest_ms, shade_ms, trace_shadow_ms, other_ms; /* per-kernel-class device time: k_trace, k_shade, stand-alone shadow passes, k_path (GLZ_LAUNCH_PATH) */
  uint64_t closest_rays, shadow_rays;
  /* traversal work counters (only filled when counting is enabled, slows rendering down) */
  uint64_t closest_nodes, closest_tris, shadow_nodes, shadow_tris, hits;
  uint64_t fresh_paths;     /* closest rays that were new camera rays (bounce 0) */
  /* tracer phase occupancy, counting build only: {node rounds, node lanes, leaf rounds, leaf lanes, refills, refilled lanes} x {closest, shadow} */
  uint64_t phase[12];
  /* shading-side work, counting build only: texture fetches that read memory (bilinear: four texels each; 1 x 1 textures live in
   * their descriptor and are not counted) and their texel bytes (16 per RGBA fetch, 4 per gray one) -- of which alpha_tex_bytes in
   * the traversal kernel's alpha tests --, light samples of omni / sun / area lights (one 112-byte RTLight each) and of the sky
   * (a binary search of the marginal table, two values of the conditional tables, one texel fetch, counted above) */
  uint64_t tex_fetches, tex_bytes, alpha_tex_bytes, light_samples, sky_samples;
} glz_render_stats;
/* bit 0: traversal work counters (slower kernels); bit 1: per-kernel hipEvent timing (on by default) */
int glz_renderer_enable_counters(glz_renderer*, int flags);
int glz_renderer_get_stats(glz_renderer*, glz_render_stats* out);

/* ---- debug / parity hooks (used by tests; run on the device, no CPU fallback) -------------- */
/* closest hit of n rays: t (inf = miss), world triangle id (instance-major), barycentric u,v */
int glz_debug_trace_closest(glz_scene*, const float* origins3, const float* dirs3, uint64_t n, float tmin,
                            float* t_out, uint32_t* tri_out, uint32_t* inst_out, float* u_out, float* v_out);
int glz_debug_trace_any(glz_scene*, const float* origins3, const float* dirs3, const float* tmax, uint64_t n,
                        float tmin, uint8_t* hit_out);
/* Derivatives buffer (normal, dpdu, dpdv as 3 x vec4 per object-space triangle), generate_derivatives.comp */
int64_t glz_debug_read_derivatives(glz_scene*, float* out12, int64_t cap_triangles);
/* Device-side sky tables / RT arrays as uploaded (for parity with the oracle's host prep) */
int64_t glz_debug_read_rt_materials(glz_scene*, void* out, int64_t cap_bytes); /* 208-byte RTMaterial records */
int64_t glz_debug_read_rt_lights(glz_scene*, void* out, int64_t cap_bytes);    /* 112-byte RTLight records */
int64_t glz_debug_read_sky(glz_scene*, float* out, int64_t cap_floats);        /* RTSky(36 f32) | header(4) | marginal arrays */
/* BVH as traversed by the kernels: 64-byte quantised 4-wide nodes (returns the node count) and 48-byte leaf
 * triangles in leaf order (n_world_triangles of them; a leaf is one triangle or two adjacent ones, bit 30 of the
 * first one's last word says which, and leaf links name the first); see DESIGN.md for the layouts. */
int64_t glz_debug_read_bvh(glz_scene*, void* nodes_out, int64_t cap_nodes, void* tris_out, int64_t cap_tris);
/* The same hierarchy as 128-byte 8-wide nodes (what the tracer of a small tile share walks; glz_scene_info::bvh_nodes8 of them, flattened
 * builds only): child k's box words at [3k .. 3k+2], its link at [24 + k], leaf links name the first triangle slot as above. */
int64_t glz_debug_read_bvh8(glz_scene*, void* nodes_out, int64_t cap_nodes);

/* one level of a scene texture's mip chain as the device holds it (level 0 = the texture; builds the chain if needed): returns the
 * byte count (0 past the last level), writes up to cap bytes and the level's dimensions */
int64_t glz_debug_read_texture_level(glz_scene*, uint32_t texture, uint32_t level, uint8_t* out, int64_t cap, uint32_t* width, uint32_t* height);
/* k_tonemap (the out32 -> RGBA8 sRGB blit, raytracer.rs:576-584) on n host pixels of RGBA32F: upload, kernel, read back */
int glz_debug_tonemap(glz_instance*, const float* rgba32f, uint64_t n_pixels, uint8_t* rgba8_out);

/* First contact with RCCL on this machine: a one-rank communicator on the instance's device (ncclCommInitAll), one
 * ncclReduce(sum, float) of n_floats values on the instance's stream, result compared bit for bit with the input, communicator
 * destroyed.  version_out (may be NULL) receives ncclGetVersion(). */
int glz_debug_rccl_selftest(glz_instance*, uint64_t n_floats, int* version_out);

/* ---- host logic, callable without a device (used by the CPU test-suite and by bindings) ------ */
/* seed + pixel offset of launch `launch` after a restart, for renderer seed `seed`
 * (rng.gen::<u32>() + WorkScheduler::next(), raytracer.rs:486-489, :1168-1206) */
int glz_host_launch_constants(uint64_t seed, uint32_t launch, uint32_t* seed_out, float offset[2]);
/* build_push_constants (raytracer.rs:1098-1120): camera2world then screen2camera, column-major */
int glz_host_push_constants(const glz_camera* camera, uint32_t width, uint32_t height, float out32[32]);
/* rank owning each pixel under glz_renderer_set_partition(., world): 64x64 tiles, tile t -> t % world */
int glz_host_tile_owner(uint32_t width, uint32_t height, uint32_t world, uint16_t* owner_out);
/* launch chains a renderer of this size and partition runs with (chains = 0: the automatic choice of glz_renderer_set_chains), and the
 * chain rendering each pixel of rank `rank` (0xFFFF for pixels of other ranks): chain s of S owns tiles t with t % (world*S) == rank + s*world */
int glz_host_chain_owner(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t chains, uint16_t* owner_out);
/* the host side of GLZ_BVH_SAH on its own (no GPU): binary hierarchy over n >= 2 leaf boxes (box_lo / box_hi: 4 floats per leaf,
 * xyz used).  children_out[2 * i], [2 * i + 1] for inner node i < n - 1: link >= 0 inner node, < 0 ~leaf; parent_out[i] for
 * inner node i (root 0: -1), parent_out[n - 1 + l] for leaf l. */
/* the 8-bit sRGB quantiser of read_rgba8 / draw (the reference's R8G8B8A8_SRGB blit, raytracer.rs:576-584): c encodes to
 * #{k in 1..255 : c >= thresholds_out[k]}; thresholds_out[0] = 0 */
int glz_host_srgb8_thresholds(float thresholds_out[256]);
/* one level of the mip chain the library generates for a texture that brings only level 0 (mipchain.h: LINEAR-blit rule); returns the
 * byte count of that level (0 past the last), writes up to cap bytes */
int64_t glz_host_mip_level(const glz_texture* texture, uint32_t level, uint8_t* out, int64_t cap, uint32_t* width, uint32_t* height);
int glz_host_build_sah(uint32_t n, const float* box_lo, const float* box_hi, int32_t* children_out, int32_t* parent_out);

#ifdef __cplusplus
}
#endif
#endif /* GLAZE_ABI_H */
