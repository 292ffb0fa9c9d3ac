// extern "C" surface of libglaze_hip.so (include/glaze_abi.h).  Nothing here touches the oracle or
// any CPU rendering path: every render call goes to the HIP kernels or fails with GLZ_E_DEVICE.
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "glaze_abi.h"
#include "parser.h"
#include "serializer.h"
#include "converter.h"
#include "codec/jpeg.h"
#include "codec/png_enc.h"
#include <strings.h>
#include "mipchain.h"
#include "rccl_dl.h"
#include "renderer.h"
#include "scene.h"

using namespace glz;

namespace {
thread_local std::string g_error;
thread_local int g_status = GLZ_OK;

int fail(const Error& e) {
  g_error = e.msg;
  g_status = e.code == GLZ_OK ? GLZ_E_ARG : e.code;
  return g_status;
}
int fail(int code, const char* msg) {
  g_error = msg;
  g_status = code;
  return code;
}
template <class T>
int64_t copy_out(const std::vector<T>& v, T* out, int64_t cap) {
  if (out && cap > 0) memcpy(out, v.data(), sizeof(T) * (size_t)std::min<int64_t>(cap, (int64_t)v.size()));
  return (int64_t)v.size();
}

template <class T>
bool to_device(DeviceBuffer<T>& b, const T* host, size_t n, hipStream_t st, Error& e) { return hip_ok(b.upload(host, n, st), "debug upload", e); }

// Guards every entry point: C++ exceptions (bad_alloc...) must not cross the C boundary.
#define GLZ_GUARD_BEGIN try {
#define GLZ_GUARD_END(ret)                                        \
  }                                                               \
  catch (const std::bad_alloc&) { fail(GLZ_E_IO, "out of host memory"); return ret; } \
  catch (const std::exception& ex) { fail(GLZ_E_ARG, ex.what()); return ret; }
}  // namespace

struct glz_parsed {
  std::unique_ptr<Parsed> p;
  std::vector<glz_texture> tex_view;
};
struct glz_instance {
  std::unique_ptr<Instance> i;
};
struct glz_scene {
  // Shared with the renderer it is handed to (raytracer.rs:109-111 moves the scene into the renderer): the handle stays usable
  // for the info / debug hooks whatever the renderer does afterwards (destroy, change_scene), and the scene is freed when
  // the last of the two lets go.  `owned` = not handed to a renderer yet.
  std::shared_ptr<Scene> s;
  bool owned = true;
};
struct glz_renderer {
  std::unique_ptr<Renderer> r;
};

extern "C" {

const char* glz_last_error(void) { return g_error.c_str(); }
int glz_last_status(void) { return g_status; }
const char* glz_version(void) { return "glaze-hip 0.1 (gfx950)"; }

// ---- parse -----------------------------------------------------------------------------------
glz_parsed* glz_parse(const char* path) {
  GLZ_GUARD_BEGIN
  if (!path) { fail(GLZ_E_ARG, "path is null"); return nullptr; }
  Error e;
  auto p = Parsed::open(path, e);
  if (!p) { fail(e); return nullptr; }
  glz_parsed* h = new glz_parsed();
  h->p = std::move(p);
  return h;
  GLZ_GUARD_END(nullptr)
}
void glz_parsed_free(glz_parsed* h) { delete h; }

#define GLZ_GETTER(NAME, TYPE, CALL)                          \
  int64_t NAME(glz_parsed* h, TYPE* out, int64_t cap) {       \
    GLZ_GUARD_BEGIN                                           \
    if (!h) return fail(GLZ_E_ARG, "parsed handle is null");  \
    Error e;                                                  \
    const std::vector<TYPE>* v = nullptr;                     \
    if (!(CALL)) return fail(e);                              \
    return copy_out(*v, out, cap);                            \
    GLZ_GUARD_END(GLZ_E_IO)                                   \
  }
GLZ_GETTER(glz_parsed_vertices, glz_vertex, h->p->vertices(v, e))
GLZ_GETTER(glz_parsed_transforms, glz_transform, h->p->transforms(v, e))
GLZ_GETTER(glz_parsed_instances, glz_mesh_instance, h->p->instances(v, e))
GLZ_GETTER(glz_parsed_cameras, glz_camera, h->p->cameras(v, e))
GLZ_GETTER(glz_parsed_materials, glz_material, h->p->materials(v, e))
GLZ_GETTER(glz_parsed_lights, glz_light, h->p->lights(v, e))
#undef GLZ_GETTER

int64_t glz_parsed_meshes(glz_parsed* h, glz_mesh* out, int64_t cap) {
  GLZ_GUARD_BEGIN
  if (!h) return fail(GLZ_E_ARG, "parsed handle is null");
  Error e;
  const std::vector<glz_mesh>* m;
  const std::vector<uint32_t>* idx;
  if (!h->p->meshes(m, idx, e)) return fail(e);
  return copy_out(*m, out, cap);
  GLZ_GUARD_END(GLZ_E_IO)
}
int64_t glz_parsed_indices(glz_parsed* h, uint32_t* out, int64_t cap) {
  GLZ_GUARD_BEGIN
  if (!h) return fail(GLZ_E_ARG, "parsed handle is null");
  Error e;
  const std::vector<glz_mesh>* m;
  const std::vector<uint32_t>* idx;
  if (!h->p->meshes(m, idx, e)) return fail(e);
  return copy_out(*idx, out, cap);
  GLZ_GUARD_END(GLZ_E_IO)
}
int64_t glz_parsed_textures(glz_parsed* h, glz_texture* out, int64_t cap) {
  GLZ_GUARD_BEGIN
  if (!h) return fail(GLZ_E_ARG, "parsed handle is null");
  Error e;
  const std::vector<TextureData>* t;
  if (!h->p->textures(t, e)) return fail(e);
  h->tex_view.clear();
  for (const TextureData& td : *t) {
    glz_texture g = td.info;
    g.pixels = td.level0.data();
    h->tex_view.push_back(g);
  }
  return copy_out(h->tex_view, out, cap);
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_parsed_meta(glz_parsed* h, glz_meta* out) {
  GLZ_GUARD_BEGIN
  if (!h || !out) return fail(GLZ_E_ARG, "null argument");
  Error e;
  bool present = false;
  if (!h->p->meta(*out, present, e)) return fail(e);
  return present ? 0 : 1;
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_serialize(const char* path, const glz_serialize_desc* d) {
  GLZ_GUARD_BEGIN
  if (!path || !d) return fail(GLZ_E_ARG, "null argument");
  SerializeInput in;
  in.vertices = d->vertices; in.n_vertices = d->n_vertices;
  in.indices = d->indices; in.n_indices = d->n_indices;
  in.meshes = d->meshes; in.n_meshes = d->n_meshes;
  in.transforms = d->transforms; in.n_transforms = d->n_transforms;
  in.instances = d->instances; in.n_instances = d->n_instances;
  in.cameras = d->cameras; in.n_cameras = d->n_cameras;
  in.textures = d->textures; in.n_textures = d->n_textures;
  in.materials = d->materials; in.n_materials = d->n_materials;
  in.lights = d->lights; in.n_lights = d->n_lights;
  in.meta = d->meta;
  Error e;
  if (!serialize_scene(path, in, e)) return fail(e);
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_parsed_update(glz_parsed* h, const glz_camera* cameras, int64_t n_cameras, const glz_material* materials, int64_t n_materials,
                      const glz_light* lights, int64_t n_lights, const glz_texture* textures, int64_t n_textures, const glz_meta* meta) {
  GLZ_GUARD_BEGIN
  if (!h) return fail(GLZ_E_ARG, "parsed handle is null");
  Parsed::Update u;
  u.cameras = cameras; u.n_cameras = n_cameras;
  u.materials = materials; u.n_materials = n_materials;
  u.lights = lights; u.n_lights = n_lights;
  u.textures = textures; u.n_textures = n_textures;
  u.meta = meta;
  Error e;
  if (!h->p->update(u, e)) return fail(e);
  h->tex_view.clear();
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_save_image(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
  GLZ_GUARD_BEGIN
  if (!path || !rgba8 || !width || !height) return fail(GLZ_E_ARG, "null argument or empty image");
  const std::string p(path);
  auto ends = [&](const char* suf) { const size_t n = strlen(suf); return p.size() >= n && strcasecmp(p.c_str() + p.size() - n, suf) == 0; };
  std::vector<uint8_t> bytes;
  bool ok;
  if (ends(".png")) ok = png_encode(rgba8, width, height, 4, bytes);
  else if (ends(".jpg") || ends(".jpeg")) ok = jpeg_encode(rgba8, width, height, 4, 75, bytes);
  else return fail(GLZ_E_INVALID_INPUT, "The output image must end with .jpg or .png");
  if (!ok) return fail(GLZ_E_INVALID_INPUT, "image cannot be encoded (JPEG is limited to 65535 x 65535)");
  FILE* f = fopen(path, "wb");
  if (!f) return fail(GLZ_E_IO, "The output file can not be written");
  ok = fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
  ok = (fclose(f) == 0) && ok;
  return ok ? GLZ_OK : fail(GLZ_E_IO, "short write");
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_convert_obj(const char* input_obj, const char* output_glaze, int gen_mipmaps, uint64_t counts[6]) {
  GLZ_GUARD_BEGIN
  if (!input_obj || !output_glaze) return fail(GLZ_E_ARG, "null argument");
  Error e;
  ConvertReport rep;
  if (!convert_obj(input_obj, output_glaze, gen_mipmaps != 0, &rep, e)) return fail(e);
  if (counts) {
    counts[0] = rep.vertices; counts[1] = rep.triangles; counts[2] = rep.meshes;
    counts[3] = rep.materials; counts[4] = rep.textures; counts[5] = rep.lights;
  }
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_converted_file(const char* path) {
  if (!path) return 0;
  FILE* f = fopen(path, "rb");
  if (!f) return 0;
  unsigned char hdr[16];
  size_t n = fread(hdr, 1, 16, f);
  fclose(f);
  return n == 16 && memcmp(hdr, "glaze", 5) == 0;
}

// ---- instance --------------------------------------------------------------------------------
glz_instance* glz_instance_create(int hip_device) {
  GLZ_GUARD_BEGIN
  Error e;
  Instance* i = Instance::create(hip_device, e);
  if (!i) { fail(e); return nullptr; }
  glz_instance* h = new glz_instance();
  h->i.reset(i);
  return h;
  GLZ_GUARD_END(nullptr)
}
void glz_instance_destroy(glz_instance* h) { delete h; }
int glz_instance_device(const glz_instance* h) { return h ? h->i->device : -1; }
void* glz_instance_stream(const glz_instance* h) { return h ? (void*)h->i->stream : nullptr; }
int glz_instance_set_bvh_builder(glz_instance* h, int builder) {
  if (!h || (builder < GLZ_BVH_LBVH || builder > GLZ_BVH_SAH_HOST)) return fail(GLZ_E_INVALID_INPUT, "glz_instance_set_bvh_builder: bad argument");
  h->i->bvh_builder = builder;
  return GLZ_OK;
}

int glz_instance_set_as_levels(glz_instance* h, int mode) {
  if (!h || mode < GLZ_AS_AUTO || mode > GLZ_AS_TWO_LEVEL) return fail(GLZ_E_INVALID_INPUT, "glz_instance_set_as_levels: bad argument");
  h->i->as_levels = mode;
  return GLZ_OK;
}

// ---- scene -----------------------------------------------------------------------------------
glz_scene* glz_scene_create(glz_instance* inst, glz_parsed* parsed) {
  GLZ_GUARD_BEGIN
  if (!inst || !parsed) { fail(GLZ_E_ARG, "null argument"); return nullptr; }
  Error e;
  SceneData data;
  parsed->p->to_scene_data(data, e);
  delete parsed;   // consumed, like the Box moved into RayTraceScene::new
  Scene* s = Scene::create(inst->i.get(), std::move(data), e);
  if (!s) { fail(e); return nullptr; }
  glz_scene* h = new glz_scene();
  h->s.reset(s);
  return h;
  GLZ_GUARD_END(nullptr)
}

glz_scene* glz_scene_create_from_desc(glz_instance* inst, const glz_scene_desc* d) {
  GLZ_GUARD_BEGIN
  if (!inst || !d) { fail(GLZ_E_ARG, "null argument"); return nullptr; }
  SceneData data;
  if (d->n_vertices) data.vertices.assign(d->vertices, d->vertices + d->n_vertices);
  if (d->n_indices) data.indices.assign(d->indices, d->indices + d->n_indices);
  if (d->n_meshes) data.meshes.assign(d->meshes, d->meshes + d->n_meshes);
  if (d->n_transforms) data.transforms.assign(d->transforms, d->transforms + d->n_transforms);
  if (d->n_instances) data.instances.assign(d->instances, d->instances + d->n_instances);
  if (d->n_materials) data.materials.assign(d->materials, d->materials + d->n_materials);
  if (d->n_lights) data.lights.assign(d->lights, d->lights + d->n_lights);
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    const glz_texture& t = d->textures[i];
    if (!t.pixels || t.format < 1 || t.format > 3) { fail(GLZ_E_ARG, "bad texture in scene description"); return nullptr; }
    TextureData td;
    td.info = t;
    const size_t bytes = (size_t)t.width * t.height * (t.format == GLZ_TEX_GRAY ? 1 : 4);
    td.level0.assign(t.pixels, t.pixels + bytes);
    td.info.pixels = nullptr;
    data.textures.push_back(std::move(td));
  }
  data.has_camera = d->camera != nullptr;
  data.camera = d->camera ? *d->camera : default_camera();
  data.has_meta = d->meta != nullptr;
  data.meta = d->meta ? *d->meta : default_meta();
  if (data.camera.type > GLZ_CAMERA_ORTHOGRAPHIC) { fail(GLZ_E_ARG, "unknown camera type"); return nullptr; }
  Error e;
  Scene* s = Scene::create(inst->i.get(), std::move(data), e);
  if (!s) { fail(e); return nullptr; }
  glz_scene* h = new glz_scene();
  h->s.reset(s);
  return h;
  GLZ_GUARD_END(nullptr)
}

void glz_scene_destroy(glz_scene* h) {
  if (!h) return;
  delete h;
}
int glz_scene_get_info(const glz_scene* h, glz_scene_info* out) {
  if (!h || !h->s || !out) return fail(GLZ_E_ARG, "null argument");
  *out = h->s->info;
  return GLZ_OK;
}
int glz_scene_camera(const glz_scene* h, glz_camera* out) {
  if (!h || !h->s || !out) return fail(GLZ_E_ARG, "null argument");
  *out = h->s->data.camera;
  return GLZ_OK;
}

// ---- renderer --------------------------------------------------------------------------------
glz_renderer* glz_renderer_create(glz_instance* inst, glz_scene* scene, uint32_t w, uint32_t h) {
  GLZ_GUARD_BEGIN
  if (!inst) { fail(GLZ_E_ARG, "instance is null"); return nullptr; }
  if (scene && !scene->owned) { fail(GLZ_E_ARG, "scene already belongs to a renderer"); return nullptr; }
  Error e;
  if (scene) scene->owned = false;   // moved into the renderer (raytracer.rs:109-111); the handle stays valid for debug hooks
  Renderer* r = Renderer::create(inst->i.get(), scene ? scene->s : std::shared_ptr<Scene>(), w, h, e);
  if (!r) {
    fail(e);
    return nullptr;
  }
  glz_renderer* hr = new glz_renderer();
  hr->r.reset(r);
  return hr;
  GLZ_GUARD_END(nullptr)
}
void glz_renderer_destroy(glz_renderer* h) { delete h; }

#define GLZ_R(h) if (!(h)) return fail(GLZ_E_ARG, "renderer is null"); Error e
#define GLZ_RET(ok) return (ok) ? GLZ_OK : fail(e)

int glz_renderer_set_integrator(glz_renderer* h, int i) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_integrator(i, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_set_exposure(glz_renderer* h, float x) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_exposure(x)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_update_camera(glz_renderer* h, const glz_camera* c) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!c) return fail(GLZ_E_ARG, "camera is null");
  GLZ_RET(h->r->update_camera(*c, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_change_resolution(glz_renderer* h, uint32_t w, uint32_t hh) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->change_resolution(w, hh, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_change_scene(glz_renderer* h, glz_scene* s) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!s || !s->s || !s->owned) return fail(GLZ_E_ARG, "scene is null or already owned by a renderer");
  s->owned = false;
  GLZ_RET(h->r->change_scene(s->s, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_update_materials_and_lights(glz_renderer* h, const glz_material* m, uint32_t nm, const glz_light* l, uint32_t nl,
                                             const glz_texture* t, uint32_t nt) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if ((!m && nm) || (!l && nl) || (t && !nt)) return fail(GLZ_E_ARG, "null array");
  GLZ_RET(h->r->update_materials_and_lights(m, nm, l, nl, t, nt, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_refresh_binded_textures(glz_renderer* h, const glz_texture* t, uint32_t nt) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!t || !nt) return fail(GLZ_E_ARG, "null array");
  GLZ_RET(h->r->refresh_binded_textures(t, nt, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_wait_idle(glz_renderer* h) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->wait_idle(e)); GLZ_GUARD_END(GLZ_E_IO) }
uint32_t glz_renderer_steps_per_sample(const glz_renderer* h) { return h ? h->r->steps_per_sample() : 0; }
int glz_renderer_draw(glz_renderer* h, size_t spp, void (*cb)(void*), void* user, uint8_t* out) {
  GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->draw(spp, cb, user, out, e)); GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_restart(glz_renderer* h) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->restart()); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_step(glz_renderer* h, uint32_t n) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->step(n, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_read_rgba8(glz_renderer* h, uint8_t* out) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!out) return fail(GLZ_E_ARG, "output is null");
  GLZ_RET(h->r->read_rgba8(out, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_set_texture_lod(glz_renderer* h, int mode) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_texture_lod(mode, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_set_seed(glz_renderer* h, uint64_t s) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_seed(s)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_set_depth(glz_renderer* h, uint32_t d) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_depth(d, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_read_hdr(glz_renderer* h, float* out) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!out) return fail(GLZ_E_ARG, "output is null");
  GLZ_RET(h->r->read_frame(false, out, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_read_result(glz_renderer* h, float* out) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!out) return fail(GLZ_E_ARG, "output is null");
  GLZ_RET(h->r->read_frame(true, out, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_launch_constants(glz_renderer* h, uint32_t launch, uint32_t* seed, float off[2]) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!seed || !off) return fail(GLZ_E_ARG, "output is null");
  GLZ_RET(h->r->launch_constants(launch, seed, off));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_push_constants(glz_renderer* h, float out[32]) {
  if (!h || !out) return fail(GLZ_E_ARG, "null argument");
  h->r->push_constants(out);
  return GLZ_OK;
}
int glz_renderer_set_partition(glz_renderer* h, uint32_t rank, uint32_t world) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_partition(rank, world, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_set_devices(glz_renderer* h, const int* devices, int n) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  GLZ_RET(h->r->set_devices(devices, n, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_set_launch_mode(glz_renderer* h, int mode) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_launch_mode(mode, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_launch_mode(glz_renderer* h) { return h ? (h->r->path_mode() ? GLZ_LAUNCH_PATH : GLZ_LAUNCH_TWO_KERNELS) : 0; }
int glz_renderer_set_node_width(glz_renderer* h, int width) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_node_width(width, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_node_width(glz_renderer* h) { return h ? (h->r->wide8() ? 8 : 4) : 0; }
int glz_renderer_set_chains(glz_renderer* h, uint32_t n) { GLZ_GUARD_BEGIN GLZ_R(h); GLZ_RET(h->r->set_chains(n, e)); GLZ_GUARD_END(GLZ_E_IO) }
int glz_renderer_export_device(glz_renderer* h, int which, void* dev) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!dev) return fail(GLZ_E_ARG, "device buffer is null");
  GLZ_RET(h->r->export_device(which, dev, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
uint64_t glz_renderer_packed_pixels(glz_renderer* h, uint32_t rank, uint32_t world) {
  if (!h) return 0;
  return (uint64_t)Renderer::packed_count(h->r->width(), h->r->height(), rank, world);
}
int glz_renderer_export_packed(glz_renderer* h, int which, void* dev) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!dev) return fail(GLZ_E_ARG, "device buffer is null");
  GLZ_RET(h->r->export_packed(which, dev, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_scatter_packed(glz_renderer* h, uint32_t rank, uint32_t world, const void* packed, void* frame) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!packed || !frame) return fail(GLZ_E_ARG, "device buffer is null");
  GLZ_RET(h->r->scatter_packed(rank, world, packed, frame, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_scatter_packed_all(glz_renderer* h, uint32_t world, const void* packed, uint64_t stride_pixels, void* frame) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!packed || !frame) return fail(GLZ_E_ARG, "device buffer is null");
  GLZ_RET(h->r->scatter_packed_all(world, packed, stride_pixels, frame, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_tonemap_device(glz_renderer* h, const void* dev, uint8_t* out) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!dev || !out) return fail(GLZ_E_ARG, "null argument");
  GLZ_RET(h->r->tonemap_device(dev, out, e));
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_renderer_enable_counters(glz_renderer* h, int flags) {
  if (!h) return fail(GLZ_E_ARG, "renderer is null");
  h->r->enable_counters(flags);
  return GLZ_OK;
}
int glz_renderer_get_stats(glz_renderer* h, glz_render_stats* out) {
  GLZ_GUARD_BEGIN GLZ_R(h);
  if (!out) return fail(GLZ_E_ARG, "output is null");
  GLZ_RET(h->r->get_stats(out, e));
  GLZ_GUARD_END(GLZ_E_IO)
}

// ---- debug / parity hooks --------------------------------------------------------------------

int glz_debug_trace_closest(glz_scene* h, const float* o, const float* d, uint64_t n, float tmin, float* t, uint32_t* tri, uint32_t* inst, float* u,
                            float* v) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s || !o || !d || !t || !tri || !inst || !u || !v) return fail(GLZ_E_ARG, "null argument");
  if (n == 0) return GLZ_OK;
  if (n > 0x7FFFFFFFull) return fail(GLZ_E_ARG, "too many rays");
  Scene* s = h->s.get();
  Error e;
  if (!hip_ok(hipSetDevice(s->instance->device), "hipSetDevice", e)) return fail(e);
  hipStream_t st = s->instance->stream;
  DeviceBuffer<float> d_o, d_d, d_t, d_u, d_v;
  DeviceBuffer<uint32_t> d_tri, d_inst, d_ovf;
  if (!to_device(d_o, o, n * 3, st, e) || !to_device(d_d, d, n * 3, st, e)) return fail(e);
  if (!hip_ok(d_t.alloc(n), "alloc", e) || !hip_ok(d_u.alloc(n), "alloc", e) || !hip_ok(d_v.alloc(n), "alloc", e) ||
      !hip_ok(d_tri.alloc(n), "alloc", e) || !hip_ok(d_inst.alloc(n), "alloc", e) || !hip_ok(d_ovf.alloc((n + 512) * s->stack_overflow_depth), "alloc", e))
    return fail(e);
  if (!hip_ok(launch_debug_closest(st, s->dev, d_o.ptr, d_d.ptr, (uint32_t)n, tmin, d_t.ptr, d_tri.ptr, d_inst.ptr, d_u.ptr, d_v.ptr, d_ovf.ptr,
                                   s->stack_overflow_depth), "k_debug_closest", e))
    return fail(e);
  (void)hipMemcpyAsync(t, d_t.ptr, n * 4, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(tri, d_tri.ptr, n * 4, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(inst, d_inst.ptr, n * 4, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(u, d_u.ptr, n * 4, hipMemcpyDeviceToHost, st);
  (void)hipMemcpyAsync(v, d_v.ptr, n * 4, hipMemcpyDeviceToHost, st);
  if (!hip_ok(hipStreamSynchronize(st), "debug trace", e)) return fail(e);
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}

int glz_debug_trace_any(glz_scene* h, const float* o, const float* d, const float* tmax, uint64_t n, float tmin, uint8_t* out) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s || !o || !d || !tmax || !out) return fail(GLZ_E_ARG, "null argument");
  if (n == 0) return GLZ_OK;
  if (n > 0x7FFFFFFFull) return fail(GLZ_E_ARG, "too many rays");
  Scene* s = h->s.get();
  Error e;
  if (!hip_ok(hipSetDevice(s->instance->device), "hipSetDevice", e)) return fail(e);
  hipStream_t st = s->instance->stream;
  DeviceBuffer<float> d_o, d_d, d_tm;
  DeviceBuffer<uint8_t> d_out;
  DeviceBuffer<uint32_t> d_ovf;
  if (!to_device(d_o, o, n * 3, st, e) || !to_device(d_d, d, n * 3, st, e) || !to_device(d_tm, tmax, n, st, e)) return fail(e);
  if (!hip_ok(d_out.alloc(n), "alloc", e) || !hip_ok(d_ovf.alloc((n + 512) * s->stack_overflow_depth), "alloc", e)) return fail(e);
  if (!hip_ok(launch_debug_any(st, s->dev, d_o.ptr, d_d.ptr, d_tm.ptr, (uint32_t)n, tmin, d_out.ptr, d_ovf.ptr, s->stack_overflow_depth), "k_debug_any", e))
    return fail(e);
  (void)hipMemcpyAsync(out, d_out.ptr, n, hipMemcpyDeviceToHost, st);
  if (!hip_ok(hipStreamSynchronize(st), "debug trace", e)) return fail(e);
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}

int64_t glz_debug_read_derivatives(glz_scene* h, float* out, int64_t cap_tris) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  Scene* s = h->s.get();
  uint32_t ntri = 0;
  for (const glz_mesh& m : s->data.meshes) ntri = std::max<uint32_t>(ntri, (m.index_offset + m.index_count) / 3);
  if (out && cap_tris > 0 && ntri > 0) {
    Error e;
    if (!hip_ok(hipSetDevice(s->instance->device), "hipSetDevice", e)) return fail(e);
    const size_t n = (size_t)std::min<int64_t>(cap_tris, ntri);
    if (!hip_ok(hipMemcpy(out, s->dev.derivatives, n * 48, hipMemcpyDeviceToHost), "read derivatives", e)) return fail(e);
  }
  return ntri;
  GLZ_GUARD_END(GLZ_E_IO)
}
int64_t glz_debug_read_rt_materials(glz_scene* h, void* out, int64_t cap) {
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  const int64_t n = (int64_t)h->s->h_materials.size() * (int64_t)sizeof(RTMaterial);
  if (out && cap > 0) {
    Error e;
    if (!hip_ok(hipSetDevice(h->s->instance->device), "hipSetDevice", e)) return fail(e);
    if (!hip_ok(hipMemcpy(out, h->s->dev.materials, (size_t)std::min(n, cap), hipMemcpyDeviceToHost), "read materials", e)) return fail(e);
  }
  return n;
}
int64_t glz_debug_read_rt_lights(glz_scene* h, void* out, int64_t cap) {
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  const int64_t n = (int64_t)h->s->h_lights.size() * (int64_t)sizeof(RTLight);
  if (out && cap > 0) {
    Error e;
    if (!hip_ok(hipSetDevice(h->s->instance->device), "hipSetDevice", e)) return fail(e);
    if (!hip_ok(hipMemcpy(out, h->s->dev.lights, (size_t)std::min(n, cap), hipMemcpyDeviceToHost), "read lights", e)) return fail(e);
  }
  return n;
}
int64_t glz_debug_read_sky(glz_scene* h, float* out, int64_t cap) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  Scene* s = h->s.get();
  std::vector<float> buf(36 + 4 + s->h_sky_marginal.size());
  memcpy(buf.data(), &s->h_sky, 144);
  memcpy(buf.data() + 36, &s->h_sky_header, 16);
  if (!s->h_sky_marginal.empty()) {
    Error e;
    if (!hip_ok(hipSetDevice(s->instance->device), "hipSetDevice", e)) return fail(e);
    if (!hip_ok(hipMemcpy(buf.data() + 40, s->dev.sky_marginal, s->h_sky_marginal.size() * 4, hipMemcpyDeviceToHost), "read sky", e)) return fail(e);
  }
  if (out && cap > 0) memcpy(out, buf.data(), (size_t)std::min<int64_t>(cap, (int64_t)buf.size()) * 4);
  return (int64_t)buf.size();
  GLZ_GUARD_END(GLZ_E_IO)
}
int64_t glz_debug_read_bvh(glz_scene* h, void* nodes_out, int64_t cap_nodes, void* tris_out, int64_t cap_tris) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  Scene* s = h->s.get();
  Error e;
  if (!hip_ok(hipSetDevice(s->instance->device), "hipSetDevice", e)) return fail(e);
  const int64_t nn = s->info.bvh_nodes, nt = (int64_t)s->info.n_as_triangles;
  if (nodes_out && cap_nodes > 0 && nn > 0 &&
      !hip_ok(hipMemcpy(nodes_out, s->dev.bvh_nodes, (size_t)std::min(cap_nodes, nn) * sizeof(BvhNode4), hipMemcpyDeviceToHost), "read nodes", e))
    return fail(e);
  if (nodes_out && cap_nodes > 0 && nn > 0 && s->dev.bvh_quads && !s->dev.two_level) {
    // The flattened build links its leaves by NUMBER (the tracer reads one 64-byte BvhQuad per leaf, which names the leaf's first
    // triangle slot); what this hook hands out is the structure as its readers walk it -- nodes whose leaf links are ~(first slot in
    // the triangle array), as in a two-level scene's meshes -- so the links are translated here.
    std::vector<BvhQuad> quads(s->d_quads_count());
    if (!quads.empty() && !hip_ok(hipMemcpy(quads.data(), s->dev.bvh_quads, quads.size() * sizeof(BvhQuad), hipMemcpyDeviceToHost), "read leaf records", e)) return fail(e);
    BvhNode4* nd = static_cast<BvhNode4*>(nodes_out);
    for (int64_t i = 0; i < std::min(cap_nodes, nn); ++i)
      for (int k = 0; k < 4; ++k) {
        const int link = (int)nd[i].w[12 + k];
        if (link < 0 && (size_t)~link < quads.size()) nd[i].w[12 + k] = (uint32_t)~(int)quads[(size_t)~link].slot;
      }
  }
  if (nodes_out && cap_nodes > 0 && nn > 0 && s->dev.bvh_quads && s->dev.two_level) {
    // the same for the meshes of a two-level scene: leaf number within the mesh -> ~(first slot within the mesh's triangles)
    std::vector<BvhQuad> quads(s->d_quads_count());
    if (!quads.empty() && !hip_ok(hipMemcpy(quads.data(), s->dev.bvh_quads, quads.size() * sizeof(BvhQuad), hipMemcpyDeviceToHost), "read leaf records", e)) return fail(e);
    BvhNode4* nd = static_cast<BvhNode4*>(nodes_out);
    for (const Scene::MeshRange& m : s->h_mesh_ranges)
      for (int64_t i = m.node_base; i < std::min<int64_t>(std::min(cap_nodes, nn), (int64_t)m.node_base + m.n_nodes); ++i)
        for (int k = 0; k < 4; ++k) {
          const int link = (int)nd[i].w[12 + k];
          const size_t q = (size_t)m.quad_base + (size_t)~link;
          if (link < 0 && link != kBvhEmptyChild && q < quads.size()) nd[i].w[12 + k] = (uint32_t)~(int)quads[q].slot;
        }
  }
  if (tris_out && cap_tris > 0 && nt > 0 &&
      !hip_ok(hipMemcpy(tris_out, s->dev.bvh_tris, (size_t)std::min(cap_tris, nt) * sizeof(BvhTri), hipMemcpyDeviceToHost), "read tris", e))
    return fail(e);
  return nn;
  GLZ_GUARD_END(GLZ_E_IO)
}

int64_t glz_debug_read_bvh8(glz_scene* h, void* nodes_out, int64_t cap_nodes) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  Scene* s = h->s.get();
  Error e;
  if (!hip_ok(hipSetDevice(s->instance->device), "hipSetDevice", e)) return fail(e);
  const int64_t nn = s->dev.bvh_nodes8 ? (int64_t)s->info.bvh_nodes8 : 0;
  if (nodes_out && cap_nodes > 0 && nn > 0) {
    if (!hip_ok(hipMemcpy(nodes_out, s->dev.bvh_nodes8, (size_t)std::min(cap_nodes, nn) * sizeof(BvhNode8), hipMemcpyDeviceToHost), "read 8-wide nodes", e)) return fail(e);
    std::vector<BvhQuad> quads(s->d_quads_count());   // leaf number -> ~(first slot), as glz_debug_read_bvh hands its links out
    if (!quads.empty() && !hip_ok(hipMemcpy(quads.data(), s->dev.bvh_quads, quads.size() * sizeof(BvhQuad), hipMemcpyDeviceToHost), "read leaf records", e)) return fail(e);
    BvhNode8* nd = static_cast<BvhNode8*>(nodes_out);
    for (int64_t i = 0; i < std::min(cap_nodes, nn); ++i)
      for (int k = 0; k < 8; ++k) {
        const int link = (int)nd[i].w[24 + k];
        if (link < 0 && (size_t)~link < quads.size()) nd[i].w[24 + k] = (uint32_t)~(int)quads[(size_t)~link].slot;
      }
  }
  return nn;
  GLZ_GUARD_END(GLZ_E_IO)
}

int64_t glz_debug_read_texture_level(glz_scene* h, uint32_t texture, uint32_t level, uint8_t* out, int64_t cap, uint32_t* width, uint32_t* height) {
  GLZ_GUARD_BEGIN
  if (!h || !h->s) return fail(GLZ_E_ARG, "scene is null");
  Error e;
  std::vector<uint8_t> px;
  uint32_t w = 0, hh = 0;
  if (!h->s->read_mip_level(texture, level, px, w, hh, e)) return fail(e);
  if (width) *width = w;
  if (height) *height = hh;
  if (out && cap > 0 && !px.empty()) memcpy(out, px.data(), (size_t)std::min<int64_t>(cap, (int64_t)px.size()));
  return (int64_t)px.size();
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_debug_tonemap(glz_instance* inst, const float* rgba32f, uint64_t n, uint8_t* out) {
  GLZ_GUARD_BEGIN
  if (!inst || !rgba32f || !out) return fail(GLZ_E_ARG, "null argument");
  if (n == 0) return GLZ_OK;
  if (n > 0x7FFFFFFFull) return fail(GLZ_E_ARG, "too many pixels");
  Error e;
  if (!hip_ok(hipSetDevice(inst->i->device), "hipSetDevice", e)) return fail(e);
  hipStream_t st = inst->i->stream;
  DeviceBuffer<float4> d_in;
  DeviceBuffer<uchar4> d_out;
  DeviceBuffer<float> d_thr;
  float thr[256];
  host::srgb8_thresholds(thr);
  if (!hip_ok(d_in.upload(reinterpret_cast<const float4*>(rgba32f), n, st), "upload", e) || !hip_ok(d_thr.upload(thr, 256, st), "upload", e) ||
      !hip_ok(d_out.alloc(n), "alloc", e))
    return fail(e);
  if (!hip_ok(launch_tonemap(st, (uint32_t)n, d_in.ptr, d_thr.ptr, d_out.ptr), "k_tonemap", e)) return fail(e);
  (void)hipMemcpyAsync(out, d_out.ptr, n * 4, hipMemcpyDeviceToHost, st);
  if (!hip_ok(hipStreamSynchronize(st), "debug tonemap", e)) return fail(e);
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}

int glz_renderer_device_count(glz_renderer* h) { return h ? (int)h->r->device_count() : 0; }
int glz_renderer_device_scene_info(glz_renderer* h, int i, glz_scene_info* out) {
  if (!h || !out) return fail(GLZ_E_ARG, "null argument");
  const Scene* s = h->r->device_scene(i);
  if (!s) return fail(GLZ_E_ARG, "glz_renderer_device_scene_info: no such device");
  *out = s->info;
  return GLZ_OK;
}
int glz_rccl_version(void) {
  GLZ_GUARD_BEGIN
  std::string why;
  const Rccl* nc = Rccl::get(why);
  if (!nc) return fail(GLZ_E_DEVICE, why.c_str());
  int version = 0;
  if (nc->GetVersion(&version) != ncclSuccess) return fail(GLZ_E_DEVICE, "ncclGetVersion failed");
  return version;
  GLZ_GUARD_END(GLZ_E_IO)
}

int glz_debug_rccl_selftest(glz_instance* inst, uint64_t n, int* version_out) {
  GLZ_GUARD_BEGIN
  if (!inst || n == 0 || n > (1ull << 30)) return fail(GLZ_E_ARG, "bad argument");
  std::string why;
  const Rccl* nc = Rccl::get(why);
  if (!nc) return fail(GLZ_E_DEVICE, why.c_str());
  Error e;
  if (!hip_ok(hipSetDevice(inst->i->device), "hipSetDevice", e)) return fail(e);
  int version = 0;
  (void)nc->GetVersion(&version);
  if (version_out) *version_out = version;
  hipStream_t st = inst->i->stream;
  std::vector<float> host(n), back(n);
  uint32_t x = 12345u;
  for (uint64_t i = 0; i < n; ++i) {   // arbitrary bit patterns that are finite floats
    x = x * 1664525u + 1013904223u;
    const uint32_t bits = (x & 0x807FFFFFu) | (((x >> 23) % 200u + 20u) << 23);
    memcpy(&host[i], &bits, 4);
  }
  DeviceBuffer<float> send, recv;
  if (!hip_ok(send.upload(host.data(), n, st), "upload", e) || !hip_ok(recv.alloc(n), "alloc", e)) return fail(e);
  if (!hip_ok(hipMemsetAsync(recv.ptr, 0, n * 4, st), "memset", e)) return fail(e);
  ncclComm_t comm = nullptr;
  const int dev = inst->i->device;
  ncclResult_t r = nc->CommInitAll(&comm, 1, &dev);
  if (r != ncclSuccess) return fail(GLZ_E_DEVICE, (std::string("ncclCommInitAll: ") + nc->GetErrorString(r)).c_str());
  r = nc->Reduce(send.ptr, recv.ptr, n, ncclFloat, ncclSum, 0, comm, st);
  bool ok = r == ncclSuccess;
  std::string msg = ok ? "" : std::string("ncclReduce: ") + nc->GetErrorString(r);
  if (ok) {
    ok = hip_ok(hipMemcpyAsync(back.data(), recv.ptr, n * 4, hipMemcpyDeviceToHost, st), "read back", e) && hip_ok(hipStreamSynchronize(st), "ncclReduce", e);
    if (!ok) msg = e.msg;
  }
  (void)nc->CommDestroy(comm);
  if (!ok) return fail(GLZ_E_DEVICE, msg.c_str());
  if (memcmp(host.data(), back.data(), n * 4) != 0) return fail(GLZ_E_DEVICE, "ncclReduce on a one-rank communicator changed the data");
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}

// ---- host logic without a device ---------------------------------------------------------------
int glz_host_launch_constants(uint64_t seed, uint32_t launch, uint32_t* seed_out, float offset[2]) {
  GLZ_GUARD_BEGIN
  if (!seed_out || !offset) return fail(GLZ_E_ARG, "output is null");
  host::SeedStream rng(seed);
  host::WorkScheduler ws;
  uint32_t s = 0;
  float o[2] = {0, 0};
  for (uint32_t i = 0; i <= launch; ++i) {
    s = rng.next();
    float ahead[2] = {-1.0f, -1.0f};
    ws.peek(ahead);   // what the launch before this one was told about it (FrameData::next_pixel_offset): must be what next() now hands out
    ws.next(o);
    if (memcmp(ahead, o, sizeof(o)) != 0) return fail(GLZ_E_IO, "WorkScheduler::peek disagrees with next()");
  }
  *seed_out = s;
  offset[0] = o[0];
  offset[1] = o[1];
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_host_push_constants(const glz_camera* camera, uint32_t width, uint32_t height, float out32[32]) {
  if (!camera || !out32 || width == 0 || height == 0 || camera->type > GLZ_CAMERA_ORTHOGRAPHIC) return fail(GLZ_E_ARG, "bad argument");
  host::push_constants(*camera, width, height, out32, out32 + 16);
  return GLZ_OK;
}
int glz_host_tile_owner(uint32_t width, uint32_t height, uint32_t world, uint16_t* owner_out) {
  if (!owner_out || world == 0 || world > 65535) return fail(GLZ_E_ARG, "bad argument");
  const uint32_t tiles_x = (width + 63) / 64;
  for (uint32_t y = 0; y < height; ++y)
    for (uint32_t x = 0; x < width; ++x) owner_out[(size_t)y * width + x] = (uint16_t)(((y / 64) * tiles_x + x / 64) % world);
  return GLZ_OK;
}
int glz_host_chain_owner(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t chains, uint16_t* owner_out) {
  if (world == 0 || rank >= world || chains > 16 || !width || !height) return fail(GLZ_E_ARG, "bad argument");
  const uint32_t S = Renderer::chains_for(width, height, rank, world, chains);
  const uint32_t tiles_x = (width + 63) / 64;
  if (owner_out)
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x) {
        const uint32_t t = (y / 64) * tiles_x + x / 64;
        owner_out[(size_t)y * width + x] = t % world == rank ? (uint16_t)((t % (world * S)) / world) : (uint16_t)0xFFFF;
      }
  return (int)S;
}

int64_t glz_host_mip_level(const glz_texture* t, uint32_t level, uint8_t* out, int64_t cap, uint32_t* width, uint32_t* height) {
  GLZ_GUARD_BEGIN
  if (!t || !t->pixels || !t->width || !t->height || t->format < 1 || t->format > 3) return fail(GLZ_E_ARG, "bad texture");
  std::vector<host::MipLevel> given(1);
  given[0].width = t->width;
  given[0].height = t->height;
  given[0].pixels.assign(t->pixels, t->pixels + (size_t)t->width * t->height * (t->format == GLZ_TEX_GRAY ? 1 : 4));
  const std::vector<host::MipLevel> chain = host::build_mip_chain(t->format, std::move(given));
  if (level >= chain.size()) {
    if (width) *width = 0;
    if (height) *height = 0;
    return 0;
  }
  const host::MipLevel& m = chain[level];
  if (width) *width = m.width;
  if (height) *height = m.height;
  if (out && cap > 0) memcpy(out, m.pixels.data(), (size_t)std::min<int64_t>(cap, (int64_t)m.pixels.size()));
  return (int64_t)m.pixels.size();
  GLZ_GUARD_END(GLZ_E_IO)
}
int glz_host_srgb8_thresholds(float thresholds_out[256]) {
  if (!thresholds_out) return fail(GLZ_E_ARG, "output is null");
  host::srgb8_thresholds(thresholds_out);
  return GLZ_OK;
}

}  // extern "C"

int glz_host_build_sah(uint32_t n, const float* box_lo, const float* box_hi, int32_t* children_out, int32_t* parent_out) {
  GLZ_GUARD_BEGIN
  if (n < 2 || !box_lo || !box_hi || !children_out || !parent_out) return fail(GLZ_E_ARG, "glz_host_build_sah: bad argument");
  static_assert(sizeof(float4) == 16 && sizeof(int2) == 8, "layout");
  build_sah_host(n, reinterpret_cast<const float4*>(box_lo), reinterpret_cast<const float4*>(box_hi), reinterpret_cast<int2*>(children_out), parent_out);
  return GLZ_OK;
  GLZ_GUARD_END(GLZ_E_IO)
}
