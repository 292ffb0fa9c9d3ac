// RayTraceRenderer on HIP -- see renderer.h.  Reference: lib/src/vulkan/raytracer.rs.
#include "renderer.h"

#include <cmath>
#include <cstring>

namespace glz {

namespace {
constexpr uint32_t kTile = 64;
}

Renderer* Renderer::create(Instance* inst, Scene* scene, uint32_t w, uint32_t h, Error& err) {
  std::unique_ptr<Renderer> r(new Renderer());
  r->inst_ = inst;
  if (!hip_ok(hipSetDevice(inst->device), "hipSetDevice", err)) {
    delete scene;
    return nullptr;
  }
  if (!scene) {
    // RayTraceRenderer::new(.., None, ..) renders an empty scene (raytracer.rs:170-174, NoScene)
    SceneData empty;
    empty.camera = default_camera();
    empty.meta = default_meta();
    scene = Scene::create(inst, std::move(empty), err);
    if (!scene) return nullptr;
  }
  r->scene_.reset(scene);
  if (w == 0 || h == 0) {
    err.code = GLZ_E_ARG;
    err.msg = "resolution must be non-zero";
    return nullptr;
  }
  r->w_ = w;
  r->h_ = h;
  r->map_.rank = 0;
  r->map_.world = 1;
  r->camera_ = scene->data.camera;
  r->exposure_ = scene->data.meta.exposure;
  host::push_constants(r->camera_, w, h, r->cam_.camera2world, r->cam_.screen2camera);
  if (!r->allocate(err)) return nullptr;
  return r.release();
}

Renderer::~Renderer() {
  if (inst_) (void)hipSetDevice(inst_->device);
  if (inst_ && inst_->stream) (void)hipStreamSynchronize(inst_->stream);
  for (auto& s : pending_events_)
    for (auto& e : s.e) (void)hipEventDestroy(e);
  for (auto& s : free_events_)
    for (auto& e : s.e) (void)hipEventDestroy(e);
}

bool Renderer::allocate(Error& err) {
  map_.width = w_;
  map_.height = h_;
  map_.tiles_x = (w_ + kTile - 1) / kTile;
  map_.tiles_y = (h_ + kTile - 1) / kTile;
  const uint32_t tiles = map_.tiles_x * map_.tiles_y;
  map_.n_local_tiles = tiles > map_.rank ? (tiles - map_.rank + map_.world - 1) / map_.world : 0;
  map_.n_local_pixels = map_.n_local_tiles * kTile * kTile;
  const size_t n = map_.n_local_pixels;
  // shadow-ray queue: 8 shards of ceil(blocks/8)*256 entries (kernels_render.hip, queue_capacity)
  const size_t n_queue = (((n + 255) / 256 + 7) / 8) * 256 * 8;
  DeviceBuffer<float4>* bufs[] = {&ray_o_, &ray_d_, &imp_[0], &imp_[1], &imp_[2], &imp_[3], &hit_, &sh_o_, &sh_d_, &contrib_, &cumulative_, &result_};
  for (auto* b : bufs)
    if (!hip_ok(b->alloc((b == &sh_o_ || b == &sh_d_ || b == &contrib_) ? n_queue : n), "alloc path state", err)) return false;
  const uint32_t od = scene_->stack_overflow_depth;
  if (!hip_ok(overflow_.alloc((2 * n + 512) * od), "alloc traversal spill", err)) return false;   // one slot per lane of the largest k_trace grid
  if (!hip_ok(frame_tmp_.alloc((size_t)w_ * h_), "alloc frame", err)) return false;
  if (!hip_ok(rgba8_.alloc((size_t)w_ * h_), "alloc rgba8", err)) return false;
  if (!hip_ok(counters_.alloc(1), "alloc counters", err)) return false;
  if (!hip_ok(queue_count_.alloc(2 * kQueueSetWords), "alloc queue counters", err)) return false;
  shadow_pending_ = false;
  request_new_frame_ = true;
  return true;
}

// the fill_buffer / clear_color_image of a new frame (raytracer.rs:506-532) + scheduler rewind (:483-485)
bool Renderer::reset_buffers(Error& err) {
  hipStream_t st = inst_->stream;
  const size_t bytes = sizeof(float4) * (size_t)map_.n_local_pixels;
  DeviceBuffer<float4>* zero[] = {&ray_o_, &ray_d_, &imp_[0], &imp_[1], &imp_[2], &imp_[3], &cumulative_, &result_, &contrib_};
  for (auto* b : zero)
    if (bytes && !hip_ok(hipMemsetAsync(b->ptr, 0, bytes, st), "clear path state", err)) return false;
  if (!hip_ok(hipMemsetAsync(counters_.ptr, 0, sizeof(TraceCounters), st), "clear counters", err)) return false;
  if (!hip_ok(hipMemsetAsync(queue_count_.ptr, 0, sizeof(uint32_t) * 2 * kQueueSetWords, st), "clear queue counters", err)) return false;
  shadow_pending_ = false;   // queued shadow rays of the abandoned frame are dropped with it
  sched_.rewind();
  rng_.reseed(seed_);   // build-defined: a restart replays the same seed stream (the reference keeps drawing from entropy)
  launches_ = 0;
  render_ms_ = closest_ms_ = shade_ms_ = shadow_ms_ = 0;
  for (auto& s : pending_events_) free_events_.push_back(s);
  pending_events_.clear();
  request_new_frame_ = false;
  return true;
}

void Renderer::fill_args(LaunchArgs& a) const {
  a.scene = scene_->dev;
  a.st.ray_o = ray_o_.ptr;
  a.st.ray_d = ray_d_.ptr;
  for (int q = 0; q < 4; ++q) a.st.imp[q] = imp_[q].ptr;
  a.st.hit = hit_.ptr;
  a.st.sh_o = sh_o_.ptr;
  a.st.sh_d = sh_d_.ptr;
  a.st.contrib = contrib_.ptr;
  a.st.queue_count = queue_count_.ptr;
  a.st.cumulative = cumulative_.ptr;
  a.st.result = result_.ptr;
  a.st.overflow = overflow_.ptr;
  a.st.overflow_depth = scene_->stack_overflow_depth;
  a.map = map_;
  a.cam = cam_;
  a.counters = counting_ ? counters_.ptr : nullptr;
  a.do_closest = a.do_shadow = 0;
  a.shade_set = pending_set_ ^ 1u;
  a.shadow_exposure = pending_exposure_;
}

bool Renderer::acquire_events(EventSet& ev, Error& err) {
  if (free_events_.empty()) {
    if (pending_events_.size() >= 64) {
      // resolve and recycle the pending sets (without the flush get_stats would do)
      if (!hip_ok(hipEventSynchronize(pending_events_.back().e[pending_events_.back().flush ? 1 : 2]), "hipEventSynchronize", err)) return false;
      for (auto& s : pending_events_) {
        float a = 0, b = 0;
        (void)hipEventElapsedTime(&a, s.e[0], s.e[1]);
        if (s.flush) {
          shadow_ms_ += a;
        } else {
          (void)hipEventElapsedTime(&b, s.e[1], s.e[2]);
          closest_ms_ += a;
          shade_ms_ += b;
        }
        render_ms_ += (double)a + b;
        free_events_.push_back(s);
      }
      pending_events_.clear();
    }
    if (free_events_.empty()) {
      EventSet fresh{};
      for (auto& e : fresh.e)
        if (!hip_ok(hipEventCreate(&e), "hipEventCreate", err)) return false;
      free_events_.push_back(fresh);
    }
  }
  ev = free_events_.back();
  free_events_.pop_back();
  return true;
}

// Stand-alone shadow pass for the rays the last launch queued: run before anything observes the accumulators.
bool Renderer::flush_shadows(Error& err) {
  if (!shadow_pending_) return true;
  hipStream_t st = inst_->stream;
  LaunchArgs a;
  fill_args(a);
  memset(&a.frame, 0, sizeof(a.frame));
  a.do_shadow = 1;
  EventSet ev{};
  if (profile_kernels_) {
    if (!acquire_events(ev, err)) return false;
    ev.flush = true;
    (void)hipEventRecord(ev.e[0], st);
  }
  if (!hip_ok(launch_trace(st, a), "k_trace (shadow pass)", err)) return false;
  if (profile_kernels_) {
    (void)hipEventRecord(ev.e[1], st);
    pending_events_.push_back(ev);
  }
  shadow_pending_ = false;
  return true;
}

// draw_frame (raytracer.rs:369-613): one path segment per pixel
bool Renderer::one_launch(Error& err) {
  if (request_new_frame_ && !reset_buffers(err)) return false;
  hipStream_t st = inst_->stream;
  LaunchArgs a;
  fill_args(a);
  FrameData& fd = a.frame;
  memset(&fd, 0, sizeof(fd));
  fd.seed = rng_.next();                  // rng.gen::<u32>(), raytracer.rs:487
  fd.lights_no = scene_->lights_no;
  sched_.next(fd.pixel_offset);           // WorkScheduler::next(), :489
  fd.scene_radius = scene_->data.meta.scene_radius;
  fd.exposure = exposure_;
  fd.scene_size[0] = (float)w_;
  fd.scene_size[1] = (float)h_;
  for (int k = 0; k < 3; ++k) fd.scene_centre[k] = scene_->data.meta.scene_centre[k];
  fd.camera_persp = camera_.type == GLZ_CAMERA_PERSPECTIVE ? 1u : 0u;
  fd.pt_steps = pt_steps_;
  fd.direct_only = integrator_ == GLZ_DIRECT ? 1u : 0u;
  ++launches_;
  if (fd.lights_no == 0) return true;   // the raygen shader returns before touching anything (path_trace.rgen:137-141)
  a.do_closest = 1;
  a.do_shadow = shadow_pending_ ? 1u : 0u;   // the previous launch's shadow rays ride in this launch's traversal kernel
  EventSet ev{};
  if (profile_kernels_) {
    if (!acquire_events(ev, err)) return false;
    ev.flush = false;
    (void)hipEventRecord(ev.e[0], st);
  }
  if (!hip_ok(launch_trace(st, a), "k_trace", err)) return false;
  if (profile_kernels_) (void)hipEventRecord(ev.e[1], st);
  if (!hip_ok(launch_shade(st, a), "k_shade", err)) return false;
  if (profile_kernels_) {
    (void)hipEventRecord(ev.e[2], st);
    pending_events_.push_back(ev);
  }
  shadow_pending_ = true;
  pending_set_ = a.shade_set;
  pending_exposure_ = exposure_;
  return true;
}

bool Renderer::get_stats(glz_render_stats* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (!flush_shadows(err)) return false;   // the counters and timings of the last launch's shadow rays belong to it
  if (!pending_events_.empty()) {
    if (!hip_ok(hipEventSynchronize(pending_events_.back().e[pending_events_.back().flush ? 1 : 2]), "hipEventSynchronize", err)) return false;
    for (auto& s : pending_events_) {
      float a = 0, b = 0;
      (void)hipEventElapsedTime(&a, s.e[0], s.e[1]);
      if (s.flush) {
        shadow_ms_ += a;
      } else {
        (void)hipEventElapsedTime(&b, s.e[1], s.e[2]);
        closest_ms_ += a;
        shade_ms_ += b;
      }
      render_ms_ += (double)a + b;
      free_events_.push_back(s);
    }
    pending_events_.clear();
  }
  memset(out, 0, sizeof(*out));
  out->launches = launches_;
  uint64_t owned = 0;
  for (uint32_t t = map_.rank; t < map_.tiles_x * map_.tiles_y; t += map_.world) {
    const uint32_t tx = t % map_.tiles_x, ty = t / map_.tiles_x;
    const uint32_t tw = std::min(kTile, w_ - tx * kTile), th = std::min(kTile, h_ - ty * kTile);
    owned += (uint64_t)tw * th;
  }
  out->samples = owned * launches_;
  out->render_ms = render_ms_;
  out->trace_closest_ms = closest_ms_;
  out->shade_ms = shade_ms_;
  out->trace_shadow_ms = shadow_ms_;
  TraceCounters c{};
  if (!hip_ok(hipMemcpyAsync(&c, counters_.ptr, sizeof(c), hipMemcpyDeviceToHost, inst_->stream), "read counters", err)) return false;
  if (!hip_ok(hipStreamSynchronize(inst_->stream), "read counters", err)) return false;
  out->closest_rays = c.closest_rays;
  out->shadow_rays = c.shadow_rays;
  out->closest_nodes = c.closest_nodes;
  out->closest_tris = c.closest_tris;
  out->shadow_nodes = c.shadow_nodes;
  out->shadow_tris = c.shadow_tris;
  out->hits = c.hits;
  out->fresh_paths = c.fresh;
  for (int i = 0; i < 12; ++i) out->phase[i] = c.phase[i];
  return true;
}

bool Renderer::set_integrator(int integrator, Error& err) {
  if (integrator != GLZ_DIRECT && integrator != GLZ_PATH_TRACE) {
    err.code = GLZ_E_ARG;
    err.msg = "unknown integrator";
    return false;
  }
  if (integrator != integrator_) {   // raytracer.rs:197
    integrator_ = integrator;
    request_new_frame_ = true;
  }
  return true;
}

bool Renderer::set_exposure(float e) {
  if (e >= 0.0f) exposure_ = e;   // raytracer.rs:186-193: no restart
  return true;
}

bool Renderer::update_camera(const glz_camera& c, Error& err) {
  if (c.type > GLZ_CAMERA_ORTHOGRAPHIC) {
    err.code = GLZ_E_ARG;
    err.msg = "unknown camera type";
    return false;
  }
  camera_ = c;
  host::push_constants(camera_, w_, h_, cam_.camera2world, cam_.screen2camera);
  request_new_frame_ = true;
  return true;
}

bool Renderer::change_resolution(uint32_t w, uint32_t h, Error& err) {
  if (w == 0 || h == 0) {
    err.code = GLZ_E_ARG;
    err.msg = "resolution must be non-zero";
    return false;
  }
  if (!wait_idle(err)) return false;
  w_ = w;
  h_ = h;
  if (!allocate(err)) return false;
  return update_camera(camera_, err);   // raytracer.rs:297
}

bool Renderer::change_scene(Scene* scene, Error& err) {
  if (!scene) {
    err.code = GLZ_E_ARG;
    err.msg = "scene is null";
    return false;
  }
  if (!wait_idle(err)) return false;
  scene_.reset(scene);
  exposure_ = scene->data.meta.exposure;
  if (!allocate(err)) return false;
  return update_camera(scene->data.camera, err);   // raytracer.rs:246-247
}

bool Renderer::update_materials_and_lights(const glz_material* m, uint32_t nm, const glz_light* l, uint32_t nl, const glz_texture* t, uint32_t nt,
                                           Error& err) {
  if (!wait_idle(err)) return false;
  const uint32_t od = scene_->stack_overflow_depth;
  if (!scene_->update_materials_and_lights(m, nm, l, nl, t, nt, err)) return false;
  if (scene_->stack_overflow_depth != od && !allocate(err)) return false;
  request_new_frame_ = true;   // raytracer.rs:325
  return true;
}

// raytracer.rs:328-356.  The reference rebuilds descriptors, pipeline and SBT and leaves the accumulation alone; here the
// kernels read the texture array through the scene struct of every launch, so re-uploading it is all there is to do.
bool Renderer::refresh_binded_textures(const glz_texture* t, uint32_t nt, Error& err) {
  if (!wait_idle(err)) return false;
  return scene_->refresh_textures(t, nt, err);
}

bool Renderer::wait_idle(Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (!flush_shadows(err)) return false;
  return hip_ok(hipStreamSynchronize(inst_->stream), "wait_idle", err);
}

bool Renderer::restart() {
  request_new_frame_ = true;
  return true;
}

bool Renderer::step(uint32_t n, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  for (uint32_t i = 0; i < n; ++i)
    if (!one_launch(err)) return false;
  return true;
}

// draw (raytracer.rs:615-687)
bool Renderer::draw(size_t spp, void (*cb)(void*), void* user, uint8_t* rgba8_out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  request_new_frame_ = true;
  const size_t steps = steps_per_sample();
  const size_t substep = spp * steps;
  for (size_t i = 0; i < substep; ++i) {
    if (!one_launch(err)) return false;
    if (cb && i % steps == 0) cb(user);   // raytracer.rs:651-653
  }
  if (spp == 0 && request_new_frame_ && !reset_buffers(err)) return false;
  if (!wait_idle(err)) return false;
  if (rgba8_out) return read_rgba8(rgba8_out, err);
  return true;
}

bool Renderer::read_frame(bool result, float* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (request_new_frame_ && !reset_buffers(err)) return false;
  if (!flush_shadows(err)) return false;
  hipStream_t st = inst_->stream;
  if (!hip_ok(launch_export(st, map_, result ? result_.ptr : cumulative_.ptr, frame_tmp_.ptr, true), "k_export", err)) return false;
  if (!hip_ok(hipMemcpyAsync(out, frame_tmp_.ptr, sizeof(float4) * (size_t)w_ * h_, hipMemcpyDeviceToHost, st), "read frame", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "read frame", err);
}

// blit out32 -> out8 (R8G8B8A8_SRGB) + export (raytracer.rs:576-584, memory.rs:269-483)
bool Renderer::read_rgba8(uint8_t* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (request_new_frame_ && !reset_buffers(err)) return false;
  if (!flush_shadows(err)) return false;
  hipStream_t st = inst_->stream;
  if (!hip_ok(launch_export(st, map_, result_.ptr, frame_tmp_.ptr, true), "k_export", err)) return false;
  if (!hip_ok(launch_tonemap(st, w_ * h_, frame_tmp_.ptr, rgba8_.ptr), "k_tonemap", err)) return false;
  if (!hip_ok(hipMemcpyAsync(out, rgba8_.ptr, (size_t)w_ * h_ * 4, hipMemcpyDeviceToHost, st), "read rgba8", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "read rgba8", err);
}

bool Renderer::set_seed(uint64_t s) {
  seed_ = s;
  request_new_frame_ = true;
  return true;
}

bool Renderer::set_depth(uint32_t d, Error& err) {
  if (d == 0 || d > 1024) {
    err.code = GLZ_E_ARG;
    err.msg = "depth (PT_STEPS) must be in 1..1024";
    return false;
  }
  pt_steps_ = d;
  request_new_frame_ = true;
  return true;
}

bool Renderer::set_partition(uint32_t rank, uint32_t world, Error& err) {
  if (world == 0 || rank >= world) {
    err.code = GLZ_E_ARG;
    err.msg = "bad tile partition";
    return false;
  }
  if (!wait_idle(err)) return false;
  map_.rank = rank;
  map_.world = world;
  return allocate(err);
}

bool Renderer::export_device(int which, void* dev, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (request_new_frame_ && !reset_buffers(err)) return false;
  if (!flush_shadows(err)) return false;
  if (!hip_ok(launch_export(inst_->stream, map_, which ? result_.ptr : cumulative_.ptr, static_cast<float4*>(dev), true), "k_export", err)) return false;
  return hip_ok(hipStreamSynchronize(inst_->stream), "export_device", err);
}

bool Renderer::tonemap_device(const void* dev_result, uint8_t* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  hipStream_t st = inst_->stream;
  if (!hip_ok(launch_tonemap(st, w_ * h_, static_cast<const float4*>(dev_result), rgba8_.ptr), "k_tonemap", err)) return false;
  if (!hip_ok(hipMemcpyAsync(out, rgba8_.ptr, (size_t)w_ * h_ * 4, hipMemcpyDeviceToHost, st), "read rgba8", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "tonemap_device", err);
}

bool Renderer::launch_constants(uint32_t launch, uint32_t* seed, float off[2]) {
  host::SeedStream rng(seed_);
  host::WorkScheduler ws;
  uint32_t s = 0;
  float o[2] = {0, 0};
  for (uint32_t i = 0; i <= launch; ++i) {
    s = rng.next();
    ws.next(o);
  }
  *seed = s;
  off[0] = o[0];
  off[1] = o[1];
  return true;
}

void Renderer::push_constants(float out[32]) const {
  memcpy(out, cam_.camera2world, 64);
  memcpy(out + 16, cam_.screen2camera, 64);
}

}  // namespace glz
