// RayTraceRenderer on HIP -- see renderer.h.  Reference: lib/src/vulkan/raytracer.rs.
#include "renderer.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

#include "rccl_dl.h"

namespace glz {

namespace {
// One host thread per additional GPU: it enqueues that device's launches while the caller's thread enqueues its own (two
// kernels per launch and device; at 1/8 of a 1080p frame per GPU a launch lasts ~0.17 ms, so eight devices fed from one
// thread would be bound by the host).  One task at a time: post(), then wait().
class Worker {
 public:
  Worker() : th_([this] { loop(); }) {}
  ~Worker() { stop(); }
  void post(std::function<void()> f) {
    std::lock_guard<std::mutex> l(m_);
    task_ = std::move(f);
    has_ = true;
    done_ = false;
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> l(m_);
    cv_.wait(l, [&] { return done_; });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> l(m_);
      if (quit_) return;
      quit_ = true;
      cv_.notify_all();
    }
    th_.join();
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> l(m_);
    for (;;) {
      cv_.wait(l, [&] { return has_ || quit_; });
      if (!has_) return;
      std::function<void()> f = std::move(task_);
      has_ = false;
      l.unlock();
      f();
      l.lock();
      done_ = true;
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::function<void()> task_;
  bool has_ = false, done_ = true, quit_ = false;
  std::thread th_;   // last member: the thread starts once everything above exists
};
}  // namespace

// Another GPU of this process: its own instance (device + stream), a replica of the scene, a renderer for the tiles
// t % n == rank, the frame it contributes to the reduce, and the host thread that drives it.
struct Renderer::Peer {
  std::unique_ptr<Instance> inst;
  std::unique_ptr<Renderer> r;
  DeviceBuffer<float4> frame;
  hipEvent_t sent = nullptr;   // peer-copy exchange: this device's tiles have left (recorded on its stream, awaited by device 0's)
  Worker worker;
  ~Peer() {
    worker.stop();
    if (inst) (void)hipSetDevice(inst->device);
    if (sent) (void)hipEventDestroy(sent);
    r.reset();
    frame.release();
  }
};
// What the peers' tasks write their outcome into.  The tasks hold pointers into it, so it must not go away while one of them
// runs: if the poster leaves early (an exception out of its local share of the work), the destructor waits for the workers.
struct Renderer::Pending {
  std::vector<Error> errs;
  std::vector<char> ok;
  Renderer* posted_on = nullptr;
  Pending() = default;
  Pending(const Pending&) = delete;
  Pending& operator=(const Pending&) = delete;
  ~Pending() {
    if (posted_on)
      for (auto& peer : posted_on->peers_) peer->worker.wait();
  }
};

// f(Peer&, Error&) -> bool on every peer's thread; f is copied into the tasks, whatever it refers to must outlive join_all()
template <class F>
void Renderer::post_all(F f, Pending& p) {
  p.errs.assign(peers_.size(), Error());
  p.ok.assign(peers_.size(), 1);
  p.posted_on = this;
  for (size_t i = 0; i < peers_.size(); ++i) {
    Peer* peer = peers_[i].get();
    Error* e = &p.errs[i];
    char* ok = &p.ok[i];
    peer->worker.post([=] {
      try {
        *ok = f(*peer, *e) ? 1 : 0;
      } catch (const std::exception& ex) {
        e->code = GLZ_E_IO;
        e->msg = ex.what();
        *ok = 0;
      } catch (...) {   // nothing may leave a peer's thread: that would be std::terminate
        e->code = GLZ_E_IO;
        e->msg = "unknown exception on a device thread";
        *ok = 0;
      }
    });
  }
}
bool Renderer::join_all(Pending& p, Error& err) {
  for (auto& peer : peers_) peer->worker.wait();
  p.posted_on = nullptr;
  for (size_t i = 0; i < p.ok.size(); ++i)
    if (!p.ok[i]) {
      err = p.errs[i];
      err.msg = "device " + std::to_string(peers_[i]->inst->device) + ": " + err.msg;
      return false;
    }
  return true;
}
template <class F>
bool Renderer::forward(F f, Error& err) {
  if (peers_.empty()) return true;
  Pending p;
  post_all(f, p);
  return join_all(p, err);
}

namespace {
constexpr uint32_t kTile = 64;
// Kernel boundaries are timed with HIP events on one launch of every `stride` (Renderer::event_stride), at a pseudo-random place
// inside each group of `stride` consecutive launches, and counted `stride` times.  A fixed place would beat against the path depth:
// launch i traces bounce i mod depth of most pixels and the bounces differ in cost.
inline bool timed_launch(uint64_t i /* 1-based */, uint64_t stride) {
  const uint64_t group = (i - 1) / stride;
  uint32_t x = (uint32_t)group * 747796405u + 2891336453u;   // PCG-RXS-M-XS-32
  x = ((x >> ((x >> 28) + 4u)) ^ x) * 277803737u;
  x ^= x >> 22;
  return (i - 1) % stride == x % stride;
}
}
// Three event records around a timed launch cost 5 - 7 us in every chain: 1 % of a full 1080p launch, 5 % of a 1/8 share's (0.130 against
// 0.123 ms per launch with one launch in four timed, tools/timeline_small_share.sh) -- a small share is timed one launch in sixteen.
uint64_t Renderer::event_stride() const {
  uint64_t pixels = 0;
  for (const auto& c : chains_) pixels += c->map.n_local_pixels;
  return pixels >= (1u << 20) ? 4u : 16u;
}

Renderer* Renderer::create(Instance* inst, std::shared_ptr<Scene> scene, uint32_t w, uint32_t h, Error& err) {
  std::unique_ptr<Renderer> r(new Renderer());
  r->inst_ = inst;
  if (!hip_ok(hipSetDevice(inst->device), "hipSetDevice", err)) return nullptr;
  if (!scene) {
    // RayTraceRenderer::new(.., None, ..) renders an empty scene (raytracer.rs:170-174, NoScene)
    SceneData empty;
    empty.camera = default_camera();
    empty.meta = default_meta();
    scene.reset(Scene::create(inst, std::move(empty), err));
    if (!scene) return nullptr;
  }
  r->scene_ = scene;
  if (w == 0 || h == 0) {
    err.code = GLZ_E_ARG;
    err.msg = "resolution must be non-zero";
    return nullptr;
  }
  r->w_ = w;
  r->h_ = h;
  r->camera_ = scene->data.camera;
  r->exposure_ = scene->data.meta.exposure;
  host::push_constants(r->camera_, w, h, r->cam_.camera2world, r->cam_.screen2camera);
  if (!r->allocate(err)) return nullptr;
  return r.release();
}

void Renderer::release_chains() {
  for (auto& c : chains_) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->pending_events)
      for (auto& e : s.e) (void)hipEventDestroy(e);
    for (auto& s : c->free_events)
      for (auto& e : s.e) (void)hipEventDestroy(e);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  }
  chains_.clear();
}

void Renderer::release_peers() {
  if (!comms_.empty()) {
    std::string why;
    if (const Rccl* nc = Rccl::get(why))
      for (void* c : comms_)
        if (c) (void)nc->CommDestroy(static_cast<ncclComm_t>(c));
    comms_.clear();
  }
  peers_.clear();
  loopback_ = false;
  if (inst_) (void)hipSetDevice(inst_->device);
}

Renderer::~Renderer() {
  release_peers();
  if (inst_) (void)hipSetDevice(inst_->device);
  release_chains();
}

// Number of concurrent chains, from the pixels this rank owns (measured on the atrium, ms per launch of one rank's share of
// a 1080p frame with 1 / 2 / 3 chains): 2.07 M pixels 1.29 / 1.36 / 1.42, 1.04 M 0.72 / 0.72 / 0.71, 518 k 0.43 / 0.38 / 0.38,
// 259 k 0.26 / 0.25 / 0.23.  A launch over a million pixels is throughput bound and wants one chain; below that it is bound
// by the latency of its longest rays and concurrent chains fill the machine.  Four chains are slower again: HIP maps streams
// onto GPU_MAX_HW_QUEUES (4) hardware queues and the fourth chain shares one (0.154 -> 0.252 ms for a 1/8 share; with 8 queues
// 0.19 ms, tools/gpu_chain_sweep.py) -- and no number of chains goes below one chain's own step, which lasts as long as its slowest wave.
uint32_t Renderer::chains_for(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t wanted) {
  const uint32_t tiles_x = (w + kTile - 1) / kTile, tiles_y = (h + kTile - 1) / kTile, tiles = tiles_x * tiles_y;
  const uint32_t local_tiles = tiles > rank ? (tiles - rank + world - 1) / world : 0;
  const uint64_t pixels = (uint64_t)local_tiles * kTile * kTile;
  uint32_t want = wanted;
  if (want == 0) want = pixels >= 1000000u ? 1u : (pixels >= 400000u ? 2u : 3u);
  if (want > local_tiles) want = local_tiles;
  return want ? want : 1u;
}
uint32_t Renderer::pick_chains() const { return path_mode_ ? 1u : chains_for(w_, h_, rank_, world_, chains_wanted_); }

// Automatic launch mode: a device runs its launches as k_path batches only when every 64-pixel group it owns gets a resident wave of its
// own (4 096 on an MI355X at k_path's four waves per SIMD: up to 262 144 pixels) -- with more groups than waves some waves carry two
// groups one after the other and the launch loop loses to the two-kernel mode (a 1/6 share: 0.217 against 0.156 ms per launch) ...
bool Renderer::allocate(Error& err) {
  release_chains();
  {
    const uint32_t tiles_x = (w_ + kTile - 1) / kTile, tiles_y = (h_ + kTile - 1) / kTile, tiles = tiles_x * tiles_y;
    const uint64_t pixels = (uint64_t)(tiles > rank_ ? (tiles - rank_ + world_ - 1) / world_ : 0) * kTile * kTile;
    bool fits = false;
    if (launch_mode_ == 0 && pixels > 0 && pixels <= (1u << 22) && scene_->dev.two_level == 0) {
      const uint32_t blocks = (uint32_t)((pixels / 64 + kTraceBlock / 64 - 1) / (kTraceBlock / 64));
      const uint32_t resident = path_resident_blocks(scene_->dev);
      // ... and the launch loop is what pays: where launches are short (a scene of a few thousand triangles: the 512 x 512 cube runs
      // 34 % faster in it, its kernel boundaries were most of a launch) or the chip is less than four fifths full (a 1/16 share of the
      // 1080p atrium: 0.096 against 0.098 ms per launch).  A share that fills every wave slot with a scene of its size is faster as two
      // kernels since round 4 (1080p / 8: 0.129 against 0.134 ms; tools/gpu_partition_timing.py, tools/gpu_batch_length.py).
      fits = resident >= blocks && (scene_->info.n_world_triangles < 4096 || (uint64_t)blocks * 5u <= (uint64_t)resident * 4u);
    }
    path_mode_ = scene_->dev.two_level == 0 && (launch_mode_ == 2 || fits);
    // The two-kernel mode's traversal walks the hierarchy's 8-wide nodes only on request (set_node_width(8), GLAZE_NODE_WIDTH=8): built to
    // shorten a small tile share's chain of dependent node fetches (17.3 against 24.9 visits per sample), measured slower at every share --
    // a 1080p / 8 share 0.140 - 0.145 against 0.1285 ms per launch, / 16 0.098 against 0.097, the full frame 0.93 against 0.79
    // (profiles/r05_wide_nodes.txt): twice the boxes and up to seven conditional pushes make a visit 1.9 x the instructions, and one wave
    // issues them one after the other, so the shorter chain takes as long.
    wide8_ = scene_->dev.two_level == 0 && scene_->dev.bvh_nodes8 != nullptr && node_width_ == 8;
  }
  const uint32_t S = pick_chains();
  const uint32_t od = scene_->stack_overflow_depth;
  for (uint32_t s = 0; s < S; ++s) {
    std::unique_ptr<Chain> c(new Chain());
    if (s == 0) {
      c->stream = inst_->stream;   // glz_instance_stream keeps naming a stream the renderer works on
    } else {
      if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate", err)) return false;
      c->own_stream = true;
    }
    TileMap& m = c->map;
    m.width = w_;
    m.height = h_;
    m.tiles_x = (w_ + kTile - 1) / kTile;
    m.tiles_y = (h_ + kTile - 1) / kTile;
    m.rank = rank_ + s * world_;   // chain s of S = the finer partition (rank + s * world, world * S)
    m.world = world_ * S;
    const uint32_t tiles = m.tiles_x * m.tiles_y;
    m.n_local_tiles = tiles > m.rank ? (tiles - m.rank + m.world - 1) / m.world : 0;
    m.n_local_pixels = m.n_local_tiles * kTile * kTile;
    const size_t n = m.n_local_pixels;
    // shadow-ray queue: 8 shards of ceil(blocks/8)*256 entries (kernels_render.hip, queue_capacity)
    const size_t n_queue = (((n + 255) / 256 + 7) / 8) * 256 * 8;
    DeviceBuffer<float4>* bufs[] = {&c->ray_o, &c->ray_d, &c->imp[0], &c->imp[1], &c->imp[2], &c->imp[3], &c->hit, &c->sh_o, &c->sh_d, &c->contrib,
                                    &c->cumulative, &c->result};
    for (auto* b : bufs)
      if (!hip_ok(b->alloc((b == &c->sh_o || b == &c->sh_d || b == &c->contrib) ? n_queue : n), "alloc path state", err)) return false;
    if (!hip_ok(c->cone.alloc(n), "alloc path state", err)) return false;
    if (!hip_ok(c->hit_inst.alloc(n), "alloc path state", err)) return false;
    c->grid = trace_grid_blocks(m.n_local_pixels, false, scene_->dev.two_level != 0);
    c->grid_counting = trace_grid_blocks(m.n_local_pixels, true, scene_->dev.two_level != 0);
    c->grid8 = wide8_ ? trace_grid_blocks(m.n_local_pixels, false, false, true) : 0u;
    c->grid_path = path_mode_ ? path_grid_blocks(m.n_local_pixels, scene_->dev) : 0u;
    // traversal spill: one slot of `od` entries per lane of the largest of the persistent grids
    if (!hip_ok(c->overflow.alloc((size_t)std::max(std::max(std::max(c->grid, c->grid_counting), c->grid_path), c->grid8) * kTraceBlock * od), "alloc traversal spill", err)) return false;
    if (!hip_ok(c->queue_count.alloc(2 * kQueueSetWords), "alloc queue counters", err)) return false;
    if (!hip_ok(c->path_cost.alloc(8 + n / 64 + 1), "alloc path costs", err)) return false;
    if (!hip_ok(hipMemsetAsync(c->path_cost.ptr, 0, sizeof(uint32_t) * (8 + n / 64 + 1), c->stream), "clear path costs", err)) return false;
    chains_.push_back(std::move(c));
  }
  if (!hip_ok(frame_tmp_.alloc((size_t)w_ * h_), "alloc frame", err)) return false;
  if (!hip_ok(rgba8_.alloc((size_t)w_ * h_), "alloc rgba8", err)) return false;
  if (!oetf_thresholds_.ptr) {
    float thr[256];
    host::srgb8_thresholds(thr);
    if (!hip_ok(oetf_thresholds_.upload(thr, 256, chains_[0]->stream), "upload OETF thresholds", err)) return false;
    if (!hip_ok(hipStreamSynchronize(chains_[0]->stream), "upload OETF thresholds", err)) return false;   // thr[] is on this stack frame
  }
  if (!hip_ok(counters_.alloc(1), "alloc counters", err)) return false;
  request_new_frame_ = true;
  return true;
}

// the fill_buffer / clear_color_image of a new frame (raytracer.rs:506-532) + scheduler rewind (:483-485)
bool Renderer::reset_buffers(Error& err) {
  // work of the abandoned frame may still be running on the chains' streams
  for (auto& c : chains_)
    if (!hip_ok(hipStreamSynchronize(c->stream), "reset", err)) return false;
  for (auto& cp : chains_) {
    Chain& c = *cp;
    const size_t bytes = sizeof(float4) * (size_t)c.map.n_local_pixels;
    DeviceBuffer<float4>* zero[] = {&c.ray_o, &c.ray_d, &c.imp[0], &c.imp[1], &c.imp[2], &c.imp[3], &c.cumulative, &c.result, &c.contrib};
    for (auto* b : zero)
      if (bytes && !hip_ok(hipMemsetAsync(b->ptr, 0, bytes, c.stream), "clear path state", err)) return false;
    if (!hip_ok(hipMemsetAsync(c.queue_count.ptr, 0, sizeof(uint32_t) * 2 * kQueueSetWords, c.stream), "clear queue counters", err)) return false;
    c.shadow_pending = false;   // queued shadow rays of the abandoned frame are dropped with it
    c.trace_ms = c.shade_ms = c.flush_ms = c.path_ms = 0;
    for (auto& s : c.pending_events) c.free_events.push_back(s);
    c.pending_events.clear();
  }
  if (!hip_ok(hipMemsetAsync(counters_.ptr, 0, sizeof(TraceCounters), chains_[0]->stream), "clear counters", err)) return false;
  if (chains_.size() > 1 && !hip_ok(hipStreamSynchronize(chains_[0]->stream), "clear counters", err)) return false;   // the other chains add to them too
  sched_.rewind();
  rng_.reseed(seed_);   // build-defined: a restart replays the same seed stream (the reference keeps drawing from entropy)
  launches_ = 0;
  request_new_frame_ = false;
  return true;
}

void Renderer::fill_args(const Chain& c, LaunchArgs& a) const {
  a.scene = scene_->dev;
  a.st.ray_o = c.ray_o.ptr;
  a.st.ray_d = c.ray_d.ptr;
  for (int q = 0; q < 4; ++q) a.st.imp[q] = c.imp[q].ptr;
  a.st.hit = c.hit.ptr;
  a.st.cone = c.cone.ptr;
  a.st.hit_inst = c.hit_inst.ptr;
  a.st.sh_o = c.sh_o.ptr;
  a.st.sh_d = c.sh_d.ptr;
  a.st.contrib = c.contrib.ptr;
  a.st.queue_count = c.queue_count.ptr;
  a.st.cumulative = c.cumulative.ptr;
  a.st.result = c.result.ptr;
  a.st.overflow = c.overflow.ptr;
  a.st.overflow_depth = scene_->stack_overflow_depth;
  a.st.path_cost = c.path_cost.ptr;
  a.map = c.map;
  a.cam = cam_;
  a.counters = counting_ ? counters_.ptr : nullptr;
  a.do_closest = a.do_shadow = 0;
  a.shade_set = c.pending_set ^ 1u;
  a.shadow_exposure = c.pending_exposure;
}

void Renderer::resolve_events(Chain& c) {
  for (auto& s : c.pending_events) {
    float a = 0, b = 0;
    (void)hipEventElapsedTime(&a, s.e[0], s.e[1]);
    if (s.kind == 1) {
      c.flush_ms += a;
    } else if (s.kind == 2) {
      c.path_ms += a;
    } else {
      (void)hipEventElapsedTime(&b, s.e[1], s.e[2]);
      const double weight = (double)s.weight;
      c.trace_ms += a * weight;
      c.shade_ms += b * weight;
    }
    c.free_events.push_back(s);
  }
  c.pending_events.clear();
}

bool Renderer::acquire_events(Chain& c, EventSet& ev, Error& err) {
  if (c.free_events.empty()) {
    if (c.pending_events.size() >= 64) {
      // resolve and recycle the pending sets (without the flush get_stats would do)
      if (!hip_ok(hipEventSynchronize(c.pending_events.back().e[c.pending_events.back().kind ? 1 : 2]), "hipEventSynchronize", err)) return false;
      resolve_events(c);
    }
    if (c.free_events.empty()) {
      EventSet fresh{};
      for (auto& e : fresh.e)
        // timing only: without the system-scope fence an event's completion otherwise carries -- the cache write-back and invalidate
        // it costs the kernels that follow (the BVH leaves the L2s at every timed kernel boundary of every chain)
        if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableSystemFence), "hipEventCreate", err)) return false;
      c.free_events.push_back(fresh);
    }
  }
  ev = c.free_events.back();
  c.free_events.pop_back();
  return true;
}

// Stand-alone shadow pass for the rays the last launch queued: run before anything observes the accumulators.
bool Renderer::flush_shadows(Chain& c, Error& err) {
  if (!c.shadow_pending) return true;
  LaunchArgs a;
  fill_args(c, a);
  memset(&a.frame, 0, sizeof(a.frame));
  a.do_shadow = 1;
  EventSet ev{};
  if (profile_kernels_) {
    if (!acquire_events(c, ev, err)) return false;
    ev.kind = 1;
    (void)hipEventRecord(ev.e[0], c.stream);
  }
  if (!hip_ok(launch_trace(c.stream, a, counting_ ? c.grid_counting : (wide8() ? c.grid8 : c.grid), wide8()), "k_trace (shadow pass)", err)) return false;
  if (profile_kernels_) {
    (void)hipEventRecord(ev.e[1], c.stream);
    c.pending_events.push_back(ev);
  }
  c.shadow_pending = false;
  return true;
}

// what all launches of a frame share in RTFrameData (raytracer.rs:369-613); seed, pixel offset and exposure are per launch
bool Renderer::launch_constants_common(FrameData& fd, Error& err) {
  memset(&fd, 0, sizeof(fd));
  fd.lights_no = scene_->lights_no;
  fd.scene_radius = scene_->data.meta.scene_radius;
  fd.scene_size[0] = (float)w_;
  fd.scene_size[1] = (float)h_;
  for (int k = 0; k < 3; ++k) fd.scene_centre[k] = scene_->data.meta.scene_centre[k];
  fd.camera_persp = camera_.type == GLZ_CAMERA_PERSPECTIVE ? 1u : 0u;
  fd.pt_steps = pt_steps_;
  fd.direct_only = integrator_ == GLZ_DIRECT ? 1u : 0u;
  fd.lod_mode = (uint32_t)lod_mode_;
  if (lod_mode_ != 0) {
    // one pixel of the image plane at unit distance (perspective: the cone's spread) or in world units (orthographic: the
    // cone's constant width), from the projection's vertical scale: screen2camera[1][1] = -tan(fovy / 2) or -scale
    const float pixel = 2.0f * fabsf(cam_.screen2camera[5]) / (float)h_;
    const bool persp = camera_.type == GLZ_CAMERA_PERSPECTIVE;
    fd.cone_spread = persp ? pixel : 0.0f;
    fd.cone_width0 = persp ? 0.0f : pixel;
    if (!scene_->mips_ready() && !scene_->ensure_mips(err)) return false;
  }
  return true;
}

// draw_frame (raytracer.rs:369-613): one path segment per pixel
bool Renderer::one_launch(Error& err) {
  if (request_new_frame_ && !reset_buffers(err)) return false;
  FrameData fd;
  if (!launch_constants_common(fd, err)) return false;
  fd.seed = rng_.next();                  // rng.gen::<u32>(), raytracer.rs:487
  sched_.next(fd.pixel_offset);           // WorkScheduler::next(), :489
  // the launch after this one, for the paths that end in this one (shade_pixel): anything that could make the next launch differ from
  // what is assumed here -- a new camera, resolution, scene, integrator, a restart -- resets the path state before it runs (reset_buffers)
  sched_.peek(fd.next_pixel_offset);
  fd.pregen = fd.direct_only ? 0u : 1u;
  fd.exposure = exposure_;
  ++launches_;
  if (fd.lights_no == 0) return true;   // the raygen shader returns before touching anything (path_trace.rgen:137-141)
  for (auto& cp : chains_) {
    Chain& c = *cp;
    hipStream_t st = c.stream;
    LaunchArgs a;
    fill_args(c, a);
    a.frame = fd;
    a.do_closest = 1;
    a.do_shadow = c.shadow_pending ? 1u : 0u;   // the previous launch's shadow rays ride in this launch's traversal kernel
    const uint64_t stride = event_stride();
    const bool timed = profile_kernels_ && timed_launch(launches_, stride);
    EventSet ev{};
    if (timed) {
      if (!acquire_events(c, ev, err)) return false;
      ev.kind = 0;
      ev.weight = (uint32_t)stride;
      (void)hipEventRecord(ev.e[0], st);
    }
    if (!hip_ok(launch_trace(st, a, counting_ ? c.grid_counting : (wide8() ? c.grid8 : c.grid), wide8()), "k_trace", err)) return false;
    if (timed) (void)hipEventRecord(ev.e[1], st);
    if (!hip_ok(launch_shade(st, a), "k_shade", err)) return false;
    if (timed) {
      (void)hipEventRecord(ev.e[2], st);
      c.pending_events.push_back(ev);
    }
    c.shadow_pending = true;
    c.pending_set = a.shade_set;
    c.pending_exposure = exposure_;
  }
  return true;
}

// n <= kPathMaxLaunches launches of draw_frame in ONE kernel (k_path): the same per-launch constants, in the same order
bool Renderer::path_batch(uint32_t n, Error& err) {
  if (request_new_frame_ && !reset_buffers(err)) return false;
  FrameData fd;
  if (!launch_constants_common(fd, err)) return false;
  PathBatch b;
  memset(&b, 0, sizeof(b));
  b.n = n;
  b.parity = chains_[0]->path_batches++ & 1u;
  for (uint32_t i = 0; i < n; ++i) {
    b.seed[i] = rng_.next();           // rng.gen::<u32>(), raytracer.rs:487
    sched_.next(b.offset[i]);          // WorkScheduler::next(), :489
    b.exposure[i] = exposure_;
  }
  sched_.peek(b.offset[n]);            // the launch after the batch (FrameData::next_pixel_offset of its last launch)
  fd.pregen = fd.direct_only ? 0u : 1u;
  launches_ += n;
  if (fd.lights_no == 0) return true;   // the raygen shader returns before touching anything (path_trace.rgen:137-141)
  Chain& c = *chains_[0];
  if (!flush_shadows(c, err)) return false;   // left by launches that ran as two kernels (work counters had been on)
  LaunchArgs a;
  fill_args(c, a);
  a.frame = fd;
  a.counters = nullptr;
  EventSet ev{};
  if (profile_kernels_) {
    if (!acquire_events(c, ev, err)) return false;
    ev.kind = 2;
    (void)hipEventRecord(ev.e[0], c.stream);
  }
  // the cost accumulator this batch adds to starts empty (the kernel reads the other one, which the batch before filled)
  if (!hip_ok(hipMemsetAsync(c.path_cost.ptr + 4u * b.parity, 0, 16, c.stream), "clear path cost accumulator", err)) return false;
  if (!hip_ok(launch_path(c.stream, a, b, c.grid_path), "k_path", err)) return false;
  if (profile_kernels_) {
    (void)hipEventRecord(ev.e[1], c.stream);
    c.pending_events.push_back(ev);
  }
  return true;   // nothing is pending: the kernel ends with the shadow rays of its last launch
}

bool Renderer::run_launches(uint32_t n, Error& err) {
  while (n != 0) {
    if (use_path()) {
      const uint32_t m = std::min(n, kPathMaxLaunches);
      if (!path_batch(m, err)) return false;
      n -= m;
    } else {
      if (!one_launch(err)) return false;
      --n;
    }
  }
  return true;
}

bool Renderer::set_launch_mode(int mode, Error& err) {
  if (mode < 0 || mode > 2) {
    err.code = GLZ_E_ARG;
    err.msg = "launch mode must be 0 (automatic), 1 (two kernels per launch) or 2 (per-wave launch loop)";
    return false;
  }
  if (!wait_idle(err)) return false;
  launch_mode_ = mode;
  if (!allocate(err)) return false;
  const bool ok = forward([=](Peer& p, Error& e) { return p.r->set_launch_mode(mode, e); }, err);
  (void)hipSetDevice(inst_->device);
  return ok;
}

bool Renderer::set_node_width(int width, Error& err) {
  if (width != 0 && width != 4 && width != 8) {
    err.code = GLZ_E_ARG;
    err.msg = "node width must be 0 (automatic), 4 or 8";
    return false;
  }
  if (!wait_idle(err)) return false;
  node_width_ = width;
  if (!allocate(err)) return false;
  const bool ok = forward([=](Peer& p, Error& e) { return p.r->set_node_width(width, e); }, err);
  (void)hipSetDevice(inst_->device);
  return ok;
}

bool Renderer::get_stats(glz_render_stats* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  double trace_ms = 0, shade_ms = 0, flush_ms = 0, path_ms = 0;
  for (auto& cp : chains_) {
    Chain& c = *cp;
    if (!flush_shadows(c, err)) return false;   // the counters and timings of the last launch's shadow rays belong to it
    if (!c.pending_events.empty()) {
      if (!hip_ok(hipEventSynchronize(c.pending_events.back().e[c.pending_events.back().kind ? 1 : 2]), "hipEventSynchronize", err)) return false;
      resolve_events(c);
    }
    trace_ms += c.trace_ms;
    shade_ms += c.shade_ms;
    flush_ms += c.flush_ms;
    path_ms += c.path_ms;
  }
  // concurrent chains overlap in time: the mean over chains is the time the rank spent in S concurrent instances of a kernel
  const double inv = chains_.empty() ? 0.0 : 1.0 / (double)chains_.size();
  memset(out, 0, sizeof(*out));
  out->launches = launches_;
  uint64_t owned = 0;
  const uint32_t tiles_x = (w_ + kTile - 1) / kTile, tiles_y = (h_ + kTile - 1) / kTile;
  for (uint32_t t = rank_; t < tiles_x * tiles_y; t += world_) {
    const uint32_t tx = t % tiles_x, ty = t / tiles_x;
    const uint32_t tw = std::min(kTile, w_ - tx * kTile), th = std::min(kTile, h_ - ty * kTile);
    owned += (uint64_t)tw * th;
  }
  out->samples = owned * launches_;
  out->trace_closest_ms = trace_ms * inv;
  out->shade_ms = shade_ms * inv;
  out->trace_shadow_ms = flush_ms * inv;
  out->other_ms = path_ms * inv;   // k_path: the per-wave launch loop of a small tile share (all three phases in one kernel)
  out->render_ms = out->trace_closest_ms + out->shade_ms + out->trace_shadow_ms + out->other_ms;
  for (auto& c : chains_)
    if (!hip_ok(hipStreamSynchronize(c->stream), "read counters", err)) return false;
  TraceCounters c{};
  if (!hip_ok(hipMemcpy(&c, counters_.ptr, sizeof(c), hipMemcpyDeviceToHost), "read counters", err)) return false;
  out->closest_rays = c.closest_rays;
  out->shadow_rays = c.shadow_rays;
  out->closest_nodes = c.closest_nodes;
  out->closest_tris = c.closest_tris;
  out->shadow_nodes = c.shadow_nodes;
  out->shadow_tris = c.shadow_tris;
  out->hits = c.hits;
  out->fresh_paths = c.fresh;
  for (int i = 0; i < 12; ++i) out->phase[i] = c.phase[i];
  out->tex_fetches = c.shade_tex[0] + c.trace_tex[0];
  out->tex_bytes = c.shade_tex[1] + c.trace_tex[1];
  out->alpha_tex_bytes = c.trace_tex[1];
  out->light_samples = c.shade_tex[2];
  out->sky_samples = c.shade_tex[3];
  // other GPUs of this process: work counters add up, kernel times overlap (the slowest device is what the job waits for)
  if (!peers_.empty()) {
    std::vector<glz_render_stats> ps(peers_.size());
    glz_render_stats* base = ps.data();
    Peer* first = peers_[0].get();
    (void)first;
    Pending pend;
    std::vector<Peer*> order;
    for (auto& p : peers_) order.push_back(p.get());
    const std::vector<Peer*>* ord = &order;
    post_all([=](Peer& p, Error& e) {
      size_t i = 0;
      while ((*ord)[i] != &p) ++i;
      return p.r->get_stats(base + i, e);
    }, pend);
    if (!join_all(pend, err)) return false;
    for (const glz_render_stats& q : ps) {
      out->samples += q.samples;
      out->trace_closest_ms = std::max(out->trace_closest_ms, q.trace_closest_ms);
      out->shade_ms = std::max(out->shade_ms, q.shade_ms);
      out->trace_shadow_ms = std::max(out->trace_shadow_ms, q.trace_shadow_ms);
      out->other_ms = std::max(out->other_ms, q.other_ms);
      out->closest_rays += q.closest_rays; out->shadow_rays += q.shadow_rays;
      out->closest_nodes += q.closest_nodes; out->closest_tris += q.closest_tris;
      out->shadow_nodes += q.shadow_nodes; out->shadow_tris += q.shadow_tris;
      out->hits += q.hits; out->fresh_paths += q.fresh_paths;
      for (int i = 0; i < 12; ++i) out->phase[i] += q.phase[i];
      out->tex_fetches += q.tex_fetches; out->tex_bytes += q.tex_bytes; out->alpha_tex_bytes += q.alpha_tex_bytes;
      out->light_samples += q.light_samples; out->sky_samples += q.sky_samples;
    }
    out->render_ms = out->trace_closest_ms + out->shade_ms + out->trace_shadow_ms + out->other_ms;
    (void)hipSetDevice(inst_->device);
  }
  return true;
}

void Renderer::enable_counters(int flags) {
  counting_ = (flags & 1) != 0;
  profile_kernels_ = (flags & 2) != 0;
  for (auto& p : peers_) p->r->enable_counters(flags);
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
  return forward([=](Peer& p, Error& e) { return p.r->set_integrator(integrator, e); }, err);
}

bool Renderer::set_exposure(float e) {
  if (e >= 0.0f) exposure_ = e;   // raytracer.rs:186-193: no restart
  for (auto& p : peers_) p->r->set_exposure(e);
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
  const glz_camera cam = c;
  return forward([=](Peer& p, Error& e) { return p.r->update_camera(cam, e); }, err);
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
  const bool want_frame = !loopback_ && exchange_ == kExchangeReduce;
  if (!forward([=](Peer& p, Error& e) {
        if (!p.r->change_resolution(w, h, e)) return false;
        return !want_frame || hip_ok(p.frame.alloc((size_t)w * h), "alloc peer frame", e);
      }, err))
    return false;
  (void)hipSetDevice(inst_->device);
  return update_camera(camera_, err);   // raytracer.rs:297
}

bool Renderer::change_scene(std::shared_ptr<Scene> scene, Error& err) {
  if (!scene) {
    err.code = GLZ_E_ARG;
    err.msg = "scene is null";
    return false;
  }
  if (!wait_idle(err)) return false;
  scene_ = scene;
  exposure_ = scene->data.meta.exposure;
  if (!allocate(err)) return false;
  // the other GPUs of this process get replicas of the new scene, in the shape (flattened / two levels) this device built
  const Scene* src = scene.get();
  const Instance* src_inst = scene->instance ? scene->instance : inst_;
  if (!forward([=](Peer& p, Error& e) {
        p.inst->copy_build_options(*src_inst);
        if (src->info.as_levels) p.inst->as_levels = (int)src->info.as_levels;
        SceneData copy = src->data;
        std::shared_ptr<Scene> replica(Scene::create(p.inst.get(), std::move(copy), e));
        return replica && p.r->change_scene(replica, e);
      }, err))
    return false;
  (void)hipSetDevice(inst_->device);
  return update_camera(scene->data.camera, err);   // raytracer.rs:246-247
}

bool Renderer::update_materials_and_lights(const glz_material* m, uint32_t nm, const glz_light* l, uint32_t nl, const glz_texture* t, uint32_t nt,
                                           Error& err) {
  if (!wait_idle(err)) return false;
  const uint32_t od = scene_->stack_overflow_depth;
  if (!scene_->update_materials_and_lights(m, nm, l, nl, t, nt, err)) return false;
  if (scene_->stack_overflow_depth != od && !allocate(err)) return false;
  request_new_frame_ = true;   // raytracer.rs:325
  return forward([=](Peer& p, Error& e) { return p.r->update_materials_and_lights(m, nm, l, nl, t, nt, e); }, err);
}

// raytracer.rs:328-356.  The reference rebuilds descriptors, pipeline and SBT and leaves the accumulation alone; here the
// kernels read the texture array through the scene struct of every launch, so re-uploading it is all there is to do.
bool Renderer::refresh_binded_textures(const glz_texture* t, uint32_t nt, Error& err) {
  if (!wait_idle(err)) return false;
  if (!scene_->refresh_textures(t, nt, err)) return false;
  return forward([=](Peer& p, Error& e) { return p.r->refresh_binded_textures(t, nt, e); }, err);
}

bool Renderer::wait_idle(Error& err) {
  Pending pend;
  if (!peers_.empty()) post_all([](Peer& p, Error& e) { return p.r->wait_idle(e); }, pend);
  bool ok = hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err);
  for (auto& c : chains_)
    if (ok && !flush_shadows(*c, err)) ok = false;
  for (auto& c : chains_)
    if (ok && !hip_ok(hipStreamSynchronize(c->stream), "wait_idle", err)) ok = false;
  if (!peers_.empty()) {
    Error pe;
    if (!join_all(pend, pe) && ok) {
      err = pe;
      ok = false;
    }
  }
  return ok;
}

bool Renderer::restart() {
  request_new_frame_ = true;
  for (auto& p : peers_) p->r->restart();
  return true;
}

bool Renderer::step_local(uint32_t n, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  return run_launches(n, err);
}

// Every device enqueues the same n launches (same seed stream, same jitter sequence) for its own tiles; the peers' host
// threads work while this thread enqueues the local share.
bool Renderer::step(uint32_t n, Error& err) {
  if (peers_.empty()) return step_local(n, err);
  Pending pend;
  post_all([=](Peer& p, Error& e) { return p.r->step_local(n, e); }, pend);
  bool ok = step_local(n, err);
  Error pe;
  if (!join_all(pend, pe) && ok) {
    err = pe;
    ok = false;
  }
  return ok;
}

// draw (raytracer.rs:615-687)
bool Renderer::draw(size_t spp, void (*cb)(void*), void* user, uint8_t* rgba8_out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  restart();
  const size_t steps = steps_per_sample();
  const size_t substep = spp * steps;
  // The launches are enqueued in chunks -- one sample's worth, or as many as one k_path kernel takes (the longer its batch, the less
  // the kernel's slowest wave weighs) -- and the callback fires on this thread once per sample, when the sample's first launch has
  // been enqueued (raytracer.rs:651-653: `if i % steps == 0`); it never said anything about the GPU's progress.
  size_t done = 0, fired = 0;
  while (done < substep) {
    // any device in the per-wave launch loop wants long batches (a root over the residency limit must not hold its peers to one
    // sample per call); with a callback -- a progress bar, a cancel hook -- at most four samples go out between two calls of it
    bool any_path = use_path();
    for (const auto& p : peers_) any_path = any_path || p->r->use_path();
    size_t chunk = any_path ? (size_t)kPathMaxLaunches : steps;
    if (cb && chunk > 4 * steps) chunk = 4 * steps;
    const uint32_t m = (uint32_t)std::min(chunk, substep - done);
    if (!(peers_.empty() ? run_launches(m, err) : step(m, err))) return false;
    done += m;
    for (; fired < (done + steps - 1) / steps; ++fired)
      if (cb) cb(user);
  }
  if (spp == 0 && request_new_frame_ && !reset_buffers(err)) return false;
  if (!wait_idle(err)) return false;
  if (rgba8_out) return read_rgba8(rgba8_out, err);
  return true;
}

// every chain scatters its tiles into the full-frame buffer `dst` (the first one clears it); all on the first chain's
// stream after the chains have drained
bool Renderer::gather(bool result, float4* dst, Error& err, bool zero_first) {
  if (!settle(err)) return false;
  hipStream_t st = chains_[0]->stream;
  bool first = zero_first;
  for (auto& c : chains_) {
    if (!hip_ok(launch_export(st, c->map, result ? c->result.ptr : c->cumulative.ptr, dst, first), "k_export", err)) return false;
    first = false;
  }
  if (!peers_.empty() && !reduce_peers(result, dst, err)) return false;
  return true;
}

// The tiles of the other GPUs meet this device's in `dst` (RCCL over xGMI; one communicator per device, every call of one
// exchange inside ONE ncclGroup issued from this thread).  Two shapes:
//  * gather (default): every peer chain SENDS its packed tile-major buffer (its share of the frame, 1/n of the bytes) straight to
//    device 0, which receives into a staging area and scatters with k_export.  xGMI is point to point: the n - 1 transfers use
//    n - 1 different links at the same time, each carrying 1/n of the frame (4 MB of a 1080p frame at n = 8).
//  * reduce (GLAZE_MULTI_EXCHANGE=reduce): one ncclReduce(sum, float) of the zero-padded W*H*4 frame per device, in place on the
//    root -- what SURVEY 8(e) names first; a ring moves the whole frame over every link (33 MB at 1080p).
//  * peer (GLAZE_MULTI_EXCHANGE=peer): the gather shape without RCCL -- one hipMemcpyPeerAsync per peer on that peer's stream into
//    the same staging area, an event per peer that device 0's stream waits for.  For machines whose RCCL cannot be loaded or will not
//    initialise (bench.py falls back to it and says so); never chosen silently.
// The tiles are disjoint, so both give the image of a one-GPU render bit for bit.  Loop-back mode (every "device" is this one;
// tests): RCCL cannot put two ranks on one GPU, and there is nothing to move -- the peers scatter their tiles straight into `dst`.
bool Renderer::reduce_peers(bool result, float4* dst, Error& err) {
  hipStream_t st = chains_[0]->stream;
  if (!hip_ok(hipStreamSynchronize(st), "exchange: local frame", err)) return false;
  const bool lb = loopback_;
  const bool packed = !lb && exchange_ != kExchangeReduce;
  if (!forward([=](Peer& p, Error& e) {
        if (!hip_ok(hipSetDevice(p.inst->device), "hipSetDevice", e)) return false;
        if (packed) {
          // the chains' own tile-major buffers are what travels: one chain is sent from where it lies, several are laid end
          // to end first so that every peer issues exactly ONE send (copies and send are ordered by the peer's stream)
          if (!p.r->settle(e)) return false;
          auto& ch = p.r->chains_;
          if (ch.size() < 2) return true;
          size_t total = 0, off = 0;
          for (auto& c : ch) total += c->map.n_local_pixels;
          if (p.frame.count < total && !hip_ok(p.frame.alloc(total), "alloc peer staging", e)) return false;
          for (auto& c : ch) {
            const size_t n = c->map.n_local_pixels;
            if (n && !hip_ok(hipMemcpyAsync(p.frame.ptr + off, result ? c->result.ptr : c->cumulative.ptr, sizeof(float4) * n, hipMemcpyDeviceToDevice, p.inst->stream), "pack tiles", e))
              return false;
            off += n;
          }
          return true;
        }
        if (!p.r->gather(result, lb ? dst : p.frame.ptr, e, !lb)) return false;
        return hip_ok(hipStreamSynchronize(p.inst->stream), "exchange: peer frame", e);
      }, err))
    return false;
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (lb) return true;
  if (exchange_ == kExchangePeerCopy) {
    size_t total = 0;
    for (auto& p : peers_)
      for (auto& c : p->r->chains_) total += c->map.n_local_pixels;
    if (recv_stage_.count < total && !hip_ok(recv_stage_.alloc(total), "alloc exchange staging", err)) return false;
    size_t off = 0;
    bool ok = true;
    for (size_t i = 0; ok && i < peers_.size(); ++i) {
      Peer& p = *peers_[i];
      auto& ch = p.r->chains_;
      size_t n = 0;
      for (auto& c : ch) n += c->map.n_local_pixels;
      if (!n) continue;
      const float4* src = ch.size() > 1 ? p.frame.ptr : (result ? ch[0]->result.ptr : ch[0]->cumulative.ptr);
      ok = hip_ok(hipSetDevice(p.inst->device), "hipSetDevice", err) && (p.sent || hip_ok(hipEventCreateWithFlags(&p.sent, hipEventDisableTiming), "event", err)) &&
           hip_ok(hipMemcpyPeerAsync(recv_stage_.ptr + off, inst_->device, src, p.inst->device, sizeof(float4) * n, p.inst->stream), "peer copy", err) &&
           hip_ok(hipEventRecord(p.sent, p.inst->stream), "peer copy", err);
      (void)hipSetDevice(inst_->device);
      ok = ok && hip_ok(hipStreamWaitEvent(st, p.sent, 0), "peer copy", err);
      off += n;
    }
    if (!ok) return false;
    off = 0;
    for (auto& p : peers_)
      for (auto& c : p->r->chains_) {
        if (!hip_ok(launch_export(st, c->map, recv_stage_.ptr + off, dst, false), "k_export (received tiles)", err)) return false;
        off += c->map.n_local_pixels;
      }
    return hip_ok(hipStreamSynchronize(st), "exchange (root)", err);   // the root waited for every peer's copy before it scattered
  }
  std::string why;
  const Rccl* nc = Rccl::get(why);
  if (!nc || comms_.size() != peers_.size() + 1) {
    err.code = GLZ_E_DEVICE;
    err.msg = nc ? "RCCL communicators are missing" : why;
    return false;
  }
  auto ck = [&](ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    if (err.code == 0 || err.msg.empty()) {
      err.code = GLZ_E_DEVICE;
      err.msg = std::string(what) + ": " + nc->GetErrorString(r);
    }
    return false;
  };
  bool ok = true;
  if (packed) {
    size_t total = 0;
    for (auto& p : peers_)
      for (auto& c : p->r->chains_) total += c->map.n_local_pixels;
    if (recv_stage_.count < total && !hip_ok(recv_stage_.alloc(total), "alloc exchange staging", err)) return false;
    if (!ck(nc->GroupStart(), "ncclGroupStart")) return false;
    size_t off = 0;
    for (size_t i = 0; ok && i < peers_.size(); ++i) {
      auto& ch = peers_[i]->r->chains_;
      size_t n = 0;
      for (auto& c : ch) n += c->map.n_local_pixels;
      if (!n) continue;
      const float4* src = ch.size() > 1 ? peers_[i]->frame.ptr : (result ? ch[0]->result.ptr : ch[0]->cumulative.ptr);
      ok = ck(nc->Send(src, n * 4, ncclFloat, 0, static_cast<ncclComm_t>(comms_[i + 1]), peers_[i]->inst->stream), "ncclSend") &&
           ck(nc->Recv(recv_stage_.ptr + off, n * 4, ncclFloat, (int)i + 1, static_cast<ncclComm_t>(comms_[0]), st), "ncclRecv");
      off += n;
    }
    const bool ended = ck(nc->GroupEnd(), "ncclGroupEnd");   // always closed, also after a failed call inside it
    if (!ok || !ended) return false;
    off = 0;
    for (auto& p : peers_)
      for (auto& c : p->r->chains_) {
        if (!hip_ok(launch_export(st, c->map, recv_stage_.ptr + off, dst, false), "k_export (received tiles)", err)) return false;
        off += c->map.n_local_pixels;
      }
  } else {
    const size_t count = (size_t)w_ * h_ * 4;
    if (!ck(nc->GroupStart(), "ncclGroupStart")) return false;
    ok = ck(nc->Reduce(dst, dst, count, ncclFloat, ncclSum, 0, static_cast<ncclComm_t>(comms_[0]), st), "ncclReduce");   // in place on the root
    for (size_t i = 0; ok && i < peers_.size(); ++i)
      ok = ck(nc->Reduce(peers_[i]->frame.ptr, nullptr, count, ncclFloat, ncclSum, 0, static_cast<ncclComm_t>(comms_[i + 1]), peers_[i]->inst->stream), "ncclReduce");
    const bool ended = ck(nc->GroupEnd(), "ncclGroupEnd");
    if (!ok || !ended) return false;
  }
  // the peers' streams first (their buffers are free again), the root's last: everything has arrived when it is idle
  for (auto& p : peers_) {
    if (!hip_ok(hipSetDevice(p->inst->device), "hipSetDevice", err)) {
      (void)hipSetDevice(inst_->device);
      return false;
    }
    if (!hip_ok(hipStreamSynchronize(p->inst->stream), "exchange (peer)", err)) {
      (void)hipSetDevice(inst_->device);
      return false;
    }
  }
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "exchange (root)", err);
}

// everything this renderer has enqueued is done and its images are final (what gather() establishes before it scatters)
bool Renderer::settle(Error& err) {
  if (request_new_frame_ && !reset_buffers(err)) return false;
  return wait_idle(err);
}

bool Renderer::read_frame(bool result, float* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (!gather(result, frame_tmp_.ptr, err)) return false;
  hipStream_t st = chains_[0]->stream;
  if (!hip_ok(hipMemcpyAsync(out, frame_tmp_.ptr, sizeof(float4) * (size_t)w_ * h_, hipMemcpyDeviceToHost, st), "read frame", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "read frame", err);
}

// blit out32 -> out8 (R8G8B8A8_SRGB) + export (raytracer.rs:576-584, memory.rs:269-483)
bool Renderer::read_rgba8(uint8_t* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (!gather(true, frame_tmp_.ptr, err)) return false;
  hipStream_t st = chains_[0]->stream;
  if (!hip_ok(launch_tonemap(st, w_ * h_, frame_tmp_.ptr, oetf_thresholds_.ptr, rgba8_.ptr), "k_tonemap", err)) return false;
  if (!hip_ok(hipMemcpyAsync(out, rgba8_.ptr, (size_t)w_ * h_ * 4, hipMemcpyDeviceToHost, st), "read rgba8", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "read rgba8", err);
}

bool Renderer::set_texture_lod(int mode, Error& err) {
  if (mode != 0 && mode != 1 && mode != 2) {
    err.code = GLZ_E_ARG;
    err.msg = "texture LOD mode must be 0 (level 0), 1 (ray cones) or 2 (ray cones, anisotropic footprint)";
    return false;
  }
  lod_mode_ = mode;
  request_new_frame_ = true;
  return forward([=](Peer& p, Error& e) { return p.r->set_texture_lod(mode, e); }, err);
}

bool Renderer::set_seed(uint64_t s) {
  seed_ = s;
  request_new_frame_ = true;
  for (auto& p : peers_) p->r->set_seed(s);
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
  for (auto& p : peers_)
    if (!p->r->set_depth(d, err)) return false;
  return true;
}

bool Renderer::set_partition(uint32_t rank, uint32_t world, Error& err) {
  if (!peers_.empty()) {
    err.code = GLZ_E_ARG;
    err.msg = "a renderer that spans several devices (set_devices) cannot also be one rank of a process partition";
    return false;
  }
  return set_partition_local(rank, world, err);
}

bool Renderer::set_partition_local(uint32_t rank, uint32_t world, Error& err) {
  if (world == 0 || rank >= world) {
    err.code = GLZ_E_ARG;
    err.msg = "bad tile partition";
    return false;
  }
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  for (auto& c : chains_) {
    if (!flush_shadows(*c, err)) return false;
    if (!hip_ok(hipStreamSynchronize(c->stream), "set_partition", err)) return false;
  }
  rank_ = rank;
  world_ = world;
  return allocate(err);
}

// what a freshly created peer renderer takes over from this one
bool Renderer::configure_peer(Renderer& p, Error& err) const {
  p.integrator_ = integrator_;
  p.lod_mode_ = lod_mode_;
  p.pt_steps_ = pt_steps_;
  p.seed_ = seed_;
  p.exposure_ = exposure_;
  p.camera_ = camera_;
  p.cam_ = cam_;
  p.chains_wanted_ = chains_wanted_;
  p.launch_mode_ = launch_mode_;
  p.node_width_ = node_width_;
  p.counting_ = counting_;
  p.profile_kernels_ = profile_kernels_;
  p.request_new_frame_ = true;
  (void)err;
  return true;
}

bool Renderer::set_devices(const int* devices, int n, Error& err) {
  auto bad = [&](const char* m) {
    err.code = GLZ_E_ARG;
    err.msg = m;
    return false;
  };
  if (!devices || n < 1 || n > 64) return bad("set_devices: between 1 and 64 devices");
  if (devices[0] != inst_->device) return bad("set_devices: the first device must be the renderer's own (glz_instance_device)");
  if (world_ != 1 && peers_.empty()) return bad("set_devices: this renderer is one rank of a process partition (set_partition)");
  bool all_same = true, any_same = false;
  for (int i = 0; i < n; ++i) {
    if (devices[i] != devices[0]) all_same = false;
    for (int j = 0; j < i; ++j) any_same |= devices[i] == devices[j];
  }
  // GLAZE_MULTI_LOOPBACK: the list may name ONE device n times (tests on a one-GPU box).  Any value but `rccl`: the tiles meet
  // without RCCL (which cannot put two ranks on one GPU).  `rccl`: the exchange still goes through the RCCL entry points -- for
  // a stand-in library named by GLAZE_RCCL_LIBRARY (tests/fake_rccl), so that the n >= 2 group construction itself runs.
  const char* lbenv = getenv("GLAZE_MULTI_LOOPBACK");
  const bool dup_ok = n > 1 && all_same && lbenv != nullptr;
  const bool loopback = dup_ok && strcmp(lbenv, "rccl") != 0 && strcmp(lbenv, "peer") != 0;   // `peer`: the peer-copy exchange with n "devices" on one GPU
  if (any_same && !dup_ok) return bad("set_devices: a device is listed twice (GLAZE_MULTI_LOOPBACK=1 allows n copies of ONE device, for tests)");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
  for (int i = 0; i < n; ++i)
    if (devices[i] < 0 || devices[i] >= count) return bad("set_devices: HIP device ordinal out of range");
  int exchange = kExchangeGather;
  if (const char* x = getenv("GLAZE_MULTI_EXCHANGE")) {
    if (!strcmp(x, "reduce")) exchange = kExchangeReduce;
    else if (!strcmp(x, "peer")) exchange = kExchangePeerCopy;
    else if (strcmp(x, "gather")) return bad("GLAZE_MULTI_EXCHANGE must be `gather`, `reduce` or `peer`");
  }
  if (dup_ok && !strcmp(lbenv, "peer")) exchange = kExchangePeerCopy;
  const bool use_rccl = n > 1 && !loopback && exchange != kExchangePeerCopy;
  // RCCL first: without it nothing is touched (a renderer that already spans devices keeps them)
  const Rccl* nc = nullptr;
  if (use_rccl) {
    std::string why;
    nc = Rccl::get(why);
    if (!nc) {
      err.code = GLZ_E_DEVICE;
      err.msg = why;
      return false;
    }
  }
  if (!wait_idle(err)) return false;
  release_peers();
  // from here on every failure leaves ONE device rendering the whole frame
  auto fail = [&]() {
    release_peers();
    Error ignored;
    (void)set_partition_local(0, 1, ignored);
    request_new_frame_ = true;
    return false;
  };
  if (n == 1) return set_partition_local(0, 1, err) ? true : fail();
  for (int i = 1; i < n; ++i) peers_.emplace_back(new Peer());
  loopback_ = loopback;
  exchange_ = exchange;
  // instance + scene replica (upload, BVH build) + renderer for the tiles t % n == i, on every peer's own thread
  const Renderer* self = this;
  const Scene* src = scene_.get();
  const Instance* src_inst = scene_->instance ? scene_->instance : inst_;
  const uint32_t w = w_, h = h_, world = (uint32_t)n;
  const bool want_frame = !loopback && exchange == kExchangeReduce;
  std::vector<Peer*> order;
  for (auto& p : peers_) order.push_back(p.get());
  const std::vector<Peer*>* ord = &order;
  const bool built = forward([=](Peer& p, Error& e) {
    size_t i = 0;
    while ((*ord)[i] != &p) ++i;
    p.inst.reset(Instance::create(devices[i + 1], e));
    if (!p.inst) return false;
    // the replica takes the shape the root's scene HAS (builder, pair leaves, flattened or two levels), not what the
    // environment of this thread would choose
    p.inst->copy_build_options(*src_inst);
    if (src->info.as_levels) p.inst->as_levels = (int)src->info.as_levels;
    SceneData copy = src->data;
    std::shared_ptr<Scene> replica(Scene::create(p.inst.get(), std::move(copy), e));
    if (!replica) return false;
    p.r.reset(Renderer::create(p.inst.get(), replica, w, h, e));
    if (!p.r || !self->configure_peer(*p.r, e)) return false;
    if (!p.r->set_partition_local((uint32_t)i + 1, world, e)) return false;
    return !want_frame || hip_ok(p.frame.alloc((size_t)w * h), "alloc peer frame", e);
  }, err);
  if (!built) return fail();
  if (!set_partition_local(0, world, err)) return fail();
  if (!loopback && !use_rccl) {
    // direct access between device 0 and every peer (without it the copies are staged through the host); "already enabled" is fine
    for (int i = 1; i < n; ++i) {
      if (devices[i] == devices[0]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) == hipSuccess && can) {
        (void)hipSetDevice(devices[0]);
        (void)hipDeviceEnablePeerAccess(devices[i], 0);
        (void)hipSetDevice(devices[i]);
        (void)hipDeviceEnablePeerAccess(devices[0], 0);
      }
    }
    (void)hipGetLastError();
    (void)hipSetDevice(inst_->device);
  }
  if (use_rccl) {
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    const ncclResult_t r = nc->CommInitAll(comms.data(), n, devices);
    (void)hipSetDevice(inst_->device);
    if (r != ncclSuccess) {
      err.code = GLZ_E_DEVICE;
      err.msg = std::string("ncclCommInitAll: ") + nc->GetErrorString(r);
      return fail();
    }
    for (ncclComm_t c : comms) comms_.push_back(c);
  }
  request_new_frame_ = true;
  return true;
}

const Scene* Renderer::device_scene(int i) const {
  if (i == 0) return scene_.get();
  if (i < 0 || (size_t)i > peers_.size() || !peers_[(size_t)i - 1]->r) return nullptr;
  return peers_[(size_t)i - 1]->r->scene_.get();
}

bool Renderer::set_chains(uint32_t n, Error& err) {
  if (n > 16) {
    err.code = GLZ_E_ARG;
    err.msg = "at most 16 chains";
    return false;
  }
  if (!wait_idle(err)) return false;
  chains_wanted_ = n;
  if (!allocate(err)) return false;
  const bool ok = forward([=](Peer& p, Error& e) { return p.r->set_chains(n, e); }, err);
  (void)hipSetDevice(inst_->device);
  return ok;
}

bool Renderer::export_device(int which, void* dev, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (!gather(which != 0, static_cast<float4*>(dev), err)) return false;
  return hip_ok(hipStreamSynchronize(chains_[0]->stream), "export_device", err);
}

// One process per GPU: what a rank hands to the exchange instead of a zero-padded frame -- its tiles only, tile-major, in the
// order of its partition (local tile j = global tile rank + j * world), packed_count() float4s.
size_t Renderer::packed_count(uint32_t w, uint32_t h, uint32_t rank, uint32_t world) {
  if (world == 0 || rank >= world) return 0;
  const uint32_t tiles = ((w + kTile - 1) / kTile) * ((h + kTile - 1) / kTile);
  return tiles > rank ? (size_t)((tiles - rank + world - 1) / world) * kTile * kTile : 0;
}
bool Renderer::export_packed(int which, void* dev, Error& err) {
  if (!peers_.empty()) {
    err.code = GLZ_E_ARG;
    err.msg = "export_packed is for one rank of a process partition (a renderer that spans devices exchanges by itself)";
    return false;
  }
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  if (!settle(err)) return false;
  hipStream_t st = chains_[0]->stream;
  const uint32_t S = (uint32_t)chains_.size();
  for (uint32_t s = 0; s < S; ++s) {
    Chain& c = *chains_[s];
    if (!hip_ok(launch_pack_tiles(st, c.map.n_local_pixels, S, s, which ? c.result.ptr : c.cumulative.ptr, static_cast<float4*>(dev)), "k_pack_tiles", err)) return false;
  }
  return hip_ok(hipStreamSynchronize(st), "export_packed", err);
}
// rank 0 of such a job: the packed tiles of partition (rank, world) go to their place in a full frame (nothing else is touched)
bool Renderer::scatter_packed(uint32_t rank, uint32_t world, const void* dev_packed, void* dev_frame, Error& err) {
  if (world == 0 || rank >= world) {
    err.code = GLZ_E_ARG;
    err.msg = "bad tile partition";
    return false;
  }
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  TileMap m{};
  m.width = w_;
  m.height = h_;
  m.tiles_x = (w_ + kTile - 1) / kTile;
  m.tiles_y = (h_ + kTile - 1) / kTile;
  m.rank = rank;
  m.world = world;
  m.n_local_pixels = (uint32_t)packed_count(w_, h_, rank, world);
  m.n_local_tiles = m.n_local_pixels / (kTile * kTile);
  hipStream_t st = chains_[0]->stream;
  if (!hip_ok(launch_export(st, m, static_cast<const float4*>(dev_packed), static_cast<float4*>(dev_frame), false), "k_export (packed tiles)", err)) return false;
  return hip_ok(hipStreamSynchronize(st), "scatter_packed", err);
}

bool Renderer::scatter_packed_all(uint32_t world, const void* dev_packed, uint64_t stride_pixels, void* dev_frame, Error& err) {
  if (world == 0 || stride_pixels < packed_count(w_, h_, 0, world)) {
    err.code = GLZ_E_ARG;
    err.msg = "scatter_packed_all: the parts must lie at least packed_pixels(0, world) pixels apart";
    return false;
  }
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  hipStream_t st = chains_[0]->stream;
  for (uint32_t rank = 0; rank < world; ++rank) {
    TileMap m{};
    m.width = w_;
    m.height = h_;
    m.tiles_x = (w_ + kTile - 1) / kTile;
    m.tiles_y = (h_ + kTile - 1) / kTile;
    m.rank = rank;
    m.world = world;
    m.n_local_pixels = (uint32_t)packed_count(w_, h_, rank, world);
    m.n_local_tiles = m.n_local_pixels / (kTile * kTile);
    if (m.n_local_pixels == 0) continue;
    if (!hip_ok(launch_export(st, m, static_cast<const float4*>(dev_packed) + (size_t)rank * stride_pixels, static_cast<float4*>(dev_frame), false), "k_export (packed tiles)", err)) return false;
  }
  return hip_ok(hipStreamSynchronize(st), "scatter_packed_all", err);
}

bool Renderer::tonemap_device(const void* dev_result, uint8_t* out, Error& err) {
  if (!hip_ok(hipSetDevice(inst_->device), "hipSetDevice", err)) return false;
  hipStream_t st = chains_[0]->stream;
  if (!hip_ok(launch_tonemap(st, w_ * h_, static_cast<const float4*>(dev_result), oetf_thresholds_.ptr, rgba8_.ptr), "k_tonemap", err)) return false;
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
