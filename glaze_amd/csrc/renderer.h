// RayTraceRenderer on HIP (lib/src/vulkan/raytracer.rs:109-687): launch loop, per-launch frame
// constants, accumulation buffers, tile partition for one-process-per-GPU jobs.
#pragma once
#include <memory>
#include <vector>

#include "host_math.h"
#include "kernels.h"
#include "scene.h"

namespace glz {

class Renderer {
 public:
  static Renderer* create(Instance* inst, std::shared_ptr<Scene> scene /* may be null: empty scene */, uint32_t w, uint32_t h, Error& err);
  ~Renderer();

  bool set_integrator(int integrator, Error& err);
  bool set_exposure(float e);
  bool update_camera(const glz_camera& c, Error& err);
  bool change_resolution(uint32_t w, uint32_t h, Error& err);
  bool change_scene(std::shared_ptr<Scene> scene, Error& err);
  bool update_materials_and_lights(const glz_material* m, uint32_t nm, const glz_light* l, uint32_t nl, const glz_texture* t, uint32_t nt, Error& err);
  bool refresh_binded_textures(const glz_texture* t, uint32_t nt, Error& err);
  bool wait_idle(Error& err);
  uint32_t steps_per_sample() const { return integrator_ == GLZ_DIRECT ? 1u : pt_steps_; }

  bool draw(size_t spp, void (*cb)(void*), void* user, uint8_t* rgba8_out, Error& err);
  bool restart();
  bool step(uint32_t n, Error& err);
  bool read_rgba8(uint8_t* out, Error& err);
  bool read_frame(bool result, float* out, Error& err);

  bool set_texture_lod(int mode, Error& err);   // 0 = level 0 (the reference), 1 = ray cones; restarts
  bool set_seed(uint64_t s);
  bool set_depth(uint32_t d, Error& err);
  bool set_partition(uint32_t rank, uint32_t world, Error& err);
  // Several GPUs inside ONE process: devices[0] must be this renderer's own device; every further entry
  // gets its own stream, a replica of the scene, a renderer for the tiles t % n == i and a host thread that enqueues its
  // launches.  Reads sum the zero-padded RGBA32F frames onto devices[0] with one ncclReduce (RCCL over xGMI; one
  // communicator per device from ncclCommInitAll).  n == 1 returns to a single device.
  bool set_devices(const int* devices, int n, Error& err);
  uint32_t device_count() const { return 1u + (uint32_t)peers_.size(); }
  const Scene* device_scene(int i) const;    // the scene (replica) device i of set_devices renders; null when out of range
  bool set_chains(uint32_t n, Error& err);   // 0 = automatic
  // How a launch reaches the device: 1 = two kernels per launch (k_trace, k_shade: throughput, the full frame), 2 = the per-wave
  // launch loop k_path (one kernel per batch of launches: latency, a small tile share per GPU), 0 = by the pixels this device
  // owns.  Images do not depend on it.
  bool set_launch_mode(int mode, Error& err);
  bool path_mode() const { return path_mode_; }
  // Which nodes the two-kernel mode's traversal walks: 4 (k_trace), 8 (k_trace8: flattened scenes, counters off), 0 = by the pixels this device owns.
  bool set_node_width(int width, Error& err);
  bool wide8() const { return wide8_ && !counting_; }
  uint32_t chains() const { return (uint32_t)chains_.size(); }
  static uint32_t chains_for(uint32_t w, uint32_t h, uint32_t rank, uint32_t world, uint32_t wanted);
  bool export_device(int which, void* dev_rgba32f, Error& err);
  static size_t packed_count(uint32_t w, uint32_t h, uint32_t rank, uint32_t world);   // float4s of a rank's packed tiles
  bool export_packed(int which, void* dev_packed, Error& err);
  bool scatter_packed(uint32_t rank, uint32_t world, const void* dev_packed, void* dev_frame, Error& err);
  bool scatter_packed_all(uint32_t world, const void* dev_packed, uint64_t stride_pixels, void* dev_frame, Error& err);   // every rank's part of one gathered buffer, one synchronisation
  bool tonemap_device(const void* dev_result, uint8_t* out, Error& err);
  bool launch_constants(uint32_t launch, uint32_t* seed, float off[2]);
  void push_constants(float out[32]) const;
  void enable_counters(int flags);
  bool get_stats(glz_render_stats* out, Error& err);

  Instance* instance() const { return inst_; }
  Scene* scene() const { return scene_.get(); }
  uint32_t width() const { return w_; }
  uint32_t height() const { return h_; }

 private:
  Renderer() = default;
  bool allocate(Error& err);
  bool reset_buffers(Error& err);
  bool one_launch(Error& err);
  bool launch_constants_common(FrameData& fd, Error& err);
  bool path_batch(uint32_t n, Error& err);
  bool run_launches(uint32_t n, Error& err);
  bool use_path() const { return path_mode_ && !counting_ && chains_.size() == 1 && chains_[0]->grid_path != 0; }
  bool gather(bool result, float4* dst, Error& err, bool zero_first = true);

  Instance* inst_ = nullptr;
  std::shared_ptr<Scene> scene_;   // shared with the glz_scene handle it came from (info / debug hooks stay valid)
  uint32_t w_ = 0, h_ = 0;
  int integrator_ = GLZ_PATH_TRACE;
  uint32_t pt_steps_ = 6;   // PT_STEPS, raytrace_structures.rs:87
  int lod_mode_ = 0;        // texture level of detail: 0 = level 0 always (what the reference's ray-tracing stages do), 1 = ray cones, 2 = ray cones with an anisotropic footprint
  float exposure_ = 1.0f;
  glz_camera camera_{};
  CameraConsts cam_{};
  uint64_t seed_ = 0;
  host::SeedStream rng_;
  host::WorkScheduler sched_;
  bool request_new_frame_ = true;
  uint32_t rank_ = 0, world_ = 1;   // tile partition of this process (glz_renderer_set_partition)
  uint32_t chains_wanted_ = 0;      // 0 = automatic (pick_chains)
  int launch_mode_ = getenv("GLAZE_LAUNCH_MODE") ? atoi(getenv("GLAZE_LAUNCH_MODE")) : 0;   // set_launch_mode
  bool path_mode_ = false;          // decided in allocate(): this device's launches run as k_path batches
  int node_width_ = getenv("GLAZE_NODE_WIDTH") ? atoi(getenv("GLAZE_NODE_WIDTH")) : 0;   // set_node_width
  bool wide8_ = false;              // decided in allocate(): k_trace8 (the 8-wide nodes) traces this device's rays while the counters are off

  // One chain = one independent sequence of launches over a subset of this rank's tiles, on its own HIP stream.
  // Pixels never interact, so the tiles of a rank can advance as several concurrent chains: chain s of S renders the
  // tiles of the finer partition (rank + s * world, world * S).  With few pixels per GPU a launch is bound by the
  // latency of its longest rays, not by throughput; concurrent chains fill the machine during those tails (strong
  // scaling of the 1080p frame over 8 GPUs).  Results are bit-identical for every S.
  struct EventSet {
    hipEvent_t e[4];
    int kind;   // 0: e[0]..e[2] around k_trace, k_shade; 1: e[0]..e[1] around a stand-alone shadow pass; 2: e[0]..e[1] around k_path
    uint32_t weight;   // kind 0: the launches this timed one stands for (event_stride())
  };
  uint64_t event_stride() const;
  struct Chain {
    TileMap map{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DeviceBuffer<float4> ray_o, ray_d, imp[4], hit, sh_o, sh_d, contrib, cumulative, result;
    DeviceBuffer<float> cone;
    DeviceBuffer<uint32_t> hit_inst;
    DeviceBuffer<uint32_t> overflow, queue_count, path_cost;
    uint32_t path_batches = 0;
    uint32_t grid = 0, grid_counting = 0;   // blocks of k_trace's persistent grid (plain / instrumented kernel)
    uint32_t grid8 = 0;                     // ... of k_trace8's (0 unless the renderer walks the 8-wide nodes)
    uint32_t grid_path = 0;                 // blocks of k_path's grid (0: this chain never runs it)
    // shadow rays queued by the last launch's k_shade and not traced yet (they ride in the next launch's k_trace, or in
    // a stand-alone pass as soon as anything looks at the images: flush_shadows)
    bool shadow_pending = false;
    uint32_t pending_set = 0;
    float pending_exposure = 1.0f;
    std::vector<EventSet> pending_events;   // per-launch kernel boundaries, resolved lazily in get_stats
    std::vector<EventSet> free_events;
    double trace_ms = 0, shade_ms = 0, flush_ms = 0, path_ms = 0;
  };
  std::vector<std::unique_ptr<Chain>> chains_;
  uint32_t pick_chains() const;
  void release_chains();
  bool flush_shadows(Chain& c, Error& err);
  bool acquire_events(Chain& c, EventSet& ev, Error& err);
  void resolve_events(Chain& c);
  void fill_args(const Chain& c, LaunchArgs& a) const;

  // ---- other GPUs of this process (set_devices) ----
  struct Peer;
  struct Pending;
  std::vector<std::unique_ptr<Peer>> peers_;
  std::vector<void*> comms_;   // ncclComm_t per device (index 0 = this renderer); empty in loop-back mode
  bool loopback_ = false;      // all "devices" are this one device (GLAZE_MULTI_LOOPBACK=1, tests on a one-GPU box): no RCCL
  enum { kExchangeGather = 0, kExchangeReduce = 1, kExchangePeerCopy = 2 };
  int exchange_ = kExchangeGather;      // how the peers' tiles reach device 0 (reduce_peers); GLAZE_MULTI_EXCHANGE at set_devices
  DeviceBuffer<float4> recv_stage_;     // device 0: the packed tiles received from the peers (gather shape)
  bool settle(Error& err);
  template <class F> void post_all(F f, Pending& p);
  bool join_all(Pending& p, Error& err);
  template <class F> bool forward(F f, Error& err);
  void release_peers();
  bool set_partition_local(uint32_t rank, uint32_t world, Error& err);
  bool configure_peer(Renderer& p, Error& err) const;
  bool reduce_peers(bool result, float4* dst, Error& err);
  bool step_local(uint32_t n, Error& err);

  DeviceBuffer<float4> frame_tmp_;
  DeviceBuffer<uchar4> rgba8_;
  DeviceBuffer<float> oetf_thresholds_;   // sRGB8 quantiser thresholds (host::srgb8_thresholds), see k_tonemap
  DeviceBuffer<TraceCounters> counters_;
  // stats
  bool counting_ = false;
  uint64_t launches_ = 0;
  bool profile_kernels_ = true;
};

}  // namespace glz
