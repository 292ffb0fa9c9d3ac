// k_path: the per-wave launch loop of a small tile share (see below), in a translation unit of its own because it is compiled
// with -mllvm -disable-machine-licm: inside the kernel's launch loop MachineLICM hoists every constant the shading code materialises
// (v_mov 2.0, 0.5, ... and scalar literals) into the loop's preheader, where they stay live through the whole body -- 170 scratch
// loads / 75 stores and 55 SGPR-to-lane spills in the 128-register kernel against 40 / 9 / 4 without the pass -- and the shading
// phase of a wave took 55 us instead of 28 (tools/gpu_path_phases.py).
#include "device/wavefront.h"

namespace glz {
using namespace dev;

// ---------------------------------------------------------------------------------------------
// k_path: the launch loop of a SMALL tile share (a 1080p frame over 8 GPUs: 259 k pixels per device) inside one persistent kernel.
// With so few pixels the chip holds one 64-pixel group per resident wave and the two-kernel launch lasts as long as its slowest
// wave -- the median wave of a 1/8 share is done after 60-67 us, the last one after 99-118 us (tools/gpu_wave_times.py), and every launch
// pays that maximum again, twice (k_trace, k_shade).  Pixels are independent (path_trace.rgen:143-168: state, RNG and accumulator are
// per pixel), so nothing forces a wave to wait for the others: here every wave carries ITS 64 pixels through
//     closest hits of launch L -> shadow rays queued by launch L-1 (+ update_count / update_result) -> shade of launch L
// for all launches of the batch (up to 192: only seed, jitter offset and exposure differ between launches, 16 bytes each in the kernel
// arguments), with no grid-wide boundary in between.  A step then costs the slowest wave's SUM over the launches instead of the
// sum over launches of the slowest wave.  (Measured: DESIGN.md section 6.)  Per pixel the operations and their order are those of k_trace / k_shade
// (same sources, same shade_pixel, shadow rays of a launch resolved before the next launch's shading), so the image is bit-identical
// -- tests/test_gpu_render.py compares the two modes and the oracle.
// A wave's closest-hit records stay in LDS, its shadow queue is its own 64 entries of the queue arrays (no atomics, no shards), and
// the batch ends with the shadow rays of its last launch, so nothing is pending when the kernel ends.  The kernel runs at k_shade's
// 128 registers, four waves per SIMD -- the objection to a fused kernel at full-frame size, where throughput counts; here every wave of
// the share is resident anyway.  The sky's marginal table stays in global memory (with it in LDS only three blocks fit a CU).
// ---------------------------------------------------------------------------------------------
// THE ORDERING CONTRACT of k_path.  Lanes of one wave hand data to each other through GLOBAL memory: the shading lane of a pixel
// writes a shadow-queue entry (sh_o / sh_d / contrib) that whichever lane picks the ray up in the next traversal pass reads; the lane
// that finishes a shadow ray read-modify-writes cumulative / result of the OWNING pixel, which that pixel's own lane reads and writes
// again in the next shading phase.  (The two-kernel mode has a kernel boundary in each of these places.)  What makes this defined is a
// release / acquire pair at WAVEFRONT scope at every phase boundary: all lanes of a wave go through one vector-memory pipeline and one
// L1, which performs a wave's accesses to an address in program order, so at this scope the fence needs no cache action and no
// s_waitcnt on gfx950 -- it costs no instruction -- but it forbids the compiler to move, merge or keep in registers any of these
// accesses across the boundary, which nothing else did.  (Not valid under tgsplit / a per-lane L1 policy: the kernel is never built that way.)
__device__ __forceinline__ void wave_handover_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GroupHitSink {   // closest-hit record of ray i of the group -> the wave's LDS slots
  float4* hit;
  __device__ __forceinline__ void store(uint32_t i, const HitRecord& h) {
    hit[i] = make_float4(h.leaf == 0xFFFFFFFFu ? INFINITY : h.t, h.u, h.v, __uint_as_float(h.leaf));
  }
};
struct GroupQueue {     // shade_pixel's shadow-queue policy: entry k of the wave's own 64 (the lanes that push, in lane order)
  uint32_t base;
  bool pushed;
  __device__ __forceinline__ uint32_t slot(bool push) {
    const unsigned long long m = __ballot(push);
    pushed = push;
    return base + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
  }
};
struct GroupShadowSource {
  const LaunchArgs& A;
  uint32_t base;
  uint32_t lid;            // per-lane: owning pixel and contribution of the ray in flight
  float4 contrib;
  // The owning pixel's accumulator, read when the ray STARTS: nobody else touches it while the ray is in flight (one shadow ray per pixel
  // and launch, the pixel's own lane only shades after this pass), and it was last written a launch ago -- read at the ray's end, where
  // update_count / update_result need it, the lane (and with it the wave's round) waited for it to come from HBM
  // (tools/gpu_sections.py: merge + retire was a fifth of a wave's tracing time).
  float4 cum;
  __device__ __forceinline__ bool load(uint32_t i, vec3& o, vec3& d, float& tmin, float& tmax) {
    const uint32_t q = base + i;
    const float4 so = A.st.sh_o[q], sd = A.st.sh_d[q];
    contrib = A.st.contrib[q];
    lid = __float_as_uint(sd.w);
    cum = A.st.cumulative[lid];
    o = mk3(so.x, so.y, so.z);
    d = mk3(sd.x, sd.y, sd.z);
    tmin = 0.001f;
    tmax = so.w;
    return true;
  }
};
struct GroupShadowSink {
  const LaunchArgs& A;
  GroupShadowSource& src;
  float exposure;          // of the launch that queued the rays
  __device__ __forceinline__ void store(uint32_t, const HitRecord& h) {
    const bool occluded = h.leaf != 0xFFFFFFFFu;
    const uint32_t flags = __float_as_uint(src.contrib.w);
    vec3 c = mk3(src.contrib.x, src.contrib.y, src.contrib.z);
    bool add = !occluded;
    if (occluded && (flags & kFlagPoison)) {
      const float nan = __uint_as_float(0x7FC00000u);
      c = mk3(nan, nan, nan);
      add = true;
    }
    accumulate_pixel(A, src.lid, c, add, true, exposure, src.cum);
  }
};

struct GroupMixedSource {   // rays 0..63: the group's pixels; 64..: the shadow rays its previous shading queued
  ClosestSource closest;
  GroupShadowSource shadow;
  bool any;
  __device__ __forceinline__ bool load(uint32_t i, vec3& o, vec3& d, float& tmin, float& tmax) {
    any = i >= 64u;
    return any ? shadow.load(i - 64u, o, d, tmin, tmax) : closest.load(i, o, d, tmin, tmax);
  }
};
struct GroupMixedSink {
  GroupHitSink closest;
  GroupShadowSink shadow;
  __device__ __forceinline__ void store(uint32_t i, const HitRecord& h) {
    if (i >= 64u) shadow.store(i - 64u, h); else closest.store(i, h);
  }
};

// The kernel's arguments, re-read: behind the empty asm the compiler no longer knows that the pointer is the one it has been loading
// from, so what a phase of k_path needs of the arguments is loaded (scalar loads from the kernarg segment) where the phase begins and
// dies where it ends -- instead of every pointer either phase uses staying in SGPRs through the whole launch loop (tracing and shading
// together use more of them than there are: 233 of them went to VGPR lanes, and the VGPRs those took to scratch).
// RTFrameData of launch L of the batch: what all launches share (LaunchArgs::frame) with the three per-launch fields from the batch
__device__ __forceinline__ FrameData launch_frame(const LaunchArgs& A, const PathBatch& B, uint32_t L) {
  FrameData F = A.frame;
  F.seed = B.seed[L];
  F.pixel_offset[0] = B.offset[L][0];
  F.pixel_offset[1] = B.offset[L][1];
  F.next_pixel_offset[0] = B.offset[L + 1][0];
  F.next_pixel_offset[1] = B.offset[L + 1][1];
  F.exposure = B.exposure[L];
  return F;
}
constexpr uint32_t kPathBatchOffset = (uint32_t)(((sizeof(LaunchArgs) + alignof(PathBatch) - 1) / alignof(PathBatch)) * alignof(PathBatch));   // PathBatch in k_path's kernarg segment
// the shading phase of k_path for the wave's 64 pixels; returns how many of them queued a shadow ray (entries 64 g .. of the queue arrays)
template <bool LOD>
__device__ __forceinline__ uint32_t path_shade(uint32_t g, uint32_t n_groups, uint32_t lane, uint32_t L, const DeviceScene& S_lds, const float4* hit) {
  const uint32_t lid0 = g * 64u;   // the wave's own 64 entries of the shadow-queue arrays
  const LaunchArgs& A = *(const LaunchArgs*)reread_kernarg();
  // the scene as the shading code sees it: the arguments' pointers, re-read, with the tables this block staged in LDS in their place
  DeviceScene S = A.scene;
  S.tex_counter = nullptr;   // k_path is never a counting pass: the checks fold away
  S.srgb_lut = S_lds.srgb_lut;
  S.materials = S_lds.materials;
  S.lights = S_lds.lights;
  S.tex_desc = S_lds.tex_desc;
  const FrameData F = launch_frame(A, *(const PathBatch*)(reread_kernarg() + kPathBatchOffset), L);
  const uint32_t lid = g * 64u + lane;
  const PixelId px = pixel_of(A.map, lid);
  GroupQueue queue{lid0, false};
  if (px.active) {
    const float4 ro = A.st.ray_o[lid], rd = A.st.ray_d[lid];
    DirectState state{A};
    shade_pixel<LOD>(A, S, F, lid, px, ro, rd, hit[lane], queue, state);
  }
  return (uint32_t)__popcll(__ballot(queue.pushed));
}

#ifdef GLZ_PATH_TIMES   // tuning builds only (tools/gpu_path_phases.py): 10 ns ticks every wave spent tracing / shading, summed over the launches of the last k_path
__device__ unsigned long long g_path_times[3 * 8192];
#endif
template <bool LOD>
__global__ void __launch_bounds__(kBlock, GLZ_PATH_WAVES) k_path(const LaunchArgs A, const PathBatch B) {
  __shared__ int s_stack[kLdsStack * kBlock];
  __shared__ alignas(1024) int s_aux[kAuxPerBlock];
  __shared__ uint4 s_top[kLdsTop ? kBvhTopNodes * 4 : 1];
  __shared__ float s_lut[256];
  __shared__ float4 s_hit[kBlock];
  extern __shared__ uint4 s_dyn[];   // [RTMaterial x n_materials | RTLight x n_rt_lights | TexDesc x n_textures] when B.tables_in_lds
  stage_top(A.scene, s_top);
  s_lut[threadIdx.x] = A.scene.srgb_lut[threadIdx.x];
  const uint32_t qm = A.scene.n_materials * (uint32_t)(sizeof(RTMaterial) / 16), ql = A.scene.n_rt_lights * (uint32_t)(sizeof(RTLight) / 16),
                 qt = A.scene.n_textures * (uint32_t)(sizeof(TexDesc) / 16);
  if (B.tables_in_lds) {
    const uint4* gm = reinterpret_cast<const uint4*>(A.scene.materials);
    const uint4* gl = reinterpret_cast<const uint4*>(A.scene.lights);
    const uint4* gt = reinterpret_cast<const uint4*>(A.scene.tex_desc);
    for (uint32_t i = threadIdx.x; i < qm + ql + qt; i += kBlock) s_dyn[i] = i < qm ? gm[i] : (i < qm + ql ? gl[i - qm] : gt[i - qm - ql]);
  }
  __syncthreads();   // the only block-wide barrier: from here on the four waves of the block never wait for each other
  DeviceScene S = A.scene;
  S.srgb_lut = s_lut;
  if (B.tables_in_lds) {
    S.materials = reinterpret_cast<const RTMaterial*>(s_dyn);
    S.lights = reinterpret_cast<const RTLight*>(s_dyn + qm);
    S.tex_desc = reinterpret_cast<const TexDesc*>(s_dyn + qm + ql);
  }
  // wave-uniform values in scalar registers: the compiler cannot see that threadIdx.x >> 6 is the same in all lanes of a wave
  const uint32_t wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t my_wave = blockIdx.x * (kBlock / 64) + wave_in_block;
  int* aux = wave_aux(s_aux, wave_in_block);
  int* links = wave_links(s_aux, wave_in_block);
  float4* hit = &s_hit[wave_in_block * 64u];
  const uint32_t lane = threadIdx.x & 63u;
  TraceTally tally;
#ifdef GLZ_PATH_TIMES
  unsigned long long pt_trace = 0, pt_shade = 0, pt_begin = wall_clock64();
#endif
  const uint32_t n_groups = (A.map.n_local_pixels + 63u) / 64u;
  for (uint32_t g = my_wave; g < n_groups; g += wave_count()) {
    const uint32_t lid0 = g * 64u;
    // Groups differ in cost, persistently (a region of the image stays as hard as it is), and the kernel lasts as long as its
    // slowest wave: a wave whose group took longer than the mean in the last batch issues ahead of the others on its SIMD.
    const unsigned long long group_begin = wall_clock64();
    {
      const uint32_t* acc = A.st.path_cost + 4u * (B.parity ^ 1u);
      const unsigned long long sum = (unsigned long long)acc[0] | ((unsigned long long)acc[1] << 32);
      const uint32_t cnt = acc[2], mine = A.st.path_cost[8u + g];
      int prio = 0;
      if (cnt != 0u && mine != 0u) {
        const float r = (float)mine * (float)cnt / (float)sum;
        prio = r > 1.2f ? 3 : (r > 1.08f ? 2 : (r > 0.97f ? 1 : 0));
      }
      prio = __builtin_amdgcn_readfirstlane(prio);
      if (prio == 3) __builtin_amdgcn_s_setprio(3); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    uint32_t n_shadow = 0;      // wave-uniform: shadow rays the group's last shading queued
    float queued_exposure = 0.0f;
    for (uint32_t L = 0;; ++L) {
      const LaunchArgs& A = *(const LaunchArgs*)reread_kernarg();   // the tracing phase's view of the arguments (shadows the parameter)
      const PathBatch& B = *(const PathBatch*)(reread_kernarg() + kPathBatchOffset);
      if (n_shadow != 0u && L == B.n) {
        // the batch ends with the shadow rays of its last launch (nothing of the next launch depends on them)
      } else if (L >= B.n) {
        break;
      }
#ifdef GLZ_PATH_TIMES
      const unsigned long long pt0 = wall_clock64();
#endif
      wave_handover_fence();   // shading (or the previous group) -> tracing: queue entries, accumulators
      if (L < B.n) {
        const FrameData F = launch_frame(A, B, L);
        // ONE traversal pass: the 64 closest-hit rays of launch L, then -- in the lanes those leave idle -- the shadow rays launch L-1 queued
        GroupMixedSource src{ClosestSource{A, F, tally, lid0}, GroupShadowSource{A, lid0, 0u, make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)}, false};
        GroupMixedSink sink{GroupHitSink{hit}, GroupShadowSink{A, src.shadow, queued_exposure}};
        trace_wave<false, false, true, GLZ_PATH_PREFETCH != 0>(A.scene, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, 64u + n_shadow, 0u, 1u, tally);
        n_shadow = 0u;
      } else {
        GroupShadowSource src{A, lid0, 0u, make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
        GroupShadowSink sink{A, src, queued_exposure};
        trace_wave<true, false, false, GLZ_PATH_PREFETCH != 0>(A.scene, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, n_shadow, 0u, 1u, tally);
        n_shadow = 0u;
      }
#ifdef GLZ_PATH_TIMES
      const unsigned long long pt1 = wall_clock64();
      pt_trace += pt1 - pt0;
#endif
      wave_handover_fence();   // tracing -> shading: the accumulators the shadow rays' lanes updated
      if (L >= B.n) break;
      n_shadow = path_shade<LOD>(g, n_groups, lane, L, S, hit);
      queued_exposure = B.exposure[L];
#ifdef GLZ_PATH_TIMES
      pt_shade += wall_clock64() - pt1;
#endif
    }
    if (lane == 0u && B.n != 0u) {
      const uint32_t ticks = (uint32_t)((wall_clock64() - group_begin) / B.n);
      A.st.path_cost[8u + g] = ticks;
      uint32_t* acc = A.st.path_cost + 4u * B.parity;
      atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)ticks);
      atomicAdd(acc + 2, 1u);
    }
  }
#ifdef GLZ_PATH_TIMES
  if ((threadIdx.x & 63) == 0 && wave_index() < 8192u) {
    g_path_times[3 * wave_index()] = pt_trace;
    g_path_times[3 * wave_index() + 1] = pt_shade;
    g_path_times[3 * wave_index() + 2] = wall_clock64() - pt_begin;
  }
#endif
}

// dynamic LDS of k_path: the scene's material / light / texture-descriptor tables when they fit kShadeTableBytes, else nothing
static uint32_t path_table_bytes(const DeviceScene& sc) {
  const uint32_t bytes = sc.n_materials * (uint32_t)sizeof(RTMaterial) + sc.n_rt_lights * (uint32_t)sizeof(RTLight) + sc.n_textures * (uint32_t)sizeof(TexDesc);
  return bytes <= kShadeTableBytes ? bytes : 0u;
}
uint32_t path_resident_blocks(const DeviceScene& sc) {
  int dev = 0, cus = 256, per_cu = 4;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_path<false>, kBlock, path_table_bytes(sc)) != hipSuccess || per_cu < 1) per_cu = 2;
  per_cu = std::min(per_cu, 8);
  return (uint32_t)cus * (uint32_t)per_cu;
}
uint32_t path_grid_blocks(uint32_t n_local_pixels, const DeviceScene& sc) {
  const uint32_t groups = (n_local_pixels + 63u) / 64u, blocks = (groups + kBlock / 64 - 1) / (kBlock / 64);
  return std::max<uint32_t>(1u, std::min<uint32_t>(blocks, path_resident_blocks(sc)));
}
hipError_t launch_path(hipStream_t st, const LaunchArgs& a, const PathBatch& batch, uint32_t blocks) {
  static_assert(sizeof(LaunchArgs) + sizeof(PathBatch) <= 4096, "kernel arguments of k_path");
  if (a.map.n_local_pixels == 0 || batch.n == 0) return hipSuccess;
  if (blocks == 0 || batch.n > kPathMaxLaunches || a.scene.two_level || a.counters) return hipErrorInvalidValue;
  PathBatch b = batch;
  const uint32_t dyn = path_table_bytes(a.scene);
  b.tables_in_lds = dyn != 0u ? 1u : 0u;
  if (a.frame.lod_mode != 0u) hipLaunchKernelGGL(k_path<true>, dim3(blocks), dim3(kBlock), dyn, st, a, b);
  else hipLaunchKernelGGL(k_path<false>, dim3(blocks), dim3(kBlock), dyn, st, a, b);
  return hipGetLastError();
}
}  // namespace glz

#ifdef GLZ_SECTION_TIMES
extern "C" int glz_debug_sections_path(unsigned long long* out, int reset) {
  int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_sections), sizeof(unsigned long long) * 16 * 8192);
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(glz::g_sections)) == hipSuccess) e = (int)hipMemset(p, 0, sizeof(unsigned long long) * 16 * 8192);
  }
  return e;
}
#endif
#ifdef GLZ_PATH_TIMES
extern "C" int glz_debug_path_times(unsigned long long* out, int n_waves) {
  if (n_waves > 8192) n_waves = 8192;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_path_times), sizeof(unsigned long long) * 3 * (size_t)n_waves);
}
#endif
