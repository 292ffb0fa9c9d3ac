// Host-callable launchers of the HIP kernels (kernels_build.hip, kernels_render.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device/types.h"

namespace glz {

// ---- scene build --------------------------------------------------------------------------------
hipError_t launch_derivatives(hipStream_t st, const float4* vertices, const uint32_t* indices, uint32_t n_tris, float4* out);

struct LbvhInputs {
  const float4* vertices;
  const uint32_t* indices;
  const RTInstance* instances;
  const uint32_t* inst_base;   // exclusive prefix sum of triangles per instance (n_instances entries)
  uint32_t n_instances;
  const TransformPair* transforms;
  const RTMaterial* materials;
  uint32_t n_world;
  int builder;                 // kBvhBuilder*
  float pair_area_ratio;       // two triangles share a leaf when area(joint box) <= ratio * (area(a) + area(b)); 0 = never (k_pair_triangles)
  // Hierarchy over GIVEN boxes instead of triangles (the instance level of a two-level structure): n_world boxes, box i becomes a
  // one-"triangle" leaf whose BvhTri::world_id is i; vertices / indices / instances are not read.  Device pointers; null = triangles.
  const float4* given_lo = nullptr;
  const float4* given_hi = nullptr;
  // The flattened world build: also emit the 64-byte per-leaf records (BvhQuad) and make leaf links ~leaf number instead of
  // ~first triangle slot (the record names the slot).  Pairs are only formed in the two vertex orders a quad record can hold.
  bool emit_quads = false;
  // ... and the same hierarchy collapsed eight wide (types.h BvhNode8; needs emit_quads): LbvhOutputs::nodes8
  bool emit_wide8 = false;
};
constexpr int kBvhBuilderLbvh = 0, kBvhBuilderPloc = 1, kBvhBuilderSah = 2, kBvhBuilderAuto = 3, kBvhBuilderSahHost = 4;   // = GLZ_BVH_LBVH / _PLOC / _SAH / _AUTO / _SAH_HOST
// host side of the SAH builder (bvh_sah.cpp): binary hierarchy over n leaf boxes -> children / parent arrays
void build_sah_host(uint32_t n, const float4* lo, const float4* hi, int2* children, int* parent);
struct LbvhOutputs {
  BvhNode4* nodes;  // n_nodes entries, hipMalloc'ed by build_lbvh: the caller owns them afterwards
  uint32_t n_nodes;
  BvhNode8* nodes8; // emit_wide8: n_nodes8 entries, hipMalloc'ed by build_lbvh (the caller owns them), else null; a scene of one leaf has none
  uint32_t n_nodes8, depth8;
  BvhGrid grid;     // quantisation grid of the node boxes
  BvhTri* tris;     // n_world + 1 entries (the tracer reads one past a leaf's first triangle), preallocated; leaf order, a leaf's triangles adjacent
  BvhQuad* quads;   // emit_quads: n_leaves records, hipMalloc'ed by build_lbvh (the caller owns them), else null
  uint32_t n_leaves;
  uint32_t depth;   // number of 4-wide nodes above the deepest leaf (the traversal stack holds at most 3 * depth + 1 entries)
  float sah;
  uint32_t rounds;  // PLOC merge rounds (0 for the LBVH)
  float bounds_lo[3], bounds_hi[3];
};
hipError_t build_lbvh(hipStream_t st, const LbvhInputs& in, LbvhOutputs& out);
// top-of-tree table for LDS staging (types.h kBvhTopNodes): `top` receives kBvhTopNodes nodes
hipError_t launch_top_table(hipStream_t st, const BvhNode4* nodes, uint32_t n_nodes, BvhNode4* top);
#ifdef GLZ_NODE48
// experiment: 64-byte nodes -> 48-byte nodes (types.h BvhNode48), n of them
hipError_t launch_compress_nodes(hipStream_t st, const BvhNode4* nodes, uint32_t n, BvhNode48* out);
#endif
// 128-byte per-leaf shading records (see k_shade_records); xf_identity[t] != 0 marks an exact identity transform
hipError_t launch_shade_records(hipStream_t st, uint32_t n, const BvhTri* tris, const RTInstance* instances, const uint32_t* indices,
                                const float4* vertices, const float4* derivatives, const uint32_t* xf_identity, float4* out);
// DeviceScene::alpha_recs for `n` triangle slots (flattened scenes with an opacity map)
hipError_t launch_alpha_records(hipStream_t st, uint32_t n, const float4* shade_tris, const RTMaterial* materials, const TexDesc* tex_desc, float4* out);

// Traversal stack entries each lane keeps in LDS (a near-first 4-wide traversal holds at most three entries per
// level of the tree; what does not fit spills to a per-lane HBM area).
constexpr int kTraversalLdsStack = 17;   // 17 levels + the staged top of the tree = 25 920 bytes per block: six blocks per CU (LDS is granted in 1 280-byte granules: 18 levels + the table would take 22 granules and leave room for five)

// ---- rendering ------------------------------------------------------------------------------------
// Per-pixel wavefront state, indexed by the LOCAL pixel id `lid` (tile-major, one wave = one 8x8 block):
//   lid = ((local_tile * 64 + sub_block) * 64 + lane)
struct PathState {
  float4* ray_o;     // origin.xyz, bounce number        (PTLastVertex.hit, raytrace_structures.rs:89-95)
  float4* ray_d;     // direction.xyz, last-bounce-specular flag (PTLastVertex.wi)
  float4* imp[4];    // importance spectrum, 4 x vec4    (PTLastVertex.importance)
  float4* hit;       // t, u, v, leaf index (bits)       closest-hit record of the current launch
  float* cone;       // ray-cone width at the ray origin (texture LOD, FrameData::lod_mode; untouched when it is off)
  uint32_t* hit_inst;// RTInstance of the closest hit (two-level scenes only: a flattened triangle record names its instance)
  // shadow-ray queue, compacted by k_shade (entry q, not pixel lid):
  float4* sh_o;      //   origin.xyz, tmax
  float4* sh_d;      //   direction.xyz, owning pixel lid (bits)
  float4* contrib;   //   rgb radiance to add if unoccluded, flags (bits)
  uint32_t* queue_count;
  float4* cumulative;// accumulate_image (xyz = sum rgb, w = launches)
  float4* result;    // result_image (out32)
  uint32_t* overflow;// traversal stack spill, `overflow_depth` words per lane slot of the k_trace grid
  uint32_t overflow_depth;
  uint32_t* path_cost;// k_path: [0..7] two accumulators {sum of per-launch ticks (u64), groups (u32), pad}, then one word per 64-pixel group: its ticks per launch in the last batch
};

struct TileMap {
  uint32_t width, height;
  uint32_t tiles_x, tiles_y;
  uint32_t rank, world;      // this renderer owns global tiles t with t % world == rank
  uint32_t n_local_tiles;
  uint32_t n_local_pixels;   // n_local_tiles * 4096
};

struct CameraConsts {
  float camera2world[16];
  float screen2camera[16];
};

struct TraceCounters {
  unsigned long long closest_rays, shadow_rays, closest_nodes, closest_tris, shadow_nodes, shadow_tris, hits, fresh;
  unsigned long long phase[12];   // {node_iters, node_lanes, leaf_iters, leaf_lanes, refill_iters, refill_lanes} x {closest, shadow}
  // DeviceScene::tex_counter of k_shade / of k_trace (alpha tests): {fetches, texel bytes, light samples other than sky, sky-light samples}
  unsigned long long shade_tex[4], trace_tex[4];
};

struct LaunchArgs {
  DeviceScene scene;
  PathState st;
  TileMap map;
  FrameData frame;
  CameraConsts cam;
  TraceCounters* counters;   // nullptr unless counting is enabled
  // k_trace phases of this call: closest-hit rays of the current launch and / or the shadow rays the PREVIOUS launch's
  // k_shade queued (they only gate an accumulation, so nothing of the current launch depends on them)
  uint32_t do_closest, do_shadow;
  uint32_t shade_set;        // which of the two shadow-queue counter sets this launch's k_shade fills (the other one is drained)
  float shadow_exposure;     // exposure of the launch that queued the shadow rays (update_result uses it)
};
// The launches one k_path call runs (kernels_path.hip): what differs between launches, by value in the kernel arguments (16 bytes a
// launch next to the 848 of LaunchArgs: 192 launches stay inside the 4 KB the arguments may take).  Long batches matter: the kernel
// ends when its slowest wave does, and a wave's time per launch scatters by ~20 % -- over 16 launches the slowest of 4 096 waves is
// 27 % above the mean, over 192 launches 8 %.
constexpr uint32_t kPathMaxLaunches = 192;
struct PathBatch {
  uint32_t n;                              // launches in this call
  uint32_t tables_in_lds;                  // filled by launch_path
  uint32_t parity;                         // which of the two cost accumulators this batch adds to (it reads the other one)
  uint32_t seed[kPathMaxLaunches];         // FrameData::seed of each launch (the rest of FrameData is LaunchArgs::frame)
  float offset[kPathMaxLaunches + 1][2];   // FrameData::pixel_offset; entry n: of the launch after the batch (FrameData::next_pixel_offset)
  float exposure[kPathMaxLaunches];        // FrameData::exposure
};
constexpr uint32_t kTraceBlock = 256;          // threads per block of the render kernels (4 waves)
constexpr uint32_t kQueueSetWords = 8 * 32;   // 8 shard counters, 128 bytes apart

// persistent grid of k_trace / k_trace_tl / (wide8) k_trace8 (device must be current); wide8: the walk over the 8-wide nodes, flattened scenes without work counters
uint32_t trace_grid_blocks(uint32_t n_local_pixels, bool counting, bool two_level, bool wide8 = false);
hipError_t launch_trace(hipStream_t st, const LaunchArgs& a, uint32_t blocks, bool wide8 = false);
hipError_t launch_shade(hipStream_t st, const LaunchArgs& a);
// the per-wave launch loop of a small tile share: `batch.n` launches for every pixel in ONE kernel (a.frame holds what the launches
// share; flattened scenes, no work counters); blocks from path_grid_blocks (device must be current)
uint32_t path_grid_blocks(uint32_t n_local_pixels, const DeviceScene& scene);
uint32_t path_resident_blocks(const DeviceScene& scene);   // blocks of k_path the chip holds at once
hipError_t launch_path(hipStream_t st, const LaunchArgs& a, const PathBatch& batch, uint32_t blocks);
// scatter the tile-major cumulative / result images into full-frame row-major RGBA32F buffers
hipError_t launch_export(hipStream_t st, const TileMap& map, const float4* tiled, float4* frame, bool zero_first);
// chain `chain` of `n_chains` -> the rank's packed tile order (local tile j = jl * n_chains + chain), see Renderer::export_packed
hipError_t launch_pack_tiles(hipStream_t st, uint32_t n_chain_pixels, uint32_t n_chains, uint32_t chain, const float4* tiled, float4* packed);
// result (out32) -> RGBA8 sRGB, full-frame row-major (the blit of raytracer.rs:576-584)
// thresholds: 256 floats, [k] = smallest linear value that encodes to k (host::srgb8_thresholds); [0] = 0
hipError_t launch_tonemap(hipStream_t st, uint32_t n_pixels, const float4* result_frame, const float* thresholds, uchar4* out);

// debug / parity hooks
hipError_t launch_debug_closest(hipStream_t st, const DeviceScene& scene, const float* origins, const float* dirs, uint32_t n, float tmin,
                                float* t, uint32_t* tri, uint32_t* inst, float* u, float* v, uint32_t* overflow, uint32_t overflow_depth);
hipError_t launch_debug_any(hipStream_t st, const DeviceScene& scene, const float* origins, const float* dirs, const float* tmax, uint32_t n,
                            float tmin, uint8_t* hit, uint32_t* overflow, uint32_t overflow_depth);

}  // namespace glz
