// Host<->device PODs.  The RT* structs keep the byte layout of the reference's
// lib/src/vulkan/raytrace_structures.rs (mirrored into GLSL by lib/build.rs:138-184) so that the
// debug read-back hooks can be compared field by field; everything else is this build's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace glz {

struct Spectrum16 {
  float w[16];
};

// raytrace_structures.rs:4-15 (52 bytes in the reference; we add the integrator knobs behind it)
struct FrameData {
  uint32_t seed;
  uint32_t lights_no;
  float pixel_offset[2];
  float scene_radius;
  float exposure;
  float scene_size[2];
  float scene_centre[4];
  uint32_t camera_persp;
  // build-defined: PT_STEPS and the DIRECT_ONLY compile-time switch as run-time values
  uint32_t pt_steps;
  uint32_t direct_only;
  // build-defined texture level of detail (the reference's ray-tracing stages sample level 0): 0 = level 0 always,
  // 1 = ray cones, 2 = ray cones with an anisotropic footprint.  The cone of a camera path starts `cone_width0` wide and widens by `cone_spread` per unit of distance.
  uint32_t lod_mode;
  float cone_spread;
  float cone_width0;
  // FrameData::pixel_offset of the launch AFTER this one (host: WorkScheduler::peek) and whether the shading of this launch may use it:
  // a path that ends in this launch gets its next camera ray from the shading code (shade_pixel, `pregen`), in whole waves of pixels
  // that ended together, instead of from the traversal kernel's refill, which runs with a quarter of its lanes (ClosestSource::load)
  float next_pixel_offset[2];
  uint32_t pregen;
};

// raytrace_structures.rs:36-42
struct RTInstance {
  uint32_t index_offset;
  uint32_t index_count;
  uint32_t material_id;
  uint32_t transform_id;
};
static_assert(sizeof(RTInstance) == 16, "RTInstance is 16 bytes");

// raytrace_structures.rs:44-64
struct alignas(16) RTMaterial {
  float diffuse_mul[4];
  float emissive_col[4];
  Spectrum16 metal_ior;
  Spectrum16 metal_fresnel;
  uint32_t diffuse;
  uint32_t roughness;
  uint32_t metalness;
  uint32_t opacity;
  uint32_t normal;
  uint32_t bsdf_index;
  float roughness_mul;
  float metalness_mul;
  float anisotropy;
  float ior_dielectric;
  uint32_t is_specular;
  uint32_t is_emissive;
};
static_assert(sizeof(RTMaterial) == 208, "RTMaterial is 208 bytes");

// raytrace_structures.rs:66-76
struct alignas(16) RTLight {
  Spectrum16 color;
  float pos[4];
  float dir[4];
  uint32_t shader;
  uint32_t instance_id;
  float intensity;
  uint32_t delta;
};
static_assert(sizeof(RTLight) == 112, "RTLight is 112 bytes");

// raytrace_structures.rs:78-85
struct alignas(16) RTSky {
  float obj2world[16];
  float world2obj[16];
  uint32_t tex_id;
  float intensity;
  uint32_t _pad[2];
};
static_assert(sizeof(RTSky) == 144, "RTSky is 144 bytes");

// SBT callable indices (light.rs:111-119, material.rs:244-258)
enum : uint32_t {
  kLightOmni = 0, kLightSun = 1, kLightArea = 2, kLightSky = 3,
  kBsdfLambert = 4, kBsdfMirror = 6, kBsdfGlass = 8, kBsdfMetal = 10, kBsdfFrosted = 12, kBsdfUber = 14
};

// Texture descriptor: texels of all textures live in one byte pool.
struct TexDesc {
  uint32_t offset;    // byte offset of level 0 in the pool
  uint32_t width, height;
  uint32_t format;    // bits 0..6: GLZ_TEX_GRAY / RGBA_SRGB / RGBA_NORM; bit 7: 1x1 texture whose texel is in `offset`; bits 8..31: tiles per row (shading.h)
};

// Object<->world matrices of one transform: column-major mat4 (as uploaded by the reference,
// scene.rs load_transforms_to_gpu) and its inverse (what the driver exposes as gl_WorldToObjectEXT).
struct TransformPair {
  float o2w[16];
  float w2o[16];
};

// BVH4 node, 64 bytes = four dwordx4 fetches per visit.  Up to four child boxes, each quantised to 16 bits per coordinate
// on a global grid spanning the scene bounds (lo rounded down, hi rounded up, so the boxes only grow); the ray is mapped
// into grid units once.  The tree is the binary LBVH / PLOC hierarchy collapsed four wide (kernels_build.hip, collapse4): half
// the dependent fetches per ray, which is what bounds a launch once a GPU holds few rays per wave.
//   child k (k = 0..3):  w[3k] = lo.x | hi.x << 16   w[3k+1] = lo.y | hi.y << 16   w[3k+2] = lo.z | hi.z << 16
//   (one word per axis, so that a per-ray byte permutation puts the plane the ray meets first into the low half)
//   w[12 + k] = link of child k: index of an inner node (>= 0), ~index of a leaf in bvh_tris (< 0), or kBvhEmptyChild
//   for an unused slot (the tracer skips it by its link; its box words are lo = kBvhGridMax, hi = 0).
constexpr uint32_t kTriNonOpaque = 0x80000000u;   // BvhTri::prim_flags: candidates on this triangle go through the alpha test
constexpr uint32_t kTriHasPartner = 0x40000000u;  // the next triangle in bvh_tris belongs to the same leaf
constexpr uint32_t kTriPrimMask = 0x3FFFFFFFu;
constexpr float kPairAreaRatio = 0.85f;           // two triangles share a leaf when area(joint box) <= ratio * (area(a) + area(b)); 0.55 ... 2: 2146 ... 2193 ... 2184 Msamples/s
struct alignas(16) BvhNode4 {
  uint32_t w[16];
};
// EXPERIMENT (-DGLZ_NODE48, tools/build_variant_full.sh): the same node in 48 bytes = three dwordx4 fetches --
// child planes as 8-bit offsets from a per-node origin on the global grid, in cells of 2^e grid cells per axis (after Ylitie et al. 2017):
//   w[0] = origin.x | origin.y << 16      w[1] = origin.z | ex << 16 | ey << 20 | ez << 24
//   w[2 + 2 a + p] = planes of axis a for children 2 p and 2 p + 1: lo | hi << 8 | lo' << 16 | hi' << 24      w[8 + k] = link of child k
// Made from the 64-byte nodes after the build (k_compress_nodes: boxes only grow); the default build does not carry it.
struct alignas(16) BvhNode48 {
  uint32_t w[12];
};
// The same hierarchy collapsed EIGHT wide, 128 bytes = two lines = eight dwordx4 fetches per visit: what the tracer of a small tile share
// reads (k_trace8: a GPU that holds one 64-ray group per wave is bound by the latency of a ray's chain of dependent node fetches, and
// this chain is a third shorter -- DESIGN.md section 6).  Same grid, same padding:
//   child k (k = 0..7):  w[3k], w[3k+1], w[3k+2] as above      w[24 + k] = link of child k (leaf links are ~leaf number)
struct alignas(128) BvhNode8 {
  uint32_t w[32];
};
constexpr int kBvhEmptyChild = 0x7FFFFFFF;
// Top of the tree, staged in LDS by the tracers ("node packets"): the root, its inner children and their inner children in
// breadth-first order, at most kBvhTopNodes nodes (1 + 4 + 16).  Inside this table -- and in `cur` of a lane that sits on
// one of its nodes -- a link to a staged node reads kBvhTopFlag | slot instead of the node index; links that leave the
// table are the ordinary ones.  Built once per scene (k_top_table), copied into LDS at the start of every tracer block.
constexpr int kBvhTopNodes = 21;
constexpr int kBvhTopFlag = 0x40000000;
constexpr uint32_t kBvhGridMax = 32767u;   // 15-bit grid coordinates: 0x47000000 | q << 8 is the float 32768 + q (box_key)
struct BvhGrid {
  float lo[3];
  float cell[3];
  float inv_cell[3];
};

// Two-level structure (instanced scenes; acceleration.rs:319-345: one BLAS per mesh, a TLAS over the instances).  The bottom
// level is the same 4-wide hierarchy over a mesh's OBJECT-space triangles (nodes, triangles and shading records of all meshes
// concatenated; links are relative to the mesh's bases), the top level the same hierarchy over the instances' world boxes whose
// leaves are TlasInstance records in leaf order.  Entering an instance only re-derives the grid-space ray (the object-space
// ray is used for the box tests alone); the triangle test stays in WORLD space on the triangle transformed exactly as the
// flattened build transforms it (k_world_tris), so a two-level scene gives bit for bit the hits of its flattened twin.
struct alignas(16) TlasInstance {
  // what ENTERING the instance reads, in the record's first 96 bytes (a line and a half; the rest is only read at the mesh's leaves)
  float w2o[12];          // rows 0..2 of world -> object (x' = w2o[0..3] . (x, 1), ...): the ray into object space
  BvhGrid grid;           // quantisation grid of the mesh's nodes (object space)
  float slack;            // slack of the object-space box tests in object units (rounding of the transformed ray and triangle); every box is widened by slack / cell + 1 cells per axis
  uint32_t node_base;     // first node of the mesh in bvh_nodes
  float w2o_norm;         // max row sum of |w2o|'s 3 x 3 part: how the rounding of a ray origin grows into object space (per-ray part of the slack)
  // what a triangle test inside the instance reads
  float o2w[16];          // object -> world, column-major as TransformPair::o2w: the triangle into world space
  uint32_t tri_base;      // first triangle / shading record of the mesh in bvh_tris / shade_tris
  uint32_t world_base;    // world triangle id of the instance's primitive 0 (tie-break key, as in the flattened build)
  uint32_t instance;      // RTInstance index
  uint32_t non_opaque;    // the instance's material has an opacity map (acceleration.rs:136-141)
  uint32_t quad_base;     // first leaf record of the mesh in bvh_quads (the meshes' leaf links are ~leaf number, relative to this)
  uint32_t _pad[3];
};
static_assert(sizeof(TlasInstance) == 192, "TlasInstance is 12 x 16 bytes");

// World-space triangle in BVH leaf order, 48 bytes (36 algorithmic + ids).  The three VERTICES, not a vertex and two edges:
// the watertight test (ray_triangle) needs the floats two triangles share to be the same floats in both records.
struct alignas(16) BvhTri {
  float v0[3]; uint32_t world_id;   // instance-major id: tie-break key for equal t
  float v1[3]; uint32_t instance;
  float v2[3]; uint32_t prim_flags; // kTriNonOpaque | kTriHasPartner | primitive in instance
};
static_assert(sizeof(BvhTri) == 48, "BvhTri is 48 bytes");

// What the flattened tracer reads at a leaf: ONE 64-byte, line-aligned record per leaf, indexed by the leaf's number (leaf links of the
// flattened build are ~leaf number; bvh_tris / shade_tris stay indexed by triangle slot, which the record names).  A leaf of two
// triangles is a quad: triangle A = (q0, q1, q2), triangle B = (q0, q2, q3) -- each in ITS OWN vertex order, so the watertight test
// computes for each exactly what it computes from the 48-byte records (the builder pairs two triangles only when one of the two
// orders of the pair has this shape, k_pair_triangles); the four vertices are sheared once and the edge q0-q2 they share is
// evaluated once: B's edge function along it is exactly the negation of A's (the products commute, tie-break terms included).
// A single triangle is A alone.  kQuadSwapped: A is the pair's SECOND triangle in slot / id order (B the first).
constexpr uint32_t kQuadSwapped = 0x20000000u;   // in prim_flags, next to kTriNonOpaque / kTriHasPartner; the primitive number keeps 29 bits
constexpr uint32_t kQuadPrimMask = 0x1FFFFFFFu;
struct alignas(64) BvhQuad {
  float q0[3]; uint32_t world_id;    // of the pair's first triangle (the second one's is + 1)
  float q1[3]; uint32_t instance;
  float q2[3]; uint32_t prim_flags;  // kTriNonOpaque | kTriHasPartner | kQuadSwapped | primitive of the pair's first triangle
  float q3[3]; uint32_t slot;        // first triangle slot of the leaf in bvh_tris / shade_tris
};
static_assert(sizeof(BvhQuad) == 64, "BvhQuad is 64 bytes");

// Sky table header following RTSky in the reference SSBO (light_sky_sample_visible.rcall:19-26)
struct SkyHeader {
  uint32_t marginal_cdf_count;
  uint32_t conditional_integral_offset;
  uint32_t conditional_cdf_count;
  float marginal_integral;
};

// Everything the kernels need about a scene (device pointers).
struct DeviceScene {
  const float4* vertices;          // 2 x float4 per vertex (VertexPacked, raytrace_commons.glsl:11-14)
  const uint32_t* indices;
  const RTInstance* instances;
  const RTMaterial* materials;
  const RTLight* lights;
  const TransformPair* transforms;
  const float4* derivatives;       // 3 x float4 per object triangle (normal, dpdu, dpdv)
  const TexDesc* tex_desc;
  const uint8_t* tex_pool;
  // mip levels 1.. (built on demand, Scene::ensure_mips; null until a renderer asks for texture LOD): one TexDesc per level in
  // tex_mip_desc, offsets into tex_mip_pool; tex_mip_base[id] = index of the texture's level-1 descriptor | level count << 24
  const TexDesc* tex_mip_desc;
  const uint32_t* tex_mip_base;
  const uint8_t* tex_mip_pool;
  const float* srgb_lut;           // 256 entries
  const BvhNode4* bvh_nodes;
  const BvhNode4* bvh_top;         // kBvhTopNodes nodes: the top levels with links into the table flagged (kBvhTopFlag)
  const BvhNode8* bvh_nodes8;      // the 8-wide collapse of the same hierarchy (flattened builds of more than one leaf; null otherwise)
  const BvhNode48* bvh_nodes48;    // -DGLZ_NODE48 builds only (null otherwise): the same nodes / staged top in the 48-byte format
  const BvhNode48* bvh_top48;
  BvhGrid bvh_grid;
  const BvhTri* bvh_tris;
  const BvhQuad* bvh_quads;        // one 64-byte record per leaf, what the tracers read: world space (flattened build) or the meshes' object space, concatenated (two levels)
  // per-leaf shading record, 8 x float4 = 128 bytes, in leaf order: VertexPacked x 3 (object space), then
  // (geometric normal.xyz, material id), (dpdu.xyz, transform id | identity flag in bit 31)
  const float4* shade_tris;
  // Flattened scenes with an opacity map (null otherwise): per triangle slot 3 x float4 = (uv0, uv1), (uv2, TexDesc::offset, ::width),
  // (::height, ::format, 0, 0) of the material's opacity map -- what the alpha test of a candidate needs (raytrace_hit.rahit:24-39) in
  // ONE line and one round trip instead of shading record -> RTMaterial -> descriptor, three dependent ones (Scene::build_alpha_records;
  // rebuilt whenever materials or textures change)
  const float4* alpha_recs;
  // two-level scenes (null / 0 otherwise): TLAS nodes, instance records in TLAS leaf order; bvh_nodes / bvh_tris / shade_tris
  // then hold the meshes' object-space hierarchies and per-OBJECT-triangle records
  const BvhNode4* tlas_nodes;
  const TlasInstance* tlas_instances;
  const uint32_t* xf_identity;     // per transform: 1 = exactly the identity
  uint32_t two_level;
  uint32_t has_non_opaque;         // some material carries an opacity map: candidates on its triangles go through the alpha test (kTriNonOpaque in the leaf records)
  uint32_t n_world_tris;
  uint32_t n_textures;
  uint32_t n_materials;            // RTMaterial records
  uint32_t n_rt_lights;            // RTLight records (area lights expand to one per instance)
  // sky
  RTSky sky;
  SkyHeader sky_header;
  const float* sky_marginal;       // cdf (H+1) | values (H) | conditional integrals (H)
  const float* sky_cdf;            // the H + 1 cdf entries the row search walks: sky_marginal, or k_shade's LDS copy of them
  const float* sky_cond_values;    // W x H
  const float* sky_cond_cdf;       // (W+1) x H
  // counting builds only (null otherwise; the host never sets it): the THREAD's tallies {texture fetches that read memory, their texel
  // bytes, light samples, sky-light samples} -- the instrumented kernels point their copy of the scene at four registers and add
  // them up per wave at their end (TraceCounters::shade_tex / trace_tex): what bench.py books for texels and light tables
  unsigned long long* tex_counter;
  uint32_t sky_w, sky_h;
};

}  // namespace glz
