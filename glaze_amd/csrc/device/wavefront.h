// Device code shared by the render kernels (kernels_render.hip: k_trace, k_shade, k_trace_tl; kernels_path.hip: k_path): pixel
// mapping, the slab and the watertight triangle test, the wave-persistent tracers, ray sources / sinks, and shade_pixel.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device/math.h"
#include "device/shading.h"
#include "device/tuning.h"
#include "device/types.h"
#include "kernels.h"

namespace glz {
using namespace dev;

constexpr int kBlock = (int)kTraceBlock;   // 4 waves
constexpr int kLdsStack = kTraversalLdsStack;   // stack entries kept in LDS per lane (17 levels: 17.4 KB of the 25.3 KB a k_trace block takes -> 6 blocks per CU); deeper levels spill to HBM
constexpr uint32_t kQueueShards = 8;     // shadow-ray sub-queues (see queue_slot)
constexpr uint32_t kCounterStride = 32;  // uint32 words between shard counters (128 bytes)
constexpr uint32_t kFlagUpdate = 1u;    // update_result() is called for this pixel in this launch
constexpr uint32_t kFlagShadow = 2u;    // the contribution is gated by a shadow ray
constexpr uint32_t kFlagPoison = 4u;    // 0 * (|cos|/pdf) * radiance is NaN: an occluded sample still poisons the pixel

// ---------------------------------------------------------------------------------------------
// pixel <-> thread mapping
// ---------------------------------------------------------------------------------------------
struct PixelId {
  uint32_t x, y;
  bool active;
};
__device__ __forceinline__ PixelId pixel_of(const TileMap& m, uint32_t lid) {
  const uint32_t lane = lid & 63u, sub = (lid >> 6) & 63u, ltile = lid >> 12;
  const uint32_t gtile = ltile * m.world + m.rank;
  const uint32_t tx = gtile % m.tiles_x, ty = gtile / m.tiles_x;
  PixelId p;
  p.x = tx * 64u + (sub & 7u) * 8u + (lane & 7u);
  p.y = ty * 64u + (sub >> 3) * 8u + (lane >> 3);
  p.active = lid < m.n_local_pixels && p.x < m.width && p.y < m.height;
  return p;
}

// ---------------------------------------------------------------------------------------------
// Ray / box and ray / triangle.  The triangle test (ray_triangle below) is watertight, accepts a candidate iff tmin < t < tmax and
// culls no face (acceleration.rs:335-345); it stands in for the driver's intersector ([ext]).  Ties on t are broken by the smaller
// world triangle id so that the result does not depend on traversal order.
// ---------------------------------------------------------------------------------------------
// Slab test on a quantised box.  It only prunes: boxes are padded by 1/16 cell when they are quantised, which covers the
// rounding of the plane distances (< 0.01 cell), so it never rejects a box whose triangle the exact test below accepts.  The ray
// is mapped into grid units once (ig = cell / d, the addend grid_addend); a node word holds lo | hi << 16 of one axis, and a per-ray
// byte permutation picks the plane the ray meets first and the other one -- no min / max per axis.  Both plane distances of an axis
// then come from one packed v_pk_fma_f32.  The tracers are VALU-issue bound, so instructions per node visit are what counts.  ig is
// kept finite (grid_inv_dir), so no plane distance is ever NaN: a ray parallel to a slab gets +-1e30-scale distances whose signs
// still say on which side of each plane the origin lies.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SlabSel { uint32_t x, y, z; };   // v_perm_b32 selectors per axis
// returns the sort key of the child: entry distance (a positive float, so its bits order like the value) with its ten lowest bits
// replaced by the child index in bits 8..9 (`k` = child << 8) -- nearer first, ties (to 2^-13 of the distance) by child index; 0xFFFFFFFF
// for a missed child or an unused slot.  Bits 8..9 because (key & 0x300) IS the byte offset of the child's link in the wave's link
// scratch ([child][lane], 256 bytes a child, the area 1 KB aligned): the address of a sorted link is one v_and_or_b32 (it was three
// instructions per link with the index in the low bits)
// (the slab test is symmetric in lo / hi, so an unused slot cannot be excluded through its box: its link says so)
// cgn / cgf: the addends of the near and the far plane.  The flattened tracer passes the same vector twice; the two-level tracer
// widens every box by the instance's slack (cg -+ pad * |ig|) at no extra instruction.
// Grid coordinates are 15 bits wide, so a byte permute turns one into a float without a conversion: the bytes {0, q_lo, q_hi, 0x47}
// are 0x47000000 | q << 8, the float 32768 + q (exponent 2^15, q in the mantissa's bits 22..8).  The 32768 is folded into the addend
// of the plane distance (grid_ray: cg - 32768 ig), the selector names which half of the node word -- the plane the ray meets first
// (lo for ig >= 0, hi for ig < 0) or the other one: kSlabSelLo / kSlabSelHi, one XOR apart.  Two permutes per axis instead of one
// permute and two v_cvt_f32_u32: 9 VALU instructions fewer per node visit (rounds 1-2 needed six live selectors for this and spilt).
constexpr uint32_t kKeyDistanceMask = 0xFFFFFC00u, kKeyChild = 0x100u, kKeyChildMask = 0x300u;   // sort key = distance bits | child << 8
constexpr uint32_t kSlabSelLo = 0x0305040Cu, kSlabSelHi = 0x0307060Cu, kSlabSelFlip = kSlabSelLo ^ kSlabSelHi, kSlabMagic = 0x47000000u;
__device__ __forceinline__ uint32_t slab_sel(float ig) { return ig < 0.0f ? kSlabSelHi : kSlabSelLo; }
// CHECK_LINK = false: an unused slot is excluded by its box alone -- lo = the grid's top, hi = 0 on every axis, and with the near /
// far plane picked by the ray's sign such a box has t_near > t_far on every axis whatever the ray (the planes are 32 767 cells the
// wrong way round; distances are never NaN) -- four compares fewer per node visit.  The two-level tracer widens boxes by a per-ray
// pad that may exceed that in extreme cases and keeps the check.
template <bool CHECK_LINK = true>
__device__ __forceinline__ uint32_t box_key(uint32_t wx, uint32_t wy, uint32_t wz, uint32_t link, uint32_t k, SlabSel sel, vec3 ig, vec3 cgn, vec3 cgf,
                                            float tmin, float tmax) {
  // v_perm_b32: bytes 0..3 of the selector index the second operand, 4..7 the first (the node word), 0x0C is a zero byte
  const float nx = __uint_as_float(__builtin_amdgcn_perm(wx, kSlabMagic, sel.x)), fx = __uint_as_float(__builtin_amdgcn_perm(wx, kSlabMagic, sel.x ^ kSlabSelFlip));
  const float ny = __uint_as_float(__builtin_amdgcn_perm(wy, kSlabMagic, sel.y)), fy = __uint_as_float(__builtin_amdgcn_perm(wy, kSlabMagic, sel.y ^ kSlabSelFlip));
  const float nz = __uint_as_float(__builtin_amdgcn_perm(wz, kSlabMagic, sel.z)), fz = __uint_as_float(__builtin_amdgcn_perm(wz, kSlabMagic, sel.z ^ kSlabSelFlip));
  const f32x2 tx = __builtin_elementwise_fma(f32x2{nx, fx}, f32x2{ig.x, ig.x}, f32x2{cgn.x, cgf.x});
  const f32x2 ty = __builtin_elementwise_fma(f32x2{ny, fy}, f32x2{ig.y, ig.y}, f32x2{cgn.y, cgf.y});
  const f32x2 tz = __builtin_elementwise_fma(f32x2{nz, fz}, f32x2{ig.z, ig.z}, f32x2{cgn.z, cgf.z});
  const float t0 = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
  const float t1 = fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax));
  return (t0 <= t1 && (!CHECK_LINK || link != (uint32_t)kBvhEmptyChild)) ? ((__float_as_uint(t0) & kKeyDistanceMask) | k) : 0xFFFFFFFFu;
}
// The same test for a child of an 8-wide node (types.h BvhNode8): the child index takes three bits of the key (8..10)
constexpr uint32_t kKey8DistanceMask = 0xFFFFF800u, kKey8ChildMask = 0x700u;
__device__ __forceinline__ uint32_t box_key8(uint32_t wx, uint32_t wy, uint32_t wz, uint32_t k, SlabSel sel, vec3 ig, vec3 cg, float tmin, float tmax) {
  const float nx = __uint_as_float(__builtin_amdgcn_perm(wx, kSlabMagic, sel.x)), fx = __uint_as_float(__builtin_amdgcn_perm(wx, kSlabMagic, sel.x ^ kSlabSelFlip));
  const float ny = __uint_as_float(__builtin_amdgcn_perm(wy, kSlabMagic, sel.y)), fy = __uint_as_float(__builtin_amdgcn_perm(wy, kSlabMagic, sel.y ^ kSlabSelFlip));
  const float nz = __uint_as_float(__builtin_amdgcn_perm(wz, kSlabMagic, sel.z)), fz = __uint_as_float(__builtin_amdgcn_perm(wz, kSlabMagic, sel.z ^ kSlabSelFlip));
  const f32x2 tx = __builtin_elementwise_fma(f32x2{nx, fx}, f32x2{ig.x, ig.x}, f32x2{cg.x, cg.x});
  const f32x2 ty = __builtin_elementwise_fma(f32x2{ny, fy}, f32x2{ig.y, ig.y}, f32x2{cg.y, cg.y});
  const f32x2 tz = __builtin_elementwise_fma(f32x2{nz, fz}, f32x2{ig.z, ig.z}, f32x2{cg.z, cg.z});
  const float t0 = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
  const float t1 = fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax));
  return t0 <= t1 ? ((__float_as_uint(t0) & kKey8DistanceMask) | k) : 0xFFFFFFFFu;   // (an unused slot's box is inverted: no ray enters it)
}
#ifdef GLZ_NODE48
// EXPERIMENT: the slab test on a 48-byte node's child (types.h BvhNode48).  `word` holds the planes of one axis for two children as bytes
// lo | hi << 8 | lo' << 16 | hi' << 24; the selector puts one of them into bits 8..15 of 0x47000000 -- the float 32768 + q -- and the
// node-local ray (a = ig 2^e, c = (origin - ray origin) ig - 32768 a, made once per visit) turns it into a distance with one fma.
constexpr uint32_t kSel48Lo = 0x030C040Cu, kSel48Hi = 0x030C050Cu;   // byte 0 / byte 1 of the word -> byte 1 of the float; ^ 0x0200: the odd child's
__device__ __forceinline__ uint32_t box_key48(uint32_t wx, uint32_t wy, uint32_t wz, uint32_t k, uint32_t odd, SlabSel sel, vec3 a, vec3 c, float tmin, float tmax) {
  const uint32_t sx = sel.x ^ odd, sy = sel.y ^ odd, sz = sel.z ^ odd;
  const float nx = __uint_as_float(__builtin_amdgcn_perm(wx, kSlabMagic, sx)), fx = __uint_as_float(__builtin_amdgcn_perm(wx, kSlabMagic, sx ^ 0x0100u));
  const float ny = __uint_as_float(__builtin_amdgcn_perm(wy, kSlabMagic, sy)), fy = __uint_as_float(__builtin_amdgcn_perm(wy, kSlabMagic, sy ^ 0x0100u));
  const float nz = __uint_as_float(__builtin_amdgcn_perm(wz, kSlabMagic, sz)), fz = __uint_as_float(__builtin_amdgcn_perm(wz, kSlabMagic, sz ^ 0x0100u));
  const f32x2 tx = __builtin_elementwise_fma(f32x2{nx, fx}, f32x2{a.x, a.x}, f32x2{c.x, c.x});
  const f32x2 ty = __builtin_elementwise_fma(f32x2{ny, fy}, f32x2{a.y, a.y}, f32x2{c.y, c.y});
  const f32x2 tz = __builtin_elementwise_fma(f32x2{nz, fz}, f32x2{a.z, a.z}, f32x2{c.z, c.z});
  const float t0 = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, tmin));
  const float t1 = fminf(fminf(tx.y, ty.y), fminf(tz.y, tmax));
  return t0 <= t1 ? ((__float_as_uint(t0) & kKeyDistanceMask) | k) : 0xFFFFFFFFu;
}
#endif
// the addend of a plane distance: plane q (a float 32768 + q out of box_key) is crossed at t = (32768 + q) ig + grid_addend = q ig - og ig
__device__ __forceinline__ float grid_addend(float og, float ig) { return fmaf(-32768.0f, ig, -(og * ig)); }

// 1 / d clamped to +-1e30: zero (or denormal) direction components must not produce inf - inf in the fma above --
// a ray with a NaN plane distance on every axis would pass every box test and walk the whole tree.
__device__ __forceinline__ float grid_inv_dir(float d) {
  const float i = 1.0f / d;
  return fabsf(i) <= 1e30f ? i : copysignf(1e30f, i);
}
// The same from the hardware's reciprocal (v_rcp_f32, one ulp) instead of the correctly rounded division (ten instructions): for the
// flattened tracer's refill, which runs with a quarter of the wave's lanes.  The grid-space ray only PRUNES -- hits come from the exact
// test on the world ray -- and an error of 2^-23 in ig moves a plane crossing by at most 32 768 cells x 2^-23 = 0.004 cell, inside the
// 1/16 cell the boxes are padded by (the rounding of the plane distances themselves takes 0.01).  The counting kernels keep the
// division: their node counts are compared with the oracle's walk.
__device__ __forceinline__ float grid_inv_dir_fast(float d) {
  const float i = __builtin_amdgcn_rcpf(d);
  return fabsf(i) <= 1e30f ? i : copysignf(1e30f, i);
}
// rays with a NaN / infinite origin or direction cannot be accepted by ray_triangle (every comparison fails): they
// are reported as misses without traversal
__device__ __forceinline__ bool ray_is_finite(vec3 o, vec3 d) {
  const float s = ((o.x + o.y) + o.z) + ((d.x + d.y) + d.z);
  return s - s == 0.0f;
}

// The watertight ray / triangle test, statement for statement the oracle's ray_tri (oracle.cpp; the reference's hits come from
// traceRayEXT on the driver's acceleration structure, path_trace.rgen:169 / acceleration.rs:319-345, which the Vulkan
// specification requires to be watertight): Woop, Benthin, Wald 2013 with the exact tie-break in single precision.
//   per ray    kz = axis of the largest |d|, shear Sz = 1 / d[kz], Sx = d[kx] Sz, Sy = d[ky] Sz        (ray_shear; once per leaf round,
//              from d alone -- nothing is kept per ray, the traversal has no register to spare)
//   per vertex A = P - o, image (A[kx] - Sx A[kz], A[ky] - Sy A[kz], Sz A[kz]): the same 2-D point in every triangle that uses P
//   per edge   U = Cx By - Cy Bx from two separately rounded products: its sign is exact unless the rounded products are equal, and
//              then the difference of their rounding errors (one fma each) is.  Exact orientation predicates on consistent points
//              cannot leave a gap at a shared edge or vertex.  -ffp-contract=off keeps the products unfused.
// Straight-line: with ~10 of 64 lanes in a leaf round an early exit is almost never taken by all of them, and without branches
// the three 16-byte loads of the triangle are issued together; only the tie-break is a (wave-uniform) branch, taken when some
// lane's ray meets an edge exactly -- axis-aligned geometry under an orthographic camera, otherwise hardly ever.
struct RayShear {
  bool z_is_x, z_is_y;   // kz == 0, kz == 1 (else 2): wave masks in SGPRs
  float sx, sy, sz;
};
__device__ __forceinline__ RayShear ray_shear(vec3 d) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  RayShear r;
  r.z_is_x = (ax >= ay) & (ax >= az);
  r.z_is_y = !r.z_is_x & (ay >= az);
  // (kx, ky, kz) = (1, 2, 0), (2, 0, 1) or (0, 1, 2)
  const float dz = r.z_is_x ? d.x : (r.z_is_y ? d.y : d.z), dx = r.z_is_x ? d.y : (r.z_is_y ? d.z : d.x), dy = r.z_is_x ? d.z : (r.z_is_y ? d.x : d.y);
  r.sz = 1.0f / dz;
  r.sx = dx * r.sz;
  r.sy = dy * r.sz;
  return r;
}
__device__ __forceinline__ vec3 shear_vertex(const RayShear& r, const float* p, vec3 o) {
  const vec3 a = mk3(p[0], p[1], p[2]) - o;
  const float az = r.z_is_x ? a.x : (r.z_is_y ? a.y : a.z), ax = r.z_is_x ? a.y : (r.z_is_y ? a.z : a.x), ay = r.z_is_x ? a.z : (r.z_is_y ? a.x : a.y);
  return mk3(fmaf(-r.sx, az, ax), fmaf(-r.sy, az, ay), r.sz * az);
}
constexpr float kDetNoise = 1.9073486e-6f;   // 2^-19 of three of the six products (about 2^-20 of their sum): see triangle_finish
__device__ __forceinline__ bool ray_triangle(const RayShear& rs, const BvhTri& tr, vec3 o, float tmin, float& t, float& u, float& v) {
  const vec3 A = shear_vertex(rs, tr.v0, o), B = shear_vertex(rs, tr.v1, o), C = shear_vertex(rs, tr.v2, o);
  const float pu = C.x * B.y, qu = C.y * B.x, pv = A.x * C.y, qv = A.y * C.x, pw = B.x * A.y, qw = B.y * A.x;
  float U = pu - qu, V = pv - qv, W = pw - qw;
  if (__builtin_expect(__any((U == 0.0f) | (V == 0.0f) | (W == 0.0f)), 0)) {   // edge_fn's second branch, for the lanes that need it
    if (U == 0.0f) U = fmaf(C.x, B.y, -pu) - fmaf(C.y, B.x, -qu);
    if (V == 0.0f) V = fmaf(A.x, C.y, -pv) - fmaf(A.y, C.x, -qv);
    if (W == 0.0f) W = fmaf(B.x, A.y, -pw) - fmaf(B.y, A.x, -qw);
  }
  const float lo = fminf(fminf(U, V), W), hi = fmaxf(fmaxf(U, V), W);   // two opposite signs <=> lo < 0 < hi (a NaN fails the distance test)
  const float det = (U + V) + W;
  const float inv = 1.0f / det;
  u = V * inv;
  v = W * inv;
  t = fmaf(W, C.z, fmaf(V, B.z, U * A.z)) * inv;
  const float noise = ((fabsf(pu) + fabsf(pv)) + fabsf(pw)) * kDetNoise;   // (triangle_finish)
  return !((lo < 0.0f) & (hi > 0.0f)) & (fabsf(det) > noise) & (t > tmin);
}

// The same test on a leaf record of the flattened build (types.h BvhQuad): triangle A = (q0, q1, q2) and, for a leaf of two,
// B = (q0, q2, q3).  Per triangle the operations and their order are ray_triangle's, so (t, u, v) and the verdict are bit for bit
// what the 48-byte records give; what the record saves is work the two triangles share: four vertex images instead of six, and the
// two products of the edge q0-q2 -- B's edge function along it is A's with the operands of the subtraction exchanged (the products
// themselves commute), tie-break terms included.  Straight-line like ray_triangle: the four 16-byte loads of a leaf issue together.
struct QuadHit {
  float t[2], u[2], v[2];   // [0] = A, [1] = B
  bool ok[2];
};
__device__ __forceinline__ vec3 shear_point(const RayShear& r, float px, float py, float pz, vec3 o) {
  const vec3 a = mk3(px, py, pz) - o;
  const float az = r.z_is_x ? a.x : (r.z_is_y ? a.y : a.z), ax = r.z_is_x ? a.y : (r.z_is_y ? a.z : a.x), ay = r.z_is_x ? a.z : (r.z_is_y ? a.x : a.y);
  return mk3(fmaf(-r.sx, az, ax), fmaf(-r.sy, az, ay), r.sz * az);
}
// the part of ray_triangle behind the edge functions.  `products` = |first product of U| + |of V| + |of W|: a det smaller than kDetNoise
// of it is the products' rounding, not a number -- the ray lies in the triangle's plane as far as single precision can tell (a shadow
// ray towards a light in the plane of the surface it leaves), and the distance that would come out of it is anything.  The oracle's
// ray_tri has the same line; oracle.cpp says what it was found by.
__device__ __forceinline__ bool triangle_finish(float U, float V, float W, float Az, float Bz, float Cz, float products, float tmin, float& t, float& u, float& v) {
  const float lo = fminf(fminf(U, V), W), hi = fmaxf(fmaxf(U, V), W);
  const float det = (U + V) + W;
  const float inv = 1.0f / det;
  u = V * inv;
  v = W * inv;
  t = fmaf(W, Cz, fmaf(V, Bz, U * Az)) * inv;
  return !((lo < 0.0f) & (hi > 0.0f)) & (fabsf(det) > products * kDetNoise) & (t > tmin);
}
__device__ __forceinline__ QuadHit ray_quad(const RayShear& rs, float4 r0, float4 r1, float4 r2, float4 r3, bool pair, vec3 o, float tmin) {
  QuadHit h;
  const vec3 S0 = shear_point(rs, r0.x, r0.y, r0.z, o), S2 = shear_point(rs, r2.x, r2.y, r2.z, o);
  const float pv = S0.x * S2.y, qv = S0.y * S2.x;   // the shared edge q0-q2: V of A, W of B
  {
    // A = (S0, S1, S2):  U = C x B, V = A x C, W = B x A  with  A = S0, B = S1, C = S2
    const vec3 S1 = shear_point(rs, r1.x, r1.y, r1.z, o);
    const float pu = S2.x * S1.y, qu = S2.y * S1.x, pw = S1.x * S0.y, qw = S1.y * S0.x;
    float U = pu - qu, V = pv - qv, W = pw - qw;
    if (__builtin_expect(__any((U == 0.0f) | (V == 0.0f) | (W == 0.0f)), 0)) {   // edge_fn's second branch, for the lanes that need it
      if (U == 0.0f) U = fmaf(S2.x, S1.y, -pu) - fmaf(S2.y, S1.x, -qu);
      if (V == 0.0f) V = fmaf(S0.x, S2.y, -pv) - fmaf(S0.y, S2.x, -qv);
      if (W == 0.0f) W = fmaf(S1.x, S0.y, -pw) - fmaf(S1.y, S0.x, -qw);
    }
    h.ok[0] = triangle_finish(U, V, W, S0.z, S1.z, S2.z, (fabsf(pu) + fabsf(pv)) + fabsf(pw), tmin, h.t[0], h.u[0], h.v[0]);
  }
  __builtin_amdgcn_sched_barrier(0);   // A is finished before B begins: interleaved for ILP the two keep twice the values alive (20 registers spilt)
  {
    // B = (S0, S2, S3):  U = S3 x S2, V = S0 x S3, W = S2 x S0 = S2.x S0.y - S2.y S0.x = qv - pv
    const vec3 S3 = shear_point(rs, r3.x, r3.y, r3.z, o);
    const float pu = S3.x * S2.y, qu = S3.y * S2.x, pv2 = S0.x * S3.y, qv2 = S0.y * S3.x;
    float U = pu - qu, V = pv2 - qv2, W = qv - pv;
    if (__builtin_expect(__any(pair & ((U == 0.0f) | (V == 0.0f) | (W == 0.0f))), 0)) {
      if (U == 0.0f) U = fmaf(S3.x, S2.y, -pu) - fmaf(S3.y, S2.x, -qu);
      if (V == 0.0f) V = fmaf(S0.x, S3.y, -pv2) - fmaf(S0.y, S3.x, -qv2);
      if (W == 0.0f) W = fmaf(S2.x, S0.y, -qv) - fmaf(S2.y, S0.x, -pv);
    }
    h.ok[1] = triangle_finish(U, V, W, S0.z, S2.z, S3.z, (fabsf(pu) + fabsf(pv2)) + fabsf(qv), tmin, h.t[1], h.u[1], h.v[1]) & pair;
  }
  return h;
}

// raytrace_hit.rahit:24-39 -- candidates on non-opaque geometry are dropped when opacity.r < 0.5
__device__ __forceinline__ bool alpha_test(const DeviceScene& S, uint32_t leaf, float u, float v) {
#ifdef GLZ_ALPHA_TIMING_NOFETCH   // TIMING ONLY: the verdict from the barycentrics alone (about half pass), no fetch
  return u + v < 0.5f;
#endif
  // One 48-byte record per triangle slot (types.h DeviceScene::alpha_recs; every flattened scene with an opacity map has them): the
  // three texture coordinates -- the values the reference's any-hit shader reads through instance -> indices -> vertices -- and the
  // descriptor of the material's opacity map: record -> texels, two round trips where shading record -> material -> descriptor ->
  // texels were four (a wave sits through them with the dozen lanes of an alpha phase: 0.07 ms of the Sponza-like atrium's k_trace).
  const float w = 1.0f - u - v;
  const float4* rec = S.alpha_recs + 3u * (size_t)leaf;
  const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
  const float tu = (r0.x * w + r0.z * u) + r1.x * v, tv = (r0.y * w + r0.w * u) + r1.y * v;
  const TexDesc t{__float_as_uint(r1.z), __float_as_uint(r1.w), __float_as_uint(r2.x), __float_as_uint(r2.y)};
  return !(bilinear_level(S, t, S.tex_pool, tu, tv).x < 0.5f);
}

// the same for a two-level scene: the shading record is per OBJECT triangle, the material is the instance's
__device__ __forceinline__ bool alpha_test_instance(const DeviceScene& S, uint32_t slot, uint32_t instance, float u, float v) {
  const float4* rec = S.shade_tris + 8u * (size_t)slot;
  const float4 a = rec[1], b = rec[3], c = rec[5];
  const uint32_t material_id = S.instances[instance].material_id;
  const float w = 1.0f - u - v;
  const float tu = (a.z * w + b.z * u) + c.z * v, tv = (a.w * w + b.w * u) + c.w * v;
  return !(texture_r(S, S.materials[material_id].opacity, vec2{tu, tv}) < 0.5f);
}

struct HitRecord {
  float t, u, v;
  uint32_t leaf;   // index into bvh_tris / shade_tris, 0xFFFFFFFF = miss
  // two-level scenes only (a flattened triangle record names its instance itself):
  uint32_t inst;       // RTInstance of the hit
  uint32_t world_id;   // world triangle id (instance-major), the tie-break key
};

// Per-lane traversal stack: the first kLdsStack levels in LDS (column `tid` of a [level][kBlock]
// array, accessed with 4-byte DS instructions, which gfx950 services in two 32-lane halves with bank = (addr / 4) % 32
// -- the 64-bank mapping only applies to the 8- and 16-byte reads: every lane hits bank tid % 32 of its own half,
// conflict-free whatever the per-lane depth), deeper
// levels in a per-lane HBM spill area.  kStolen marks an LDS entry that was handed to an idle lane (work sharing
// at the tail of trace_wave); pop_live() skips such entries.
constexpr int kRayDone = 0x7FFFFFFF;   // `cur` of a lane without a node to visit (inner nodes are >= 0, leaves < 0)
constexpr int kStolen = 0x7FFFFFFE;
// The LDS column is addressed through a pointer that KEEPS its address space: with a generic pointer the compiler turned pop() -- LDS
// level or spilt level -- into ONE flat_load_dword behind a pointer select, i.e. every pop of the traversal went through the flat path
// and waited for vmcnt(0) AND lgkmcnt(0) (with it all of the lane's loads in flight): the largest single piece of a node iteration
// (tools/gpu_sections.py, round 4: ~1 000 of ~2 400 clocks on an otherwise idle chip).  The staged top of the tree had the same
// problem in round 2 (LdsNodePtr).
typedef __attribute__((address_space(3))) int* LdsIntPtr;
template <int kLevels>
struct StackT {
  LdsIntPtr lds;        // &s_stack[threadIdx.x]
  uint32_t* spill;      // overflow words of this lane
  int sp;
  __device__ __forceinline__ StackT(int* lds_column, uint32_t* spill_words, int sp0) : lds((LdsIntPtr)lds_column), spill(spill_words), sp(sp0) {}
  __device__ __forceinline__ void push(int v) {
    if (sp < kLevels) lds[sp * kBlock] = v; else spill[sp - kLevels] = (uint32_t)v;
    ++sp;
  }
  __device__ __forceinline__ int pop() {
    --sp;
    int v;
    if (__builtin_expect(sp < kLevels, 1)) v = lds[sp * kBlock]; else v = (int)spill[sp - kLevels];   // two loads of two address spaces: not to be merged
    return v;
  }
  // Hand-overs take the OLDEST live entry of a stack (the lowest level, aux_sb) and move that mark up by one, so the stolen entries are
  // one run at the bottom: a pop that finds kStolen has found the end of the lane's own work -- everything below is stolen as well.
  // (Walking down through the marks one LDS read at a time, as rounds 1-3 did, was most of the 1 100 clocks a node iteration of a small
  // share spent behind its box tests: the whole wave waits while one lane scans.)
  __device__ __forceinline__ int pop_live() {
    if (sp > 0) {
      const int v = pop();
      if (v != kStolen) return v;
      sp = 0;
    }
    return kRayDone;
  }
};
using Stack = StackT<kLdsStack>;
constexpr int kLdsStack8 = GLZ_TRACE8_STACK;   // LDS levels of the 8-wide tracer's stacks (a visit pushes up to seven; its blocks run four to a CU: 28 KB + 8 KB of scratch each)

struct TraceTally {
  unsigned long long rays = 0, nodes = 0, tris = 0, hits = 0, fresh = 0;
  // phase occupancy (instrumented build only): rounds executed and lanes doing useful work in them, counted on lane 0
  unsigned long long node_iters = 0, node_lanes = 0, leaf_iters = 0, leaf_lanes = 0, refill_iters = 0, refill_lanes = 0;
};

// ---------------------------------------------------------------------------------------------
// Wave-persistent traversal.  A wave owns a strided sequence of 64-ray groups (group g of wave w is rays
// [64 * (g * n_waves + w), +64)) and keeps its 64 lanes busy: a lane whose ray has finished takes the next
// ray of the wave's sequence as soon as kRefill lanes are idle (no atomics: the sequence pointer is wave
// uniform).  Each round is  [refill] -> [inner-node phase, a share step before each of its iterations] -> [leaf phase] -> [merge] -> [retire]:
//   * inner-node phase: lanes sitting on an inner node test its four child boxes, descend into the nearest
//     hit child and push the others farthest first; lanes that reached a leaf wait.  The phase ends when no lane is on an
//     inner node, or when at least kLeafQuorum lanes are waiting on a leaf.
//   * leaf phase: every lane on a leaf runs the exact ray/triangle test once, then pops its stack.
//   * share / merge (only once the wave's sequence is exhausted, i.e. in the tail): an idle lane takes the OLDEST
//     pending subtree off the LDS stack of a busy lane (the stacks are LDS columns, so any lane can reach them),
//     copies that lane's ray through shuffles and traverses the subtree as a helper; its result is merged back into
//     the owner (smaller t, then smaller world id; any hit for shadow rays), which retires when no helper is left.  Owner and
//     helpers prune with the closest distance any of them has found so far (aux_t).
//     The longest rays then finish in a fraction of their serial time: they set the duration of a launch once a GPU
//     holds few rays per wave (tile sharding over 8 GPUs: k_trace's floor was 0.17 ms whatever the share of the frame).
//     Closest-hit and any-hit results do not depend on the visit order, so sharing changes no result; it is compiled
//     out of the instrumented kernels, whose node / triangle counts are defined by the serial walk.
// This replaces the one-ray-per-thread loop whose VALU lane utilisation was 24 % on the atrium
// (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), profiles/r01b_sq_counters.txt).
//
// ANY = false: closest hit in (tmin, tmax); ties on t go to the smaller world triangle id, so the result
// does not depend on visit order.  ANY = true: the first accepted hit ends the ray.
// Source: bool load(uint32_t ray, vec3& o, vec3& d, float& tmin, float& tmax)   (false = nothing to trace or report)
// Sink:   void store(uint32_t ray, const HitRecord&)
// lds_col: this lane's stack column; aux: this WAVE's 3 x 64 ints of LDS scratch (helpers per owner, donor list, stack bottoms)
// ---------------------------------------------------------------------------------------------
// (thresholds: device/tuning.h.  Leaf quorum: 12 / 16 / 20 / 24 / 32 lanes -> 0.571 / 0.554 / 0.542 / 0.541 / 0.557 ms per k_trace;
// the tail of a small share and the shadow rays have the same optimum.)
#ifdef GLZ_NO_LDS_TOP   // debugging switch: every node from memory
constexpr bool kLdsTop = false;
#else
constexpr bool kLdsTop = true;
#endif   // the top kBvhTopNodes nodes of the tree come from a per-block LDS copy ("LDS-staged node packets")
constexpr int kRefill = GLZ_REFILL;
constexpr int kTlRefill = GLZ_TL_REFILL;
constexpr int kLeafQuorum = GLZ_LEAF_QUORUM;
constexpr int kRefillAny = GLZ_REFILL_ANY, kLeafQuorumAny = GLZ_LEAF_QUORUM_ANY;   // the same for a pass of shadow rays only (k_trace's second pass, its shadow waves)
constexpr int kAlphaQuorum = GLZ_ALPHA_QUORUM;   // lanes waiting for the alpha test at which the alpha phase runs (trace_wave)
constexpr int kAuxPerWave = 3 * 64 + 4 * 64;   // work sharing (3 x 64) + the four child links of the node a lane is visiting
// The block's scratch, as the kernels declare it (__shared__ alignas(1024) int s_aux[kAuxPerBlock]): the waves' link areas first -- 256 ints
// each, so that every one of them starts on a 1 KB boundary (sorted_link) -- then their work-sharing words.
constexpr int kAuxPerBlock = (kBlock / 64) * kAuxPerWave;
__device__ __forceinline__ int* wave_links(int* s_aux, uint32_t wave_in_block) { return s_aux + 256u * wave_in_block; }
__device__ __forceinline__ int* wave_aux(int* s_aux, uint32_t wave_in_block) { return s_aux + 256u * (kBlock / 64) + 192u * wave_in_block; }

// The SIMD issues its OLDEST ready wave first, and the tracers are bound by VALU issue: with one priority for all, the waves of the
// blocks dispatched first ran 1.6 x faster than the last ones' through the same amount of work (1.67 against 2.66 us per node
// iteration) and then sat idle while those finished -- the closest-hit phase of a full frame ended between 240 and 396 us
// (tools/gpu_wave_times.py, by position in the grid, not by XCD).  Every wave changes its issue priority once per round, starting from
// the sixth of the grid its block is in: all get the same share, the phase ends between 307 and 389 us, k_trace 0.540 -> 0.514 ms
// (only for shares that give a wave at least two whole groups, see `rotate` in trace_wave).
// (Per node iteration instead of per round, keyed by the hardware wave slot instead of the block index, every second round: the
// same; every fourth round 0.523; priority by the wave's own progress -- groups behind first -- 0.545; the youngest first 0.566.)
// Six waves per SIMD, four levels: a cycle of six turns 3 2 2 1 1 0 (with four turns, waves four apart always tie and the older one wins: 0.514 -> 0.509 ms).
__device__ __forceinline__ void rotate_priority(uint32_t turn) {
  const uint32_t p = turn & 3u;   // s_setprio takes an immediate
  if (p == 0u) __builtin_amdgcn_s_setprio(0); else if (p == 1u) __builtin_amdgcn_s_setprio(1); else if (p == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
}

// the link of the child a sort key names, out of the wave's link scratch (`base` = LDS byte address of this lane's word of child 0, bits 8..9 zero)
__device__ __forceinline__ int sorted_link(uint32_t base, uint32_t key) {
  return *(LdsIntPtr)(uintptr_t)((key & kKeyChildMask) | base);
}
__device__ __forceinline__ void sort2(uint32_t& a, uint32_t& b) {
  const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}

// The staged nodes are read through a pointer that keeps its LDS address space: with a generic pointer the compiler
// merges the LDS and the global fetch of a node into ONE flat_load behind a pointer select -- every node of the tree then
// comes in through the flat path (measured: k_trace 0.586 -> 0.786 ms).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* LdsNodePtr;

// The rays a wave works through, in the order it takes them (all wave-uniform but the position asked for): the 64-ray groups are
// dealt round-robin, group g to wave g % n_waves.  (Dealing the left-over rays in pieces smaller than a group, so that every wave gets
// the same share of them, is slower everywhere -- a wave's iteration costs the same whatever the number of its lanes that work:
// EXPERIMENTS.md.)
struct RaySequence {
  uint32_t wave, n_waves, total;
  uint32_t own_full;     // rays this wave takes in whole rounds of the deal (a last, partial round may add one more group)
  __device__ __forceinline__ RaySequence(uint32_t wave_, uint32_t n_waves_, uint32_t total_) : wave(wave_), n_waves(n_waves_), total(total_) {
    own_full = (((total + 63u) >> 6) / n_waves) * 64u;
  }
  // ray at position `pos` of this wave's sequence; >= total: the sequence has ended (ray_at is monotonic in pos)
  // (A wave's groups neighbours of EACH OTHER -- groups 6 w .. 6 w + 5 -- instead of its block-mates': k_trace 0.510 -> 0.561 ms.)
  __device__ __forceinline__ uint32_t ray_at(uint32_t pos) const { return (wave + (pos >> 6) * n_waves) * 64u + (pos & 63u); }
};

#ifdef GLZ_SECTION_TIMES   // tuning builds only (tools/gpu_sections.py): shader clocks every wave spent in each part of trace_wave's round, and how often
static __device__ unsigned long long g_sections[16 * 8192];   // per wave: clocks {refill, share, node visit, loop control, leaf, merge + retire}, counts {rounds, node iterations, leaf phases, hand-over steps with a taker}
#define GLZ_SEC_STAMP(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (acc) += now_ - sec_t; sec_t = now_; } while (0)
#else
#define GLZ_SEC_STAMP(acc) do { } while (0)
#endif
#ifdef GLZ_WAVE_TIMES
__device__ unsigned long long g_wave_times[3 * 8192];
__device__ unsigned int g_wave_stats[8 * 8192];   // closest-hit phase: rounds, node iterations, lanes in them, leaf iterations, lanes in them, rounds with helpers
__device__ unsigned long long g_tl_stats[8];      // two-level tracer, summed over lanes: rays, top-level node visits, mesh node visits, instances entered, triangle tests, node iterations, leaf iterations
#endif

// MIXED (with ANY = false): the sequence holds rays of both kinds, the source says which after every load (src.any) -- k_path
// traces a wave's closest-hit rays and the shadow rays its previous launch queued in ONE pass, the shadow rays in the lanes
// the closest-hit rays leave idle.
// PREFETCH (k_path): the four 16-byte loads of a lane's NEXT inner node are issued the moment the node is known -- at the end of the visit
// that chose it, after a leaf phase's pop, when an idle lane takes a subtree over -- instead of at the top of the next node iteration, so
// that they are in flight during the pushes, the loop's ballots and the hand-overs to idle lanes in between.  A small share of the frame is
// bound by the LATENCY of this dependent chain (tools/gpu_sections.py: a node visit of a 1/32 share, on an otherwise idle chip, still takes
// 1 800 clocks, most of them waiting for the node), not by issue; the full-frame k_trace is issue bound and has no 16 registers to spare.
// WIDE8 (k_trace8, the tracer of a small tile share): the hierarchy's 8-wide nodes (types.h BvhNode8, two lines a visit).  A GPU that holds
// one 64-ray group per resident wave is bound by the LATENCY of a ray's chain of dependent node fetches (1 800 - 2 100 clocks per node
// iteration whatever the load), and eight-wide that chain is 31 % shorter (17.3 against 24.9 visits per sample on the bench scene,
// tools/bvh_lab); the full frame is bound by VALU issue and by the address units, where twice the boxes per visit cost more than the
// visits saved (it keeps the 4-wide nodes).  The nearest child is entered, the others are pushed in slot order -- no sort: ordering the
// rest by distance as well saves 0.7 % of the visits (tools/bvh_lab order=nearest) for 38 instructions a visit; the child links are
// picked in registers (a select tree on the key's child bits) rather than through LDS, one round trip less on the chain.  Hits do not
// depend on the visit order, so the images are those of the 4-wide walk bit for bit.  No staged top (the root is node 0), a deeper LDS
// stack (kLdsStack8).
// ALPHA: what becomes of a candidate on non-opaque geometry -- kAlphaNone: the scene has none (DeviceScene::has_non_opaque == 0; the
// kernel carries no alpha code), kAlphaInline: tested where it is met (the counting kernels, whose fetch counts are defined by the
// serial walk; k_path; k_trace8), kAlphaPhase: it waits for an alpha phase of its own (k_trace for scenes with opacity maps).
constexpr int kAlphaNone = 0, kAlphaInline = 1, kAlphaPhase = 2;
template <bool ANY, bool COUNT, bool MIXED = false, bool PREFETCH = false, bool WIDE8 = false, int ALPHA = kAlphaInline, class Source, class Sink>
__device__ __forceinline__ void trace_wave(const DeviceScene& S, Source& src, Sink& sink, int* __restrict__ lds_col, int* aux, int* link_scratch, LdsNodePtr top_lds,
                                           uint32_t* __restrict__ spill, uint32_t spill_depth, uint32_t total, uint32_t wave, uint32_t n_waves, TraceTally& tally) {
#if defined(GLZ_NO_SHARE_ANY)      // debugging switches (tools/build_variant.sh): the tail's work sharing off in one pass, or in both
  constexpr bool SHARE = !COUNT && !ANY;
#elif defined(GLZ_NO_SHARE_CLOSEST)
  constexpr bool SHARE = !COUNT && ANY;
#elif defined(GLZ_NO_SHARE)
  constexpr bool SHARE = false;
#else
  constexpr bool SHARE = !COUNT;
#endif
  constexpr bool TOP = kLdsTop && !WIDE8;
  constexpr int kLevels = WIDE8 ? kLdsStack8 : kLdsStack;
  const BvhNode8* __restrict__ nodes8 = S.bvh_nodes8;
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  // The wave's place in the deal is the same in all of its lanes, and the compiler has to KNOW that: everything that steers the rounds below
  // (`seq`, `exhausted`, the loops' exits) derives from these three, and a build in which they arrived through variables the compiler
  // could not prove uniform turned the loops into divergent ones -- lanes "leave" one by one, EXEC is empty behind the last exit -- and
  // placed the reloads of spilt registers in front of the instruction that restores EXEC: what was kept across the pass came back as
  // whatever the registers held (EXPERIMENTS.md, round 5: a few wrong pixels from run to run; clang 22).  v_readfirstlane says it.
  wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
  n_waves = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_waves);
  total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
  const BvhNode4* __restrict__ nodes = S.bvh_nodes;
  const BvhGrid grid = S.bvh_grid;
  const int lane = threadIdx.x & 63;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  int* aux_out = aux;          // [owner lane] helpers currently working for that lane's ray
  uint32_t* aux_t = reinterpret_cast<uint32_t*>(aux) + 64;   // [owner lane] bits of the smallest hit distance the ray's owner or any helper has found (tail only)
  int* aux_sb = aux + 128;     // [lane] lowest LDS stack level that may still hold a live entry
  int* aux_pair = link_scratch;   // [k] lane of the k-th donor of this round; shares its words with the child links, which only live inside a node visit
  if (SHARE) {
    aux_out[lane] = 0;
    aux_sb[lane] = 0;
  }
  uint32_t seq = 0;                                         // wave-uniform: rays of this wave's sequence handed out so far
  const RaySequence rays(wave, n_waves, total);
  // A share of the frame that gives a wave fewer than two whole groups is one long tail (work sharing from the first round on): there
  // the rotation costs 3 % (1/4 share 0.245 -> 0.253 ms per launch, 1/8 0.146 -> 0.150) where the full frame gains 4 % and a half 4.5 %.
  // (Rotating only until the wave's sequence is exhausted gains nothing anywhere: what the rotation evens out is the waves' last groups.)
  const bool rotate = rays.own_full >= 128u;
  // (giving each XCD one contiguous eighth of the groups -- rays of one image band per L2 -- measured 5 % slower: the bands
  // differ in cost and the static split loses more to imbalance than the L2 gains)
  // (Drawing the groups from a counter instead of the stride: the waves of a full-frame launch end between 257 and 406 us of a
  // 410 us closest-hit phase -- 5 or 6 groups each -- tools/gpu_wave_times.py.  One counter: 577 us, device-scope atomics on one
  // address are served at ~15 ns each; 32 interleaved counters: the ends move together, 306 - 400 us, but every wave gets slower --
  // neighbouring groups no longer run on one CU at one time -- 0.572 ms per k_trace either way; whole rounds by the stride and
  // only the last partial round drawn: 0.601 ms.  A wave's last group runs without refills behind it whoever hands it out.)
  bool exhausted = rays.ray_at(0u) >= total;                // wave-uniform
  // per-lane ray state
  bool open = false;                                        // a ray of this lane's own is in flight and its result has not been stored
  bool helper = false;                                      // this lane traverses a subtree of lane `ray`'s ray (work sharing)
  bool found_own = false;                                   // ... and has accepted a hit of its own since it took the subtree over (merge)
  bool any_lane = ANY;                                      // the ray in this lane ends with its first accepted hit (MIXED: per ray)
  bool alpha_wait = false;                                  // the lane sits on a leaf (cur < 0) with a candidate that needs the alpha test: it waits for the alpha phase
  int cur = kRayDone;
  uint32_t ray = 0;                                         // ray index (open) or owner lane (helper)
  vec3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f);
  vec3 ig = mk3(0.0f, 0.0f, 0.0f), cg = mk3(0.0f, 0.0f, 0.0f);   // grid-space ray: plane q is crossed at t = q * ig + cg
  SlabSel sel{kSlabSelLo, kSlabSelLo, kSlabSelLo};             // near-plane selectors, from the signs of ig
  float tmin = 0.0f, tmax = 0.0f;
  HitRecord best{0.0f, 0.0f, 0.0f, kNone};
  uint32_t best_id = kNone;
  // the spill area is indexed by the physical lane slot of the grid (a lane traverses one ray or subtree at a time)
  StackT<kLevels> st{lds_col, spill + ((size_t)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64u + (uint32_t)lane) * spill_depth, 0};
  // PREFETCH: the node `pf_cur` as loaded (or on its way)
  u32x4 pf0 = {0u, 0u, 0u, 0u}, pf1 = pf0, pf2 = pf0, pf3 = pf0, pf4 = pf0, pf5 = pf0, pf6 = pf0, pf7 = pf0;
  int pf_cur = -1;
  auto prefetch_node = [&]() {
    if (PREFETCH && WIDE8 && cur >= 0 && cur < kStolen) {
      const u32x4* np = reinterpret_cast<const u32x4*>(nodes8 + cur);
      pf0 = np[0]; pf1 = np[1]; pf2 = np[2]; pf3 = np[3]; pf4 = np[4]; pf5 = np[5]; pf6 = np[6]; pf7 = np[7];
      pf_cur = cur;
    } else
    if (PREFETCH && !WIDE8 && cur >= 0 && !(cur & kBvhTopFlag)) {   // a node of the table in memory (staged nodes, kRayDone and kStolen carry bit 30)
#ifdef GLZ_NODE48
      const u32x4* np = reinterpret_cast<const u32x4*>(S.bvh_nodes48 + cur);
      pf0 = np[0]; pf1 = np[1]; pf2 = np[2];
#else
      const u32x4* np = reinterpret_cast<const u32x4*>(nodes + cur);
      pf0 = np[0]; pf1 = np[1]; pf2 = np[2]; pf3 = np[3];
#endif
      pf_cur = cur;
    }
    // (The 64-byte record of a LEAF requested the same way, while the lane waits for the leaf phase's quorum: slower, a 1/8 share
    // 0.1379 -> 0.1418 ms per launch, 1/16 0.0997 -> 0.1058.)
  };
#ifdef GLZ_WAVE_TIMES
  unsigned int wt_rounds = 0, wt_node_iters = 0, wt_node_lanes = 0, wt_leaf_iters = 0, wt_leaf_lanes = 0, wt_helper_rounds = 0, wt_wait_rounds = 0;
#endif
#ifdef GLZ_SECTION_TIMES
  unsigned long long sec_t = __builtin_amdgcn_s_memtime(), sec_refill = 0, sec_share = 0, sec_node = 0, sec_ctl = 0, sec_leaf = 0, sec_tail = 0;
  unsigned long long sec_rounds = 0, sec_iters = 0, sec_leaves = 0, sec_takes = 0, sec_merges = 0, sec_merge = 0, sec_f0 = 0, sec_f1 = 0, sec_f2 = 0;
#endif
  // issue-priority rotation (rotate_priority above); k_path's MIXED pass keeps the priority its own kernel set
  // (inside k_path's mixed pass as well: a 1/4 share 0.276 -> 0.263 ms per launch, 1/8 0.1456 -> 0.1449, 1/16 0.118 -> 0.123)
  constexpr bool ROTATE = !MIXED;
  const uint32_t prio_gen = (blockIdx.x * 6u) / gridDim.x;   // which sixth of the grid: the order the blocks of a CU were dispatched in
  uint32_t prio_round = 0;
  // ---- share: idle lanes adopt the oldest pending subtree of a busy lane (called before every node iteration, see below) ----
  auto share_step = [&]() {
  if (SHARE && exhausted)
  for (int rep = 0; rep < 1; ++rep) {   // (more than one hand-over per donor and node iteration costs more in shuffles than it gains: 0.145 / 0.148 / 0.152 ms for 1 / 2 / 3)
    bool more = false;
    const bool busy = open || helper;
    const unsigned long long idle_m = __ballot(!busy);
    if (idle_m != 0ull) {
      if (ANY || MIXED) {   // helpers of a ray whose hit has been found have nothing left to decide
        const int owner_found = __shfl((int)(best.leaf != kNone), helper ? (int)ray : lane);
        if (helper && any_lane && owner_found) cur = kRayDone;
      }
      int sb = 0, lim = 0;
      if (busy && cur != kRayDone) {
        sb = aux_sb[lane];
        if (sb > st.sp) sb = st.sp;
        lim = st.sp < kLevels ? st.sp : kLevels;
        while (sb < lim && st.lds[sb * kBlock] == kStolen) ++sb;
        aux_sb[lane] = sb;
      }
      const bool can_give = busy && cur != kRayDone && sb < lim;
      const unsigned long long give_m = __ballot(can_give);
      const int n_give = __popcll(give_m), n_take = __popcll(idle_m);
      const int n_pairs = n_give < n_take ? n_give : n_take;
      if (n_pairs > 0) {
        int give = 0;
        if (can_give && __popcll(give_m & lanes_below) < n_pairs) {
          give = st.lds[sb * kBlock];
          st.lds[sb * kBlock] = kStolen;
          aux_sb[lane] = sb + 1;
          aux_pair[__popcll(give_m & lanes_below)] = lane;
        }
        const int take_rank = __popcll(idle_m & lanes_below);
        const bool take = !busy && take_rank < n_pairs;
        const int donor = take ? aux_pair[take_rank] : lane;   // same-wave LDS: the stores above are complete (in-order)
        // Every lane runs the shuffles.  Lanes that take nothing read their own lane (donor == lane), so the ray registers
        // can be assigned unconditionally: no temporaries stay live across the block (register pressure: 72 VGPRs).
        const int t_node = __shfl(give, donor);
        const int t_owner = __shfl(helper ? (int)ray : lane, donor);
        o.x = __shfl(o.x, donor); o.y = __shfl(o.y, donor); o.z = __shfl(o.z, donor);
        d.x = __shfl(d.x, donor); d.y = __shfl(d.y, donor); d.z = __shfl(d.z, donor);
        ig.x = __shfl(ig.x, donor); ig.y = __shfl(ig.y, donor); ig.z = __shfl(ig.z, donor);
        cg.x = __shfl(cg.x, donor); cg.y = __shfl(cg.y, donor); cg.z = __shfl(cg.z, donor);
        sel = SlabSel{slab_sel(ig.x), slab_sel(ig.y), slab_sel(ig.z)};
        tmin = __shfl(tmin, donor); tmax = __shfl(tmax, donor);
        best.t = __shfl(best.t, donor); best.u = __shfl(best.u, donor); best.v = __shfl(best.v, donor);
        best.leaf = (uint32_t)__shfl((int)best.leaf, donor);
        best_id = (uint32_t)__shfl((int)best_id, donor);
        if (MIXED) any_lane = __shfl((int)any_lane, donor) != 0;
#ifdef GLZ_SECTION_TIMES
        sec_takes += 1;
#endif
        if (take) {
          ray = (uint32_t)t_owner;
          cur = t_node;
          st.sp = 0;
          aux_sb[lane] = 0;
          helper = true;
          found_own = false;
          alpha_wait = false;
          atomicAdd(&aux_out[t_owner], 1);
          prefetch_node();
        }
        more = n_take > n_give;   // idle lanes are left over: the donors may have more to give
      }
    }
    if (!more) break;
  }
  };
  for (;;) {
#ifdef GLZ_WAVE_TIMES
    wt_rounds += 1;
    wt_helper_rounds += __ballot(helper) != 0ull;
    wt_wait_rounds += __ballot(open && cur == kRayDone) != 0ull && __ballot(open && cur != kRayDone) == 0ull;   // owners only waiting for helpers
#endif
    if (ROTATE && rotate) {   // six waves per SIMD, four levels: a cycle of six turns 3 2 2 1 1 0
      const uint32_t pos = (prio_gen + prio_round++) % 6u;
      rotate_priority(pos == 0u ? 3u : (pos < 3u ? 2u : (pos < 5u ? 1u : 0u)));
    }
#ifdef GLZ_SECTION_TIMES
    sec_rounds += 1;
#endif
    GLZ_SEC_STAMP(sec_tail);
    // ---- refill ----
    const unsigned long long idle = __ballot(!(open || helper));
    const int n_idle = __popcll(idle);
    if (!exhausted && n_idle >= (ANY ? kRefillAny : kRefill)) {
      if (COUNT && lane == 0) { tally.refill_iters += 1; tally.refill_lanes += (unsigned)n_idle; }
      const uint32_t next_ray = rays.ray_at(seq + (uint32_t)__popcll(idle & lanes_below));
      if (!open && next_ray < total) {
        if (src.load(next_ray, o, d, tmin, tmax)) {
          ray = next_ray;
          alpha_wait = false;
          best = HitRecord{tmax, 0.0f, 0.0f, kNone};
          best_id = kNone;
          if constexpr (MIXED) any_lane = src.any;
          if (COUNT) tally.rays += 1;
          if (S.n_world_tris == 0 || !ray_is_finite(o, d)) {
            sink.store(ray, best);                          // nothing to intersect / nothing can be hit: a miss
          } else {
            const vec3 og = mk3((o.x - grid.lo[0]) * grid.inv_cell[0], (o.y - grid.lo[1]) * grid.inv_cell[1], (o.z - grid.lo[2]) * grid.inv_cell[2]);
            if (COUNT) ig = mk3(grid_inv_dir(d.x) * grid.cell[0], grid_inv_dir(d.y) * grid.cell[1], grid_inv_dir(d.z) * grid.cell[2]);
            else ig = mk3(grid_inv_dir_fast(d.x) * grid.cell[0], grid_inv_dir_fast(d.y) * grid.cell[1], grid_inv_dir_fast(d.z) * grid.cell[2]);
            cg = mk3(grid_addend(og.x, ig.x), grid_addend(og.y, ig.y), grid_addend(og.z, ig.z));
            sel = SlabSel{slab_sel(ig.x), slab_sel(ig.y), slab_sel(ig.z)};
            st.sp = 0;
            if (SHARE) aux_sb[lane] = 0;
            cur = TOP ? kBvhTopFlag : 0;   // the root (slot 0 of the staged table)
            open = true;
          }
        }
      }
      seq += (uint32_t)n_idle;
      exhausted = rays.ray_at(seq) >= total;
      // The tail begins: from here on a ray may be worked on by several lanes, which tell each other the closest distance found so
      // far through aux_t -- a helper walking a far subtree with the bound it was handed at the start would go through all of
      // it after the owner has long found something nearer, and the owner cannot retire before its helpers are back.
      if (SHARE && !ANY && exhausted && open) aux_t[lane] = __float_as_uint(best.t);
    }
    GLZ_SEC_STAMP(sec_refill);
    if (__ballot(open || helper) == 0ull) {
      if (exhausted) break;
      continue;
    }
    // ---- inner-node phase ----
    for (;;) {
      // Idle lanes take over pending subtrees before EVERY node iteration of the tail, not once per round: a round is several
      // iterations long, and with one hand-over per round the helpers of a long ray multiplied too slowly to matter before it
      // was over (a 1/8 share: 0.153 -> 0.147 ms per launch; the full frame, where only each wave's last group is a tail: 0.930 -> 0.914).
      // (Every 2nd / 3rd node iteration instead: a 1/8 share 0.1352 -> 0.1382 / 0.1404 ms per launch.)
      GLZ_SEC_STAMP(sec_ctl);
      share_step();
      GLZ_SEC_STAMP(sec_share);
      // (Reading the first word of the triangle as soon as a lane of the tail arrives at a leaf, so that the line is on its way while
      // the others finish their node iterations: slower, 0.146 -> 0.149 ms for a 1/8 share and 0.905 -> 0.924 ms for the full frame.)
      const bool at_node = cur >= 0 && cur < kStolen;
      const unsigned long long m_node = __ballot(at_node);
      if (m_node == 0ull) break;
      if (COUNT && lane == 0) { tally.node_iters += 1; tally.node_lanes += (unsigned)__popcll(m_node); }
#ifdef GLZ_WAVE_TIMES
      wt_node_iters += 1; wt_node_lanes += (unsigned)__popcll(m_node);
#endif
#ifdef GLZ_SECTION_TIMES
      sec_iters += 1;
#endif
      GLZ_SEC_STAMP(sec_ctl);
      if (WIDE8) {
       if (at_node) {
        // 128-byte node = 8 x dwordx4: eight child boxes and eight links.  The nearest child is entered, the others pushed in slot order.
        u32x4 w0, w1, w2, w3, w4, w5, w6, w7;
        if (PREFETCH) {
          if (pf_cur != cur) prefetch_node();
          w0 = pf0; w1 = pf1; w2 = pf2; w3 = pf3; w4 = pf4; w5 = pf5; w6 = pf6; w7 = pf7;
        } else {
          const u32x4* np = reinterpret_cast<const u32x4*>(nodes8 + cur);
          w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3]; w4 = np[4]; w5 = np[5]; w6 = np[6]; w7 = np[7];
        }
        if (COUNT) tally.nodes += 1;
        float bound = best.t;
        if (SHARE && !ANY && exhausted) bound = fminf(bound, __uint_as_float(aux_t[helper ? (int)ray : lane]));
        uint32_t key[8];
        key[0] = box_key8(w0.x, w0.y, w0.z, 0u << 8, sel, ig, cg, tmin, bound); key[1] = box_key8(w0.w, w1.x, w1.y, 1u << 8, sel, ig, cg, tmin, bound);
        key[2] = box_key8(w1.z, w1.w, w2.x, 2u << 8, sel, ig, cg, tmin, bound); key[3] = box_key8(w2.y, w2.z, w2.w, 3u << 8, sel, ig, cg, tmin, bound);
        key[4] = box_key8(w3.x, w3.y, w3.z, 4u << 8, sel, ig, cg, tmin, bound); key[5] = box_key8(w3.w, w4.x, w4.y, 5u << 8, sel, ig, cg, tmin, bound);
        key[6] = box_key8(w4.z, w4.w, w5.x, 6u << 8, sel, ig, cg, tmin, bound); key[7] = box_key8(w5.y, w5.z, w5.w, 7u << 8, sel, ig, cg, tmin, bound);
        const uint32_t ka = key[0] < key[1] ? key[0] : key[1], kb = key[2] < key[3] ? key[2] : key[3], kc = key[4] < key[5] ? key[4] : key[5], kd = key[6] < key[7] ? key[6] : key[7];
        const uint32_t kab = ka < kb ? ka : kb, kcd = kc < kd ? kc : kd;
        const uint32_t kmin = kab < kcd ? kab : kcd;
        if (kmin == 0xFFFFFFFFu) {
          cur = SHARE ? st.pop_live() : (st.sp ? st.pop() : kRayDone);
          prefetch_node();
        } else {
          const int link[8] = {(int)w6.x, (int)w6.y, (int)w6.z, (int)w6.w, (int)w7.x, (int)w7.y, (int)w7.z, (int)w7.w};
          const bool b0 = (kmin & 0x100u) != 0u, b1 = (kmin & 0x200u) != 0u, b2 = (kmin & 0x400u) != 0u;
          const int s01 = b0 ? link[1] : link[0], s23 = b0 ? link[3] : link[2], s45 = b0 ? link[5] : link[4], s67 = b0 ? link[7] : link[6];
          const int t03 = b1 ? s23 : s01, t47 = b1 ? s67 : s45;
          const int nearest = b2 ? t47 : t03;
          if (__ballot(st.sp + 7 > kLevels) == 0ull) {   // wave-uniform: every lane stays inside the LDS part of its stack
#pragma unroll
            for (int k = 7; k >= 0; --k)
              if (key[k] != 0xFFFFFFFFu && key[k] != kmin) { st.lds[st.sp * kBlock] = link[k]; ++st.sp; }
          } else {
#pragma unroll
            for (int k = 7; k >= 0; --k)
              if (key[k] != 0xFFFFFFFFu && key[k] != kmin) st.push(link[k]);
          }
          cur = nearest;
          prefetch_node();
        }
       }
      } else
      if (at_node) {
        // 64-byte node = 4 x dwordx4: four child boxes in 16-bit grid coordinates (the ray was mapped into grid units at
        // refill) and four links.  Children are entered nearest first; the others are pushed farthest first.
        // Nodes of the top levels come out of the block's LDS copy (their `cur` carries kBvhTopFlag | slot); lanes that read the
        // same staged node broadcast.
#ifdef GLZ_NODE48
        u32x4 w0, w1, w3;
        if (kLdsTop && (cur & kBvhTopFlag)) {
          LdsNodePtr np = top_lds + 3 * (cur & 0xFFFF);
          w0 = np[0]; w1 = np[1]; w3 = np[2];
        } else if (PREFETCH) {
          if (pf_cur != cur) prefetch_node();
          w0 = pf0; w1 = pf1; w3 = pf2;
        } else {
          const u32x4* np = reinterpret_cast<const u32x4*>(S.bvh_nodes48 + cur);
          w0 = np[0]; w1 = np[1]; w3 = np[2];
        }
        if (COUNT) tally.nodes += 1;
        float bound = best.t;
        if (SHARE && !ANY && exhausted) bound = fminf(bound, __uint_as_float(aux_t[helper ? (int)ray : lane]));
        // the node-local ray: planes are origin + q 2^e grid cells
        const float ox = __uint_as_float(__builtin_amdgcn_perm(w0.x, kSlabMagic, kSlabSelLo)), oy = __uint_as_float(__builtin_amdgcn_perm(w0.x, kSlabMagic, kSlabSelHi)),
                    oz = __uint_as_float(__builtin_amdgcn_perm(w0.y, kSlabMagic, kSlabSelLo));   // 32768 + origin
        const vec3 na = mk3(ldexpf(ig.x, (int)((w0.y >> 16) & 15u)), ldexpf(ig.y, (int)((w0.y >> 20) & 15u)), ldexpf(ig.z, (int)((w0.y >> 24) & 15u)));
        const vec3 nc = mk3(fmaf(-32768.0f, na.x, fmaf(ox, ig.x, cg.x)), fmaf(-32768.0f, na.y, fmaf(oy, ig.y, cg.y)), fmaf(-32768.0f, na.z, fmaf(oz, ig.z, cg.z)));
        const SlabSel s48{ig.x < 0.0f ? kSel48Hi : kSel48Lo, ig.y < 0.0f ? kSel48Hi : kSel48Lo, ig.z < 0.0f ? kSel48Hi : kSel48Lo};
        uint32_t k0 = box_key48(w0.z, w1.x, w1.z, 0u, 0u, s48, na, nc, tmin, bound), k1 = box_key48(w0.z, w1.x, w1.z, kKeyChild, 0x0200u, s48, na, nc, tmin, bound);
        uint32_t k2 = box_key48(w0.w, w1.y, w1.w, 2u * kKeyChild, 0u, s48, na, nc, tmin, bound), k3 = box_key48(w0.w, w1.y, w1.w, 3u * kKeyChild, 0x0200u, s48, na, nc, tmin, bound);
#else
        u32x4 w0, w1, w2, w3;
        if (TOP && (cur & kBvhTopFlag)) {
          LdsNodePtr np = top_lds + 4 * (cur & 0xFFFF);
          w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3];
        } else if (PREFETCH) {
          if (pf_cur != cur) prefetch_node();   // (the ray has just started, or its stack was popped by someone who could not know)
          w0 = pf0; w1 = pf1; w2 = pf2; w3 = pf3;
        } else {
          const u32x4* np = reinterpret_cast<const u32x4*>(nodes + cur);
          w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3];
          // (A fifth 16-byte load from the node's own line costs 2.4 % of the kernel, 0.580 -> 0.594 ms: a 48-byte node format --
          // 8-bit boxes relative to a per-node origin -- would buy about that and pay ~12 VALU instructions per visit for it.)
        }
        if (COUNT) tally.nodes += 1;
#if defined(GLZ_SECTION_TIMES) && GLZ_SECTION_TIMES >= 2   // -DGLZ_SECTION_TIMES=2: the node visit in pieces
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        GLZ_SEC_STAMP(sec_f0);   // the node's words are here
#endif
        float bound = best.t;
        if (SHARE && !ANY && exhausted) bound = fminf(bound, __uint_as_float(aux_t[helper ? (int)ray : lane]));   // positive floats order like their bits
        uint32_t k0 = box_key<false>(w0.x, w0.y, w0.z, w3.x, 0u, sel, ig, cg, cg, tmin, bound), k1 = box_key<false>(w0.w, w1.x, w1.y, w3.y, kKeyChild, sel, ig, cg, cg, tmin, bound);
        uint32_t k2 = box_key<false>(w1.z, w1.w, w2.x, w3.z, 2u * kKeyChild, sel, ig, cg, cg, tmin, bound), k3 = box_key<false>(w2.y, w2.z, w2.w, w3.w, 3u * kKeyChild, sel, ig, cg, cg, tmin, bound);
#endif
        sort2(k0, k1); sort2(k2, k3); sort2(k0, k2); sort2(k1, k3); sort2(k1, k2);
        // The links go through LDS: picking one of four registers by a per-lane index costs 6 VALU instructions (the
        // kernel's bottleneck), an LDS read at a computed address 2 (k_trace 0.714 -> 0.691 ms).  The scratch is laid out
        // [child][lane] so that every access of a wave instruction has bank = lane % 32 (the [lane][child] layout with one
        // 16-byte store put lanes l, l + 8, l + 16, l + 24 of a half-wave on the same banks: 4.6 M conflict cycles per launch,
        // 22 % of the LDS-active cycles), and all four sorted links are fetched before the first one is used: the reads
        // are independent, so one LDS round trip covers them instead of one per push (read -> wait -> write, four times over).
#if defined(GLZ_SECTION_TIMES) && GLZ_SECTION_TIMES >= 2   // -DGLZ_SECTION_TIMES=2: the node visit in pieces
        asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));
        GLZ_SEC_STAMP(sec_f1);   // box tests and sort
#endif
        int* links = link_scratch + lane;
        links[0] = (int)w3.x; links[64] = (int)w3.y; links[128] = (int)w3.z; links[192] = (int)w3.w;
        const uint32_t link_base = (uint32_t)(uintptr_t)(LdsIntPtr)links;   // the wave's area is 1 KB aligned: bits 8..9 are the child's
        const int l0 = sorted_link(link_base, k0), l1 = sorted_link(link_base, k1), l2 = sorted_link(link_base, k2), l3 = sorted_link(link_base, k3);
        // (Three unconditional stores with the stack pointer advancing by one per valid key -- the invalid links of the sorted
        // sequence are overwritten by the next store or stay above the top -- remove 12 scalar / branch instructions per round
        // and measured slower, 0.587 -> 0.597 ms: the extra DS stores cost more than the exec-mask branches.)
#if defined(GLZ_SECTION_TIMES) && GLZ_SECTION_TIMES >= 2   // -DGLZ_SECTION_TIMES=2: the node visit in pieces
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GLZ_SEC_STAMP(sec_f2);   // links through LDS
#endif
        if (k0 == 0xFFFFFFFFu) {
          cur = SHARE ? st.pop_live() : (st.sp ? st.pop() : kRayDone);
          prefetch_node();
        } else {
          if (__ballot(st.sp + 3 > kLdsStack) == 0ull) {   // wave-uniform: every lane stays inside the LDS part of its stack (no spill branches)
            if (k3 != 0xFFFFFFFFu) { st.lds[st.sp * kBlock] = l3; ++st.sp; }
            if (k2 != 0xFFFFFFFFu) { st.lds[st.sp * kBlock] = l2; ++st.sp; }
            if (k1 != 0xFFFFFFFFu) { st.lds[st.sp * kBlock] = l1; ++st.sp; }
          } else {
            if (k3 != 0xFFFFFFFFu) st.push(l3);
            if (k2 != 0xFFFFFFFFu) st.push(l2);
            if (k1 != 0xFFFFFFFFu) st.push(l1);
          }
          cur = l0;
          prefetch_node();
        }
      }
#ifdef GLZ_SECTION_TIMES
      if (!PREFETCH) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the visit's loads and LDS traffic are charged to the visit
#endif
      GLZ_SEC_STAMP(sec_node);
      if (__popcll(__ballot(cur < 0 && !(ALPHA == kAlphaPhase && alpha_wait))) >= (ANY ? kLeafQuorumAny : kLeafQuorum)) break;
      // (Postponed leaves -- a lane parks the first leaf it arrives at and goes on with its stack, blocks at the second, the parked
      // leaves are tested first in the next leaf phase; Aila & Laine's speculative traversal -- k_trace 0.512 -> 0.540 ms with the
      // leaf phase at 24 waiting lanes, 0.542 / 0.555 at 16 / 32: the visits made without the parked leaf's bound and the second
      // leaf pass cost more than the fuller node iterations save.)
      // (Leaving for a refill as soon as kRefill finished lanes have piled up, without a leaf phase for the few lanes that wait
      // on a leaf, measured slower: 0.588 -> 0.611 ms, node rounds 41.1 -> 42.1 of 64 lanes.  The idle lanes are not what
      // holds the utilisation down.)
    }
    GLZ_SEC_STAMP(sec_ctl);
#ifdef GLZ_SECTION_TIMES
    sec_leaves += __ballot(cur < 0) != 0ull;
#endif
    // ---- leaf phase ----
    if (COUNT) {
      const unsigned long long m_leaf = __ballot(cur < 0);
      if (lane == 0 && m_leaf) { tally.leaf_iters += 1; tally.leaf_lanes += (unsigned)__popcll(m_leaf); }
    }
#ifdef GLZ_WAVE_TIMES
    { const unsigned long long ml = __ballot(cur < 0); if (ml) { wt_leaf_iters += 1; wt_leaf_lanes += (unsigned)__popcll(ml); } }
#endif
    // Candidates on NON-OPAQUE geometry go through the alpha test (raytrace_hit.rahit:24-39) -- the triangle's texture coordinates, the
    // material's opacity map, its descriptor, four texels: a chain of dependent fetches that the whole wave used to sit through whenever ONE
    // of its lanes met such a candidate (with a twentieth of the rays meeting one, most leaf phases: the Sponza-like atrium's k_trace took
    // 0.72 ms against 0.48 without the opacity maps, tools/gpu_sponza_like.py).  So the test has a phase of its own, with a quorum like the
    // leaf phase's: a lane whose leaf holds such a candidate stays on the leaf (alpha_wait) while the others go on, and the waiting lanes
    // take the test together.  The leaf is then tested again from the start -- same operations, same bits -- so nothing is kept per lane
    // but the flag; the candidates of one ray may be decided in another order than the serial walk's, which changes no result (the closest
    // hit is a minimum over the candidates that pass, ties by world id; an occluded ray is occluded).  The counting kernels keep the serial
    // walk: their texture-fetch counts are defined by it.  Measured on one box (tools/gpu_sponza_variants.py, profiles/r05_alpha_phase.txt):
    // the atrium with opacity maps 0.716 -> 0.670 ms per k_trace (quorum 1 / 4 / 8 / 12 / 16 / 24 / 48: 0.807 / 0.745 / 0.686 / 0.673 / 0.670 /
    // 0.701 / 1.018; with the verdict for free 0.601: what is left are the rays that go on through the holes), and the atrium WITHOUT
    // non-opaque geometry 0.489 -> 0.502 for the two ballots a round and the second copy of the leaf code -- so a scene without opacity
    // maps runs the kernel that has no alpha code at all (kAlphaNone).
    constexpr bool DEFER = ALPHA == kAlphaPhase;
    static_assert(!(COUNT && DEFER), "the counting kernels keep the serial walk");
    auto leaf_visit = [&](auto with_alpha_tag) {
      constexpr bool WITH_ALPHA = decltype(with_alpha_tag)::value;
      // A leaf is one 64-byte record (types.h BvhQuad): one triangle or two that share an edge, tested together (ray_quad).  The
      // candidates are then taken in slot order, the order the 48-byte records were walked in (the alpha test's fetches are counted).
      const uint32_t leaf = (uint32_t)~cur;
      const float4* qp = reinterpret_cast<const float4*>(S.bvh_quads + leaf);
      const float4 r0 = qp[0], r1 = qp[1], r2 = qp[2], r3 = qp[3];
      const uint32_t id0 = __float_as_uint(r0.w), qflags = __float_as_uint(r2.w), slot0 = __float_as_uint(r3.w);
      const bool pair = (qflags & kTriHasPartner) != 0u;
      if (COUNT) tally.tris += pair ? 2 : 1;
      const RayShear rs = ray_shear(d);   // (kept in registers with the ray instead: fits without spills, 0.555 against 0.552 ms -- no gain)
      const QuadHit qh = ray_quad(rs, r0, r1, r2, r3, pair, o, tmin);
      const uint32_t swapped = (qflags & kQuadSwapped) ? 1u : 0u;
      const bool non_opaque = ALPHA != kAlphaNone && (qflags & kTriNonOpaque) != 0u;
      bool finished = false, wait = false;
#pragma nounroll
      for (uint32_t which = 0; which < 2u; ++which) {   // the leaf's first triangle, then its partner
        const bool is_b = (which ^ swapped) != 0u;
        const float t = is_b ? qh.t[1] : qh.t[0], u = is_b ? qh.u[1] : qh.u[0], v = is_b ? qh.v[1] : qh.v[0];
        if ((is_b ? qh.ok[1] : qh.ok[0]) && t < tmax) {
          const uint32_t wid = id0 + which, slot = slot0 + which;
          const bool better = best.leaf == kNone ? true : (t < best.t || (t == best.t && wid < best_id));
          if (better && non_opaque && !WITH_ALPHA) {
            wait = true;   // decided in the alpha phase
          } else if (better && (!non_opaque || alpha_test(S, slot, u, v))) {
            best = HitRecord{t, u, v, slot};
            best_id = wid;
            found_own = true;
            finished = any_lane;
            if (SHARE && !ANY && exhausted) atomicMin(&aux_t[helper ? (int)ray : lane], __float_as_uint(t));
          }
        }
      }
      if (!WITH_ALPHA && wait) {
        alpha_wait = true;   // stays on the leaf
      } else {
        alpha_wait = false;
        cur = finished ? kRayDone : (SHARE ? st.pop_live() : (st.sp ? st.pop() : kRayDone));
        prefetch_node();
      }
    };
    if (cur < 0 && !(DEFER && alpha_wait)) {
      if constexpr (DEFER) leaf_visit(std::false_type{}); else leaf_visit(std::true_type{});
    }
    if constexpr (DEFER) {
      // ---- alpha phase: when enough lanes wait for it, or when nobody has anything else to do
      const unsigned long long m_wait = __ballot(alpha_wait && cur < 0);
      if (m_wait != 0ull && (__popcll(m_wait) >= kAlphaQuorum || __ballot((open || helper) && cur != kRayDone && !(alpha_wait && cur < 0)) == 0ull)) {
        if (alpha_wait && cur < 0) leaf_visit(std::true_type{});
      }
    }
#ifdef GLZ_SECTION_TIMES
    if (!PREFETCH) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    GLZ_SEC_STAMP(sec_leaf);
    // ---- merge: finished helpers hand their result to the owner of the ray ----
    if (SHARE) {
      // All helpers that are done leave together; only those that can have CHANGED their owner's result take a turn in the loop: a
      // helper that accepted a hit of its own (it starts from a copy of the donor's best, which the owner has) and, for closest-hit
      // rays, whose distance is still the smallest anyone has found for that ray (aux_t, kept by every lane of the tail at each
      // accepted hit: a result behind it cannot win, and whoever holds the smallest one either is the owner or will be here when it is
      // done).  A turn reads the helper's lane with v_readlane -- its index is wave uniform -- instead of through the LDS permute.
      // (tools/gpu_sections.py: a 1/8 share merged 71 helper results per wave and launch one after the other, 457 clocks each, 17 % of
      // the tracing time.)
      const bool done = helper && cur == kRayDone;
      bool cand = done && found_own;
      if (cand && !ANY && !any_lane) cand = __float_as_uint(best.t) == aux_t[ray];
      unsigned long long fin = __ballot(cand);
      while (fin != 0ull) {
#ifdef GLZ_SECTION_TIMES
        sec_merges += 1;
#endif
        const int h = __ffsll((long long)fin) - 1;   // wave uniform
        fin &= fin - 1ull;
        const int ow = __builtin_amdgcn_readlane((int)ray, h);
        const float bt = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(best.t), h));
        const float bu = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(best.u), h));
        const float bv = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(best.v), h));
        const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)best.leaf, h), bi = (uint32_t)__builtin_amdgcn_readlane((int)best_id, h);
        if (lane == ow) {
          const bool better = best.leaf == kNone ? true : (bt < best.t || (bt == best.t && bi < best_id));
          if (better) {
            best = HitRecord{bt, bu, bv, bl};
            best_id = bi;
          }
          if (any_lane) cur = kRayDone;   // occluded: the rest of the owner's stack does not matter
        }
      }
      if (done) {
        helper = false;
        atomicSub(&aux_out[ray], 1);
      }
    }
    GLZ_SEC_STAMP(sec_merge);
    // ---- retire ----
    // (Storing finished rays only when they make up a refill together with the idle lanes -- a fifth as many executions of the sink's code,
    // with a quarter of the wave in it instead of a lane or two: k_trace 0.483 against 0.483 ms, a 1/8 share 0.1312 against 0.1302.)
    if (open && cur == kRayDone && (!SHARE || aux_out[lane] == 0)) {
      if (COUNT) tally.hits += best.leaf != kNone;
      sink.store(ray, best);
      open = false;
    }
  }
  if (ROTATE) __builtin_amdgcn_s_setprio(0);
#ifdef GLZ_SECTION_TIMES
  {
    GLZ_SEC_STAMP(sec_tail);
    const uint32_t gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (lane == 0 && gw < 8192u) {
      unsigned long long* g = g_sections + 16 * gw;
      g[0] += sec_refill; g[1] += sec_share; g[2] += sec_node; g[3] += sec_ctl; g[4] += sec_leaf; g[5] += sec_tail;
      g[6] += sec_rounds; g[7] += sec_iters; g[8] += sec_leaves; g[9] += sec_takes; g[10] += 1; g[11] += sec_merges; g[12] += sec_merge; g[13] += sec_f0; g[14] += sec_f1; g[15] += sec_f2;
    }
  }
#endif
#ifdef GLZ_WAVE_TIMES
  if (!ANY && lane == 0 && wave < 8192u) {
    unsigned int* o = g_wave_stats + 8 * wave;
    o[0] = wt_rounds; o[1] = wt_node_iters; o[2] = wt_node_lanes; o[3] = wt_leaf_iters; o[4] = wt_leaf_lanes; o[5] = wt_helper_rounds; o[6] = wt_wait_rounds;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Two-level traversal (instanced scenes, DeviceScene::two_level; types.h TlasInstance): the same wave-persistent rounds over a
// top level whose leaves are instances and, inside an instance, the mesh's object-space hierarchy.
//   * Entering an instance (a top-level leaf, handled in the leaf phase) pushes an exit marker, takes the ray into object space
//     ONLY to re-derive the grid-space ray of the mesh's own quantisation grid -- with every box widened by the instance's slack,
//     folded into the slab test's addends -- and continues at the mesh's root.  Popping the marker re-derives the top-level
//     grid-space ray from the world ray, which never leaves its registers.
//   * A mesh leaf transforms its one or two OBJECT triangles to world space with the instance's matrix, operation for
//     operation what k_world_tris does for the flattened build, and runs the same world-space Moeller-Trumbore test: hits
//     (t, u, v, tie-break by world triangle id) are bit-identical to the flattened twin of the scene, whatever the hierarchy.
// (Tail work sharing of TOP-level entries -- an idle lane takes the oldest top-level entry below a busy lane's exit marker, with the
// world ray through __shfl and the top-level grid ray from the donor's LDS column, and enters instances on its own -- was built and
// measured: bit-identical, and slower everywhere, forest x 200 0.820 -> 0.864 ms per launch, a 1/8 share 0.160 -> 0.180; once per
// round instead of per node iteration 0.856 / 0.174.  A stolen top-level subtree costs its helper instance entries that the owner,
// with the bound of the hit it finds first, mostly never makes.)
// Simpler than trace_wave on purpose (no tail work sharing; the staged top is built in and off, kTlLdsTop): instanced scenes are about memory -- O(meshes + instances) instead of
// O(instances x triangles) -- and must not put the tuned flattened path at risk.
// ---------------------------------------------------------------------------------------------
constexpr int kExitInstance = 0x7FFFFFFD;   // stack marker: the entries below belong to the top level
// refill / leaf-phase thresholds of the two-level tracer (lanes): defaults = the flattened tracer's
// (Leaf quorum of this tracer, GLZ_TL_LEAF_QUORUM in device/tuning.h: 8 / 16 / 24 / 32 / 40 / 48 lanes -> 0.993 / 0.889 / 0.844 / 0.825 / 0.822 / 0.835 ms
// per launch (forest x 200): a leaf visit here is an instance entry or a triangle taken to world space, dearer than the flattened tracer's.
// The top level's first nodes from a per-block LDS copy, as in the flattened tracer: 0.823 -> 0.833 ms -- built in, off.)
constexpr bool kTlLdsTop = false;


template <bool ANY, bool COUNT, class Source, class Sink>
__device__ __forceinline__ void trace_wave_tl(const DeviceScene& S, Source& src, Sink& sink, int* __restrict__ lds_col, int* aux, int* link_scratch, float* __restrict__ top_ray, LdsNodePtr top_lds,
                                              uint32_t* __restrict__ spill, uint32_t spill_depth, uint32_t total, uint32_t wave, uint32_t n_waves, TraceTally& tally) {
  constexpr uint32_t kNone = 0xFFFFFFFFu;
  wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);   // (uniform, and said so: see trace_wave)
  n_waves = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_waves);
  total = (uint32_t)__builtin_amdgcn_readfirstlane((int)total);
  const BvhNode4* __restrict__ nodes = S.bvh_nodes;        // top-level nodes first, the meshes' after them (TlasInstance::node_base)
  const TlasInstance* __restrict__ instances = S.tlas_instances;
  const int lane = threadIdx.x & 63;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  uint32_t seq = 0;
  const RaySequence rays(wave, n_waves, total);
  bool exhausted = rays.ray_at(0u) >= total;
  bool open = false;
  int cur = kRayDone;
  uint32_t ray = 0, nbase = 0, cur_inst = kNone;
  vec3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f);                                  // the WORLD ray, always
  vec3 ig = mk3(0.0f, 0.0f, 0.0f), cgn = mk3(0.0f, 0.0f, 0.0f), cgf = mk3(0.0f, 0.0f, 0.0f);    // grid-space ray of the level the lane is in
  SlabSel sel{kSlabSelLo, kSlabSelLo, kSlabSelLo};
  float tmin = 0.0f, tmax = 0.0f;
  HitRecord best{0.0f, 0.0f, 0.0f, kNone, 0u, kNone};
  Stack st{lds_col, spill + ((size_t)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 64u + (uint32_t)lane) * spill_depth, 0};
  // grid-space ray for grid g from a ray (oo, dd) given in that grid's space; boxes widened by `slack` (a length in that space)
  // plus `cells` cells on every side
  auto set_grid_ray = [&](const float* glo, const float* gcell, const float* ginv, vec3 oo, vec3 dd, float slack, float cells) {
    const vec3 og = mk3((oo.x - glo[0]) * ginv[0], (oo.y - glo[1]) * ginv[1], (oo.z - glo[2]) * ginv[2]);
    const vec3 id = mk3(grid_inv_dir(dd.x), grid_inv_dir(dd.y), grid_inv_dir(dd.z));
    ig = mk3(id.x * gcell[0], id.y * gcell[1], id.z * gcell[2]);
    const vec3 cg = mk3(grid_addend(og.x, ig.x), grid_addend(og.y, ig.y), grid_addend(og.z, ig.z));
    // The pad as a LENGTH of that space (slack + `cells` cells) over the direction, not as a number of cells: across the thin side of a
    // flat mesh a cell is 1e-13 of a unit, and a pad counted in cells (it was, capped at twice the grid's span) is then far less than the
    // rounding of the transformed ray it stands for -- coplanar meshes lost near ties, thin ones hits (tools/gpu_fuzz_parity.py).
    const vec3 w = mk3((slack + cells * gcell[0]) * fabsf(id.x), (slack + cells * gcell[1]) * fabsf(id.y), (slack + cells * gcell[2]) * fabsf(id.z));
    cgn = cg - w;
    cgf = cg + w;
    sel = SlabSel{slab_sel(ig.x), slab_sel(ig.y), slab_sel(ig.z)};
  };
#ifdef GLZ_WAVE_TIMES
  unsigned long long tl_rays = 0, tl_top = 0, tl_mesh = 0, tl_enter = 0, tl_tris = 0, tl_niter = 0, tl_liter = 0;
#endif
  // The top level's grid-space ray is derived once per ray and parked in the lane's LDS column top_ray[k * kBlock] (k = 0..8: ig, cgn,
  // cgf): leaving an instance reloads it instead of re-deriving it from the world ray (three correctly rounded divisions).
  auto to_top_level = [&](bool fresh) {
    cur_inst = kNone;
    nbase = 0u;
    if (fresh) {
      set_grid_ray(S.bvh_grid.lo, S.bvh_grid.cell, S.bvh_grid.inv_cell, o, d, 0.0f, 0.0f);
      top_ray[0] = ig.x; top_ray[kBlock] = ig.y; top_ray[2 * kBlock] = ig.z;
      top_ray[3 * kBlock] = cgn.x; top_ray[4 * kBlock] = cgn.y; top_ray[5 * kBlock] = cgn.z;
      top_ray[6 * kBlock] = cgf.x; top_ray[7 * kBlock] = cgf.y; top_ray[8 * kBlock] = cgf.z;
    } else {
      ig = mk3(top_ray[0], top_ray[kBlock], top_ray[2 * kBlock]);
      cgn = mk3(top_ray[3 * kBlock], top_ray[4 * kBlock], top_ray[5 * kBlock]);
      cgf = mk3(top_ray[6 * kBlock], top_ray[7 * kBlock], top_ray[8 * kBlock]);
      sel = SlabSel{slab_sel(ig.x), slab_sel(ig.y), slab_sel(ig.z)};
    }
  };
  auto pop_next = [&]() -> int {
    for (;;) {
      if (st.sp == 0) return kRayDone;
      const int v = st.pop();
      if (v != kExitInstance) return v;
      to_top_level(false);
    }
  };
  const uint32_t prio_gen = (blockIdx.x * 4u) / gridDim.x;   // (four blocks per CU here)
  uint32_t prio_round = 0;
  for (;;) {
    if (rays.own_full >= 128u) rotate_priority(prio_gen + prio_round++);
    // ---- refill ----
    const unsigned long long idle = __ballot(!open);
    const int n_idle = __popcll(idle);
    if (!exhausted && n_idle >= kTlRefill) {
      const uint32_t next_ray = rays.ray_at(seq + (uint32_t)__popcll(idle & lanes_below));
      if (!open && next_ray < total) {
        if (src.load(next_ray, o, d, tmin, tmax)) {
          ray = next_ray;
          best = HitRecord{tmax, 0.0f, 0.0f, kNone, 0u, kNone};
          if (COUNT) tally.rays += 1;
          if (S.n_world_tris == 0 || !ray_is_finite(o, d)) {
            sink.store(ray, best);
          } else {
            st.sp = 0;
            to_top_level(true);
            cur = kTlLdsTop ? kBvhTopFlag : 0;   // the top level's root (slot 0 of the staged table)
            open = true;
#ifdef GLZ_WAVE_TIMES
            tl_rays += 1;
#endif
          }
        }
      }
      seq += (uint32_t)n_idle;
      exhausted = rays.ray_at(seq) >= total;
    }
    if (__ballot(open) == 0ull) {
      if (exhausted) break;
      continue;
    }
    // ---- inner-node phase (either level) ----
    for (;;) {
      const bool at_node = cur >= 0 && cur < kExitInstance;
      if (__ballot(at_node) == 0ull) break;
#ifdef GLZ_WAVE_TIMES
      if (lane == 0) tl_niter += 1;
      if (at_node) { if (cur_inst == kNone) tl_top += 1; else tl_mesh += 1; }
#endif
      if (at_node) {
        if (COUNT) tally.nodes += 1;   // node visits of either level
        // the top level's first kBvhTopNodes nodes come out of the block's LDS copy (`cur` = kBvhTopFlag | slot), like the flattened tracer's
        u32x4 w0, w1, w2, w3;
        if (kTlLdsTop && (cur & kBvhTopFlag)) {
          LdsNodePtr np = top_lds + 4 * (cur & 0xFFFF);
          w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3];
        } else {
          const u32x4* np = reinterpret_cast<const u32x4*>(nodes + nbase + (uint32_t)cur);
          w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3];
        }
        uint32_t k0 = box_key(w0.x, w0.y, w0.z, w3.x, 0u, sel, ig, cgn, cgf, tmin, best.t), k1 = box_key(w0.w, w1.x, w1.y, w3.y, kKeyChild, sel, ig, cgn, cgf, tmin, best.t);
        uint32_t k2 = box_key(w1.z, w1.w, w2.x, w3.z, 2u * kKeyChild, sel, ig, cgn, cgf, tmin, best.t), k3 = box_key(w2.y, w2.z, w2.w, w3.w, 3u * kKeyChild, sel, ig, cgn, cgf, tmin, best.t);
        sort2(k0, k1); sort2(k2, k3); sort2(k0, k2); sort2(k1, k3); sort2(k1, k2);
        int* links = link_scratch + lane;
        links[0] = (int)w3.x; links[64] = (int)w3.y; links[128] = (int)w3.z; links[192] = (int)w3.w;
        const uint32_t link_base = (uint32_t)(uintptr_t)(LdsIntPtr)links;   // the wave's area is 1 KB aligned: bits 8..9 are the child's
        const int l0 = sorted_link(link_base, k0), l1 = sorted_link(link_base, k1), l2 = sorted_link(link_base, k2), l3 = sorted_link(link_base, k3);
        if (k0 == 0xFFFFFFFFu) {
          cur = pop_next();
        } else {
          if (__ballot(st.sp + 3 > kLdsStack) == 0ull) {   // wave-uniform: every lane stays inside the LDS part of its stack (no spill branches)
            if (k3 != 0xFFFFFFFFu) { st.lds[st.sp * kBlock] = l3; ++st.sp; }
            if (k2 != 0xFFFFFFFFu) { st.lds[st.sp * kBlock] = l2; ++st.sp; }
            if (k1 != 0xFFFFFFFFu) { st.lds[st.sp * kBlock] = l1; ++st.sp; }
          } else {
            if (k3 != 0xFFFFFFFFu) st.push(l3);
            if (k2 != 0xFFFFFFFFu) st.push(l2);
            if (k1 != 0xFFFFFFFFu) st.push(l1);
          }
          cur = l0;
        }
      }
      // (Instance entries and triangle tests with a quorum each -- 8 / 16 / 24 lanes for entries, 24 / 32 for triangles -- so that neither kind of
      // leaf work runs for a handful of lanes: forest x 200 0.803 -> 0.816 ... 0.838 ms per launch, x 2 000 1.082 -> 1.106 ... 1.129: slower.)
      if (__popcll(__ballot(cur < 0)) >= GLZ_TL_LEAF_QUORUM) break;
    }
    // ---- leaf phase: an instance to enter (top level) or triangles to test (inside an instance) ----
    if (cur < 0) {
      if (cur_inst == kNone) {
        cur_inst = (uint32_t)~cur;
        const TlasInstance* ti = instances + cur_inst;
        const float4* q = reinterpret_cast<const float4*>(ti->w2o);
        const float4 r0 = q[0], r1 = q[1], r2 = q[2];
        // object-space ray (a point and a vector through the 3 x 4 matrix): only the box tests see it
        const vec3 oo = mk3(((r0.x * o.x + r0.y * o.y) + r0.z * o.z) + r0.w, ((r1.x * o.x + r1.y * o.y) + r1.z * o.z) + r1.w, ((r2.x * o.x + r2.y * o.y) + r2.z * o.z) + r2.w);
        const vec3 dd = mk3((r0.x * d.x + r0.y * d.y) + r0.z * d.z, (r1.x * d.x + r1.y * d.y) + r1.z * d.z, (r2.x * d.x + r2.y * d.y) + r2.z * d.z);
        st.push(kExitInstance);
        nbase = ti->node_base;
        // The slack the build computed covers ray origins inside the scene's bounds; the rounding of oo grows with |o|, wherever the
        // ray starts (a camera far outside a small instanced scene): 32 eps |W2O|_inf |o|_1 on top, in object units like the rest.
        const float slack = ti->slack + (3.8146973e-6f * ti->w2o_norm) * ((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z));
        set_grid_ray(ti->grid.lo, ti->grid.cell, ti->grid.inv_cell, oo, dd, slack, 1.0f);
        cur = 0;   // the mesh's root
      } else {
        const TlasInstance* ti = instances + cur_inst;
        // The mesh's leaf record (types.h BvhQuad; its hierarchy was built over the mesh under the identity transform, so the
        // vertices are the object-space ones): one 64-byte line for one triangle or two.  The world triangles are exactly what
        // k_world_tris builds -- points through o2w -- four of them for a pair instead of six.
        const float4* qp = reinterpret_cast<const float4*>(S.bvh_quads + ti->quad_base + (uint32_t)~cur);
        const float4 r0 = qp[0], r1 = qp[1], r2 = qp[2], r3 = qp[3];
        const uint32_t id0 = __float_as_uint(r0.w), qflags = __float_as_uint(r2.w), slot0 = ti->tri_base + __float_as_uint(r3.w);
        const bool pair = (qflags & kTriHasPartner) != 0u;
        if (COUNT) tally.tris += pair ? 2 : 1;
        const vec3 w0 = xform_point(ti->o2w, mk3(r0.x, r0.y, r0.z)), w1 = xform_point(ti->o2w, mk3(r1.x, r1.y, r1.z));
        const vec3 w2 = xform_point(ti->o2w, mk3(r2.x, r2.y, r2.z)), w3 = xform_point(ti->o2w, mk3(r3.x, r3.y, r3.z));
        const RayShear rs = ray_shear(d);
        const QuadHit qh = ray_quad(rs, make_float4(w0.x, w0.y, w0.z, 0.0f), make_float4(w1.x, w1.y, w1.z, 0.0f), make_float4(w2.x, w2.y, w2.z, 0.0f),
                                    make_float4(w3.x, w3.y, w3.z, 0.0f), pair, o, tmin);
        const uint32_t swapped = (qflags & kQuadSwapped) ? 1u : 0u;
        bool finished = false;
#pragma nounroll
        for (uint32_t which = 0; which < 2u; ++which) {   // the leaf's first triangle, then its partner (the order the 48-byte records were walked in)
          const bool is_b = (which ^ swapped) != 0u;
          const float t = is_b ? qh.t[1] : qh.t[0], u = is_b ? qh.u[1] : qh.u[0], v = is_b ? qh.v[1] : qh.v[0];
          if ((is_b ? qh.ok[1] : qh.ok[0]) && t < tmax) {
            const uint32_t world_id = ti->world_base + id0 + which, slot = slot0 + which;
            const bool better = best.leaf == kNone ? true : (t < best.t || (t == best.t && world_id < best.world_id));
            if (better && (ti->non_opaque == 0u || alpha_test_instance(S, slot, ti->instance, u, v))) {
              best = HitRecord{t, u, v, slot, ti->instance, world_id};
              finished = ANY;
            }
          }
        }
        cur = finished ? kRayDone : pop_next();
      }
    }
    // ---- retire ----
    if (open && cur == kRayDone) {
      if (COUNT) tally.hits += best.leaf != kNone;
      sink.store(ray, best);
      open = false;
    }
  }
  __builtin_amdgcn_s_setprio(0);
#ifdef GLZ_WAVE_TIMES
  if (!ANY) {
    atomicAdd(&g_tl_stats[0], tl_rays); atomicAdd(&g_tl_stats[1], tl_top); atomicAdd(&g_tl_stats[2], tl_mesh); atomicAdd(&g_tl_stats[3], tl_enter);
    atomicAdd(&g_tl_stats[4], tl_tris); atomicAdd(&g_tl_stats[5], tl_niter); atomicAdd(&g_tl_stats[6], tl_liter);
  }
#endif
}

__device__ __forceinline__ void flush_counters(TraceCounters* c, bool shadow, TraceTally t) {
  // wave-level reduction first, one atomic per wave and counter (Guideline 12)
  for (int off = 32; off > 0; off >>= 1) {
    t.rays += __shfl_down(t.rays, off);
    t.nodes += __shfl_down(t.nodes, off);
    t.tris += __shfl_down(t.tris, off);
    t.hits += __shfl_down(t.hits, off);
    t.fresh += __shfl_down(t.fresh, off);
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* ph = shadow ? c->phase + 6 : c->phase;
    atomicAdd(&ph[0], t.node_iters); atomicAdd(&ph[1], t.node_lanes); atomicAdd(&ph[2], t.leaf_iters); atomicAdd(&ph[3], t.leaf_lanes);
    atomicAdd(&ph[4], t.refill_iters); atomicAdd(&ph[5], t.refill_lanes);
  }
  if ((threadIdx.x & 63) == 0) {
    if (shadow) {
      atomicAdd(&c->shadow_rays, t.rays); atomicAdd(&c->shadow_nodes, t.nodes); atomicAdd(&c->shadow_tris, t.tris);
    } else {
      atomicAdd(&c->closest_rays, t.rays); atomicAdd(&c->closest_nodes, t.nodes); atomicAdd(&c->closest_tris, t.tris);
      atomicAdd(&c->hits, t.hits);
      atomicAdd(&c->fresh, t.fresh);
    }
  }
}

// a counting kernel's per-thread texture / light tallies (DeviceScene::tex_counter) -> the launch's counters: one atomic per wave and
// tally; every lane of the wave must get here
__device__ __forceinline__ void flush_tex_tallies(unsigned long long* dst, const unsigned long long* t) {
  unsigned long long a = t[0], b = t[1], c = t[2], d = t[3];
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); d += __shfl_down(d, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(dst, a); atomicAdd(dst + 1, b); atomicAdd(dst + 2, c); atomicAdd(dst + 3, d);
  }
}

// persistent launch geometry: every wave of the grid is one independent tracer
// (XCD-aware numbering -- the blocks with b % 8 == x, which share an L2, taking one contiguous run of groups / pixels each,
// cdna_hip_programming.md T1 -- measured slower for both kernels: k_trace 0.588 -> 0.621 ms, k_shade 0.348 -> 0.357 ms, a 1/8
// share 0.162 -> 0.188 ms.  Neighbouring regions differ in cost; dealing them round-robin over the XCDs balances that, and
// the L2s' hit rates are not what bounds either kernel.)
__device__ __forceinline__ uint32_t wave_index() { return blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t wave_count() { return gridDim.x * (kBlock / 64); }

// ---------------------------------------------------------------------------------------------
// closest-hit phase of k_trace: path_trace.rgen:143-169
// ---------------------------------------------------------------------------------------------
// (k_path with a wave's 64 pixels dealt from 4 ... 64 different groups, to average the groups' persistent cost differences: it does at a
// half-empty machine and not at a 1/8 share, where the coherence lost costs more: EXPERIMENTS.md.)
// The camera ray of a new path through the jittered pixel (ray_origin / ray_dir, path_trace.rgen:47-73).  Two callers, the same
// operations in the same order: the traversal kernel's refill for a pixel whose path is new and has no ray yet (the first launch after
// a restart, the direct-light integrator), and the shading code for a path that has just ended, with the NEXT launch's pixel offset.
__device__ __forceinline__ void camera_ray(const LaunchArgs& A, const FrameData& F, PixelId px, float off_x, float off_y, vec3& origin, vec3& direction) {
  const float pxf = (float)px.x + off_x, pyf = (float)px.y + off_y;
  const float ndcx = -1.0f + 2.0f * (pxf / F.scene_size[0]), ndcy = -1.0f + 2.0f * (pyf / F.scene_size[1]);
  const float* c2w = A.cam.camera2world;
  const float* s2c = A.cam.screen2camera;
  const float ortho = gl_step(0.5f, F.camera_persp ? 0.0f : 1.0f), persp = gl_step(0.5f, F.camera_persp ? 1.0f : 0.0f);
  const float ox = ndcx * ortho, oy = ndcy * ortho;
  origin = mk3((c2w[0] * ox + c2w[4] * oy) + c2w[12], (c2w[1] * ox + c2w[5] * oy) + c2w[13], (c2w[2] * ox + c2w[6] * oy) + c2w[14]);
  const float fx = ndcx * persp, fy = ndcy * persp;
  const vec3 target = mk3(((s2c[0] * fx + s2c[4] * fy) + s2c[8]) + s2c[12], ((s2c[1] * fx + s2c[5] * fy) + s2c[9]) + s2c[13],
                          ((s2c[2] * fx + s2c[6] * fy) + s2c[10]) + s2c[14]);
  const vec3 nt = normalize3(target);
  const float dx = (c2w[0] * nt.x + c2w[4] * nt.y) + c2w[8] * nt.z, dy = (c2w[1] * nt.x + c2w[5] * nt.y) + c2w[9] * nt.z;
  const float dz = (c2w[2] * nt.x + c2w[6] * nt.y) + c2w[10] * nt.z, dw = (c2w[3] * nt.x + c2w[7] * nt.y) + c2w[11] * nt.z;
  const float inv = 1.0f / sqrtf(((dx * dx + dy * dy) + dz * dz) + dw * dw);   // normalize() of the vec4
  direction = mk3(dx * inv, dy * inv, dz * inv);
}
// ray_o.w of a pixel is the bounce its path is at; 0 = a new path.  +0.0: the camera ray is still to be made (by the refill below);
// -0.0 (kPregenBounce): the shading code of the previous launch has made it already (shade_pixel) and ray_o / ray_d hold it.  Both
// compare equal to 0.0f, which is all the shading code asks.
constexpr uint32_t kPregenBounceBits = 0x80000000u;
struct ClosestSource {
  const LaunchArgs& A;
  const FrameData& F;   // the launch's constants (k_trace: A.frame; k_path: one entry of its batch)
  TraceTally& tally;
  uint32_t base;        // ray i is local pixel base + i (k_trace: 0; k_path: the first pixel of the wave's group)
  // ray generation / resume for local pixel `lid`
  // (Dealing the rays of a group from 4, 16 or 64 different tiles instead of one row of one tile -- to level the waves of a small
  // share, whose ends spread from 60 (median) to 105 us -- changes nothing: the spread is not regional, a wave is as slow as the
  // longest dependent chain among its 64 rays.  Median and end of the phase moved by +3 ... +8 % with the coherence lost.)
  __device__ __forceinline__ bool load(uint32_t i, vec3& origin, vec3& direction, float& tmin, float& tmax) {
    const uint32_t lid = base + i;
    if (lid >= A.map.n_local_pixels) return false;
    const float4 ro = A.st.ray_o[lid], rd = A.st.ray_d[lid];
    // A refill runs with the 16 - 24 lanes that were idle, and a quarter of the pixels start a new path in every launch: making their
    // camera rays here -- ~170 VALU instructions with four divisions and two square roots, at a quarter of the lanes, in nearly every
    // refill of a kernel that is bound by VALU issue -- was 8 % of k_trace's instructions.  The shading code makes them now where the
    // paths end (whole waves of misses after k_shade's regrouping), and the branch below is taken by the launch after a restart only.
    // (Where the pixel is -- a division by the tiles per row -- only matters here: the pixels of an edge tile that lie outside the image
    // are never written by anybody, stay at +0.0 and come this way in every launch.)
    if (F.direct_only || __float_as_uint(ro.w) == 0u) {
      const PixelId px = pixel_of(A.map, lid);
      if (!px.active) return false;
      tally.fresh += 1;
      camera_ray(A, F, px, F.pixel_offset[0], F.pixel_offset[1], origin, direction);
      A.st.ray_o[lid] = make_float4(origin.x, origin.y, origin.z, ro.w);
      A.st.ray_d[lid] = make_float4(direction.x, direction.y, direction.z, rd.w);
    } else {
      if (ro.w == 0.0f) tally.fresh += 1;
      origin = mk3(ro.x, ro.y, ro.z);
      direction = mk3(rd.x, rd.y, rd.z);
    }
    tmin = 0.0001f;
    tmax = INFINITY;
    return true;
  }
};
struct ClosestSink {
  const LaunchArgs& A;
  __device__ __forceinline__ void store(uint32_t lid, const HitRecord& h) {
    A.st.hit[lid] = make_float4(h.leaf == 0xFFFFFFFFu ? INFINITY : h.t, h.u, h.v, __uint_as_float(h.leaf));
  }
};

// update_count() + update_result() of path_trace.rgen:119-133 for one pixel; `cum` = cumulative[lid] as read before
__device__ __forceinline__ void accumulate_pixel(const LaunchArgs& A, uint32_t lid, vec3 c, bool add, bool update, float exposure, float4 cum) {
  cum.w += 1.0f;
  if (update) {
    if (add) { cum.x += c.x; cum.y += c.y; cum.z += c.z; }
    A.st.result[lid] = make_float4(cum.x * exposure / cum.w, cum.y * exposure / cum.w, cum.z * exposure / cum.w, 1.0f);
  }
  A.st.cumulative[lid] = cum;
}
__device__ __forceinline__ void accumulate_pixel(const LaunchArgs& A, uint32_t lid, vec3 c, bool add, bool update, float exposure) {
  float4 cum = A.st.cumulative[lid];
  cum.w += 1.0f;
  if (update) {
    if (add) { cum.x += c.x; cum.y += c.y; cum.z += c.z; }
    A.st.result[lid] = make_float4(cum.x * exposure / cum.w, cum.y * exposure / cum.w, cum.z * exposure / cum.w, 1.0f);
  }
  A.st.cumulative[lid] = cum;
}

// Shadow-ray queue: 8 sub-queues ("shards"), shard = blockIdx % 8.  Blocks b and b+8 are observed to land on
// the same XCD, so a shard's counter line tends to stay in one XCD's L2; more importantly eight counters on
// separate 128-byte lines take eight times the append rate of one word (MI355X_MICROARCH.md, row `dequeue`).
// A shard only receives entries from its own blocks, so its capacity ceil(blocks/8) * kBlock can never overflow.
__device__ __forceinline__ uint32_t queue_capacity(uint32_t n_local_pixels) {
  const uint32_t blocks = (n_local_pixels + kBlock - 1) / kBlock;
  return ((blocks + kQueueShards - 1) / kQueueShards) * kBlock;
}
// Appends the lanes with `push` set: one atomic per wave (ballot + popcount); the wave's entries are contiguous so
// the three float4 stores stay coalesced.  Returns the entry index in the queue arrays.
__device__ __forceinline__ uint32_t queue_slot(uint32_t* counters, uint32_t n_local_pixels, bool push) {
  const unsigned long long m = __ballot(push);
  uint32_t slot = 0;
  if (push) {
    const uint32_t shard = ((blockIdx.x * blockDim.x + threadIdx.x) / kBlock) % kQueueShards;   // by 256-pixel segment, whatever the block size (queue_capacity)
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counters + shard * kCounterStride, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    slot = shard * queue_capacity(n_local_pixels) + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  return slot;
}

// ---------------------------------------------------------------------------------------------
// One pixel of path_trace.rgen:170-237 minus the two traceRayEXT calls, with raytrace_hit.rchit:30-71 in front: what k_shade
// runs for the pixel at its sorted slot and what k_path (the per-wave launch loop of a small tile share) runs for each of a
// wave's 64 pixels.  `hr` is the closest-hit record of this launch, `queue.slot(push)` hands out the shadow-queue entry (all
// lanes that get this far call it together).
// ---------------------------------------------------------------------------------------------
struct SharedQueue {   // k_shade: the rank's sharded queue in HBM, drained by the next k_trace
  const LaunchArgs& A;
  __device__ __forceinline__ uint32_t slot(bool push) { return queue_slot(A.st.queue_count + A.shade_set * kQueueSetWords, A.map.n_local_pixels, push); }
};
// LOD: the build with the texture level of detail (FrameData::lod_mode != 0); the default build carries none of its code
// Where the pixel's next path state goes.  DirectState: straight into the state arrays (k_path: a wave's 64 pixels are neighbours, every
// store is whole lines).  StagedState (k_shade, whose threads shade pixels in regrouped order): kept in registers, the kernel writes
// them after the block's last barrier, transposed through LDS so that thread i stores pixel i's state.
// (k_shade: every thread storing its pixel's state itself 0.346 ms, the path state through the LDS transpose 0.322, the accumulator
// update through it too 0.329 -- so only the path state is staged.)
#ifdef GLZ_SECTION_TIMES   // tools/gpu_shade_sections.py: shader clocks between the stamps of shade_pixel, per wave (k_shade only)
#define GLZ_SHADE_STAMP(k) out.stamp(k)
#else
#define GLZ_SHADE_STAMP(k) do { } while (0)
#endif
struct DirectState {
  const LaunchArgs& A;
  __device__ __forceinline__ void stamp(int) {}
  __device__ __forceinline__ void ray_o(uint32_t lid, float4 v) { A.st.ray_o[lid] = v; }
  __device__ __forceinline__ void ray_d(uint32_t lid, float4 v) { A.st.ray_d[lid] = v; }
  __device__ __forceinline__ void imp(int q, uint32_t lid, float4 v) { A.st.imp[q][lid] = v; }
  __device__ __forceinline__ float4 read_imp(int q, uint32_t lid) const { return A.st.imp[q][lid]; }
  __device__ __forceinline__ void accumulate(uint32_t lid, vec3 c, bool add, bool update, float exposure) { accumulate_pixel(A, lid, c, add, update, exposure); }
};
struct StagedState {
#ifdef GLZ_SECTION_TIMES
  unsigned long long sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sec_last = 0;
  __device__ __forceinline__ void stamp(int k) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();   // (no wait: a section is charged what the wave waited for in it, not what it asked for)
    sec[k] += now - sec_last;
    sec_last = now;
  }
#else
  __device__ __forceinline__ void stamp(int) {}
#endif
  float4 ro, rd, im[4];
  uint32_t mask = 0;   // 1: ro, 2: rd, 4: im
  __device__ __forceinline__ void ray_o(uint32_t, float4 v) { ro = v; mask |= 1u; }
  __device__ __forceinline__ void ray_d(uint32_t, float4 v) { rd = v; mask |= 2u; }
  __device__ __forceinline__ void imp(int q, uint32_t, float4 v) { im[q] = v; mask |= 4u; }
  const LaunchArgs* A = nullptr;   // the accumulator is updated where the pixel is shaded
  // The importance the pixel arrived with: k_shade's prologue reads the block's 4 x 4 KB in whole lines, in pixel order, into LDS, and the
  // (up to three) reads of shade_pixel come from there -- read where they are used, by threads in regrouped order, they were twelve
  // scattered 16-byte accesses per pixel on the vector-memory path, which is what bounds the kernel.
  LdsNodePtr lds_imp = nullptr;   // &s_imp[pixel's index in the block] (a pointer that keeps its address space: ds_read_b128); component q at [q * kShadeBlockPixels]
  static constexpr uint32_t kShadeBlockPixels = 256;
  __device__ __forceinline__ float4 read_imp(int q, uint32_t) const {
    const u32x4 v = lds_imp[q * kShadeBlockPixels];
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  __device__ __forceinline__ void accumulate(uint32_t lid, vec3 cc, bool add, bool update, float exposure) { accumulate_pixel(*A, lid, cc, add, update, exposure); }
};
// Returns 0 when the pixel's next path state has been written (or, with the direct-light integrator, is not needed), 1 / 2 when the path
// has ENDED and its reset is left to shade_pixel below: 1 = only ray_o is to be written (a miss: ray_d keeps its flag), 2 = ray_o and
// ray_d, the latter with the flag `end_w`.
template <bool LOD, class Queue, class State>
__device__ __forceinline__ int shade_pixel_body(const LaunchArgs& A, const DeviceScene& S, const FrameData& F, uint32_t lid, PixelId px, float4 ro, float4 rd, float4 hr, Queue& queue,
                                                State& out, float& end_w) {
  const bool fresh = F.direct_only || ro.w == 0.0f;
  float bounce = F.direct_only ? 0.0f : ro.w;
  const vec3 direction = mk3(rd.x, rd.y, rd.z);
  // The path's importance (16 floats) is read where it is used -- the radiance of the light sample, the roulette, the final product --
  // instead of once up front: held through texture fetches, light sampling and the two BSDF calls it set the kernel's register peak.
  // The re-reads hit the lines the first read brought in.
  auto load_importance = [&]() {
    asm volatile("" ::: "memory");   // a fresh read every time: merged with an earlier one the values would stay in registers in between
    Spec imp;
    if (fresh) {
      imp = spec_set(1.0f);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = out.read_imp(q, lid);
        imp.w[4 * q] = v.x; imp.w[4 * q + 1] = v.y; imp.w[4 * q + 2] = v.z; imp.w[4 * q + 3] = v.w;
      }
    }
    return imp;
  };
  const uint32_t leaf = __float_as_uint(hr.w);
  if (leaf == 0xFFFFFFFFu) {
    // miss: optional sky radiance, path reset (path_trace.rgen:170-179)
    uint32_t flags = 0;
    vec3 c = mk3(0.0f, 0.0f, 0.0f);
    if ((bounce == 0.0f || rd.w == 1.0f) && S.sky.tex_id > 0) {
      const vec3 w = normalize3(xform_dir(S.sky.world2obj, direction));   // sky_radiance, :75-82
      const float phi = glz_atan2f(w.y, w.x), theta = glz_acosf(w.z);
      const vec3 texel = texture_rgb(S, S.sky.tex_id, vec2{phi * kInv2Pi, theta * kInvPi});
      c = spec_to_rgb(spec_mul(load_importance(), from_illuminant_color(texel)));
      flags = kFlagUpdate;
    }
    out.accumulate(lid, c, true, flags != 0, F.exposure);
    end_w = rd.w;
    return F.direct_only ? 0 : 1;   // RESET_PATH
  }
  // ---- closest-hit shader (raytrace_hit.rchit:30-71), inputs from the 128-byte per-leaf shading record ----
  const float4* rec = S.shade_tris + 8u * (size_t)leaf;
  const float4 va0 = rec[0], va1 = rec[1], vb0 = rec[2], vb1 = rec[3], vc0 = rec[4], vc1 = rec[5], dn = rec[6], du = rec[7];
  uint32_t material_id = __float_as_uint(dn.w), xf_bits = __float_as_uint(du.w);
  if (S.two_level) {   // the record is per OBJECT triangle: material and transform are the instance's
    const RTInstance in = S.instances[A.st.hit_inst[lid]];
    material_id = in.material_id;
    xf_bits = in.transform_id | (S.xf_identity[in.transform_id] ? 0x80000000u : 0u);
  }
  const float b0 = 1.0f - hr.y - hr.z, b1 = hr.y, b2 = hr.z;
  vec3 point = (mk3(va0.x, va0.y, va0.z) * b0 + mk3(vb0.x, vb0.y, vb0.z) * b1) + mk3(vc0.x, vc0.y, vc0.z) * b2;
  const vec2 uv = vec2{(va1.z * b0 + vb1.z * b1) + vc1.z * b2, (va1.w * b0 + vb1.w * b1) + vc1.w * b2};
  vec3 ng = mk3(dn.x, dn.y, dn.z), dpdu = mk3(du.x, du.y, du.z);   // dpdv is transformed by the reference but never read afterwards
  vec3 ns = (mk3(va0.w, va1.x, va1.y) * b0 + mk3(vb0.w, vb1.x, vb1.y) * b1) + mk3(vc0.w, vc1.x, vc1.y) * b2;
  const MatScalars mat = load_material(&S.materials[material_id]);
  GLZ_SHADE_STAMP(0);   // hit record -> shading record -> material scalars
  // ---- texture level of detail by ray cones (build-defined, off by default: the reference's stages sample level 0) ----
  // The cone of a camera path starts cone_width0 wide and widens by cone_spread per unit of distance along the whole path;
  // at a hit the footprint on the surface is width / |cos|, and a texture of W x H texels over a triangle with texture-space
  // area A_uv and world area A_w is minified by sqrt(A_uv W H / A_w) texels per unit length:
  // level = 0.5 log2(A_uv / A_w * width^2 / cos^2) + 0.5 log2(W H)      (Akenine-Moeller et al., ray cones)
  // lod mode 2 (anisotropic): the footprint is cone_w across and cone_w / |cos| along the projection m of the ray direction onto the
  // surface; taps = ceil(min(1 / |cos|, 16)) probes along m, each at the level of a footprint cone_w / |cos| / taps wide; m written in
  // the triangle's edges (least squares: it lies in their plane) gives the footprint's long axis in texture space.
  TexFootprint fp{kNoLod, 0.0f, 0.0f, 1u};
  float cone_w = 0.0f;
  if constexpr (LOD) {
    cone_w = (fresh ? F.cone_width0 : A.st.cone[lid]) + F.cone_spread * hr.x;
    vec3 e1 = mk3(vb0.x, vb0.y, vb0.z) - mk3(va0.x, va0.y, va0.z), e2 = mk3(vc0.x, vc0.y, vc0.z) - mk3(va0.x, va0.y, va0.z);
    vec3 n = mk3(dn.x, dn.y, dn.z);
    if (!(xf_bits >> 31)) {
      const TransformPair* xf = &S.transforms[xf_bits & 0x7FFFFFFFu];
      e1 = xform_dir(xf->o2w, e1);
      e2 = xform_dir(xf->o2w, e2);
      n = xform_tdir(xf->w2o, n);
    }
    const vec3 cr = cross3(e1, e2);
    const float area2 = sqrtf(dot3(cr, cr));
    const float uva2 = fabsf((vb1.z - va1.z) * (vc1.w - va1.w) - (vc1.z - va1.z) * (vb1.w - va1.w));
    const float nn = dot3(n, n), nd = dot3(n, direction);
    const float cosv = fabsf(nd) / sqrtf(nn);
    const float x = ((uva2 / area2) * (cone_w * cone_w)) / (cosv * cosv);
    if (x >= 1.17549435e-38f && x <= 3.4e38f) {
      fp.lod_base = 0.5f * glz_log2f(x);
      if (F.lod_mode == 2u) {
        float ratio = 1.0f / cosv;
        ratio = ratio < 16.0f ? ratio : 16.0f;
        const float taps = -glz_floorf(-ratio);   // ceil
        const vec3 m = direction - n * (nd / nn);
        const float mm = dot3(m, m);
        if (taps > 1.0f && mm > 0.0f) {
          const float g11 = dot3(e1, e1), g12 = dot3(e1, e2), g22 = dot3(e2, e2), r1 = dot3(m, e1), r2 = dot3(m, e2);
          const float det = g11 * g22 - g12 * g12;
          const float ca = (r1 * g22 - r2 * g12) / det, cb = (r2 * g11 - r1 * g12) / det;
          const float len = (cone_w / cosv) / sqrtf(mm);
          const float du = (ca * (vb1.z - va1.z) + cb * (vc1.z - va1.z)) * len;
          const float dv = (ca * (vb1.w - va1.w) + cb * (vc1.w - va1.w)) * len;
          if (fabsf(du) <= 3.4e38f && fabsf(dv) <= 3.4e38f) {
            fp.du = du;
            fp.dv = dv;
            fp.taps = (uint32_t)taps;
            fp.lod_base = fp.lod_base - glz_log2f(taps);
          }
        }
      }
    }
  }
  if (mat.normal != 0) {
    const vec4 tx = texture2d_lod(S, mat.normal, uv.x, uv.y, fp);
    Frame old;
    old.s = normalize3(dpdu);
    old.n = ns;
    old.t = normalize3(cross3(old.n, old.s));
    ns = normalize3(to_world(mk3(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f), old));
    ns = ns * gl_sign(dot3(ng, ns));
  }
  if (!(xf_bits >> 31)) {
    // object -> world.  Skipped for an exact identity transform: m*x with m = I reproduces x bit for bit
    // (x*1 + y*0 + z*0 + 0 for finite coordinates), so the result is unchanged and ~25 scalar loads are saved.
    const float4* xq = reinterpret_cast<const float4*>(&S.transforms[xf_bits & 0x7FFFFFFFu]);
    const float4 m0 = xq[0], m1 = xq[1], m2 = xq[2], m3 = xq[3], w0 = xq[4], w1 = xq[5], w2 = xq[6];
    const float o2w[16] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w, m3.x, m3.y, m3.z, m3.w};
    const float w2o[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    point = xform_point(o2w, point);
    dpdu = xform_point(o2w, dpdu);   // transformed as a point, w = 1 (Q8)
    ng = xform_tdir(w2o, ng);
    ns = xform_tdir(w2o, ns);
  }
  (void)ng;
  // ---- raygen continues (path_trace.rgen:180-237) ----
  uint32_t rng = pcg(__float_as_uint((float)F.seed) ^ pcg(__float_as_uint((float)px.x) ^ pcg(__float_as_uint((float)px.y))));   // :143, Q11
  SurfacePoint P;
  P.woW = -direction;
  P.uv = uv;
  P.frame = make_frame(dpdu, ns);
  P.mat = mat;
  fetch_material_textures(S, P, fp);
  GLZ_SHADE_STAMP(1);   // normal map, transform, frame, the material's textures
  float spec_flag;
  float imp_lum = 0.0f;      // luminance of the importance, taken when the light-sampling block reads it: the roulette needs nothing else of it
  bool have_lum = false;
  if (mat.is_specular == 0) {
    // direct_light(), :84-117
    const uint32_t li = (uint32_t)gl_min(rand01(rng) * (float)F.lights_no, (float)(F.lights_no - 1u));
    vec3 xi;
    xi.x = rand01(rng); xi.y = rand01(rng); xi.z = rand01(rng);
    LightSample ls;
    ls.pdf = 0.0f;
    sample_light(S, li, point, xi, F.scene_radius, ls);
    GLZ_SHADE_STAMP(2);   // light sample
    vec3 c = mk3(0.0f, 0.0f, 0.0f);
    uint32_t flags = kFlagUpdate;
    vec3 sh_dir = mk3(0.0f, 0.0f, 0.0f);
    float sh_tmax = 0.0f;
    if (ls.pdf > 0.0f) {
      const float xi_b = rand01(rng);
      Spec value = spec_set(0.0f);
      const float bpdf = bsdf_eval(S, P, ls.wiW, xi_b, value);
      if (bpdf > 0.0f) {
        // weight_light = (1 or 0) * |cos| / pdf; radiance = value*emission*weight*lights_no*importance
        const float w_vis = 1.0f * (fabsf(dot3(ls.wiW, ns)) / ls.pdf);
        const float w_occ = 0.0f * (fabsf(dot3(ls.wiW, ns)) / ls.pdf);
        const float nl = (float)F.lights_no;
        const Spec emission = light_emission(ls);
        const Spec importance = load_importance();
        imp_lum = spec_luminance(importance);
        have_lum = true;
        Spec rad;
        float poison = 0.0f;
        GLZ_BINS {
          const float rl = value.w[i] * emission.w[i];
          rad.w[i] = ((rl * w_vis) * nl) * importance.w[i];
          poison += ((rl * w_occ) * nl) * importance.w[i];
        }
        c = spec_to_rgb(rad);
        flags |= kFlagShadow | (poison == poison ? 0u : kFlagPoison);
        sh_dir = ls.wiW;
        sh_tmax = ls.distance - 1e-3f;
      }
    }
    if (!(flags & kFlagShadow)) {
      // no light sample: the reference still adds rgb(0 * lights_no * importance), which is NaN for a non-finite importance
      const Spec importance = load_importance();
      imp_lum = spec_luminance(importance);
      have_lum = true;
      float probe = 0.0f;
      GLZ_BINS probe += 0.0f * importance.w[i];
      if (probe != probe) c = spec_to_rgb(spec_scale(importance, 0.0f * (float)F.lights_no));
    }
    // shadow-ray queue (consumed by the next launch's k_trace); pixels without a shadow ray are accumulated right here
    const bool push = (flags & kFlagShadow) != 0;
    GLZ_SHADE_STAMP(3);   // BSDF evaluation, radiance, importance read
    const uint32_t slot = queue.slot(push);
    if (push) {
      A.st.sh_o[slot] = make_float4(point.x, point.y, point.z, sh_tmax);
      A.st.sh_d[slot] = make_float4(sh_dir.x, sh_dir.y, sh_dir.z, __uint_as_float(lid));
      A.st.contrib[slot] = make_float4(c.x, c.y, c.z, __uint_as_float(flags));
    } else {
      out.accumulate(lid, c, true, true, F.exposure);
    }
    spec_flag = 0.0f;
  } else {
    out.accumulate(lid, mk3(0.0f, 0.0f, 0.0f), false, false, F.exposure);
    spec_flag = 1.0f;
  }
  if (F.direct_only) return 0;
  GLZ_SHADE_STAMP(4);   // queue entry / accumulator update
  // Russian roulette (:197-210)
  float rr_scale = 1.0f;   // importance * 1.0f is importance, bit for bit: the paths that skip the roulette multiply by it too
  if (bounce > (float)(F.pt_steps / 2u)) {
    const float kill = gl_max(0.05f, 1.0f - (have_lum ? imp_lum : spec_luminance(load_importance())));
    if (rand01(rng) < kill) {
      end_w = spec_flag;
      return 2;
    }
    rr_scale = 1.0f / (1.0f - kill);
  }
  vec3 xi;
  xi.x = rand01(rng); xi.y = rand01(rng); xi.z = rand01(rng);
  Spec value = spec_set(0.0f);
  vec3 wiW = mk3(0.0f, 0.0f, 0.0f);
  const float pdf = bsdf_sample(S, P, xi, value, wiW);   // :212-218
  if (pdf == 0.0f) {
    end_w = spec_flag;
    return 2;
  }
  float weight = fabsf(dot3(wiW, ns));
  weight /= pdf;
  const Spec importance = spec_scale(load_importance(), rr_scale);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    out.imp(q, lid, make_float4(importance.w[4 * q] * (value.w[4 * q] * weight), importance.w[4 * q + 1] * (value.w[4 * q + 1] * weight),
                                importance.w[4 * q + 2] * (value.w[4 * q + 2] * weight), importance.w[4 * q + 3] * (value.w[4 * q + 3] * weight)));
  bounce = bounce < (float)F.pt_steps ? bounce + 1.0f : 0.0f;   // :230-237
  GLZ_SHADE_STAMP(5);   // roulette, BSDF sample, new importance
  if constexpr (LOD) A.st.cone[lid] = cone_w;
  if (F.pregen && bounce == 0.0f) {   // the path has reached its last step: the next launch starts a new one (only its flag survives)
    end_w = spec_flag;
    return 2;
  }
  out.ray_o(lid, make_float4(point.x, point.y, point.z, bounce));
  out.ray_d(lid, make_float4(wiW.x, wiW.y, wiW.z, spec_flag));
  return 0;
}
// shade_pixel_body, then the reset of a path that ended (RESET_PATH, path_trace.rgen:170-179 / :197-218): ray_o.w = 0 tells the next launch
// to start a new path at this pixel.  With FrameData::pregen the new path's camera ray is made right here, from the next launch's
// pixel offset (camera_ray: the operations ClosestSource::load would run in the next launch, bit for bit), and ray_o.w = -0.0 says so --
// k_shade's regrouping puts the pixels that missed into waves of their own, so the code runs with full waves where the traversal
// kernel's refill ran it with a quarter of the lanes.
template <bool LOD, class Queue, class State>
__device__ __forceinline__ void shade_pixel(const LaunchArgs& A, const DeviceScene& S, const FrameData& F, uint32_t lid, PixelId px, float4 ro, float4 rd, float4 hr, Queue& queue,
                                            State& out) {
  float end_w = 0.0f;
  const int ended = shade_pixel_body<LOD>(A, S, F, lid, px, ro, rd, hr, queue, out, end_w);
  if (ended != 0) {
    if (F.pregen) {
      vec3 co, cd;
      camera_ray(A, F, px, F.next_pixel_offset[0], F.next_pixel_offset[1], co, cd);
      out.ray_o(lid, make_float4(co.x, co.y, co.z, __uint_as_float(kPregenBounceBits)));
      out.ray_d(lid, make_float4(cd.x, cd.y, cd.z, end_w));
    } else {
      out.ray_o(lid, make_float4(ro.x, ro.y, ro.z, 0.0f));
      if (ended == 2) out.ray_d(lid, make_float4(rd.x, rd.y, rd.z, end_w));
    }
  }
}

#ifndef GLZ_SHADE_TABLE_BYTES
#define GLZ_SHADE_TABLE_BYTES 16384
#endif
constexpr uint32_t kShadeTableBytes = GLZ_SHADE_TABLE_BYTES;   // LDS copy of the material / light / texture-descriptor tables (78 materials alone would fill it)

// ---------------------------------------------------------------------------------------------
// Shadow rays: the shadow traceRayEXT (path_trace.rgen:106-110) for the compacted queue written by k_shade,
// followed by update_count / update_result (:119-133) of the owning pixel (source / sink of k_trace's second phase).
// ---------------------------------------------------------------------------------------------
struct ShadowSource {
  const LaunchArgs& A;
  const uint32_t* start;   // prefix sums of the shard counts (kQueueShards + 1 entries)
  uint32_t cap;
  uint32_t lid;            // per-lane: owning pixel and contribution of the ray in flight
  float4 contrib;
  __device__ __forceinline__ bool load(uint32_t i, vec3& o, vec3& d, float& tmin, float& tmax) {
    uint32_t shard = 0;
#pragma unroll
    for (uint32_t k = 1; k < kQueueShards; ++k) shard += i >= start[k] ? 1u : 0u;
    const uint32_t q = shard * cap + (i - start[shard]);
    const float4 so = A.st.sh_o[q], sd = A.st.sh_d[q];
    contrib = A.st.contrib[q];
    lid = __float_as_uint(sd.w);
    o = mk3(so.x, so.y, so.z);
    d = mk3(sd.x, sd.y, sd.z);
    tmin = 0.001f;
    tmax = so.w;
    return true;
  }
};
struct ShadowSink {
  const LaunchArgs& A;
  ShadowSource& src;
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
    accumulate_pixel(A, src.lid, c, add, true, A.shadow_exposure);
  }
};

// ---------------------------------------------------------------------------------------------
// k_trace: ONE persistent traversal kernel per launch.  Every wave first works through its share of the closest-hit
// rays of launch L (ray generation / resume + traversal -> hit[lid]), then through its share of the shadow rays that
// launch L-1's k_shade queued (any-hit traversal, then update_count / update_result of the owning pixel).  The shadow
// test of a launch only gates an accumulation -- the path itself continues from k_shade's output -- so deferring it
// into the next launch's traversal changes no result, takes one kernel and one dependent drain off every launch's
// critical path, and lets waves that finish their closest-hit share early start on shadow rays instead of idling
// (strong scaling: at 1/8 of a 1080p frame per GPU the launch was 0.146 + 0.073 + 0.183 ms with three kernels).
// k_shade of launch L runs after this kernel, so the accumulations of launch L-1 land before those of launch L:
// the per-pixel order of `cum += c` is the reference's.
// Two counter sets: this kernel drains set shade_set ^ 1 and clears set shade_set for the k_shade that follows.
// ---------------------------------------------------------------------------------------------
// copies the scene's top-of-tree table (types.h kBvhTopNodes) into the block's LDS; ends with a block barrier
__device__ __forceinline__ void stage_top(const DeviceScene& S, uint4* s_top) {
  if (kLdsTop) {
#ifdef GLZ_NODE48
    const uint4* src = reinterpret_cast<const uint4*>(S.bvh_top48);
    if (threadIdx.x < kBvhTopNodes * 3) s_top[threadIdx.x] = src[threadIdx.x];
#else
    const uint4* src = reinterpret_cast<const uint4*>(S.bvh_top);
    if (threadIdx.x < kBvhTopNodes * 4) s_top[threadIdx.x] = src[threadIdx.x];
#endif
    __syncthreads();
  }
}

#ifdef GLZ_WAVE_TIMES   // tuning builds only (tools/build_variant.sh, tools/gpu_wave_times.py): when each wave of the last k_trace with closest-hit rays started, finished those and ended
#define GLZ_WAVE_STAMP(k) do { if (A.do_closest && (threadIdx.x & 63) == 0 && wave_index() < 8192u) g_wave_times[3 * wave_index() + (k)] = wall_clock64(); } while (0)
#else
#define GLZ_WAVE_STAMP(k) do { } while (0)
#endif

// The kernel's arguments, re-read: behind the empty asm the compiler no longer knows that the pointer is the one it has been loading
// from, so what follows loads the arguments it needs where it needs them (scalar loads from the kernarg segment) instead of keeping
// every pointer of LaunchArgs in SGPRs from the top of the kernel -- there are more of them than SGPRs, the overflow goes to VGPR
// lanes (v_writelane / v_readlane) and takes registers from the shading code.
typedef const __attribute__((address_space(4))) char* KernargPtr;
__device__ __forceinline__ KernargPtr reread_kernarg() {
  KernargPtr p = (KernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
}  // namespace glz
