// Wavefront path-tracing kernels for gfx950 (wave64).  One reference "launch" (one
// vkCmdTraceRaysKHR of path_trace.rgen = one path segment per pixel, raytracer.rs:553-562) is
// three kernels over the SoA path state in HBM:
//
//   k_trace_closest   ray generation / resume + closest-hit LBVH traversal      -> hit record
//   k_shade           hit attributes, light sample, BSDF eval, Russian roulette,
//                     BSDF sample, state update                                 -> shadow ray + contribution
//   k_shadow_queue    any-hit traversal of the compacted shadow-ray queue (persistent grid), update_count/update_result
//
// A wave owns one 8x8 pixel block of a 64x64 tile, so primary rays of a wave are coherent and all
// per-pixel arrays are read and written as one contiguous 1 KiB (float4) line per wave.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device/math.h"
#include "device/shading.h"
#include "device/types.h"
#include "kernels.h"

namespace glz {
using namespace dev;

constexpr int kBlock = 256;       // 4 waves
constexpr int kLdsStack = kTraversalLdsStack;   // stack entries kept in LDS per lane (18 KB per block -> 8 blocks per CU); deeper levels spill to HBM
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
// Ray / box and ray / triangle.  The triangle test is Moeller-Trumbore with the candidate accepted
// iff tmin < t < tmax, no face culling (acceleration.rs:335-345); it stands in for the driver's
// intersector ([ext]).  Ties on t are broken by the smaller world triangle id so that the result
// does not depend on traversal order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_entry(const float* __restrict__ lo, const float* __restrict__ hi, vec3 o, vec3 inv, float tmin, float tmax) {
  float t0 = tmin, t1 = tmax;
  {
    float a = (lo[0] - o.x) * inv.x, b = (hi[0] - o.x) * inv.x;
    t0 = fmaxf(t0, fminf(a, b));
    t1 = fminf(t1, fmaxf(a, b) * 1.0000005f);
  }
  {
    float a = (lo[1] - o.y) * inv.y, b = (hi[1] - o.y) * inv.y;
    t0 = fmaxf(t0, fminf(a, b));
    t1 = fminf(t1, fmaxf(a, b) * 1.0000005f);
  }
  {
    float a = (lo[2] - o.z) * inv.z, b = (hi[2] - o.z) * inv.z;
    t0 = fmaxf(t0, fminf(a, b));
    t1 = fminf(t1, fmaxf(a, b) * 1.0000005f);
  }
  return t0 <= t1 ? t0 : INFINITY;
}

__device__ __forceinline__ bool ray_triangle(const BvhTri& tr, vec3 o, vec3 d, float tmin, float& t, float& u, float& v) {
  const vec3 e1 = mk3(tr.e1[0], tr.e1[1], tr.e1[2]), e2 = mk3(tr.e2[0], tr.e2[1], tr.e2[2]);
  const vec3 pvec = cross3(d, e2);
  const float det = dot3(e1, pvec);
  if (det == 0.0f) return false;
  const float inv = 1.0f / det;
  const vec3 tvec = o - mk3(tr.v0[0], tr.v0[1], tr.v0[2]);
  u = dot3(tvec, pvec) * inv;
  if (!(u >= 0.0f && u <= 1.0f)) return false;
  const vec3 qvec = cross3(tvec, e1);
  v = dot3(d, qvec) * inv;
  if (!(v >= 0.0f && u + v <= 1.0f)) return false;
  t = dot3(e2, qvec) * inv;
  return t > tmin;
}

// raytrace_hit.rahit:24-39 -- candidates on non-opaque geometry are dropped when opacity.r < 0.5
__device__ __forceinline__ bool alpha_test(const DeviceScene& S, const BvhTri& tr, float u, float v) {
  const RTInstance in = S.instances[tr.instance];
  const uint32_t prim = tr.prim_flags & 0x7FFFFFFFu;
  const uint32_t* ix = S.indices + (in.index_offset / 3u + prim) * 3u;
  const float4 a = S.vertices[2u * ix[0] + 1u], b = S.vertices[2u * ix[1] + 1u], c = S.vertices[2u * ix[2] + 1u];
  const float w = 1.0f - u - v;
  const float tu = (a.z * w + b.z * u) + c.z * v, tv = (a.w * w + b.w * u) + c.w * v;
  return !(texture_r(S, S.materials[in.material_id].opacity, vec2{tu, tv}) < 0.5f);
}

struct HitRecord {
  float t, u, v;
  uint32_t leaf;   // index into bvh_tris, 0xFFFFFFFF = miss
};

// Per-lane traversal stack: the first kLdsStack levels in LDS (column `tid` of a [level][kBlock]
// array: every lane always hits bank tid % 32, conflict-free whatever the per-lane depth), deeper
// levels in a per-pixel HBM spill area.
struct Stack {
  int* lds;             // &s_stack[threadIdx.x]
  uint32_t* spill;      // overflow words of this lane
  int sp;
  __device__ __forceinline__ void push(int v) {
    if (sp < kLdsStack) lds[sp * kBlock] = v; else spill[sp - kLdsStack] = (uint32_t)v;
    ++sp;
  }
  __device__ __forceinline__ int pop() {
    --sp;
    return sp < kLdsStack ? lds[sp * kBlock] : (int)spill[sp - kLdsStack];
  }
};

// ANY = false: closest hit in (tmin, tmax).  ANY = true: first accepted hit terminates.
// Visit order: near child first (entry distance; ties -> child0), far child pushed.
template <bool ANY, bool COUNT>
__device__ __forceinline__ HitRecord traverse(const DeviceScene& S, vec3 o, vec3 d, float tmin, float tmax, Stack st,
                                              unsigned long long& n_nodes, unsigned long long& n_tris) {
  HitRecord best{tmax, 0.0f, 0.0f, 0xFFFFFFFFu};
  if (S.n_world_tris == 0) return best;
  uint32_t best_id = 0xFFFFFFFFu;
  const vec3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  const BvhNode* __restrict__ nodes = S.bvh_nodes;
  const BvhTri* __restrict__ tris = S.bvh_tris;
  st.sp = 0;
  int cur = 0;
  for (;;) {
    if (cur >= 0) {
      // inner node: 64 bytes = 4 x dwordx4
      const float4* np = reinterpret_cast<const float4*>(nodes + cur);
      const float4 a = np[0], b = np[1], c = np[2], e = np[3];
      if (COUNT) ++n_nodes;
      const float lo0[3] = {a.x, a.y, a.z}, hi0[3] = {b.x, b.y, b.z}, lo1[3] = {c.x, c.y, c.z}, hi1[3] = {e.x, e.y, e.z};
      const float e0 = box_entry(lo0, hi0, o, inv, tmin, best.t), e1 = box_entry(lo1, hi1, o, inv, tmin, best.t);
      const int c0 = __float_as_int(a.w), c1 = __float_as_int(b.w);
      const bool h0 = e0 < INFINITY, h1 = e1 < INFINITY;
      if (h0 && h1) {
        const bool swap = e1 < e0;
        st.push(swap ? c0 : c1);
        cur = swap ? c1 : c0;
        continue;
      }
      if (h0) { cur = c0; continue; }
      if (h1) { cur = c1; continue; }
    } else {
      const uint32_t leaf = (uint32_t)~cur;
      const float4* tp = reinterpret_cast<const float4*>(tris + leaf);
      const float4 a = tp[0], b = tp[1], c = tp[2];
      if (COUNT) ++n_tris;
      BvhTri tr;
      tr.v0[0] = a.x; tr.v0[1] = a.y; tr.v0[2] = a.z; tr.world_id = __float_as_uint(a.w);
      tr.e1[0] = b.x; tr.e1[1] = b.y; tr.e1[2] = b.z; tr.instance = __float_as_uint(b.w);
      tr.e2[0] = c.x; tr.e2[1] = c.y; tr.e2[2] = c.z; tr.prim_flags = __float_as_uint(c.w);
      float t, u, v;
      if (ray_triangle(tr, o, d, tmin, t, u, v) && t < tmax) {
        const bool better = best.leaf == 0xFFFFFFFFu ? true : (t < best.t || (t == best.t && tr.world_id < best_id));
        if (better && (!(tr.prim_flags >> 31) || alpha_test(S, tr, u, v))) {
          best = HitRecord{t, u, v, leaf};
          best_id = tr.world_id;
          if (ANY) return best;
        }
      }
    }
    if (st.sp == 0) break;
    cur = st.pop();
  }
  return best;
}

__device__ __forceinline__ void flush_counters(TraceCounters* c, bool shadow, unsigned long long rays, unsigned long long nodes,
                                               unsigned long long tris, unsigned long long hits, unsigned long long fresh = 0) {
  // wave-level reduction first, one atomic per wave and counter (Guideline 12)
  for (int off = 32; off > 0; off >>= 1) {
    rays += __shfl_down(rays, off);
    nodes += __shfl_down(nodes, off);
    tris += __shfl_down(tris, off);
    hits += __shfl_down(hits, off);
    fresh += __shfl_down(fresh, off);
  }
  if ((threadIdx.x & 63) == 0) {
    if (shadow) {
      atomicAdd(&c->shadow_rays, rays); atomicAdd(&c->shadow_nodes, nodes); atomicAdd(&c->shadow_tris, tris);
    } else {
      atomicAdd(&c->closest_rays, rays); atomicAdd(&c->closest_nodes, nodes); atomicAdd(&c->closest_tris, tris);
      atomicAdd(&c->hits, hits);
      atomicAdd(&c->fresh, fresh);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_trace_closest: path_trace.rgen:143-169
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(kBlock) k_trace_closest(const LaunchArgs A) {
  __shared__ int s_stack[kLdsStack * kBlock];
  const uint32_t lid = blockIdx.x * kBlock + threadIdx.x;
  const PixelId px = pixel_of(A.map, lid);
  unsigned long long n_nodes = 0, n_tris = 0, n_rays = 0, n_hits = 0, n_fresh = 0;
  if (lid < kQueueShards) A.st.queue_count[lid * kCounterStride] = 0;   // the previous launch's k_shadow_queue has drained the queue; k_shade refills it
  if (px.active) {
    float4 ro = A.st.ray_o[lid], rd = A.st.ray_d[lid];
    vec3 origin, direction;
    if (A.frame.direct_only || ro.w == 0.0f) {
      n_fresh = 1;
      // new path: camera ray through the jittered pixel (ray_origin / ray_dir, path_trace.rgen:47-73)
      const float pxf = (float)px.x + A.frame.pixel_offset[0], pyf = (float)px.y + A.frame.pixel_offset[1];
      const float ndcx = -1.0f + 2.0f * (pxf / A.frame.scene_size[0]), ndcy = -1.0f + 2.0f * (pyf / A.frame.scene_size[1]);
      const float* c2w = A.cam.camera2world;
      const float* s2c = A.cam.screen2camera;
      const float ortho = gl_step(0.5f, A.frame.camera_persp ? 0.0f : 1.0f), persp = gl_step(0.5f, A.frame.camera_persp ? 1.0f : 0.0f);
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
      A.st.ray_o[lid] = make_float4(origin.x, origin.y, origin.z, ro.w);
      A.st.ray_d[lid] = make_float4(direction.x, direction.y, direction.z, rd.w);
    } else {
      origin = mk3(ro.x, ro.y, ro.z);
      direction = mk3(rd.x, rd.y, rd.z);
    }
    Stack st{&s_stack[threadIdx.x], A.st.overflow + (size_t)lid * A.st.overflow_depth, 0};
    const HitRecord h = traverse<false, COUNT>(A.scene, origin, direction, 0.0001f, INFINITY, st, n_nodes, n_tris);
    A.st.hit[lid] = make_float4(h.leaf == 0xFFFFFFFFu ? INFINITY : h.t, h.u, h.v, __uint_as_float(h.leaf));
    n_rays = 1;
    n_hits = h.leaf != 0xFFFFFFFFu;
  }
  if (COUNT) flush_counters(A.counters, false, n_rays, n_nodes, n_tris, n_hits, n_fresh);
}

// update_count() + update_result() of path_trace.rgen:119-133 for one pixel
__device__ __forceinline__ void accumulate_pixel(const LaunchArgs& A, uint32_t lid, vec3 c, bool add, bool update) {
  float4 cum = A.st.cumulative[lid];
  cum.w += 1.0f;
  if (update) {
    if (add) { cum.x += c.x; cum.y += c.y; cum.z += c.z; }
    A.st.result[lid] = make_float4(cum.x * A.frame.exposure / cum.w, cum.y * A.frame.exposure / cum.w, cum.z * A.frame.exposure / cum.w, 1.0f);
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
    const uint32_t shard = blockIdx.x % kQueueShards;
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
// k_shade: path_trace.rgen:170-237 minus the two traceRayEXT calls, raytrace_hit.rchit:30-71
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_shade(const LaunchArgs A) {
  const uint32_t lid = blockIdx.x * kBlock + threadIdx.x;
  const PixelId px = pixel_of(A.map, lid);
  if (!px.active) return;
  const DeviceScene& S = A.scene;
  const FrameData& F = A.frame;
  const float4 ro = A.st.ray_o[lid], rd = A.st.ray_d[lid], hr = A.st.hit[lid];
  const bool fresh = F.direct_only || ro.w == 0.0f;
  float bounce = F.direct_only ? 0.0f : ro.w;
  const vec3 direction = mk3(rd.x, rd.y, rd.z);
  Spec importance;
  if (fresh) {
    importance = spec_set(1.0f);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = A.st.imp[q][lid];
      importance.w[4 * q] = v.x; importance.w[4 * q + 1] = v.y; importance.w[4 * q + 2] = v.z; importance.w[4 * q + 3] = v.w;
    }
  }
  const uint32_t leaf = __float_as_uint(hr.w);
  if (leaf == 0xFFFFFFFFu) {
    // miss: optional sky radiance, path reset (path_trace.rgen:170-179)
    uint32_t flags = 0;
    vec3 c = mk3(0.0f, 0.0f, 0.0f);
    if ((bounce == 0.0f || rd.w == 1.0f) && S.sky.tex_id > 0) {
      const vec3 w = normalize3(xform_dir(S.sky.world2obj, direction));   // sky_radiance, :75-82
      const float phi = glz_atan2f(w.y, w.x), theta = glz_acosf(w.z);
      const vec3 texel = texture_rgb(S, S.sky.tex_id, vec2{phi * kInv2Pi, theta * kInvPi});
      c = spec_to_rgb(spec_mul(importance, from_illuminant_color(texel)));
      flags = kFlagUpdate;
    }
    accumulate_pixel(A, lid, c, true, flags != 0);
    if (!F.direct_only) A.st.ray_o[lid] = make_float4(ro.x, ro.y, ro.z, 0.0f);   // RESET_PATH
    return;
  }
  // ---- closest-hit shader (raytrace_hit.rchit:30-71) ----
  const float4* tp = reinterpret_cast<const float4*>(S.bvh_tris + leaf);
  const uint32_t inst_id = __float_as_uint(tp[1].w), prim = __float_as_uint(tp[2].w) & 0x7FFFFFFFu;
  const RTInstance inst = S.instances[inst_id];
  const uint32_t tri_id = inst.index_offset / 3u + prim;
  const float b0 = 1.0f - hr.y - hr.z, b1 = hr.y, b2 = hr.z;
  const uint32_t i0 = S.indices[3u * tri_id], i1 = S.indices[3u * tri_id + 1u], i2 = S.indices[3u * tri_id + 2u];
  const float4 va0 = S.vertices[2u * i0], va1 = S.vertices[2u * i0 + 1u];
  const float4 vb0 = S.vertices[2u * i1], vb1 = S.vertices[2u * i1 + 1u];
  const float4 vc0 = S.vertices[2u * i2], vc1 = S.vertices[2u * i2 + 1u];
  vec3 point = (mk3(va0.x, va0.y, va0.z) * b0 + mk3(vb0.x, vb0.y, vb0.z) * b1) + mk3(vc0.x, vc0.y, vc0.z) * b2;
  const vec2 uv = vec2{(va1.z * b0 + vb1.z * b1) + vc1.z * b2, (va1.w * b0 + vb1.w * b1) + vc1.w * b2};
  const float4 dn = S.derivatives[3u * tri_id], du = S.derivatives[3u * tri_id + 1u], dv = S.derivatives[3u * tri_id + 2u];
  vec3 ng = mk3(dn.x, dn.y, dn.z), dpdu = mk3(du.x, du.y, du.z);
  (void)dv;   // dpdv is transformed by the reference but never read afterwards
  vec3 ns = (mk3(va0.w, va1.x, va1.y) * b0 + mk3(vb0.w, vb1.x, vb1.y) * b1) + mk3(vc0.w, vc1.x, vc1.y) * b2;
  const RTMaterial* mat = &S.materials[inst.material_id];
  if (mat->normal != 0) {
    const vec4 tx = texture2d(S, mat->normal, uv.x, uv.y);
    Frame old;
    old.s = normalize3(dpdu);
    old.n = ns;
    old.t = normalize3(cross3(old.n, old.s));
    ns = normalize3(to_world(mk3(tx.x * 2.0f - 1.0f, tx.y * 2.0f - 1.0f, tx.z * 2.0f - 1.0f), old));
    ns = ns * gl_sign(dot3(ng, ns));
  }
  const TransformPair* xf = &S.transforms[inst.transform_id];
  point = xform_point(xf->o2w, point);
  dpdu = xform_point(xf->o2w, dpdu);   // transformed as a point, w = 1 (Q8)
  ng = xform_tdir(xf->w2o, ng);
  ns = xform_tdir(xf->w2o, ns);
  (void)ng;
  // ---- raygen continues (path_trace.rgen:180-237) ----
  uint32_t rng = pcg(__float_as_uint((float)F.seed) ^ pcg(__float_as_uint((float)px.x) ^ pcg(__float_as_uint((float)px.y))));   // :143, Q11
  SurfacePoint P;
  P.woW = -direction;
  P.uv = uv;
  P.frame = make_frame(dpdu, ns);
  P.mat = mat;
  float spec_flag;
  if (mat->is_specular == 0) {
    // direct_light(), :84-117
    const uint32_t li = (uint32_t)gl_min(rand01(rng) * (float)F.lights_no, (float)(F.lights_no - 1u));
    vec3 xi;
    xi.x = rand01(rng); xi.y = rand01(rng); xi.z = rand01(rng);
    LightSample ls;
    ls.pdf = 0.0f;
    sample_light(S, li, point, xi, F.scene_radius, ls);
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
        Spec rad;
        float poison = 0.0f;
        GLZ_BINS {
          const float rl = value.w[i] * ls.emission.w[i];
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
      float probe = 0.0f;
      GLZ_BINS probe += 0.0f * importance.w[i];
      if (probe != probe) c = spec_to_rgb(spec_scale(importance, 0.0f * (float)F.lights_no));
    }
    // shadow-ray queue (consumed by k_shadow_queue); pixels without a shadow ray are accumulated right here
    const bool push = (flags & kFlagShadow) != 0;
    const uint32_t slot = queue_slot(A.st.queue_count, A.map.n_local_pixels, push);
    if (push) {
      A.st.sh_o[slot] = make_float4(point.x, point.y, point.z, sh_tmax);
      A.st.sh_d[slot] = make_float4(sh_dir.x, sh_dir.y, sh_dir.z, __uint_as_float(lid));
      A.st.contrib[slot] = make_float4(c.x, c.y, c.z, __uint_as_float(flags));
    } else {
      accumulate_pixel(A, lid, c, true, true);
    }
    spec_flag = 0.0f;
  } else {
    accumulate_pixel(A, lid, mk3(0.0f, 0.0f, 0.0f), false, false);
    spec_flag = 1.0f;
  }
  if (F.direct_only) return;
  // Russian roulette (:197-210)
  if (bounce > (float)(F.pt_steps / 2u)) {
    const float kill = gl_max(0.05f, 1.0f - spec_luminance(importance));
    if (rand01(rng) < kill) {
      A.st.ray_o[lid] = make_float4(ro.x, ro.y, ro.z, 0.0f);
      A.st.ray_d[lid] = make_float4(rd.x, rd.y, rd.z, spec_flag);
      return;
    }
    importance = spec_scale(importance, 1.0f / (1.0f - kill));
  }
  vec3 xi;
  xi.x = rand01(rng); xi.y = rand01(rng); xi.z = rand01(rng);
  Spec value = spec_set(0.0f);
  vec3 wiW = mk3(0.0f, 0.0f, 0.0f);
  const float pdf = bsdf_sample(S, P, xi, value, wiW);   // :212-218
  if (pdf == 0.0f) {
    A.st.ray_o[lid] = make_float4(ro.x, ro.y, ro.z, 0.0f);
    A.st.ray_d[lid] = make_float4(rd.x, rd.y, rd.z, spec_flag);
    return;
  }
  float weight = fabsf(dot3(wiW, ns));
  weight /= pdf;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    A.st.imp[q][lid] = make_float4(importance.w[4 * q] * (value.w[4 * q] * weight), importance.w[4 * q + 1] * (value.w[4 * q + 1] * weight),
                                   importance.w[4 * q + 2] * (value.w[4 * q + 2] * weight), importance.w[4 * q + 3] * (value.w[4 * q + 3] * weight));
  bounce = bounce < (float)F.pt_steps ? bounce + 1.0f : 0.0f;   // :230-237
  A.st.ray_o[lid] = make_float4(point.x, point.y, point.z, bounce);
  A.st.ray_d[lid] = make_float4(wiW.x, wiW.y, wiW.z, spec_flag);
}

// ---------------------------------------------------------------------------------------------
// k_shadow_queue: the shadow traceRayEXT (path_trace.rgen:106-110) for the compacted queue written by
// k_shade, followed by update_count / update_result (:119-133) of the owning pixel.  Persistent grid: each
// thread strides over the queue, so every wave traverses with (almost) all lanes active.
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(kBlock) k_shadow_queue(const LaunchArgs A) {
  __shared__ int s_stack[kLdsStack * kBlock];
  // prefix sums of the eight shard counts (final: k_shade has completed)
  uint32_t start[kQueueShards + 1];
  start[0] = 0;
#pragma unroll
  for (uint32_t k = 0; k < kQueueShards; ++k) start[k + 1] = start[k] + A.st.queue_count[k * kCounterStride];
  const uint32_t count = start[kQueueShards], cap = queue_capacity(A.map.n_local_pixels);
  unsigned long long n_nodes = 0, n_tris = 0, n_rays = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    uint32_t shard = 0;
#pragma unroll
    for (uint32_t k = 1; k < kQueueShards; ++k) shard += i >= start[k] ? 1u : 0u;
    const uint32_t q = shard * cap + (i - start[shard]);
    const float4 so = A.st.sh_o[q], sd = A.st.sh_d[q], cb = A.st.contrib[q];
    const uint32_t lid = __float_as_uint(sd.w), flags = __float_as_uint(cb.w);
    Stack st{&s_stack[threadIdx.x], A.st.overflow + (size_t)lid * A.st.overflow_depth, 0};
    const HitRecord h = traverse<true, COUNT>(A.scene, mk3(so.x, so.y, so.z), mk3(sd.x, sd.y, sd.z), 0.001f, so.w, st, n_nodes, n_tris);
    const bool occluded = h.leaf != 0xFFFFFFFFu;
    n_rays += 1;
    vec3 c = mk3(cb.x, cb.y, cb.z);
    bool add = !occluded;
    if (occluded && (flags & kFlagPoison)) {
      const float nan = __uint_as_float(0x7FC00000u);
      c = mk3(nan, nan, nan);
      add = true;
    }
    accumulate_pixel(A, lid, c, add, true);
  }
  if (COUNT) flush_counters(A.counters, true, n_rays, n_nodes, n_tris, 0);
}

// ---------------------------------------------------------------------------------------------
// image plumbing
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_export(const TileMap map, const float4* __restrict__ tiled, float4* __restrict__ frame) {
  const uint32_t lid = blockIdx.x * kBlock + threadIdx.x;
  const PixelId px = pixel_of(map, lid);
  if (px.active) frame[(size_t)px.y * map.width + px.x] = tiled[lid];
}

// linear -> sRGB OETF, 8 bit, round to nearest (what the R8G8B8A8_SRGB blit does, raytracer.rs:576-584 [ext])
__device__ __forceinline__ unsigned char srgb8(float c) {
  if (!(c > 0.0f)) return 0;
  if (c >= 1.0f) return 255;
  const float v = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
  const int q = (int)(v * 255.0f + 0.5f);
  return (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
}
__global__ void __launch_bounds__(kBlock) k_tonemap(uint32_t n, const float4* __restrict__ result, uchar4* __restrict__ out) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 r = result[i];
  out[i] = make_uchar4(srgb8(r.x), srgb8(r.y), srgb8(r.z), r.w >= 1.0f ? 255 : 0);
}
// ---------------------------------------------------------------------------------------------
// debug / parity kernels: arbitrary rays through the same traversal code
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_debug_closest(const DeviceScene S, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                                          float tmin, float* t, uint32_t* tri, uint32_t* inst, float* u, float* v,
                                                          uint32_t* overflow, uint32_t overflow_depth) {
  __shared__ int s_stack[kLdsStack * kBlock];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  unsigned long long a = 0, b = 0;
  Stack st{&s_stack[threadIdx.x], overflow + (size_t)i * overflow_depth, 0};
  const HitRecord h = traverse<false, false>(S, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, INFINITY, st, a, b);
  const bool hit = h.leaf != 0xFFFFFFFFu;
  t[i] = hit ? h.t : INFINITY;
  tri[i] = hit ? S.bvh_tris[h.leaf].world_id : 0xFFFFFFFFu;
  inst[i] = hit ? S.bvh_tris[h.leaf].instance : 0xFFFFFFFFu;
  u[i] = hit ? h.u : 0.0f;
  v[i] = hit ? h.v : 0.0f;
}
__global__ void __launch_bounds__(kBlock) k_debug_any(const DeviceScene S, const float* __restrict__ o, const float* __restrict__ d,
                                                      const float* __restrict__ tmax, uint32_t n, float tmin, uint8_t* out, uint32_t* overflow,
                                                      uint32_t overflow_depth) {
  __shared__ int s_stack[kLdsStack * kBlock];
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  unsigned long long a = 0, b = 0;
  Stack st{&s_stack[threadIdx.x], overflow + (size_t)i * overflow_depth, 0};
  const HitRecord h = traverse<true, false>(S, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmin, tmax[i], st, a, b);
  out[i] = h.leaf != 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }

hipError_t launch_trace_closest(hipStream_t st, const LaunchArgs& a) {
  if (a.map.n_local_pixels == 0) return hipSuccess;
  if (a.counters) hipLaunchKernelGGL(k_trace_closest<true>, grid_for(a.map.n_local_pixels), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL(k_trace_closest<false>, grid_for(a.map.n_local_pixels), dim3(kBlock), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_shade(hipStream_t st, const LaunchArgs& a) {
  if (a.map.n_local_pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(k_shade, grid_for(a.map.n_local_pixels), dim3(kBlock), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_shadow_accumulate(hipStream_t st, const LaunchArgs& a) {
  if (a.map.n_local_pixels == 0) return hipSuccess;
  // persistent grid: 256 CUs x 8 resident blocks at most, never more blocks than queue capacity
  const uint32_t blocks = std::min<uint32_t>((a.map.n_local_pixels + kBlock - 1) / kBlock, 256u * 8u);
  if (a.counters) hipLaunchKernelGGL(k_shadow_queue<true>, dim3(blocks), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL(k_shadow_queue<false>, dim3(blocks), dim3(kBlock), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_export(hipStream_t st, const TileMap& map, const float4* tiled, float4* frame, bool zero_first) {
  if (zero_first) {
    hipError_t e = hipMemsetAsync(frame, 0, sizeof(float4) * (size_t)map.width * map.height, st);
    if (e != hipSuccess) return e;
  }
  if (map.n_local_pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(k_export, grid_for(map.n_local_pixels), dim3(kBlock), 0, st, map, tiled, frame);
  return hipGetLastError();
}
hipError_t launch_tonemap(hipStream_t st, uint32_t n, const float4* result_frame, uchar4* out) {
  hipLaunchKernelGGL(k_tonemap, grid_for(n), dim3(kBlock), 0, st, n, result_frame, out);
  return hipGetLastError();
}
hipError_t launch_debug_closest(hipStream_t st, const DeviceScene& scene, const float* o, const float* d, uint32_t n, float tmin, float* t,
                                uint32_t* tri, uint32_t* inst, float* u, float* v, uint32_t* overflow, uint32_t overflow_depth) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_debug_closest, grid_for(n), dim3(kBlock), 0, st, scene, o, d, n, tmin, t, tri, inst, u, v, overflow, overflow_depth);
  return hipGetLastError();
}
hipError_t launch_debug_any(hipStream_t st, const DeviceScene& scene, const float* o, const float* d, const float* tmax, uint32_t n, float tmin,
                            uint8_t* hit, uint32_t* overflow, uint32_t overflow_depth) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_debug_any, grid_for(n), dim3(kBlock), 0, st, scene, o, d, tmax, n, tmin, hit, overflow, overflow_depth);
  return hipGetLastError();
}

}  // namespace glz
