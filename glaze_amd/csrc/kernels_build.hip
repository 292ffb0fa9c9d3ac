// Scene-build kernels for gfx950: per-triangle derivatives, instance flattening, and the LBVH
// (Morton codes -> bitonic sort -> Karras hierarchy -> bottom-up fit) that replaces the driver's
// BLAS/TLAS build (lib/src/vulkan/acceleration.rs:89-494).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "device/math.h"
#include "device/types.h"
#include "kernels.h"

namespace glz {
using namespace dev;

// ---------------------------------------------------------------------------------------------
// generate_derivatives.comp:23-64 -- one thread per object-space triangle, 48 bytes out
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tri_derivatives(const float4* __restrict__ vertices, const uint32_t* __restrict__ indices,
                                                         uint32_t n_tris, float4* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tris) return;
  const uint32_t i0 = indices[3 * t], i1 = indices[3 * t + 1], i2 = indices[3 * t + 2];
  const float4 a0 = vertices[2 * i0], a1 = vertices[2 * i0 + 1];
  const float4 b0 = vertices[2 * i1], b1 = vertices[2 * i1 + 1];
  const float4 c0 = vertices[2 * i2], c1 = vertices[2 * i2 + 1];
  const vec3 p0 = mk3(a0.x, a0.y, a0.z), p1 = mk3(b0.x, b0.y, b0.z), p2 = mk3(c0.x, c0.y, c0.z);
  // texcoords are the last two floats of the packed vertex (raytrace_commons.glsl:28-31)
  const float duv02x = a1.z - c1.z, duv02y = a1.w - c1.w;
  const float duv12x = b1.z - c1.z, duv12y = b1.w - c1.w;
  const float det = duv02x * duv12y - duv02y * duv12x;
  const vec3 n = normalize3(cross3(p1 - p0, p2 - p0));
  vec3 dpdu, dpdv;
  if (det == 0.0f) {
    if (fabsf(n.x) > fabsf(n.y)) dpdu = mk3(-n.z, 0.0f, n.x) / sqrtf(n.x * n.x + n.z * n.z);
    else dpdu = mk3(0.0f, n.z, -n.y) / sqrtf(n.y * n.y + n.z * n.z);
    dpdv = cross3(n, dpdu);
  } else {
    const vec3 dp02 = p0 - p2, dp12 = p1 - p2;
    const float invdet = 1.0f / det;
    dpdu = (duv12y * dp02 - duv02y * dp12) * invdet;
    dpdv = ((-duv12x) * dp02 + duv02x * dp12) * invdet;
  }
  out[3 * t] = make_float4(n.x, n.y, n.z, 0.0f);
  out[3 * t + 1] = make_float4(dpdu.x, dpdu.y, dpdu.z, 0.0f);
  out[3 * t + 2] = make_float4(dpdv.x, dpdv.y, dpdv.z, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// Instance flattening: world triangle w -> (instance, primitive), world-space v0/e1/e2 + AABB.
// Scene bounds are reduced per block in LDS, then one ordered-int atomic per block and axis.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int float_to_ordered(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ void __launch_bounds__(256) k_world_tris(const float4* __restrict__ vertices, const uint32_t* __restrict__ indices,
                                                    const RTInstance* __restrict__ instances, const uint32_t* __restrict__ inst_base,
                                                    uint32_t n_instances, const TransformPair* __restrict__ transforms,
                                                    const RTMaterial* __restrict__ materials, uint32_t n_world,
                                                    BvhTri* __restrict__ tris, float4* __restrict__ box_lo, float4* __restrict__ box_hi,
                                                    int* __restrict__ scene_bounds /* 6 ordered ints: centroid lo xyz, hi xyz */) {
  __shared__ float s_lo[3][256], s_hi[3][256];
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (w < n_world) {
    // binary search: last instance whose first world triangle is <= w
    uint32_t lo = 0, hi = n_instances - 1;
    while (lo < hi) {
      uint32_t mid = (lo + hi + 1) >> 1;
      if (inst_base[mid] <= w) lo = mid; else hi = mid - 1;
    }
    const uint32_t inst = lo, prim = w - inst_base[inst];
    const RTInstance in = instances[inst];
    const uint32_t* ix = indices + in.index_offset + 3 * prim;
    const float* M = transforms[in.transform_id].o2w;
    vec3 v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 p = vertices[2 * ix[k]];
      v[k] = xform_point(M, mk3(p.x, p.y, p.z));
    }
    BvhTri t;
    t.v0[0] = v[0].x; t.v0[1] = v[0].y; t.v0[2] = v[0].z; t.world_id = w;
    t.v1[0] = v[1].x; t.v1[1] = v[1].y; t.v1[2] = v[1].z; t.instance = inst;
    t.v2[0] = v[2].x; t.v2[1] = v[2].y; t.v2[2] = v[2].z;
    t.prim_flags = prim | (materials[in.material_id].opacity != 0 ? kTriNonOpaque : 0u);   // acceleration.rs:136-141
    tris[w] = t;
    const float* A = &v[0].x; const float* B = &v[1].x; const float* C = &v[2].x;
    float l[3], h[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      l[k] = fminf(A[k], fminf(B[k], C[k]));
      h[k] = fmaxf(A[k], fmaxf(B[k], C[k]));
      // conservative pad: the ray/triangle test may accept points a few ulps outside the exact box
      const float pad = 1e-5f * fmaxf(fmaxf(fabsf(l[k]), fabsf(h[k])), 1e-3f);
      l[k] -= pad; h[k] += pad;
      clo[k] = chi[k] = 0.5f * (l[k] + h[k]);
    }
    box_lo[w] = make_float4(l[0], l[1], l[2], 0.0f);
    box_hi[w] = make_float4(h[0], h[1], h[2], 0.0f);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) { s_lo[k][threadIdx.x] = clo[k]; s_hi[k][threadIdx.x] = chi[k]; }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        s_lo[k][threadIdx.x] = fminf(s_lo[k][threadIdx.x], s_lo[k][threadIdx.x + s]);
        s_hi[k][threadIdx.x] = fmaxf(s_hi[k][threadIdx.x], s_hi[k][threadIdx.x + s]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    atomicMin(&scene_bounds[threadIdx.x], float_to_ordered(s_lo[threadIdx.x][0]));
    atomicMax(&scene_bounds[3 + threadIdx.x], float_to_ordered(s_hi[threadIdx.x][0]));
  }
}

// 21 bits per axis interleaved to a 63-bit Morton code
__device__ __forceinline__ uint64_t spread21(uint64_t x) {
  x &= 0x1FFFFFull;
  x = (x | x << 32) & 0x1F00000000FFFFull;
  x = (x | x << 16) & 0x1F0000FF0000FFull;
  x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void __launch_bounds__(256) k_morton(const float4* __restrict__ box_lo, const float4* __restrict__ box_hi,
                                                const int* __restrict__ scene_bounds, uint32_t n, uint32_t n_padded,
                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_padded) return;
  if (i >= n) {   // padding of the power-of-two bitonic network sorts to the end
    keys[i] = ~0ull;
    vals[i] = 0xFFFFFFFFu;
    return;
  }
  float q[3];
  const float4 l = box_lo[i], h = box_hi[i];
  const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lo = ordered_to_float(scene_bounds[k]), hi = ordered_to_float(scene_bounds[3 + k]);
    const float ext = hi - lo;
    float t = ext > 0.0f ? (c[k] - lo) / ext : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    q[k] = fminf(t * 2097152.0f, 2097151.0f);
  }
  keys[i] = (spread21((uint64_t)q[0]) << 2) | (spread21((uint64_t)q[1]) << 1) | spread21((uint64_t)q[2]);
  vals[i] = i;
}

// ---------------------------------------------------------------------------------------------
// Bitonic sort of (key, value) pairs, n a power of two.  Strides >= 1024 run one compare-exchange
// per launch in global memory; all strides below are fused in LDS (2048 pairs per 1024-thread block).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool key_greater(uint64_t ka, uint32_t va, uint64_t kb, uint32_t vb) {
  return ka > kb || (ka == kb && va > vb);
}

__global__ void __launch_bounds__(256) k_bitonic_global(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n, uint32_t k,
                                                        uint32_t j) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per pair
  if (t >= n / 2) return;
  const uint32_t i = 2 * t - (t & (j - 1));   // index with bit j cleared
  const uint32_t p = i + j;
  const bool up = (i & k) == 0;
  const uint64_t ka = keys[i], kb = keys[p];
  const uint32_t va = vals[i], vb = vals[p];
  if (key_greater(ka, va, kb, vb) == up) {
    keys[i] = kb; keys[p] = ka;
    vals[i] = vb; vals[p] = va;
  }
}

constexpr uint32_t kSortTile = 2048;
__global__ void __launch_bounds__(1024) k_bitonic_lds(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n, uint32_t k_first,
                                                      uint32_t k_last, uint32_t j_first) {
  // Runs, for k = k_first..k_last (doubling), the strides j = min(j_first or k/2, 1024) .. 1 inside one tile.
  __shared__ uint64_t s_k[kSortTile];
  __shared__ uint32_t s_v[kSortTile];
  const uint32_t base = blockIdx.x * kSortTile;
  for (uint32_t t = threadIdx.x; t < kSortTile; t += blockDim.x) {
    s_k[t] = keys[base + t];
    s_v[t] = vals[base + t];
  }
  __syncthreads();
  for (uint32_t k = k_first; k <= k_last; k <<= 1) {
    uint32_t j = (k == k_first && j_first) ? j_first : k >> 1;
    if (j > kSortTile / 2) j = kSortTile / 2;
    for (; j > 0; j >>= 1) {
      const uint32_t t = threadIdx.x;
      const uint32_t i = 2 * t - (t & (j - 1));
      const uint32_t p = i + j;
      const bool up = ((base + i) & k) == 0;
      const uint64_t ka = s_k[i], kb = s_k[p];
      const uint32_t va = s_v[i], vb = s_v[p];
      if (key_greater(ka, va, kb, vb) == up) {
        s_k[i] = kb; s_k[p] = ka;
        s_v[i] = vb; s_v[p] = va;
      }
      __syncthreads();
    }
  }
  for (uint32_t t = threadIdx.x; t < kSortTile; t += blockDim.x) {
    keys[base + t] = s_k[t];
    vals[base + t] = s_v[t];
  }
}

// ---------------------------------------------------------------------------------------------
// Leaves.  Two triangles of one instance that follow each other in the index buffer, share an edge and have largely the
// same box (the two halves of a quad: what tessellated grids and triangulated quad meshes consist of) form ONE leaf: the
// hierarchy is built over half as many primitives and the tracer tests both triangles in one leaf round.  A leaf's
// triangles are adjacent in bvh_tris (the first one carries kTriHasPartner); leaf links point at the first.
// Pairs start at even primitives first, then at odd ones between triangles that are still single.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float box_area3(float4 l, float4 h) {
  const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
__global__ void __launch_bounds__(256) k_pair_triangles(uint32_t n_world, uint32_t parity, const BvhTri* __restrict__ tris,
                                                        const uint32_t* __restrict__ indices, const RTInstance* __restrict__ instances,
                                                        const float4* __restrict__ box_lo, const float4* __restrict__ box_hi,
                                                        float area_ratio, uint32_t quads_only,
                                                        uint8_t* role /* 0 single, 1 / 5 first of a pair (5: kQuadSwapped order), 2 second of a pair */) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w + 1 >= n_world) return;
  const BvhTri a = tris[w], b = tris[w + 1];
  const uint32_t prim = a.prim_flags & kTriPrimMask;
  if ((prim & 1u) != parity || a.instance != b.instance) return;
  if (parity == 1u && (role[w] != 0 || role[w + 1] != 0)) return;
  const RTInstance in = instances[a.instance];
  const uint32_t* ia = indices + in.index_offset + 3 * prim;
  const uint32_t* ib = ia + 3;
  int shared = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) shared += ia[i] == ib[j] ? 1 : 0;
  if (shared != 2) return;
  // The vertex order of the second triangle in terms of the first one's corners (3 = its own fourth vertex).  A quad record
  // (types.h BvhQuad) holds A = (q0, q1, q2), B = (q0, q2, q3): the pair fits as it is when the second triangle reads (0, 2, 3), and
  // with the two triangles in the other order when it reads (0, 3, 1) -- then the SECOND one is A = (p0, d, p1) and the first one
  // (p0, p1, p2) = (q0, q2, q3) is B.  Together these are the two ways a quad is cut into a fan from a shared first vertex (all pairs of the
  // atrium, 98 % of mattest.glaze's); other orders stay single triangles, because re-ordering a triangle's vertices would change the
  // rounding of its (t, u, v).
  uint32_t pat = 0;
  for (int j = 0; j < 3; ++j) {
    uint32_t m = 3;
    for (int i = 0; i < 3; ++i) m = ia[i] == ib[j] ? (uint32_t)i : m;
    pat |= m << (4 * j);
  }
  const bool fits = pat == 0x320u, fits_swapped = pat == 0x130u;   // (0, 2, 3) / (0, 3, 1), first entry in the low nibble
  if (quads_only && !fits && !fits_swapped) return;
  const float4 la = box_lo[w], ha = box_hi[w], lb = box_lo[w + 1], hb = box_hi[w + 1];
  const float4 lm = make_float4(fminf(la.x, lb.x), fminf(la.y, lb.y), fminf(la.z, lb.z), 0.0f);
  const float4 hm = make_float4(fmaxf(ha.x, hb.x), fmaxf(ha.y, hb.y), fmaxf(ha.z, hb.z), 0.0f);
  // one box for both must not cost more than it saves: identical boxes give 0.5, two squares side by side 0.83
  if (!(box_area3(lm, hm) <= area_ratio * (box_area3(la, ha) + box_area3(lb, hb)))) return;
  role[w] = (quads_only && fits_swapped) ? 5 : 1;
  role[w + 1] = 2;
}
__global__ void __launch_bounds__(256) k_leaf_flags(uint32_t n_world, const uint8_t* __restrict__ role, unsigned long long* __restrict__ flags) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < n_world) flags[w] = role[w] != 2 ? 1ull : 0ull;
}
// leaf l (numbered in world-triangle order) -> its first triangle and its box
__global__ void __launch_bounds__(256) k_leaf_boxes(uint32_t n_world, const uint8_t* __restrict__ role, const unsigned long long* __restrict__ pos,
                                                    const float4* __restrict__ box_lo, const float4* __restrict__ box_hi,
                                                    uint32_t* __restrict__ leaf_first, float4* __restrict__ leaf_lo, float4* __restrict__ leaf_hi) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_world || role[w] == 2) return;
  const uint32_t l = (uint32_t)pos[w];
  float4 lo = box_lo[w], hi = box_hi[w];
  if (role[w] & 1) {
    const float4 l2 = box_lo[w + 1], h2 = box_hi[w + 1];
    lo = make_float4(fminf(lo.x, l2.x), fminf(lo.y, l2.y), fminf(lo.z, l2.z), 0.0f);
    hi = make_float4(fmaxf(hi.x, h2.x), fmaxf(hi.y, h2.y), fmaxf(hi.z, h2.z), 0.0f);
  }
  leaf_first[l] = w;
  leaf_lo[l] = lo;
  leaf_hi[l] = hi;
}
// triangles of the leaf at sorted place j (for the prefix sum that gives its first slot in bvh_tris)
__global__ void __launch_bounds__(256) k_leaf_sizes(uint32_t n, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ leaf_first,
                                                    const uint8_t* __restrict__ role, unsigned long long* __restrict__ sizes) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) sizes[j] = (role[leaf_first[vals[j]]] & 1) ? 2ull : 1ull;
}

// gathers triangles and boxes into leaf (sorted) order
__global__ void __launch_bounds__(256) k_gather_leaves(const uint32_t* __restrict__ vals, uint32_t n, const uint32_t* __restrict__ leaf_first,
                                                       const uint8_t* __restrict__ role, const unsigned long long* __restrict__ slot,
                                                       const BvhTri* __restrict__ tris_in, const float4* __restrict__ lo_in,
                                                       const float4* __restrict__ hi_in, BvhTri* __restrict__ tris_out, float4* __restrict__ node_lo,
                                                       float4* __restrict__ node_hi, BvhQuad* __restrict__ quads_out /* null: none */) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint32_t leaf = vals[j], first = leaf_first[leaf];
  const uint32_t s = (uint32_t)slot[j];
  BvhTri t = tris_in[first];
  if (role[first] & 1) {
    t.prim_flags |= kTriHasPartner;
    tris_out[s + 1] = tris_in[first + 1];
  }
  tris_out[s] = t;
  if (quads_out) {
    // the leaf's record for the flattened tracer (types.h BvhQuad): A = (q0, q1, q2), B = (q0, q2, q3)
    BvhQuad q;
    q.world_id = t.world_id;
    q.instance = t.instance;
    q.prim_flags = t.prim_flags;
    q.slot = s;
    const float* c[4] = {t.v0, t.v1, t.v2, t.v2};   // a single triangle: q3 repeats q2 (never looked at)
    if (role[first] == 1) {            // second triangle = (p0, p2, d)
      c[3] = tris_in[first + 1].v2;
    } else if (role[first] == 5) {     // second triangle = (p0, d, p1): it is A, the first one B
      const BvhTri& b = tris_in[first + 1];
      c[1] = b.v1;
      c[2] = t.v1;
      c[3] = t.v2;
      q.prim_flags |= kQuadSwapped;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { q.q0[k] = c[0][k]; q.q1[k] = c[1][k]; q.q2[k] = c[2][k]; q.q3[k] = c[3][k]; }
    quads_out[j] = q;
  }
  node_lo[(n - 1) + j] = lo_in[leaf];   // leaf j's box lives at slot (n-1)+j, inner node i's at slot i
  node_hi[(n - 1) + j] = hi_in[leaf];
}

// ---------------------------------------------------------------------------------------------
// Karras 2012: one thread per internal node finds its key range and split
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int common_prefix(const uint64_t* __restrict__ keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t x = keys[i] ^ keys[j];
  if (x == 0) return 64 + __clz(i ^ j);   // duplicate codes: fall back to the index bits
  return __clzll((long long)x);
}

__global__ void __launch_bounds__(256) k_hierarchy(const uint64_t* __restrict__ keys, int n, int2* __restrict__ children,
                                                   int* __restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = common_prefix(keys, n, i, i - d);
  int lmax = 2;
  while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t > 0; t >>= 1)
    if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = common_prefix(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
    if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  // child link: >= 0 inner node, < 0 ~leaf
  const int left = (lo == gamma) ? ~gamma : gamma;
  const int right = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  children[i] = make_int2(left, right);
  parent[left >= 0 ? left : (n - 1) + ~left] = i;
  parent[right >= 0 ? right : (n - 1) + ~right] = i;
  if (i == 0) parent[0] = -1;
}

// Bottom-up passes run level by level: k_node_depth numbers every inner node with its distance from the root, then one
// launch per level (deepest first) lets each node of that level combine its two children, which the previous launch
// finished.  The kernel boundary is the only synchronisation: no arrival counters and no agent-scope fences (the
// per-XCD L2s are not coherent, cdna_hip_programming.md Guideline 16, and a fence per visited node costs an L2
// write-back: the arrival-counter version of these passes took 38 + 33 ms on 7 M triangles, this one 3 ms).
__global__ void __launch_bounds__(256) k_node_depth(int n, const int* __restrict__ parent, int* __restrict__ node_depth,
                                                    int* __restrict__ max_inner_depth, int* __restrict__ max_leaf_depth) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;   // inner nodes 0..n-2, leaf i at (n-1)+i
  if (t >= 2 * n - 1) return;
  int depth = 0;
  for (int p = parent[t]; p >= 0; p = parent[p]) ++depth;
  // one atomic per wave
  int m = depth;
  if (t >= n - 1) m = -1;
  int ml = t >= n - 1 ? depth : -1;
  for (int off = 32; off > 0; off >>= 1) {
    m = max(m, __shfl_xor(m, off));
    ml = max(ml, __shfl_xor(ml, off));
  }
  if (t < n - 1) node_depth[t] = depth;
  if ((threadIdx.x & 63) == 0) {
    if (m >= 0) atomicMax(max_inner_depth, m);
    if (ml >= 0) atomicMax(max_leaf_depth, ml);
  }
}

// One level of the bottom-up pass: boxes (FIT) and the number of inner nodes per subtree (for the depth-first layout).
template <bool FIT>
__global__ void __launch_bounds__(256) k_level_up(int n, int level, const int* __restrict__ node_depth, const int2* __restrict__ children,
                                                  float4* node_lo, float4* node_hi, int* counts) {
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n - 1 || node_depth[node] != level) return;
  const int2 c = children[node];
  if (FIT) {
    const int s0 = c.x >= 0 ? c.x : (n - 1) + ~c.x, s1 = c.y >= 0 ? c.y : (n - 1) + ~c.y;
    const float4 l0 = node_lo[s0], l1 = node_lo[s1];
    const float4 h0 = node_hi[s0], h1 = node_hi[s1];
    node_lo[node] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
    node_hi[node] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
  }
  counts[node] = 1 + (c.x >= 0 ? counts[c.x] : 0) + (c.y >= 0 ? counts[c.y] : 0);
}

// ---------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018): bottom-up agglomerative build over the
// Morton-ordered leaves.  Every round each cluster looks kPlocRadius positions to both sides for the neighbour
// whose merged box has the smallest surface area; mutually nearest pairs merge into a new inner node and the
// cluster array is compacted IN ORDER (prefix sum), so it stays spatially sorted.  Optional builder
// (glz_instance_set_bvh_builder): on the atrium its trees cost the same as the Karras LBVH's overall.
// Inner-node ids are handed out downwards from n-2, so the last merge creates the root as node 0.
// ---------------------------------------------------------------------------------------------
constexpr int kPlocRadius = 16;
constexpr int kScanTile = 1024;   // elements per block of the scan kernels (256 threads x 4)

__device__ __forceinline__ int box_slot(int ref, int n) { return ref >= 0 ? ref : (n - 1) + ~ref; }

__global__ void __launch_bounds__(256) k_ploc_nearest(int m, int n, const int* __restrict__ refs, const float4* __restrict__ node_lo,
                                                      const float4* __restrict__ node_hi, int* __restrict__ nearest) {
  __shared__ float4 s_lo[256 + 2 * kPlocRadius], s_hi[256 + 2 * kPlocRadius];
  const int base = (int)(blockIdx.x * 256) - kPlocRadius;
  for (int k = threadIdx.x; k < 256 + 2 * kPlocRadius; k += 256) {
    const int idx = base + k;
    if (idx >= 0 && idx < m) {
      const int slot = box_slot(refs[idx], n);
      s_lo[k] = node_lo[slot];
      s_hi[k] = node_hi[slot];
    }
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const float4 lo = s_lo[threadIdx.x + kPlocRadius], hi = s_hi[threadIdx.x + kPlocRadius];
  float best = INFINITY;
  int best_j = -1;
  for (int d = -kPlocRadius; d <= kPlocRadius; ++d) {
    const int j = i + d;
    if (d == 0 || j < 0 || j >= m) continue;
    const float4 l = s_lo[threadIdx.x + kPlocRadius + d], h = s_hi[threadIdx.x + kPlocRadius + d];
    const float dx = fmaxf(hi.x, h.x) - fminf(lo.x, l.x), dy = fmaxf(hi.y, h.y) - fminf(lo.y, l.y), dz = fmaxf(hi.z, h.z) - fminf(lo.z, l.z);
    const float area = dx * dy + dy * dz + dz * dx;
    if (area < best) {   // d ascends, so ties keep the smaller index: the globally closest pair is then always mutual
      best = area;
      best_j = j;
    }
  }
  nearest[i] = best_j;
}

// flags: low word = the cluster survives this round (merged pairs survive as their left member), high word = it is the
// left member of a merging pair (it allocates the new node)
__global__ void __launch_bounds__(256) k_ploc_flags(int m, const int* __restrict__ nearest, unsigned long long* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const int j = nearest[i];
  const bool mutual = j >= 0 && nearest[j] == i;
  const bool leader = mutual && i < j, absorbed = mutual && i > j;
  flags[i] = (absorbed ? 0ull : 1ull) | (leader ? (1ull << 32) : 0ull);
}

// exclusive prefix sum of packed counters, three phases: per-tile scan + tile totals, scan of the totals (recursive), add
__global__ void __launch_bounds__(256) k_scan_tiles(int m, const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                    unsigned long long* __restrict__ tile_sums) {
  __shared__ unsigned long long s_wave[4];
  const int base = blockIdx.x * kScanTile + threadIdx.x * 4;
  unsigned long long v[4], run = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = base + k < m ? in[base + k] : 0ull;
    run += v[k];
  }
  unsigned long long incl = run;   // inclusive scan of the per-thread sums: wave shuffle, then the 4 wave totals
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long before = 0;
  for (int w = 0; w < wave; ++w) before += s_wave[w];
  unsigned long long excl = before + incl - run;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < m) out[base + k] = excl;
    excl += v[k];
  }
  if (threadIdx.x == 255) tile_sums[blockIdx.x] = before + incl;
}
__global__ void __launch_bounds__(256) k_scan_add(int m, unsigned long long* __restrict__ out, const unsigned long long* __restrict__ tile_offsets) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) out[i] += tile_offsets[i / kScanTile];
}
// total = exclusive[m-1] + in[m-1]
__global__ void k_scan_total(int m, const unsigned long long* __restrict__ in, const unsigned long long* __restrict__ out,
                             unsigned long long* __restrict__ total) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = out[m - 1] + in[m - 1];
}
// in/out: m elements; tmp: scratch for the tile sums of every level (>= m / 1023 + 8 elements)
static hipError_t scan_exclusive(hipStream_t st, int m, const unsigned long long* in, unsigned long long* out, unsigned long long* tmp) {
  const int tiles = (m + kScanTile - 1) / kScanTile;
  hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(256), 0, st, m, in, out, tmp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || tiles == 1) return e;
  unsigned long long* sums_scanned = tmp + tiles;
  e = scan_exclusive(st, tiles, tmp, sums_scanned, sums_scanned + tiles);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_scan_add, dim3((m + 255) / 256), dim3(256), 0, st, m, out, sums_scanned);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_ploc_merge(int m, int n, int next_free, const int* __restrict__ refs, const int* __restrict__ nearest,
                                                    const unsigned long long* __restrict__ flags, const unsigned long long* __restrict__ pos,
                                                    int* __restrict__ refs_out, int2* __restrict__ children, int* __restrict__ parent,
                                                    float4* node_lo, float4* node_hi) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const unsigned long long f = flags[i], p = pos[i];
  if (!(f & 1ull)) return;   // absorbed by its left partner
  int ref = refs[i];
  if (f >> 32) {
    const int id = next_free - (int)(p >> 32);
    const int a = ref, b = refs[nearest[i]];
    const int sa = box_slot(a, n), sb = box_slot(b, n);
    const float4 l0 = node_lo[sa], l1 = node_lo[sb], h0 = node_hi[sa], h1 = node_hi[sb];
    children[id] = make_int2(a, b);
    node_lo[id] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
    node_hi[id] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 0.0f);
    parent[sa] = id;
    parent[sb] = id;
    ref = id;
  }
  refs_out[(uint32_t)p] = ref;
}
__global__ void __launch_bounds__(256) k_ploc_init(int n, int* __restrict__ refs, int* __restrict__ parent) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) refs[i] = ~i;
  if (i == 0) parent[0] = -1;
}

// ---------------------------------------------------------------------------------------------
// Top-down binned SAH on the GPU (GLZ_BVH_SAH): the same algorithm, arithmetic and tie-breaks as the host reference in
// bvh_sah.cpp, so both give the same tree node for node (tests/test_gpu_scene_trace.py).  One launch per level, one block
// per node of the level: centroid bounds of the node's range (LDS reduction) -> 3 x 16 bins (LDS atomics on ordered-int
// box coordinates) -> thread 0 walks the 45 candidate splits exactly as the host does -> stable partition of the range
// into the other index array (block-wide prefix sums over tiles) -> children.  A subtree over c leaves owns c - 1
// consecutive node ids (left child id + 1, right child id + c_left), so ids do not depend on which block runs when.
// The top levels are few blocks over long ranges (level 0 of 131 k leaves: 1.5 ms), the rest is wide and short.
// ---------------------------------------------------------------------------------------------
constexpr int kSahBins = 16;
struct SahTask {
  uint32_t b, e;
  int node;
};
__device__ __forceinline__ int sah_bin_of(float c, float lo, float scale) {
  const float f = (c - lo) * scale;
  return f >= 0.0f ? (f < (float)kSahBins ? (int)f : kSahBins - 1) : 0;
}
__device__ __forceinline__ float sah_area(const float* lo, const float* hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return dx < 0.0f ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
}
// The split of a node from its bins: bvh_sah.cpp, Builder::split, statement for statement (candidate order, strict '<').
// box: [3][kSahBins][6] ordered-int lo xyz / hi xyz, count: [3][kSahBins]; axis < 0 when binning separates nothing.
__device__ inline void sah_pick_split(const int* box, const uint32_t* count, const float* scale, int& axis_out, int& bin_out, uint32_t& n_left_out) {
  float best_cost = INFINITY;
  int best_axis = -1, best_bin = -1;
  for (int a = 0; a < 3; ++a) {
    if (!(scale[a] > 0.0f)) continue;
    const int* bx = box + a * kSahBins * 6;
    const uint32_t* cn = count + a * kSahBins;
    float right_area[kSahBins];
    uint32_t right_cnt[kSahBins];
    float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t c = 0;
    for (int k = kSahBins - 1; k > 0; --k) {
      for (int d = 0; d < 3; ++d) { alo[d] = fminf(alo[d], ordered_to_float(bx[k * 6 + d])); ahi[d] = fmaxf(ahi[d], ordered_to_float(bx[k * 6 + 3 + d])); }
      c += cn[k];
      right_area[k] = sah_area(alo, ahi);
      right_cnt[k] = c;
    }
    for (int d = 0; d < 3; ++d) { alo[d] = INFINITY; ahi[d] = -INFINITY; }
    c = 0;
    for (int k = 0; k < kSahBins - 1; ++k) {
      for (int d = 0; d < 3; ++d) { alo[d] = fminf(alo[d], ordered_to_float(bx[k * 6 + d])); ahi[d] = fmaxf(ahi[d], ordered_to_float(bx[k * 6 + 3 + d])); }
      c += cn[k];
      if (c == 0 || right_cnt[k + 1] == 0) continue;
      const float cost = sah_area(alo, ahi) * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
      if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
    }
  }
  uint32_t n_left = 0;
  if (best_axis >= 0)
    for (int k = 0; k <= best_bin; ++k) n_left += count[best_axis * kSahBins + k];
  axis_out = best_axis;
  bin_out = best_bin;
  n_left_out = n_left;
}
// Children of node t once its range is in order in idx_out: leaves are linked, longer ranges queued for the next level.
__device__ inline void sah_emit_children(const SahTask& t, uint32_t mid, const uint32_t* idx_out, int n_leaves, int2* children, int* parent,
                                         SahTask* queue_out, uint32_t* n_out) {
  int link[2];
  const uint32_t rb[2] = {t.b, mid}, re[2] = {mid, t.e};
  int next_id = t.node + 1;
  for (int s = 0; s < 2; ++s) {
    const uint32_t c = re[s] - rb[s];
    if (c == 1) {
      const uint32_t leaf = idx_out[rb[s]];
      link[s] = ~(int)leaf;
      parent[(n_leaves - 1) + (int)leaf] = t.node;
    } else {
      link[s] = next_id;
      parent[next_id] = t.node;
      queue_out[atomicAdd(n_out, 1u)] = SahTask{rb[s], re[s], next_id};
      next_id += (int)c - 1;
    }
  }
  children[t.node] = make_int2(link[0], link[1]);
}
// Stable partition of the elements [begin, end) of node t (a whole range or one chunk of it) into idx_out: lefts go to
// t.b + done_left..., rights to mid + done_right..., a tile of B elements at a time.  s_wave_sum: B / 64 words, s_done: 2.
template <int B>
__device__ inline void sah_scatter(uint32_t begin, uint32_t end, uint32_t node_b, uint32_t mid, int axis, int bin, float lo_a, float scale_a,
                                   uint32_t done_left, uint32_t done_right, const uint32_t* __restrict__ idx_in, uint32_t* __restrict__ idx_out,
                                   const float4* __restrict__ leaf_lo, const float4* __restrict__ leaf_hi, uint32_t* s_wave_sum, uint32_t* s_done) {
  const int tid = threadIdx.x;
  if (tid == 0) { s_done[0] = done_left; s_done[1] = done_right; }
  __syncthreads();
  for (uint32_t base = begin; base < end; base += B) {
    const uint32_t i = base + tid;
    uint32_t p = 0;
    bool left = false;
    const bool valid = i < end;
    if (valid) {
      p = idx_in[i];
      const float4 l = leaf_lo[p], h = leaf_hi[p];
      const float c = axis == 0 ? 0.5f * (l.x + h.x) : (axis == 1 ? 0.5f * (l.y + h.y) : 0.5f * (l.z + h.z));
      left = sah_bin_of(c, lo_a, scale_a) <= bin;
    }
    const unsigned long long m = __ballot(valid && left);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1ull));
    if ((tid & 63) == 0) s_wave_sum[tid >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, tile_left = 0;
    for (int w = 0; w < B / 64; ++w) {
      if (w < (tid >> 6)) before += s_wave_sum[w];
      tile_left += s_wave_sum[w];
    }
    const uint32_t lpos = before + in_wave;                 // lefts of the tile before this element
    if (valid) {
      if (left) idx_out[node_b + s_done[0] + lpos] = p;
      else idx_out[mid + s_done[1] + ((uint32_t)tid - lpos)] = p;
    }
    __syncthreads();
    if (tid == 0) {
      const uint32_t tile = min((uint32_t)B, end - base);
      s_done[0] += tile_left;
      s_done[1] += tile - tile_left;
    }
    __syncthreads();
  }
}

// One block per node of the level.  A one-wave block whose node holds at most 64 leaves finishes the whole subtree by
// itself (children go on a stack in LDS, each reading the index array its parent wrote): the wide bottom of the tree --
// millions of two- and three-leaf nodes over six or seven levels -- costs one level.
template <int kSahBlock>
__global__ void __launch_bounds__(kSahBlock) k_sah_level(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in,
                                                         uint32_t* idx_a /* read by the level's nodes */, uint32_t* idx_b /* written */,
                                                         SahTask* __restrict__ queue_out, uint32_t* __restrict__ n_out, int n_leaves,
                                                         const float4* __restrict__ leaf_lo, const float4* __restrict__ leaf_hi,
                                                         int2* __restrict__ children, int* __restrict__ parent, int force_halve) {
  if (blockIdx.x >= *n_in) return;
  constexpr int kLocalLeaves = 64;
  SahTask t = queue_in[blockIdx.x];
  const bool local = kSahBlock == 64 && t.e - t.b <= (uint32_t)kLocalLeaves;
  const int tid = threadIdx.x;
  __shared__ float s_red[6][kSahBlock / 64];
  __shared__ float s_clo[3], s_chi[3], s_scale[3];
  __shared__ int s_box[3][kSahBins][6];      // ordered-int lo xyz, hi xyz
  __shared__ uint32_t s_count[3][kSahBins];
  __shared__ int s_axis, s_bin;
  __shared__ uint32_t s_n_left, s_wave_sum[kSahBlock / 64], s_done[2];
  __shared__ SahTask s_stack[kSahBlock == 64 ? kLocalLeaves : 1];   // bit 31 of .node: the task reads idx_b (its parent wrote there)
  __shared__ int s_sp;
  if (tid == 0) s_sp = 0;
  bool flip = false;
  for (;;) {
    const uint32_t* idx_in = flip ? idx_b : idx_a;
    uint32_t* idx_out = flip ? idx_a : idx_b;
    const uint32_t cnt = t.e - t.b;
    uint32_t mid = t.b + cnt / 2;
    bool found = false;
    if (cnt > 2 && !force_halve) {
      // ---- centroid bounds ----
      float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (uint32_t i = t.b + tid; i < t.e; i += kSahBlock) {
        const uint32_t p = idx_in[i];
        const float4 l = leaf_lo[p], h = leaf_hi[p];
        const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
        for (int k = 0; k < 3; ++k) { clo[k] = fminf(clo[k], c[k]); chi[k] = fmaxf(chi[k], c[k]); }
      }
      for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; ++k) { clo[k] = fminf(clo[k], __shfl_xor(clo[k], off)); chi[k] = fmaxf(chi[k], __shfl_xor(chi[k], off)); }
      if ((tid & 63) == 0)
        for (int k = 0; k < 3; ++k) { s_red[k][tid >> 6] = clo[k]; s_red[3 + k][tid >> 6] = chi[k]; }
      for (int i = tid; i < 3 * kSahBins * 6; i += kSahBlock) (&s_box[0][0][0])[i] = (i % 6) < 3 ? float_to_ordered(INFINITY) : float_to_ordered(-INFINITY);
      for (int i = tid; i < 3 * kSahBins; i += kSahBlock) (&s_count[0][0])[i] = 0;
      __syncthreads();
      if (tid < 3) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < kSahBlock / 64; ++w) { l = fminf(l, s_red[tid][w]); h = fmaxf(h, s_red[3 + tid][w]); }
        s_clo[tid] = l;
        s_chi[tid] = h;
        s_scale[tid] = h - l > 0.0f ? (float)kSahBins / (h - l) : 0.0f;
      }
      __syncthreads();
      // ---- binning ----
      for (uint32_t i = t.b + tid; i < t.e; i += kSahBlock) {
        const uint32_t p = idx_in[i];
        const float4 l = leaf_lo[p], h = leaf_hi[p];
        const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
        for (int a = 0; a < 3; ++a) {
          if (!(s_scale[a] > 0.0f)) continue;
          const int k = sah_bin_of(c[a], s_clo[a], s_scale[a]);
          atomicMin(&s_box[a][k][0], float_to_ordered(l.x)); atomicMin(&s_box[a][k][1], float_to_ordered(l.y)); atomicMin(&s_box[a][k][2], float_to_ordered(l.z));
          atomicMax(&s_box[a][k][3], float_to_ordered(h.x)); atomicMax(&s_box[a][k][4], float_to_ordered(h.y)); atomicMax(&s_box[a][k][5], float_to_ordered(h.z));
          atomicAdd(&s_count[a][k], 1u);
        }
      }
      __syncthreads();
      if (tid == 0) {
        int axis, bin;
        uint32_t n_left;
        sah_pick_split(&s_box[0][0][0], &s_count[0][0], s_scale, axis, bin, n_left);
        s_axis = axis;
        s_bin = bin;
        s_n_left = n_left;
      }
      __syncthreads();
      found = s_axis >= 0 && s_n_left > 0 && s_n_left < cnt;
    }
    if (found) {
      mid = t.b + s_n_left;
      sah_scatter<kSahBlock>(t.b, t.e, t.b, mid, s_axis, s_bin, s_clo[s_axis], s_scale[s_axis], 0u, 0u, idx_in, idx_out, leaf_lo, leaf_hi, s_wave_sum, s_done);
    } else {
      for (uint32_t i = t.b + tid; i < t.e; i += kSahBlock) idx_out[i] = idx_in[i];   // two leaves, or every centroid in one place: halve the range as it stands
      __syncthreads();
    }
    if (!local) {
      if (tid == 0) sah_emit_children(t, mid, idx_out, n_leaves, children, parent, queue_out, n_out);
      return;
    }
    // ---- this wave goes on with the children ----
    if (tid == 0) {
      int link[2];
      const uint32_t rb[2] = {t.b, mid}, re[2] = {mid, t.e};
      int next_id = t.node + 1;
      for (int s = 0; s < 2; ++s) {
        const uint32_t c = re[s] - rb[s];
        if (c == 1) {
          const uint32_t leaf = idx_out[rb[s]];
          link[s] = ~(int)leaf;
          parent[(n_leaves - 1) + (int)leaf] = t.node;
        } else {
          link[s] = next_id;
          parent[next_id] = t.node;
          s_stack[s_sp++] = SahTask{rb[s], re[s], next_id | (flip ? 0 : (int)0x80000000)};   // the child reads what this node wrote
          next_id += (int)c - 1;
        }
      }
      children[t.node] = make_int2(link[0], link[1]);
    }
    __syncthreads();
    if (s_sp == 0) return;
    t = s_stack[s_sp - 1];
    __syncthreads();
    if (tid == 0) --s_sp;
    flip = (t.node & (int)0x80000000) != 0;
    t.node &= 0x7FFFFFFF;
    __syncthreads();
  }
}
// ---- the top levels: long ranges, several blocks per node ("chunks" of kSahChunk elements) ----
// A level whose mean range is long would leave a handful of blocks looping over millions of elements (level 0 of 3.6 M
// leaves: 70 ms in one block).  Here every pass of the level runs over (node, chunk) pairs: centroid bounds and bins are
// combined per node with global atomics on ordered ints (min / max / counts: the result does not depend on the order), the
// split is picked by one thread per node with the same routine, lefts are counted per chunk, and every chunk scatters
// its elements behind those of the chunks before it -- the stable partition of the one-block version, hence the same tree.
constexpr int kSahMaxSplitLevels = 256;   // levels of SAH splits before the rest of the tree is built by halving ranges
constexpr uint32_t kSahChunk = 4096;
constexpr uint32_t kSahWideMean = 16384;   // levels whose mean range is at least this long take the several-blocks-per-node path
constexpr int kSahWideBlock = 1024;
struct SahWideNode {
  int bounds[6];                       // ordered-int centroid lo xyz, hi xyz
  int box[3 * kSahBins * 6];
  uint32_t count[3 * kSahBins];
  int axis, bin;
  uint32_t n_left, found;
  uint32_t chunk_base;                 // number of the node's first chunk in the level
};
__device__ __forceinline__ uint32_t sah_chunks_of(const SahTask& t) { return (t.e - t.b + kSahChunk - 1) / kSahChunk; }
// block -> (node, chunk of the node); false when the block is beyond the level's chunks
__device__ __forceinline__ bool sah_locate(const SahWideNode* __restrict__ wide, uint32_t n_nodes, uint32_t total_chunks, uint32_t block,
                                           uint32_t& node, uint32_t& chunk) {
  if (block >= total_chunks) return false;
  uint32_t lo = 0, hi = n_nodes - 1;
  while (lo < hi) {   // last node whose first chunk is <= block
    const uint32_t m = (lo + hi + 1) >> 1;
    if (wide[m].chunk_base <= block) lo = m; else hi = m - 1;
  }
  node = lo;
  chunk = block - wide[lo].chunk_base;
  return true;
}
__global__ void __launch_bounds__(1024) k_wide_plan(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in, SahWideNode* __restrict__ wide,
                                                    uint32_t* __restrict__ total_chunks) {
  const uint32_t n_nodes = *n_in;
  for (uint32_t i = threadIdx.x; i < n_nodes; i += blockDim.x) {
    SahWideNode& w = wide[i];
    for (int k = 0; k < 3; ++k) { w.bounds[k] = float_to_ordered(INFINITY); w.bounds[3 + k] = float_to_ordered(-INFINITY); }
    for (int k = 0; k < 3 * kSahBins * 6; ++k) w.box[k] = (k % 6) < 3 ? float_to_ordered(INFINITY) : float_to_ordered(-INFINITY);
    for (int k = 0; k < 3 * kSahBins; ++k) w.count[k] = 0;
    w.axis = -1; w.bin = -1; w.n_left = 0; w.found = 0;
  }
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (uint32_t i = 0; i < n_nodes; ++i) { wide[i].chunk_base = acc; acc += sah_chunks_of(queue_in[i]); }
    *total_chunks = acc;
  }
}
__global__ void __launch_bounds__(kSahWideBlock) k_wide_bounds(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in,
                                                               const uint32_t* __restrict__ total_chunks, SahWideNode* __restrict__ wide,
                                                               const uint32_t* __restrict__ idx_in, const float4* __restrict__ leaf_lo,
                                                               const float4* __restrict__ leaf_hi) {
  uint32_t node, chunk;
  if (!sah_locate(wide, *n_in, *total_chunks, blockIdx.x, node, chunk)) return;
  const SahTask t = queue_in[node];
  if (t.e - t.b <= 2) return;
  const uint32_t begin = t.b + chunk * kSahChunk, end = min(t.e, begin + kSahChunk);
  __shared__ float s_red[6][kSahWideBlock / 64];
  const int tid = threadIdx.x;
  float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = begin + tid; i < end; i += kSahWideBlock) {
    const uint32_t p = idx_in[i];
    const float4 l = leaf_lo[p], h = leaf_hi[p];
    const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
    for (int k = 0; k < 3; ++k) { clo[k] = fminf(clo[k], c[k]); chi[k] = fmaxf(chi[k], c[k]); }
  }
  for (int off = 32; off > 0; off >>= 1)
    for (int k = 0; k < 3; ++k) { clo[k] = fminf(clo[k], __shfl_xor(clo[k], off)); chi[k] = fmaxf(chi[k], __shfl_xor(chi[k], off)); }
  if ((tid & 63) == 0)
    for (int k = 0; k < 3; ++k) { s_red[k][tid >> 6] = clo[k]; s_red[3 + k][tid >> 6] = chi[k]; }
  __syncthreads();
  if (tid < 3) {
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < kSahWideBlock / 64; ++w) { l = fminf(l, s_red[tid][w]); h = fmaxf(h, s_red[3 + tid][w]); }
    atomicMin(&wide[node].bounds[tid], float_to_ordered(l));
    atomicMax(&wide[node].bounds[3 + tid], float_to_ordered(h));
  }
}
__device__ __forceinline__ void sah_wide_scale(const SahWideNode& w, float clo[3], float scale[3]) {
  for (int a = 0; a < 3; ++a) {
    const float l = ordered_to_float(w.bounds[a]), h = ordered_to_float(w.bounds[3 + a]);
    clo[a] = l;
    scale[a] = h - l > 0.0f ? (float)kSahBins / (h - l) : 0.0f;
  }
}
__global__ void __launch_bounds__(kSahWideBlock) k_wide_bin(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in,
                                                            const uint32_t* __restrict__ total_chunks, SahWideNode* __restrict__ wide,
                                                            const uint32_t* __restrict__ idx_in, const float4* __restrict__ leaf_lo,
                                                            const float4* __restrict__ leaf_hi) {
  uint32_t node, chunk;
  if (!sah_locate(wide, *n_in, *total_chunks, blockIdx.x, node, chunk)) return;
  const SahTask t = queue_in[node];
  if (t.e - t.b <= 2) return;
  const uint32_t begin = t.b + chunk * kSahChunk, end = min(t.e, begin + kSahChunk);
  __shared__ int s_box[3 * kSahBins * 6];
  __shared__ uint32_t s_count[3 * kSahBins];
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * kSahBins * 6; i += kSahWideBlock) s_box[i] = (i % 6) < 3 ? float_to_ordered(INFINITY) : float_to_ordered(-INFINITY);
  for (int i = tid; i < 3 * kSahBins; i += kSahWideBlock) s_count[i] = 0;
  float clo[3], scale[3];
  sah_wide_scale(wide[node], clo, scale);
  __syncthreads();
  for (uint32_t i = begin + tid; i < end; i += kSahWideBlock) {
    const uint32_t p = idx_in[i];
    const float4 l = leaf_lo[p], h = leaf_hi[p];
    const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
    for (int a = 0; a < 3; ++a) {
      if (!(scale[a] > 0.0f)) continue;
      int* bx = &s_box[(a * kSahBins + sah_bin_of(c[a], clo[a], scale[a])) * 6];
      atomicMin(&bx[0], float_to_ordered(l.x)); atomicMin(&bx[1], float_to_ordered(l.y)); atomicMin(&bx[2], float_to_ordered(l.z));
      atomicMax(&bx[3], float_to_ordered(h.x)); atomicMax(&bx[4], float_to_ordered(h.y)); atomicMax(&bx[5], float_to_ordered(h.z));
      atomicAdd(&s_count[a * kSahBins + sah_bin_of(c[a], clo[a], scale[a])], 1u);
    }
  }
  __syncthreads();
  for (int i = tid; i < 3 * kSahBins; i += kSahWideBlock) {
    if (s_count[i] == 0) continue;
    atomicAdd(&wide[node].count[i], s_count[i]);
    for (int d = 0; d < 3; ++d) { atomicMin(&wide[node].box[i * 6 + d], s_box[i * 6 + d]); atomicMax(&wide[node].box[i * 6 + 3 + d], s_box[i * 6 + 3 + d]); }
  }
}
__global__ void __launch_bounds__(64) k_wide_pick(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in, SahWideNode* __restrict__ wide) {
  const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= *n_in) return;
  const SahTask t = queue_in[node];
  const uint32_t cnt = t.e - t.b;
  if (cnt <= 2) return;
  float clo[3], scale[3];
  sah_wide_scale(wide[node], clo, scale);
  int axis, bin;
  uint32_t n_left;
  sah_pick_split(wide[node].box, wide[node].count, scale, axis, bin, n_left);
  wide[node].axis = axis;
  wide[node].bin = bin;
  wide[node].n_left = n_left;
  wide[node].found = (axis >= 0 && n_left > 0 && n_left < cnt) ? 1u : 0u;
}
__global__ void __launch_bounds__(kSahWideBlock) k_wide_count(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in,
                                                              const uint32_t* __restrict__ total_chunks, const SahWideNode* __restrict__ wide,
                                                              const uint32_t* __restrict__ idx_in, const float4* __restrict__ leaf_lo,
                                                              const float4* __restrict__ leaf_hi, uint32_t* __restrict__ chunk_left) {
  uint32_t node, chunk;
  if (!sah_locate(wide, *n_in, *total_chunks, blockIdx.x, node, chunk)) return;
  const SahWideNode& w = wide[node];
  if (!w.found) return;
  const SahTask t = queue_in[node];
  const uint32_t begin = t.b + chunk * kSahChunk, end = min(t.e, begin + kSahChunk);
  __shared__ uint32_t s_sum[kSahWideBlock / 64];
  float clo[3], scale[3];
  sah_wide_scale(w, clo, scale);
  const int axis = w.axis, bin = w.bin, tid = threadIdx.x;
  uint32_t mine = 0;
  for (uint32_t i = begin + tid; i < end; i += kSahWideBlock) {
    const uint32_t p = idx_in[i];
    const float4 l = leaf_lo[p], h = leaf_hi[p];
    const float c = axis == 0 ? 0.5f * (l.x + h.x) : (axis == 1 ? 0.5f * (l.y + h.y) : 0.5f * (l.z + h.z));
    mine += sah_bin_of(c, clo[axis], scale[axis]) <= bin ? 1u : 0u;
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((tid & 63) == 0) s_sum[tid >> 6] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
    for (int k = 0; k < kSahWideBlock / 64; ++k) total += s_sum[k];
    chunk_left[blockIdx.x] = total;
  }
}
__global__ void __launch_bounds__(kSahWideBlock) k_wide_scatter(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in,
                                                                const uint32_t* __restrict__ total_chunks, const SahWideNode* __restrict__ wide,
                                                                const uint32_t* __restrict__ chunk_left, const uint32_t* __restrict__ idx_in,
                                                                uint32_t* __restrict__ idx_out, const float4* __restrict__ leaf_lo,
                                                                const float4* __restrict__ leaf_hi) {
  uint32_t node, chunk;
  if (!sah_locate(wide, *n_in, *total_chunks, blockIdx.x, node, chunk)) return;
  const SahWideNode& w = wide[node];
  const SahTask t = queue_in[node];
  const uint32_t begin = t.b + chunk * kSahChunk, end = min(t.e, begin + kSahChunk);
  const int tid = threadIdx.x;
  if (!w.found) {   // the range stays as it is
    for (uint32_t i = begin + tid; i < end; i += kSahWideBlock) idx_out[i] = idx_in[i];
    return;
  }
  __shared__ uint32_t s_sum[kSahWideBlock / 64], s_wave_sum[kSahWideBlock / 64], s_done[2], s_before;
  // lefts in the chunks of this node before this one
  uint32_t mine = 0;
  for (uint32_t c = tid; c < chunk; c += kSahWideBlock) mine += chunk_left[w.chunk_base + c];
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
  if ((tid & 63) == 0) s_sum[tid >> 6] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
    for (int k = 0; k < kSahWideBlock / 64; ++k) total += s_sum[k];
    s_before = total;
  }
  __syncthreads();
  const uint32_t left_before = s_before, right_before = chunk * kSahChunk - left_before;
  float clo[3], scale[3];
  sah_wide_scale(w, clo, scale);
  sah_scatter<kSahWideBlock>(begin, end, t.b, t.b + w.n_left, w.axis, w.bin, clo[w.axis], scale[w.axis], left_before, right_before, idx_in, idx_out, leaf_lo,
                             leaf_hi, s_wave_sum, s_done);
}
__global__ void __launch_bounds__(64) k_wide_children(const SahTask* __restrict__ queue_in, const uint32_t* __restrict__ n_in,
                                                      const SahWideNode* __restrict__ wide, const uint32_t* __restrict__ idx_out, int n_leaves,
                                                      int2* __restrict__ children, int* __restrict__ parent, SahTask* __restrict__ queue_out,
                                                      uint32_t* __restrict__ n_out) {
  const uint32_t node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= *n_in) return;
  const SahTask t = queue_in[node];
  const uint32_t mid = wide[node].found ? t.b + wide[node].n_left : t.b + (t.e - t.b) / 2;
  sah_emit_children(t, mid, idx_out, n_leaves, children, parent, queue_out, n_out);
}

__global__ void k_sah_init(uint32_t n, uint32_t* __restrict__ idx, SahTask* __restrict__ queue, uint32_t* __restrict__ counts, int* __restrict__ parent) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = i;
  if (i == 0) {
    queue[0] = SahTask{0u, n, 0};
    counts[0] = 1;
    counts[1] = 0;
    parent[0] = -1;
  }
}

// ---- layout: depth-first (pre-order) numbering, so every subtree is one contiguous run of nodes and a node's left
// child is its neighbour.  counts[t] = inner nodes in the subtree of t (k_level_up).
__global__ void __launch_bounds__(256) k_dfs_ids(int n, const int2* __restrict__ children, const int* __restrict__ parent,
                                                 const int* __restrict__ counts, int* __restrict__ new_id) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n - 1) return;
  int id = 0;
  for (int cur = t, p = parent[t]; p >= 0; cur = p, p = parent[p]) {
    const int2 c = children[p];
    id += 1 + ((c.y == cur && c.x >= 0) ? counts[c.x] : 0);
  }
  new_id[t] = id;
}

// Quantisation grid from the root box: kBvhGridMax (32 767) cells per axis, stretched by 2^-16 so the top plane stays below it.  15 bits:
// the tracer turns a coordinate into the float 32768 + q with one byte permute (device/wavefront.h box_key).
constexpr float kGridReach = 1.9073486e-6f;   // 2^-19
__global__ void k_grid_params(const float4* __restrict__ node_lo, const float4* __restrict__ node_hi, BvhGrid* __restrict__ grid) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const float lo[3] = {node_lo[0].x, node_lo[0].y, node_lo[0].z}, hi[3] = {node_hi[0].x, node_hi[0].y, node_hi[0].z};
  // the grid reaches kGridReach of the largest coordinate past the bounds on every side, so that the margin the boxes get at
  // quantisation (grid_margin below) has cells to be counted in: across the thin side of a flat scene the grid is then all margin
  float largest = 0.0f;
  for (int k = 0; k < 3; ++k) largest = fmaxf(largest, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
  const float reach = kGridReach * largest;
  for (int k = 0; k < 3; ++k) {
    float ext = (hi[k] - lo[k]) + 2.0f * reach;
    if (!(ext > 0.0f)) ext = 1.0f;
    const float cell = ext * 1.00002f / (float)kBvhGridMax;
    grid->lo[k] = (lo[k] - reach) - 0.5f * cell;
    grid->cell[k] = cell;
    grid->inv_cell[k] = 1.0f / cell;
  }
}

// grid coordinate of a world coordinate; the SAME expression maps the ray origin in the tracer
__device__ __forceinline__ float to_grid(float x, float lo, float inv_cell) { return (x - lo) * inv_cell; }
// `margin`: cells every box grows by on that axis beyond the 1/16 (grid_margin below)
__device__ __forceinline__ uint32_t quant_lo(float x, float lo, float inv_cell, float margin) {
  // 1/16 cell of slack covers the rounding of to_grid() at grid coordinates up to 32767 (ulp 2^-9)
  const float g = floorf(to_grid(x, lo, inv_cell) - 0.0625f - margin);
  return (uint32_t)fminf(fmaxf(g, 0.0f), (float)kBvhGridMax);
}
__device__ __forceinline__ uint32_t quant_hi(float x, float lo, float inv_cell, float margin) {
  const float g = ceilf(to_grid(x, lo, inv_cell) + 0.0625f + margin);
  return (uint32_t)fminf(fmaxf(g, 0.0f), (float)kBvhGridMax);
}
// The tracer leaves a box out when its entry distance lies behind the hit it already has, so a box's entry must not round past the
// distance of a triangle inside it: the plane distance (an fma in grid space) and the triangle's t (sheared coordinates) each carry a
// few ulps of the coordinates involved.  Where a cell is a fair fraction of the scene the 1/16 above is that margin; across the thin
// side of a FLAT scene (a floor plan: every box of no thickness, cells of 1e-13 units, the builders' relative pad nothing near
// coordinate 0) it is not, and exact ties between coincident triangles went to whichever was met first (tools/gpu_fuzz_parity.py,
// seed 61907).  So every box also grows by 2^-19 of the largest coordinate of the grid's bounds, whatever that is in cells
// (k_grid_params lets the grid reach that far past the bounds).
__device__ __forceinline__ void grid_margin(const BvhGrid& g, float* margin) {
  float largest = 0.0f;
  for (int k = 0; k < 3; ++k) largest = fmaxf(largest, fmaxf(fabsf(g.lo[k]), fabsf(g.lo[k] + (float)kBvhGridMax * g.cell[k])));
  for (int k = 0; k < 3; ++k) margin[k] = fminf((kGridReach * largest) * g.inv_cell[k], 65536.0f);
}

// ---- 4-wide collapse.  A BVH4 node starts from the two children of a binary node and keeps opening the inner child
// with the LARGEST surface area (the one a ray is most likely to enter anyway) until it has four children or only leaves
// are left; an opened child's two children take its place, so the children stay in the left-to-right order of the binary
// hierarchy (the order that breaks distance ties in the tracer).  The binary nodes that head a BVH4 node are found top
// down, one launch per level of the binary hierarchy (a node marks its final children, which lie 1-3 levels below).
// flags[dfs id] = 1 for those heads, so an exclusive scan over the depth-first order numbers the BVH4 nodes depth-first.
// (Folding every odd level instead -- children = grandchildren -- left 3.0 children per node; this leaves 3.4+.)
template <int W>
struct Kids {
  int link[W];
  int n;
};
using Kids4 = Kids<4>;
__device__ __forceinline__ float box_area(const float4* __restrict__ node_lo, const float4* __restrict__ node_hi, int slot) {
  const float4 l = node_lo[slot], h = node_hi[slot];
  const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
// W = 4: the nodes every tracer reads; W = 8: the 128-byte nodes of the tracer for small tile shares (types.h BvhNode8) -- the same
// rule carried on until eight children are open
template <int W>
__device__ __forceinline__ Kids<W> collapse_wide(int node, const int2* __restrict__ children, const float4* __restrict__ node_lo,
                                                 const float4* __restrict__ node_hi) {
  Kids<W> k;
  const int2 c = children[node];
  float area[W];
#pragma unroll
  for (int i = 0; i < W; ++i) { k.link[i] = kBvhEmptyChild; area[i] = -1.0f; }
  k.link[0] = c.x; k.link[1] = c.y;
  k.n = 2;
  area[0] = c.x >= 0 ? box_area(node_lo, node_hi, c.x) : -1.0f;
  area[1] = c.y >= 0 ? box_area(node_lo, node_hi, c.y) : -1.0f;
  while (k.n < W) {
    int j = -1;
    float best = -1.0f;
    for (int i = 0; i < W; ++i)
      if (i < k.n && k.link[i] >= 0 && area[i] > best) { best = area[i]; j = i; }   // ties: the leftmost
    if (j < 0) break;   // only leaves left
    const int2 g = children[k.link[j]];
    for (int i = W - 1; i > 0; --i)
      if (i > j + 1) { k.link[i] = k.link[i - 1]; area[i] = area[i - 1]; }
    k.link[j] = g.x; area[j] = g.x >= 0 ? box_area(node_lo, node_hi, g.x) : -1.0f;
    k.link[j + 1] = g.y; area[j + 1] = g.y >= 0 ? box_area(node_lo, node_hi, g.y) : -1.0f;
    ++k.n;
  }
  return k;
}
__device__ __forceinline__ Kids4 collapse4(int node, const int2* __restrict__ children, const float4* __restrict__ node_lo, const float4* __restrict__ node_hi) {
  return collapse_wide<4>(node, children, node_lo, node_hi);
}
// head[t] = level of the BVH4 node headed by binary node t (root = 1), 0 = folded into an ancestor
template <int W>
__global__ void __launch_bounds__(256) k_mark_heads(int n, int level, const int* __restrict__ node_depth, const int2* __restrict__ children,
                                                    const float4* __restrict__ node_lo, const float4* __restrict__ node_hi,
                                                    int* head, int* __restrict__ max_level) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n - 1 || node_depth[t] != level) return;
  const int mine = t == 0 ? 1 : head[t];
  if (t == 0) head[0] = 1;
  if (mine == 0) return;
  const Kids<W> k = collapse_wide<W>(t, children, node_lo, node_hi);
  for (int i = 0; i < W; ++i)
    if (i < k.n && k.link[i] >= 0) head[k.link[i]] = mine + 1;
  atomicMax(max_level, mine);
}
__global__ void __launch_bounds__(256) k_head_flags(int n, const int* __restrict__ head, const int* __restrict__ new_id,
                                                    unsigned long long* __restrict__ flags) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n - 1) return;
  flags[new_id[t]] = head[t] ? 1ull : 0ull;
}

__global__ void __launch_bounds__(256) k_emit_nodes4(int n, const int2* __restrict__ children, const float4* __restrict__ node_lo,
                                                     const float4* __restrict__ node_hi, const BvhGrid* __restrict__ grid,
                                                     const int* __restrict__ new_id, const unsigned long long* __restrict__ flags,
                                                     const unsigned long long* __restrict__ pos, const unsigned long long* __restrict__ slot,
                                                     uint32_t leaf_links_by_number, BvhNode4* __restrict__ nodes, float* __restrict__ sah) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int2 c = children[i];
  {
    // SAH cost numerator of the binary hierarchy: surface areas of inner nodes (1.2) and leaves (1.0), normalised on the host
    auto area = [](float4 l, float4 h) { float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z; return 2.0f * (dx * dy + dy * dz + dz * dx); };
    float acc = 1.2f * area(node_lo[i], node_hi[i]);
    if (c.x < 0) acc += area(node_lo[(n - 1) + ~c.x], node_hi[(n - 1) + ~c.x]);
    if (c.y < 0) acc += area(node_lo[(n - 1) + ~c.y], node_hi[(n - 1) + ~c.y]);
    atomicAdd(sah, acc);   // reported only (the sum's order is not fixed)
  }
  if (!flags[new_id[i]]) return;   // folded into an ancestor
  const Kids4 kk = collapse4(i, children, node_lo, node_hi);
  const int nk = kk.n;
  const int* kids = kk.link;
  const BvhGrid g = *grid;
  float gm[3];
  grid_margin(g, gm);
  BvhNode4 nd;
  for (int k = 0; k < 4; ++k) {
    const int ch = kids[k];
    if (k >= nk) {
      nd.w[3 * k] = nd.w[3 * k + 1] = nd.w[3 * k + 2] = kBvhGridMax;   // lo = the grid's top, hi = 0 (the tracer skips the slot by its link)
      nd.w[12 + k] = (uint32_t)kBvhEmptyChild;
      continue;
    }
    const int box = ch >= 0 ? ch : (n - 1) + ~ch;
    const float4 l = node_lo[box], h = node_hi[box];
    const uint32_t q[6] = {quant_lo(l.x, g.lo[0], g.inv_cell[0], gm[0]), quant_lo(l.y, g.lo[1], g.inv_cell[1], gm[1]), quant_lo(l.z, g.lo[2], g.inv_cell[2], gm[2]),
                           quant_hi(h.x, g.lo[0], g.inv_cell[0], gm[0]), quant_hi(h.y, g.lo[1], g.inv_cell[1], gm[1]), quant_hi(h.z, g.lo[2], g.inv_cell[2], gm[2])};
    nd.w[3 * k] = q[0] | (q[3] << 16);       // one word per axis: lo | hi << 16
    nd.w[3 * k + 1] = q[1] | (q[4] << 16);
    nd.w[3 * k + 2] = q[2] | (q[5] << 16);
    // inner: its number among the BVH4 nodes; leaf: ~(first slot of the leaf in bvh_tris), or ~(leaf number) when the tracer reads
    // per-leaf records that name the slot (LbvhInputs::emit_quads)
    nd.w[12 + k] = (uint32_t)(ch >= 0 ? (int)pos[new_id[ch]] : (leaf_links_by_number ? ch : ~(int)slot[~ch]));
  }
  nodes[pos[new_id[i]]] = nd;
}

// The same hierarchy collapsed eight wide (types.h BvhNode8): heads found by k_mark_heads<8>, numbered depth-first like the 4-wide
// nodes; same grid, same padding, leaf links by leaf number (only the flattened build with per-leaf records carries these nodes).
__global__ void __launch_bounds__(256) k_emit_nodes8(int n, const int2* __restrict__ children, const float4* __restrict__ node_lo,
                                                     const float4* __restrict__ node_hi, const BvhGrid* __restrict__ grid,
                                                     const int* __restrict__ new_id, const unsigned long long* __restrict__ flags,
                                                     const unsigned long long* __restrict__ pos, BvhNode8* __restrict__ nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1 || !flags[new_id[i]]) return;
  const Kids<8> kk = collapse_wide<8>(i, children, node_lo, node_hi);
  const BvhGrid g = *grid;
  float gm[3];
  grid_margin(g, gm);
  BvhNode8 nd;
  for (int k = 0; k < 8; ++k) {
    const int ch = kk.link[k];
    if (k >= kk.n) {
      nd.w[3 * k] = nd.w[3 * k + 1] = nd.w[3 * k + 2] = kBvhGridMax;   // an inverted box: no ray enters it
      nd.w[24 + k] = (uint32_t)kBvhEmptyChild;
      continue;
    }
    const int box = ch >= 0 ? ch : (n - 1) + ~ch;
    const float4 l = node_lo[box], h = node_hi[box];
    nd.w[3 * k] = quant_lo(l.x, g.lo[0], g.inv_cell[0], gm[0]) | (quant_hi(h.x, g.lo[0], g.inv_cell[0], gm[0]) << 16);
    nd.w[3 * k + 1] = quant_lo(l.y, g.lo[1], g.inv_cell[1], gm[1]) | (quant_hi(h.y, g.lo[1], g.inv_cell[1], gm[1]) << 16);
    nd.w[3 * k + 2] = quant_lo(l.z, g.lo[2], g.inv_cell[2], gm[2]) | (quant_hi(h.z, g.lo[2], g.inv_cell[2], gm[2]) << 16);
    nd.w[24 + k] = (uint32_t)(ch >= 0 ? (int)pos[new_id[ch]] : ch);
  }
  nodes[pos[new_id[i]]] = nd;
}

// Per-leaf shading records: everything raytrace_hit.rchit reads for a hit (3 packed vertices, the triangle's
// derivatives, material and transform ids) gathered into one contiguous 128-byte record so that k_shade fetches it
// with 8 dwordx4 loads instead of walking leaf -> instance -> indices -> vertices -> derivatives (14 scattered loads,
// three levels of dependent latency).
__global__ void __launch_bounds__(256) k_shade_records(uint32_t n, const BvhTri* __restrict__ tris, const RTInstance* __restrict__ instances,
                                                       const uint32_t* __restrict__ indices, const float4* __restrict__ vertices,
                                                       const float4* __restrict__ derivatives, const uint32_t* __restrict__ xf_identity,
                                                       float4* __restrict__ out) {
  const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
  if (leaf >= n) return;
  const BvhTri t = tris[leaf];
  const RTInstance in = instances[t.instance];
  const uint32_t tri_id = in.index_offset / 3u + (t.prim_flags & kTriPrimMask);
  float4* r = out + 8 * (size_t)leaf;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint32_t v = indices[3u * tri_id + k];
    r[2 * k] = vertices[2u * v];
    r[2 * k + 1] = vertices[2u * v + 1u];
  }
  const float4 dn = derivatives[3u * tri_id], du = derivatives[3u * tri_id + 1u];
  r[6] = make_float4(dn.x, dn.y, dn.z, __uint_as_float(in.material_id));
  r[7] = make_float4(du.x, du.y, du.z, __uint_as_float(in.transform_id | (xf_identity[in.transform_id] ? 0x80000000u : 0u)));
}

// The alpha test's inputs per triangle slot (types.h DeviceScene::alpha_recs): the three texture coordinates out of the shading record
// and the descriptor of the material's opacity map, side by side.  A slot whose material has no opacity map gets a record nobody reads.
__global__ void __launch_bounds__(256) k_alpha_records(uint32_t n, const float4* __restrict__ shade_tris, const RTMaterial* __restrict__ materials,
                                                       const TexDesc* __restrict__ tex_desc, float4* __restrict__ out) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const float4* rec = shade_tris + 8u * (size_t)slot;
  const float4 a = rec[1], b = rec[3], c = rec[5];
  const uint32_t opacity = materials[__float_as_uint(rec[6].w)].opacity;
  const TexDesc t = tex_desc[opacity];
  float4* r = out + 3u * (size_t)slot;
  r[0] = make_float4(a.z, a.w, b.z, b.w);
  r[1] = make_float4(c.z, c.w, __uint_as_float(t.offset), __uint_as_float(t.width));
  r[2] = make_float4(__uint_as_float(t.height), __uint_as_float(t.format), 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------------------
#define GLZ_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return e_;               \
  } while (0)

hipError_t launch_derivatives(hipStream_t st, const float4* vertices, const uint32_t* indices, uint32_t n_tris, float4* out) {
  if (n_tris == 0) return hipSuccess;
  // the reference dispatches (triangles/256)+1 groups of 256 (scene.rs:2162)
  hipLaunchKernelGGL(k_tri_derivatives, dim3(n_tris / 256 + 1), dim3(256), 0, st, vertices, indices, n_tris, out);
  return hipGetLastError();
}

static uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Top-of-tree table (types.h kBvhTopNodes): breadth-first from the root, one thread -- 21 dependent 64-byte reads, once per
// scene.  An inner child gets the next free slot and its link in the table becomes kBvhTopFlag | slot; leaves, empty
// slots and inner children beyond the table keep their links.  Unused slots stay zero (never referenced).
__global__ void k_top_table(const BvhNode4* __restrict__ nodes, uint32_t n_nodes, BvhNode4* __restrict__ top) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int s = 0; s < kBvhTopNodes; ++s)
    for (int k = 0; k < 16; ++k) top[s].w[k] = 0u;
  if (n_nodes == 0) return;
  int source[kBvhTopNodes];
  int used = 1;
  source[0] = 0;
  for (int s = 0; s < used; ++s) {
    BvhNode4 nd = nodes[source[s]];
    for (int k = 0; k < 4; ++k) {
      const int link = (int)nd.w[12 + k];
      if (link >= 0 && link != kBvhEmptyChild && (uint32_t)link < n_nodes && used < kBvhTopNodes) {
        source[used] = link;
        nd.w[12 + k] = (uint32_t)(kBvhTopFlag | used);
        ++used;
      }
    }
    top[s] = nd;
  }
}
#ifdef GLZ_NODE48
// 64-byte node -> 48-byte node: origin = the children's lowest planes less one cell, per-axis cell = the smallest power of two (in
// grid cells) that spans the children in 253 steps, every plane moved outwards to the next cell and one more (the plane distances of
// the tracer are exact to 2^-9 of a node cell, which at e >= 5 is more than the 1/16 grid cell the global boxes were padded by).
__global__ void __launch_bounds__(256) k_compress_nodes(const BvhNode4* __restrict__ nodes, uint32_t n, BvhNode48* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const BvhNode4 nd = nodes[i];
  BvhNode48 o;
  uint32_t org[3], ex[3];
  uint32_t lo[4][3], hi[4][3];
  bool used[4];
  for (int k = 0; k < 4; ++k) {
    used[k] = nd.w[12 + k] != (uint32_t)kBvhEmptyChild;
    for (int a = 0; a < 3; ++a) { lo[k][a] = nd.w[3 * k + a] & 0xFFFFu; hi[k][a] = nd.w[3 * k + a] >> 16; }
  }
  for (int a = 0; a < 3; ++a) {
    uint32_t mn = 0xFFFFu, mx = 0u;
    for (int k = 0; k < 4; ++k)
      if (used[k]) { mn = min(mn, lo[k][a]); mx = max(mx, hi[k][a]); }
    if (mn > mx) { mn = 0; mx = 0; }
    uint32_t e = 0;
    while (e < 15u && (mx - mn) > (253u << e)) ++e;
    const uint32_t cell = 1u << e;
    org[a] = mn >= cell ? mn - cell : 0u;
    if ((mx - org[a] + cell - 1u) / cell + 1u > 255u) { ++e; org[a] = mn >= (1u << e) ? mn - (1u << e) : 0u; }
    ex[a] = e;
  }
  o.w[0] = org[0] | (org[1] << 16);
  o.w[1] = org[2] | (ex[0] << 16) | (ex[1] << 20) | (ex[2] << 24);
  for (int a = 0; a < 3; ++a)
    for (int p2 = 0; p2 < 2; ++p2) {
      uint32_t word = 0;
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * p2 + h;
        uint32_t bl = 255u, bh = 0u;   // an unused slot: an inverted box, as in the 64-byte format
        if (used[k]) {
          const uint32_t l = (lo[k][a] - org[a]) >> ex[a], u = ((hi[k][a] - org[a]) + (1u << ex[a]) - 1u) >> ex[a];
          bl = l > 0u ? l - 1u : 0u;
          bh = min(u + 1u, 255u);
        }
        word |= (bl | (bh << 8)) << (16 * h);
      }
      o.w[2 + 2 * a + p2] = word;
    }
  for (int k = 0; k < 4; ++k) o.w[8 + k] = nd.w[12 + k];
  out[i] = o;
}
hipError_t launch_compress_nodes(hipStream_t st, const BvhNode4* nodes, uint32_t n, BvhNode48* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_compress_nodes, dim3((n + 255) / 256), dim3(256), 0, st, nodes, n, out);
  return hipGetLastError();
}
#endif
hipError_t launch_top_table(hipStream_t st, const BvhNode4* nodes, uint32_t n_nodes, BvhNode4* top) {
  hipLaunchKernelGGL(k_top_table, dim3(1), dim3(64), 0, st, nodes, n_nodes, top);
  return hipGetLastError();
}

// Leaves from given boxes (LbvhInputs::given_lo / given_hi): the box arrays the rest of the build works on, a placeholder
// triangle per box that carries its index, and the bounds of the box centres.
__global__ void __launch_bounds__(256) k_given_boxes(const float4* __restrict__ glo, const float4* __restrict__ ghi, uint32_t n, BvhTri* __restrict__ tris,
                                                     float4* __restrict__ box_lo, float4* __restrict__ box_hi, int* __restrict__ scene_bounds) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 l = glo[i], h = ghi[i];
  BvhTri t{};
  t.world_id = i;
  t.instance = i;
  tris[i] = t;
  box_lo[i] = make_float4(l.x, l.y, l.z, 0.0f);
  box_hi[i] = make_float4(h.x, h.y, h.z, 0.0f);
  const float c[3] = {0.5f * (l.x + h.x), 0.5f * (l.y + h.y), 0.5f * (l.z + h.z)};
#pragma unroll
  for (int k = 0; k < 3; ++k) {   // one atomic pair per box: instance counts are small next to triangle counts
    atomicMin(&scene_bounds[k], float_to_ordered(c[k]));
    atomicMax(&scene_bounds[3 + k], float_to_ordered(c[k]));
  }
}

hipError_t launch_shade_records(hipStream_t st, uint32_t n, const BvhTri* tris, const RTInstance* instances, const uint32_t* indices,
                                const float4* vertices, const float4* derivatives, const uint32_t* xf_identity, float4* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_shade_records, dim3((n + 255) / 256), dim3(256), 0, st, n, tris, instances, indices, vertices, derivatives, xf_identity, out);
  return hipGetLastError();
}

hipError_t launch_alpha_records(hipStream_t st, uint32_t n, const float4* shade_tris, const RTMaterial* materials, const TexDesc* tex_desc, float4* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_alpha_records, dim3((n + 255) / 256), dim3(256), 0, st, n, shade_tris, materials, tex_desc, out);
  return hipGetLastError();
}

hipError_t build_lbvh(hipStream_t st, const LbvhInputs& in, LbvhOutputs& out) {
  const uint32_t nw = in.n_world;   // world triangles; n (below) = leaves of the hierarchy <= nw
  const int builder = in.builder == kBvhBuilderAuto ? kBvhBuilderSah : in.builder;
  out.depth = 0;
  out.sah = 0.0f;
  out.rounds = 0;
  out.nodes = nullptr;
  out.nodes8 = nullptr;
  out.n_nodes8 = 0;
  out.depth8 = 0;
  out.quads = nullptr;
  out.n_nodes = 0;
  out.n_leaves = 0;
  if (nw == 0) return hipSuccess;
  const uint32_t np_max = std::max<uint32_t>(next_pow2(nw), kSortTile);
  hipError_t e;
  BvhTri* tris_unsorted = nullptr;
  float4 *lo = nullptr, *hi = nullptr, *leaf_lo = nullptr, *leaf_hi = nullptr, *node_lo = nullptr, *node_hi = nullptr;
  uint64_t* keys = nullptr;
  uint32_t *vals = nullptr, *leaf_first = nullptr;
  uint8_t* role = nullptr;
  int2* children = nullptr;
  int *parent = nullptr, *node_depth = nullptr, *scalars = nullptr, *counts = nullptr, *new_id = nullptr;
  int *refs_a = nullptr, *refs_b = nullptr, *nearest = nullptr;
  uint32_t *sah_idx_a = nullptr, *sah_idx_b = nullptr, *sah_counts = nullptr, *sah_total_chunks = nullptr, *sah_chunk_left = nullptr;
  SahWideNode* sah_wide = nullptr;
  SahTask *sah_queue_a = nullptr, *sah_queue_b = nullptr;
  unsigned long long *flags = nullptr, *pos = nullptr, *slot = nullptr, *scan_tmp = nullptr, *scan_total = nullptr;
  float* sah = nullptr;
  BvhGrid* grid = nullptr;
  auto cleanup = [&]() {
    void* bufs[] = {tris_unsorted, lo, hi, leaf_lo, leaf_hi, leaf_first, role, node_lo, node_hi, keys, vals, children, parent, node_depth, scalars, sah,
                    grid, counts, new_id, refs_a, refs_b, nearest, flags, pos, slot, scan_tmp, scan_total, sah_idx_a, sah_idx_b, sah_counts, sah_queue_a, sah_queue_b, sah_total_chunks, sah_chunk_left, sah_wide};
    for (void* b : bufs)
      if (b) (void)hipFree(b);
  };
#define GLZ_TRY(x) do { e = (x); if (e != hipSuccess) { cleanup(); return e; } } while (0)
  GLZ_TRY(hipMalloc(&tris_unsorted, sizeof(BvhTri) * nw));
  GLZ_TRY(hipMalloc(&lo, sizeof(float4) * nw));
  GLZ_TRY(hipMalloc(&hi, sizeof(float4) * nw));
  GLZ_TRY(hipMalloc(&leaf_lo, sizeof(float4) * nw));
  GLZ_TRY(hipMalloc(&leaf_hi, sizeof(float4) * nw));
  GLZ_TRY(hipMalloc(&leaf_first, sizeof(uint32_t) * nw));
  GLZ_TRY(hipMalloc(&role, nw));
  GLZ_TRY(hipMalloc(&node_lo, sizeof(float4) * (2 * (size_t)nw)));
  GLZ_TRY(hipMalloc(&node_hi, sizeof(float4) * (2 * (size_t)nw)));
  GLZ_TRY(hipMalloc(&keys, sizeof(uint64_t) * np_max));
  GLZ_TRY(hipMalloc(&vals, sizeof(uint32_t) * np_max));
  GLZ_TRY(hipMalloc(&children, sizeof(int2) * nw));
  GLZ_TRY(hipMalloc(&parent, sizeof(int) * (2 * (size_t)nw)));
  GLZ_TRY(hipMalloc(&node_depth, sizeof(int) * nw));
  GLZ_TRY(hipMalloc(&scalars, sizeof(int) * 8));
  GLZ_TRY(hipMalloc(&sah, sizeof(float)));
  GLZ_TRY(hipMalloc(&grid, sizeof(BvhGrid)));
  GLZ_TRY(hipMalloc(&counts, sizeof(int) * nw));
  {
    const size_t tiles = ((size_t)nw + kScanTile - 1) / kScanTile;
    GLZ_TRY(hipMalloc(&flags, sizeof(unsigned long long) * nw));
    GLZ_TRY(hipMalloc(&pos, sizeof(unsigned long long) * nw));
    GLZ_TRY(hipMalloc(&slot, sizeof(unsigned long long) * nw));
    GLZ_TRY(hipMalloc(&scan_tmp, sizeof(unsigned long long) * (2 * tiles + 4096)));
    GLZ_TRY(hipMalloc(&scan_total, sizeof(unsigned long long)));
  }
  GLZ_TRY(hipMalloc(&new_id, sizeof(int) * nw));
  GLZ_TRY(hipMemsetAsync(sah, 0, sizeof(float), st));
  GLZ_TRY(hipMemsetAsync(role, 0, nw, st));
  {
    // ordered-int encodings of +inf / -inf, then depth counter
    const int init[8] = {0x7F800000, 0x7F800000, 0x7F800000, (int)0xFF800000 ^ 0x7FFFFFFF, (int)0xFF800000 ^ 0x7FFFFFFF,
                         (int)0xFF800000 ^ 0x7FFFFFFF, 0, 0};
    GLZ_TRY(hipMemcpyAsync(scalars, init, sizeof(init), hipMemcpyHostToDevice, st));
  }
  const dim3 blk(256), grdw((nw + 255) / 256);
  if (in.given_lo && in.given_hi) {
    hipLaunchKernelGGL(k_given_boxes, grdw, blk, 0, st, in.given_lo, in.given_hi, nw, tris_unsorted, lo, hi, scalars);
  } else {
    hipLaunchKernelGGL(k_world_tris, grdw, blk, 0, st, in.vertices, in.indices, in.instances, in.inst_base, in.n_instances, in.transforms,
                       in.materials, nw, tris_unsorted, lo, hi, scalars);
  }
  GLZ_TRY(hipGetLastError());
  // leaves: pairs of triangles where they qualify, single triangles otherwise
  if (in.pair_area_ratio > 0.0f && !in.given_lo) {
    for (uint32_t parity = 0; parity < 2; ++parity) {
      hipLaunchKernelGGL(k_pair_triangles, grdw, blk, 0, st, nw, parity, tris_unsorted, in.indices, in.instances, lo, hi, in.pair_area_ratio, in.emit_quads ? 1u : 0u, role);
      GLZ_TRY(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(k_leaf_flags, grdw, blk, 0, st, nw, role, flags);
  GLZ_TRY(hipGetLastError());
  GLZ_TRY(scan_exclusive(st, (int)nw, flags, pos, scan_tmp));
  hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, st, (int)nw, flags, pos, scan_total);
  GLZ_TRY(hipGetLastError());
  unsigned long long n_leaves = 0;
  GLZ_TRY(hipMemcpyAsync(&n_leaves, scan_total, sizeof(n_leaves), hipMemcpyDeviceToHost, st));
  GLZ_TRY(hipStreamSynchronize(st));
  const uint32_t n = (uint32_t)n_leaves;
  if (n == 0 || n > nw) { cleanup(); return hipErrorUnknown; }
  out.n_leaves = n;
  hipLaunchKernelGGL(k_leaf_boxes, grdw, blk, 0, st, nw, role, pos, lo, hi, leaf_first, leaf_lo, leaf_hi);
  GLZ_TRY(hipGetLastError());
  const uint32_t np = std::max<uint32_t>(next_pow2(n), kSortTile);
  const dim3 grd((n + 255) / 256);
  hipLaunchKernelGGL(k_morton, dim3((np + 255) / 256), blk, 0, st, leaf_lo, leaf_hi, scalars, n, np, keys, vals);
  GLZ_TRY(hipGetLastError());
  // bitonic network: stages k = 2..np; strides j = k/2..1
  hipLaunchKernelGGL(k_bitonic_lds, dim3(np / kSortTile), dim3(1024), 0, st, keys, vals, np, 2u, kSortTile, 0u);
  GLZ_TRY(hipGetLastError());
  for (uint32_t k = kSortTile * 2; k <= np; k <<= 1) {
    for (uint32_t j = k >> 1; j >= kSortTile; j >>= 1) {
      hipLaunchKernelGGL(k_bitonic_global, dim3((np / 2 + 255) / 256), blk, 0, st, keys, vals, np, k, j);
      GLZ_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_bitonic_lds, dim3(np / kSortTile), dim3(1024), 0, st, keys, vals, np, k, k, kSortTile / 2);
    GLZ_TRY(hipGetLastError());
  }
  // first slot of every leaf in bvh_tris (leaf order, one or two triangles each)
  hipLaunchKernelGGL(k_leaf_sizes, grd, blk, 0, st, n, vals, leaf_first, role, flags);
  GLZ_TRY(hipGetLastError());
  GLZ_TRY(scan_exclusive(st, (int)n, flags, slot, scan_tmp));
  out.quads = nullptr;
  if (in.emit_quads) {
    BvhQuad* q = nullptr;
    GLZ_TRY(hipMalloc(&q, sizeof(BvhQuad) * (size_t)n));
    out.quads = q;   // the caller's from here on, also when a later step fails
  }
  hipLaunchKernelGGL(k_gather_leaves, grd, blk, 0, st, vals, n, leaf_first, role, slot, tris_unsorted, leaf_lo, leaf_hi, out.tris, node_lo, node_hi, out.quads);
  GLZ_TRY(hipGetLastError());
  if (n >= 2) {
    if (builder == kBvhBuilderLbvh) {
      hipLaunchKernelGGL(k_hierarchy, grd, blk, 0, st, keys, (int)n, children, parent);
      GLZ_TRY(hipGetLastError());
    } else if (builder == kBvhBuilderSah) {
      // binned SAH, one launch per level (k_sah_level); the leaf boxes are node_lo / node_hi slots (n-1)+j
      const size_t qcap = (size_t)n / 2 + 2;
      GLZ_TRY(hipMalloc(&sah_idx_a, sizeof(uint32_t) * n));
      GLZ_TRY(hipMalloc(&sah_idx_b, sizeof(uint32_t) * n));
      GLZ_TRY(hipMalloc(&sah_queue_a, sizeof(SahTask) * qcap));
      GLZ_TRY(hipMalloc(&sah_queue_b, sizeof(SahTask) * qcap));
      GLZ_TRY(hipMalloc(&sah_counts, sizeof(uint32_t) * 2));
      GLZ_TRY(hipMalloc(&sah_total_chunks, sizeof(uint32_t)));
      GLZ_TRY(hipMalloc(&sah_wide, sizeof(SahWideNode) * ((size_t)n / kSahWideMean + 2)));
      GLZ_TRY(hipMalloc(&sah_chunk_left, sizeof(uint32_t) * ((size_t)n / kSahWideMean + (size_t)n / kSahChunk + 4)));
      hipLaunchKernelGGL(k_sah_init, grd, blk, 0, st, n, sah_idx_a, sah_queue_a, sah_counts, parent);
      GLZ_TRY(hipGetLastError());
      uint32_t active = 1;
      uint32_t *idx_in = sah_idx_a, *idx_out = sah_idx_b;
      SahTask *q_in = sah_queue_a, *q_out = sah_queue_b;
      for (int level = 0, in = 0; active > 0; ++level, in ^= 1) {
        if (level > kSahMaxSplitLevels + 64) { cleanup(); return hipErrorUnknown; }   // cannot happen: halving ends after 32 levels
        // by the mean range of the level: several blocks per node while the ranges are long, then one block per node --
        // many threads for a long range (it is one block's loop), one wave for the wide bottom levels (its barriers cost nothing)
        // A tree this deep means input that defeats the binning level after level (a geometric progression of scales); the
        // remaining ranges are halved as they stand so that the depth stays bounded.
        const int force_halve = level >= kSahMaxSplitLevels ? 1 : 0;
        const uint32_t mean = n / active;
        if (mean >= kSahWideMean && !force_halve) {
          const uint32_t max_chunks = active + n / kSahChunk + 1;
          const dim3 gc(max_chunks), gn((active + 63) / 64);
          hipLaunchKernelGGL(k_wide_plan, dim3(1), dim3(1024), 0, st, q_in, sah_counts + in, sah_wide, sah_total_chunks);
          hipLaunchKernelGGL(k_wide_bounds, gc, dim3(kSahWideBlock), 0, st, q_in, sah_counts + in, sah_total_chunks, sah_wide, idx_in, node_lo + (n - 1),
                             node_hi + (n - 1));
          hipLaunchKernelGGL(k_wide_bin, gc, dim3(kSahWideBlock), 0, st, q_in, sah_counts + in, sah_total_chunks, sah_wide, idx_in, node_lo + (n - 1),
                             node_hi + (n - 1));
          hipLaunchKernelGGL(k_wide_pick, gn, dim3(64), 0, st, q_in, sah_counts + in, sah_wide);
          hipLaunchKernelGGL(k_wide_count, gc, dim3(kSahWideBlock), 0, st, q_in, sah_counts + in, sah_total_chunks, sah_wide, idx_in, node_lo + (n - 1),
                             node_hi + (n - 1), sah_chunk_left);
          hipLaunchKernelGGL(k_wide_scatter, gc, dim3(kSahWideBlock), 0, st, q_in, sah_counts + in, sah_total_chunks, sah_wide, sah_chunk_left, idx_in, idx_out,
                             node_lo + (n - 1), node_hi + (n - 1));
          hipLaunchKernelGGL(k_wide_children, gn, dim3(64), 0, st, q_in, sah_counts + in, sah_wide, idx_out, (int)n, children, parent, q_out,
                             sah_counts + (in ^ 1));
        } else {
#define GLZ_SAH_LAUNCH(B) hipLaunchKernelGGL(k_sah_level<B>, dim3(active), dim3(B), 0, st, q_in, sah_counts + in, idx_in, idx_out, q_out, \
                                             sah_counts + (in ^ 1), (int)n, node_lo + (n - 1), node_hi + (n - 1), children, parent, force_halve)
          if (mean >= 4096) GLZ_SAH_LAUNCH(1024);
          else if (mean >= 128) GLZ_SAH_LAUNCH(256);
          else GLZ_SAH_LAUNCH(64);
#undef GLZ_SAH_LAUNCH
        }
        GLZ_TRY(hipGetLastError());
        GLZ_TRY(hipMemcpyAsync(&active, sah_counts + (in ^ 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        GLZ_TRY(hipMemsetAsync(sah_counts + in, 0, sizeof(uint32_t), st));   // this level's input counter is the output counter of the level after next
        GLZ_TRY(hipStreamSynchronize(st));
        std::swap(idx_in, idx_out);
        std::swap(q_in, q_out);
        ++out.rounds;
      }
    } else if (builder == kBvhBuilderSahHost) {
      // the same builder on the host cores (bvh_sah.cpp): the reference the GPU builder is tested against
      std::vector<float4> h_lo(n), h_hi(n);
      std::vector<int2> h_children(n);
      std::vector<int> h_parent(2 * (size_t)n);
      GLZ_TRY(hipMemcpyAsync(h_lo.data(), node_lo + (n - 1), sizeof(float4) * n, hipMemcpyDeviceToHost, st));
      GLZ_TRY(hipMemcpyAsync(h_hi.data(), node_hi + (n - 1), sizeof(float4) * n, hipMemcpyDeviceToHost, st));
      GLZ_TRY(hipStreamSynchronize(st));
      build_sah_host(n, h_lo.data(), h_hi.data(), h_children.data(), h_parent.data());
      GLZ_TRY(hipMemcpyAsync(children, h_children.data(), sizeof(int2) * (n - 1), hipMemcpyHostToDevice, st));
      GLZ_TRY(hipMemcpyAsync(parent, h_parent.data(), sizeof(int) * (2 * (size_t)n - 1), hipMemcpyHostToDevice, st));
      GLZ_TRY(hipStreamSynchronize(st));
    } else {
      GLZ_TRY(hipMalloc(&refs_a, sizeof(int) * n));
      GLZ_TRY(hipMalloc(&refs_b, sizeof(int) * n));
      GLZ_TRY(hipMalloc(&nearest, sizeof(int) * n));
      hipLaunchKernelGGL(k_ploc_init, grd, blk, 0, st, (int)n, refs_a, parent);
      GLZ_TRY(hipGetLastError());
      int m = (int)n, next_free = (int)n - 2;
      out.rounds = 0;
      while (m > 1) {
        const dim3 gm((m + 255) / 256);
        hipLaunchKernelGGL(k_ploc_nearest, gm, blk, 0, st, m, (int)n, refs_a, node_lo, node_hi, nearest);
        GLZ_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_ploc_flags, gm, blk, 0, st, m, nearest, flags);
        GLZ_TRY(hipGetLastError());
        GLZ_TRY(scan_exclusive(st, m, flags, pos, scan_tmp));
        hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, st, m, flags, pos, scan_total);
        GLZ_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_ploc_merge, gm, blk, 0, st, m, (int)n, next_free, refs_a, nearest, flags, pos, refs_b, children, parent, node_lo,
                           node_hi);
        GLZ_TRY(hipGetLastError());
        unsigned long long total = 0;
        GLZ_TRY(hipMemcpyAsync(&total, scan_total, sizeof(total), hipMemcpyDeviceToHost, st));
        GLZ_TRY(hipStreamSynchronize(st));
        const int survivors = (int)(total & 0xFFFFFFFFull), merges = (int)(total >> 32);
        if (merges <= 0 || survivors != m - merges) { cleanup(); return hipErrorUnknown; }   // cannot happen: the closest pair is always mutual
        next_free -= merges;
        m = survivors;
        std::swap(refs_a, refs_b);
        ++out.rounds;
      }
    }
    // bottom-up, one launch per level: boxes (LBVH; PLOC made them while merging) and subtree sizes
    hipLaunchKernelGGL(k_node_depth, dim3((2 * n + 255) / 256), blk, 0, st, (int)n, parent, node_depth, scalars + 7, scalars + 6);
    GLZ_TRY(hipGetLastError());
    int inner_depth = 0;
    GLZ_TRY(hipMemcpyAsync(&inner_depth, scalars + 7, sizeof(int), hipMemcpyDeviceToHost, st));
    GLZ_TRY(hipStreamSynchronize(st));
    for (int level = inner_depth; level >= 0; --level) {
      if (builder != kBvhBuilderPloc)
        hipLaunchKernelGGL(k_level_up<true>, grd, blk, 0, st, (int)n, level, node_depth, children, node_lo, node_hi, counts);
      else
        hipLaunchKernelGGL(k_level_up<false>, grd, blk, 0, st, (int)n, level, node_depth, children, node_lo, node_hi, counts);
      GLZ_TRY(hipGetLastError());
    }
    // depth-first layout of the finished hierarchy
    hipLaunchKernelGGL(k_dfs_ids, grd, blk, 0, st, (int)n, children, parent, counts, new_id);
    GLZ_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(64), 0, st, node_lo, node_hi, grid);
    GLZ_TRY(hipGetLastError());
    // 4-wide collapse: find the heads of the BVH4 nodes top down, number them in depth-first order, then emit them
    int* head = counts;   // k_dfs_ids was the last reader of the subtree sizes
    GLZ_TRY(hipMemsetAsync(head, 0, sizeof(int) * n, st));
    GLZ_TRY(hipMemsetAsync(scalars + 7, 0, sizeof(int), st));
    for (int level = 0; level <= inner_depth; ++level) {
      hipLaunchKernelGGL(k_mark_heads<4>, grd, blk, 0, st, (int)n, level, node_depth, children, node_lo, node_hi, head, scalars + 7);
      GLZ_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_head_flags, grd, blk, 0, st, (int)n, head, new_id, flags);
    GLZ_TRY(hipGetLastError());
    GLZ_TRY(scan_exclusive(st, (int)n - 1, flags, pos, scan_tmp));
    hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, st, (int)n - 1, flags, pos, scan_total);
    GLZ_TRY(hipGetLastError());
    unsigned long long n4 = 0;
    int depth4_keep = 0;
    GLZ_TRY(hipMemcpyAsync(&n4, scan_total, sizeof(n4), hipMemcpyDeviceToHost, st));
    GLZ_TRY(hipMemcpyAsync(&depth4_keep, scalars + 7, sizeof(int), hipMemcpyDeviceToHost, st));
    GLZ_TRY(hipStreamSynchronize(st));
    out.n_nodes = (uint32_t)n4;
    GLZ_TRY(hipMalloc(&out.nodes, sizeof(BvhNode4) * (size_t)n4));
    hipLaunchKernelGGL(k_emit_nodes4, grd, blk, 0, st, (int)n, children, node_lo, node_hi, grid, new_id, flags, pos, slot, in.emit_quads ? 1u : 0u, out.nodes, sah);
    GLZ_TRY(hipGetLastError());
    if (in.emit_wide8 && in.emit_quads) {
      // the 8-wide collapse of the same binary hierarchy: heads, depth-first numbers, nodes (head / flags / pos are free again)
      GLZ_TRY(hipMemsetAsync(head, 0, sizeof(int) * n, st));
      GLZ_TRY(hipMemsetAsync(scalars + 7, 0, sizeof(int), st));
      for (int level = 0; level <= inner_depth; ++level) {
        hipLaunchKernelGGL(k_mark_heads<8>, grd, blk, 0, st, (int)n, level, node_depth, children, node_lo, node_hi, head, scalars + 7);
        GLZ_TRY(hipGetLastError());
      }
      hipLaunchKernelGGL(k_head_flags, grd, blk, 0, st, (int)n, head, new_id, flags);
      GLZ_TRY(hipGetLastError());
      GLZ_TRY(scan_exclusive(st, (int)n - 1, flags, pos, scan_tmp));
      hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, st, (int)n - 1, flags, pos, scan_total);
      GLZ_TRY(hipGetLastError());
      unsigned long long n8 = 0;
      int depth8 = 0;
      GLZ_TRY(hipMemcpyAsync(&n8, scan_total, sizeof(n8), hipMemcpyDeviceToHost, st));
      GLZ_TRY(hipMemcpyAsync(&depth8, scalars + 7, sizeof(int), hipMemcpyDeviceToHost, st));
      GLZ_TRY(hipStreamSynchronize(st));
      out.n_nodes8 = (uint32_t)n8;
      out.depth8 = (uint32_t)depth8;
      GLZ_TRY(hipMalloc(&out.nodes8, sizeof(BvhNode8) * (size_t)n8));
      hipLaunchKernelGGL(k_emit_nodes8, grd, blk, 0, st, (int)n, children, node_lo, node_hi, grid, new_id, flags, pos, out.nodes8);
      GLZ_TRY(hipGetLastError());
      GLZ_TRY(hipMemcpyAsync(scalars + 7, &depth4_keep, sizeof(int), hipMemcpyHostToDevice, st));   // host_scalars[7] below is the 4-wide depth
    }
  }
  int host_scalars[8];
  float host_sah = 0.0f;
  float4 root_lo, root_hi;
  GLZ_TRY(hipMemcpyAsync(host_scalars, scalars, sizeof(host_scalars), hipMemcpyDeviceToHost, st));
  GLZ_TRY(hipMemcpyAsync(&host_sah, sah, sizeof(float), hipMemcpyDeviceToHost, st));
  GLZ_TRY(hipMemcpyAsync(&root_lo, node_lo, sizeof(float4), hipMemcpyDeviceToHost, st));
  GLZ_TRY(hipMemcpyAsync(&root_hi, node_hi, sizeof(float4), hipMemcpyDeviceToHost, st));
  GLZ_TRY(hipStreamSynchronize(st));
  if (n == 1) {
    // single triangle: one node whose first child is leaf 0 with a box spanning the whole grid
    hipLaunchKernelGGL(k_grid_params, dim3(1), dim3(64), 0, st, node_lo, node_hi, grid);   // slot (n-1)+0 = 0 is the leaf box
    GLZ_TRY(hipGetLastError());
    BvhNode4 nd{};
    for (int k = 0; k < 4; ++k) {
      nd.w[3 * k] = nd.w[3 * k + 1] = nd.w[3 * k + 2] = kBvhGridMax;
      nd.w[12 + k] = (uint32_t)kBvhEmptyChild;
    }
    nd.w[0] = nd.w[1] = nd.w[2] = 0u | (kBvhGridMax << 16);   // lo 0, hi the grid's top on every axis
    nd.w[12] = ~0u;   // leaf 0
    GLZ_TRY(hipMalloc(&out.nodes, sizeof(BvhNode4)));
    out.n_nodes = 1;
    GLZ_TRY(hipMemcpyAsync(out.nodes, &nd, sizeof(nd), hipMemcpyHostToDevice, st));
    GLZ_TRY(hipStreamSynchronize(st));
    out.depth = 1;
  } else {
    out.depth = (uint32_t)host_scalars[7];   // levels of 4-wide nodes above the deepest leaf
    const float dx = root_hi.x - root_lo.x, dy = root_hi.y - root_lo.y, dz = root_hi.z - root_lo.z;
    const float ra = 2.0f * (dx * dy + dy * dz + dz * dx);
    out.sah = ra > 0.0f ? host_sah / ra : 0.0f;
  }
  GLZ_TRY(hipMemcpy(&out.grid, grid, sizeof(BvhGrid), hipMemcpyDeviceToHost));
  out.bounds_lo[0] = root_lo.x; out.bounds_lo[1] = root_lo.y; out.bounds_lo[2] = root_lo.z;
  out.bounds_hi[0] = root_hi.x; out.bounds_hi[1] = root_hi.y; out.bounds_hi[2] = root_hi.z;
  cleanup();
#undef GLZ_TRY
  return hipSuccess;
}

}  // namespace glz
