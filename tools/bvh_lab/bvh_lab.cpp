// bvh_lab: an OFFLINE (CPU) bench for hierarchy quality.  Not part of the product and not test infrastructure -- a tool for deciding
// what is worth building into kernels_build.hip: it builds hierarchies over the bench scene's leaves (quads: two triangles that share
// an edge, as the product pairs them) in several ways, collapses them to 4- or 8-wide nodes, and counts node and leaf visits of the
// traversal rule the tracers use (children entered nearest first, pruned by the closest hit so far; any-hit rays stop at the first hit)
// over a ray population shaped like a path tracer's: camera rays, cosine-distributed bounces, shadow rays towards the sun and the sky.
//
//   g++ -O2 -std=c++17 -fopenmp tools/bvh_lab/bvh_lab.cpp -o /tmp/bvh_lab && /tmp/bvh_lab /tmp/atrium.bin [options]
//   options: builder=binned|sweep|sbvh  bins=16  alpha=1e-5  width=4|8  collapse=area|sah  order=full|nearest  leafmax=1  presplit=0|N  rays=WxH  bounces=8
//
// Boxes are kept in floats (no 15-bit grid): the numbers are for comparing variants with each other, not with the GPU counters.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <random>
#include <string>
#include <vector>

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 norm(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
static inline float comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
static inline void setc(V3& v, int a, float f) { (a == 0 ? v.x : (a == 1 ? v.y : v.z)) = f; }

struct Box {
  V3 lo{1e30f, 1e30f, 1e30f}, hi{-1e30f, -1e30f, -1e30f};
  void grow(V3 p) { lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)}; hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)}; }
  void grow(const Box& b) { if (b.hi.x < b.lo.x) return; grow(b.lo); grow(b.hi); }
  bool empty() const { return hi.x < lo.x; }
  float area() const { if (empty()) return 0.0f; const V3 d = hi - lo; return 2.0f * (d.x * d.y + d.y * d.z + d.z * d.x); }
  V3 centre() const { return (lo + hi) * 0.5f; }
};
static Box intersect(const Box& a, const Box& b) {
  Box r;
  r.lo = {std::max(a.lo.x, b.lo.x), std::max(a.lo.y, b.lo.y), std::max(a.lo.z, b.lo.z)};
  r.hi = {std::min(a.hi.x, b.hi.x), std::min(a.hi.y, b.hi.y), std::min(a.hi.z, b.hi.z)};
  return r;
}

struct Prim { V3 v[4]; int nv; };   // a quad (two triangles v0 v1 v2, v0 v2 v3) or a triangle
struct Ref { int prim; Box box; };

static std::vector<Prim> prims;

// the box of the part of a primitive inside [lo, hi] along `axis` (polygon clipping of each triangle; Stich et al. 2009)
static Box clip_prim(const Prim& p, int axis, float lo, float hi) {
  Box out;
  for (int t = 0; t < (p.nv == 4 ? 2 : 1); ++t) {
    V3 poly[8]; int n = 3;
    poly[0] = p.v[0]; poly[1] = p.v[t == 0 ? 1 : 2]; poly[2] = p.v[t == 0 ? 2 : 3];
    for (int side = 0; side < 2; ++side) {
      V3 q[8]; int m = 0;
      const float plane = side == 0 ? lo : hi, sgn = side == 0 ? 1.0f : -1.0f;
      for (int i = 0; i < n; ++i) {
        const V3 a = poly[i], b = poly[(i + 1) % n];
        const float da = sgn * (comp(a, axis) - plane), db = sgn * (comp(b, axis) - plane);
        if (da >= 0.0f) q[m++] = a;
        if ((da > 0.0f && db < 0.0f) || (da < 0.0f && db > 0.0f)) {
          const float s = da / (da - db);
          V3 c = a + (b - a) * s;
          setc(c, axis, plane);
          q[m++] = c;
        }
      }
      n = m;
      for (int i = 0; i < n; ++i) poly[i] = q[i];
      if (n == 0) break;
    }
    for (int i = 0; i < n; ++i) out.grow(poly[i]);
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// binary hierarchy
// ---------------------------------------------------------------------------------------------
struct BNode { Box box; int left = -1, right = -1; int first = 0, count = 0; };   // count > 0: leaf over refs [first, first + count)
static std::vector<BNode> bnodes;
static std::vector<Ref> leaf_refs;   // refs in leaf order

struct Opt {
  std::string builder = "binned", collapse = "area", order = "full";
  int bins = 16, width = 4, leafmax = 1, presplit = 0, rw = 480, rh = 270, bounces = 8, sbins = 32;
  float alpha = 1e-5f, ct = 1.2f, ci = 1.0f;
} opt;
static float root_area = 1.0f;
static long long n_spatial = 0, n_object = 0;

static int build(std::vector<Ref>& refs, int depth) {
  const int id = (int)bnodes.size();
  bnodes.emplace_back();
  Box nb, cb;
  for (auto& r : refs) { nb.grow(r.box); cb.grow(r.box.centre()); }
  bnodes[id].box = nb;
  auto make_leaf = [&]() {
    bnodes[id].first = (int)leaf_refs.size();
    bnodes[id].count = (int)refs.size();
    for (auto& r : refs) leaf_refs.push_back(r);
    return id;
  };
  if ((int)refs.size() <= opt.leafmax || depth > 60) return make_leaf();
  // ---- object split
  float best = 1e30f; int best_axis = -1; float best_pos = 0.0f; int best_k = 0;
  const int n = (int)refs.size();
  if (opt.builder == "sweep") {
    std::vector<int> order(n);
    std::vector<float> right(n + 1);
    for (int a = 0; a < 3; ++a) {
      for (int i = 0; i < n; ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](int i, int j) { return comp(refs[i].box.centre(), a) < comp(refs[j].box.centre(), a); });
      Box b;
      for (int i = n - 1; i > 0; --i) { b.grow(refs[order[i]].box); right[i] = b.area(); }
      b = Box();
      for (int i = 0; i < n - 1; ++i) {
        b.grow(refs[order[i]].box);
        const float c = b.area() * (i + 1) + right[i + 1] * (n - 1 - i);
        if (c < best) { best = c; best_axis = a; best_k = i + 1; }
      }
    }
  } else {
    const int B = opt.bins;
    for (int a = 0; a < 3; ++a) {
      const float lo = comp(cb.lo, a), ext = comp(cb.hi, a) - lo;
      if (!(ext > 0.0f)) continue;
      std::vector<Box> bb(B); std::vector<int> cn(B, 0);
      const float scale = B / ext;
      for (auto& r : refs) { int k = std::min(B - 1, std::max(0, (int)((comp(r.box.centre(), a) - lo) * scale))); bb[k].grow(r.box); cn[k]++; }
      std::vector<float> ra(B); std::vector<int> rc(B);
      Box b; int c = 0;
      for (int k = B - 1; k > 0; --k) { b.grow(bb[k]); c += cn[k]; ra[k] = b.area(); rc[k] = c; }
      b = Box(); c = 0;
      for (int k = 0; k < B - 1; ++k) {
        b.grow(bb[k]); c += cn[k];
        if (c == 0 || rc[k + 1] == 0) continue;
        const float cost = b.area() * c + ra[k + 1] * rc[k + 1];
        if (cost < best) { best = cost; best_axis = a; best_k = k; best_pos = lo + (k + 1) / scale; }
      }
    }
  }
  // ---- spatial split (sbvh): only when the object split's children overlap enough
  bool spatial = false; int s_axis = -1; float s_pos = 0.0f;
  if (opt.builder == "sbvh" && best_axis >= 0) {
    // overlap of the object split's children
    Box l, r;
    {
      const int a = best_axis; const float lo = comp(cb.lo, a), scale = opt.bins / (comp(cb.hi, a) - lo);
      for (auto& rf : refs) { int k = std::min(opt.bins - 1, std::max(0, (int)((comp(rf.box.centre(), a) - lo) * scale))); (k <= best_k ? l : r).grow(rf.box); }
    }
    const Box ov = intersect(l, r);
    const bool overlap = ov.lo.x <= ov.hi.x && ov.lo.y <= ov.hi.y && ov.lo.z <= ov.hi.z;
    if (overlap && ov.area() / root_area > opt.alpha) {
      const int B = opt.sbins;
      for (int a = 0; a < 3; ++a) {
        const float lo = comp(nb.lo, a), ext = comp(nb.hi, a) - lo;
        if (!(ext > 0.0f)) continue;
        const float w = ext / B;
        std::vector<Box> bb(B); std::vector<int> en(B, 0), ex(B, 0);
        for (auto& rf : refs) {
          int k0 = std::min(B - 1, std::max(0, (int)((comp(rf.box.lo, a) - lo) / w))), k1 = std::min(B - 1, std::max(0, (int)((comp(rf.box.hi, a) - lo) / w)));
          en[k0]++; ex[k1]++;
          for (int k = k0; k <= k1; ++k) {
            Box c = k0 == k1 ? rf.box : intersect(clip_prim(prims[rf.prim], a, lo + k * w, lo + (k + 1) * w), rf.box);
            if (!c.empty()) bb[k].grow(c);
          }
        }
        std::vector<float> ra(B); std::vector<int> rc(B);
        Box b; int c = 0;
        for (int k = B - 1; k > 0; --k) { b.grow(bb[k]); c += ex[k]; ra[k] = b.area(); rc[k] = c; }
        b = Box(); c = 0;
        for (int k = 0; k < B - 1; ++k) {
          b.grow(bb[k]); c += en[k];
          if (c == 0 || rc[k + 1] == 0) continue;
          const float cost = b.area() * c + ra[k + 1] * rc[k + 1];
          if (cost < best) { best = cost; spatial = true; s_axis = a; s_pos = lo + (k + 1) * w; }
        }
      }
    }
  }
  if (best_axis < 0 && !spatial) {
    if (n <= 4) return make_leaf();
    best_axis = 0; best_k = n / 2;   // halve
    std::vector<Ref> L(refs.begin(), refs.begin() + n / 2), R(refs.begin() + n / 2, refs.end());
    refs.clear(); refs.shrink_to_fit();
    const int l = build(L, depth + 1), r = build(R, depth + 1);
    bnodes[id].left = l; bnodes[id].right = r;
    return id;
  }
  // leaf if cheaper (only with leafmax > 1)
  if (n <= opt.leafmax) return make_leaf();
  std::vector<Ref> L, R;
  if (spatial) {
    n_spatial++;
    for (auto& rf : refs) {
      const float lo = comp(rf.box.lo, s_axis), hi = comp(rf.box.hi, s_axis);
      if (hi <= s_pos) L.push_back(rf);
      else if (lo >= s_pos) R.push_back(rf);
      else {
        Ref a = rf, b = rf;
        a.box = intersect(clip_prim(prims[rf.prim], s_axis, -1e30f, s_pos), rf.box);
        b.box = intersect(clip_prim(prims[rf.prim], s_axis, s_pos, 1e30f), rf.box);
        if (!a.box.empty()) L.push_back(a);
        if (!b.box.empty()) R.push_back(b);
      }
    }
    if (L.empty() || R.empty() || (int)L.size() == n && (int)R.size() == n) { L.clear(); R.clear(); spatial = false; }
  }
  if (!spatial) {
    n_object++;
    if (opt.builder == "sweep") {
      std::vector<int> order(n);
      for (int i = 0; i < n; ++i) order[i] = i;
      const int a = best_axis;
      std::sort(order.begin(), order.end(), [&](int i, int j) { return comp(refs[i].box.centre(), a) < comp(refs[j].box.centre(), a); });
      for (int i = 0; i < n; ++i) (i < best_k ? L : R).push_back(refs[order[i]]);
    } else {
      const int a = best_axis; const float lo = comp(cb.lo, a), scale = opt.bins / (comp(cb.hi, a) - lo);
      for (auto& rf : refs) { int k = std::min(opt.bins - 1, std::max(0, (int)((comp(rf.box.centre(), a) - lo) * scale))); (k <= best_k ? L : R).push_back(rf); }
    }
  }
  (void)best_pos;
  refs.clear(); refs.shrink_to_fit();
  const int l = build(L, depth + 1), r = build(R, depth + 1);
  bnodes[id].left = l; bnodes[id].right = r;
  return id;
}

// ---------------------------------------------------------------------------------------------
// wide collapse
// ---------------------------------------------------------------------------------------------
struct WNode { int n = 0; Box box[8]; int link[8]; };   // link >= 0: wide node, < 0: ~leaf (index of a binary leaf node)
static std::vector<WNode> wnodes;

// SAH-optimal collapse (Ylitie, Karras, Laine 2017): cost[n][i] = cheapest way to represent the subtree of n as a forest of at most i + 1 roots
static std::vector<std::array<float, 8>> dp_cost;
static std::vector<std::array<int8_t, 8>> dp_split;   // how many roots go to the left subtree (0: the node itself is a root)
static void dp_prepare(int W) {
  dp_cost.assign(bnodes.size(), {});
  dp_split.assign(bnodes.size(), {});
  // post-order
  std::vector<int> order; order.reserve(bnodes.size());
  std::vector<int> st{0};
  while (!st.empty()) { int n = st.back(); st.pop_back(); order.push_back(n); if (bnodes[n].count == 0) { st.push_back(bnodes[n].left); st.push_back(bnodes[n].right); } }
  for (int oi = (int)order.size() - 1; oi >= 0; --oi) {
    const int n = order[oi];
    const BNode& b = bnodes[n];
    const float A = b.box.area();
    if (b.count > 0) { for (int i = 0; i < W; ++i) { dp_cost[n][i] = A * opt.ci * b.count; dp_split[n][i] = 0; } continue; }
    // as one root (i = 0): a wide node whose children are the best forest of <= W roots below
    auto forest = [&](int k, int8_t& split) {   // best distribution of k roots (k >= 2) over the two children
      float best = 1e30f;
      for (int l = 1; l < k; ++l) { const float c = dp_cost[b.left][l - 1] + dp_cost[b.right][k - l - 1]; if (c < best) { best = c; split = (int8_t)l; } }
      return best;
    };
    int8_t s;
    dp_cost[n][0] = A * opt.ct + forest(W, s);
    dp_split[n][0] = 0;
    (void)s;
    for (int i = 1; i < W; ++i) {   // at most i + 1 roots
      int8_t sp = 0;
      float c = forest(i + 1, sp);
      if (dp_cost[n][i - 1] <= c) { c = dp_cost[n][i - 1]; sp = dp_split[n][i - 1]; }   // fewer roots is allowed
      dp_cost[n][i] = c; dp_split[n][i] = sp;
    }
  }
}
static void dp_roots(int n, int k, std::vector<int>& out) {   // the roots of the best forest of <= k roots under n, left to right
  const BNode& b = bnodes[n];
  if (b.count > 0) { out.push_back(n); return; }
  // which i gives the cost? walk down to the first i' <= k-1 with the same split decision
  int i = k - 1;
  while (i > 0 && dp_cost[n][i - 1] <= dp_cost[n][i] && dp_split[n][i] == dp_split[n][i - 1] && false) --i;
  // recompute the decision for exactly k roots
  if (k == 1) { out.push_back(n); return; }
  float best = 1e30f; int l_best = 1;
  for (int l = 1; l < k; ++l) { const float c = dp_cost[b.left][l - 1] + dp_cost[b.right][k - l - 1]; if (c < best) { best = c; l_best = l; } }
  if (dp_cost[n][0] <= best) { out.push_back(n); return; }
  dp_roots(b.left, l_best, out);
  dp_roots(b.right, k - l_best, out);
}

static int collapse(int bn) {   // bn: an inner binary node that heads a wide node
  const int id = (int)wnodes.size();
  wnodes.emplace_back();
  std::vector<int> kids;
  const int W = opt.width;
  if (opt.collapse == "sah") {
    const BNode& b = bnodes[bn];
    float best = 1e30f; int l_best = 1;
    for (int l = 1; l < W; ++l) { const float c = dp_cost[b.left][l - 1] + dp_cost[b.right][W - l - 1]; if (c < best) { best = c; l_best = l; } }
    dp_roots(b.left, l_best, kids);
    dp_roots(b.right, W - l_best, kids);
  } else {
    kids = {bnodes[bn].left, bnodes[bn].right};
    while ((int)kids.size() < W) {
      int j = -1; float best = -1.0f;
      for (int i = 0; i < (int)kids.size(); ++i)
        if (bnodes[kids[i]].count == 0 && bnodes[kids[i]].box.area() > best) { best = bnodes[kids[i]].box.area(); j = i; }
      if (j < 0) break;
      const int k = kids[j];
      kids[j] = bnodes[k].left;
      kids.insert(kids.begin() + j + 1, bnodes[k].right);
    }
  }
  wnodes[id].n = (int)kids.size();
  for (int i = 0; i < (int)kids.size(); ++i) {
    wnodes[id].box[i] = bnodes[kids[i]].box;
    const int link = bnodes[kids[i]].count > 0 ? ~kids[i] : collapse(kids[i]);
    wnodes[id].link[i] = link;
  }
  return id;
}

// ---------------------------------------------------------------------------------------------
// traversal with counters
// ---------------------------------------------------------------------------------------------
struct Count { double nodes = 0, leaves = 0, tris = 0, rays = 0, hits = 0, empty_visits = 0; };
static bool ray_tri(V3 o, V3 d, V3 a, V3 b, V3 c, float tmin, float tmax, float& t) {
  const V3 e1 = b - a, e2 = c - a, p = cross(d, e2);
  const float det = dot(e1, p);
  if (det == 0.0f) return false;
  const float inv = 1.0f / det;
  const V3 s = o - a;
  const float u = dot(s, p) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  const V3 q = cross(s, e1);
  const float v = dot(d, q) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot(e2, q) * inv;
  return t > tmin && t < tmax;
}
static bool trace(V3 o, V3 d, float tmin, float tmax, bool any, Count& c, float& t_out, int& prim_out) {
  const V3 inv{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
  int stack[256]; int sp = 0; int cur = 0;
  float best = tmax; bool found = false;
  c.rays += 1;
  for (;;) {
    if (cur >= 0) {
      const WNode& w = wnodes[cur];
      c.nodes += 1;
      std::pair<float, int> key[8]; int nk = 0;
      for (int k = 0; k < w.n; ++k) {
        const Box& b = w.box[k];
        float t0 = tmin, t1 = best;
        const float ax = (b.lo.x - o.x) * inv.x, bx = (b.hi.x - o.x) * inv.x; t0 = std::max(t0, std::min(ax, bx)); t1 = std::min(t1, std::max(ax, bx));
        const float ay = (b.lo.y - o.y) * inv.y, by = (b.hi.y - o.y) * inv.y; t0 = std::max(t0, std::min(ay, by)); t1 = std::min(t1, std::max(ay, by));
        const float az = (b.lo.z - o.z) * inv.z, bz = (b.hi.z - o.z) * inv.z; t0 = std::max(t0, std::min(az, bz)); t1 = std::min(t1, std::max(az, bz));
        if (t0 <= t1) key[nk++] = {t0, k};
      }
      if (nk == 0) c.empty_visits += 1;
      if (opt.order == "full") std::sort(key, key + nk);
      else if (nk > 1) {   // "nearest": the nearest child first, the others in slot order
        int m = 0;
        for (int k = 1; k < nk; ++k) if (key[k] < key[m]) m = k;
        const auto first = key[m];
        for (int k = m; k > 0; --k) key[k] = key[k - 1];
        key[0] = first;
      }
      if (nk > 0) {
        for (int k = nk - 1; k >= 1; --k) stack[sp++] = w.link[key[k].second];
        cur = w.link[key[0].second];
        continue;
      }
    } else {
      const BNode& leaf = bnodes[~cur];
      bool stop = false;
      for (int i = leaf.first; i < leaf.first + leaf.count; ++i) {
        c.leaves += 1;
        const Prim& p = prims[leaf_refs[i].prim];
        for (int t = 0; t < (p.nv == 4 ? 2 : 1); ++t) {
          c.tris += 1;
          float tt;
          if (ray_tri(o, d, p.v[0], p.v[t == 0 ? 1 : 2], p.v[t == 0 ? 2 : 3], tmin, best, tt)) { best = tt; found = true; prim_out = leaf_refs[i].prim; if (any) stop = true; }
        }
      }
      if (stop) break;
    }
    if (!sp) break;
    cur = stack[--sp];
  }
  t_out = best;
  if (found) c.hits += 1;
  return found;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: bvh_lab scene.bin [key=value ...]\n"); return 2; }
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i]; const size_t eq = a.find('=');
    if (eq == std::string::npos) continue;
    const std::string k = a.substr(0, eq), v = a.substr(eq + 1);
    if (k == "builder") opt.builder = v; else if (k == "order") opt.order = v; else if (k == "collapse") opt.collapse = v; else if (k == "bins") opt.bins = std::atoi(v.c_str());
    else if (k == "sbins") opt.sbins = std::atoi(v.c_str()); else if (k == "width") opt.width = std::atoi(v.c_str()); else if (k == "leafmax") opt.leafmax = std::atoi(v.c_str());
    else if (k == "presplit") opt.presplit = std::atoi(v.c_str()); else if (k == "alpha") opt.alpha = (float)std::atof(v.c_str()); else if (k == "bounces") opt.bounces = std::atoi(v.c_str());
    else if (k == "ct") opt.ct = (float)std::atof(v.c_str());
    else if (k == "rays") std::sscanf(v.c_str(), "%dx%d", &opt.rw, &opt.rh);
  }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 1; }
  uint32_t hdr[2];
  if (std::fread(hdr, 4, 2, f) != 2) return 1;
  std::vector<V3> pos(hdr[0]); std::vector<uint32_t> idx(3 * (size_t)hdr[1]);
  if (std::fread(pos.data(), 12, hdr[0], f) != hdr[0] || std::fread(idx.data(), 4, idx.size(), f) != idx.size()) return 1;
  std::fclose(f);
  // leaves as the product pairs them: triangles 2k, 2k+1 that read (0, 2, 3) in each other's corners
  const uint32_t nt = hdr[1];
  for (uint32_t t = 0; t < nt;) {
    Prim p; p.nv = 3; p.v[0] = pos[idx[3 * t]]; p.v[1] = pos[idx[3 * t + 1]]; p.v[2] = pos[idx[3 * t + 2]];
    if (t + 1 < nt && idx[3 * t + 3] == idx[3 * t] && idx[3 * t + 4] == idx[3 * t + 2]) { p.v[3] = pos[idx[3 * t + 5]]; p.nv = 4; t += 2; }
    else if (t + 1 < nt && idx[3 * t + 3] == idx[3 * t] && idx[3 * t + 5] == idx[3 * t + 1]) { const V3 b = p.v[2], c = p.v[1]; p.v[1] = pos[idx[3 * t + 4]]; p.v[2] = c; p.v[3] = b; p.nv = 4; t += 2; }   // (0, 3, 1): the second triangle first
    else t += 1;
    prims.push_back(p);
  }
  std::vector<Ref> refs;
  Box scene;
  for (int i = 0; i < (int)prims.size(); ++i) { Ref r; r.prim = i; for (int k = 0; k < prims[i].nv; ++k) r.box.grow(prims[i].v[k]); scene.grow(r.box); refs.push_back(r); }
  root_area = scene.area();
  const size_t n_prims = refs.size();
  if (opt.presplit > 0) {   // early split clipping: a reference whose box is larger than the budget allows is cut across its longest axis
    // budget: total references <= (1 + presplit / 100) * prims, spent on the boxes with the largest area first (priority by area)
    const size_t budget = n_prims * (size_t)opt.presplit / 100;
    auto cmp = [](const Ref& a, const Ref& b) { return a.box.area() < b.box.area(); };
    std::make_heap(refs.begin(), refs.end(), cmp);
    for (size_t s = 0; s < budget; ++s) {
      std::pop_heap(refs.begin(), refs.end(), cmp);
      Ref r = refs.back(); refs.pop_back();
      const V3 e = r.box.hi - r.box.lo;
      const int a = e.x >= e.y && e.x >= e.z ? 0 : (e.y >= e.z ? 1 : 2);
      const float mid = 0.5f * (comp(r.box.lo, a) + comp(r.box.hi, a));
      Ref l = r, h = r;
      l.box = intersect(clip_prim(prims[r.prim], a, -1e30f, mid), r.box);
      h.box = intersect(clip_prim(prims[r.prim], a, mid, 1e30f), r.box);
      if (l.box.empty() || h.box.empty()) { refs.push_back(r); std::push_heap(refs.begin(), refs.end(), cmp); break; }
      refs.push_back(l); std::push_heap(refs.begin(), refs.end(), cmp);
      refs.push_back(h); std::push_heap(refs.begin(), refs.end(), cmp);
    }
  }
  const size_t n_refs_in = refs.size();
  build(refs, 0);
  if (opt.collapse == "sah") dp_prepare(opt.width);
  if (bnodes[0].count > 0) { std::fprintf(stderr, "one leaf\n"); return 1; }
  collapse(0);
  // SAH cost of the binary tree and of the wide tree
  double sah_bin = 0, sah_wide = 0;
  for (auto& b : bnodes) sah_bin += (b.count > 0 ? opt.ci * b.count : opt.ct) * b.box.area();
  for (auto& w : wnodes) for (int k = 0; k < w.n; ++k) sah_wide += (w.link[k] < 0 ? 0.0 : 1.0) * w.box[k].area();
  double kids = 0; int depth = 0;
  for (auto& w : wnodes) kids += w.n;
  { std::function<int(int)> dep = [&](int n) { int m = 0; for (int k = 0; k < wnodes[n].n; ++k) if (wnodes[n].link[k] >= 0) m = std::max(m, dep(wnodes[n].link[k])); return m + 1; }; depth = dep(0); }
  std::printf("%zu leaves (%u triangles), %zu references in, %zu in leaves (+%.1f %%), %lld spatial / %lld object splits; binary SAH %.2f; %d-wide: %zu nodes, %.2f children per node, depth %d, wide-node area sum %.2f\n",
              n_prims, nt, n_refs_in, leaf_refs.size(), 100.0 * (leaf_refs.size() - (double)n_prims) / n_prims, n_spatial, n_object, sah_bin / root_area, opt.width, wnodes.size(), kids / wnodes.size(), depth,
              (sah_wide + scene.area()) / root_area);
  // ---- rays: the bench camera (atrium_scene), cosine bounces, shadow rays to the sun and to random sky directions
  const V3 eye{-18.0f + 4.0f, 2.2f, 0.6f}, target{18.0f, 4.5f, -0.8f};
  const V3 fw = norm(target - eye), rt = norm(cross(fw, V3{0, 1, 0})), up = cross(rt, fw);
  const float tanx = std::tan(0.5f * 75.0f * 3.14159265f / 180.0f), tany = tanx * opt.rh / opt.rw;
  const V3 sun = norm(V3{0.35f, 0.85f, -0.25f});
  std::vector<Count> closest(opt.bounces), shadow(opt.bounces);
  const int W = opt.rw, H = opt.rh;
#pragma omp parallel
  {
    std::vector<Count> lc(opt.bounces), ls(opt.bounces);
#pragma omp for schedule(dynamic, 64)
    for (int pix = 0; pix < W * H; ++pix) {
      std::mt19937 rng(pix * 9781 + 7);
      std::uniform_real_distribution<float> U(0.0f, 1.0f);
      const float sx = ((pix % W + U(rng)) / W * 2.0f - 1.0f) * tanx, sy = (1.0f - (pix / W + U(rng)) / H * 2.0f) * tany;
      V3 o = eye, d = norm(fw + rt * sx + up * sy);
      for (int b = 0; b < opt.bounces; ++b) {
        float t; int prim = -1;
        if (!trace(o, d, 1e-4f, 1e30f, false, lc[b], t, prim)) break;
        const V3 p = o + d * t;
        const Prim& pr = prims[prim];
        V3 n = norm(cross(pr.v[1] - pr.v[0], pr.v[2] - pr.v[0]));
        if (dot(n, d) > 0.0f) n = n * -1.0f;
        // light sample: sun or a uniform sky direction, half each (lights_no = 2)
        V3 ld = sun;
        float lmax = 1e30f;
        if (U(rng) < 0.5f) { const float z = 1.0f - 2.0f * U(rng), ph = 6.2831853f * U(rng), r = std::sqrt(std::max(0.0f, 1.0f - z * z)); ld = V3{r * std::cos(ph), z, r * std::sin(ph)}; }
        if (dot(ld, n) > 0.0f) { float ts; int ps; trace(p, ld, 1e-3f, lmax, true, ls[b], ts, ps); }
        // cosine bounce
        const float u1 = U(rng), u2 = U(rng), r = std::sqrt(u1), ph = 6.2831853f * u2;
        const V3 tx = norm(std::fabs(n.x) > 0.5f ? cross(n, V3{0, 1, 0}) : cross(n, V3{1, 0, 0})), ty = cross(n, tx);
        d = norm(tx * (r * std::cos(ph)) + ty * (r * std::sin(ph)) + n * std::sqrt(std::max(0.0f, 1.0f - u1)));
        o = p;
      }
    }
#pragma omp critical
    for (int b = 0; b < opt.bounces; ++b) {
      closest[b].nodes += lc[b].nodes; closest[b].leaves += lc[b].leaves; closest[b].rays += lc[b].rays; closest[b].hits += lc[b].hits; closest[b].tris += lc[b].tris; closest[b].empty_visits += lc[b].empty_visits;
      shadow[b].nodes += ls[b].nodes; shadow[b].leaves += ls[b].leaves; shadow[b].rays += ls[b].rays; shadow[b].hits += ls[b].hits; shadow[b].tris += ls[b].tris;
    }
  }
  Count c, s;
  for (int b = 0; b < opt.bounces; ++b) {
    c.nodes += closest[b].nodes; c.leaves += closest[b].leaves; c.rays += closest[b].rays; c.hits += closest[b].hits; c.tris += closest[b].tris; c.empty_visits += closest[b].empty_visits;
    s.nodes += shadow[b].nodes; s.leaves += shadow[b].leaves; s.rays += shadow[b].rays; s.hits += shadow[b].hits; s.tris += shadow[b].tris;
  }
  std::printf("closest: %.0f rays, %.2f node visits (%.2f camera, %.2f bounce), %.2f leaf visits per ray, hit %.2f, %.1f %% of visits enter nothing | shadow: %.0f rays, %.2f node visits, %.2f leaf visits per ray, occluded %.2f\n",
              c.rays, c.nodes / c.rays, closest[0].nodes / closest[0].rays, (c.nodes - closest[0].nodes) / std::max(1.0, c.rays - closest[0].rays), c.leaves / c.rays, c.hits / c.rays, 100.0 * c.empty_visits / c.nodes,
              s.rays, s.nodes / s.rays, s.leaves / s.rays, s.hits / s.rays);
  std::printf("per sample (1 closest ray + %.3f shadow rays): %.2f node visits, %.2f leaf visits\n", s.rays / c.rays, (c.nodes + s.nodes) / c.rays, (c.leaves + s.leaves) / c.rays);
  return 0;
}
