// Top-down binned-SAH builder on the host (GLZ_BVH_SAH): a slower build for a cheaper tree (atrium: SAH cost 64.1 against
// 72.2 for the LBVH, 5 % more samples per second).  Input: the leaf boxes in leaf order (leaf i = triangle i of bvh_tris);
// output: the binary hierarchy in the arrays the device passes of kernels_build.hip continue from (children links >= 0
// inner / < 0 ~leaf, parents; inner boxes are fitted on the device).
//
// Every leaf is one triangle, so a subtree over c leaves has exactly c - 1 inner nodes: the node of a range gets the first
// id of a block of c - 1 ids, its left subtree the next c_left - 1, its right subtree the rest.  Ids are therefore the
// depth-first pre-order whatever thread builds which subtree: the tree does not depend on scheduling.
// Threads: the ranges at the top are split one after the other with the binning of a large range spread over the threads;
// once there are enough independent subtrees they are built in parallel.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "kernels.h"

namespace glz {
namespace {
constexpr int kBins = 16;   // 8 / 16 / 32 / 64 bins: SAH cost 64.3 / 64.1 / 64.2 / 64.6 on the atrium, the same render speed
constexpr uint32_t kParallelBinning = 1u << 16;   // ranges at least this large are binned by several threads

struct Box {
  float lo[3], hi[3];
  void reset() { for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
  void grow(const float4& l, const float4& h) {
    lo[0] = std::min(lo[0], l.x); lo[1] = std::min(lo[1], l.y); lo[2] = std::min(lo[2], l.z);
    hi[0] = std::max(hi[0], h.x); hi[1] = std::max(hi[1], h.y); hi[2] = std::max(hi[2], h.z);
  }
  void grow(const Box& b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
  float area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx < 0.0f ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};

struct Bins {
  Box box[3][kBins];
  uint32_t count[3][kBins];
  void reset() {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < kBins; ++b) { box[a][b].reset(); count[a][b] = 0; }
  }
  void merge(const Bins& o) {
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < kBins; ++b) { box[a][b].grow(o.box[a][b]); count[a][b] += o.count[a][b]; }
  }
};

struct Task { uint32_t b, e; int node; };

struct Builder {
  uint32_t n;
  const float4 *lo, *hi;
  int2* children;
  int* parent;
  std::vector<uint32_t> idx;
  std::vector<float> cen;   // 3 per leaf
  std::vector<uint32_t> scratch;   // parallel partition of the large ranges at the top
  unsigned threads;

  static int bin_of(float c, float lo, float scale) {
    const float f = (c - lo) * scale;   // NaN centroids (degenerate input) land in bin 0
    return f >= 0.0f ? (f < (float)kBins ? (int)f : kBins - 1) : 0;
  }
  void centroid_bounds(uint32_t b, uint32_t e, float clo[3], float chi[3]) const {
    for (int k = 0; k < 3; ++k) { clo[k] = INFINITY; chi[k] = -INFINITY; }
    for (uint32_t i = b; i < e; ++i)
      for (int k = 0; k < 3; ++k) {
        const float c = cen[3 * (size_t)idx[i] + k];
        clo[k] = std::min(clo[k], c);
        chi[k] = std::max(chi[k], c);
      }
  }
  void fill_bins(uint32_t b, uint32_t e, const float clo[3], const float scale[3], Bins& bins) const {
    bins.reset();
    for (uint32_t i = b; i < e; ++i) {
      const uint32_t p = idx[i];
      for (int a = 0; a < 3; ++a) {
        if (!(scale[a] > 0.0f)) continue;
        const int k = bin_of(cen[3 * (size_t)p + a], clo[a], scale[a]);
        bins.box[a][k].grow(lo[p], hi[p]);
        ++bins.count[a][k];
      }
    }
  }
  // in parallel over equal chunks of [b, e): f(chunk, chunk_begin, chunk_end)
  template <class F>
  void chunks(uint32_t b, uint32_t e, bool parallel, F f) const {
    const unsigned parts = parallel ? threads : 1u;
    if (parts <= 1) { f(0u, b, e); return; }
    std::vector<std::thread> pool;
    const uint32_t step = (e - b + parts - 1) / parts;
    for (unsigned t = 0; t < parts; ++t) {
      const uint32_t cb = std::min(e, b + t * step), ce = std::min(e, cb + step);
      pool.emplace_back([=] { f(t, cb, ce); });
    }
    for (auto& th : pool) th.join();
  }

  // splits [b, e) (at least 2 leaves), writes the node, returns the sub-ranges that still need a node
  int split(const Task& t, Task out[2], bool parallel) {
    const uint32_t cnt = t.e - t.b;
    uint32_t mid = t.b + cnt / 2;
    bool found = false;
    float clo[3], chi[3];
    if (cnt > 2) {
      if (parallel) {
        std::vector<float> plo(3 * threads, INFINITY), phi(3 * threads, -INFINITY);
        chunks(t.b, t.e, true, [&](unsigned c, uint32_t cb, uint32_t ce) { if (cb < ce) centroid_bounds(cb, ce, &plo[3 * c], &phi[3 * c]); });
        for (int k = 0; k < 3; ++k) {
          clo[k] = INFINITY; chi[k] = -INFINITY;
          for (unsigned c = 0; c < threads; ++c) { clo[k] = std::min(clo[k], plo[3 * c + k]); chi[k] = std::max(chi[k], phi[3 * c + k]); }
        }
      } else {
        centroid_bounds(t.b, t.e, clo, chi);
      }
      float scale[3];
      for (int a = 0; a < 3; ++a) scale[a] = chi[a] - clo[a] > 0.0f ? (float)kBins / (chi[a] - clo[a]) : 0.0f;
      Bins bins;
      if (parallel) {
        std::vector<Bins> part(threads);
        chunks(t.b, t.e, true, [&](unsigned c, uint32_t cb, uint32_t ce) { fill_bins(cb, ce, clo, scale, part[c]); });
        bins.reset();
        for (auto& p : part) bins.merge(p);   // min / max / integer sums: the same whatever the chunking
      } else {
        fill_bins(t.b, t.e, clo, scale, bins);
      }
      float best_cost = INFINITY;
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; ++a) {
        if (!(scale[a] > 0.0f)) continue;
        float right_area[kBins];
        uint32_t right_cnt[kBins];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int k = kBins - 1; k > 0; --k) {
          acc.grow(bins.box[a][k]);
          c += bins.count[a][k];
          right_area[k] = acc.area();
          right_cnt[k] = c;
        }
        acc.reset();
        c = 0;
        for (int k = 0; k < kBins - 1; ++k) {
          acc.grow(bins.box[a][k]);
          c += bins.count[a][k];
          if (c == 0 || right_cnt[k + 1] == 0) continue;
          const float cost = acc.area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
        }
      }
      if (best_axis >= 0) {
        const float l = clo[best_axis], s = scale[best_axis];
        auto left = [&](uint32_t p) { return bin_of(cen[3 * (size_t)p + best_axis], l, s) <= best_bin; };
        if (parallel) {
          // stable partition in chunks: every thread splits its chunk into the scratch array, then the pieces are
          // copied back side by side (left pieces in chunk order, then right pieces): the result of the serial algorithm
          if (scratch.size() < n) scratch.resize(n);
          std::vector<uint32_t> n_left(threads, 0), cb(threads, 0), ce(threads, 0);
          chunks(t.b, t.e, true, [&](unsigned c, uint32_t b0, uint32_t e0) {
            cb[c] = b0; ce[c] = e0;
            uint32_t lpos = b0, rpos = e0;   // left elements grow up from b0, right elements down from e0 (reversed)
            for (uint32_t i = b0; i < e0; ++i) {
              const uint32_t p = idx[i];
              if (left(p)) scratch[lpos++] = p; else scratch[--rpos] = p;
            }
            n_left[c] = lpos - b0;
          });
          uint32_t total_left = 0;
          for (unsigned c = 0; c < threads; ++c) total_left += n_left[c];
          std::vector<uint32_t> lbase(threads), rbase(threads);
          uint32_t lacc = t.b, racc = t.b + total_left;
          for (unsigned c = 0; c < threads; ++c) { lbase[c] = lacc; rbase[c] = racc; lacc += n_left[c]; racc += (ce[c] - cb[c]) - n_left[c]; }
          chunks(t.b, t.e, true, [&](unsigned c, uint32_t, uint32_t) {
            for (uint32_t i = 0; i < n_left[c]; ++i) idx[lbase[c] + i] = scratch[cb[c] + i];
            const uint32_t nr = (ce[c] - cb[c]) - n_left[c];
            for (uint32_t i = 0; i < nr; ++i) idx[rbase[c] + i] = scratch[ce[c] - 1 - i];   // undo the reversal
          });
          mid = t.b + total_left;
        } else {
          auto it = std::stable_partition(idx.begin() + t.b, idx.begin() + t.e, left);
          mid = (uint32_t)(it - idx.begin());
        }
        found = mid > t.b && mid < t.e;
      }
    }
    if (!found) {   // two leaves, or every centroid in one place: halve the range as it stands
      mid = t.b + cnt / 2;
    }
    int link[2], n_out = 0;
    const uint32_t rb[2] = {t.b, mid}, re[2] = {mid, t.e};
    int next_id = t.node + 1;
    for (int s = 0; s < 2; ++s) {
      const uint32_t c = re[s] - rb[s];
      if (c == 1) {
        const uint32_t leaf = idx[rb[s]];
        link[s] = ~(int)leaf;
        parent[(n - 1) + leaf] = t.node;
      } else {
        link[s] = next_id;
        parent[next_id] = t.node;
        out[n_out++] = Task{rb[s], re[s], next_id};
        next_id += (int)c - 1;
      }
    }
    children[t.node] = make_int2(link[0], link[1]);
    return n_out;
  }

  void subtree(const Task& root) {
    std::vector<Task> stack{root};
    while (!stack.empty()) {
      const Task t = stack.back();
      stack.pop_back();
      Task out[2];
      const int k = split(t, out, false);
      for (int i = 0; i < k; ++i) stack.push_back(out[i]);
    }
  }
};
}  // namespace

void build_sah_host(uint32_t n, const float4* lo, const float4* hi, int2* children, int* parent) {
  if (n < 2) return;
  Builder B;
  B.n = n; B.lo = lo; B.hi = hi; B.children = children; B.parent = parent;
  B.threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  B.idx.resize(n);
  B.cen.resize(3 * (size_t)n);
  for (uint32_t i = 0; i < n; ++i) {
    B.idx[i] = i;
    B.cen[3 * (size_t)i] = 0.5f * (lo[i].x + hi[i].x);
    B.cen[3 * (size_t)i + 1] = 0.5f * (lo[i].y + hi[i].y);
    B.cen[3 * (size_t)i + 2] = 0.5f * (lo[i].z + hi[i].z);
  }
  parent[0] = -1;
  // top of the tree: largest range first until there are plenty of independent subtrees
  std::vector<Task> open{Task{0, n, 0}};
  const size_t want = B.threads > 1 ? 8 * (size_t)B.threads : 1;
  while (B.threads > 1 && open.size() < want) {
    size_t big = 0;
    for (size_t i = 1; i < open.size(); ++i)
      if (open[i].e - open[i].b > open[big].e - open[big].b) big = i;
    if (open[big].e - open[big].b < 4096) break;
    const Task t = open[big];
    open.erase(open.begin() + big);
    Task out[2];
    const int k = B.split(t, out, t.e - t.b >= kParallelBinning);
    for (int i = 0; i < k; ++i) open.push_back(out[i]);
  }
  if (B.threads <= 1) {
    for (auto& t : open) B.subtree(t);
    return;
  }
  std::atomic<size_t> next{0};
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < B.threads; ++t)
    pool.emplace_back([&] {
      for (size_t i = next.fetch_add(1); i < open.size(); i = next.fetch_add(1)) B.subtree(open[i]);
    });
  for (auto& th : pool) th.join();
}

}  // namespace glz
