// Binned-SAH BVH builder: bit-exact replacement for build_bvh() of the reference
// (src/bvh/bvh.h:42, src/bvh/bvh.cc:15-169).
//
// Same tree, same node numbering, same in-place permutation of the triangle list as the
// reference's recursive builder, but organised for large inputs (cfg4: 5 M triangles):
//   * per-triangle centroid and bounds are computed once and permuted together with the
//     index triples (the reference recomputes the centroid ~4x per triangle per level,
//     bvh.cc:104,116,64).  glm::min/max keep the leftmost extremal operand, so folding a
//     triangle's pre-reduced box into a node box gives the same bits (signed zeros
//     included) as expanding vertex by vertex (bvh.h:13-28);
//   * an explicit work stack replaces the recursion; children are numbered at split time
//     and the left child is processed first, which reproduces the reference's pre-order
//     allocation (bvh.cc:74-87);
//   * the partition is the hand-written bidirectional algorithm std::partition uses for
//     random-access iterators (bvh.cc:66), so the order never depends on
//     the standard library.
// This translation unit must be compiled with -ffp-contract=off.
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "pathtracer_amd.h"

namespace {

constexpr float kInf = std::numeric_limits<float>::infinity();

struct Vec3 {
  float v[3];
};

// glm::min / glm::max scalar definitions
inline float MinG(float x, float y) { return (y < x) ? y : x; }
inline float MaxG(float x, float y) { return (x < y) ? y : x; }

struct Box {
  Vec3 lo{{kInf, kInf, kInf}};
  Vec3 hi{{-kInf, -kInf, -kInf}};

  void Grow(const Vec3& p) {  // AABB::expand(vec3): max first (bvh.h:13-16)
    for (int k = 0; k < 3; ++k) hi.v[k] = MaxG(hi.v[k], p.v[k]);
    for (int k = 0; k < 3; ++k) lo.v[k] = MinG(lo.v[k], p.v[k]);
  }
  void Grow(const Box& o) {  // AABB::expand(AABB) (bvh.h:25-28)
    for (int k = 0; k < 3; ++k) hi.v[k] = MaxG(hi.v[k], o.hi.v[k]);
    for (int k = 0; k < 3; ++k) lo.v[k] = MinG(lo.v[k], o.lo.v[k]);
  }
  float HalfArea() const {  // AABB::area (bvh.h:30-33)
    float ex = hi.v[0] - lo.v[0], ey = hi.v[1] - lo.v[1], ez = hi.v[2] - lo.v[2];
    return ex * ey + ey * ez + ez * ex;
  }
};

struct TriRef {  // travels with the index triple through every partition
  Vec3 centroid;
  Box bounds;
};

struct Split {
  uint32_t axis = 0;
  float pos = 0.0f;
  float cost = kInf;
};

class Builder {
 public:
  Builder(const pt_vertex* vertices, uint32_t* triangles, uint32_t triangle_count, uint32_t bin_count,
          pt_bvh_node* nodes)
      : tris_(triangles), n_(triangle_count), bins_(bin_count), nodes_(nodes), refs_(triangle_count) {
    for (uint32_t t = 0; t < n_; ++t) {
      const uint32_t* idx = tris_ + 3 * static_cast<size_t>(t);
      Vec3 p[3];
      for (int k = 0; k < 3; ++k) std::memcpy(p[k].v, vertices[idx[k]].position, sizeof(float) * 3);
      TriRef& r = refs_[t];
      for (int c = 0; c < 3; ++c)  // BVHBuilder::centroid (bvh.cc:152-159)
        r.centroid.v[c] = ((p[0].v[c] + p[1].v[c]) + p[2].v[c]) / 3.0f;
      for (int k = 0; k < 3; ++k) r.bounds.Grow(p[k]);
    }
    bin_box_.resize(bins_);
    bin_cnt_.resize(bins_);
    l_area_.resize(bins_);
    r_area_.resize(bins_);
    l_cnt_.resize(bins_);
    r_cnt_.resize(bins_);
  }

  // BVHBuilder::build (bvh.cc:36-50)
  uint32_t Run(uint32_t* node_count_out) {
    for (uint32_t i = 0; i < 2 * n_; ++i) {
      pt_bvh_node& nd = nodes_[i];
      for (int k = 0; k < 3; ++k) {
        nd.aabb_min[k] = kInf;
        nd.aabb_max[k] = -kInf;
      }
      nd.left_child_index = nd.begin_index = nd.triangle_count = 0;
    }
    nodes_[0].triangle_count = n_;
    uint32_t used = 2;  // node 1 is never used: sibling pairs sit at even indices (bvh.cc:19)
    uint32_t max_depth = 0;

    struct Work {
      uint32_t node, depth;
    };
    std::vector<Work> todo;
    todo.push_back({0, 1});
    while (!todo.empty()) {
      Work w = todo.back();
      todo.pop_back();
      if (w.depth > max_depth) max_depth = w.depth;
      pt_bvh_node& nd = nodes_[w.node];
      const uint32_t begin = nd.begin_index, count = nd.triangle_count;

      Box box;  // update_bounds (bvh.cc:90-94)
      for (uint32_t i = 0; i < count; ++i) box.Grow(refs_[begin + i].bounds);
      std::memcpy(nd.aabb_min, box.lo.v, sizeof(float) * 3);
      std::memcpy(nd.aabb_max, box.hi.v, sizeof(float) * 3);

      Split s = FindSplit(begin, count);
      float nosplit_cost = box.HalfArea() * static_cast<float>(count);
      if (s.cost >= nosplit_cost) continue;  // leaf (bvh.cc:58-59)

      uint32_t mid = Partition(begin, begin + count, s.axis, s.pos);
      uint32_t left_n = mid - begin;
      if (left_n == 0 || left_n == count) continue;  // leaf (bvh.cc:70-72)

      nd.left_child_index = used;
      pt_bvh_node& l = nodes_[used];
      pt_bvh_node& r = nodes_[used + 1];
      l.begin_index = begin;
      l.triangle_count = left_n;
      r.begin_index = mid;
      r.triangle_count = count - left_n;
      nd.triangle_count = 0;
      todo.push_back({used + 1, w.depth + 1});
      todo.push_back({used, w.depth + 1});  // popped first: left subtree is numbered first
      used += 2;
    }
    *node_count_out = used;
    return max_depth;
  }

 private:
  // BVHBuilder::find_split (bvh.cc:96-150)
  Split FindSplit(uint32_t begin, uint32_t count) {
    Split best;
    Vec3 cmin{{kInf, kInf, kInf}}, cmax{{-kInf, -kInf, -kInf}};
    for (uint32_t i = 0; i < count; ++i) {
      const Vec3& c = refs_[begin + i].centroid;
      for (int k = 0; k < 3; ++k) {
        cmin.v[k] = MinG(cmin.v[k], c.v[k]);
        cmax.v[k] = MaxG(cmax.v[k], c.v[k]);
      }
    }
    const uint32_t nb = bins_;
    for (uint32_t axis = 0; axis < 3; ++axis) {
      const float lo = cmin.v[axis], hi = cmax.v[axis];
      // Degenerate extent: the reference's bin scale is inf, every cost on
      // this axis is NaN and `cost < split_cost` never holds — the axis cannot win.
      if (!(hi > lo)) continue;
      for (uint32_t b = 0; b < nb; ++b) {
        bin_box_[b] = Box{};
        bin_cnt_[b] = 0;
      }
      float scale = static_cast<float>(nb) / (hi - lo);
      for (uint32_t i = 0; i < count; ++i) {
        const TriRef& r = refs_[begin + i];
        uint32_t b = static_cast<uint32_t>((r.centroid.v[axis] - lo) * scale);
        if (b > nb - 1) b = nb - 1;
        ++bin_cnt_[b];
        bin_box_[b].Grow(r.bounds);
      }
      Box lbox, rbox;
      uint32_t lsum = 0, rsum = 0;
      for (uint32_t i = 0; i + 1 < nb; ++i) {
        lsum += bin_cnt_[i];
        l_cnt_[i] = lsum;
        lbox.Grow(bin_box_[i]);
        l_area_[i] = lbox.HalfArea();
        rsum += bin_cnt_[nb - i - 1];
        r_cnt_[nb - i - 2] = rsum;
        rbox.Grow(bin_box_[nb - i - 1]);
        r_area_[nb - i - 2] = rbox.HalfArea();
      }
      scale = (hi - lo) / static_cast<float>(nb);
      for (uint32_t i = 0; i + 1 < nb; ++i) {
        float cost = static_cast<float>(l_cnt_[i]) * l_area_[i] + static_cast<float>(r_cnt_[i]) * r_area_[i];
        if (cost < best.cost) {
          best.axis = axis;
          best.pos = lo + scale * static_cast<float>(i + 1);
          best.cost = cost;
        }
      }
    }
    return best;
  }

  void SwapTris(uint32_t a, uint32_t b) {
    uint32_t* ta = tris_ + 3 * static_cast<size_t>(a);
    uint32_t* tb = tris_ + 3 * static_cast<size_t>(b);
    for (int k = 0; k < 3; ++k) {
      uint32_t t = ta[k];
      ta[k] = tb[k];
      tb[k] = t;
    }
    TriRef t = refs_[a];
    refs_[a] = refs_[b];
    refs_[b] = t;
  }

  // std::partition, bidirectional form (libstdc++/libc++/MSVC agree for random access)
  uint32_t Partition(uint32_t first, uint32_t last, uint32_t axis, float pos) {
    auto goes_left = [&](uint32_t i) { return refs_[i].centroid.v[axis] < pos; };
    for (;;) {
      for (;;) {
        if (first == last) return first;
        if (goes_left(first)) ++first;
        else break;
      }
      --last;
      for (;;) {
        if (first == last) return first;
        if (!goes_left(last)) --last;
        else break;
      }
      SwapTris(first, last);
      ++first;
    }
  }

  uint32_t* tris_;
  uint32_t n_;
  uint32_t bins_;
  pt_bvh_node* nodes_;
  std::vector<TriRef> refs_;
  std::vector<Box> bin_box_;
  std::vector<uint32_t> bin_cnt_, l_cnt_, r_cnt_;
  std::vector<float> l_area_, r_area_;
};

}  // namespace

extern "C" int pt_build_bvh(const pt_vertex* vertices, uint32_t vertex_count, uint32_t* triangles,
                            uint32_t triangle_count, uint32_t bin_count, pt_bvh_node* nodes_out,
                            uint32_t* node_count_out, uint32_t* depth_out) {
  if (!vertices || !triangles || !nodes_out || !node_count_out || !depth_out) return PT_ERR_INVALID;
  if (triangle_count == 0 || bin_count < 2 || triangle_count > 0x7FFFFFFFu) return PT_ERR_INVALID;
  for (size_t i = 0; i < 3 * static_cast<size_t>(triangle_count); ++i)
    if (triangles[i] >= vertex_count) return PT_ERR_INVALID;
  try {
    Builder b(vertices, triangles, triangle_count, bin_count, nodes_out);
    *depth_out = b.Run(node_count_out);
  } catch (...) {
    return PT_ERR_INVALID;
  }
  return PT_OK;
}
