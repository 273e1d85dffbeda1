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
//
// pt_build_bvh_mt builds the SAME arrays with several host threads:
//   * nodes with many triangles are split one at a time by all threads together: bounds,
//     centroid range and bins are reduced per contiguous chunk and merged in chunk order
//     (glm::min/max keep the leftmost extremal operand, an associative choice, so the merged
//     values carry the same bits, signed zeros included), and the partition performs exactly
//     the swaps of the bidirectional algorithm: its k-th swap exchanges the k-th element from
//     the left that belongs right with the k-th element from the right that belongs left,
//     so the swap list can be computed from two compacted position lists and applied in
//     parallel;
//   * the remaining subtrees are independent (disjoint triangle ranges): each is built by one
//     thread with the sequential code into a private node array in subtree-local numbering;
//   * numbering: the sequential builder numbers a subtree's nodes contiguously (pair first,
//     then everything under the left child, then everything under the right child), so one
//     pre-order walk over the top of the tree assigns every private array its base index.
// This translation unit must be compiled with -ffp-contract=off.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <stdexcept>
#include <thread>
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

struct Scratch {  // find_split's per-bin tables: one set per thread
  explicit Scratch(uint32_t bins) : bin_box(bins), bin_cnt(bins), l_cnt(bins), r_cnt(bins), l_area(bins), r_area(bins) {}
  std::vector<Box> bin_box;
  std::vector<uint32_t> bin_cnt, l_cnt, r_cnt;
  std::vector<float> l_area, r_area;
};

void ClearNode(pt_bvh_node& nd) {
  for (int k = 0; k < 3; ++k) {
    nd.aabb_min[k] = kInf;
    nd.aabb_max[k] = -kInf;
  }
  nd.left_child_index = nd.begin_index = nd.triangle_count = 0;
}

// Fork-join helper: Run(f) calls f(tid) on `size` threads (the caller is tid 0) and returns when all are done.
class Pool {
 public:
  explicit Pool(unsigned size) : size_(size) {
    try {
      for (unsigned t = 1; t < size_; ++t) threads_.emplace_back([this, t] { Loop(t); });
    } catch (...) {  // the system refuses more threads: go on with the ones that started
      size_ = static_cast<unsigned>(threads_.size()) + 1;
    }
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      ++generation_;
    }
    go_.notify_all();
    for (std::thread& t : threads_) t.join();
  }
  unsigned size() const { return size_; }
  bool failed() const { return failed_.load(); }

  void Run(const std::function<void(unsigned)>& f) {
    {
      std::lock_guard<std::mutex> g(m_);
      job_ = &f;
      pending_ = size_ - 1;
      ++generation_;
    }
    go_.notify_all();
    Call(f, 0);
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
  }

 private:
  void Call(const std::function<void(unsigned)>& f, unsigned tid) {
    try {
      f(tid);
    } catch (...) {
      failed_.store(true);
    }
  }
  void Loop(unsigned tid) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* job;
      {
        std::unique_lock<std::mutex> g(m_);
        go_.wait(g, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
        job = job_;
      }
      Call(*job, tid);
      {
        std::lock_guard<std::mutex> g(m_);
        --pending_;
      }
      done_.notify_one();
    }
  }
  unsigned size_;
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable go_, done_;
  const std::function<void(unsigned)>* job_ = nullptr;
  uint64_t generation_ = 0;
  unsigned pending_ = 0;
  bool stop_ = false;
  std::atomic<bool> failed_{false};
};

class Builder {
 public:
  Builder(const pt_vertex* vertices, uint32_t* triangles, uint32_t triangle_count, uint32_t bin_count)
      : vertices_(vertices), tris_(triangles), n_(triangle_count), bins_(bin_count), refs_(triangle_count) {}

  void MakeRefs(uint32_t begin, uint32_t end) {
    for (uint32_t t = begin; t < end; ++t) {
      const uint32_t* idx = tris_ + 3 * static_cast<size_t>(t);
      Vec3 p[3];
      for (int k = 0; k < 3; ++k) std::memcpy(p[k].v, vertices_[idx[k]].position, sizeof(float) * 3);
      TriRef& r = refs_[t];
      for (int c = 0; c < 3; ++c)  // BVHBuilder::centroid (bvh.cc:152-159)
        r.centroid.v[c] = ((p[0].v[c] + p[1].v[c]) + p[2].v[c]) / 3.0f;
      for (int k = 0; k < 3; ++k) r.bounds.Grow(p[k]);
    }
  }

  // BVHBuilder::build / subdivide (bvh.cc:36-87) below nodes[root], whose begin_index and
  // triangle_count are set. New pairs are numbered from `used` on; returns the next free
  // index. The whole tree: root 0, used 2 (node 1 is never used: sibling pairs sit at even
  // indices, bvh.cc:19). A private subtree: root 0, used 1.
  uint32_t Subdivide(pt_bvh_node* nodes, uint32_t root, uint32_t used, uint32_t root_depth, Scratch& sc,
                     uint32_t* max_depth) const {
    struct Work {
      uint32_t node, depth;
    };
    std::vector<Work> todo;
    todo.push_back({root, root_depth});
    while (!todo.empty()) {
      Work w = todo.back();
      todo.pop_back();
      if (w.depth > *max_depth) *max_depth = w.depth;
      pt_bvh_node& nd = nodes[w.node];
      const uint32_t begin = nd.begin_index, count = nd.triangle_count;

      Box box;  // update_bounds (bvh.cc:90-94)
      for (uint32_t i = 0; i < count; ++i) box.Grow(refs_[begin + i].bounds);
      std::memcpy(nd.aabb_min, box.lo.v, sizeof(float) * 3);
      std::memcpy(nd.aabb_max, box.hi.v, sizeof(float) * 3);

      Split s = FindSplit(begin, count, sc);
      float nosplit_cost = box.HalfArea() * static_cast<float>(count);
      if (s.cost >= nosplit_cost) continue;  // leaf (bvh.cc:58-59)

      uint32_t mid = Partition(begin, begin + count, s.axis, s.pos);
      uint32_t left_n = mid - begin;
      if (left_n == 0 || left_n == count) continue;  // leaf (bvh.cc:70-72)

      nd.left_child_index = used;
      pt_bvh_node& l = nodes[used];
      pt_bvh_node& r = nodes[used + 1];
      l.begin_index = begin;
      l.triangle_count = left_n;
      r.begin_index = mid;
      r.triangle_count = count - left_n;
      nd.triangle_count = 0;
      todo.push_back({used + 1, w.depth + 1});
      todo.push_back({used, w.depth + 1});  // popped first: left subtree is numbered first
      used += 2;
    }
    return used;
  }

  uint32_t RunSequential(pt_bvh_node* nodes, uint32_t* node_count_out) {
    MakeRefs(0, n_);
    for (uint32_t i = 0; i < 2 * n_; ++i) ClearNode(nodes[i]);
    nodes[0].triangle_count = n_;
    Scratch sc(bins_);
    uint32_t max_depth = 0;
    *node_count_out = Subdivide(nodes, 0, 2, 1, sc, &max_depth);
    return max_depth;
  }

  uint32_t RunParallel(pt_bvh_node* nodes, uint32_t* node_count_out, unsigned threads);

 private:
  // The sweep over the bins of one axis (bvh.cc:121-147); the bins are in sc.bin_box / sc.bin_cnt
  void SweepAxis(uint32_t axis, float lo, float hi, Scratch& sc, Split& best) const {
    const uint32_t nb = bins_;
    Box lbox, rbox;
    uint32_t lsum = 0, rsum = 0;
    for (uint32_t i = 0; i + 1 < nb; ++i) {
      lsum += sc.bin_cnt[i];
      sc.l_cnt[i] = lsum;
      lbox.Grow(sc.bin_box[i]);
      sc.l_area[i] = lbox.HalfArea();
      rsum += sc.bin_cnt[nb - i - 1];
      sc.r_cnt[nb - i - 2] = rsum;
      rbox.Grow(sc.bin_box[nb - i - 1]);
      sc.r_area[nb - i - 2] = rbox.HalfArea();
    }
    const float scale = (hi - lo) / static_cast<float>(nb);
    for (uint32_t i = 0; i + 1 < nb; ++i) {
      float cost = static_cast<float>(sc.l_cnt[i]) * sc.l_area[i] + static_cast<float>(sc.r_cnt[i]) * sc.r_area[i];
      if (cost < best.cost) {
        best.axis = axis;
        best.pos = lo + scale * static_cast<float>(i + 1);
        best.cost = cost;
      }
    }
  }

  // one axis' bins over [begin, end) into (box, cnt), which the caller has cleared
  void BinRange(uint32_t begin, uint32_t end, uint32_t axis, float lo, float scale, Box* box, uint32_t* cnt) const {
    const uint32_t nb = bins_;
    for (uint32_t i = begin; i < end; ++i) {
      const TriRef& r = refs_[i];
      uint32_t b = static_cast<uint32_t>((r.centroid.v[axis] - lo) * scale);
      if (b > nb - 1) b = nb - 1;
      ++cnt[b];
      box[b].Grow(r.bounds);
    }
  }

  // BVHBuilder::find_split (bvh.cc:96-150)
  Split FindSplit(uint32_t begin, uint32_t count, Scratch& sc) const {
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
        sc.bin_box[b] = Box{};
        sc.bin_cnt[b] = 0;
      }
      BinRange(begin, begin + count, axis, lo, static_cast<float>(nb) / (hi - lo), sc.bin_box.data(), sc.bin_cnt.data());
      SweepAxis(axis, lo, hi, sc, best);
    }
    return best;
  }

  void SwapTris(uint32_t a, uint32_t b) const {
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
  uint32_t Partition(uint32_t first, uint32_t last, uint32_t axis, float pos) const {
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

  const pt_vertex* vertices_;
  uint32_t* tris_;
  uint32_t n_;
  uint32_t bins_;
  mutable std::vector<TriRef> refs_;  // permuted together with tris_ (disjoint ranges per thread)
};

// ---- the parallel build

struct TopNode {  // a node split by all threads together, or the root of a private subtree
  uint32_t begin, count, depth;
  Box box;
  int32_t left = -1, right = -1;  // TopNode indices when split
  int32_t task = -1;              // private subtree, else (left < 0): a leaf decided at the top
};

struct Subtree {
  uint32_t top;                    // its TopNode
  std::vector<pt_bvh_node> nodes;  // [0] = the root, pairs from 1 on, subtree-local child indices
  uint32_t used = 1;
  uint32_t max_depth = 0;
  uint32_t base = 0;  // global index of local node 1
};

inline uint32_t ChunkEdge(uint32_t begin, uint32_t count, unsigned parts, unsigned k) {
  return begin + static_cast<uint32_t>(static_cast<uint64_t>(count) * k / parts);
}

uint32_t Builder::RunParallel(pt_bvh_node* nodes, uint32_t* node_count_out, unsigned threads) {
  Pool pool(threads);
  const unsigned T = pool.size();
  const uint32_t nb = bins_;
  pool.Run([&](unsigned t) {
    MakeRefs(ChunkEdge(0, n_, T, t), ChunkEdge(0, n_, T, t + 1));
    for (uint32_t i = ChunkEdge(0, 2 * n_, T, t), e = ChunkEdge(0, 2 * n_, T, t + 1); i < e; ++i) ClearNode(nodes[i]);
  });

  // --- top of the tree: nodes above `big` triangles, one at a time, all threads on each
  const uint32_t big = std::max<uint32_t>(n_ / (8 * T), 4096);
  std::vector<TopNode> top;
  std::vector<Subtree> subtrees;
  top.reserve(64 * T + 64);
  {
    TopNode r;
    r.begin = 0;
    r.count = n_;
    r.depth = 1;
    top.push_back(r);
  }
  Scratch sc(nb);
  struct ChunkStat {
    Box box;
    Vec3 cmin{{kInf, kInf, kInf}}, cmax{{-kInf, -kInf, -kInf}};
    uint32_t n_left = 0, n_a = 0, n_b = 0;
  };
  std::vector<ChunkStat> stat(T);
  std::vector<Box> tbox(static_cast<size_t>(T) * 3 * nb);
  std::vector<uint32_t> tcnt(static_cast<size_t>(T) * 3 * nb);
  std::vector<uint32_t> pos_a(n_), pos_b(n_);  // compacted positions, per chunk at the chunk's own offset
  std::vector<uint32_t> todo{0};
  while (!todo.empty()) {
    const uint32_t ti = todo.back();
    todo.pop_back();
    const uint32_t begin = top[ti].begin, count = top[ti].count;
    if (count <= big) {
      top[ti].task = static_cast<int32_t>(subtrees.size());
      Subtree st;
      st.top = ti;
      subtrees.push_back(std::move(st));
      continue;
    }
    // update_bounds + the centroid range, per chunk, merged in chunk order
    pool.Run([&](unsigned t) {
      ChunkStat cs;
      for (uint32_t i = ChunkEdge(begin, count, T, t), e = ChunkEdge(begin, count, T, t + 1); i < e; ++i) {
        const TriRef& r = refs_[i];
        cs.box.Grow(r.bounds);
        for (int k = 0; k < 3; ++k) {
          cs.cmin.v[k] = MinG(cs.cmin.v[k], r.centroid.v[k]);
          cs.cmax.v[k] = MaxG(cs.cmax.v[k], r.centroid.v[k]);
        }
      }
      stat[t] = cs;
    });
    Box box;
    Vec3 cmin{{kInf, kInf, kInf}}, cmax{{-kInf, -kInf, -kInf}};
    for (unsigned t = 0; t < T; ++t) {
      box.Grow(stat[t].box);
      for (int k = 0; k < 3; ++k) {
        cmin.v[k] = MinG(cmin.v[k], stat[t].cmin.v[k]);
        cmax.v[k] = MaxG(cmax.v[k], stat[t].cmax.v[k]);
      }
    }
    top[ti].box = box;
    // find_split: the three axes' bins per chunk in one pass, merged per bin in chunk order
    pool.Run([&](unsigned t) {
      const uint32_t b0 = ChunkEdge(begin, count, T, t), b1 = ChunkEdge(begin, count, T, t + 1);
      for (uint32_t axis = 0; axis < 3; ++axis) {
        Box* bx = tbox.data() + (static_cast<size_t>(t) * 3 + axis) * nb;
        uint32_t* bc = tcnt.data() + (static_cast<size_t>(t) * 3 + axis) * nb;
        const float lo = cmin.v[axis], hi = cmax.v[axis];
        if (!(hi > lo)) continue;
        for (uint32_t b = 0; b < nb; ++b) {
          bx[b] = Box{};
          bc[b] = 0;
        }
        BinRange(b0, b1, axis, lo, static_cast<float>(nb) / (hi - lo), bx, bc);
      }
    });
    Split s;
    for (uint32_t axis = 0; axis < 3; ++axis) {
      const float lo = cmin.v[axis], hi = cmax.v[axis];
      if (!(hi > lo)) continue;
      for (uint32_t b = 0; b < nb; ++b) {
        sc.bin_box[b] = Box{};
        sc.bin_cnt[b] = 0;
      }
      for (unsigned t = 0; t < T; ++t) {
        const Box* bx = tbox.data() + (static_cast<size_t>(t) * 3 + axis) * nb;
        const uint32_t* bc = tcnt.data() + (static_cast<size_t>(t) * 3 + axis) * nb;
        for (uint32_t b = 0; b < nb; ++b) {
          sc.bin_cnt[b] += bc[b];
          if (bc[b]) sc.bin_box[b].Grow(bx[b]);
        }
      }
      SweepAxis(axis, lo, hi, sc, s);
    }
    const float nosplit_cost = box.HalfArea() * static_cast<float>(count);
    if (s.cost >= nosplit_cost) continue;  // leaf

    // partition, pass 1: where the boundary will be
    const uint32_t axis = s.axis;
    const float pos = s.pos;
    pool.Run([&](unsigned t) {
      uint32_t c = 0;
      for (uint32_t i = ChunkEdge(begin, count, T, t), e = ChunkEdge(begin, count, T, t + 1); i < e; ++i)
        c += refs_[i].centroid.v[axis] < pos ? 1u : 0u;
      stat[t].n_left = c;
    });
    uint32_t left_n = 0;
    for (unsigned t = 0; t < T; ++t) left_n += stat[t].n_left;
    if (left_n == 0 || left_n == count) continue;  // leaf: the sequential partition swaps nothing in either case
    const uint32_t mid = begin + left_n;
    // pass 2: A = misplaced positions left of the boundary (ascending), B = right of it
    pool.Run([&](unsigned t) {
      uint32_t a0 = ChunkEdge(begin, left_n, T, t), a1 = ChunkEdge(begin, left_n, T, t + 1), na = 0;
      for (uint32_t i = a0; i < a1; ++i)
        if (!(refs_[i].centroid.v[axis] < pos)) pos_a[a0 + na++] = i;
      uint32_t b0 = ChunkEdge(mid, count - left_n, T, t), b1 = ChunkEdge(mid, count - left_n, T, t + 1), nbq = 0;
      for (uint32_t i = b0; i < b1; ++i)
        if (refs_[i].centroid.v[axis] < pos) pos_b[b0 + nbq++] = i;
      stat[t].n_a = na;
      stat[t].n_b = nbq;
    });
    std::vector<uint32_t> off_a(T + 1, 0), off_b(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) {
      off_a[t + 1] = off_a[t] + stat[t].n_a;
      off_b[t + 1] = off_b[t] + stat[t].n_b;
    }
    const uint32_t swaps = off_a[T];
    if (swaps != off_b[T]) throw std::runtime_error("partition bookkeeping");
    // pass 3: swap k exchanges A[k] with B[swaps-1-k]
    pool.Run([&](unsigned t) {
      uint32_t k = off_a[t];
      if (stat[t].n_a == 0) return;
      const uint32_t a0 = ChunkEdge(begin, left_n, T, t);
      uint32_t rank_b = swaps - 1 - k;  // rank in B (ascending order) of the first partner
      unsigned cb = static_cast<unsigned>(std::upper_bound(off_b.begin(), off_b.end(), rank_b) - off_b.begin()) - 1;
      for (uint32_t j = 0; j < stat[t].n_a; ++j, ++k) {
        rank_b = swaps - 1 - k;
        while (rank_b < off_b[cb]) --cb;
        const uint32_t b = pos_b[ChunkEdge(mid, count - left_n, T, cb) + (rank_b - off_b[cb])];
        SwapTris(pos_a[a0 + j], b);
      }
    });

    TopNode l, r;
    l.begin = begin;
    l.count = left_n;
    r.begin = mid;
    r.count = count - left_n;
    l.depth = r.depth = top[ti].depth + 1;
    top[ti].left = static_cast<int32_t>(top.size());
    top[ti].right = top[ti].left + 1;
    top.push_back(l);
    top.push_back(r);
    todo.push_back(static_cast<uint32_t>(top[ti].right));
    todo.push_back(static_cast<uint32_t>(top[ti].left));
  }
  if (pool.failed()) throw std::runtime_error("worker failed");

  // --- the private subtrees, largest first, one thread each
  std::vector<uint32_t> order(subtrees.size());
  for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const uint32_t ca = top[subtrees[a].top].count, cb = top[subtrees[b].top].count;
    return ca != cb ? ca > cb : a < b;
  });
  std::atomic<uint32_t> next{0};
  pool.Run([&](unsigned) {
    Scratch mine(nb);
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= order.size()) break;
      Subtree& st = subtrees[order[k]];
      const TopNode& tn = top[st.top];
      st.nodes.resize(2 * static_cast<size_t>(tn.count));
      for (pt_bvh_node& nd : st.nodes) ClearNode(nd);
      st.nodes[0].begin_index = tn.begin;
      st.nodes[0].triangle_count = tn.count;
      st.used = Subdivide(st.nodes.data(), 0, 1, tn.depth, mine, &st.max_depth);
    }
  });
  if (pool.failed()) throw std::runtime_error("worker failed");

  // --- numbering: the sequential builder's pre-order, pairs allocated at split time
  uint32_t used = 2, max_depth = 0;
  struct Visit {
    uint32_t top, index;
  };
  std::vector<Visit> walk{{0, 0}};
  std::vector<uint32_t> global_of(top.size(), 0);
  while (!walk.empty()) {
    const Visit v = walk.back();
    walk.pop_back();
    const TopNode& tn = top[v.top];
    global_of[v.top] = v.index;
    if (tn.depth > max_depth) max_depth = tn.depth;
    if (tn.task >= 0) {
      Subtree& st = subtrees[static_cast<size_t>(tn.task)];
      st.base = used;
      used += st.used - 1;
      if (st.max_depth > max_depth) max_depth = st.max_depth;
      continue;
    }
    pt_bvh_node& nd = nodes[v.index];
    std::memcpy(nd.aabb_min, tn.box.lo.v, sizeof(float) * 3);
    std::memcpy(nd.aabb_max, tn.box.hi.v, sizeof(float) * 3);
    nd.begin_index = tn.begin;
    nd.triangle_count = tn.count;
    if (tn.left >= 0) {
      nd.left_child_index = used;
      nd.triangle_count = 0;
      walk.push_back({static_cast<uint32_t>(tn.right), used + 1});
      walk.push_back({static_cast<uint32_t>(tn.left), used});
      used += 2;
    }
  }
  next.store(0);
  pool.Run([&](unsigned) {
    for (;;) {
      const uint32_t k = next.fetch_add(1);
      if (k >= order.size()) break;
      const Subtree& st = subtrees[order[k]];
      for (uint32_t j = 0; j < st.used; ++j) {
        pt_bvh_node nd = st.nodes[j];
        if (nd.triangle_count == 0 && nd.left_child_index != 0) nd.left_child_index += st.base - 1;
        nodes[j == 0 ? global_of[st.top] : st.base + j - 1] = nd;
      }
    }
  });
  *node_count_out = used;
  return max_depth;
}

}  // namespace

static int BuildChecked(const pt_vertex* vertices, uint32_t vertex_count, uint32_t* triangles, uint32_t triangle_count,
                        uint32_t bin_count, pt_bvh_node* nodes_out, uint32_t* node_count_out, uint32_t* depth_out,
                        unsigned threads) {
  if (!vertices || !triangles || !nodes_out || !node_count_out || !depth_out) return PT_ERR_INVALID;
  if (triangle_count == 0 || bin_count < 2 || triangle_count > 0x7FFFFFFFu) return PT_ERR_INVALID;
  for (size_t i = 0; i < 3 * static_cast<size_t>(triangle_count); ++i)
    if (triangles[i] >= vertex_count) return PT_ERR_INVALID;
  try {
    Builder b(vertices, triangles, triangle_count, bin_count);
    *depth_out = threads > 1 ? b.RunParallel(nodes_out, node_count_out, threads) : b.RunSequential(nodes_out, node_count_out);
  } catch (...) {
    return PT_ERR_INVALID;
  }
  return PT_OK;
}

extern "C" int pt_build_bvh_mt(const pt_vertex* vertices, uint32_t vertex_count, uint32_t* triangles,
                               uint32_t triangle_count, uint32_t bin_count, pt_bvh_node* nodes_out,
                               uint32_t* node_count_out, uint32_t* depth_out, uint32_t threads) {
  if (threads == 0) {
    threads = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("PT_BUILD_THREADS")) threads = static_cast<uint32_t>(std::atoi(e));
    if (threads > 16) threads = 16;  // the partition passes are memory-bound well before that
    if (threads < 1) threads = 1;
  }
  if (threads > 256) threads = 256;
  return BuildChecked(vertices, vertex_count, triangles, triangle_count, bin_count, nodes_out, node_count_out, depth_out, threads);
}

extern "C" int pt_build_bvh(const pt_vertex* vertices, uint32_t vertex_count, uint32_t* triangles,
                            uint32_t triangle_count, uint32_t bin_count, pt_bvh_node* nodes_out,
                            uint32_t* node_count_out, uint32_t* depth_out) {
  // same arrays either way; small inputs are not worth starting threads for
  return triangle_count < 65536u
             ? BuildChecked(vertices, vertex_count, triangles, triangle_count, bin_count, nodes_out, node_count_out, depth_out, 1)
             : pt_build_bvh_mt(vertices, vertex_count, triangles, triangle_count, bin_count, nodes_out, node_count_out, depth_out, 0);
}
