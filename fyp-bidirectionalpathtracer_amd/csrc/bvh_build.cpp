// bvh_build.cpp — binned-SAH top-down builder producing the flat 64-byte-node layout of bvh.h.
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace bdpt {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    lo[0] = lo[1] = lo[2] = 1e30f;
    hi[0] = hi[1] = hi[2] = -1e30f;
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], b.lo[a]);
      hi[a] = std::max(hi[a], b.hi[a]);
    }
  }
  void grow(const float* p) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], p[a]);
      hi[a] = std::max(hi[a], p[a]);
    }
  }
  float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};

struct TmpNode {
  Box box;
  int32_t left = -1, right = -1;  // children (tmp indices) or -1
  uint32_t first = 0, count = 0;  // leaf range in `order`
  uint32_t depth = 0;
};

constexpr int kBins = 16;
constexpr uint32_t kLeafMax = 4;
constexpr float kCostTraverse = 1.0f, kCostTri = 1.0f;

inline uint32_t ceilLog2(uint32_t x) {
  uint32_t l = 0;
  while ((1u << l) < x) l++;
  return l;
}

}  // namespace

void buildBvh(const float* positions, const uint32_t* indices, uint32_t n, const uint32_t* triFlags, Bvh& out) {
  out.nodes.clear();
  out.tris.clear();
  out.maxDepth = 0;
  out.sahCost = 0.0f;

  // Triangle records exactly as the device intersects them: the "actual" triangle is
  // (v0, v0+e1, v0+e2) in fp32, so bounds are taken from those points.
  std::vector<BvhTri> recs(n);
  std::vector<Box> boxes(n);
  std::vector<float> cent((size_t)n * 3);
  Box scene;
  scene.reset();
  for (uint32_t t = 0; t < n; t++) {
    const float* a = positions + (size_t)indices[(size_t)t * 3] * 3;
    const float* b = positions + (size_t)indices[(size_t)t * 3 + 1] * 3;
    const float* c = positions + (size_t)indices[(size_t)t * 3 + 2] * 3;
    BvhTri& r = recs[t];
    float p1[3], p2[3];
    for (int k = 0; k < 3; k++) {
      r.v0[k] = a[k];
      r.e1[k] = b[k] - a[k];
      r.e2[k] = c[k] - a[k];
      p1[k] = r.v0[k] + r.e1[k];
      p2[k] = r.v0[k] + r.e2[k];
    }
    r.prim = t;
    r.flags = triFlags ? triFlags[t] : 0u;
    r.pad = 0;
    Box bx;
    bx.reset();
    bx.grow(r.v0);
    bx.grow(p1);
    bx.grow(p2);
    bx.grow(b);
    bx.grow(c);
    boxes[t] = bx;
    for (int k = 0; k < 3; k++) cent[(size_t)t * 3 + k] = 0.5f * (bx.lo[k] + bx.hi[k]);
    scene.grow(bx);
  }
  float diag = 0.0f;
  if (n) {
    float dx = scene.hi[0] - scene.lo[0], dy = scene.hi[1] - scene.lo[1], dz = scene.hi[2] - scene.lo[2];
    diag = std::sqrt(dx * dx + dy * dy + dz * dz);
  }
  // Slab tests run in fp32 on boxes that must never reject a hit the triangle test accepts:
  // pad every box by a small fraction of the scene diagonal (covers rounding in both tests).
  const float pad = 2e-5f * diag + 1e-30f;

  std::vector<uint32_t> order(n);
  for (uint32_t i = 0; i < n; i++) order[i] = i;

  std::vector<TmpNode> tmp;
  tmp.reserve((size_t)n / 2 + 16);
  std::vector<uint32_t> todo;
  {
    TmpNode root;
    root.first = 0;
    root.count = n;
    root.depth = 0;
    tmp.push_back(root);
    todo.push_back(0);
  }
  while (!todo.empty()) {
    uint32_t ni = todo.back();
    todo.pop_back();
    uint32_t first = tmp[ni].first, count = tmp[ni].count, depth = tmp[ni].depth;
    Box nb, cb;
    nb.reset();
    cb.reset();
    for (uint32_t k = 0; k < count; k++) {
      uint32_t t = order[first + k];
      nb.grow(boxes[t]);
      cb.grow(&cent[(size_t)t * 3]);
    }
    tmp[ni].box = nb;
    out.maxDepth = std::max(out.maxDepth, depth);
    if (count <= kLeafMax) continue;

    // Depth budget: once the remaining levels are only just enough for a balanced split of
    // `count` triangles into leaves, stop trusting SAH and split at the median.
    bool forceMedian = depth + ceilLog2((count + kLeafMax - 1) / kLeafMax) + 1 >= (uint32_t)kBvhMaxDepth;

    int bestAxis = -1, bestSplit = -1;
    float bestCost = 1e30f;
    if (!forceMedian) {
      for (int axis = 0; axis < 3; axis++) {
        float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
        if (!(ext > 0.0f)) continue;
        Box bb[kBins];
        uint32_t bc[kBins];
        for (int b = 0; b < kBins; b++) {
          bb[b].reset();
          bc[b] = 0;
        }
        float scale = (float)kBins / ext;
        for (uint32_t k = 0; k < count; k++) {
          uint32_t t = order[first + k];
          int b = (int)((cent[(size_t)t * 3 + axis] - lo) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          bb[b].grow(boxes[t]);
          bc[b]++;
        }
        float rightArea[kBins];
        uint32_t rightCnt[kBins];
        Box acc;
        acc.reset();
        uint32_t cnt = 0;
        for (int b = kBins - 1; b > 0; b--) {
          acc.grow(bb[b]);
          cnt += bc[b];
          rightArea[b] = acc.area();
          rightCnt[b] = cnt;
        }
        acc.reset();
        cnt = 0;
        for (int b = 0; b < kBins - 1; b++) {
          acc.grow(bb[b]);
          cnt += bc[b];
          if (cnt == 0 || rightCnt[b + 1] == 0) continue;
          float cost = acc.area() * (float)cnt + rightArea[b + 1] * (float)rightCnt[b + 1];
          if (cost < bestCost) {
            bestCost = cost;
            bestAxis = axis;
            bestSplit = b;
          }
        }
      }
    }
    uint32_t mid = 0;
    if (bestAxis >= 0) {
      float lo = cb.lo[bestAxis], ext = cb.hi[bestAxis] - cb.lo[bestAxis];
      float scale = (float)kBins / ext;
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
        int b = (int)((cent[(size_t)t * 3 + bestAxis] - lo) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        return b <= bestSplit;
      });
      mid = (uint32_t)(it - (order.begin() + first));
    }
    if (mid == 0 || mid == count) {
      // median split on the widest centroid axis (also the degenerate all-equal case)
      int axis = 0;
      float e0 = cb.hi[0] - cb.lo[0], e1 = cb.hi[1] - cb.lo[1], e2 = cb.hi[2] - cb.lo[2];
      if (e1 > e0 && e1 >= e2) axis = 1;
      if (e2 > e0 && e2 > e1) axis = 2;
      mid = count / 2;
      std::nth_element(order.begin() + first, order.begin() + first + mid, order.begin() + first + count,
                       [&](uint32_t a, uint32_t b) {
                         float ka = cent[(size_t)a * 3 + axis], kb = cent[(size_t)b * 3 + axis];
                         return ka < kb || (ka == kb && a < b);
                       });
    }
    TmpNode l, r;
    l.first = first;
    l.count = mid;
    l.depth = depth + 1;
    r.first = first + mid;
    r.count = count - mid;
    r.depth = depth + 1;
    uint32_t li = (uint32_t)tmp.size();
    tmp.push_back(l);
    tmp.push_back(r);
    tmp[ni].left = (int32_t)li;
    tmp[ni].right = (int32_t)li + 1;
    tmp[ni].count = 0;
    todo.push_back(li + 1);
    todo.push_back(li);
  }

  // Leaf-ordered triangle list.
  out.tris.resize(n);
  for (uint32_t i = 0; i < n; i++) out.tris[i] = recs[order[i]];

  // Flatten: interior nodes in depth-first order (top of the tree first), each carrying its
  // children's padded boxes.
  auto leafRef = [](uint32_t first, uint32_t count) -> int32_t { return -1 - (int32_t)((first << 3) | (count - 1)); };
  auto writeChild = [&](BvhNode& nd, int which, const TmpNode* c, int32_t ref) {
    float lo[3], hi[3];
    if (c) {
      for (int k = 0; k < 3; k++) {
        lo[k] = c->box.lo[k] - pad;
        hi[k] = c->box.hi[k] + pad;
      }
    } else {
      lo[0] = lo[1] = lo[2] = 1e30f;
      hi[0] = hi[1] = hi[2] = -1e30f;
    }
    if (which == 0) {
      nd.lo0[0] = lo[0];
      nd.lo0[1] = lo[1];
      nd.lo0[2] = lo[2];
      nd.hi0x = hi[0];
      nd.hi0yz[0] = hi[1];
      nd.hi0yz[1] = hi[2];
      nd.child0 = ref;
    } else {
      nd.lo1xy[0] = lo[0];
      nd.lo1xy[1] = lo[1];
      nd.lo1z = lo[2];
      nd.hi1[0] = hi[0];
      nd.hi1[1] = hi[1];
      nd.hi1[2] = hi[2];
      nd.child1 = ref;
    }
  };
  if (n == 0) {
    BvhNode nd;
    std::memset(&nd, 0, sizeof(nd));
    writeChild(nd, 0, nullptr, -1);
    writeChild(nd, 1, nullptr, -1);
    out.nodes.push_back(nd);
    return;
  }
  if (tmp[0].left < 0) {  // whole scene is one leaf: wrap it in a root
    BvhNode nd;
    std::memset(&nd, 0, sizeof(nd));
    writeChild(nd, 0, &tmp[0], leafRef(tmp[0].first, tmp[0].count));
    writeChild(nd, 1, nullptr, -1);
    out.nodes.push_back(nd);
    out.sahCost = kCostTri * (float)n;
    return;
  }
  // assign flat indices to interior tmp nodes in DFS preorder
  std::vector<int32_t> flatIndex(tmp.size(), -1);
  std::vector<uint32_t> stack;
  stack.push_back(0);
  uint32_t numInner = 0;
  std::vector<uint32_t> preorder;
  while (!stack.empty()) {
    uint32_t t = stack.back();
    stack.pop_back();
    if (tmp[t].left < 0) continue;
    flatIndex[t] = (int32_t)numInner++;
    preorder.push_back(t);
    stack.push_back((uint32_t)tmp[t].right);
    stack.push_back((uint32_t)tmp[t].left);
  }
  out.nodes.resize(numInner);
  float rootArea = tmp[0].box.area();
  double cost = 0.0;
  for (uint32_t t : preorder) {
    BvhNode nd;
    std::memset(&nd, 0, sizeof(nd));
    const TmpNode& l = tmp[(size_t)tmp[t].left];
    const TmpNode& r = tmp[(size_t)tmp[t].right];
    writeChild(nd, 0, &l, l.left < 0 ? leafRef(l.first, l.count) : flatIndex[(size_t)tmp[t].left]);
    writeChild(nd, 1, &r, r.left < 0 ? leafRef(r.first, r.count) : flatIndex[(size_t)tmp[t].right]);
    out.nodes[(size_t)flatIndex[t]] = nd;
    if (rootArea > 0) {
      cost += kCostTraverse * tmp[t].box.area() / rootArea;
      if (l.left < 0) cost += kCostTri * l.count * l.box.area() / rootArea;
      if (r.left < 0) cost += kCostTri * r.count * r.box.area() / rootArea;
    }
  }
  out.sahCost = (float)cost;
}

}  // namespace bdpt
