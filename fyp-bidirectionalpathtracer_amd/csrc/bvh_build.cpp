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
constexpr int kBinaryMaxDepth = 48;  // depth budget of the intermediate binary tree

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
    if (count <= kLeafMax) continue;

    // Depth budget: once the remaining levels are only just enough for a balanced split of
    // `count` triangles into leaves, stop trusting SAH and split at the median.
    bool forceMedian = depth + ceilLog2((count + kLeafMax - 1) / kLeafMax) + 1 >= (uint32_t)kBinaryMaxDepth;

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

  // ---- collapse the binary tree into four-wide nodes and quantise the child boxes ----------------
  auto leafRef = [](uint32_t first, uint32_t count) -> int32_t { return -1 - (int32_t)((first << 3) | (count - 1)); };
  struct Wide {
    uint32_t src;            // tmp index of the subtree root this node covers
    uint32_t kids[4];        // tmp indices of the (up to 4) children
    int nk;
    uint32_t depth;
  };
  if (n == 0) {
    BvhNode nd;
    std::memset(&nd, 0, sizeof(nd));
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 4; c++) {
        nd.lo[a][c] = 255;
        nd.hi[a][c] = 0;
      }
    for (int c = 0; c < 4; c++) nd.child[c] = -1;
    nd.exp[0] = nd.exp[1] = nd.exp[2] = 127;
    out.nodes.push_back(nd);
    return;
  }
  // Binary height of every subtree: the stack need of a subtree left two-wide is its height, so a
  // node at stack level u may widen to k children only while u + (k-1) + max child height stays
  // within the device stack.  Shallow subtrees (most of the nodes) become four-wide; only the few
  // deep, skinny paths of a SAH tree keep two-wide nodes.
  std::vector<uint16_t> height(tmp.size(), 0);
  for (size_t t = tmp.size(); t-- > 0;)  // children are always created after their parent
    if (tmp[t].left >= 0) height[t] = (uint16_t)(1 + std::max(height[(size_t)tmp[t].left], height[(size_t)tmp[t].right]));
  std::vector<Wide> wide;
  std::vector<std::pair<int32_t, int32_t>> slots;  // per wide node: where its index must be written
  uint32_t wDepth = 0, wStack = 0;
  {
    struct Job {
      uint32_t src, depth, stackAbove;
      int32_t slotNode, slotIdx;
    };
    std::vector<Job> jobs;
    jobs.push_back(Job{0, 0, 0, -1, -1});
    while (!jobs.empty()) {
      Job j = jobs.back();
      jobs.pop_back();
      Wide w;
      w.src = j.src;
      w.depth = j.depth;
      w.nk = 0;
      if (tmp[j.src].left < 0) {  // root is a single leaf
        w.kids[w.nk++] = j.src;
      } else {
        w.kids[w.nk++] = (uint32_t)tmp[j.src].left;
        w.kids[w.nk++] = (uint32_t)tmp[j.src].right;
        while (w.nk < 4) {
          int best = -1;
          float bestArea = -1.0f;
          for (int k = 0; k < w.nk; k++)
            if (tmp[w.kids[k]].left >= 0) {
              float ar = tmp[w.kids[k]].box.area();
              if (ar > bestArea) {
                bestArea = ar;
                best = k;
              }
            }
          if (best < 0) break;
          // stack need if we widen: j.stackAbove + nk (= (nk+1)-1) + tallest remaining child
          const uint32_t t = w.kids[best];
          uint32_t tallest = std::max<uint32_t>(height[(size_t)tmp[t].left], height[(size_t)tmp[t].right]);
          for (int k = 0; k < w.nk; k++)
            if (k != best) tallest = std::max<uint32_t>(tallest, height[w.kids[k]]);
          if (j.stackAbove + (uint32_t)w.nk + tallest > (uint32_t)kBvhMaxStack) break;
          w.kids[best] = (uint32_t)tmp[t].left;
          w.kids[w.nk++] = (uint32_t)tmp[t].right;
        }
      }
      const uint32_t self = (uint32_t)wide.size();
      wide.push_back(w);
      slots.push_back({j.slotNode, j.slotIdx});
      wDepth = std::max(wDepth, j.depth);
      const uint32_t need = j.stackAbove + (uint32_t)(w.nk - 1);
      wStack = std::max(wStack, need);
      for (int k = w.nk - 1; k >= 0; k--)
        if (tmp[w.kids[k]].left >= 0) jobs.push_back(Job{w.kids[k], j.depth + 1, need, (int32_t)self, k});
    }
  }
  out.maxDepth = wDepth;
  out.maxStack = wStack;
  out.nodes.resize(wide.size());
  const float rootArea = tmp[0].box.area();
  double cost = 0.0;
  for (size_t wi = 0; wi < wide.size(); wi++) {
    const Wide& w = wide[wi];
    BvhNode nd;
    std::memset(&nd, 0, sizeof(nd));
    // node box = union of the padded child boxes
    float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
    float clo[4][3], chi[4][3];
    for (int k = 0; k < w.nk; k++)
      for (int a = 0; a < 3; a++) {
        clo[k][a] = tmp[w.kids[k]].box.lo[a] - pad;
        chi[k][a] = tmp[w.kids[k]].box.hi[a] + pad;
        blo[a] = std::min(blo[a], clo[k][a]);
        bhi[a] = std::max(bhi[a], chi[k][a]);
      }
    for (int a = 0; a < 3; a++) {
      nd.origin[a] = blo[a];
      // smallest power of two s with 254*s >= extent (one code of headroom for outward rounding)
      const float ext = std::max(bhi[a] - blo[a], 1e-30f);
      int e = 0;
      (void)std::frexp(ext / 254.0f, &e);  // ext/254 = m * 2^e, m in [0.5,1) -> s = 2^e >= ext/254
      int biased = e + 127;
      if (biased < 1) biased = 1;
      if (biased > 254) biased = 254;
      nd.exp[a] = (uint8_t)biased;
      for (int k = 0; k < 4; k++) {
        if (k >= w.nk) {
          nd.lo[a][k] = 255;
          nd.hi[a][k] = 0;
          continue;
        }
        union {
          uint32_t u;
          float f;
        } sc;
        sc.u = (uint32_t)biased << 23;
        int ql = (int)std::floor((clo[k][a] - blo[a]) / sc.f);
        ql = std::min(std::max(ql, 0), 255);
        while (ql > 0 && blo[a] + (float)ql * sc.f > clo[k][a]) ql--;
        int qh = (int)std::ceil((chi[k][a] - blo[a]) / sc.f);
        qh = std::min(std::max(qh, 0), 255);
        while (qh < 255 && blo[a] + (float)qh * sc.f < chi[k][a]) qh++;
        nd.lo[a][k] = (uint8_t)ql;
        nd.hi[a][k] = (uint8_t)qh;
      }
    }
    nd.numChildren = (uint8_t)w.nk;
    for (int k = 0; k < 4; k++) nd.child[k] = -1;
    for (int k = 0; k < w.nk; k++) {
      const TmpNode& c = tmp[w.kids[k]];
      if (c.left < 0) nd.child[k] = leafRef(c.first, c.count);  // interior refs are patched below
      if (rootArea > 0) cost += (c.left < 0 ? kCostTri * c.count : kCostTraverse) * c.box.area() / rootArea;
    }
    out.nodes[wi] = nd;
  }
  for (size_t wi = 1; wi < wide.size(); wi++) out.nodes[(size_t)slots[wi].first].child[slots[wi].second] = (int32_t)wi;
  out.sahCost = (float)cost + kCostTraverse;
}

}  // namespace bdpt
