// bvh_build.cpp — binned-SAH top-down builder producing the flat 64-byte-node layout of bvh.h.
//
// Threaded (std::thread): the top of the tree is split by one thread with the binning of large nodes shared
// among all, then the subtrees below a grain size are built concurrently on disjoint ranges of the
// triangle order; the per-triangle and per-node passes run as parallel loops.  The result does not depend on
// the thread count, bit for bit: bin bounds and counts are order-independent (min / max / integer sums), every
// partition is the same serial std::partition on the same range, the four-wide collapse only looks at the tree's
// shape, and the one floating-point sum (the SAH cost) is taken in node order by one thread.
#include "bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <chrono>
#include <mutex>
#include <thread>

#include <sched.h>

namespace bdpt {
namespace {

using Box = BvhBox;            // bvh.h: shared with the device-side tree builder
using TmpNode = BvhBuildNode;  // the binary tree: children always behind their parent
inline TmpNode makeTmpNode(uint32_t first, uint32_t count, uint32_t depth) {
  TmpNode n;
  n.box.reset();
  n.left = n.right = -1;
  n.first = first;
  n.count = count;
  n.depth = depth;
  return n;
}

#ifndef BDPT_SAH_BINS
#define BDPT_SAH_BINS 16
#endif
#ifndef BDPT_LEAF_MAX
#define BDPT_LEAF_MAX 2
#endif
// Default budgets.  Opaque triangles: none — measured on the bench atrium (tools/bvh_eval.py, profiles/README.md r3)
// pre-splitting by priority makes an evenly tessellated scene's tree worse at every setting tried (SAH cost 31.3 ->
// 31.5-35.2, +0-14 % node visits); the knob (also the environment variable BDPT_SPLIT_BUDGET) stays for scenes that
// mix huge and tiny triangles.  Non-opaque triangles: four extra references each — the 10 M-triangle
// courtyard's closest-hit rays go from 76 node visits + 55 triangle tests to 40 + 12 and its 4K depth-16 frame from 905 ms to
// 373 ms (budgets 2 / 4 / 8: 435 / 373 / 361 ms at 7 / 13 / 16 s of scene set-up).
#ifndef BDPT_SPLIT_BUDGET
#define BDPT_SPLIT_BUDGET 0.0f
#endif
#ifndef BDPT_SPLIT_BUDGET_ALPHA
#define BDPT_SPLIT_BUDGET_ALPHA 4.0f
#endif
constexpr int kBins = kBvhBins;
constexpr uint32_t kStablePartitionMin = 1u << 16;  // nodes of at least this many references are partitioned stably, in parallel
constexpr size_t kPartitionChunk = 1u << 14;
constexpr uint32_t kCollapseGrain = 1u << 15;  // binary subtrees of at most this many nodes are collapsed to four-wide nodes by one worker each
// Leaves hold at most two triangles.  Measured on the bench frame (profiles/README.md r2): leaves of <= 1 / 2 / 3 / 4 / 8
// triangles give 23.6 / 19.1 / 19.5 / 20.1 / 23.0 ms per frame — a triangle test costs half a node visit and leaf runs of
// different lengths idle lanes, while one-triangle leaves double the node array past the 4 MiB L2 of an XCD.
constexpr uint32_t kLeafMax = kBvhLeafMax;
constexpr float kCostTraverse = 1.0f, kCostTri = 1.0f;
constexpr int kBinaryMaxDepth = kBvhBinaryMaxDepth;  // depth budget of the intermediate binary tree: a two-wide path stacks one reference per level, so the device stack bounds it (a budget of 48 let a 10 M-triangle scene of overlapping cards through that bdpt_set_scene then had to refuse)

inline uint32_t ceilLog2(uint32_t x) {
  uint32_t l = 0;
  while ((1u << l) < x) l++;
  return l;
}

// [0, n) in `threads` contiguous chunks, one std::thread each (the caller's thread takes chunk 0)
template <class F>
void parallelFor(size_t n, int threads, const F& f) {
  if (threads <= 1 || n < 4096) {
    f((size_t)0, n, 0);
    return;
  }
  WorkerScope pool;
  const size_t chunk = (n + (size_t)threads - 1) / (size_t)threads;
  for (int t = 1; t < threads; t++) {
    const size_t a = std::min(n, chunk * (size_t)t), b = std::min(n, a + chunk);
    if (a < b) pool.spawn([&f, a, b, t] { f(a, b, t); });
  }
  f((size_t)0, std::min(n, chunk), 0);
  pool.join();
}

// [0, n) in chunks of `chunk` items handed out through an atomic counter (work per item may differ by orders of
// magnitude: a split triangle against an untouched one); f(chunkIndex, first, last).  What a chunk produces must only
// depend on its index, so that results assembled in chunk order do not depend on the thread count.
template <class F>
void parallelChunks(size_t n, int threads, size_t chunk, const F& f) {
  const size_t numChunks = (n + chunk - 1) / chunk;
  if (threads <= 1 || numChunks <= 1) {
    for (size_t c = 0; c < numChunks; c++) f(c, c * chunk, std::min(n, (c + 1) * chunk));
    return;
  }
  std::atomic<size_t> next{0};
  auto worker = [&] {
    for (;;) {
      const size_t c = next.fetch_add(1);
      if (c >= numChunks) return;
      f(c, c * chunk, std::min(n, (c + 1) * chunk));
    }
  };
  WorkerScope pool;
  for (int t = 1; t < threads; t++) pool.spawn(worker);
  worker();
  pool.join();
}

// What the tree is built over: one 40-byte record per reference — the box of the piece it stands for, the box's
// centre, and its index in the reference list.  The records THEMSELVES are permuted as nodes are partitioned (round 4;
// before, an index array was permuted and every scan of a node gathered boxes and centres through it): every pass over
// a node — bounds, binning, the partition — then streams a contiguous range instead of gathering 36 bytes per element
// from a gigabyte of boxes, which is what bounded the builder on 16 host threads.
using Ref = BvhBuildRef;

struct BuildData {
  BigVec<Ref>& refs;     // permuted in place: a node owns a contiguous range
  BigVec<Ref>& scratch;  // as large as refs: the stable partition of large nodes scatters through it
};

struct Bins {
  Box bb[3][kBins];
  uint32_t bc[3][kBins];
  void reset() {
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < kBins; b++) {
        bb[a][b].reset();
        bc[a][b] = 0;
      }
  }
};

// Bounds of a node and the position of its split (0 = leaf).  `threads` > 1 shares the two O(count) scans.
uint32_t splitNode(const BuildData& B, uint32_t first, uint32_t count, uint32_t depth, int threads, Box& nodeBox) {
  Ref* const base = B.refs.data() + first;
  Box nb, cb;
  nb.reset();
  cb.reset();
  if (threads > 1 && count >= (1u << 16)) {
    std::vector<Box> pn((size_t)threads), pc((size_t)threads);
    for (int t = 0; t < threads; t++) {
      pn[(size_t)t].reset();
      pc[(size_t)t].reset();
    }
    parallelFor(count, threads, [&](size_t a, size_t b, int t) {
      Box n0, c0;
      n0.reset();
      c0.reset();
      for (size_t k = a; k < b; k++) {
        n0.grow(base[k].box);
        c0.grow(base[k].cent);
      }
      pn[(size_t)t] = n0;
      pc[(size_t)t] = c0;
    });
    for (int t = 0; t < threads; t++) {
      nb.grow(pn[(size_t)t]);
      cb.grow(pc[(size_t)t]);
    }
  } else {
    for (uint32_t k = 0; k < count; k++) {
      nb.grow(base[k].box);
      cb.grow(base[k].cent);
    }
  }
  nodeBox = nb;
  if (count <= kLeafMax) return 0;

  // Depth budget: once the remaining levels are only just enough for a balanced split of
  // `count` triangles into leaves, stop trusting SAH and split at the median.
  const bool forceMedian = depth + ceilLog2((count + kLeafMax - 1) / kLeafMax) + 1 >= (uint32_t)kBinaryMaxDepth;

  int bestAxis = -1, bestSplit = -1;
  float bestCost = 1e30f;
  if (!forceMedian) {
    float lo[3], scale[3];
    bool use[3];
    for (int axis = 0; axis < 3; axis++) {
      const float ext = cb.hi[axis] - cb.lo[axis];
      lo[axis] = cb.lo[axis];
      use[axis] = ext > 0.0f;
      scale[axis] = use[axis] ? (float)kBins / ext : 0.0f;
    }
    auto binRange = [&](size_t a, size_t b, Bins& bins) {
      for (size_t k = a; k < b; k++) {
        const Ref& r = base[k];
        for (int axis = 0; axis < 3; axis++) {
          if (!use[axis]) continue;
          int bi = (int)((r.cent[axis] - lo[axis]) * scale[axis]);
          bi = std::min(std::max(bi, 0), kBins - 1);
          bins.bb[axis][bi].grow(r.box);
          bins.bc[axis][bi]++;
        }
      }
    };
    Bins bins;
    bins.reset();
    if (threads > 1 && count >= (1u << 16)) {
      std::vector<Bins> part((size_t)threads);
      for (Bins& pb : part) pb.reset();
      parallelFor(count, threads, [&](size_t a, size_t b, int t) { binRange(a, b, part[(size_t)t]); });
      for (const Bins& pb : part)
        for (int axis = 0; axis < 3; axis++)
          for (int b = 0; b < kBins; b++) {
            bins.bb[axis][b].grow(pb.bb[axis][b]);
            bins.bc[axis][b] += pb.bc[axis][b];
          }
    } else {
      binRange(0, count, bins);
    }
    for (int axis = 0; axis < 3; axis++) {
      if (!use[axis]) continue;
      const Box* bb = bins.bb[axis];
      const uint32_t* bc = bins.bc[axis];
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
        const float cost = acc.area() * (float)cnt + rightArea[b + 1] * (float)rightCnt[b + 1];
        if (cost < bestCost) {
          bestCost = cost;
          bestAxis = axis;
          bestSplit = b;
        }
      }
    }
  }
  // A STABLE partition of the node's range — lefts in their order, then rights in theirs — is the one way references
  // move, whatever decides who goes left.  Large nodes (the top of the tree: a few dozen nodes that together touch every
  // reference several times) do it in three parallel passes: count the lefts of every chunk, scatter the records into
  // the scratch array at offsets from those counts, copy back.  The chunks are fixed-size, not per-thread, and the rule is
  // chosen by the node's SIZE, so the permutation — and with it the tree — does not depend on the thread count; small
  // nodes do the same by one thread.  One rule for every node size: a builder that partitions in parallel — the passes
  // here, or a device (bvh_device.hip: flags, scan, scatter) — produces the very same permutation.
  auto partitionStable = [&](auto&& goesLeft) -> uint32_t {
    Ref* const tmp = B.scratch.data() + first;
    if (count >= kStablePartitionMin) {
      const size_t nChunks = ((size_t)count + kPartitionChunk - 1) / kPartitionChunk;
      std::vector<uint32_t> lefts(nChunks + 1, 0);
      parallelChunks(count, threads, kPartitionChunk, [&](size_t ci, size_t a, size_t b) {
        uint32_t n = 0;
        for (size_t k = a; k < b; k++) n += goesLeft(base[k]) ? 1u : 0u;
        lefts[ci + 1] = n;
      });
      for (size_t ci = 0; ci < nChunks; ci++) lefts[ci + 1] += lefts[ci];
      const uint32_t nLeft = lefts[nChunks];
      parallelChunks(count, threads, kPartitionChunk, [&](size_t ci, size_t a, size_t b) {
        uint32_t l = lefts[ci], r = nLeft + ((uint32_t)a - lefts[ci]);
        for (size_t k = a; k < b; k++) {
          if (goesLeft(base[k]))
            tmp[l++] = base[k];
          else
            tmp[r++] = base[k];
        }
      });
      parallelChunks(count, threads, kPartitionChunk, [&](size_t, size_t a, size_t b) { std::memcpy(base + a, tmp + a, (b - a) * sizeof(Ref)); });
      return nLeft;
    }
    uint32_t l = 0, r = 0;
    for (uint32_t k = 0; k < count; k++) {
      if (goesLeft(base[k]))
        base[l++] = base[k];  // (l <= k: never overwrites an element not yet read)
      else
        tmp[r++] = base[k];
    }
    std::memcpy(base + l, tmp, (size_t)r * sizeof(Ref));
    return l;
  };
  uint32_t mid = 0;
  if (bestAxis >= 0) {
    const float lo = cb.lo[bestAxis], ext = cb.hi[bestAxis] - cb.lo[bestAxis];
    const float scale = (float)kBins / ext;
    mid = partitionStable([&](const Ref& r) {
      int b = (int)((r.cent[bestAxis] - lo) * scale);
      b = std::min(std::max(b, 0), kBins - 1);
      return b <= bestSplit;
    });
  }
  if (mid == 0 || mid == count) {
    // Median split on the widest centroid axis (also the degenerate all-equal case): the count / 2 references that come
    // first by (centroid, reference id) go left — found by SELECTING the pivot, not by sorting — and the same stable
    // partition moves them: no order inside the halves has to be defined beyond the one they already have, and a device
    // does it with a radix select and the partition machinery it has anyway.
    int axis = 0;
    const float e0 = cb.hi[0] - cb.lo[0], e1 = cb.hi[1] - cb.lo[1], e2 = cb.hi[2] - cb.lo[2];
    if (e1 > e0 && e1 >= e2) axis = 1;
    if (e2 > e0 && e2 > e1) axis = 2;
    std::vector<std::pair<float, uint32_t>> keys(count);
    for (uint32_t k = 0; k < count; k++) keys[k] = {base[k].cent[axis], base[k].id};
    std::nth_element(keys.begin(), keys.begin() + count / 2, keys.end());
    const std::pair<float, uint32_t> pivot = keys[count / 2];
    mid = partitionStable([&](const Ref& r) { return std::pair<float, uint32_t>(r.cent[axis], r.id) < pivot; });
  }
  return mid;
}

// Whole subtree under nodes[root] (its first / count / depth already set), depth first, appended to `nodes`.
// Children are always created after their parent.
void buildSubtree(const BuildData& B, BigVec<TmpNode>& nodes, uint32_t root, uint32_t splitBelow, int threads,
                  std::vector<uint32_t>* deferred) {
  std::vector<uint32_t> todo{root};
  while (!todo.empty()) {
    const uint32_t ni = todo.back();
    todo.pop_back();
    const uint32_t first = nodes[ni].first, count = nodes[ni].count, depth = nodes[ni].depth;
    if (deferred && ni != root && count <= splitBelow) {  // small enough: some thread builds it later
      deferred->push_back(ni);
      continue;
    }
    Box nb;
    const uint32_t mid = splitNode(B, first, count, depth, threads, nb);
    nodes[ni].box = nb;
    if (mid == 0) continue;
    const TmpNode l = makeTmpNode(first, mid, depth + 1), r = makeTmpNode(first + mid, count - mid, depth + 1);
    const uint32_t li = (uint32_t)nodes.size();
    nodes.push_back(l);
    nodes.push_back(r);
    nodes[ni].left = (int32_t)li;
    nodes[ni].right = (int32_t)li + 1;
    nodes[ni].count = 0;
    todo.push_back(li + 1);
    todo.push_back(li);
  }
}


// ------------------------------------------------------------------------------------------------
// References.  The tree is built over REFERENCES, not triangles: a reference is (triangle, box of the piece of it
// the reference stands for).  One reference per triangle gives the classic object-split tree; spatial
// pre-splitting cuts the box of a triangle at spatial-median planes of the scene box and gives every piece its own
// reference, so that large or diagonal triangles (and, above all, the overlapping alpha-masked cards of foliage)
// stop inflating every box above them.  The leaf still intersects the WHOLE triangle (device_trace.hpp triGeom): a
// hit found through any reference of a triangle is that triangle's one hit, and the closest-hit tie rule (lowest
// primitive index; the same primitive never replaces itself) makes duplicates harmless.
//
// How many splits a triangle gets follows Karras & Aila, "Fast parallel construction of high-quality bounding
// volume hierarchies" (HPG 2013), section 4.3: priority p = (2^-level * (A_box - A_ideal))^(1/3), where level is
// that of the most important spatial-median plane cutting the box and A_ideal = |e1 x e2|_1 is the box area the
// triangle would reach if split without end; split counts s_t = floor(D p_t) with D chosen so that their sum meets
// the budget; a piece hands its remaining splits to its two halves in proportion to their extents.
//
// Pieces are convex polygons in the triangle's barycentric plane (vertex = (bu, bv), P = v0 + bu e1 + bv e2),
// clipped in double precision; a piece's box is rounded outwards to float (the builder's pad covers the fp32
// rounding of the device's triangle test, as it does for whole triangles).  For non-opaque triangles the caller's
// BvhRefClipper shrinks a piece to where the alpha test can pass, or drops it.
// ------------------------------------------------------------------------------------------------
struct Piece {
  double b[kBvhPolyMax][2];
  int n;
  Box box;
  uint32_t splits;
};

inline float floatDown(double x) {
  float f = (float)x;
  if ((double)f > x) f = std::nextafterf(f, -INFINITY);
  return f;
}
inline float floatUp(double x) {
  float f = (float)x;
  if ((double)f < x) f = std::nextafterf(f, INFINITY);
  return f;
}

Box polyBox(const BvhTri& r, const double (*b)[2], int n) {
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int k = 0; k < n; k++)
    for (int a = 0; a < 3; a++) {
      const double p = (double)r.v0[a] + b[k][0] * (double)r.e1[a] + b[k][1] * (double)r.e2[a];
      lo[a] = std::min(lo[a], p);
      hi[a] = std::max(hi[a], p);
    }
  Box bx;
  for (int a = 0; a < 3; a++) {
    bx.lo[a] = floatDown(lo[a]);
    bx.hi[a] = floatUp(hi[a]);
  }
  return bx;
}

// Sutherland-Hodgman against the closed half-plane A + B bu + C bv <= 0
int clipHalfPlane(const double (*in)[2], int n, double A, double B, double C, double (*out)[2]) {
  int m = 0;
  for (int k = 0; k < n; k++) {
    const double* p = in[k];
    const double* q = in[(k + 1) % n];
    const double fp = A + B * p[0] + C * p[1], fq = A + B * q[0] + C * q[1];
    if (fp <= 0.0 && m < kBvhPolyMax) {
      out[m][0] = p[0];
      out[m][1] = p[1];
      m++;
    }
    if (((fp < 0.0 && fq > 0.0) || (fp > 0.0 && fq < 0.0)) && m < kBvhPolyMax) {
      const double t = fp / (fp - fq);
      out[m][0] = p[0] + t * (q[0] - p[0]);
      out[m][1] = p[1] + t * (q[1] - p[1]);
      m++;
    }
  }
  return m;
}

struct SplitGrid {  // spatial-median planes of the scene box on a 2^30 grid per axis
  double lo[3], ext[3];
  // most important plane strictly inside [a, b] on `axis`: its importance (bit position, higher = nearer the root) or -1
  int plane(int axis, float a, float b, double& coord) const {
    if (!(ext[axis] > 0.0) || !(b > a)) return -1;
    const double s = 1073741824.0 / ext[axis];
    double ua = std::floor(((double)a - lo[axis]) * s), ub = std::floor(((double)b - lo[axis]) * s);
    ua = std::min(std::max(ua, 0.0), 1073741823.0);
    ub = std::min(std::max(ub, 0.0), 1073741823.0);
    const uint32_t ia = (uint32_t)ua, ib = (uint32_t)ub;
    if (ia == ib) return -1;
    const uint32_t diff = ia ^ ib;
    const int h = 31 - __builtin_clz(diff);
    const uint32_t pl = (ib >> h) << h;
    coord = lo[axis] + (double)pl / s;
    if (!(coord > (double)a && coord < (double)b)) return -1;  // (rounding at the ends of the interval)
    return h;
  }
  int dominant(const Box& bx, int& axis, double& coord) const {
    int best = -1;
    float bestExt = -1.0f;
    for (int a = 0; a < 3; a++) {
      double c;
      const int h = plane(a, bx.lo[a], bx.hi[a], c);
      const float e = bx.hi[a] - bx.lo[a];
      if (h > best || (h == best && h >= 0 && e > bestExt)) {
        best = h;
        bestExt = e;
        axis = a;
        coord = c;
      }
    }
    return best;
  }
};

double splitPriority(const SplitGrid& G, const BvhTri& r, const Box& bx, double polyShare) {
  int axis = 0;
  double c = 0;
  const int h = G.dominant(bx, axis, c);
  if (h < 0) return 0.0;
  const double cx = (double)r.e1[1] * r.e2[2] - (double)r.e1[2] * r.e2[1], cy = (double)r.e1[2] * r.e2[0] - (double)r.e1[0] * r.e2[2],
               cz = (double)r.e1[0] * r.e2[1] - (double)r.e1[1] * r.e2[0];
  const double ideal = (std::fabs(cx) + std::fabs(cy) + std::fabs(cz)) * polyShare;
  const double dx = (double)bx.hi[0] - bx.lo[0], dy = (double)bx.hi[1] - bx.lo[1], dz = (double)bx.hi[2] - bx.lo[2];
  const double gain = 2.0 * (dx * dy + dy * dz + dz * dx) - ideal;
  if (!(gain > 0.0)) return 0.0;
  return bvhCbrt(std::ldexp(gain, h - 30));
}

double polyArea2(const double (*b)[2], int n) {  // twice the area in barycentric units (the whole triangle: 1)
  double s = 0;
  for (int k = 0; k < n; k++) {
    const double* p = b[k];
    const double* q = b[(k + 1) % n];
    s += p[0] * q[1] - q[0] * p[1];
  }
  return std::fabs(s);
}

inline Box intersectBox(const Box& a, const Box& b) {
  Box r;
  for (int k = 0; k < 3; k++) {
    r.lo[k] = std::max(a.lo[k], b.lo[k]);
    r.hi[k] = std::min(a.hi[k], b.hi[k]);
    if (r.hi[k] < r.lo[k]) r.hi[k] = r.lo[k];  // (outward rounding of two disjoint-by-an-ulp intervals)
  }
  return r;
}

struct RefOut {
  std::vector<Box> boxes;
  std::vector<uint32_t> tri;
};

// All references of one triangle, appended to `out` in a fixed order.  Returns the number appended.
uint32_t splitTriangle(const SplitGrid& G, const BvhTri& r, uint32_t t, const Piece& whole, const BvhRefClipper* clipper, RefOut& out) {
  std::vector<Piece> todo{whole};
  uint32_t made = 0;
  while (!todo.empty()) {
    Piece pc = todo.back();
    todo.pop_back();
    for (int guard = 0;; guard++) {
      int axis = 0;
      double c = 0;
      if (pc.splits == 0 || guard > 96 || pc.n + 2 > kBvhPolyMax || G.dominant(pc.box, axis, c) < 0) {
        out.boxes.push_back(pc.box);
        out.tri.push_back(t);
        made++;
        break;
      }
      const double A = (double)r.v0[axis] - c, B = (double)r.e1[axis], C = (double)r.e2[axis];
      Piece lo, hi;
      lo.n = clipHalfPlane(pc.b, pc.n, A, B, C, lo.b);
      hi.n = clipHalfPlane(pc.b, pc.n, -A, -B, -C, hi.b);
      bool haveLo = lo.n >= 3 && polyArea2(lo.b, lo.n) > 0.0, haveHi = hi.n >= 3 && polyArea2(hi.b, hi.n) > 0.0;
      if (clipper) {
        if (haveLo) haveLo = clipper->clip(t, lo.b, lo.n) && lo.n >= 3;
        if (haveHi) haveHi = clipper->clip(t, hi.b, hi.n) && hi.n >= 3;
      }
      if (haveLo) lo.box = intersectBox(polyBox(r, lo.b, lo.n), pc.box);
      if (haveHi) hi.box = intersectBox(polyBox(r, hi.b, hi.n), pc.box);
      if (!haveLo && !haveHi) {
        if (!clipper) {  // (a sliver the clip lost to rounding: keep the piece as it was)
          out.boxes.push_back(pc.box);
          out.tri.push_back(t);
          made++;
        }
        break;
      }
      if (!haveLo || !haveHi) {  // the polygon lies on one side of the plane although its box straddles it: shrink and go on
        const uint32_t s = pc.splits;
        pc = haveLo ? lo : hi;
        pc.splits = s;
        continue;
      }
      const uint32_t rest = pc.splits - 1;
      const double wl = ((double)lo.box.hi[0] - lo.box.lo[0]) + ((double)lo.box.hi[1] - lo.box.lo[1]) + ((double)lo.box.hi[2] - lo.box.lo[2]);
      const double wh = ((double)hi.box.hi[0] - hi.box.lo[0]) + ((double)hi.box.hi[1] - hi.box.lo[1]) + ((double)hi.box.hi[2] - hi.box.lo[2]);
      uint32_t sl = (wl + wh > 0.0) ? (uint32_t)std::floor((double)rest * wl / (wl + wh) + 0.5) : rest / 2;
      if (sl > rest) sl = rest;
      lo.splits = sl;
      hi.splits = rest - sl;
      todo.push_back(hi);
      pc = lo;
      guard = 0;
    }
  }
  return made;
}

}  // namespace

static BvhTreeBuilder gDefaultTreeBuilder = nullptr;
static void* gDefaultTreeBuilderUser = nullptr;
void bvhSetDefaultTreeBuilder(BvhTreeBuilder f, void* user) {
  gDefaultTreeBuilder = f;
  gDefaultTreeBuilderUser = user;
}

int bvhBuildThreads() {
  if (const char* e = std::getenv("BDPT_BUILD_THREADS")) {
    const int v = std::atoi(e);
    if (v >= 1) return std::min(v, 256);
  }
  int n = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // container CPU quota: "<quota> <period>" or "max <period>"
    char q[64];
    double period = 0;
    if (std::fscanf(f, "%63s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
      n = std::min(n, std::max(1, (int)(std::atof(q) / period + 0.5)));
    std::fclose(f);
  }
  return std::max(1, std::min(n, 64));
}

void buildBvh(const float* positions, const uint32_t* indices, uint32_t n, const uint32_t* triFlags, Bvh& out, int threads,
              const uint32_t* triAux) {
  BvhBuildOptions opt;
  opt.threads = threads;
  buildBvh(positions, indices, n, triFlags, out, opt, triAux);
}

void buildBvh(const float* positions, const uint32_t* indices, uint32_t nTris, const uint32_t* triFlags, Bvh& out,
              const BvhBuildOptions& opt, const uint32_t* triAux) {
  int threads = opt.threads;
  if (threads <= 0) threads = bvhBuildThreads();
  const bool verbose = std::getenv("BDPT_BUILD_VERBOSE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto tStart = now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    auto t = now();
    std::fprintf(stderr, "[bvh] %-12s %.3f s\n", what, std::chrono::duration<double>(t - tStart).count());
    tStart = t;
  };
  out.nodes.clear();
  out.tris.clear();
  out.refBox.clear();
  out.maxDepth = 0;
  out.sahCost = 0.0f;
  out.numDropped = 0;
  out.numNodes = 0;
  out.numRefs = 0;
  out.deviceRecs = nullptr;
  out.deviceNumRecs = 0;

  auto budgetDefault = [](const char* env, float dflt) {
    if (const char* e = std::getenv(env)) {
      const float v = (float)std::atof(e);
      if (v >= 0.0f && v <= 64.0f) return v;
    }
    return dflt;
  };
  const float budgetOpaque = opt.splitBudget >= 0.0f ? opt.splitBudget : budgetDefault("BDPT_SPLIT_BUDGET", (float)BDPT_SPLIT_BUDGET);
  const float budgetAlpha = opt.splitBudgetAlpha >= 0.0f ? opt.splitBudgetAlpha : budgetDefault("BDPT_SPLIT_BUDGET_ALPHA", (float)BDPT_SPLIT_BUDGET_ALPHA);
  // (a plugged-in reference maker — bdpt_set_scene: the device — may do all of the reference stage, pass 1 and the split counts, itself)
  const BvhRefMaker refMaker = (opt.treeBuilder && opt.packer) ? opt.refMaker : nullptr;
  const bool decideElsewhere = refMaker && opt.prioritiesInRefMaker;
  // ... and then makes the triangle records and their boxes where it works, from positions + indices (BvhRefInput), and
  // the packer of the same pipeline reads them there: this side only needs the scene's box, not 72 bytes per triangle
  // written and page-faulted in (0.05-0.15 s of a 10 M-triangle bdpt_set_scene).
  const bool recsElsewhere = decideElsewhere && opt.collapseInPacker && opt.numVertices != 0 && !(budgetOpaque > 0.0f) &&
                             std::getenv("BDPT_UPLOAD_TRI_RECS") == nullptr;

  // Triangle records exactly as the device intersects them: the "actual" triangle is
  // (v0, v0+e1, v0+e2) in fp32, so bounds are taken from those points.
  BigVec<BvhTri> recs(recsElsewhere ? 0 : nTris);  // (every field of every record is written by the loop below)
  BigVec<Box> triBox(recsElsewhere ? 0 : nTris);
  Box scene;
  scene.reset();
  {
    std::vector<Box> part((size_t)threads);
    for (Box& b : part) b.reset();
    parallelFor(nTris, threads, [&](size_t t0, size_t t1, int th) {
      Box acc;
      acc.reset();
      for (size_t t = t0; t < t1; t++) {
        const float* a = positions + (size_t)indices[t * 3] * 3;
        const float* b = positions + (size_t)indices[t * 3 + 1] * 3;
        const float* c = positions + (size_t)indices[t * 3 + 2] * 3;
        BvhTri r;
        float p1[3], p2[3];
        for (int k = 0; k < 3; k++) {
          r.v0[k] = a[k];
          r.e1[k] = b[k] - a[k];
          r.e2[k] = c[k] - a[k];
          p1[k] = r.v0[k] + r.e1[k];
          p2[k] = r.v0[k] + r.e2[k];
        }
        r.prim = (uint32_t)t;
        r.flags = triFlags ? triFlags[t] : 0u;
        r.aux = triAux ? triAux[t] : 0u;
        Box bx;
        bx.reset();
        bx.grow(r.v0);
        bx.grow(p1);
        bx.grow(p2);
        bx.grow(b);
        bx.grow(c);
        if (!recsElsewhere) {
          recs[t] = r;
          triBox[t] = bx;
        }
        acc.grow(bx);
      }
      part[(size_t)th].grow(acc);
    });
    for (const Box& b : part) scene.grow(b);
  }
  float diag = 0.0f;
  if (nTris) {
    float dx = scene.hi[0] - scene.lo[0], dy = scene.hi[1] - scene.lo[1], dz = scene.hi[2] - scene.lo[2];
    diag = std::sqrt(dx * dx + dy * dy + dz * dz);
  }
  // Slab tests run in fp32 on boxes that must never reject a hit the triangle test accepts:
  // pad every box by a small fraction of the scene diagonal (covers rounding in both tests).
  const float pad = 2e-5f * diag + 1e-30f;

  // ---- references (see "References" above): the whole triangle, shrunk by the clipper where it is non-opaque,
  // then split s_t times
  SplitGrid G;
  for (int a = 0; a < 3; a++) {
    G.lo[a] = nTris ? (double)scene.lo[a] : 0.0;
    G.ext[a] = nTris ? (double)scene.hi[a] - (double)scene.lo[a] : 0.0;
  }
  float outlierArea = 0.0f;  // opaque triangles below this box area are never split
  if (budgetOpaque > 0.0f && nTris) {
    std::vector<float> areas;
    areas.reserve(nTris);
    for (uint32_t t = 0; t < nTris; t++)
      if (!(recs[t].flags & kTriNonOpaque)) areas.push_back(triBox[t].area());
    if (!areas.empty()) {
      std::nth_element(areas.begin(), areas.begin() + areas.size() / 2, areas.end());
      outlierArea = (float)BDPT_SPLIT_OUTLIER * areas[areas.size() / 2];
    }
  }
  const size_t nHere = decideElsewhere ? 0 : nTris;
  // (sized without being touched, filled side by side: 130 MB of one-thread value-initialisation at 10 M triangles otherwise)
  BigVec<double> prio(nHere);
  BigVec<float> capOf(nHere);    // splits a triangle may get at most
  BigVec<uint8_t> state(nHere);  // 0 = plain reference (triBox), 1 = shrunk by the clipper, 2 = dropped
  BigVec<uint32_t> splits(nHere);
  parallelFor(nHere, threads, [&](size_t t0, size_t t1, int) {
    for (size_t t = t0; t < t1; t++) {
      prio[t] = 0.0;
      capOf[t] = (float)BDPT_SPLIT_MAX_PER_TRI;
      state[t] = 0;
      splits[t] = 0;
    }
  });
  const bool anySplit = budgetOpaque > 0.0f || budgetAlpha > 0.0f;
  // the whole triangle as a piece, shrunk by the clipper where it is non-opaque; false: nothing of it can be hit.
  // (Recomputed where it is needed again instead of kept: a piece is ~400 bytes and a scene may hold millions.)
  auto wholePiece = [&](size_t t, Piece& pc, bool& shrunk) {
    const BvhTri& r = recs[t];
    pc.n = 3;
    pc.b[0][0] = 0.0;
    pc.b[0][1] = 0.0;
    pc.b[1][0] = 1.0;
    pc.b[1][1] = 0.0;
    pc.b[2][0] = 0.0;
    pc.b[2][1] = 1.0;
    pc.splits = 0;
    pc.box = triBox[t];
    shrunk = false;
    if (!((r.flags & kTriNonOpaque) != 0 && opt.clipper != nullptr)) return true;
    if (!opt.clipper->clip((uint32_t)t, pc.b, pc.n) || pc.n < 3) return false;
    shrunk = !(pc.n == 3 && pc.b[0][0] == 0.0 && pc.b[0][1] == 0.0 && pc.b[1][0] == 1.0 && pc.b[1][1] == 0.0 && pc.b[2][0] == 0.0 && pc.b[2][1] == 1.0);
    if (shrunk) pc.box = intersectBox(polyBox(r, pc.b, pc.n), triBox[t]);
    return true;
  };
  // pass 1: what the clipper leaves of every non-opaque triangle, and every triangle's priority
  constexpr size_t kRefChunk = 8192;
  parallelChunks(nHere, threads, kRefChunk, [&](size_t, size_t t0, size_t t1) {
    for (size_t t = t0; t < t1; t++) {
      const BvhTri& r = recs[t];
      Piece pc;
      bool shrunk = false;
      if (!wholePiece(t, pc, shrunk)) {
        state[t] = 2;
        continue;
      }
      state[t] = shrunk ? 1 : 0;
      const float budget = (r.flags & kTriNonOpaque) ? budgetAlpha : budgetOpaque;
      if (budget > 0.0f && ((r.flags & kTriNonOpaque) || pc.box.area() >= outlierArea)) {
        prio[t] = splitPriority(G, r, pc.box, shrunk ? polyArea2(pc.b, pc.n) : 1.0);
        // an opaque outlier is cut down to about the size of its neighbours, not further
        if (!(r.flags & kTriNonOpaque) && outlierArea > 0.0f)
          capOf[t] = std::min((float)BDPT_SPLIT_MAX_PER_TRI, std::floor((float)BDPT_SPLIT_OUTLIER * pc.box.area() / outlierArea));
      }
    }
  });
  // split counts per class: the largest D with sum floor(D p_t) <= budget (integer sums: thread-count independent)
  if (anySplit && nTris && !decideElsewhere) {
    for (int cls = 0; cls < 2; cls++) {
      const float budgetF = cls ? budgetAlpha : budgetOpaque;
      if (!(budgetF > 0.0f)) continue;
      uint64_t members = 0;
      double pmax = 0.0;
      {  // (a count and a maximum: what the threads find does not depend on how the range was shared out)
        std::vector<uint64_t> pm((size_t)threads, 0);
        std::vector<double> px((size_t)threads, 0.0);
        parallelFor(nTris, threads, [&](size_t t0, size_t t1, int th) {
          uint64_t m = 0;
          double x = 0.0;
          for (size_t t = t0; t < t1; t++)
            if (state[t] != 2 && (((recs[t].flags & kTriNonOpaque) != 0) == (cls == 1))) {
              m++;
              x = std::max(x, prio[t]);
            }
          pm[(size_t)th] = m;
          px[(size_t)th] = x;
        });
        for (int th = 0; th < threads; th++) {
          members += pm[(size_t)th];
          pmax = std::max(pmax, px[(size_t)th]);
        }
      }
      const uint64_t budget = (uint64_t)((double)members * (double)budgetF);
      if (!members || !budget || !(pmax > 0.0)) continue;
      auto total = [&](double D) {
        std::vector<uint64_t> part((size_t)threads, 0);
        parallelFor(nTris, threads, [&](size_t t0, size_t t1, int th) {
          uint64_t acc = 0;
          for (size_t t = t0; t < t1; t++)
            if (state[t] != 2 && (((recs[t].flags & kTriNonOpaque) != 0) == (cls == 1)))
              acc += (uint64_t)std::min<double>(std::floor(D * prio[t]), (double)capOf[t]);
          part[(size_t)th] = acc;
        });
        uint64_t s = 0;
        for (uint64_t v : part) s += v;
        return s;
      };
      double dLo = 0.0, dHi = ((double)BDPT_SPLIT_MAX_PER_TRI + 1.0) / pmax;  // at dHi the largest priority is capped
      if (total(dHi) <= budget) {
        dLo = dHi;
      } else {
        for (int it = 0; it < 40; it++) {
          const double mid = 0.5 * (dLo + dHi);
          if (total(mid) <= budget)
            dLo = mid;
          else
            dHi = mid;
        }
      }
      parallelFor(nTris, threads, [&](size_t t0, size_t t1, int) {
        for (size_t t = t0; t < t1; t++)
          if (state[t] != 2 && (((recs[t].flags & kTriNonOpaque) != 0) == (cls == 1)))
            splits[t] = (uint32_t)std::min<double>(std::floor(dLo * prio[t]), (double)capOf[t]);
      });
    }
  }
  lap("priorities");
  // pass 2: the references, triangle order (chunks are contiguous triangle ranges, appended in order), as the
  // 40-byte records the tree is built over (box, centre, reference index) + the triangle every reference belongs to
  BigVec<Ref> refs;
  BigVec<uint32_t> refTri;
  // (a plugged-in reference maker — bdpt_set_scene: the device — makes the same references in the same order and keeps them)
  uint32_t madeElsewhere = 0, droppedElsewhere = 0;
  if (refMaker) {
    BvhRefInput in;
    in.triRecs = recsElsewhere ? nullptr : recs.data();
    in.triBox = recsElsewhere ? nullptr : triBox.data();
    in.splits = decideElsewhere ? nullptr : splits.data();
    in.state = decideElsewhere ? nullptr : state.data();
    in.budgetOpaque = budgetOpaque;
    in.budgetAlpha = budgetAlpha;
    in.outlierArea = outlierArea;
    in.numDroppedOut = &droppedElsewhere;
    in.numTris = nTris;
    for (int a = 0; a < 3; a++) {
      in.gridLo[a] = G.lo[a];
      in.gridExt[a] = G.ext[a];
    }
    in.clipper = opt.clipper;
    in.positions = positions;
    in.indices = indices;
    in.triFlags = triFlags;
    in.triAux = triAux;
    in.numVertices = opt.numVertices;
    std::string err;
    if (!refMaker(opt.treeBuilderUser, in, madeElsewhere, err)) {
      if (opt.error) *opt.error = err.empty() ? "reference maker failed" : err;
      return;
    }
  } else {
    std::vector<RefOut> part((nTris + kRefChunk - 1) / kRefChunk);  // one per chunk, appended in chunk order below
    parallelChunks(nTris, threads, kRefChunk, [&](size_t ci, size_t t0, size_t t1) {
      RefOut& o = part[ci];
      for (size_t t = t0; t < t1; t++) {
        if (state[t] == 2) continue;
        if (splits[t] == 0 && state[t] == 0) {
          o.boxes.push_back(triBox[t]);
          o.tri.push_back((uint32_t)t);
          continue;
        }
        Piece pc;
        bool shrunk = false;
        if (!wholePiece(t, pc, shrunk)) continue;  // (cannot happen: pass 1 kept it)
        if (splits[t] == 0) {
          o.boxes.push_back(pc.box);
          o.tri.push_back((uint32_t)t);
          continue;
        }
        pc.splits = splits[t];
        const bool alpha = (recs[t].flags & kTriNonOpaque) != 0 && opt.clipper != nullptr;
        splitTriangle(G, recs[t], (uint32_t)t, pc, alpha ? opt.clipper : nullptr, o);
      }
    });
    std::vector<size_t> at(part.size() + 1, 0);
    for (size_t ci = 0; ci < part.size(); ci++) at[ci + 1] = at[ci] + part[ci].tri.size();
    refs.resize(at.back());
    refTri.resize(at.back());
    parallelChunks(part.size(), threads, 16, [&](size_t, size_t c0, size_t c1) {  // every chunk knows where it lands
      for (size_t ci = c0; ci < c1; ci++) {
        const RefOut& o = part[ci];
        for (size_t j = 0; j < o.tri.size(); j++) {
          Ref& r = refs[at[ci] + j];
          for (int k = 0; k < 3; k++) {  // (+ 0.0f: -0 becomes +0, so that no minimum or maximum depends on the order in which equal zeros meet)
            r.box.lo[k] = o.boxes[j].lo[k] + 0.0f;
            r.box.hi[k] = o.boxes[j].hi[k] + 0.0f;
            r.cent[k] = 0.5f * (r.box.lo[k] + r.box.hi[k]) + 0.0f;
          }
          r.id = (uint32_t)(at[ci] + j);
          refTri[at[ci] + j] = o.tri[j];
        }
        RefOut().boxes.swap(part[ci].boxes);  // (release as we go: the pieces are as large as the result)
        RefOut().tri.swap(part[ci].tri);
      }
    });
  }
  if (decideElsewhere) {
    out.numDropped = droppedElsewhere;
  } else {
    uint32_t dropped = 0;
    for (uint32_t t = 0; t < nTris; t++) dropped += state[t] == 2 ? 1u : 0u;
    out.numDropped = dropped;
  }
  const uint32_t n = refMaker ? madeElsewhere : (uint32_t)refs.size();  // references from here on
  if (verbose) std::fprintf(stderr, "[bvh] %u triangles -> %u references (%u dropped)\n", nTris, n, out.numDropped);
  BigVec<TmpNode> tmp;
  BigVec<uint32_t> order;  // filled by a plugged-in tree builder: the leaf order as reference ids (the host code permutes `refs` itself)
  const BvhTreeBuilder treeBuilder = opt.treeBuilder ? opt.treeBuilder : gDefaultTreeBuilder;
  void* const treeBuilderUser = opt.treeBuilder ? opt.treeBuilderUser : gDefaultTreeBuilderUser;
  const bool plugged = treeBuilder && n > 0;
  const BvhPacker packer = (plugged && opt.treeBuilder) ? opt.packer : nullptr;
  BigVec<Ref> scratch(plugged ? 0 : n);  // every partition scatters through the node's own range of it
  const BuildData B{refs, scratch};
  lap("records");
  if (plugged) {
    // the binary tree is built elsewhere (bdpt_set_scene: on the device, bvh_device.hip) — the same decisions, the same
    // order of the references, the same tree as the host code below; children come after their parents there too
    std::string err;
    if (!treeBuilder(treeBuilderUser, refMaker ? nullptr : refs.data(), n, order, tmp, err) || tmp.empty() || (!refMaker && order.size() != n)) {
      if (opt.error) *opt.error = err.empty() ? "tree builder failed" : err;
      out.nodes.clear();
      out.tris.clear();
      out.refBox.clear();
      out.recs.clear();
      return;
    }
    lap("device tree");
  } else if (threads <= 1 || n < (1u << 15)) {
    // The host code.  Phase 1: one thread splits the top of the tree (large nodes share their scans among all threads)
    // and defers every subtree of at most `grain` triangles.  Phase 2: the deferred subtrees are built concurrently,
    // largest first, each into its own node list.  Phase 3: the lists are appended; children stay after parents.
    tmp.reserve((size_t)n / 2 + 16);
    tmp.push_back(makeTmpNode(0, n, 0));
    buildSubtree(B, tmp, 0, 0, 1, nullptr);
  } else {
    tmp.reserve((size_t)n / 2 + 16);
    tmp.push_back(makeTmpNode(0, n, 0));
    const uint32_t grain = std::max<uint32_t>(4096, n / (uint32_t)(threads * 8));
    std::vector<uint32_t> deferred;
    buildSubtree(B, tmp, 0, grain, threads, &deferred);
    lap("top");
    std::sort(deferred.begin(), deferred.end(), [&](uint32_t a, uint32_t b) {
      return tmp[a].count > tmp[b].count || (tmp[a].count == tmp[b].count && a < b);
    });
    std::vector<BigVec<TmpNode>> local(deferred.size());
    std::atomic<size_t> next{0};
    auto worker = [&] {
      for (;;) {
        const size_t j = next.fetch_add(1);
        if (j >= deferred.size()) return;
        BigVec<TmpNode>& L = local[j];
        L.reserve((size_t)tmp[deferred[j]].count / 2 + 4);
        L.push_back(tmp[deferred[j]]);
        buildSubtree(B, L, 0, 0, 1, nullptr);
      }
    };
    {
      WorkerScope pool;
      for (int t = 1; t < threads; t++) pool.spawn(worker);
      worker();
      pool.join();
    }
    lap("subtrees");
    // the lists are appended in the order of `deferred` (children stay after parents); every list knows where it
    // lands, so the copies run side by side
    std::vector<size_t> at(deferred.size() + 1);
    at[0] = tmp.size();
    for (size_t j = 0; j < deferred.size(); j++) at[j + 1] = at[j] + local[j].size() - 1;
    tmp.resize(at.back());
    parallelChunks(deferred.size(), threads, 1, [&](size_t j, size_t, size_t) {
      const BigVec<TmpNode>& L = local[j];
      const int32_t off = (int32_t)at[j] - 1;  // local index i >= 1 -> off + i
      TmpNode rootNode = L[0];
      if (rootNode.left >= 0) {
        rootNode.left += off;
        rootNode.right += off;
      }
      tmp[deferred[j]] = rootNode;
      for (size_t i = 1; i < L.size(); i++) {
        TmpNode nd = L[i];
        if (nd.left >= 0) {
          nd.left += off;
          nd.right += off;
        }
        tmp[at[j] + i - 1] = nd;
      }
    });
  }

  lap("append");
  out.numRefs = n;
  if (packer && refMaker && opt.collapseInPacker) {
    // everything from here on — collapse, quantisation, packing, the summary — happens where the tree is (bvh_device.hip)
    BvhPackInput in;
    in.triRecs = recs.data();
    in.numTris = nTris;
    in.refTri = nullptr;
    in.numRefs = n;
    in.wide = nullptr;
    in.slots = nullptr;
    in.numWide = 0;
    in.pad = pad;
    std::string err;
    if (!packer(treeBuilderUser, in, out, err)) {
      if (opt.error) *opt.error = err.empty() ? "packer failed" : err;
      out.deviceRecs = nullptr;
      out.deviceNumRecs = 0;
    }
    lap("device collapse + pack");
    return;
  }
  // Leaf-ordered triangle list (a packer gathers it on the device).
  out.tris.resize(packer ? 0 : n);
  out.refBox.resize(packer ? 0 : (size_t)n * 6);
  if (!packer) parallelFor(n, threads, [&](size_t a, size_t b, int) {
    for (size_t i = a; i < b; i++) {
      const Ref& r = plugged ? refs[order[i]] : refs[i];
      out.tris[i] = recs[refTri[r.id]];
      for (int k = 0; k < 3; k++) {
        out.refBox[i * 6 + (size_t)k] = r.box.lo[k];
        out.refBox[i * 6 + 3 + (size_t)k] = r.box.hi[k];
      }
    }
  });

  // ---- collapse the binary tree into four-wide nodes and quantise the child boxes ----------------
  auto leafRef = [](uint32_t first, uint32_t count) -> int32_t { return -1 - (int32_t)((first << 3) | (count - 1)); };
  using Wide = BvhWideNode;  // (src: tmp index of the subtree root this node covers; kids: tmp indices of the up to 4 children)
  if (n == 0) {
    BvhNode nd;
    std::memset(&nd, 0, sizeof(nd));
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 4; c++) {
        nd.lo[a][c] = 255;
        nd.hi[a][c] = 0;
      }
    for (int c = 0; c < 4; c++) nd.child[c] = -1;
    nd.scale[0] = nd.scale[1] = nd.scale[2] = 1.0f;
    out.nodes.push_back(nd);
    out.numNodes = 1;
    packBvh(out, 1);
    return;
  }
  // Binary height of every subtree: the stack need of a subtree left two-wide is its height, so a
  // node at stack level u may widen to k children only while u + (k-1) + max child height stays
  // within the device stack.  Shallow subtrees (most of the nodes) become four-wide; only the few
  // deep, skinny paths of a SAH tree keep two-wide nodes.
  std::vector<uint16_t> height(tmp.size(), 0);
  std::vector<uint32_t> subtreeNodes(tmp.size(), 1);  // binary nodes in the subtree (itself included)
  for (size_t t = tmp.size(); t-- > 0;)  // children are always created after their parent
    if (tmp[t].left >= 0) {
      height[t] = (uint16_t)(1 + std::max(height[(size_t)tmp[t].left], height[(size_t)tmp[t].right]));
      subtreeNodes[t] = 1 + subtreeNodes[(size_t)tmp[t].left] + subtreeNodes[(size_t)tmp[t].right];
    }
  std::vector<Wide> wide;
  std::vector<BvhSlot> slots;  // per wide node: where its index must be written
  uint32_t wDepth = 0, wStack = 0;
  {
    struct Job {
      uint32_t src, depth, stackAbove;
      int32_t slotNode, slotIdx;
    };
    // One subtree, depth first, appended to (W, SL); `defer` (may be null) receives the jobs of subtrees of at most
    // kCollapseGrain binary nodes instead of descending into them.
    auto collapse = [&](Job rootJob, std::vector<Wide>& W, std::vector<BvhSlot>& SL, std::vector<Job>* defer, uint32_t& dMax,
                        uint32_t& sMax) {
      std::vector<Job> jobs;
      jobs.push_back(rootJob);
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
        const uint32_t self = (uint32_t)W.size();
        W.push_back(w);
        SL.push_back(BvhSlot{j.slotNode, j.slotIdx});
        dMax = std::max(dMax, j.depth);
        const uint32_t need = j.stackAbove + (uint32_t)(w.nk - 1);
        sMax = std::max(sMax, need);
        for (int k = w.nk - 1; k >= 0; k--)
          if (tmp[w.kids[k]].left >= 0) {
            const Job child{w.kids[k], j.depth + 1, need, (int32_t)self, k};
            if (defer && subtreeNodes[w.kids[k]] <= kCollapseGrain)
              defer->push_back(child);
            else
              jobs.push_back(child);
          }
      }
    };
    // The top of the tree by one thread; the subtrees below kCollapseGrain nodes side by side, each into its own list,
    // appended in the order they were met.  The grain is a constant, so the node order does not depend on the thread count.
    std::vector<Job> deferred;
    collapse(Job{0, 0, 0, -1, -1}, wide, slots, tmp.size() > 4 * (size_t)kCollapseGrain ? &deferred : nullptr, wDepth, wStack);
    if (!deferred.empty()) {
      struct Local {
        std::vector<Wide> w;
        std::vector<BvhSlot> sl;
        uint32_t dMax = 0, sMax = 0;
      };
      std::vector<Local> local(deferred.size());
      parallelChunks(deferred.size(), threads, 1, [&](size_t j, size_t, size_t) {
        Local& L = local[j];
        L.w.reserve(subtreeNodes[deferred[j].src] / 2 + 4);
        L.sl.reserve(subtreeNodes[deferred[j].src] / 2 + 4);
        Job r = deferred[j];
        const int32_t parentNode = r.slotNode, parentIdx = r.slotIdx;
        r.slotNode = -2;  // marks the list's root: its slot is a node of the top part
        collapse(r, L.w, L.sl, nullptr, L.dMax, L.sMax);
        L.sl[0] = BvhSlot{parentNode, parentIdx};
      });
      std::vector<size_t> at(deferred.size() + 1);
      at[0] = wide.size();
      for (size_t j = 0; j < deferred.size(); j++) at[j + 1] = at[j] + local[j].w.size();
      wide.resize(at.back());
      slots.resize(at.back());
      parallelChunks(deferred.size(), threads, 1, [&](size_t j, size_t, size_t) {
        const Local& L = local[j];
        std::copy(L.w.begin(), L.w.end(), wide.begin() + (long)at[j]);
        slots[at[j]] = L.sl[0];  // (a node of the top part: global index already)
        for (size_t i = 1; i < L.sl.size(); i++) slots[at[j] + i] = BvhSlot{L.sl[i].node + (int32_t)at[j], L.sl[i].idx};
      });
      for (const Local& L : local) {
        wDepth = std::max(wDepth, L.dMax);
        wStack = std::max(wStack, L.sMax);
      }
    }
  }
  lap("collapse");
  out.maxDepth = wDepth;
  out.maxStack = wStack;
  out.numNodes = (uint32_t)wide.size();
  const float rootArea = tmp[0].box.area();
  auto sahCost = [&] {
    // blocks of kCostBlock nodes summed in node order side by side, the block sums added in block order: the rounding
    // depends on neither the thread count nor on who builds the tree
    constexpr size_t kCostBlock = (size_t)1 << 16;
    std::vector<double> part((wide.size() + kCostBlock - 1) / kCostBlock, 0.0);
    if (rootArea > 0)
      parallelChunks(wide.size(), threads, kCostBlock, [&](size_t ci, size_t w0, size_t w1) {
        double cost = 0.0;
        for (size_t wi = w0; wi < w1; wi++)
          for (int k = 0; k < wide[wi].nk; k++) {
            const TmpNode& c = tmp[wide[wi].kids[k]];
            cost += (c.left < 0 ? kCostTri * c.count : kCostTraverse) * c.box.area() / rootArea;
          }
        part[ci] = cost;
      });
    double cost = 0.0;
    for (double v : part) cost += v;
    return (float)cost + kCostTraverse;
  };
  if (packer) {
    // quantisation and packing happen where the tree was built; the host keeps only the summary
    BvhPackInput in;
    in.triRecs = recs.data();
    in.numTris = nTris;
    in.refTri = refMaker ? nullptr : refTri.data();
    in.numRefs = n;
    in.wide = wide.data();
    in.slots = slots.data();
    in.numWide = wide.size();
    in.pad = pad;
    std::string err;
    std::future<float> cost = std::async(std::launch::async, sahCost);  // (the host's one job meanwhile)
    const bool ok = packer(treeBuilderUser, in, out, err);
    out.sahCost = cost.get();
    if (!ok) {
      if (opt.error) *opt.error = err.empty() ? "packer failed" : err;
      out.deviceRecs = nullptr;
      out.deviceNumRecs = 0;
    }
    lap("device pack");
    return;
  }
  out.nodes.resize(wide.size());
  parallelFor(wide.size(), threads, [&](size_t w0, size_t w1, int) {
  for (size_t wi = w0; wi < w1; wi++) {
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
      {
        union {
          uint32_t u;
          float f;
        } sc0;
        sc0.u = (uint32_t)biased << 23;
        nd.scale[a] = sc0.f;
      }
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
    for (int k = 0; k < 4; k++) nd.child[k] = -1;
    for (int k = 0; k < w.nk; k++) {
      const TmpNode& c = tmp[w.kids[k]];
      if (c.left < 0) nd.child[k] = leafRef(c.first, c.count);  // interior refs are patched below
    }
    out.nodes[wi] = nd;
  }
  });
  for (size_t wi = 1; wi < wide.size(); wi++) out.nodes[(size_t)slots[wi].node].child[slots[wi].idx] = (int32_t)wi;
  out.sahCost = sahCost();
  lap("quantise");
  if (!packBvh(out, threads)) out.recs.clear();  // (leaves of at most 8 triangles always fit: 3 x 8 < 256)
  lap("pack");
}

bool packBvh(Bvh& bvh, int threads) {
  if (threads <= 0) threads = bvhBuildThreads();
  const size_t nn = bvh.nodes.size();
  // record index of every node, and of every node's first child: parents come before their children in `nodes`,
  // so one pass in index order hands out the child blocks (a node's block follows the blocks of all earlier nodes)
  std::vector<uint32_t> pos(nn, 0), base(nn, 0);
  uint64_t next = 1;  // record 0 = the root
  for (size_t i = 0; i < nn; i++) {
    const BvhNode& n = bvh.nodes[i];
    const int nk = bvhNumChildren(n);
    base[i] = (uint32_t)next;
    for (int c = 0; c < nk; c++) {
      const int32_t r = n.child[c];
      if (r >= 0) {
        if ((size_t)r <= i || (size_t)r >= nn) return false;
        pos[(size_t)r] = (uint32_t)next;
        next += 1;
      } else {
        next += (uint64_t)(((uint32_t)(-1 - r)) & 7u) + 1u;
      }
    }
    if (next >= 0x7fffffffull) return false;
  }
  // every record below `next` is written by the loop below (a node at pos[i], a leaf's triangles behind base[i]); the pad
  // records behind them — the device fetches up to four records per leaf visit — are zero
  bvh.recs.resize((size_t)next + kBvhPadRecs);
  for (uint32_t k = 0; k < kBvhPadRecs; k++) bvh.recs[(size_t)next + k] = BvhRec{};
  bool ok = true;
  std::mutex failMutex;
  parallelFor(nn, threads, [&](size_t i0, size_t i1, int) {
    for (size_t i = i0; i < i1; i++) {
      const BvhNode& n = bvh.nodes[i];
      BvhRec rec{};
      std::memcpy(&rec.w[0], n.origin, 12);
      uint32_t ex[3];
      for (int a = 0; a < 3; a++) {
        uint32_t bits;
        std::memcpy(&bits, &n.scale[a], 4);
        ex[a] = (bits >> 23) & 0xffu;  // the scales are powers of two: mantissa 0, sign 0
      }
      std::memcpy(&rec.w[4], n.lo, 12);
      std::memcpy(&rec.w[7], n.hi, 12);
      const int nk = bvhNumChildren(n);
      uint32_t leafBits = 0, offs = 0, off = 0;
      for (int c = 0; c < nk; c++) {
        if (off > 255u) {
          std::lock_guard<std::mutex> g(failMutex);
          ok = false;
          break;
        }
        offs |= off << (8 * c);
        const int32_t r = n.child[c];
        if (r >= 0) {
          off += 1;
        } else {
          leafBits |= 1u << c;
          const uint32_t enc = (uint32_t)(-1 - r), first = enc >> 3, cnt = (enc & 7u) + 1u;
          for (uint32_t k = 0; k < cnt; k++) {
            BvhTri t = bvh.tris[first + k];
            if (k + 1 == cnt) t.flags |= kTriLastOfLeaf;
            std::memcpy(&bvh.recs[(size_t)base[i] + off + k], &t, sizeof(BvhTri));
          }
          off += cnt;
        }
      }
      rec.w[3] = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (leafBits << 24);
      rec.w[10] = base[i];
      rec.w[11] = offs;
      bvh.recs[pos[i]] = rec;
    }
  });
  return ok;
}

}  // namespace bdpt
