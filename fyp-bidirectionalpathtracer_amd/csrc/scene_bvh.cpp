// scene_bvh.cpp — see scene_bvh.h.
#include "scene_bvh.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>

#include "../../include/bdpt.h"

namespace bdpt {
namespace {

template <class F>
void hostFor(size_t n, int threads, const F& f) {
  if (threads <= 1 || n < 4096) {
    f((size_t)0, n);
    return;
  }
  WorkerScope pool;  // bvh.h: a worker's exception is rethrown here, after every thread has been joined
  const size_t chunk = (n + (size_t)threads - 1) / (size_t)threads;
  for (int t = 1; t < threads; t++) {
    const size_t a = std::min(n, chunk * (size_t)t), b = std::min(n, a + chunk);
    if (a < b) pool.spawn([&f, a, b] { f(a, b); });
  }
  f((size_t)0, std::min(n, chunk));
  pool.join();
}

}  // namespace

void buildSceneBvh(const bdpt_scene_desc* d, int threads, float splitBudget, float splitBudgetAlpha, bool classify, SceneBvh& out,
                   BvhTreeBuilder treeBuilder, void* treeBuilderUser, std::string* error, BvhPacker packer, BvhRefMaker refMaker, bool collapseInPacker, bool prioritiesInRefMaker) {
  if (threads <= 0) threads = bvhBuildThreads();
  const bool verbose = std::getenv("BDPT_BUILD_VERBOSE") != nullptr;
  auto tLap = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[scene_bvh] %-12s %.3f s\n", what, std::chrono::duration<double>(t - tLap).count());
    tLap = t;
  };
  const uint32_t n = d->numTriangles;
  // per-triangle traversal flags: BLAS OPAQUE iff AlphaModeOpaque (Falcor Raytracing/RtModel.cpp:221-224),
  // TRIANGLE_CULL_DISABLE iff double-sided (Raytracing/RtScene.cpp:175-178)
  out.triFlags.assign(n, 0u);
  hostFor(n, threads, [&](size_t t0, size_t t1) {
    for (size_t t = t0; t < t1; t++) {
      const uint32_t f = d->materials[d->triMaterial[t]].flags;
      out.triFlags[t] = (BDPT_FLAG_ALPHA_MODE(f) != BDPT_ALPHA_MODE_OPAQUE ? kTriNonOpaque : 0u) | (BDPT_FLAG_DOUBLE_SIDED(f) ? kTriDoubleSided : 0u);
    }
  });
  out.numAlphaMode = out.numAlwaysPass = 0;
  for (uint32_t t = 0; t < n; t++) out.numAlphaMode += (out.triFlags[t] & kTriNonOpaque) ? 1u : 0u;
  lap("flags");
  out.clipper.reset();
  if (out.numAlphaMode) out.clipper.reset(new AlphaClipper(d));  // (the host-side trace hook runs the alpha test through it)
  lap("clipper");
  if (classify && out.numAlphaMode) {
    std::atomic<uint32_t> pass{0};
    hostFor(n, threads, [&](size_t t0, size_t t1) {
      uint32_t acc = 0;
      for (size_t t = t0; t < t1; t++)
        if ((out.triFlags[t] & kTriNonOpaque) && out.clipper->classify((uint32_t)t) == 1) {
          out.triFlags[t] &= ~kTriNonOpaque;
          acc++;
        }
      pass += acc;
    });
    out.numAlwaysPass = pass.load();
  }
  lap("classify");
  // non-opaque triangles get an alpha-test record (device_scene.hpp alphaTestFails); its index travels in BvhTri::aux
  out.triAux.assign(n, 0u);
  out.alphaTris.clear();
  for (uint32_t t = 0; t < n; t++)
    if (out.triFlags[t] & kTriNonOpaque) {
      out.triAux[t] = (uint32_t)out.alphaTris.size();
      out.alphaTris.push_back(t);
    }
  BvhBuildOptions opt;
  opt.threads = threads;
  opt.splitBudget = splitBudget;
  opt.splitBudgetAlpha = splitBudgetAlpha;
  opt.clipper = classify ? out.clipper.get() : nullptr;
  opt.numVertices = d->numVertices;
  opt.treeBuilder = treeBuilder;
  opt.treeBuilderUser = treeBuilderUser;
  opt.packer = packer;
  opt.refMaker = refMaker;
  opt.collapseInPacker = collapseInPacker;
  opt.prioritiesInRefMaker = prioritiesInRefMaker;
  opt.error = error;
  lap("aux");
  buildBvh(d->positions, d->indices, n, out.triFlags.data(), out.bvh, opt, out.triAux.data());
}

// ------------------------------------------------------------------------------------------------
// Host-side trace hook: the queries of device_trace.hpp (MODE 0 closest hit, 1 closest hit with back-face culling,
// 2 any hit; hit iff tmin < t < tmax; ties to the lowest primitive index; the any-hit alpha test) walked on the CPU
// over the builder's own node list, plus the linear scan over every triangle that defines the right answer.
// ------------------------------------------------------------------------------------------------
namespace {

struct V3 {
  float x, y, z;
};
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct HostHit {
  int prim = -1;
  float t = 0, u = 0, v = 0;
};

// Moeller-Trumbore in the order device_trace.hpp triGeom evaluates it
inline bool triGeomHost(int mode, const BvhTri& r, V3 o, V3 d, float tmin, float tmax, float& t, float& u, float& v) {
  const V3 v0{r.v0[0], r.v0[1], r.v0[2]}, e1{r.e1[0], r.e1[1], r.e1[2]}, e2{r.e2[0], r.e2[1], r.e2[2]};
  const V3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  if (mode == 1 && !(r.flags & kTriDoubleSided)) {
    if (!(det > 0.0f)) return false;
  } else {
    if (det == 0.0f) return false;
  }
  const float inv = 1.0f / det;
  const V3 tvec = sub(o, v0);
  u = dot(tvec, pvec) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  const V3 qvec = cross(tvec, e1);
  v = dot(d, qvec) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot(e2, qvec) * inv;
  return (t > tmin) && (t < tmax);
}

struct HostScene {
  const bdpt_scene_desc* d;
  SceneBvh sb;
  std::vector<BvhTri> raw;  // one record per input triangle with the reference's per-material flags (the linear scan)
};

inline float clampedRcpHost(float d) {
  float r = 1.0f / d;
  if (!(std::fabs(r) <= 1.0e30f)) r = std::copysign(1.0e30f, d);
  return r;
}

// true: the query is finished (any hit)
inline bool considerHost(const HostScene& S, int mode, const BvhTri& r, V3 o, V3 d, float tmin, float tmax, HostHit& best, float& bestT) {
  float t, u, v;
  if (!triGeomHost(mode, r, o, d, tmin, tmax, t, u, v)) return false;
  if ((r.flags & kTriNonOpaque) && (mode == 2 || t <= bestT) && S.sb.clipper && S.sb.clipper->testFails(r.prim, u, v)) return false;
  if (mode == 2) {
    best.prim = 0;
    best.t = t;
    return true;
  }
  if (t < bestT || (t == bestT && best.prim >= 0 && (int)r.prim < best.prim)) {
    best.prim = (int)r.prim;
    best.t = t;
    best.u = u;
    best.v = v;
    bestT = t;
  }
  return false;
}

HostHit traceHost(const HostScene& S, int mode, const float* ray, uint64_t& nNodes, uint64_t& nTris) {
  HostHit best;
  const V3 o{ray[0], ray[1], ray[2]}, d{ray[3], ray[4], ray[5]};
  const float tmin = ray[6], tmax = ray[7];
  float bestT = tmax;
  const bool finite = (o.x == o.x) && (o.y == o.y) && (o.z == o.z) && (d.x == d.x) && (d.y == d.y) && (d.z == d.z);
  if (!(finite && tmax > tmin)) return best;
  const Bvh& bvh = S.sb.bvh;
  if (bvh.tris.empty()) return best;
  const float idir[3] = {clampedRcpHost(d.x), clampedRcpHost(d.y), clampedRcpHost(d.z)};
  const float org[3] = {o.x, o.y, o.z};
  const bool neg[3] = {d.x < 0.0f, d.y < 0.0f, d.z < 0.0f};
  int32_t stack[128];
  int sp = 0;
  int32_t cur = 0;
  for (;;) {
    if (cur >= 0) {
      nNodes++;
      const BvhNode& nd = bvh.nodes[(size_t)cur];
      float tn[4];
      bool hit[4];
      for (int c = 0; c < 4; c++) {
        float n = tmin, f = bestT;
        for (int a = 0; a < 3; a++) {
          const float A = nd.scale[a] * idir[a], B = (nd.origin[a] - org[a]) * idir[a];
          const float qn = (float)(neg[a] ? nd.hi[a][c] : nd.lo[a][c]), qf = (float)(neg[a] ? nd.lo[a][c] : nd.hi[a][c]);
          const float t0 = std::fma(qn, A, B), t1 = std::fma(qf, A, B);
          n = std::fmax(n, t0);
          f = std::fmin(f, t1);
        }
        hit[c] = n <= f;
        tn[c] = n;
      }
      int idx[4], m = 0;
      for (int c = 0; c < 4; c++)
        if (hit[c]) idx[m++] = c;
      if (mode != 2) std::stable_sort(idx, idx + m, [&](int a, int b) { return tn[a] < tn[b]; });
      for (int k = m - 1; k >= 1; k--)
        if (sp < 127) stack[sp++] = nd.child[idx[k]];
      if (m > 0) {
        cur = nd.child[idx[0]];
        continue;
      }
    } else {
      const uint32_t enc = (uint32_t)(-1 - cur), first = enc >> 3, cnt = (enc & 7u) + 1u;
      for (uint32_t k = 0; k < cnt; k++) {
        nTris++;
        if (considerHost(S, mode, bvh.tris[first + k], o, d, tmin, tmax, best, bestT)) return best;
      }
    }
    if (sp == 0) break;
    cur = stack[--sp];
  }
  return best;
}

HostHit bruteHost(const HostScene& S, int mode, const float* ray) {
  HostHit best;
  const V3 o{ray[0], ray[1], ray[2]}, d{ray[3], ray[4], ray[5]};
  const float tmin = ray[6], tmax = ray[7];
  float bestT = tmax;
  const bool finite = (o.x == o.x) && (o.y == o.y) && (o.z == o.z) && (d.x == d.x) && (d.y == d.y) && (d.z == d.z);
  if (!(finite && tmax > tmin)) return best;
  for (const BvhTri& r : S.raw)
    if (considerHost(S, mode, r, o, d, tmin, tmax, best, bestT)) return best;
  return best;
}

}  // namespace
}  // namespace bdpt

using namespace bdpt;

extern "C" {

// ---- host-only checks of the builder (no GPU, no context) ------------------------------------------------------
int bdpt_bvh_build_check(const bdpt_scene_desc* d, bdpt_bvh_info* out, char* msg, uint32_t msgCap) try {
  auto say = [&](const std::string& m) {
    if (msg && msgCap) {
      std::snprintf(msg, msgCap, "%s", m.c_str());
    }
    return BDPT_E_INVALID;
  };
  if (!d || !out || !d->positions || !d->indices) return say("null scene");
  Bvh bvh;
  buildBvh(d->positions, d->indices, d->numTriangles, nullptr, bvh);
  out->numNodes = (uint32_t)bvh.nodes.size();
  out->numTriangles = d->numTriangles;
  out->maxDepth = bvh.maxDepth;
  out->nodeBytes = sizeof(BvhRec);
  out->triBytes = sizeof(BvhTri);
  out->sahCost = bvh.sahCost;
  out->maxStack = bvh.maxStack;
  out->numReferences = (uint32_t)bvh.tris.size();
  out->numDropped = bvh.numDropped;
  out->numAlphaMode = out->numAlwaysPass = 0;
  if (bvh.maxStack > (uint32_t)kBvhMaxStack) return say("worst-case stack exceeds kBvhMaxStack");
  if (bvh.tris.size() < d->numTriangles || bvh.refBox.size() != bvh.tris.size() * 6) return say("leaf entry list size");
  {
    // every triangle referenced, and its pieces cover it: sample points of the triangle (a barycentric grid incl. the
    // vertices and edges) must each lie in the box of at least one of its references
    std::vector<uint32_t> start(d->numTriangles + 1, 0);
    for (const BvhTri& t : bvh.tris) {
      if (t.prim >= d->numTriangles) return say("primitive index out of range in the leaf order");
      start[t.prim + 1]++;
    }
    for (uint32_t t = 0; t < d->numTriangles; t++) {
      if (!start[t + 1]) return say("a triangle has no reference");
      start[t + 1] += start[t];
    }
    std::vector<uint32_t> refs(bvh.tris.size()), fill(start.begin(), start.end() - 1);
    for (uint32_t i = 0; i < bvh.tris.size(); i++) refs[fill[bvh.tris[i].prim]++] = i;
    constexpr int kGrid = 8;
    for (uint32_t t = 0; t < d->numTriangles; t++) {
      const BvhTri& r = bvh.tris[refs[start[t]]];
      float ext = 0.0f;
      for (int k = 0; k < 3; k++) ext = std::max(ext, std::max(std::fabs(r.e1[k]), std::fabs(r.e2[k])) + std::fabs(r.v0[k]));
      const float eps = 4e-6f * ext + 1e-30f;
      if (start[t + 1] - start[t] == 1) continue;  // one reference: its box is the triangle's (checked by the walk below)
      for (int i = 0; i <= kGrid; i++)
        for (int j = 0; i + j <= kGrid; j++) {
          const double bu = (double)i / kGrid, bv = (double)j / kGrid;
          double p[3];
          for (int k = 0; k < 3; k++) p[k] = (double)r.v0[k] + bu * (double)r.e1[k] + bv * (double)r.e2[k];
          bool in = false;
          for (uint32_t q = start[t]; q < start[t + 1] && !in; q++) {
            const float* bx = &bvh.refBox[(size_t)refs[q] * 6];
            in = true;
            for (int k = 0; k < 3; k++) in = in && p[k] >= (double)bx[k] - eps && p[k] <= (double)bx[3 + k] + eps;
          }
          if (!in) return say("the pieces of a split triangle do not cover it");
        }
    }
  }
  if (d->numTriangles == 0) return BDPT_OK;
  // walk: returns the exact bounds of a subtree and checks them against the decoded (quantised) box
  std::vector<uint8_t> covered(bvh.tris.size(), 0);
  struct Bounds {
    float lo[3], hi[3];
  };
  bool ok = true;
  std::string why;
  std::function<Bounds(int32_t, uint32_t, uint32_t)> walk = [&](int32_t ref, uint32_t depth, uint32_t stackAbove) -> Bounds {
    Bounds b;
    for (int k = 0; k < 3; k++) {
      b.lo[k] = 1e30f;
      b.hi[k] = -1e30f;
    }
    if (depth > 64) {
      ok = false;
      why = "walk deeper than 64 levels";
      return b;
    }
    if (ref < 0) {
      const uint32_t enc = (uint32_t)(-1 - ref), first = enc >> 3, cnt = (enc & 7u) + 1u;
      for (uint32_t i = 0; i < cnt; i++) {
        if (first + i >= bvh.tris.size() || covered[first + i]) {
          ok = false;
          why = "leaf range out of bounds or shared";
          return b;
        }
        covered[first + i] = 1;
        for (int k = 0; k < 3; k++) {  // the piece this entry stands for (a whole triangle's box holds its three corners)
          b.lo[k] = std::min(b.lo[k], bvh.refBox[(size_t)(first + i) * 6 + (size_t)k]);
          b.hi[k] = std::max(b.hi[k], bvh.refBox[(size_t)(first + i) * 6 + 3 + (size_t)k]);
        }
      }
      return b;
    }
    if ((size_t)ref >= bvh.nodes.size()) {
      ok = false;
      why = "child index out of range";
      return b;
    }
    const BvhNode& n = bvh.nodes[(size_t)ref];
    const uint32_t numChildren = (uint32_t)bvhNumChildren(n);
    if (numChildren < 1 || numChildren > 4 || stackAbove + numChildren - 1 > (uint32_t)kBvhMaxStack) {
      ok = false;
      why = "bad child count or stack bound";
      return b;
    }
    for (int c = 0; c < 4; c++) {
      if (c >= (int)numChildren) {
        for (int k = 0; k < 3; k++)
          if (!(n.lo[k][c] == 255 && n.hi[k][c] == 0)) {
            ok = false;
            why = "unused child slot is not inverted";
          }
        continue;
      }
      Bounds cb = walk(n.child[c], depth + 1, stackAbove + numChildren - 1);
      for (int k = 0; k < 3; k++) {
        if (cb.lo[k] < bvhDecodePlane(n, k, n.lo[k][c]) || cb.hi[k] > bvhDecodePlane(n, k, n.hi[k][c])) {
          ok = false;
          why = "decoded child box does not contain its subtree";
        }
        b.lo[k] = std::min(b.lo[k], cb.lo[k]);
        b.hi[k] = std::max(b.hi[k], cb.hi[k]);
      }
    }
    return b;
  };
  walk(0, 0, 0);
  if (!ok) return say(why);
  for (size_t i = 0; i < covered.size(); i++)
    if (!covered[i]) return say("a leaf entry is not referenced by any leaf");
  // the packed 48-byte records (what the device traverses) must decode to the same tree
  if (bvh.recs.empty()) return say("packed records missing");
  std::vector<uint8_t> used(bvh.recs.size(), 0);
  std::function<void(uint32_t, uint32_t, uint32_t)> walkPacked = [&](uint32_t rec, uint32_t node, uint32_t depth) {
    if (!ok) return;
    if (rec >= bvh.recs.size() || used[rec] || depth > 64) {
      ok = false;
      why = "packed node record out of range or shared";
      return;
    }
    used[rec] = 1;
    const BvhRec& r = bvh.recs[rec];
    const BvhNode& n = bvh.nodes[node];
    bool same = std::memcmp(&r.w[0], n.origin, 12) == 0 && std::memcmp(&r.w[4], n.lo, 12) == 0 && std::memcmp(&r.w[7], n.hi, 12) == 0;
    for (int a = 0; a < 3; a++) {
      const uint32_t bits = ((r.w[3] >> (8 * a)) & 0xffu) << 23;  // the device's decode
      float f;
      std::memcpy(&f, &bits, 4);
      same = same && f == n.scale[a];
    }
    if (!same) {
      ok = false;
      why = "packed node does not decode to its node";
      return;
    }
    const uint32_t numChildren = (uint32_t)bvhNumChildren(n);
    for (uint32_t c = 0; c < numChildren; c++) {
      const uint32_t idx = r.w[10] + ((r.w[11] >> (8 * c)) & 0xffu);
      const bool leaf = ((r.w[3] >> (24 + c)) & 1u) != 0;
      const int32_t ref = n.child[c];
      if (leaf != (ref < 0)) {
        ok = false;
        why = "packed child kind differs";
        return;
      }
      if (!leaf) {
        walkPacked(idx, (uint32_t)ref, depth + 1);
        continue;
      }
      const uint32_t enc = (uint32_t)(-1 - ref), first = enc >> 3, cnt = (enc & 7u) + 1u;
      for (uint32_t k = 0; k < cnt; k++) {
        if (idx + k >= bvh.recs.size() || used[idx + k]) {
          ok = false;
          why = "packed leaf out of range or shared";
          return;
        }
        used[idx + k] = 1;
        BvhTri t = bvh.tris[first + k];
        if (k + 1 == cnt) t.flags |= kTriLastOfLeaf;
        if (std::memcmp(&bvh.recs[idx + k], &t, sizeof(BvhTri)) != 0) {
          ok = false;
          why = "packed leaf triangle differs";
          return;
        }
      }
    }
  };
  walkPacked(0, 0, 0);
  if (!ok) return say(why);
  if (used.size() < kBvhPadRecs) return say("pad records missing");
  for (size_t i = 0; i + kBvhPadRecs < used.size(); i++)  // (the last kBvhPadRecs records are the pad behind the array)
    if (!used[i]) return say("a packed record is not referenced");
  for (size_t i = used.size() - kBvhPadRecs; i < used.size(); i++)
    if (used[i]) return say("a pad record is referenced");
  return BDPT_OK;
} catch (const std::bad_alloc&) {
  return BDPT_E_NOMEM;  // (also when a worker thread ran out of memory: bvh.h WorkerScope)
} catch (...) {
  return BDPT_E_INVALID;
}

int bdpt_bvh_build_hash(const bdpt_scene_desc* d, int threads, uint64_t* out_hash, bdpt_bvh_info* out_info) try {
  if (!d || !out_hash || !d->positions || !d->indices) return BDPT_E_INVALID;
  // a scene with materials goes through everything bdpt_set_scene does (flags, alpha classification, pre-splitting)
  SceneBvh sb;
  const bool full = d->materials && d->triMaterial && d->numMaterials;
  if (full)
    buildSceneBvh(d, threads, -1.0f, -1.0f, true, sb);
  else
    buildBvh(d->positions, d->indices, d->numTriangles, nullptr, sb.bvh, threads);
  Bvh& bvh = sb.bvh;
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the node array, the leaf-ordered triangles and the summary
  auto mix = [&](const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  mix(bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode));
  mix(bvh.tris.data(), bvh.tris.size() * sizeof(BvhTri));
  mix(bvh.refBox.data(), bvh.refBox.size() * sizeof(float));
  mix(bvh.recs.data(), bvh.recs.size() * sizeof(BvhRec));
  mix(&bvh.maxDepth, sizeof(bvh.maxDepth));
  mix(&bvh.maxStack, sizeof(bvh.maxStack));
  mix(&bvh.sahCost, sizeof(bvh.sahCost));
  *out_hash = h;
  if (out_info) {
    out_info->numNodes = (uint32_t)bvh.nodes.size();
    out_info->numTriangles = d->numTriangles;
    out_info->maxDepth = bvh.maxDepth;
    out_info->nodeBytes = sizeof(BvhRec);
    out_info->triBytes = sizeof(BvhTri);
    out_info->sahCost = bvh.sahCost;
    out_info->maxStack = bvh.maxStack;
    out_info->reserved = (uint32_t)bvhBuildThreads();
    out_info->numReferences = (uint32_t)bvh.tris.size();
    out_info->numDropped = bvh.numDropped;
    out_info->numAlphaMode = sb.numAlphaMode;
    out_info->numAlwaysPass = sb.numAlwaysPass;
  }
  return BDPT_OK;
} catch (const std::bad_alloc&) {
  return BDPT_E_NOMEM;  // (also when a worker thread ran out of memory: bvh.h WorkerScope)
} catch (...) {
  return BDPT_E_INVALID;
}

void* bdpt_host_bvh_create(const bdpt_scene_desc* d, int threads, float splitBudget, float splitBudgetAlpha, int classify, bdpt_bvh_info* info) {
  if (!d || !d->positions || !d->indices || !d->triMaterial || !d->materials) return nullptr;
  HostScene* S = nullptr;
  try {
    S = new HostScene();
    S->d = d;
    buildSceneBvh(d, threads, splitBudget, splitBudgetAlpha, classify != 0, S->sb);
    S->raw.resize(d->numTriangles);
    for (uint32_t t = 0; t < d->numTriangles; t++) {
      BvhTri& r = S->raw[t];
      const float* a = d->positions + (size_t)d->indices[(size_t)t * 3] * 3;
      const float* b = d->positions + (size_t)d->indices[(size_t)t * 3 + 1] * 3;
      const float* c = d->positions + (size_t)d->indices[(size_t)t * 3 + 2] * 3;
      for (int k = 0; k < 3; k++) {
        r.v0[k] = a[k];
        r.e1[k] = b[k] - a[k];
        r.e2[k] = c[k] - a[k];
      }
      r.prim = t;
      const uint32_t f = d->materials[d->triMaterial[t]].flags;
      r.flags = (BDPT_FLAG_ALPHA_MODE(f) != BDPT_ALPHA_MODE_OPAQUE ? kTriNonOpaque : 0u) | (BDPT_FLAG_DOUBLE_SIDED(f) ? kTriDoubleSided : 0u);
      r.aux = 0;
    }
  } catch (...) {
    delete S;
    return nullptr;
  }
  if (info) {
    std::memset(info, 0, sizeof(*info));
    info->numNodes = (uint32_t)S->sb.bvh.nodes.size();
    info->numTriangles = d->numTriangles;
    info->maxDepth = S->sb.bvh.maxDepth;
    info->nodeBytes = sizeof(BvhRec);
    info->triBytes = sizeof(BvhTri);
    info->sahCost = S->sb.bvh.sahCost;
    info->maxStack = S->sb.bvh.maxStack;
    info->numReferences = (uint32_t)S->sb.bvh.tris.size();
    info->numDropped = S->sb.bvh.numDropped;
    info->numAlphaMode = S->sb.numAlphaMode;
    info->numAlwaysPass = S->sb.numAlwaysPass;
  }
  return S;
}

void bdpt_host_bvh_destroy(void* h) { delete static_cast<HostScene*>(h); }

int bdpt_host_bvh_trace(void* h, const float* rays, uint32_t n, int mode, int brute, int threads, int32_t* out_prim, float* out_tuv,
                        uint64_t* out_visits) {
  if (!h || !rays || mode < 0 || mode > 2) return BDPT_E_INVALID;
  const HostScene& S = *static_cast<HostScene*>(h);
  if (threads <= 0) threads = bvhBuildThreads();
  std::atomic<uint64_t> nodes{0}, tris{0};
  auto work = [&](size_t a, size_t b) {
    uint64_t nn = 0, nt = 0;
    for (size_t i = a; i < b; i++) {
      const HostHit hh = brute ? bruteHost(S, mode, rays + i * 8) : traceHost(S, mode, rays + i * 8, nn, nt);
      if (out_prim) out_prim[i] = hh.prim;
      if (out_tuv) {
        const bool rec = mode != 2 && hh.prim >= 0;
        out_tuv[i * 3] = rec ? hh.t : 0.0f;
        out_tuv[i * 3 + 1] = rec ? hh.u : 0.0f;
        out_tuv[i * 3 + 2] = rec ? hh.v : 0.0f;
      }
    }
    nodes += nn;
    tris += nt;
  };
  if (threads <= 1 || n < 64) {
    work(0, n);
  } else {
    WorkerScope pool;
    const size_t chunk = ((size_t)n + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; t++) {
      const size_t a = std::min<size_t>(n, chunk * (size_t)t), b = std::min<size_t>(n, a + chunk);
      if (a < b) pool.spawn([&work, a, b] { work(a, b); });
    }
    try {
      pool.join();
    } catch (...) {
      return BDPT_E_NOMEM;
    }
  }
  if (out_visits) {
    out_visits[0] = nodes.load();
    out_visits[1] = tris.load();
  }
  return BDPT_OK;
}

}  // extern "C"
