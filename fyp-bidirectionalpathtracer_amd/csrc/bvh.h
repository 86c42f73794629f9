// bvh.h — host-side acceleration-structure build.  Replaces the DXR driver's
// BuildRaytracingAccelerationStructure calls issued by Falcor
// (Raytracing/RtModel.cpp:181-254 bottom level, Raytracing/RtScene.cpp:220-308 top level);
// instancing is flattened on input so one level suffices.
#pragma once
#include <cstdint>
#include <vector>

#include <cstdlib>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace bdpt {
// std::vector whose resize() leaves new elements UNINITIALISED (default-initialisation instead of value-initialisation):
// the builder's arrays are hundreds of megabytes each and are filled by parallel loops right after they are sized;
// letting one thread zero them first — and take every page fault — cost more than the loops (round 4: 8 GB of such
// fills in a 10 M-triangle build).  Only for element types whose default-initialisation does nothing, and only where
// every element is written before it is read.
template <class T>
struct NoInitAllocator : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = NoInitAllocator<U>;
  };
  NoInitAllocator() = default;
  template <class U>
  NoInitAllocator(const NoInitAllocator<U>&) {}
  template <class U>
  void construct(U* p) {
    ::new (static_cast<void*>(p)) U;
  }
  template <class U, class A0, class... A>
  void construct(U* p, A0&& a0, A&&... a) {
    ::new (static_cast<void*>(p)) U(static_cast<A0&&>(a0), static_cast<A&&>(a)...);
  }
};
template <class T>
using BigVec = std::vector<T, NoInitAllocator<T>>;

// Worker threads of the host-side builder.  An exception thrown on a worker (std::bad_alloc from the large per-chunk
// allocations of a 10 M-triangle scene, above all) must not reach std::terminate: it is kept, every thread is joined,
// and join() rethrows it on the caller's thread, where bdpt_set_scene turns it into BDPT_E_NOMEM.  Leaving the scope
// early (the caller's own chunk threw) joins the threads as well.
class WorkerScope {
 public:
  template <class F>
  void spawn(F f) {
    mPool.emplace_back([this, f] {
      try {
        if (std::getenv("BDPT_TEST_THROW_IN_WORKER")) throw std::bad_alloc();  // test hook: tests/test_bvh_builder.py
        f();
      } catch (...) {
        std::lock_guard<std::mutex> g(mLock);
        if (!mError) mError = std::current_exception();
      }
    });
  }
  void join() {
    for (std::thread& t : mPool) t.join();
    mPool.clear();
    if (mError) {
      std::exception_ptr e = mError;
      mError = nullptr;
      std::rethrow_exception(e);
    }
  }
  ~WorkerScope() {
    for (std::thread& t : mPool)
      if (t.joinable()) t.join();
  }

 private:
  std::vector<std::thread> mPool;
  std::exception_ptr mError;
  std::mutex mLock;
};


// 64-byte four-child node with child boxes quantised to 8 bits per plane relative to the node's
// own box: plane = origin[axis] + q * scale[axis], scale a power of two.  Quantisation rounds outward, so a
// decoded child box always contains the (already padded) exact one.  One node = four 16-byte
// loads per lane, the same as a two-child fp32 node, for half the dependent fetches per ray.
//   ref >= 0 : interior node index
//   ref <  0 : leaf, -1 - ((firstTriangle << 3) | (count - 1)), count in 1..8
// Unused child slots have lo = 255, hi = 0 on every axis (never entered: the slab test picks
// near/far planes by ray direction sign, so an inverted box has tnear > tfar).
// (The scales are stored as ready-to-use floats, not as exponent bytes: the node visit is bound by VALU issue and the
// decode cost six instructions per visit; the child count is implied by the inverted boxes of unused slots.)
struct alignas(16) BvhNode {
  float origin[3];
  float scale[3];    // 2^e per axis
  uint8_t lo[3][4];  // [axis][child]
  uint8_t hi[3][4];
  int32_t child[4];
};
inline int bvhNumChildren(const BvhNode& n) {
  int k = 0;
  for (int c = 0; c < 4; c++)
    if (!(n.lo[0][c] == 255 && n.hi[0][c] == 0 && n.lo[1][c] == 255 && n.hi[1][c] == 0 && n.lo[2][c] == 255 && n.hi[2][c] == 0)) k = c + 1;
  return k;
}
static_assert(sizeof(BvhNode) == 64, "node must be 64 bytes");

// 48-byte leaf triangle as intersected: v0, e1 = v1 - v0, e2 = v2 - v0 (+ ids in the w lanes).
struct alignas(16) BvhTri {
  float v0[3];
  uint32_t prim;  // index into the caller's triangle list
  float e1[3];
  uint32_t flags;  // bit0 non-opaque (any-hit alpha test), bit1 double-sided (no back-face cull)
  float e2[3];
  uint32_t aux;  // caller's word per triangle (bdpt_set_scene: index of the alpha-test record of a non-opaque triangle)
};
static_assert(sizeof(BvhTri) == 48, "triangle must be 48 bytes");

#ifndef KSTACK
#define KSTACK 32
#endif
#ifndef BDPT_BVH_STACK_BUDGET
#define BDPT_BVH_STACK_BUDGET (KSTACK - 1)
#endif
constexpr int kBvhMaxStack = BDPT_BVH_STACK_BUDGET;  // worst-case traversal stack entries (the device stack holds KSTACK per lane)
constexpr uint32_t kTriNonOpaque = 1u, kTriDoubleSided = 2u;

// What the device traverses: ONE array of 48-byte records in which a node's children — interior nodes (one record
// each) and leaves (their triangles, one BvhTri record each) — follow one another from `childBase` on.
// A divergent wave pays for every 16-byte load of every lane (the vector-memory address unit takes one lane-load per
// clock per CU: profiles/README.md), so a node visit is three loads here instead of four:
//   word 0-2  origin.xyz (float)
//   word 3    biased scale exponents ex | ey << 8 | ez << 16 (scale = 2^(e-127)), leaf bits << 24 (bit c: child c is a leaf)
//   word 4-9  lo.x lo.y lo.z hi.x hi.y hi.z, four child bytes each (as BvhNode)
//   word 10   childBase: record index of the first child
//   word 11   record offset of child c from childBase in byte c (child 0: 0)
// Device references: ref >= 0 interior record index; ref < 0 leaf, ~ref = record index of its first triangle; the last
// triangle of a leaf carries kTriLastOfLeaf in its flags.  The root is record 0; kBvhPadRecs zero pad records end the array.
constexpr int kRecF4 = 3;  // 16-byte words per record
struct alignas(16) BvhRec {
  uint32_t w[4 * kRecF4];
};
static_assert(sizeof(BvhRec) == 48, "record size");
constexpr uint32_t kTriLastOfLeaf = 4u;  // BvhTri::flags bit set by packBvh (device-side records only)
constexpr uint32_t kBvhPadRecs = 4;      // zero records behind the array (a leaf fetch reads past a leaf's last triangle)

struct Bvh {
  BigVec<BvhNode> nodes;
  BigVec<BvhTri> tris;   // in leaf order: one record per REFERENCE (a split triangle appears once per piece)
  BigVec<float> refBox;  // 6 floats (lo, hi) per entry of `tris`: bounds of the piece the reference stands for
  BigVec<BvhRec> recs;   // packed device form of the two (packBvh)
  uint32_t maxDepth = 0;     // depth of the four-wide tree
  uint32_t maxStack = 0;     // worst-case number of simultaneously stacked references
  float sahCost = 0.0f;
  uint32_t numDropped = 0;   // input triangles with no reference at all (the clipper found nothing that can be hit)
  uint32_t numNodes = 0;     // four-wide nodes and references (= nodes.size(), tris.size() when the host code packed)
  uint32_t numRefs = 0;
  // With BvhBuildOptions::packer: the packed records are built in device memory and nodes / tris / refBox / recs stay empty.
  void* deviceRecs = nullptr;  // hipMalloc'ed, the caller's to free
  size_t deviceNumRecs = 0;    // (including the kBvhPadRecs pad records)
};

// Lets the caller shrink or drop the part of a triangle a reference stands for.  Used for non-opaque triangles
// (flag kTriNonOpaque): a hit is only ever reported where the any-hit alpha test passes, so a piece whose texels
// all fail can never produce a hit and needs no reference, and a reference only has to bound the part of its
// piece where the test can pass.  `poly` is a convex polygon in the triangle's barycentric plane, vertex k =
// (bu, bv) with P = v0 + bu e1 + bv e2, at most kBvhPolyMax vertices.  Returns false when nothing is left.
constexpr int kBvhPolyMax = 24;
// Opaque triangles are only split when their box is an outlier — at least BDPT_SPLIT_OUTLIER times the median box area of
// the scene's opaque triangles — and only down to about that size (bvh_build.cpp "References"); no triangle is split more
// than BDPT_SPLIT_MAX_PER_TRI times.
#ifndef BDPT_SPLIT_OUTLIER
#define BDPT_SPLIT_OUTLIER 8.0f
#endif
#ifndef BDPT_SPLIT_MAX_PER_TRI
#define BDPT_SPLIT_MAX_PER_TRI 255
#endif
#ifdef __HIPCC__
#define BVH_HD __host__ __device__
#else
#define BVH_HD
#endif
// Cube root of a non-negative finite double in plain arithmetic — exponent reduced to a multiple of three by bit
// manipulation, six Newton steps on the mantissa — so that the host compiler and the device compiler produce the same bits
// (libm's cbrt is correctly rounded on neither in general, and the two differ): the split priorities are computed with it.
BVH_HD inline double bvhCbrt(double x) {
  if (!(x > 0.0)) return 0.0;
  int scaled = 0;
  if (x < 2.2250738585072014e-308) {  // denormal: an exact scaling by 2^108 first, 2^-36 at the end
    x *= 324518553658426726783156020576256.0;
    scaled = -36;
  }
  unsigned long long u;
  __builtin_memcpy(&u, &x, 8);
  const int e = (int)((u >> 52) & 0x7ffull) - 1023;
  const int q = e >= 0 ? e / 3 : -((-e + 2) / 3);
  const int r = e - 3 * q;  // 0, 1, 2
  u = (u & 0x000fffffffffffffull) | ((unsigned long long)(1023 + r) << 52);
  double m;
  __builtin_memcpy(&m, &u, 8);  // in [1, 8)
  double t = m < 2.0 ? 1.1 : (m < 4.0 ? 1.4 : 1.8);
  for (int i = 0; i < 6; i++) t = t - (t * t * t - m) / (3.0 * t * t);
  const unsigned long long p = (unsigned long long)(1023 + q + scaled) << 52;
  double s2;
  __builtin_memcpy(&s2, &p, 8);
  return t * s2;
}
// What a clipper decides with, as plain tables — for an implementation of the same decisions somewhere else (the device:
// bvh_device.hip runs alpha_clip.cpp's clip() from these).  All pointers stay the clipper's / the scene's.
struct BvhClipTables {
  const uint32_t* triMaterial = nullptr;  // per triangle
  const uint32_t* indices = nullptr;      // 3 per triangle
  const float* texcoords = nullptr;       // 3 floats per vertex, or null
  uint32_t numTriangles = 0, numVertices = 0;
  std::vector<int32_t> matMask;     // per material: index into masks, -1: no texture decides
  std::vector<int32_t> matVerdict;  // for matMask < 0: 1 the test always passes, 2 it always fails
  struct Mask {
    int32_t w, h;
    const uint32_t* mayPass;  // summed-area table, (w + 1) x (h + 1): cells a sample may pass in
  };
  std::vector<Mask> masks;
};
struct BvhRefClipper {
  virtual ~BvhRefClipper() {}
  virtual bool clip(uint32_t tri, double (*poly)[2], int& n) const = 0;
  virtual bool tables(BvhClipTables&) const { return false; }  // false: this clipper's decisions exist as code only
};

// ---- what the binary-tree stage of the builder works on (bvh_build.cpp on the host, bvh_device.hip on the device) ----
#ifndef BDPT_SAH_BINS
#define BDPT_SAH_BINS 16
#endif
#ifndef BDPT_LEAF_MAX
#define BDPT_LEAF_MAX 2
#endif
constexpr int kBvhBins = BDPT_SAH_BINS;           // SAH bins per axis
constexpr uint32_t kBvhLeafMax = BDPT_LEAF_MAX;   // references per leaf, at most 8 (three count bits in a leaf reference)
constexpr int kBvhBinaryMaxDepth = kBvhMaxStack;  // depth budget of the binary tree: a two-wide path stacks one reference per level
struct BvhBox {
  float lo[3], hi[3];
  // (host-side helpers; the device code of bvh_device.hip spells the same arithmetic out itself)
  void reset() {
    lo[0] = lo[1] = lo[2] = 1e30f;
    hi[0] = hi[1] = hi[2] = -1e30f;
  }
  void grow(const BvhBox& b) {
    for (int a = 0; a < 3; a++) {
      lo[a] = b.lo[a] < lo[a] ? b.lo[a] : lo[a];  // std::min(lo, b.lo)
      hi[a] = hi[a] < b.hi[a] ? b.hi[a] : hi[a];  // std::max(hi, b.hi)
    }
  }
  void grow(const float* p) {
    for (int a = 0; a < 3; a++) {
      lo[a] = p[a] < lo[a] ? p[a] : lo[a];
      hi[a] = hi[a] < p[a] ? p[a] : hi[a];
    }
  }
  float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
  }
};
// One reference: the box of the piece it stands for, the box's centre, its index in the reference list.  The
// records themselves are permuted as nodes are partitioned; a node owns a contiguous range.
struct BvhBuildRef {
  BvhBox box;
  float cent[3];
  uint32_t id;
};
static_assert(sizeof(BvhBuildRef) == 40, "reference record");
struct BvhBuildNode {  // (no member initialisers: arrays of these are sized without being touched)
  BvhBox box;
  int32_t left, right;    // children (indices into the node list, always behind their parent) or -1
  uint32_t first, count;  // leaf: its range of the reference array
  uint32_t depth;
};
static_assert(sizeof(BvhBuildNode) == 44, "binary node");
// Builds the binary tree over refs[0, n) — binned SAH, stable partitions, median fallback: the decisions, the order of the
// references and the tree of bvh_build.cpp's host code, bit for bit — somewhere else (bdpt_set_scene plugs in the device
// implementation, bvh_device.hip).  `refs` stays as it is; order[i] is the reference (its index = its id) at position i
// of the leaf order the nodes' ranges speak of.  nodes[0] is the root; children come after their parents.  false + err
// on failure.
using BvhTreeBuilder = bool (*)(void* user, const BvhBuildRef* refs, uint32_t n, BigVec<uint32_t>& order, BigVec<BvhBuildNode>& nodes, std::string& err);
// After the collapse (host): the four-wide nodes as lists of binary nodes, and where each one's index goes in its parent.
struct BvhWideNode {
  uint32_t src;      // binary node this wide node covers
  uint32_t kids[4];  // the binary nodes that became its (up to 4) children
  int32_t nk;
  uint32_t depth;
};
struct BvhSlot {
  int32_t node, idx;  // wide node and child slot that refer to this wide node (-1: the root)
};
struct Bvh;
struct BvhPackInput {
  const BvhTri* triRecs;  // one per input triangle
  uint32_t numTris;
  const uint32_t* refTri;  // the triangle of every reference, by reference id (null: the BvhRefMaker of this build kept it, and triRecs)
  uint32_t numRefs;
  const BvhWideNode* wide;  // null: the collapse has not happened — the packer does it from the tree of this build and fills
  const BvhSlot* slots;     // out.numNodes / maxDepth / maxStack / sahCost as well
  size_t numWide;
  float pad;  // what every child box is padded by before it is quantised
};
// Makes the references (bvh_build.cpp "References": every triangle's whole piece, shrunk by the clipper, split `splits`
// times, every piece clipped again) somewhere else, in the host code's order, and keeps them there for the
// BvhTreeBuilder of the same `user` (which is then called with refs = null) and the BvhPacker.
struct BvhRefInput {
  const BvhTri* triRecs;   // one per input triangle
  const BvhBox* triBox;
  const uint32_t* splits;  // split count per triangle; null: the maker also decides what the clipper leaves of every
  const uint8_t* state;    // triangle (state: 0 = plain reference (triBox), 1 = shrunk by the clipper, 2 = dropped), its split
  uint32_t numTris;        // priority and the split counts that meet the budgets below (bvh_build.cpp pass 1 + "split counts")
  float budgetOpaque, budgetAlpha;  // extra references per triangle of the class, on average (0: the class is not split)
  float outlierArea;                // opaque triangles below this box area are never split
  uint32_t* numDroppedOut;          // (with splits == null) receives the number of dropped triangles
  double gridLo[3], gridExt[3];      // the scene box: the split planes are its spatial medians
  const BvhRefClipper* clipper;      // for the non-opaque triangles; may be null
  // What triRecs / triBox were made from (buildBvh's inputs; numVertices 0: not given).  A maker that works on another
  // device can make the two arrays there from these — (12 numVertices + 20 numTris) bytes to hand over instead of
  // 72 numTris — with the loop of buildBvh "Triangle records", which is plain fp32 arithmetic (bit-identical).
  const float* positions = nullptr;
  const uint32_t* indices = nullptr;
  const uint32_t* triFlags = nullptr;  // may be null (all 0)
  const uint32_t* triAux = nullptr;    // may be null (all 0)
  uint32_t numVertices = 0;
};
using BvhRefMaker = bool (*)(void* user, const BvhRefInput& in, uint32_t& numRefs, std::string& err);
// Quantises the child boxes and packs nodes and leaf triangles into the device's record array — what the host code does
// between "collapse" and the upload — from the order and the nodes the BvhTreeBuilder of the same `user` left behind.
// Fills out.deviceRecs / out.deviceNumRecs (the caller owns the allocation: hipFree) and nothing else.
using BvhPacker = bool (*)(void* user, const BvhPackInput& in, Bvh& out, std::string& err);
// Test hook: the tree builder (and packer) buildBvh uses when its options name none (null = the host code).
// bdpt_test_tree_builder (api.cpp) points it at the device implementation so that the host-only hash / check hooks can be
// run over a device-built tree and compared with the host-built one.
void bvhSetDefaultTreeBuilder(BvhTreeBuilder f, void* user);

// The device implementation (bvh_device.hip).  One BvhDeviceBuild per build: the stages hand their results to one another
// in device memory through it (`user` of both functions).
struct BvhDeviceBuild;
BvhDeviceBuild* bvhDeviceBuildBegin(int device, bool collapseOnDevice = false);
void bvhDeviceBuildEnd(BvhDeviceBuild* b);
// pageable host memory -> the current device through pinned staging buffers and a few copy threads (bvh_device.hip)
bool bvhUploadStaged(void* dst, const void* src, size_t bytes, std::string& err);
// pins the staging buffers of bvhUploadStaged on a thread of its own (once per process; bdpt_create calls it)
void bvhPrewarmStaging(int device);
bool buildBinaryTreeOnDevice(void* user, const BvhBuildRef* refs, uint32_t n, BigVec<uint32_t>& order, BigVec<BvhBuildNode>& nodes, std::string& err);
bool packOnDevice(void* user, const BvhPackInput& in, Bvh& out, std::string& err);
bool makeReferencesOnDevice(void* user, const BvhRefInput& in, uint32_t& numRefs, std::string& err);

struct BvhBuildOptions {
  int threads = 0;                // <= 0: bvhBuildThreads()
  // Spatial pre-splitting (bvh_build.cpp "References"): extra references the builder may create, as a fraction of the
  // number of opaque / non-opaque triangles.  0 = one reference per triangle (object splits only).
  float splitBudget = -1.0f;      // < 0: the build default (BDPT_SPLIT_BUDGET)
  float splitBudgetAlpha = -1.0f; // < 0: the build default (BDPT_SPLIT_BUDGET_ALPHA)
  const BvhRefClipper* clipper = nullptr;  // applied to the pieces of triangles flagged kTriNonOpaque
  uint32_t numVertices = 0;                // vertices `positions` holds (0: unknown; only a plugged-in reference maker asks)
  BvhTreeBuilder treeBuilder = nullptr;    // null: the host code builds the binary tree
  bool prioritiesInRefMaker = false;       // (with a refMaker) it also classifies, rates and assigns the split counts (BvhRefInput::splits = null)
  BvhRefMaker refMaker = nullptr;          // (with a treeBuilder and a packer only) null: the host code makes the references
  bool collapseInPacker = false;           // (with refMaker, treeBuilder and packer) the packer also does the four-wide collapse (in.wide = null) and fills the summary
  BvhPacker packer = nullptr;              // (with a treeBuilder only) null: the host code quantises and packs; else Bvh::deviceRecs is the result
  void* treeBuilderUser = nullptr;
  std::string* error = nullptr;            // receives the tree builder's message when it fails (the build then has no nodes)
};

// positions: 3 floats per vertex; indices: 3 per triangle; triFlags: per triangle (may be null).
// threads <= 0: bvhBuildThreads().  The tree does not depend on the thread count, bit for bit.
void buildBvh(const float* positions, const uint32_t* indices, uint32_t numTriangles, const uint32_t* triFlags, Bvh& out,
              int threads = 0, const uint32_t* triAux = nullptr);
void buildBvh(const float* positions, const uint32_t* indices, uint32_t numTriangles, const uint32_t* triFlags, Bvh& out,
              const BvhBuildOptions& opt, const uint32_t* triAux = nullptr);
// host threads the builder uses by default: BDPT_BUILD_THREADS, else the affinity mask capped by the cgroup CPU quota
int bvhBuildThreads();

// Re-encodes nodes + tris as the 48-byte record array (called by buildBvh; a pure function of the two).
// Returns false when a child block does not fit the format (more than 255 records before a node's last child).
bool packBvh(Bvh& bvh, int threads = 0);

// Decode one quantised plane exactly as the device does.
inline float bvhDecodePlane(const BvhNode& n, int axis, uint8_t q) { return n.origin[axis] + (float)q * n.scale[axis]; }

}  // namespace bdpt
