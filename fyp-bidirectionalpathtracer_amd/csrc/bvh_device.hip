// bvh_device.hip — the acceleration-structure build ON THE DEVICE.
//
// Replaces what the reference leaves to the DXR driver on the GPU as well — BLAS / TLAS builds issued by
// Falcor/Framework/Source/Raytracing/RtModel.cpp:181-254 and RtScene.cpp:220-308.  Four stages of buildBvh
// (bvh_build.cpp) are plugged in from here by bdpt_set_scene, each producing the host code's result bit for bit (the host
// code stays as what the CPU tests run and as the definition; tests/test_gpu_configs.py compares the two):
//   makeReferencesOnDevice   classification, split priorities and split counts; then the references: whole piece, clipped by
//                            the alpha clipper, split, every piece clipped again
//   buildBinaryTreeOnDevice  the binned-SAH binary tree over them
//   packOnDevice             the four-wide collapse (collapseOnDevice), child boxes quantised, nodes + leaf triangles packed
//                            into the record array the kernels traverse
// (on the host: the triangle records, before all of it).
//
// The tree is level-synchronous: every level is a handful of launches —
//   bounds   node box + centroid box per active node ("slot")   (ordered-uint atomics; a block first reduces the
//   bin      3 x 16 bins per split candidate: box + count        references of its dominant slot in registers / LDS)
//   prepare  leaf / forced-median / split candidate; bin grid    (one thread per slot)
//   decide   the SAH sweep of bvh_build.cpp splitNode            (one thread per slot, the same float operations in the same order)
//   small    all of the above for a slot of <= kSmall references (one wave per slot, bins in LDS, no atomics on memory)
//   flags -> exclusive scan -> stable partition scatter          (lefts in their order, then rights in theirs)
//   children two nodes per split slot, in slot order
// — and produces, bit for bit, the permutation of the references and the tree the host code produces (one partition
// rule there too).  Median fallback (no useful SAH split, or the depth budget): the count / 2 references that come first by
// (centroid, reference id) go left — the pivot found by one wave's ranking for up to kMedianWave references, by a radix
// select over the 64-bit key beyond that — through the same stable partition.
// Float min / max through atomics on an order-preserving uint encoding: exact, order-independent.  -0 is turned into +0
// when the references are made, so no box component depends on the order in which equal zeros met.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh.h"

namespace bdpt {

// The scratch of one build: chunks of device memory the stages carve their arrays from and give back when they end, so
// that a stage re-uses what the one before it held.  (The driver clears previously used VRAM when it is allocated again,
// at some 35 GB/s: three stages taking and freeing 8-14 GB each paid for that three times, and so did whoever allocated
// next; it is also sixty hipMalloc calls less.)
struct BvhDeviceArena {
  struct Chunk {
    char* p;
    size_t size, used;
  };
  std::vector<Chunk> chunks;
  static constexpr size_t kMinChunk = (size_t)1 << 30;
  void* alloc(size_t bytes) {
    // test hook: BDPT_TEST_FAIL_DEVICE_ALLOC=<n> makes the n-th request of the process fail (tests/test_gpu_configs.py)
    static const long failAt = std::getenv("BDPT_TEST_FAIL_DEVICE_ALLOC") ? std::atol(std::getenv("BDPT_TEST_FAIL_DEVICE_ALLOC")) : -1;
    static std::atomic<long> requests{0};
    if (failAt >= 0 && ++requests == failAt) return nullptr;
    bytes = (std::max<size_t>(bytes, 16) + 255) & ~(size_t)255;
    for (Chunk& c : chunks)
      if (c.used + bytes <= c.size) {
        void* q = c.p + c.used;
        c.used += bytes;
        return q;
      }
    void* q = nullptr;
    const size_t size = std::max(bytes, kMinChunk);
    if (hipMalloc(&q, size) != hipSuccess) {
      // (a chunk of the minimum size may not fit where the request itself does)
      if (size == bytes || hipMalloc(&q, bytes) != hipSuccess) return nullptr;
      chunks.push_back(Chunk{static_cast<char*>(q), bytes, bytes});
      return q;
    }
    chunks.push_back(Chunk{static_cast<char*>(q), size, bytes});
    return q;
  }
  void reset() {  // a stage has ended: everything it took is free for the next one
    for (Chunk& c : chunks) c.used = 0;
  }
  void release() {
    for (Chunk& c : chunks) (void)hipFree(c.p);
    chunks.clear();
  }
};
struct BvhDeviceArenaScope {
  BvhDeviceArena& a;
  ~BvhDeviceArenaScope() { a.reset(); }
};

// What one build's stages hand to one another in device memory.
struct BvhDeviceBuild {
  int device = 0;
  BvhDeviceArena arena;
  BvhBuildNode* nodes = nullptr;  // the binary tree (buildBinaryTreeOnDevice)
  uint32_t numNodes = 0;
  uint32_t* order = nullptr;      // reference id at every position of the leaf order
  uint32_t numRefs = 0;
  BvhBuildRef* refs = nullptr;    // makeReferencesOnDevice: the references (until the tree builder takes them) ...
  uint32_t numMadeRefs = 0;
  uint32_t* refTri = nullptr;     // ... the triangle of each, and the triangle records (for the packer)
  BvhTri* triRecs = nullptr;
  uint32_t numTris = 0;
  bool collapseHere = false;          // the four-wide collapse happens on the device too: the tree is not read back
  std::vector<uint32_t> levelStart;   // first node of every level of the binary tree, then numNodes (for the collapse)
  void releaseTree() {
    if (nodes) (void)hipFree(nodes);
    if (order) (void)hipFree(order);
    nodes = nullptr;
    order = nullptr;
    numNodes = numRefs = 0;
  }
  void release() {
    releaseTree();
    if (refs) (void)hipFree(refs);
    if (refTri) (void)hipFree(refTri);
    if (triRecs) (void)hipFree(triRecs);
    refs = nullptr;
    refTri = nullptr;
    triRecs = nullptr;
    numMadeRefs = numTris = 0;
    arena.release();
  }
};
BvhDeviceBuild* bvhDeviceBuildBegin(int device, bool collapseOnDevice) {
  BvhDeviceBuild* b = new BvhDeviceBuild();
  b->device = device;
  b->collapseHere = collapseOnDevice;
  return b;
}
void bvhDeviceBuildEnd(BvhDeviceBuild* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  b->release();
  delete b;
}

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
// Slots of at most kSmall references are handled by one wave each (k_small: bounds, bins in LDS, the sweep — no atomics
// on memory); larger ones by the flat launches over the references.  The slot number a reference carries has
// kSmallBit set when its slot is a small one (kNone has it too: "nothing here for the flat launches").
constexpr uint32_t kSmall = 1024, kSmallBit = 0x80000000u;
constexpr uint32_t kMedianWave = 2048;  // median-fallback slots up to this many references find their pivot by one wave's ranking, larger ones by a radix select
enum : uint32_t { ST_LEAF = 0u, ST_SPLIT = 1u, ST_MEDIAN = 2u };
constexpr int kBinWords = 7;  // lo3 hi3 (encoded) + count

#define BDV __device__ __forceinline__
BDV uint32_t enc(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
BDV float dec(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
BDV float areaOf(const float* lo, const float* hi) {  // BvhBox::area
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
  return 2.0f * (dx * dy + dy * dz + dz * dx);
}
BDV uint32_t ceilLog2(uint32_t x) {
  uint32_t l = 0;
  while ((1u << l) < x) l++;
  return l;
}

// The bins of a large slot: ranges of more than kSmall references are disjoint, so first / (kSmall + 1) is unique.
BDV size_t binsOf(uint32_t first) { return (size_t)(first / (kSmall + 1)) * 3 * kBvhBins * kBinWords; }

// One level's active nodes ("slots").  Two of these ping-pong.
struct Level {
  uint32_t* node;    // tree node index of the slot
  uint32_t* first;   // its range of the reference array
  uint32_t* count;
  uint32_t* depth;
  uint32_t* state;   // ST_*
  uint32_t* bnd;     // 12 encoded floats per slot: node box lo3 hi3, centroid box lo3 hi3
  float* lo;         // 3 per slot: bin grid origin per axis (centroid box lo)
  float* scale;      // 3 per slot: kBins / extent, 0 where the axis is not used
  uint32_t* axis;    // SPLIT: best axis
  uint32_t* maxis;   // the median fallback's axis: the widest centroid axis
  uint32_t* split;   // SPLIT: last bin of the left side
  uint32_t* nLeft;
  uint32_t* sel;     // MEDIAN slot of more than kMedianWave references: its row in the radix select's tables (else kNone)
  uint32_t* pivotHi;  // MEDIAN: the pivot key — (encoded centroid on maxis, reference id) of the reference that comes
  uint32_t* pivotLo;  //         count / 2-th in that order: the keys below it go left
  uint32_t* child;   // slot of the left child in the next level (right = +1)
};

struct Tree {  // node arrays (struct of arrays), capacity 2n
  float* box;  // 6 per node
  int32_t* left;
  int32_t* right;
  uint32_t* first;
  uint32_t* count;
  uint32_t* depth;
};

struct Counters {
  uint32_t numSplit;       // slots that get two children this level
  uint32_t numMedianNew;   // MEDIAN slots of this level
  uint32_t numMedianBig;   // of those: too large for the one-wave ranking
  uint32_t pad;
};

__global__ void k_fill_u32(uint32_t* p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_init_root(Level L, Tree T, uint32_t n) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  L.node[0] = 0;
  L.first[0] = 0;
  L.count[0] = n;
  L.depth[0] = 0;
  for (int j = 0; j < 12; j++) L.bnd[j] = ((j % 6) < 3) ? enc(1e30f) : enc(-1e30f);
  T.left[0] = -1;
  T.right[0] = -1;
  T.first[0] = 0;
  T.count[0] = n;
  T.depth[0] = 0;
}

// A block of the flat launches over the references covers kChunk consecutive ones.  Near the root a slot's range spans
// many blocks, and all of a block's atomics would meet on the same few words: so a block first reduces, among its own
// references, those of ONE large slot — that of its first reference that has one — in registers / LDS, and only the
// result goes to memory; references of other large slots (range borders) go straight there.
constexpr uint32_t kChunkPer = 8, kChunk = 256 * kChunkPer;
BDV uint32_t blockSlot(const uint32_t* __restrict__ nodeOf, uint32_t base, uint32_t n, uint32_t* sFirst) {
  if (threadIdx.x == 0) *sFirst = kNone;
  __syncthreads();
  for (uint32_t j = 0; j < kChunkPer; j++) {  // (stops at the first round that finds one: the rounds are in index order)
    const uint32_t k = base + j * 256 + threadIdx.x;
    if (k < n && !(nodeOf[k] & kSmallBit)) atomicMin(sFirst, k - base);
    __syncthreads();
    const uint32_t f = *sFirst;
    __syncthreads();  // (nobody starts the next round's atomics before everybody has read this round's result)
    if (f != kNone) return nodeOf[base + f];
  }
  return kNone;
}

// node box and centroid box of every active node
__global__ __launch_bounds__(256) void k_bounds(const BvhBuildRef* __restrict__ refs, const uint32_t* __restrict__ nodeOf, uint32_t n, Level L) {
  __shared__ uint32_t sFirst;
  const uint32_t base = blockIdx.x * kChunk;
  const uint32_t a0 = blockSlot(nodeOf, base, n, &sFirst);
  if (a0 == kNone) return;
  float v[12];
  for (int j = 0; j < 12; j++) v[j] = ((j % 6) < 3) ? 1e30f : -1e30f;
  bool any = false;
  for (uint32_t i = 0; i < kChunkPer; i++) {
    const uint32_t k = base + i * 256 + threadIdx.x;
    if (k >= n) break;
    const uint32_t a = nodeOf[k];
    if (a & kSmallBit) continue;
    const BvhBuildRef r = refs[k];
    if (a == a0) {
      any = true;
      for (int j = 0; j < 3; j++) {
        v[j] = fminf(v[j], r.box.lo[j]);
        v[3 + j] = fmaxf(v[3 + j], r.box.hi[j]);
        v[6 + j] = fminf(v[6 + j], r.cent[j]);
        v[9 + j] = fmaxf(v[9 + j], r.cent[j]);
      }
    } else {
      uint32_t* b = L.bnd + (size_t)a * 12;
      for (int j = 0; j < 3; j++) {
        atomicMin(&b[j], enc(r.box.lo[j]));
        atomicMax(&b[3 + j], enc(r.box.hi[j]));
        atomicMin(&b[6 + j], enc(r.cent[j]));
        atomicMax(&b[9 + j], enc(r.cent[j]));
      }
    }
  }
  if (__ballot(any) == 0ull) return;
  for (int j = 0; j < 12; j++) {
    float x = v[j];
    for (int off = 32; off > 0; off >>= 1) {
      const float y = __shfl_xor(x, off);
      x = ((j % 6) < 3) ? fminf(x, y) : fmaxf(x, y);
    }
    v[j] = x;
  }
  if ((threadIdx.x & 63u) == 0) {
    uint32_t* b = L.bnd + (size_t)a0 * 12;
    for (int j = 0; j < 12; j++) {
      if ((j % 6) < 3)
        atomicMin(&b[j], enc(v[j]));
      else
        atomicMax(&b[j], enc(v[j]));
    }
  }
}

// per slot: the node's box goes to the tree; leaf / forced median / split candidate; the bin grid (splitNode)
__global__ void k_prepare(Level L, Tree T, uint32_t nA, Counters* C, uint32_t* __restrict__ bins) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA || L.count[a] <= kSmall) return;
  const uint32_t* b = L.bnd + (size_t)a * 12;
  float nb[6], cb[6];
  for (int j = 0; j < 6; j++) {
    nb[j] = dec(b[j]);
    cb[j] = dec(b[6 + j]);
  }
  const uint32_t node = L.node[a], count = L.count[a], depth = L.depth[a];
  for (int j = 0; j < 6; j++) T.box[(size_t)node * 6 + j] = nb[j];
  L.sel[a] = kNone;
  L.nLeft[a] = 0;
  L.child[a] = kNone;
  if (count <= kBvhLeafMax) {
    L.state[a] = ST_LEAF;
    return;
  }
  const bool forceMedian = depth + ceilLog2((count + kBvhLeafMax - 1) / kBvhLeafMax) + 1 >= (uint32_t)kBvhBinaryMaxDepth;
  for (int axis = 0; axis < 3; axis++) {
    const float ext = cb[3 + axis] - cb[axis];
    L.lo[(size_t)a * 3 + axis] = cb[axis];
    L.scale[(size_t)a * 3 + axis] = (ext > 0.0f) ? (float)kBvhBins / ext : 0.0f;
  }
  // the median fallback's axis: the widest centroid axis (splitNode)
  int maxis = 0;
  const float e0 = cb[3] - cb[0], e1 = cb[4] - cb[1], e2 = cb[5] - cb[2];
  if (e1 > e0 && e1 >= e2) maxis = 1;
  if (e2 > e0 && e2 > e1) maxis = 2;
  L.axis[a] = (uint32_t)maxis;
  L.maxis[a] = (uint32_t)maxis;
  L.split[a] = 0;
  if (forceMedian) {
    L.state[a] = ST_MEDIAN;
    atomicAdd(&C->numMedianNew, 1u);
    if (count > kMedianWave) atomicAdd(&C->numMedianBig, 1u);
  } else {
    L.state[a] = ST_SPLIT;
    uint32_t* w = bins + binsOf(L.first[a]);
    for (int i = 0; i < 3 * kBvhBins * kBinWords; i++) {
      const int j = i % kBinWords;
      w[i] = j < 3 ? enc(1e30f) : (j < 6 ? enc(-1e30f) : 0u);
    }
  }
}

__global__ __launch_bounds__(256) void k_bin(const BvhBuildRef* __restrict__ refs, const uint32_t* __restrict__ nodeOf, uint32_t n, Level L,
                                             uint32_t* __restrict__ bins) {
  __shared__ uint32_t sFirst;
  __shared__ uint32_t sBins[3 * kBvhBins * kBinWords];
  const uint32_t base = blockIdx.x * kChunk;
  uint32_t a0 = blockSlot(nodeOf, base, n, &sFirst);
  if (a0 == kNone) return;
  for (uint32_t i = threadIdx.x; i < 3 * kBvhBins * kBinWords; i += 256) {
    const uint32_t j = i % kBinWords;
    sBins[i] = j < 3 ? enc(1e30f) : (j < 6 ? enc(-1e30f) : 0u);
  }
  __syncthreads();
  float lo0[3], sc0[3];
  const bool split0 = L.state[a0] == ST_SPLIT;
  for (int axis = 0; axis < 3; axis++) {
    lo0[axis] = L.lo[(size_t)a0 * 3 + axis];
    sc0[axis] = split0 ? L.scale[(size_t)a0 * 3 + axis] : 0.0f;
  }
  for (uint32_t i = 0; i < kChunkPer; i++) {
    const uint32_t k = base + i * 256 + threadIdx.x;
    if (k >= n) break;
    const uint32_t a = nodeOf[k];
    if (a & kSmallBit) continue;
    if (a == a0) {
      if (!split0) continue;
      const BvhBuildRef r = refs[k];
      for (int axis = 0; axis < 3; axis++) {
        if (!(sc0[axis] > 0.0f)) continue;  // (use[axis] = ext > 0)
        int bi = (int)((r.cent[axis] - lo0[axis]) * sc0[axis]);
        bi = bi < 0 ? 0 : (bi > kBvhBins - 1 ? kBvhBins - 1 : bi);
        uint32_t* w = sBins + (axis * kBvhBins + bi) * kBinWords;
        for (int j = 0; j < 3; j++) {
          atomicMin(&w[j], enc(r.box.lo[j]));
          atomicMax(&w[3 + j], enc(r.box.hi[j]));
        }
        atomicAdd(&w[6], 1u);
      }
    } else {
      if (L.state[a] != ST_SPLIT) continue;
      const BvhBuildRef r = refs[k];
      for (int axis = 0; axis < 3; axis++) {
        const float sc = L.scale[(size_t)a * 3 + axis];
        if (!(sc > 0.0f)) continue;
        int bi = (int)((r.cent[axis] - L.lo[(size_t)a * 3 + axis]) * sc);
        bi = bi < 0 ? 0 : (bi > kBvhBins - 1 ? kBvhBins - 1 : bi);
        uint32_t* w = bins + binsOf(L.first[a]) + ((size_t)axis * kBvhBins + (size_t)bi) * kBinWords;
        for (int j = 0; j < 3; j++) {
          atomicMin(&w[j], enc(r.box.lo[j]));
          atomicMax(&w[3 + j], enc(r.box.hi[j]));
        }
        atomicAdd(&w[6], 1u);
      }
    }
  }
  __syncthreads();
  if (!split0) return;
  uint32_t* g = bins + binsOf(L.first[a0]);
  for (uint32_t i = threadIdx.x; i < 3 * kBvhBins * kBinWords; i += 256) {
    const uint32_t j = i % kBinWords;
    if (sBins[i - j + 6] == 0u) continue;  // empty bin
    if (j < 3)
      atomicMin(&g[i], sBins[i]);
    else if (j < 6)
      atomicMax(&g[i], sBins[i]);
    else
      atomicAdd(&g[i], sBins[i]);
  }
}

// One wave per small slot: what k_bounds, k_prepare, k_bin and k_decide do for the large ones — the same values (minima,
// maxima and counts do not depend on the order; the sweep is the same float operations in the same order, one lane per
// axis, the axes then compared in their order) — without an atomic on memory.
__global__ __launch_bounds__(64) void k_small(const BvhBuildRef* __restrict__ refs, Level L, Tree T, uint32_t nA, Counters* C) {
  __shared__ uint32_t sBins[3 * kBvhBins * kBinWords];
  const uint32_t a = blockIdx.x, lane = threadIdx.x;
  if (a >= nA) return;
  const uint32_t count = L.count[a];
  if (count > kSmall) return;
  const uint32_t first = L.first[a];
  float v[12];
  for (int j = 0; j < 12; j++) v[j] = ((j % 6) < 3) ? 1e30f : -1e30f;
  for (uint32_t i = lane; i < count; i += 64) {
    const BvhBuildRef r = refs[first + i];
    for (int j = 0; j < 3; j++) {
      v[j] = fminf(v[j], r.box.lo[j]);
      v[3 + j] = fmaxf(v[3 + j], r.box.hi[j]);
      v[6 + j] = fminf(v[6 + j], r.cent[j]);
      v[9 + j] = fmaxf(v[9 + j], r.cent[j]);
    }
  }
  for (int j = 0; j < 12; j++) {
    float x = v[j];
    for (int off = 32; off > 0; off >>= 1) {
      const float y = __shfl_xor(x, off);
      x = ((j % 6) < 3) ? fminf(x, y) : fmaxf(x, y);
    }
    v[j] = x;
  }
  const uint32_t node = L.node[a], depth = L.depth[a];
  if (lane == 0) {
    for (int j = 0; j < 6; j++) T.box[(size_t)node * 6 + j] = v[j];
    L.sel[a] = kNone;
    L.nLeft[a] = 0;
    L.child[a] = kNone;
  }
  if (count <= kBvhLeafMax) {
    if (lane == 0) L.state[a] = ST_LEAF;
    return;
  }
  const float* cb = v + 6;
  float lo[3], sc[3];
  for (int axis = 0; axis < 3; axis++) {
    const float ext = cb[3 + axis] - cb[axis];
    lo[axis] = cb[axis];
    sc[axis] = (ext > 0.0f) ? (float)kBvhBins / ext : 0.0f;
  }
  int maxis = 0;
  const float e0 = cb[3] - cb[0], e1 = cb[4] - cb[1], e2 = cb[5] - cb[2];
  if (e1 > e0 && e1 >= e2) maxis = 1;
  if (e2 > e0 && e2 > e1) maxis = 2;
  if (lane == 0) {
    for (int axis = 0; axis < 3; axis++) {
      L.lo[(size_t)a * 3 + axis] = lo[axis];
      L.scale[(size_t)a * 3 + axis] = sc[axis];
    }
    L.maxis[a] = (uint32_t)maxis;
  }
  const bool forceMedian = depth + ceilLog2((count + kBvhLeafMax - 1) / kBvhLeafMax) + 1 >= (uint32_t)kBvhBinaryMaxDepth;
  if (forceMedian) {
    if (lane == 0) {
      L.axis[a] = (uint32_t)maxis;
      L.split[a] = 0;
      L.state[a] = ST_MEDIAN;
      atomicAdd(&C->numMedianNew, 1u);
    }
    return;
  }
  for (uint32_t i = lane; i < 3 * kBvhBins * kBinWords; i += 64) {
    const uint32_t j = i % kBinWords;
    sBins[i] = j < 3 ? enc(1e30f) : (j < 6 ? enc(-1e30f) : 0u);
  }
  __syncthreads();
  for (uint32_t i = lane; i < count; i += 64) {
    const BvhBuildRef r = refs[first + i];
    for (int axis = 0; axis < 3; axis++) {
      if (!(sc[axis] > 0.0f)) continue;  // (use[axis] = ext > 0)
      int bi = (int)((r.cent[axis] - lo[axis]) * sc[axis]);
      bi = bi < 0 ? 0 : (bi > kBvhBins - 1 ? kBvhBins - 1 : bi);
      uint32_t* w = sBins + (axis * kBvhBins + bi) * kBinWords;
      for (int j = 0; j < 3; j++) {
        atomicMin(&w[j], enc(r.box.lo[j]));
        atomicMax(&w[3 + j], enc(r.box.hi[j]));
      }
      atomicAdd(&w[6], 1u);
    }
  }
  __syncthreads();
  // the sweep: lane `axis` does its axis
  int mySplit = -1;
  float myCost = 1e30f;
  const float mySc = lane == 0 ? sc[0] : (lane == 1 ? sc[1] : sc[2]);
  if (lane < 3 && mySc > 0.0f) {
    const uint32_t* w = sBins + lane * kBvhBins * kBinWords;
    float rightArea[kBvhBins];
    uint32_t rightCnt[kBvhBins];
    float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
    uint32_t cnt = 0;
    for (int b = kBvhBins - 1; b > 0; b--) {
      for (int j = 0; j < 3; j++) {
        const float bl = dec(w[b * kBinWords + j]), bh = dec(w[b * kBinWords + 3 + j]);
        blo[j] = bl < blo[j] ? bl : blo[j];
        bhi[j] = bhi[j] < bh ? bh : bhi[j];
      }
      cnt += w[b * kBinWords + 6];
      rightArea[b] = areaOf(blo, bhi);
      rightCnt[b] = cnt;
    }
    for (int j = 0; j < 3; j++) {
      blo[j] = 1e30f;
      bhi[j] = -1e30f;
    }
    cnt = 0;
    for (int b = 0; b < kBvhBins - 1; b++) {
      for (int j = 0; j < 3; j++) {
        const float bl = dec(w[b * kBinWords + j]), bh = dec(w[b * kBinWords + 3 + j]);
        blo[j] = bl < blo[j] ? bl : blo[j];
        bhi[j] = bhi[j] < bh ? bh : bhi[j];
      }
      cnt += w[b * kBinWords + 6];
      if (cnt == 0 || rightCnt[b + 1] == 0) continue;
      const float cost = areaOf(blo, bhi) * (float)cnt + rightArea[b + 1] * (float)rightCnt[b + 1];
      if (cost < myCost) {
        myCost = cost;
        mySplit = b;
      }
    }
  }
  int bestAxis = -1, bestSplit = -1;
  float bestCost = 1e30f;
  for (int axis = 0; axis < 3; axis++) {
    const float c = __shfl(myCost, axis);
    const int sp = __shfl(mySplit, axis);
    if (sp >= 0 && c < bestCost) {
      bestCost = c;
      bestAxis = axis;
      bestSplit = sp;
    }
  }
  if (lane == 0) {
    if (bestAxis >= 0) {
      L.axis[a] = (uint32_t)bestAxis;
      L.split[a] = (uint32_t)bestSplit;
      L.state[a] = ST_SPLIT;
    } else {
      L.axis[a] = (uint32_t)maxis;
      L.split[a] = 0;
      L.state[a] = ST_MEDIAN;
      atomicAdd(&C->numMedianNew, 1u);
    }
  }
}

// the SAH sweep of splitNode: the same float operations in the same order
__global__ void k_decide(Level L, uint32_t nA, const uint32_t* __restrict__ bins, Counters* C) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA || L.count[a] <= kSmall || L.state[a] != ST_SPLIT) return;
  int bestAxis = -1, bestSplit = -1;
  float bestCost = 1e30f;
  for (int axis = 0; axis < 3; axis++) {
    if (!(L.scale[(size_t)a * 3 + axis] > 0.0f)) continue;
    const uint32_t* w = bins + binsOf(L.first[a]) + (size_t)axis * kBvhBins * kBinWords;
    float rightArea[kBvhBins];
    uint32_t rightCnt[kBvhBins];
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    uint32_t cnt = 0;
    for (int b = kBvhBins - 1; b > 0; b--) {
      for (int j = 0; j < 3; j++) {
        const float bl = dec(w[b * kBinWords + j]), bh = dec(w[b * kBinWords + 3 + j]);
        lo[j] = bl < lo[j] ? bl : lo[j];
        hi[j] = hi[j] < bh ? bh : hi[j];
      }
      cnt += w[b * kBinWords + 6];
      rightArea[b] = areaOf(lo, hi);
      rightCnt[b] = cnt;
    }
    for (int j = 0; j < 3; j++) {
      lo[j] = 1e30f;
      hi[j] = -1e30f;
    }
    cnt = 0;
    for (int b = 0; b < kBvhBins - 1; b++) {
      for (int j = 0; j < 3; j++) {
        const float bl = dec(w[b * kBinWords + j]), bh = dec(w[b * kBinWords + 3 + j]);
        lo[j] = bl < lo[j] ? bl : lo[j];
        hi[j] = hi[j] < bh ? bh : hi[j];
      }
      cnt += w[b * kBinWords + 6];
      if (cnt == 0 || rightCnt[b + 1] == 0) continue;
      const float cost = areaOf(lo, hi) * (float)cnt + rightArea[b + 1] * (float)rightCnt[b + 1];
      if (cost < bestCost) {
        bestCost = cost;
        bestAxis = axis;
        bestSplit = b;
      }
    }
  }
  if (bestAxis >= 0) {
    L.axis[a] = (uint32_t)bestAxis;
    L.split[a] = (uint32_t)bestSplit;
  } else {  // no split position: the median fallback (the axis prepared above)
    L.state[a] = ST_MEDIAN;
    atomicAdd(&C->numMedianNew, 1u);
    if (L.count[a] > kMedianWave) atomicAdd(&C->numMedianBig, 1u);
  }
}

__global__ __launch_bounds__(256) void k_flags(const BvhBuildRef* __restrict__ refs, const uint32_t* __restrict__ nodeOf, uint32_t n, Level L,
                                               uint32_t* __restrict__ F) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t f = 0;
  uint32_t a = nodeOf[k];
  if (a != kNone) {
    a &= ~kSmallBit;
    const uint32_t st = L.state[a];
    if (st == ST_SPLIT) {
      const uint32_t axis = L.axis[a];
      int b = (int)((refs[k].cent[axis] - L.lo[(size_t)a * 3 + axis]) * L.scale[(size_t)a * 3 + axis]);
      b = b < 0 ? 0 : (b > kBvhBins - 1 ? kBvhBins - 1 : b);
      f = b <= (int)L.split[a] ? 1u : 0u;
    } else if (st == ST_MEDIAN) {
      const BvhBuildRef r = refs[k];
      const uint32_t hi = enc(r.cent[L.maxis[a]]), pHi = L.pivotHi[a];
      f = (hi < pHi || (hi == pHi && r.id < L.pivotLo[a])) ? 1u : 0u;  // (the pivot is known by the time this matters)
    }
  }
  F[k] = f;
}

// ---- exclusive scan of a uint32 array (three launches: block sums, scan of the sums by one block, local scans) ----
constexpr uint32_t kScanBlock = 256, kScanPer = 8, kScanTile = kScanBlock * kScanPer;
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t s[kScanBlock];
  const size_t base = (size_t)blockIdx.x * kScanTile;
  uint32_t acc = 0;
  for (uint32_t j = 0; j < kScanPer; j++) {
    const size_t i = base + (size_t)j * kScanBlock + threadIdx.x;
    if (i < n) acc += in[i];
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t off = kScanBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(kScanBlock) void k_scan_of_sums(uint32_t* sums, uint32_t m, uint32_t* total) {  // one block
  __shared__ uint32_t s[kScanBlock];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < m; base += kScanBlock) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < m ? sums[i] : 0u;
    s[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < kScanBlock; off <<= 1) {  // inclusive Hillis-Steele
      const uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < m) sums[i] = carry + s[threadIdx.x] - v;  // exclusive
    __syncthreads();
    if (threadIdx.x == kScanBlock - 1) carry += s[kScanBlock - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_local(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ sums,
                                                         const uint32_t* __restrict__ total, uint32_t* __restrict__ out) {
  __shared__ uint32_t s[kScanBlock];
  const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanPer;  // each thread owns kScanPer consecutive elements
  uint32_t v[kScanPer], acc = 0;
  for (uint32_t j = 0; j < kScanPer; j++) {
    v[j] = (base + j < n) ? in[base + j] : 0u;
    acc += v[j];
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t off = 1; off < kScanBlock; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + s[threadIdx.x] - acc;
  for (uint32_t j = 0; j < kScanPer; j++) {
    if (base + j < n) out[base + j] = run;
    run += v[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// lefts of every partitioned slot; a SAH split that puts everything on one side falls back to the median
__global__ void k_nleft(Level L, uint32_t nA, const uint32_t* __restrict__ S, Counters* C) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA) return;
  const uint32_t st = L.state[a];
  if (st == ST_LEAF) return;
  const uint32_t first = L.first[a], count = L.count[a];
  const uint32_t nl = S[first + count] - S[first];
  L.nLeft[a] = nl;
  if (st == ST_SPLIT && (nl == 0 || nl == count)) {
    L.state[a] = ST_MEDIAN;
    atomicAdd(&C->numMedianNew, 1u);
    if (count > kMedianWave) atomicAdd(&C->numMedianBig, 1u);
  }
}

// The median fallback's pivot: the reference that comes count / 2-th by (centroid on the slot's axis, reference id).
// Slots of at most kMedianWave references: one wave ranks every reference against all the others.
__global__ __launch_bounds__(64) void k_median_pivot_small(const BvhBuildRef* __restrict__ refs, Level L, uint32_t nA) {
  const uint32_t a = blockIdx.x;
  if (a >= nA || L.state[a] != ST_MEDIAN || L.count[a] > kMedianWave) return;
  const uint32_t first = L.first[a], count = L.count[a], axis = L.maxis[a], mid = count / 2;
  for (uint32_t i = threadIdx.x; i < count; i += 64) {
    const float ki = refs[first + i].cent[axis];
    const uint32_t idi = refs[first + i].id;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < count; j++) {
      const float kj = refs[first + j].cent[axis];
      const uint32_t idj = refs[first + j].id;
      rank += (kj < ki || (kj == ki && idj < idi)) ? 1u : 0u;
    }
    if (rank == mid) {
      L.pivotHi[a] = enc(ki);
      L.pivotLo[a] = idi;
    }
  }
}
// Larger slots: a radix select over the 64-bit key, most significant byte first — per pass a histogram of the byte
// among the keys that match the prefix found so far (k_sel_hist), then the bin the wanted rank falls in (k_sel_pick).
__global__ void k_median_list_big(Level L, uint32_t nA, uint32_t* list, uint32_t* listCount, uint32_t* selHi, uint32_t* selLo, uint32_t* selK) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA || L.state[a] != ST_MEDIAN || L.count[a] <= kMedianWave) return;
  const uint32_t j = atomicAdd(listCount, 1u);
  list[j] = a;
  L.sel[a] = j;
  selHi[j] = 0;
  selLo[j] = 0;
  selK[j] = L.count[a] / 2;
}
__global__ __launch_bounds__(256) void k_sel_hist(const BvhBuildRef* __restrict__ refs, const uint32_t* __restrict__ nodeOf, uint32_t n, Level L, int pass,
                                                  const uint32_t* __restrict__ selHi, const uint32_t* __restrict__ selLo, uint32_t* __restrict__ hist) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t a = nodeOf[k];
  if (a == kNone) return;
  a &= ~kSmallBit;
  if (L.state[a] != ST_MEDIAN) return;
  const uint32_t j = L.sel[a];
  if (j == kNone) return;
  const BvhBuildRef r = refs[k];
  const uint32_t hi = enc(r.cent[L.maxis[a]]), lo = r.id;
  // the bytes above byte `pass` must be the prefix found so far
  bool match;
  uint32_t digit;
  if (pass >= 4) {
    const int sh = 8 * (pass - 4);
    match = sh == 24 || (hi >> (sh + 8)) == (selHi[j] >> (sh + 8));
    digit = (hi >> sh) & 255u;
  } else {
    const int sh = 8 * pass;
    match = hi == selHi[j] && (sh == 24 || (lo >> (sh + 8)) == (selLo[j] >> (sh + 8)));
    digit = (lo >> sh) & 255u;
  }
  if (match) atomicAdd(&hist[(size_t)j * 256 + digit], 1u);
}
__global__ void k_sel_pick(uint32_t m, int pass, uint32_t* selHi, uint32_t* selLo, uint32_t* selK, uint32_t* hist, Level L, const uint32_t* __restrict__ list) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  uint32_t* h = hist + (size_t)j * 256;
  uint32_t kk = selK[j], d = 0;
  for (; d < 255; d++) {
    if (kk < h[d]) break;
    kk -= h[d];
  }
  for (uint32_t i = 0; i < 256; i++) h[i] = 0;
  selK[j] = kk;
  if (pass >= 4)
    selHi[j] |= d << (8 * (pass - 4));
  else
    selLo[j] |= d << (8 * pass);
  if (pass == 0) {
    L.pivotHi[list[j]] = selHi[j];
    L.pivotLo[list[j]] = selLo[j];
  }
}

// slots that split: G[a] = 1 (scanned into the children's slots)
__global__ void k_split_flags(Level L, uint32_t nA, uint32_t* G) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < nA) G[a] = L.state[a] != ST_LEAF ? 1u : 0u;
}
// two children per splitting slot, in slot order: tree nodes nodeBase + 2 rank, + 1; slots 2 rank, + 1 of the next level
__global__ void k_children(Level L, uint32_t nA, const uint32_t* __restrict__ R, Level N, Tree T, uint32_t nodeBase) {
  const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nA || L.state[a] == ST_LEAF) return;
  const uint32_t rank = R[a], node = L.node[a];
  const uint32_t first = L.first[a], count = L.count[a], nl = L.nLeft[a], depth = L.depth[a];
  L.child[a] = 2 * rank;
  for (int side = 0; side < 2; side++) {
    const uint32_t c = nodeBase + 2 * rank + (uint32_t)side, s = 2 * rank + (uint32_t)side;
    const uint32_t cf = side ? first + nl : first, cc = side ? count - nl : nl;
    T.left[c] = -1;
    T.right[c] = -1;
    T.first[c] = cf;
    T.count[c] = cc;
    T.depth[c] = depth + 1;
    N.node[s] = c;
    N.first[s] = cf;
    N.count[s] = cc;
    N.depth[s] = depth + 1;
    for (int j = 0; j < 12; j++) N.bnd[(size_t)s * 12 + j] = ((j % 6) < 3) ? enc(1e30f) : enc(-1e30f);
  }
  T.left[node] = (int32_t)(nodeBase + 2 * rank);
  T.right[node] = (int32_t)(nodeBase + 2 * rank + 1);
  T.count[node] = 0;  // (interior: bvh_build.cpp sets it so)
}

// stable partition of every splitting slot's range; everything else is carried over
__global__ __launch_bounds__(256) void k_scatter(const BvhBuildRef* __restrict__ refs, const uint32_t* __restrict__ nodeOf, uint32_t n, Level L,
                                                 const uint32_t* __restrict__ F, const uint32_t* __restrict__ S,
                                                 BvhBuildRef* __restrict__ refsOut, uint32_t* __restrict__ nodeOfOut) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t a = nodeOf[k];
  if (a != kNone) a &= ~kSmallBit;
  if (a == kNone || L.state[a] == ST_LEAF) {
    refsOut[k] = refs[k];
    nodeOfOut[k] = kNone;
    return;
  }
  const uint32_t first = L.first[a], nl = L.nLeft[a];
  const uint32_t l = S[k] - S[first];
  const uint32_t f = F[k];
  const uint32_t dest = f ? first + l : first + nl + ((k - first) - l);
  refsOut[dest] = refs[k];
  const uint32_t childCount = f ? nl : L.count[a] - nl;
  nodeOfOut[dest] = (L.child[a] + (f ? 0u : 1u)) | (childCount <= kSmall ? kSmallBit : 0u);
}

// the results in the host's formats: the leaf order as reference ids, the nodes as BvhBuildNode records
__global__ void k_order(const BvhBuildRef* __restrict__ refs, uint32_t n, uint32_t* __restrict__ order) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) order[k] = refs[k].id;
}
__global__ void k_pack_nodes(Tree T, uint32_t numNodes, BvhBuildNode* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= numNodes) return;
  BvhBuildNode nd;
  for (int j = 0; j < 3; j++) {
    nd.box.lo[j] = T.box[(size_t)i * 6 + j];
    nd.box.hi[j] = T.box[(size_t)i * 6 + 3 + j];
  }
  nd.left = T.left[i];
  nd.right = T.right[i];
  nd.first = T.first[i];
  nd.count = T.count[i];
  nd.depth = T.depth[i];
  out[i] = nd;
}

// ------------------------------------------------------------------------------------------------
// Quantise + pack on the device: what bvh_build.cpp does between "collapse" and packBvh, one thread per four-wide node,
// the same float operations (subtractions, a division by a power of two, floor / ceil, comparisons).
// ------------------------------------------------------------------------------------------------
// child boxes -> origin, scale exponents, 8-bit planes, leaf bits, child offsets: everything of the node's record but
// where it and its child block go; blockSize[i] = records in node i's child block
__global__ void k_quantise(const BvhWideNode* __restrict__ wide, uint32_t nWide, const BvhBuildNode* __restrict__ nodes, float pad,
                           BvhRec* __restrict__ recTmp, uint32_t* __restrict__ blockSize) {
  const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= nWide) return;
  const BvhWideNode w = wide[wi];
  float blo[3] = {1e30f, 1e30f, 1e30f}, bhi[3] = {-1e30f, -1e30f, -1e30f};
  float clo[4][3], chi[4][3];
  uint32_t leafBits = 0, offs = 0, off = 0;
  for (int k = 0; k < 4; k++) {
    if (k >= w.nk) break;
    const BvhBuildNode c = nodes[w.kids[k]];
    for (int a = 0; a < 3; a++) {
      clo[k][a] = c.box.lo[a] - pad;
      chi[k][a] = c.box.hi[a] + pad;
      blo[a] = clo[k][a] < blo[a] ? clo[k][a] : blo[a];  // std::min(blo, clo)
      bhi[a] = bhi[a] < chi[k][a] ? chi[k][a] : bhi[a];  // std::max(bhi, chi)
    }
    offs |= off << (8 * k);
    if (c.left < 0) {
      leafBits |= 1u << k;
      off += c.count;
    } else {
      off += 1;
    }
  }
  BvhRec rec;
  for (int i = 0; i < 12; i++) rec.w[i] = 0;
  uint32_t ex[3];
  for (int a = 0; a < 3; a++) {
    rec.w[a] = __float_as_uint(blo[a]);
    const float d = bhi[a] - blo[a];
    const float ext = d < 1e-30f ? 1e-30f : d;  // std::max(d, 1e-30f)
    // frexp(ext / 254): ext / 254 = m 2^e, m in [0.5, 1) — a normal number here, so e = exponent field - 126
    const int e = (int)((__float_as_uint(ext / 254.0f) >> 23) & 0xffu) - 126;
    int biased = e + 127;
    if (biased < 1) biased = 1;
    if (biased > 254) biased = 254;
    ex[a] = (uint32_t)biased;
    const float sc = __uint_as_float((uint32_t)biased << 23);
    uint32_t lo4 = 0, hi4 = 0;
    for (int k = 0; k < 4; k++) {
      if (k >= w.nk) {
        lo4 |= 255u << (8 * k);
        continue;
      }
      int ql = (int)floorf((clo[k][a] - blo[a]) / sc);
      ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
      while (ql > 0 && blo[a] + (float)ql * sc > clo[k][a]) ql--;
      int qh = (int)ceilf((chi[k][a] - blo[a]) / sc);
      qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
      while (qh < 255 && blo[a] + (float)qh * sc < chi[k][a]) qh++;
      lo4 |= (uint32_t)ql << (8 * k);
      hi4 |= (uint32_t)qh << (8 * k);
    }
    rec.w[4 + a] = lo4;
    rec.w[7 + a] = hi4;
  }
  rec.w[3] = ex[0] | (ex[1] << 8) | (ex[2] << 16) | (leafBits << 24);
  rec.w[11] = offs;
  recTmp[wi] = rec;
  blockSize[wi] = off;
}
// where every node's record goes: into its parent's child block (the root: record 0)
__global__ void k_node_pos(const BvhSlot* __restrict__ slots, uint32_t nWide, const BvhRec* __restrict__ recTmp, const uint32_t* __restrict__ baseExcl,
                           uint32_t* __restrict__ pos) {
  const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= nWide) return;
  if (wi == 0) {
    pos[0] = 0;
    return;
  }
  const BvhSlot s = slots[wi];
  pos[wi] = 1u + baseExcl[s.node] + ((recTmp[s.node].w[11] >> (8 * s.idx)) & 0xffu);
}
__global__ void k_write_recs(const BvhWideNode* __restrict__ wide, uint32_t nWide, const BvhBuildNode* __restrict__ nodes, const BvhRec* __restrict__ recTmp,
                             const uint32_t* __restrict__ baseExcl, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ order,
                             const uint32_t* __restrict__ refTri, const BvhTri* __restrict__ triRecs, BvhRec* __restrict__ recs) {
  const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= nWide) return;
  const BvhWideNode w = wide[wi];
  BvhRec rec = recTmp[wi];
  const uint32_t base = 1u + baseExcl[wi];
  rec.w[10] = base;
  recs[pos[wi]] = rec;
  for (int k = 0; k < 4; k++) {
    if (k >= w.nk || !((rec.w[3] >> (24 + k)) & 1u)) continue;
    const uint32_t first = nodes[w.kids[k]].first, count = nodes[w.kids[k]].count;
    const uint32_t at = base + ((rec.w[11] >> (8 * k)) & 0xffu);
    for (uint32_t j = 0; j < count; j++) {
      BvhTri t = triRecs[refTri[order[first + j]]];
      if (j + 1 == count) t.flags |= kTriLastOfLeaf;
      BvhRec out;
      static_assert(sizeof(BvhTri) == sizeof(BvhRec), "one record per triangle");
      __builtin_memcpy(&out, &t, sizeof(out));
      recs[at + j] = out;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// References on the device: bvh_build.cpp's wholePiece + splitTriangle and alpha_clip.cpp's clip(), one thread per
// triangle — the same double-precision operations in the same order (no contraction: the Makefile's -ffp-contract=off),
// so the same pieces, the same boxes, in the same order.
// ------------------------------------------------------------------------------------------------
struct DevPiece {
  double b[kBvhPolyMax][2];
  int n;
  uint32_t splits;
  float lo[3], hi[3];
};
struct DevMask {
  int32_t w, h;
  const uint32_t* mayPass;
};
struct RefArgs {
  const BvhTri* triRecs;
  const BvhBox* triBox;
  const uint32_t* splits;
  const uint8_t* state;
  uint32_t numTris;
  double gridLo[3], gridExt[3];
  int haveClipper;
  const uint32_t* triMaterial;
  const uint32_t* indices;
  const float* texcoords;  // may be null
  const int32_t* matMask;
  const int32_t* matVerdict;
  const DevMask* masks;
};

// buildBvh "Triangle records" (bvh_build.cpp) on the device: v0, e1 = v1 - v0, e2 = v2 - v0, ids; the box of the five
// points the host takes (v0, v0 + e1, v0 + e2, v1, v2), with its min / max spelled as BvhBox::grow spells them.
__global__ void k_tri_recs(const float* __restrict__ pos, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ flags,
                           const uint32_t* __restrict__ aux, uint32_t n, BvhTri* __restrict__ recs, BvhBox* __restrict__ boxes) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float* a = pos + (size_t)idx[(size_t)t * 3] * 3;
  const float* b = pos + (size_t)idx[(size_t)t * 3 + 1] * 3;
  const float* c = pos + (size_t)idx[(size_t)t * 3 + 2] * 3;
  BvhTri r;
  BvhBox bx;
  for (int k = 0; k < 3; k++) {
    const float va = a[k], vb = b[k], vc = c[k];
    r.v0[k] = va;
    r.e1[k] = vb - va;
    r.e2[k] = vc - va;
    const float p1 = r.v0[k] + r.e1[k], p2 = r.v0[k] + r.e2[k];
    float lo = 1e30f, hi = -1e30f;
    const float pts[5] = {va, p1, p2, vb, vc};
    for (int j = 0; j < 5; j++) {
      lo = pts[j] < lo ? pts[j] : lo;
      hi = hi < pts[j] ? pts[j] : hi;
    }
    bx.lo[k] = lo;
    bx.hi[k] = hi;
  }
  r.prim = t;
  r.flags = flags ? flags[t] : 0u;
  r.aux = aux ? aux[t] : 0u;
  recs[t] = r;
  boxes[t] = bx;
}

BDV float floatDownD(double x) {
  float f = (float)x;
  if ((double)f > x) {  // nextafterf(f, -inf)
    const uint32_t u = __float_as_uint(f);
    f = (f == 0.0f) ? -__uint_as_float(1u) : __uint_as_float((u & 0x80000000u) ? u + 1u : u - 1u);
  }
  return f;
}
BDV float floatUpD(double x) {
  float f = (float)x;
  if ((double)f < x) {  // nextafterf(f, +inf)
    const uint32_t u = __float_as_uint(f);
    f = (f == 0.0f) ? __uint_as_float(1u) : __uint_as_float((u & 0x80000000u) ? u - 1u : u + 1u);
  }
  return f;
}
BDV double dmin(double a, double b) { return b < a ? b : a; }  // std::min(a, b)
BDV double dmax(double a, double b) { return a < b ? b : a; }  // std::max(a, b)

// Sutherland-Hodgman against the closed half-plane A + B bu + C bv <= 0 (clipHalfPlane)
__device__ int devClipHalfPlane(const double (*in)[2], int n, double A, double B, double C, double (*out)[2]) {
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
__device__ int devClipInPlace(double (*poly)[2], int n, double A, double B, double C) {
  double out[kBvhPolyMax][2];
  const int m = devClipHalfPlane(poly, n, A, B, C, out);
  for (int k = 0; k < m; k++) {
    poly[k][0] = out[k][0];
    poly[k][1] = out[k][1];
  }
  return m;
}
__device__ double devPolyArea2(const double (*b)[2], int n) {
  double s = 0;
  for (int k = 0; k < n; k++) {
    const double* p = b[k];
    const double* q = b[(k + 1) % n];
    s += p[0] * q[1] - q[0] * p[1];
  }
  return fabs(s);
}
// intersectBox(polyBox(r, b, n), outer)
__device__ void devPolyBoxIn(const BvhTri& r, const double (*b)[2], int n, const float* outerLo, const float* outerHi, float* lo, float* hi) {
  double dlo[3] = {1e300, 1e300, 1e300}, dhi[3] = {-1e300, -1e300, -1e300};
  for (int k = 0; k < n; k++)
    for (int a = 0; a < 3; a++) {
      const double p = (double)r.v0[a] + b[k][0] * (double)r.e1[a] + b[k][1] * (double)r.e2[a];
      dlo[a] = dmin(dlo[a], p);
      dhi[a] = dmax(dhi[a], p);
    }
  for (int a = 0; a < 3; a++) {
    const float pl = floatDownD(dlo[a]), ph = floatUpD(dhi[a]);
    lo[a] = pl < outerLo[a] ? outerLo[a] : pl;  // std::max(a.lo, b.lo)
    hi[a] = outerHi[a] < ph ? outerHi[a] : ph;  // std::min(a.hi, b.hi)
    if (hi[a] < lo[a]) hi[a] = lo[a];
  }
}
// SplitGrid::plane
__device__ int devPlane(const RefArgs& A, int axis, float a, float b, double& coord) {
  if (!(A.gridExt[axis] > 0.0) || !(b > a)) return -1;
  const double s = 1073741824.0 / A.gridExt[axis];
  double ua = floor(((double)a - A.gridLo[axis]) * s), ub = floor(((double)b - A.gridLo[axis]) * s);
  ua = dmin(dmax(ua, 0.0), 1073741823.0);
  ub = dmin(dmax(ub, 0.0), 1073741823.0);
  const uint32_t ia = (uint32_t)ua, ib = (uint32_t)ub;
  if (ia == ib) return -1;
  const uint32_t diff = ia ^ ib;
  const int h = 31 - __clz((int)diff);
  const uint32_t pl = (ib >> h) << h;
  coord = A.gridLo[axis] + (double)pl / s;
  if (!(coord > (double)a && coord < (double)b)) return -1;
  return h;
}
__device__ int devDominant(const RefArgs& A, const float* lo, const float* hi, int& axis, double& coord) {
  int best = -1;
  float bestExt = -1.0f;
  for (int a = 0; a < 3; a++) {
    double c = 0.0;
    const int h = devPlane(A, a, lo[a], hi[a], c);
    const float e = hi[a] - lo[a];
    if (h > best || (h == best && h >= 0 && e > bestExt)) {
      best = h;
      bestExt = e;
      axis = a;
      coord = c;
    }
  }
  return best;
}
BDV long long devFloorDiv(long long a, long long n) {
  long long q = a / n;
  if ((a % n) != 0 && ((a < 0) != (n < 0))) q--;
  return q;
}
// AlphaClipper::Mask::count over mayPass
__device__ uint32_t devCount(const DevMask& m, long long x0, long long x1, long long y0, long long y1) {
  if (x1 < x0 || y1 < y0) return 0;
  const size_t W1 = (size_t)m.w + 1;
  const uint32_t* sat = m.mayPass;
  long long xs[2][2], ys[2][2];
  int nx = 0, ny = 0;
  {
    const long long s = devFloorDiv(x0, m.w) * m.w, a = x0 - s, b = x1 - s;
    if (b < m.w) {
      xs[nx][0] = a, xs[nx][1] = b, nx++;
    } else {
      xs[nx][0] = a, xs[nx][1] = m.w - 1, nx++;
      xs[nx][0] = 0, xs[nx][1] = (b - m.w) < (long long)(m.w - 1) ? (b - m.w) : (long long)(m.w - 1), nx++;
    }
  }
  {
    const long long s = devFloorDiv(y0, m.h) * m.h, a = y0 - s, b = y1 - s;
    if (b < m.h) {
      ys[ny][0] = a, ys[ny][1] = b, ny++;
    } else {
      ys[ny][0] = a, ys[ny][1] = m.h - 1, ny++;
      ys[ny][0] = 0, ys[ny][1] = (b - m.h) < (long long)(m.h - 1) ? (b - m.h) : (long long)(m.h - 1), ny++;
    }
  }
  uint32_t c = 0;
  for (int i = 0; i < nx; i++)
    for (int j = 0; j < ny; j++) {
      const long long xa = xs[i][0], xb = xs[i][1], ya = ys[j][0], yb = ys[j][1];
      c += sat[((size_t)yb + 1) * W1 + (size_t)xb + 1] - sat[(size_t)ya * W1 + (size_t)xb + 1] - sat[((size_t)yb + 1) * W1 + (size_t)xa] +
           sat[(size_t)ya * W1 + (size_t)xa];
    }
  return c;
}
// AlphaClipper::clip
__device__ bool devClip(const RefArgs& A, uint32_t tri, double (*poly)[2], int& n) {
  const uint32_t mat = A.triMaterial[tri];
  const int32_t mask = A.matMask[mat];
  if (mask < 0) return A.matVerdict[mat] != 2;
  // cellRect
  if (!A.texcoords) return true;
  const DevMask m = A.masks[mask];
  double uv[3][2], big = 0.0;
  for (int k = 0; k < 3; k++) {
    const uint32_t vi = A.indices[(size_t)tri * 3 + (size_t)k];
    uv[k][0] = (double)A.texcoords[(size_t)vi * 3];
    uv[k][1] = (double)A.texcoords[(size_t)vi * 3 + 1];
    if (!isfinite(uv[k][0]) || !isfinite(uv[k][1])) return true;
    big = dmax(big, dmax(fabs(uv[k][0]), fabs(uv[k][1])));
  }
  if (big > 4096.0) return true;
  const double margin = 0.5 + 1e-5 * (big + 1.0) * (double)(m.w < m.h ? m.h : m.w);
  double xa = 1e300, xb = -1e300, ya = 1e300, yb = -1e300;
  for (int k = 0; k < n; k++) {
    const double b0 = 1.0 - poly[k][0] - poly[k][1];
    const double u = uv[0][0] * b0 + uv[1][0] * poly[k][0] + uv[2][0] * poly[k][1];
    const double v = uv[0][1] * b0 + uv[1][1] * poly[k][0] + uv[2][1] * poly[k][1];
    const double x = u * (double)m.w - 0.5, y = v * (double)m.h - 0.5;
    xa = dmin(xa, x);
    xb = dmax(xb, x);
    ya = dmin(ya, y);
    yb = dmax(yb, y);
  }
  long long x0 = (long long)floor(xa - margin), x1 = (long long)floor(xb + margin);
  long long y0 = (long long)floor(ya - margin), y1 = (long long)floor(yb + margin);
  const bool fullX = x1 - x0 + 1 >= m.w, fullY = y1 - y0 + 1 >= m.h;
  if (fullX) x0 = 0, x1 = m.w - 1;
  if (fullY) y0 = 0, y1 = m.h - 1;
  if (devCount(m, x0, x1, y0, y1) == 0) return false;
  const double du1 = uv[1][0] - uv[0][0], du2 = uv[2][0] - uv[0][0], dv1 = uv[1][1] - uv[0][1], dv2 = uv[2][1] - uv[0][1];
  if (!fullX) {
    long long a = x0, b = x1;  // firstCol
    while (a < b) {
      const long long mid = a + (b - a) / 2;
      if (devCount(m, x0, mid, y0, y1) > 0)
        b = mid;
      else
        a = mid + 1;
    }
    const long long c0 = a;
    a = x0, b = x1;  // lastCol
    while (a < b) {
      const long long mid = a + (b - a + 1) / 2;
      if (devCount(m, mid, x1, y0, y1) > 0)
        a = mid;
      else
        b = mid - 1;
    }
    const long long c1 = a;
    const double uLo = ((double)c0 + 0.5 - margin) / (double)m.w, uHi = ((double)c1 + 1.5 + margin) / (double)m.w;
    if (c0 > x0) n = devClipInPlace(poly, n, uLo - uv[0][0], -du1, -du2);
    if (n >= 3 && c1 < x1) n = devClipInPlace(poly, n, uv[0][0] - uHi, du1, du2);
  }
  if (n >= 3 && !fullY) {
    long long a = y0, b = y1;  // firstRow
    while (a < b) {
      const long long mid = a + (b - a) / 2;
      if (devCount(m, x0, x1, y0, mid) > 0)
        b = mid;
      else
        a = mid + 1;
    }
    const long long r0 = a;
    a = y0, b = y1;  // lastRow
    while (a < b) {
      const long long mid = a + (b - a + 1) / 2;
      if (devCount(m, x0, x1, mid, y1) > 0)
        a = mid;
      else
        b = mid - 1;
    }
    const long long r1 = a;
    const double vLo = ((double)r0 + 0.5 - margin) / (double)m.h, vHi = ((double)r1 + 1.5 + margin) / (double)m.h;
    if (r0 > y0) n = devClipInPlace(poly, n, vLo - uv[0][1], -dv1, -dv2);
    if (n >= 3 && r1 < y1) n = devClipInPlace(poly, n, uv[0][1] - vHi, dv1, dv2);
  }
  return n >= 3;
}

// ---- pass 1 of bvh_build.cpp and its split counts on the device (BvhRefInput::splits == null) ----
// splitPriority
__device__ double devSplitPriority(const RefArgs& A, const BvhTri& r, const float* lo, const float* hi, double polyShare) {
  int axis = 0;
  double c = 0;
  const int h = devDominant(A, lo, hi, axis, c);
  if (h < 0) return 0.0;
  const double cx = (double)r.e1[1] * r.e2[2] - (double)r.e1[2] * r.e2[1], cy = (double)r.e1[2] * r.e2[0] - (double)r.e1[0] * r.e2[2],
               cz = (double)r.e1[0] * r.e2[1] - (double)r.e1[1] * r.e2[0];
  const double ideal = (fabs(cx) + fabs(cy) + fabs(cz)) * polyShare;
  const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
  const double gain = 2.0 * (dx * dy + dy * dz + dz * dx) - ideal;
  if (!(gain > 0.0)) return 0.0;
  return bvhCbrt(ldexp(gain, h - 30));
}
// what the clipper leaves of every triangle, its priority and the most splits it may get
__global__ __launch_bounds__(64) void k_prio(RefArgs A, float budgetOpaque, float budgetAlpha, float outlierArea, uint8_t* __restrict__ state,
                                             double* __restrict__ prio, float* __restrict__ capOf, uint32_t* __restrict__ splits) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.numTris) return;
  const BvhTri r = A.triRecs[t];
  const BvhBox tb = A.triBox[t];
  double b[kBvhPolyMax][2];
  int n = 3;
  b[0][0] = 0.0;
  b[0][1] = 0.0;
  b[1][0] = 1.0;
  b[1][1] = 0.0;
  b[2][0] = 0.0;
  b[2][1] = 1.0;
  float lo[3] = {tb.lo[0], tb.lo[1], tb.lo[2]}, hi[3] = {tb.hi[0], tb.hi[1], tb.hi[2]};
  bool shrunk = false;
  splits[t] = 0;
  prio[t] = 0.0;
  capOf[t] = (float)BDPT_SPLIT_MAX_PER_TRI;
  const bool nonOpaque = (r.flags & kTriNonOpaque) != 0;
  if (nonOpaque && A.haveClipper) {
    if (!devClip(A, t, b, n) || n < 3) {
      state[t] = 2;
      return;
    }
    shrunk = !(n == 3 && b[0][0] == 0.0 && b[0][1] == 0.0 && b[1][0] == 1.0 && b[1][1] == 0.0 && b[2][0] == 0.0 && b[2][1] == 1.0);
    if (shrunk) devPolyBoxIn(r, b, n, tb.lo, tb.hi, lo, hi);
  }
  state[t] = shrunk ? 1 : 0;
  const float budget = nonOpaque ? budgetAlpha : budgetOpaque;
  const float area = areaOf(lo, hi);
  if (budget > 0.0f && (nonOpaque || area >= outlierArea)) {
    prio[t] = devSplitPriority(A, r, lo, hi, shrunk ? devPolyArea2(b, n) : 1.0);
    if (!nonOpaque && outlierArea > 0.0f) {
      const float f = floorf((float)BDPT_SPLIT_OUTLIER * area / outlierArea);
      capOf[t] = f < (float)BDPT_SPLIT_MAX_PER_TRI ? f : (float)BDPT_SPLIT_MAX_PER_TRI;  // std::min(MAX, f)
    }
  }
}
BDV bool inClass(const RefArgs& A, const uint8_t* state, uint32_t t, int cls) {
  return state[t] != 2 && (((A.triRecs[t].flags & kTriNonOpaque) != 0) == (cls == 1));
}
// members of a class and their largest priority (non-negative doubles order like their bit patterns)
__global__ void k_class_stats(RefArgs A, const uint8_t* __restrict__ state, const double* __restrict__ prio, int cls, unsigned long long* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = t < A.numTris && inClass(A, state, t, cls);
  const unsigned long long m = __ballot(in);
  unsigned long long best = in ? (unsigned long long)__double_as_longlong(prio[t]) : 0ull;
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(best, off);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63u) == 0u && m) {
    atomicAdd(&out[0], (unsigned long long)__popcll(m));
    if (best) atomicMax(&out[1], best);
  }
}
// sum over the class of min(floor(D p), cap): integers, so the order of the additions does not matter
__global__ __launch_bounds__(256) void k_split_total(RefArgs A, const uint8_t* __restrict__ state, const double* __restrict__ prio, const float* __restrict__ capOf,
                                                     int cls, double D, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long s[256];
  unsigned long long acc = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < A.numTris; t += gridDim.x * blockDim.x)
    if (inClass(A, state, t, cls)) {
      const double v = floor(D * prio[t]), c = (double)capOf[t];
      acc += (unsigned long long)(c < v ? c : v);
    }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && s[0]) atomicAdd(out, s[0]);
}
__global__ void k_split_assign(RefArgs A, const uint8_t* __restrict__ state, const double* __restrict__ prio, const float* __restrict__ capOf, int cls, double D,
                               uint32_t* __restrict__ splits) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.numTris || !inClass(A, state, t, cls)) return;
  const double v = floor(D * prio[t]), c = (double)capOf[t];
  splits[t] = (uint32_t)(c < v ? c : v);
}
// dropped triangles, reference slots (splits + 1 per kept triangle), triangles that need the reference kernel
__global__ void k_ref_summary(uint32_t numTris, const uint8_t* __restrict__ state, const uint32_t* __restrict__ splits, unsigned long long* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long dropped = 0, slots = 0, work = 0;
  if (t < numTris) {
    if (state[t] == 2) {
      dropped = 1;
    } else {
      slots = (unsigned long long)splits[t] + 1ull;
      work = (splits[t] != 0 || state[t] != 0) ? 1ull : 0ull;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {  // (one atomic per wave and counter: ten million lanes on three words otherwise)
    dropped += __shfl_xor(dropped, off);
    slots += __shfl_xor(slots, off);
    work += __shfl_xor(work, off);
  }
  if ((threadIdx.x & 63u) == 0u) {
    if (dropped) atomicAdd(&out[0], dropped);
    if (slots) atomicAdd(&out[1], slots);
    if (work) atomicAdd(&out[2], work);
  }
}

// upper bounds: a triangle makes at most splits + 1 references and stacks at most `splits` pieces
__global__ void k_ref_caps(RefArgs A, uint32_t* __restrict__ capRefs, uint32_t* __restrict__ capStack) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.numTris) return;
  const bool dropped = A.state[t] == 2;
  capRefs[t] = dropped ? 0u : A.splits[t] + 1u;
  capStack[t] = dropped ? 0u : A.splits[t];
}
// pass 2 of bvh_build.cpp for triangle t: its references' boxes into boxes[slotAt[t] ...], their number into made[t]
__global__ __launch_bounds__(64) void k_make_refs(RefArgs A, const uint32_t* __restrict__ slotAt, const uint32_t* __restrict__ stackAt,
                                                  DevPiece* __restrict__ stacks, BvhBox* __restrict__ boxes, uint32_t* __restrict__ made) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.numTris) return;
  const uint32_t st = A.state[t];
  if (st == 2) {
    made[t] = 0;
    return;
  }
  BvhBox* const out = boxes + slotAt[t];
  const uint32_t nSplits = A.splits[t];
  const BvhBox tb = A.triBox[t];
  if (nSplits == 0 && st == 0) {
    out[0] = tb;
    made[t] = 1;
    return;
  }
  const BvhTri r = A.triRecs[t];
  const bool alpha = (r.flags & kTriNonOpaque) != 0 && A.haveClipper;
  // wholePiece
  DevPiece pc;
  pc.n = 3;
  pc.b[0][0] = 0.0;
  pc.b[0][1] = 0.0;
  pc.b[1][0] = 1.0;
  pc.b[1][1] = 0.0;
  pc.b[2][0] = 0.0;
  pc.b[2][1] = 1.0;
  for (int a = 0; a < 3; a++) {
    pc.lo[a] = tb.lo[a];
    pc.hi[a] = tb.hi[a];
  }
  if (alpha) {
    if (!devClip(A, t, pc.b, pc.n) || pc.n < 3) {  // (cannot happen: pass 1 kept it)
      made[t] = 0;
      return;
    }
    const bool shrunk =
        !(pc.n == 3 && pc.b[0][0] == 0.0 && pc.b[0][1] == 0.0 && pc.b[1][0] == 1.0 && pc.b[1][1] == 0.0 && pc.b[2][0] == 0.0 && pc.b[2][1] == 1.0);
    if (shrunk) devPolyBoxIn(r, pc.b, pc.n, tb.lo, tb.hi, pc.lo, pc.hi);
  }
  uint32_t m = 0;
  auto emit = [&](const DevPiece& p) {
    BvhBox bx;
    for (int a = 0; a < 3; a++) {
      bx.lo[a] = p.lo[a];
      bx.hi[a] = p.hi[a];
    }
    out[m++] = bx;
  };
  if (nSplits == 0) {
    emit(pc);
    made[t] = m;
    return;
  }
  pc.splits = nSplits;
  // splitTriangle
  DevPiece* const stack = stacks + stackAt[t];
  uint32_t sp = 0;
  for (;;) {
    for (int guard = 0;; guard++) {
      int axis = 0;
      double c = 0;
      if (pc.splits == 0 || guard > 96 || pc.n + 2 > kBvhPolyMax || devDominant(A, pc.lo, pc.hi, axis, c) < 0) {
        emit(pc);
        break;
      }
      const double PA = (double)r.v0[axis] - c, PB = (double)r.e1[axis], PC = (double)r.e2[axis];
      DevPiece lo, hi;
      lo.n = devClipHalfPlane(pc.b, pc.n, PA, PB, PC, lo.b);
      hi.n = devClipHalfPlane(pc.b, pc.n, -PA, -PB, -PC, hi.b);
      bool haveLo = lo.n >= 3 && devPolyArea2(lo.b, lo.n) > 0.0, haveHi = hi.n >= 3 && devPolyArea2(hi.b, hi.n) > 0.0;
      if (alpha) {
        if (haveLo) haveLo = devClip(A, t, lo.b, lo.n) && lo.n >= 3;
        if (haveHi) haveHi = devClip(A, t, hi.b, hi.n) && hi.n >= 3;
      }
      if (haveLo) devPolyBoxIn(r, lo.b, lo.n, pc.lo, pc.hi, lo.lo, lo.hi);
      if (haveHi) devPolyBoxIn(r, hi.b, hi.n, pc.lo, pc.hi, hi.lo, hi.hi);
      if (!haveLo && !haveHi) {
        if (!alpha) emit(pc);  // (a sliver the clip lost to rounding: keep the piece as it was)
        break;
      }
      if (!haveLo || !haveHi) {
        const uint32_t s = pc.splits;
        pc = haveLo ? lo : hi;
        pc.splits = s;
        continue;
      }
      const uint32_t rest = pc.splits - 1;
      const double wl = ((double)lo.hi[0] - lo.lo[0]) + ((double)lo.hi[1] - lo.lo[1]) + ((double)lo.hi[2] - lo.lo[2]);
      const double wh = ((double)hi.hi[0] - hi.lo[0]) + ((double)hi.hi[1] - hi.lo[1]) + ((double)hi.hi[2] - hi.lo[2]);
      uint32_t sl = (wl + wh > 0.0) ? (uint32_t)floor((double)rest * wl / (wl + wh) + 0.5) : rest / 2;
      if (sl > rest) sl = rest;
      lo.splits = sl;
      hi.splits = rest - sl;
      stack[sp++] = hi;
      pc = lo;
      guard = 0;
    }
    if (sp == 0) break;
    pc = stack[--sp];
  }
  made[t] = m;
}
// no triangle is split or shrunk (a scene without alpha-mode materials, built without opaque splits): one reference per
// triangle that was not dropped, its box the triangle's — without k_make_refs and the scratch its private arrays need
__global__ void k_plain_refs(RefArgs A, const uint32_t* __restrict__ slotAt, BvhBox* __restrict__ boxes, uint32_t* __restrict__ made) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= A.numTris) return;
  const bool keep = A.state[t] != 2;
  if (keep) boxes[slotAt[t]] = A.triBox[t];
  made[t] = keep ? 1u : 0u;
}
// the references in triangle order: box (-0 -> +0), centre, id = index; and the triangle of each
__global__ void k_compact_refs(uint32_t numTris, const uint32_t* __restrict__ slotAt, const uint32_t* __restrict__ made, const uint32_t* __restrict__ refAt,
                               const BvhBox* __restrict__ boxes, BvhBuildRef* __restrict__ refs, uint32_t* __restrict__ refTri) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= numTris) return;
  const uint32_t m = made[t], at = refAt[t];
  for (uint32_t j = 0; j < m; j++) {
    const BvhBox bx = boxes[slotAt[t] + j];
    BvhBuildRef r;
    for (int k = 0; k < 3; k++) {
      r.box.lo[k] = bx.lo[k] + 0.0f;
      r.box.hi[k] = bx.hi[k] + 0.0f;
      r.cent[k] = 0.5f * (r.box.lo[k] + r.box.hi[k]) + 0.0f;
    }
    r.id = at + j;
    refs[at + j] = r;
    refTri[at + j] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// The four-wide collapse on the device: bvh_build.cpp's `collapse` — the same greedy widening under the stack budget, the
// same node ORDER (the top of the tree depth first; then, in the order the top met them, the subtrees of at most
// kCollapseGrain binary nodes, each depth first) — computed level by level instead of by a walk:
//   metrics   binary height and subtree size, bottom-up over the tree's levels
//   expand    the wide nodes breadth first: a level's jobs choose their children; a scan places the next level's jobs
//   sizes     wide-subtree sizes bottom-up (all nodes / top nodes only), deferred children per top node
//   number    the final index of every wide node, top-down: top part, deferred roots (by the top's order), the rest
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kCollapseGrainDev = 1u << 15;  // (bvh_build.cpp kCollapseGrain)
enum : uint32_t { WF_TOP = 1u, WF_DEFROOT = 2u };
struct WideTmp {  // breadth-first arrays, one entry per wide node
  uint32_t* src;
  uint32_t* kids;    // 4 per node
  int32_t* nk;
  uint32_t* depth;
  uint32_t* above;   // stack entries above this node's own (the parent's `need`)
  uint32_t* need;
  int32_t* parent;   // breadth-first index of the parent (-1: the root)
  int32_t* kIdx;
  uint32_t* flags;
  int32_t* child;    // 4 per node: breadth-first index of the wide node under kid k (-1: a leaf)
  uint32_t* nInterior;
  uint32_t* sizeAll;
  uint32_t* sizeTop;
  uint32_t* defCount;
  uint32_t* idx;     // final index
};
__global__ void k_tree_metrics(const BvhBuildNode* __restrict__ nodes, uint32_t start, uint32_t end, uint16_t* __restrict__ hgt, uint32_t* __restrict__ sub) {
  const uint32_t t = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= end) return;
  const BvhBuildNode nd = nodes[t];
  if (nd.left >= 0) {
    const uint16_t hl = hgt[nd.left], hr = hgt[nd.right];
    hgt[t] = (uint16_t)(1 + (hl > hr ? hl : hr));
    sub[t] = 1u + sub[nd.left] + sub[nd.right];
  } else {
    hgt[t] = 0;
    sub[t] = 1;
  }
}
BDV float nodeArea(const BvhBuildNode& n) { return areaOf(n.box.lo, n.box.hi); }
// one job = one wide node: its children (the collapse's inner loop), how much stack it needs, how many of them are interior
__global__ void k_wide_expand(WideTmp W, uint32_t start, uint32_t end, const BvhBuildNode* __restrict__ nodes, const uint16_t* __restrict__ hgt,
                              uint32_t* __restrict__ maxDepth, uint32_t* __restrict__ maxStack) {
  const uint32_t b = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= end) return;
  const uint32_t src = W.src[b], above = W.above[b];
  uint32_t kids[4] = {0, 0, 0, 0};
  int nk = 0;
  if (nodes[src].left < 0) {  // the root is a single leaf
    kids[nk++] = src;
  } else {
    kids[nk++] = (uint32_t)nodes[src].left;
    kids[nk++] = (uint32_t)nodes[src].right;
    while (nk < 4) {
      int best = -1;
      float bestArea = -1.0f;
      for (int k = 0; k < nk; k++)
        if (nodes[kids[k]].left >= 0) {
          const float ar = nodeArea(nodes[kids[k]]);
          if (ar > bestArea) {
            bestArea = ar;
            best = k;
          }
        }
      if (best < 0) break;
      const uint32_t t = kids[best];
      uint32_t tallest = hgt[nodes[t].left] > hgt[nodes[t].right] ? hgt[nodes[t].left] : hgt[nodes[t].right];
      for (int k = 0; k < nk; k++)
        if (k != best) tallest = tallest > hgt[kids[k]] ? tallest : (uint32_t)hgt[kids[k]];
      if (above + (uint32_t)nk + tallest > (uint32_t)kBvhMaxStack) break;
      kids[best] = (uint32_t)nodes[t].left;
      kids[nk++] = (uint32_t)nodes[t].right;
    }
  }
  const uint32_t need = above + (uint32_t)(nk - 1);
  uint32_t nInt = 0;
  for (int k = 0; k < 4; k++) {
    W.kids[(size_t)b * 4 + k] = k < nk ? kids[k] : 0u;
    W.child[(size_t)b * 4 + k] = -1;
    if (k < nk && nodes[kids[k]].left >= 0) nInt++;
  }
  W.nk[b] = nk;
  W.need[b] = need;
  W.nInterior[b] = nInt;
  atomicMax(maxDepth, W.depth[b]);
  atomicMax(maxStack, need);
}
// the jobs of the next level: the interior kids of this level's nodes, in (node, k) order
__global__ void k_wide_children(WideTmp W, uint32_t start, uint32_t end, const uint32_t* __restrict__ at, const BvhBuildNode* __restrict__ nodes,
                                const uint32_t* __restrict__ sub, int deferral) {
  const uint32_t b = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= end) return;
  uint32_t c = end + at[b - start];
  const uint32_t pf = W.flags[b];
  const int nk = W.nk[b];
  for (int k = 0; k < nk; k++) {
    const uint32_t kid = W.kids[(size_t)b * 4 + k];
    if (nodes[kid].left < 0) continue;
    W.child[(size_t)b * 4 + k] = (int32_t)c;
    W.src[c] = kid;
    W.depth[c] = W.depth[b] + 1;
    W.above[c] = W.need[b];
    W.parent[c] = (int32_t)b;
    W.kIdx[c] = k;
    const bool parentTop = (pf & WF_TOP) != 0;
    const bool def = parentTop && deferral && sub[kid] <= kCollapseGrainDev;
    W.flags[c] = def ? WF_DEFROOT : (parentTop ? WF_TOP : 0u);
    c++;
  }
}
__global__ void k_wide_sizes(WideTmp W, uint32_t start, uint32_t end) {
  const uint32_t b = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= end) return;
  uint32_t all = 1, top = 1, def = 0;
  for (int k = 0; k < 4; k++) {
    const int32_t c = W.child[(size_t)b * 4 + k];
    if (c < 0) continue;
    all += W.sizeAll[c];
    if (W.flags[c] & WF_TOP) top += W.sizeTop[c];
    if (W.flags[c] & WF_DEFROOT) def++;
  }
  W.sizeAll[b] = all;
  W.sizeTop[b] = (W.flags[b] & WF_TOP) ? top : 0u;
  W.defCount[b] = (W.flags[b] & WF_TOP) ? def : 0u;
}
// pass 1: the top part, depth first (children in k order, top children only)
__global__ void k_wide_number_top(WideTmp W, uint32_t start, uint32_t end, uint32_t* __restrict__ defCountByIdx) {
  const uint32_t b = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= end || !(W.flags[b] & WF_TOP)) return;
  if (W.parent[b] < 0) W.idx[b] = 0;
  const uint32_t me = W.idx[b];
  defCountByIdx[me] = W.defCount[b];
  uint32_t next = me + 1;
  for (int k = 0; k < 4; k++) {
    const int32_t c = W.child[(size_t)b * 4 + k];
    if (c < 0 || !(W.flags[c] & WF_TOP)) continue;
    W.idx[c] = next;
    next += W.sizeTop[c];
  }
}
// pass 2: the deferred roots, in the order the top's walk met them (by parent index; within a parent by DEcreasing k)
__global__ void k_wide_def_rank(WideTmp W, uint32_t nWide, const uint32_t* __restrict__ defBaseByIdx, uint32_t* __restrict__ rankOf, uint32_t* __restrict__ sizeByRank) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nWide || !(W.flags[b] & WF_TOP) || W.defCount[b] == 0) return;
  uint32_t r = defBaseByIdx[W.idx[b]];
  for (int k = 3; k >= 0; k--) {
    const int32_t c = W.child[(size_t)b * 4 + k];
    if (c < 0 || !(W.flags[c] & WF_DEFROOT)) continue;
    rankOf[c] = r;
    sizeByRank[r] = W.sizeAll[c];
    r++;
  }
}
__global__ void k_wide_number_def(WideTmp W, uint32_t nWide, const uint32_t* __restrict__ rankOf, const uint32_t* __restrict__ offByRank, uint32_t numTop) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nWide || !(W.flags[b] & WF_DEFROOT)) return;
  W.idx[b] = numTop + offByRank[rankOf[b]];
}
// pass 3: inside the deferred subtrees, depth first
__global__ void k_wide_number_rest(WideTmp W, uint32_t start, uint32_t end) {
  const uint32_t b = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= end || (W.flags[b] & WF_TOP)) return;
  uint32_t next = W.idx[b] + 1;
  for (int k = 0; k < 4; k++) {
    const int32_t c = W.child[(size_t)b * 4 + k];
    if (c < 0) continue;
    W.idx[c] = next;
    next += W.sizeAll[c];
  }
}
__global__ void k_wide_emit(WideTmp W, uint32_t nWide, BvhWideNode* __restrict__ wide, BvhSlot* __restrict__ slots) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nWide) return;
  BvhWideNode w;
  w.src = W.src[b];
  for (int k = 0; k < 4; k++) w.kids[k] = W.kids[(size_t)b * 4 + k];
  w.nk = W.nk[b];
  w.depth = W.depth[b];
  const uint32_t i = W.idx[b];
  wide[i] = w;
  BvhSlot s;
  s.node = W.parent[b] < 0 ? -1 : (int32_t)W.idx[W.parent[b]];
  s.idx = W.parent[b] < 0 ? -1 : W.kIdx[b];
  slots[i] = s;
}
// the SAH cost as bvh_build.cpp sums it: float terms (computed side by side), summed in double in node order within blocks
// of 65536 nodes (one thread per block: the order of the additions is the sum's definition)
__global__ void k_sah_terms(const BvhWideNode* __restrict__ wide, uint32_t nWide, const BvhBuildNode* __restrict__ nodes, float rootArea, float* __restrict__ term) {
  const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= nWide) return;
  const BvhWideNode w = wide[wi];
  for (int k = 0; k < 4; k++) {
    float t = 0.0f;
    if (k < w.nk) {
      const BvhBuildNode c = nodes[w.kids[k]];
      t = (c.left < 0 ? 1.0f * (float)c.count : 1.0f) * nodeArea(c) / rootArea;
    }
    term[(size_t)wi * 4 + k] = t;
  }
}
// (one wave per block of nodes: all lanes fetch the next 256 terms into LDS, lane 0 adds them in order — a term of a child
// slot that does not exist is +0 and adding it changes nothing, so the four slots of every node are simply added)
__global__ __launch_bounds__(64) void k_sah_sum(const float* __restrict__ term, uint32_t nWide, double* __restrict__ part) {
  __shared__ float4 s[64];
  const uint32_t blk = blockIdx.x, lane = threadIdx.x;
  const uint32_t w0 = blk * 65536u;
  if (w0 >= nWide) return;
  const uint32_t w1 = w0 + 65536u < nWide ? w0 + 65536u : nWide;
  const float4* t4 = reinterpret_cast<const float4*>(term);
  double cost = 0.0;
  for (uint32_t base = w0; base < w1; base += 64) {
    const uint32_t wi = base + lane;
    s[lane] = wi < w1 ? t4[wi] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    if (lane == 0) {
      const uint32_t n = w1 - base < 64u ? w1 - base : 64u;
      for (uint32_t i = 0; i < n; i++) {
        const float4 v = s[i];
        cost += (double)v.x;
        cost += (double)v.y;
        cost += (double)v.z;
        cost += (double)v.w;
      }
    }
    __syncthreads();
  }
  if (lane == 0) part[blk] = cost;
}

// Device -> pageable host memory through pinned staging buffers, the host-side copies shared among a few threads (a
// plain hipMemcpy into pageable memory runs at ~3 GB/s here; the node list of a 10 M-triangle scene is 1 GB).
bool downloadStaged(void* dst, const void* src, size_t bytes, std::string& err) {
  constexpr size_t kStage = 32u << 20;
  constexpr int kBufs = 2, kCopyThreads = 4;
  auto bad = [&](hipError_t e) {
    err = std::string("device tree builder: download: ") + hipGetErrorString(e);
    return false;
  };
  if (bytes <= (8u << 20)) {
    const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
    return e == hipSuccess ? true : bad(e);
  }
  void* stage[kBufs] = {};
  hipEvent_t ev[kBufs] = {};
  hipStream_t st = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int i = 0; i < kBufs && e == hipSuccess; i++) {
    e = hipHostMalloc(&stage[i], kStage, hipHostMallocPortable);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  }
  const size_t nChunks = (bytes + kStage - 1) / kStage;
  auto sizeOf = [&](size_t c) { return std::min(kStage, bytes - c * kStage); };
  auto issue = [&](size_t c) {
    hipError_t r = hipMemcpyAsync(stage[c % kBufs], static_cast<const char*>(src) + c * kStage, sizeOf(c), hipMemcpyDeviceToHost, st);
    if (r == hipSuccess) r = hipEventRecord(ev[c % kBufs], st);
    return r;
  };
  if (e == hipSuccess) e = issue(0);
  for (size_t c = 0; c < nChunks && e == hipSuccess; c++) {
    if (c + 1 < nChunks) e = issue(c + 1);  // (buffer (c + 1) % 2 was consumed in the previous round)
    if (e == hipSuccess) e = hipEventSynchronize(ev[c % kBufs]);
    if (e != hipSuccess) break;
    const size_t sz = sizeOf(c), per = (sz + kCopyThreads - 1) / kCopyThreads;
    char* const d = static_cast<char*>(dst) + c * kStage;
    const char* const sp = static_cast<const char*>(stage[c % kBufs]);
    std::thread th[kCopyThreads];
    for (int t = 1; t < kCopyThreads; t++)
      if ((size_t)t * per < sz) th[t] = std::thread([=] { std::memcpy(d + (size_t)t * per, sp + (size_t)t * per, std::min(per, sz - (size_t)t * per)); });
    std::memcpy(d, sp, std::min(per, sz));
    for (int t = 1; t < kCopyThreads; t++)
      if (th[t].joinable()) th[t].join();
  }
  if (st) (void)hipStreamSynchronize(st);
  for (int i = 0; i < kBufs; i++) {
    if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (stage[i]) (void)hipHostFree(stage[i]);
  }
  if (st) (void)hipStreamDestroy(st);
  return e == hipSuccess ? true : bad(e);
}

// Pageable host memory -> device, the mirror of downloadStaged: the host-side copies into pinned staging buffers are
// shared among a few threads while the previous buffer is on its way to the device (a plain hipMemcpy from pageable
// memory runs at 3-4.5 GB/s here: the 2.3 GB bdpt_set_scene hands over for a 10 M-triangle scene were what it waited
// for).  Staging buffers are kept for the life of the process (pinning costs milliseconds per call otherwise): a small
// pool, so that the build's thread and bdpt_set_scene's side thread can upload at the same time.
namespace {
constexpr size_t kUpStage = 32u << 20;
std::mutex gStageLock;
std::vector<void*> gStageFree;  // pinned buffers of kUpStage bytes, hipHostMallocPortable: one pool for every device of the process
// bvhPrewarmStaging (bdpt_create): four buffers pinned on a thread of their own while the host does other things —
// pinning 128 MB takes the driver over 0.1 s, which the first bdpt_set_scene of a process would otherwise wait for
struct Prewarm {
  std::thread th;             // touched under joinLock only (and by the destructor, after main)
  std::mutex joinLock;
  std::once_flag once;
  std::atomic<bool> pending{false};  // started and not joined yet
  ~Prewarm() {
    if (th.joinable()) th.join();
  }
} gPrewarm;
void waitForPrewarm() {
  if (!gPrewarm.pending.load(std::memory_order_acquire)) return;
  std::lock_guard<std::mutex> j(gPrewarm.joinLock);
  if (gPrewarm.th.joinable()) gPrewarm.th.join();
  gPrewarm.pending.store(false, std::memory_order_release);
}
void* takeStage() {
  for (int attempt = 0; attempt < 2; attempt++) {
    {
      std::lock_guard<std::mutex> g(gStageLock);
      if (!gStageFree.empty()) {
        void* p = gStageFree.back();
        gStageFree.pop_back();
        return p;
      }
    }
    if (attempt == 0) waitForPrewarm();  // the warm-up has not delivered yet: wait for it rather than pin more
  }
  void* p = nullptr;
  return hipHostMalloc(&p, kUpStage, hipHostMallocPortable) == hipSuccess ? p : nullptr;
}
void giveStage(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> g(gStageLock);
  if (gStageFree.size() < 6)
    gStageFree.push_back(p);
  else
    (void)hipHostFree(p);
}
}  // namespace
bool uploadStagedImpl(void* dst, const void* src, size_t bytes, std::string& err) {
  constexpr int kBufs = 2, kCopyThreads = 8;
  auto bad = [&](hipError_t e) {
    err = std::string("upload: ") + hipGetErrorString(e);
    return false;
  };
  if (bytes == 0) return true;
  if (bytes <= (4u << 20)) {
    const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    return e == hipSuccess ? true : bad(e);
  }
  void* stage[kBufs] = {takeStage(), takeStage()};
  hipEvent_t ev[kBufs] = {};
  hipStream_t st = nullptr;
  hipError_t e = (stage[0] && stage[1]) ? hipStreamCreateWithFlags(&st, hipStreamNonBlocking) : hipErrorOutOfMemory;
  for (int i = 0; i < kBufs && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
  const size_t nChunks = (bytes + kUpStage - 1) / kUpStage;
  bool used[kBufs] = {false, false};
  for (size_t c = 0; c < nChunks && e == hipSuccess; c++) {
    const int b = (int)(c % kBufs);
    if (used[b]) e = hipEventSynchronize(ev[b]);  // the copy that last read this buffer has left it
    if (e != hipSuccess) break;
    const size_t sz = std::min(kUpStage, bytes - c * kUpStage), per = (sz + kCopyThreads - 1) / kCopyThreads;
    const char* const sp = static_cast<const char*>(src) + c * kUpStage;
    char* const d = static_cast<char*>(stage[b]);
    std::thread th[kCopyThreads];
    for (int t = 1; t < kCopyThreads; t++)
      if ((size_t)t * per < sz) th[t] = std::thread([=] { std::memcpy(d + (size_t)t * per, sp + (size_t)t * per, std::min(per, sz - (size_t)t * per)); });
    std::memcpy(d, sp, std::min(per, sz));
    for (int t = 1; t < kCopyThreads; t++)
      if (th[t].joinable()) th[t].join();
    e = hipMemcpyAsync(static_cast<char*>(dst) + c * kUpStage, stage[b], sz, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(ev[b], st);
    used[b] = true;
  }
  if (st) {
    const hipError_t s2 = hipStreamSynchronize(st);
    if (e == hipSuccess) e = s2;
  }
  for (int i = 0; i < kBufs; i++) {
    if (ev[i]) (void)hipEventDestroy(ev[i]);
    giveStage(stage[i]);
  }
  if (st) (void)hipStreamDestroy(st);
  return e == hipSuccess ? true : bad(e);
}

template <class T>
bool devAllocT(BvhDeviceArena& pool, T** p, size_t count, std::string& err) {
  void* q = pool.alloc(count * sizeof(T));
  if (!q) {
    err = "device builder: out of device memory";
    return false;
  }
  *p = static_cast<T*>(q);
  return true;
}

struct Scan {
  uint32_t* sums = nullptr;
  uint32_t* total = nullptr;
  uint32_t capTiles = 0;
  void run(const uint32_t* in, uint32_t n, uint32_t* out, hipStream_t st) const {  // out has n + 1 entries
    const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_sums, dim3(tiles), dim3(kScanBlock), 0, st, in, n, sums);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(kScanBlock), 0, st, sums, tiles, total);
    hipLaunchKernelGGL(k_scan_local, dim3(tiles), dim3(kScanBlock), 0, st, in, n, sums, total, out);
  }
};

bool allocLevel(BvhDeviceArena& pool, Level& L, size_t cap, std::string& err) {
  return devAllocT(pool, &L.node, cap, err) && devAllocT(pool, &L.first, cap, err) && devAllocT(pool, &L.count, cap, err) &&
         devAllocT(pool, &L.depth, cap, err) && devAllocT(pool, &L.state, cap, err) && devAllocT(pool, &L.bnd, cap * 12, err) &&
         devAllocT(pool, &L.lo, cap * 3, err) && devAllocT(pool, &L.scale, cap * 3, err) && devAllocT(pool, &L.axis, cap, err) && devAllocT(pool, &L.maxis, cap, err) &&
         devAllocT(pool, &L.split, cap, err) && devAllocT(pool, &L.nLeft, cap, err) && devAllocT(pool, &L.sel, cap, err) &&
         devAllocT(pool, &L.pivotHi, cap, err) && devAllocT(pool, &L.pivotLo, cap, err) &&
         devAllocT(pool, &L.child, cap, err);
}

// The collapse of this build's tree (see the kernels above): wide nodes and slots in device memory, the summary to `out`.
bool collapseOnDevice(BvhDeviceBuild* build, BvhDeviceArena& pool, BvhWideNode** dWideOut, BvhSlot** dSlotsOut, uint32_t* nWideOut, Bvh& out, std::string& err) {
  auto ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err = std::string("device collapse: ") + what + ": " + hipGetErrorString(e);
    return false;
  };
  hipStream_t st = nullptr;
  const dim3 blk(256);
  auto gridFor = [](size_t m) { return dim3((unsigned)((m + 255) / 256)); };
  const uint32_t numNodes = build->numNodes;
  const BvhBuildNode* nodes = build->nodes;
  const std::vector<uint32_t>& ls = build->levelStart;  // first node of every level of the binary tree, + numNodes
  if (ls.size() < 2 || ls.back() != numNodes) {
    err = "device collapse: no tree of this build in device memory";
    return false;
  }
  uint16_t* hgt = nullptr;
  uint32_t* sub = nullptr;
  if (!devAllocT(pool, &hgt, numNodes, err) || !devAllocT(pool, &sub, numNodes, err)) return false;
  for (size_t l = ls.size() - 1; l-- > 0;)
    if (ls[l + 1] > ls[l]) hipLaunchKernelGGL(k_tree_metrics, gridFor(ls[l + 1] - ls[l]), blk, 0, st, nodes, ls[l], ls[l + 1], hgt, sub);
  const uint32_t capWide = numNodes / 2 + 2;  // every wide node covers an interior binary node of its own (or is the lone leaf root)
  WideTmp W;
  uint32_t *at = nullptr, *counters = nullptr;
  Scan scan;
  scan.capTiles = (capWide + kScanTile - 1) / kScanTile + 1;
  if (!devAllocT(pool, &W.src, capWide, err) || !devAllocT(pool, &W.kids, (size_t)capWide * 4, err) || !devAllocT(pool, &W.nk, capWide, err) ||
      !devAllocT(pool, &W.depth, capWide, err) || !devAllocT(pool, &W.above, capWide, err) || !devAllocT(pool, &W.need, capWide, err) ||
      !devAllocT(pool, &W.parent, capWide, err) || !devAllocT(pool, &W.kIdx, capWide, err) || !devAllocT(pool, &W.flags, capWide, err) ||
      !devAllocT(pool, &W.child, (size_t)capWide * 4, err) || !devAllocT(pool, &W.nInterior, capWide, err) || !devAllocT(pool, &W.sizeAll, capWide, err) ||
      !devAllocT(pool, &W.sizeTop, capWide, err) || !devAllocT(pool, &W.defCount, capWide, err) || !devAllocT(pool, &W.idx, capWide, err) ||
      !devAllocT(pool, &at, (size_t)capWide + 1, err) || !devAllocT(pool, &counters, 2, err) || !devAllocT(pool, &scan.sums, scan.capTiles, err) ||
      !devAllocT(pool, &scan.total, 1, err))
    return false;
  const int deferral = numNodes > 4u * kCollapseGrainDev ? 1 : 0;
  {  // the root job
    const uint32_t zero = 0, top = WF_TOP;
    const int32_t minus = -1;
    if (!ok(hipMemsetAsync(counters, 0, 8, st), "memset") || !ok(hipMemcpy(W.src, &zero, 4, hipMemcpyHostToDevice), "root") ||
        !ok(hipMemcpy(W.depth, &zero, 4, hipMemcpyHostToDevice), "root") || !ok(hipMemcpy(W.above, &zero, 4, hipMemcpyHostToDevice), "root") ||
        !ok(hipMemcpy(W.parent, &minus, 4, hipMemcpyHostToDevice), "root") || !ok(hipMemcpy(W.kIdx, &minus, 4, hipMemcpyHostToDevice), "root") ||
        !ok(hipMemcpy(W.flags, &top, 4, hipMemcpyHostToDevice), "root"))
      return false;
  }
  std::vector<uint32_t> wl{0, 1};  // breadth-first ranges of the wide levels: [wl[i], wl[i + 1])
  for (;;) {
    const uint32_t s0 = wl[wl.size() - 2], s1 = wl.back();
    hipLaunchKernelGGL(k_wide_expand, gridFor(s1 - s0), blk, 0, st, W, s0, s1, nodes, hgt, counters, counters + 1);
    scan.run(W.nInterior + s0, s1 - s0, at, st);
    uint32_t nNext = 0;
    if (!ok(hipMemcpy(&nNext, at + (s1 - s0), 4, hipMemcpyDeviceToHost), "level")) return false;
    if (nNext == 0) break;
    if ((size_t)s1 + nNext > capWide || wl.size() > 4096) {
      err = "device collapse: node count out of bounds";
      return false;
    }
    hipLaunchKernelGGL(k_wide_children, gridFor(s1 - s0), blk, 0, st, W, s0, s1, at, nodes, sub, deferral);
    wl.push_back(s1 + nNext);
  }
  const uint32_t nWide = wl.back();
  for (size_t l = wl.size() - 1; l-- > 0;) hipLaunchKernelGGL(k_wide_sizes, gridFor(wl[l + 1] - wl[l]), blk, 0, st, W, wl[l], wl[l + 1]);
  uint32_t numTop = 0;
  if (!ok(hipMemcpy(&numTop, W.sizeTop, 4, hipMemcpyDeviceToHost), "sizes")) return false;
  uint32_t *defCountByIdx = nullptr, *defBaseByIdx = nullptr, *rankOf = nullptr, *sizeByRank = nullptr, *offByRank = nullptr;
  if (!devAllocT(pool, &defCountByIdx, (size_t)numTop + 1, err) || !devAllocT(pool, &defBaseByIdx, (size_t)numTop + 1, err) || !devAllocT(pool, &rankOf, nWide, err) ||
      !devAllocT(pool, &sizeByRank, (size_t)nWide + 1, err) || !devAllocT(pool, &offByRank, (size_t)nWide + 1, err))
    return false;
  for (size_t l = 0; l + 1 < wl.size(); l++) hipLaunchKernelGGL(k_wide_number_top, gridFor(wl[l + 1] - wl[l]), blk, 0, st, W, wl[l], wl[l + 1], defCountByIdx);
  scan.run(defCountByIdx, numTop, defBaseByIdx, st);
  uint32_t numDef = 0;
  if (!ok(hipMemcpy(&numDef, defBaseByIdx + numTop, 4, hipMemcpyDeviceToHost), "deferred")) return false;
  if (numDef > 0) {
    hipLaunchKernelGGL(k_wide_def_rank, gridFor(nWide), blk, 0, st, W, nWide, defBaseByIdx, rankOf, sizeByRank);
    scan.run(sizeByRank, numDef, offByRank, st);
    hipLaunchKernelGGL(k_wide_number_def, gridFor(nWide), blk, 0, st, W, nWide, rankOf, offByRank, numTop);
    for (size_t l = 0; l + 1 < wl.size(); l++) hipLaunchKernelGGL(k_wide_number_rest, gridFor(wl[l + 1] - wl[l]), blk, 0, st, W, wl[l], wl[l + 1]);
  }
  BvhWideNode* dWide = nullptr;
  BvhSlot* dSlots = nullptr;
  float* term = nullptr;
  double* part = nullptr;
  const uint32_t nBlocks = (nWide + 65535u) / 65536u;
  if (!devAllocT(pool, &dWide, nWide, err) || !devAllocT(pool, &dSlots, nWide, err) || !devAllocT(pool, &term, (size_t)nWide * 4, err) || !devAllocT(pool, &part, nBlocks, err))
    return false;
  hipLaunchKernelGGL(k_wide_emit, gridFor(nWide), blk, 0, st, W, nWide, dWide, dSlots);
  BvhBuildNode root;
  uint32_t hc[2] = {0, 0};
  if (!ok(hipMemcpy(&root, nodes, sizeof(root), hipMemcpyDeviceToHost), "root") || !ok(hipMemcpy(hc, counters, 8, hipMemcpyDeviceToHost), "summary")) return false;
  const float rootArea = root.box.area();
  double cost = 0.0;
  if (rootArea > 0) {
    hipLaunchKernelGGL(k_sah_terms, gridFor(nWide), blk, 0, st, dWide, nWide, nodes, rootArea, term);
    hipLaunchKernelGGL(k_sah_sum, dim3(nBlocks), dim3(64), 0, st, term, nWide, part);
    std::vector<double> hp(nBlocks);
    if (!ok(hipMemcpy(hp.data(), part, (size_t)nBlocks * 8, hipMemcpyDeviceToHost), "cost")) return false;
    for (double v : hp) cost += v;
  }
  if (!ok(hipGetLastError(), "launch") || !ok(hipDeviceSynchronize(), "synchronise")) return false;
  out.maxDepth = hc[0];
  out.maxStack = hc[1];
  out.numNodes = nWide;
  out.sahCost = (float)cost + 1.0f;  // (+ kCostTraverse)
  *dWideOut = dWide;
  *dSlotsOut = dSlots;
  *nWideOut = nWide;
  return true;
}

}  // namespace

// The BvhTreeBuilder bdpt_set_scene plugs into buildBvh; `user` is the build's BvhDeviceBuild.
bool buildBinaryTreeOnDevice(void* user, const BvhBuildRef* refs, uint32_t n, BigVec<uint32_t>& order, BigVec<BvhBuildNode>& nodes, std::string& err) {
  BvhDeviceBuild* const build = static_cast<BvhDeviceBuild*>(user);
  if (!build) {
    err = "device tree builder: no build object";
    return false;
  }
  const int device = build->device;
  if (n == 0) return false;
  if (hipSetDevice(device) != hipSuccess) {
    err = "device tree builder: no such device";
    return false;
  }
  const bool verbose = std::getenv("BDPT_BUILD_VERBOSE") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[bvh]   device %-10s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  // BDPT_BUILD_VERBOSE=2: synchronise after every stage and report where the levels' time goes
  const bool stages = verbose && std::atoi(std::getenv("BDPT_BUILD_VERBOSE")) >= 2;
  enum { S_BOUNDS, S_PREPARE, S_SMALL, S_BIN, S_DECIDE, S_FLAGS, S_SCAN, S_NLEFT, S_MEDIAN, S_CHILDREN, S_SCATTER, S_COUNT };
  static const char* const stageName[S_COUNT] = {"bounds", "prepare", "small", "bin", "decide", "flags", "scan", "nleft+read", "median", "children", "scatter"};
  double stageTime[S_COUNT] = {};
  auto ts = std::chrono::steady_clock::now();
  auto stage = [&](int id) {
    if (!stages) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    stageTime[id] += std::chrono::duration<double>(t1 - ts).count();
    ts = t1;
  };
  BvhDeviceArena& pool = build->arena;
  BvhDeviceArenaScope scratch{pool};
  struct Owned {  // the references of makeReferencesOnDevice, once taken over
    void* p = nullptr;
    ~Owned() {
      if (p) (void)hipFree(p);
    }
  } takenRefs;
  hipStream_t st = nullptr;  // the default stream: the build is a blocking call of set-up, not of the frame
  auto ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err = std::string("device tree builder: ") + what + ": " + hipGetErrorString(e);
    return false;
  };
  const size_t cap = n;                  // active slots per level: disjoint non-empty ranges
  const size_t maxNodes = 2 * (size_t)n;  // a binary tree over n references has at most 2n - 1 nodes
  BvhBuildRef *rA = nullptr, *rB = nullptr;
  uint32_t *ofA = nullptr, *ofB = nullptr, *F = nullptr, *S = nullptr, *G = nullptr, *R = nullptr, *bins = nullptr, *bigList = nullptr, *bigCount = nullptr;
  Counters* C = nullptr;
  Level L[2];
  Tree T;
  Scan scan;
  const size_t binWords = ((size_t)n / (kSmall + 1) + 1) * 3 * kBvhBins * kBinWords;  // (binsOf)
  if (!refs) {  // the references of makeReferencesOnDevice: this function's from here on
    if (!build->refs || build->numMadeRefs != n) {
      err = "device tree builder: no references of this build in device memory";
      return false;
    }
    rA = build->refs;
    takenRefs.p = rA;
    build->refs = nullptr;
  }
  if ((refs && !devAllocT(pool, &rA, n, err)) || !devAllocT(pool, &rB, n, err) || !devAllocT(pool, &ofA, n, err) || !devAllocT(pool, &ofB, n, err) ||
      !devAllocT(pool, &F, (size_t)n + 1, err) || !devAllocT(pool, &S, (size_t)n + 1, err) || !devAllocT(pool, &G, cap + 1, err) ||
      !devAllocT(pool, &R, cap + 1, err) || !devAllocT(pool, &bins, binWords, err) || !devAllocT(pool, &bigList, cap, err) ||
      !devAllocT(pool, &bigCount, 1, err) || !devAllocT(pool, &C, 1, err) || !allocLevel(pool, L[0], cap, err) || !allocLevel(pool, L[1], cap, err) ||
      !devAllocT(pool, &T.box, maxNodes * 6, err) || !devAllocT(pool, &T.left, maxNodes, err) || !devAllocT(pool, &T.right, maxNodes, err) ||
      !devAllocT(pool, &T.first, maxNodes, err) || !devAllocT(pool, &T.count, maxNodes, err) || !devAllocT(pool, &T.depth, maxNodes, err))
    return false;
  scan.capTiles = (uint32_t)((std::max<size_t>(n, cap) + kScanTile - 1) / kScanTile) + 1;
  if (!devAllocT(pool, &scan.sums, scan.capTiles, err) || !devAllocT(pool, &scan.total, 1, err)) return false;
  lap("alloc");
  if (refs && !ok(hipMemcpy(rA, refs, (size_t)n * sizeof(BvhBuildRef), hipMemcpyHostToDevice), "upload")) return false;
  lap("upload");

  const dim3 blk(256);
  auto gridFor = [](size_t m) { return dim3((unsigned)((m + 255) / 256)); };
  auto chunksOf = [](size_t m) { return dim3((unsigned)((m + kChunk - 1) / kChunk)); };
  hipLaunchKernelGGL(k_fill_u32, dim3(1024), blk, 0, st, ofA, n <= kSmall ? kSmallBit : 0u, (size_t)n);
  hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, st, L[0], T, n);
  uint32_t nA = 1, numNodes = 1;
  int cur = 0;
  build->levelStart.assign({0u, 1u});
  uint32_t *selHist = nullptr, *selHi = nullptr, *selLo = nullptr, *selK = nullptr;  // (allocated when a level first needs them)
  for (int level = 0; nA > 0; level++) {
    if (level > kBvhBinaryMaxDepth + 2) {
      err = "device tree builder: depth budget exceeded";
      return false;
    }
    Level& Lc = L[cur];
    Level& Ln = L[cur ^ 1];
    if (!ok(hipMemsetAsync(C, 0, sizeof(Counters), st), "memset")) return false;
    if (stages) {
      (void)hipDeviceSynchronize();
      ts = std::chrono::steady_clock::now();
    }
    const auto tLevel = ts;
    hipLaunchKernelGGL(k_bounds, chunksOf(n), blk, 0, st, rA, ofA, n, Lc);
    stage(S_BOUNDS);
    hipLaunchKernelGGL(k_prepare, gridFor(nA), blk, 0, st, Lc, T, nA, C, bins);
    stage(S_PREPARE);
    hipLaunchKernelGGL(k_small, dim3(nA), dim3(64), 0, st, rA, Lc, T, nA, C);
    stage(S_SMALL);
    hipLaunchKernelGGL(k_bin, chunksOf(n), blk, 0, st, rA, ofA, n, Lc, bins);
    stage(S_BIN);
    hipLaunchKernelGGL(k_decide, gridFor(nA), blk, 0, st, Lc, nA, bins, C);
    stage(S_DECIDE);
    hipLaunchKernelGGL(k_flags, gridFor(n), blk, 0, st, rA, ofA, n, Lc, F);
    stage(S_FLAGS);
    scan.run(F, n, S, st);
    stage(S_SCAN);
    hipLaunchKernelGGL(k_nleft, gridFor(nA), blk, 0, st, Lc, nA, S, C);
    Counters hc;
    if (!ok(hipMemcpy(&hc, C, sizeof(hc), hipMemcpyDeviceToHost), "counters")) return false;
    stage(S_NLEFT);
    if (hc.numMedianNew > 0) {
      // median fallback: every such slot's pivot (by one wave's ranking, or by a radix select), then the flags again
      hipLaunchKernelGGL(k_median_pivot_small, dim3(nA), dim3(64), 0, st, rA, Lc, nA);
      if (hc.numMedianBig > 0) {
        const uint32_t capBig = n / (kMedianWave + 1) + 1;  // (disjoint ranges of more than kMedianWave references)
        if (!selHist && (!devAllocT(pool, &selHist, (size_t)capBig * 256, err) || !devAllocT(pool, &selHi, capBig, err) || !devAllocT(pool, &selLo, capBig, err) ||
                         !devAllocT(pool, &selK, capBig, err)))
          return false;
        if (!ok(hipMemsetAsync(bigCount, 0, 4, st), "memset") || !ok(hipMemsetAsync(selHist, 0, (size_t)capBig * 256 * 4, st), "memset")) return false;
        hipLaunchKernelGGL(k_median_list_big, gridFor(nA), blk, 0, st, Lc, nA, bigList, bigCount, selHi, selLo, selK);
        uint32_t m = 0;
        if (!ok(hipMemcpy(&m, bigCount, 4, hipMemcpyDeviceToHost), "list")) return false;
        for (int pass = 7; pass >= 0 && m > 0; pass--) {
          hipLaunchKernelGGL(k_sel_hist, gridFor(n), blk, 0, st, rA, ofA, n, Lc, pass, selHi, selLo, selHist);
          hipLaunchKernelGGL(k_sel_pick, gridFor(m), blk, 0, st, m, pass, selHi, selLo, selK, selHist, Lc, bigList);
        }
      }
      hipLaunchKernelGGL(k_flags, gridFor(n), blk, 0, st, rA, ofA, n, Lc, F);
      scan.run(F, n, S, st);
      hipLaunchKernelGGL(k_nleft, gridFor(nA), blk, 0, st, Lc, nA, S, C);
    }
    stage(S_MEDIAN);
    // children: slots that split, ranked in slot order
    hipLaunchKernelGGL(k_split_flags, gridFor(nA), blk, 0, st, Lc, nA, G);
    scan.run(G, nA, R, st);
    uint32_t numSplit = 0;
    if (!ok(hipMemcpy(&numSplit, R + nA, 4, hipMemcpyDeviceToHost), "split count")) return false;
    if ((size_t)numNodes + 2 * (size_t)numSplit > maxNodes || 2 * (size_t)numSplit > cap) {
      err = "device tree builder: node count out of bounds";
      return false;
    }
    if (numSplit > 0) hipLaunchKernelGGL(k_children, gridFor(nA), blk, 0, st, Lc, nA, R, Ln, T, numNodes);
    stage(S_CHILDREN);
    hipLaunchKernelGGL(k_scatter, gridFor(n), blk, 0, st, rA, ofA, n, Lc, F, S, rB, ofB);
    stage(S_SCATTER);
    if (stages)
      std::fprintf(stderr, "[bvh]   device level %2d: %8u slots, %8u split, %.4f s\n", level, nA, numSplit, std::chrono::duration<double>(ts - tLevel).count());
    std::swap(rA, rB);
    std::swap(ofA, ofB);
    numNodes += 2 * numSplit;
    if (numSplit > 0) build->levelStart.push_back(numNodes);
    nA = 2 * numSplit;
    cur ^= 1;
    if (!ok(hipGetLastError(), "launch")) return false;
  }
  if (!ok(hipDeviceSynchronize(), "synchronise")) return false;
  lap("levels");
  if (stages)
    for (int i = 0; i < S_COUNT; i++) std::fprintf(stderr, "[bvh]   device stage %-10s %.4f s\n", stageName[i], stageTime[i]);
  // back to the host: the leaf order and the tree, in the host's formats (the boxes of the references are the caller's own)
  // (both stay in device memory for the packer: the build object owns them from here on)
  build->releaseTree();
  {
    void *q0 = nullptr, *q1 = nullptr;
    if (hipMalloc(&q0, (size_t)n * 4) != hipSuccess || hipMalloc(&q1, (size_t)numNodes * sizeof(BvhBuildNode)) != hipSuccess) {
      if (q0) (void)hipFree(q0);
      err = "device tree builder: out of device memory";
      return false;
    }
    build->order = static_cast<uint32_t*>(q0);
    build->nodes = static_cast<BvhBuildNode*>(q1);
    build->numRefs = n;
    build->numNodes = numNodes;
  }
  uint32_t* const dOrder = build->order;
  BvhBuildNode* const dNodes = build->nodes;
  hipLaunchKernelGGL(k_order, gridFor(n), blk, 0, st, rA, n, dOrder);
  hipLaunchKernelGGL(k_pack_nodes, gridFor(numNodes), blk, 0, st, T, numNodes, dNodes);
  if (!ok(hipDeviceSynchronize(), "pack")) return false;
  order.resize(refs ? n : 0);  // (references that never left the device: the packer reads the order there)
  const bool keepHere = build->collapseHere && !refs;  // the collapse happens here too: only the root goes back (a tree was built)
  nodes.resize(keepHere ? 1 : numNodes);
  if ((refs && !downloadStaged(order.data(), dOrder, (size_t)n * 4, err)) ||
      !downloadStaged(nodes.data(), dNodes, (size_t)(keepHere ? 1 : numNodes) * sizeof(BvhBuildNode), err))
    return false;
  lap("download");
  return true;
}

// The BvhPacker bdpt_set_scene plugs into buildBvh (after buildBinaryTreeOnDevice of the same build).
bool packOnDevice(void* user, const BvhPackInput& in, Bvh& out, std::string& err) {
  BvhDeviceBuild* const build = static_cast<BvhDeviceBuild*>(user);
  const bool collapse = in.wide == nullptr;  // the four-wide collapse has not happened yet: it happens here
  if (!build || !build->nodes || !build->order || build->numRefs != in.numRefs || (!collapse && (in.numWide == 0 || in.numWide > 0x7fffffffull))) {
    err = "device packer: no tree of this build in device memory";
    return false;
  }
  if (hipSetDevice(build->device) != hipSuccess) {
    err = "device packer: no such device";
    return false;
  }
  const bool verbose = std::getenv("BDPT_BUILD_VERBOSE") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[bvh]   device %-10s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  BvhDeviceArena& pool = build->arena;
  BvhDeviceArenaScope scratch{pool};
  auto ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err = std::string("device packer: ") + what + ": " + hipGetErrorString(e);
    return false;
  };
  uint32_t nWide = (uint32_t)in.numWide;
  BvhTri* dTri = nullptr;
  uint32_t *dRefTri = nullptr, *dBlock = nullptr, *dBase = nullptr, *dPos = nullptr;
  BvhWideNode* dWide = nullptr;
  BvhSlot* dSlots = nullptr;
  BvhRec* dTmp = nullptr;
  Scan scan;
  const bool kept = in.refTri == nullptr;  // the reference maker of this build kept refTri and the triangle records
  if (kept && (!build->refTri || !build->triRecs || build->numMadeRefs != in.numRefs || build->numTris != in.numTris)) {
    err = "device packer: no references of this build in device memory";
    return false;
  }
  if (kept) {
    dTri = build->triRecs;
    dRefTri = build->refTri;
  }
  if (collapse) {
    if (!collapseOnDevice(build, pool, &dWide, &dSlots, &nWide, out, err)) return false;
    lap("collapse");
  }
  if ((!kept && (!devAllocT(pool, &dTri, in.numTris, err) || !devAllocT(pool, &dRefTri, in.numRefs, err))) || (!collapse && !devAllocT(pool, &dWide, nWide, err)) ||
      (!collapse && !devAllocT(pool, &dSlots, nWide, err)) || !devAllocT(pool, &dTmp, nWide, err) || !devAllocT(pool, &dBlock, (size_t)nWide + 1, err) ||
      !devAllocT(pool, &dBase, (size_t)nWide + 1, err) || !devAllocT(pool, &dPos, nWide, err))
    return false;
  scan.capTiles = (nWide + kScanTile - 1) / kScanTile + 1;
  if (!devAllocT(pool, &scan.sums, scan.capTiles, err) || !devAllocT(pool, &scan.total, 1, err)) return false;
  if ((!kept && (!ok(hipMemcpy(dTri, in.triRecs, (size_t)in.numTris * sizeof(BvhTri), hipMemcpyHostToDevice), "upload") ||
                 !ok(hipMemcpy(dRefTri, in.refTri, (size_t)in.numRefs * 4, hipMemcpyHostToDevice), "upload"))) ||
      (!collapse && (!ok(hipMemcpy(dWide, in.wide, (size_t)nWide * sizeof(BvhWideNode), hipMemcpyHostToDevice), "upload") ||
                     !ok(hipMemcpy(dSlots, in.slots, (size_t)nWide * sizeof(BvhSlot), hipMemcpyHostToDevice), "upload"))))
    return false;
  lap("pack upload");
  hipStream_t st = nullptr;
  const dim3 blk(256), grid((nWide + 255) / 256);
  hipLaunchKernelGGL(k_quantise, grid, blk, 0, st, dWide, nWide, build->nodes, in.pad, dTmp, dBlock);
  scan.run(dBlock, nWide, dBase, st);
  uint32_t total = 0;
  if (!ok(hipMemcpy(&total, dBase + nWide, 4, hipMemcpyDeviceToHost), "block sizes")) return false;
  const uint64_t next = 1ull + total;
  if (next >= 0x7fffffffull || total < nWide - 1) {  // (the 32-bit sum wrapped, or the format's 2^31 records are not enough)
    err = "bvh does not fit the packed record format (2^31 records)";
    return false;
  }
  void* q = nullptr;
  const size_t numRecs = (size_t)next + kBvhPadRecs;
  if (hipMalloc(&q, numRecs * sizeof(BvhRec)) != hipSuccess) {
    err = "device packer: out of device memory";
    return false;
  }
  BvhRec* const dRecs = static_cast<BvhRec*>(q);
  bool good = ok(hipMemsetAsync(dRecs + next, 0, kBvhPadRecs * sizeof(BvhRec), st), "memset");
  if (good) {
    hipLaunchKernelGGL(k_node_pos, grid, blk, 0, st, dSlots, nWide, dTmp, dBase, dPos);
    hipLaunchKernelGGL(k_write_recs, grid, blk, 0, st, dWide, nWide, build->nodes, dTmp, dBase, dPos, build->order, dRefTri, dTri, dRecs);
    good = ok(hipGetLastError(), "launch") && ok(hipDeviceSynchronize(), "synchronise");
  }
  if (!good) {
    (void)hipFree(dRecs);
    return false;
  }
  lap("pack");
  build->release();
  out.deviceRecs = dRecs;
  out.deviceNumRecs = numRecs;
  return true;
}

// The BvhRefMaker bdpt_set_scene plugs into buildBvh: the references stay in device memory (in the build object) for
// buildBinaryTreeOnDevice and packOnDevice.
bool makeReferencesOnDevice(void* user, const BvhRefInput& in, uint32_t& numRefs, std::string& err) {
  BvhDeviceBuild* const build = static_cast<BvhDeviceBuild*>(user);
  if (!build) {
    err = "device reference maker: no build object";
    return false;
  }
  if (hipSetDevice(build->device) != hipSuccess) {
    err = "device reference maker: no such device";
    return false;
  }
  build->release();
  numRefs = 0;
  if (in.numTris == 0) return true;
  const bool verbose = std::getenv("BDPT_BUILD_VERBOSE") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[bvh]   device %-10s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  BvhDeviceArena& pool = build->arena;
  BvhDeviceArenaScope scratch{pool};
  auto ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    err = std::string("device reference maker: ") + what + ": " + hipGetErrorString(e);
    return false;
  };
  auto upload = [&](auto** dst, const auto* src, size_t count) {
    if (!devAllocT(pool, dst, count, err)) return false;
    return count == 0 || uploadStagedImpl(*dst, src, count * sizeof(**dst), err);
  };
  const uint32_t nT = in.numTris;
  RefArgs A{};
  {  // the triangle records stay for the packer
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)nT * sizeof(BvhTri)) != hipSuccess) {
      err = "device reference maker: out of device memory";
      return false;
    }
    build->triRecs = static_cast<BvhTri*>(q);
    build->numTris = nT;
  }
  A.triRecs = build->triRecs;
  BvhBox* dTriBox = nullptr;
  uint32_t* dSplits = nullptr;
  uint8_t* dState = nullptr;
  uint32_t* dIdxAll = nullptr;  // the index list, shared with the clipper's tables below when they name the same array
  const bool decideHere = in.splits == nullptr;  // classification, priorities and split counts happen here too
  // BDPT_UPLOAD_TRI_RECS (measurement knob): hand the host's records and boxes over instead of making them here
  if (in.positions && in.indices && in.numVertices && nT && std::getenv("BDPT_UPLOAD_TRI_RECS") == nullptr) {
    // records and boxes made here from what they are made of: 12 B per vertex + 20 B per triangle cross the bus
    // instead of 72 B per triangle
    float* dPos = nullptr;
    uint32_t *dFlags = nullptr, *dAux = nullptr;
    if (!upload(&dPos, in.positions, (size_t)in.numVertices * 3) || !upload(&dIdxAll, in.indices, (size_t)nT * 3)) return false;
    if (in.triFlags && !upload(&dFlags, in.triFlags, nT)) return false;
    if (in.triAux && !upload(&dAux, in.triAux, nT)) return false;
    if (!devAllocT(pool, &dTriBox, nT, err)) return false;
    hipLaunchKernelGGL(k_tri_recs, dim3((nT + 255) / 256), dim3(256), 0, nullptr, dPos, dIdxAll, dFlags, dAux, nT, build->triRecs, dTriBox);
  } else {
    if (!in.triRecs || !in.triBox) {
      err = "device reference maker: neither triangle records nor what they are made of were handed over";
      return false;
    }
    if (!uploadStagedImpl(build->triRecs, in.triRecs, (size_t)nT * sizeof(BvhTri), err)) return false;
    if (!upload(&dTriBox, in.triBox, nT)) return false;
  }
  if (decideHere ? (!devAllocT(pool, &dSplits, nT, err) || !devAllocT(pool, &dState, nT, err)) : (!upload(&dSplits, in.splits, nT) || !upload(&dState, in.state, nT))) return false;
  A.triBox = dTriBox;
  A.splits = dSplits;
  A.state = dState;
  A.numTris = nT;
  for (int a = 0; a < 3; a++) {
    A.gridLo[a] = in.gridLo[a];
    A.gridExt[a] = in.gridExt[a];
  }
  if (in.clipper) {
    BvhClipTables tb;
    if (!in.clipper->tables(tb) || tb.numTriangles != nT) {
      err = "device reference maker: the clipper's decisions are not available as tables";
      return false;
    }
    uint32_t *dMat = nullptr, *dIdx = nullptr;
    float* dTex = nullptr;
    int32_t *dMask = nullptr, *dVerdict = nullptr;
    DevMask* dMasks = nullptr;
    if (dIdxAll && tb.indices == in.indices)
      dIdx = dIdxAll;  // (already here)
    else if (!upload(&dIdx, tb.indices, (size_t)nT * 3))
      return false;
    if (!upload(&dMat, tb.triMaterial, nT) || !upload(&dMask, tb.matMask.data(), tb.matMask.size()) ||
        !upload(&dVerdict, tb.matVerdict.data(), tb.matVerdict.size()))
      return false;
    if (tb.texcoords && !upload(&dTex, tb.texcoords, (size_t)tb.numVertices * 3)) return false;
    std::vector<DevMask> masks;
    for (const BvhClipTables::Mask& m : tb.masks) {
      uint32_t* dSat = nullptr;
      if (!upload(&dSat, m.mayPass, ((size_t)m.w + 1) * ((size_t)m.h + 1))) return false;
      masks.push_back(DevMask{m.w, m.h, dSat});
    }
    if (!upload(&dMasks, masks.data(), masks.size())) return false;
    A.haveClipper = 1;
    A.triMaterial = dMat;
    A.indices = dIdx;
    A.texcoords = dTex;
    A.matMask = dMask;
    A.matVerdict = dVerdict;
    A.masks = dMasks;
  }
  lap("refs upload");
  hipStream_t st = nullptr;
  const dim3 blk(256), grid((nT + 255) / 256);
  unsigned long long* dSum = nullptr;
  if (!devAllocT(pool, &dSum, 4, err)) return false;
  if (decideHere) {
    double* prio = nullptr;
    float* capOf = nullptr;
    if (!devAllocT(pool, &prio, nT, err) || !devAllocT(pool, &capOf, nT, err)) return false;
    hipLaunchKernelGGL(k_prio, dim3((nT + 63) / 64), dim3(64), 0, st, A, in.budgetOpaque, in.budgetAlpha, in.outlierArea, dState, prio, capOf, dSplits);
    for (int cls = 0; cls < 2; cls++) {  // split counts per class: the largest D with sum min(floor(D p), cap) <= budget
      const float budgetF = cls ? in.budgetAlpha : in.budgetOpaque;
      if (!(budgetF > 0.0f)) continue;
      unsigned long long stats[2] = {0, 0};
      if (!ok(hipMemsetAsync(dSum, 0, 16, st), "memset")) return false;
      hipLaunchKernelGGL(k_class_stats, grid, blk, 0, st, A, dState, prio, cls, dSum);
      if (!ok(hipMemcpy(stats, dSum, 16, hipMemcpyDeviceToHost), "class")) return false;
      const uint64_t members = stats[0];
      double pmax;
      std::memcpy(&pmax, &stats[1], 8);
      const uint64_t budget = (uint64_t)((double)members * (double)budgetF);
      if (!members || !budget || !(pmax > 0.0)) continue;
      bool failed = false;
      auto total = [&](double D) -> uint64_t {
        unsigned long long v = 0;
        if (hipMemsetAsync(dSum, 0, 8, st) != hipSuccess) failed = true;
        hipLaunchKernelGGL(k_split_total, dim3(1024), blk, 0, st, A, dState, prio, capOf, cls, D, dSum);
        if (hipMemcpy(&v, dSum, 8, hipMemcpyDeviceToHost) != hipSuccess) failed = true;
        return v;
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
      if (failed) {
        err = "device reference maker: split counts failed";
        return false;
      }
      hipLaunchKernelGGL(k_split_assign, grid, blk, 0, st, A, dState, prio, capOf, cls, dLo, dSplits);
    }
    lap("refs decide");
  }
  uint32_t *capRefs = nullptr, *capStack = nullptr, *slotAt = nullptr, *stackAt = nullptr, *made = nullptr, *refAt = nullptr;
  Scan scan;
  scan.capTiles = (nT + kScanTile - 1) / kScanTile + 1;
  if (!devAllocT(pool, &capRefs, nT, err) || !devAllocT(pool, &capStack, nT, err) || !devAllocT(pool, &slotAt, (size_t)nT + 1, err) ||
      !devAllocT(pool, &stackAt, (size_t)nT + 1, err) || !devAllocT(pool, &made, nT, err) || !devAllocT(pool, &refAt, (size_t)nT + 1, err) ||
      !devAllocT(pool, &scan.sums, scan.capTiles, err) || !devAllocT(pool, &scan.total, 1, err))
    return false;
  hipLaunchKernelGGL(k_ref_caps, grid, blk, 0, st, A, capRefs, capStack);
  scan.run(capRefs, nT, slotAt, st);
  scan.run(capStack, nT, stackAt, st);
  uint32_t totalSlots = 0, totalStack = 0;
  if (!ok(hipMemcpy(&totalSlots, slotAt + nT, 4, hipMemcpyDeviceToHost), "caps") || !ok(hipMemcpy(&totalStack, stackAt + nT, 4, hipMemcpyDeviceToHost), "caps")) return false;
  bool plain = true;  // nothing to split, nothing shrunk
  {  // (the scans are 32-bit: make sure they did not wrap)
    unsigned long long sum[3] = {0, 0, 0};
    if (!ok(hipMemsetAsync(dSum, 0, 24, st), "memset")) return false;
    hipLaunchKernelGGL(k_ref_summary, grid, blk, 0, st, nT, A.state, A.splits, dSum);
    if (!ok(hipMemcpy(sum, dSum, 24, hipMemcpyDeviceToHost), "summary")) return false;
    if (sum[1] != totalSlots || sum[1] >= 0x7fffffffull) {
      err = "bvh does not fit the packed record format (2^31 records)";
      return false;
    }
    plain = sum[2] == 0;
    if (in.numDroppedOut) *in.numDroppedOut = (uint32_t)sum[0];
  }
  BvhBox* boxes = nullptr;
  DevPiece* stacks = nullptr;
  if (!devAllocT(pool, &boxes, totalSlots, err) || !devAllocT(pool, &stacks, totalStack, err)) return false;
  if (plain)
    hipLaunchKernelGGL(k_plain_refs, grid, blk, 0, st, A, slotAt, boxes, made);
  else
    hipLaunchKernelGGL(k_make_refs, dim3((nT + 63) / 64), dim3(64), 0, st, A, slotAt, stackAt, stacks, boxes, made);
  scan.run(made, nT, refAt, st);
  uint32_t total = 0;
  if (!ok(hipMemcpy(&total, refAt + nT, 4, hipMemcpyDeviceToHost), "count")) return false;
  lap("refs make");
  if (total > 0) {
    void *q0 = nullptr, *q1 = nullptr;
    if (hipMalloc(&q0, (size_t)total * sizeof(BvhBuildRef)) != hipSuccess || hipMalloc(&q1, (size_t)total * 4) != hipSuccess) {
      if (q0) (void)hipFree(q0);
      err = "device reference maker: out of device memory";
      return false;
    }
    build->refs = static_cast<BvhBuildRef*>(q0);
    build->refTri = static_cast<uint32_t*>(q1);
    hipLaunchKernelGGL(k_compact_refs, grid, blk, 0, st, nT, slotAt, made, refAt, boxes, build->refs, build->refTri);
  }
  if (!ok(hipGetLastError(), "launch") || !ok(hipDeviceSynchronize(), "synchronise")) return false;
  build->numMadeRefs = total;
  numRefs = total;
  lap("refs compact");
  return true;
}

bool bvhUploadStaged(void* dst, const void* src, size_t bytes, std::string& err) { return uploadStagedImpl(dst, src, bytes, err); }
void bvhPrewarmStaging(int device) {
  std::call_once(gPrewarm.once, [device] {
    std::lock_guard<std::mutex> j(gPrewarm.joinLock);
    gPrewarm.pending.store(true, std::memory_order_release);
    gPrewarm.th = std::thread([device] {
      if (hipSetDevice(device) != hipSuccess) return;
      for (int i = 0; i < 4; i++) {
        void* p = nullptr;
        if (hipHostMalloc(&p, kUpStage, hipHostMallocPortable) != hipSuccess) return;
        giveStage(p);
      }
    });
  });
}

}  // namespace bdpt
