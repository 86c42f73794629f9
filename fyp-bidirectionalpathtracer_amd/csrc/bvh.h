// bvh.h — host-side acceleration-structure build.  Replaces the DXR driver's
// BuildRaytracingAccelerationStructure calls issued by Falcor
// (Raytracing/RtModel.cpp:181-254 bottom level, Raytracing/RtScene.cpp:220-308 top level);
// instancing is flattened on input so one level suffices.
#pragma once
#include <cstdint>
#include <vector>

namespace bdpt {

// 64-byte two-child node: both children's boxes + two child references.
//   ref >= 0 : interior node index
//   ref <  0 : leaf, -1 - ((firstTriangle << 3) | (count - 1)), count in 1..8
// An absent child has an inverted box (lo > hi) and is never entered.
struct alignas(16) BvhNode {
  float lo0[3], hi0x;  // child 0: lo.xyz, hi.x
  float hi0yz[2], lo1xy[2];
  float lo1z, hi1[3];
  int32_t child0, child1;
  int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "node must be 64 bytes");

// 48-byte leaf triangle as intersected: v0, e1 = v1 - v0, e2 = v2 - v0 (+ ids in the w lanes).
struct alignas(16) BvhTri {
  float v0[3];
  uint32_t prim;  // index into the caller's triangle list
  float e1[3];
  uint32_t flags;  // bit0 non-opaque (any-hit alpha test), bit1 double-sided (no back-face cull)
  float e2[3];
  uint32_t pad;
};
static_assert(sizeof(BvhTri) == 48, "triangle must be 48 bytes");

constexpr int kBvhMaxDepth = 30;  // traversal stack holds 32 entries per lane
constexpr uint32_t kTriNonOpaque = 1u, kTriDoubleSided = 2u;

struct Bvh {
  std::vector<BvhNode> nodes;
  std::vector<BvhTri> tris;  // in leaf order
  uint32_t maxDepth = 0;
  float sahCost = 0.0f;
};

// positions: 3 floats per vertex; indices: 3 per triangle; triFlags: per triangle (may be null).
void buildBvh(const float* positions, const uint32_t* indices, uint32_t numTriangles, const uint32_t* triFlags, Bvh& out);

}  // namespace bdpt
