// alpha_clip.h — what the any-hit alpha test (BDPT/BDPTUtils.hlsli:115-127 alphaTestFails; device_scene.hpp) can and
// cannot let through, decided per texel at bdpt_set_scene so that the acceleration structure only bounds what can be
// hit.  The reference decides opacity per MESH (Falcor Raytracing/RtModel.cpp:221-224: a BLAS geometry is OPAQUE iff its
// material's alpha mode is); this build decides it per triangle and per piece of a triangle:
//   * a triangle whose bilinear footprint only touches texels at or above the threshold always passes: it loses its
//     non-opaque flag and its hits skip the test;
//   * a piece of a triangle whose footprint only touches texels below the threshold can never be reported: it gets no
//     reference in the tree (an always-ignored candidate is never a hit);
//   * a reference only bounds the part of its piece where the test can pass (BvhRefClipper, bvh.h).
// All three leave every reported hit — and so every image — unchanged.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/bdpt.h"
#include "bvh.h"

namespace bdpt {

class AlphaClipper : public BvhRefClipper {
 public:
  AlphaClipper(const bdpt_scene_desc* d);
  // 0: both outcomes occur (or could not be decided), 1: the test always passes, 2: it always fails
  int classify(uint32_t tri) const;
  bool clip(uint32_t tri, double (*poly)[2], int& n) const override;
  bool tables(BvhClipTables& out) const override;
  // the device's alpha test in the same fp32 arithmetic (device_scene.hpp alphaTestFails), for the host-side trace hook
  bool testFails(uint32_t tri, float bu, float bv) const;

 private:
  struct Mask {
    int w = 0, h = 0;
    std::vector<uint32_t> mayPass, mayFail;  // summed-area tables, (w + 1) x (h + 1): cells a sample may pass / fail in
    uint32_t count(const std::vector<uint32_t>& sat, long x0, long x1, long y0, long y1) const;
  };
  struct MatInfo {
    int mask = -1;        // index into masks, -1: no texture decides
    int verdict = 0;      // for mask < 0: 1 always passes, 2 always fails
  };
  bool cellRect(uint32_t tri, const double (*poly)[2], int n, const Mask*& m, long& x0, long& x1, long& y0, long& y1, double& margin,
                double uv[3][2]) const;
  const bdpt_scene_desc* d_;
  std::vector<MatInfo> mats_;
  std::vector<Mask> masks_;
};

}  // namespace bdpt
