// scene_bvh.h — from a scene description to the acceleration structure bdpt_set_scene uploads: per-triangle traversal
// flags, the alpha classification of alpha_clip.h, and the build.  Shared by bdpt_set_scene (api.cpp) and the
// host-side trace hook (bdpt_host_bvh_*), which walks the same tree on the CPU so that the builder can be tested
// and its trees compared without a GPU.
#pragma once
#include <memory>
#include <vector>

#include "alpha_clip.h"
#include "bvh.h"

namespace bdpt {

struct SceneBvh {
  std::vector<uint32_t> triFlags;   // kTriNonOpaque | kTriDoubleSided per triangle, after classification
  std::vector<uint32_t> triAux;     // index of the alpha-test record of a (still) non-opaque triangle
  std::vector<uint32_t> alphaTris;  // the triangles that have an alpha-test record, in record order
  uint32_t numAlphaMode = 0;        // triangles whose material is not AlphaModeOpaque
  uint32_t numAlwaysPass = 0;       // of those: the test always passes (flag cleared)
  std::unique_ptr<AlphaClipper> clipper;
  Bvh bvh;
};

// threads / budgets as BvhBuildOptions (negative budgets: build defaults); classify = false keeps the reference's
// per-material opacity (every triangle of an alpha-mode material is tested, nothing is dropped or clipped).
// refMaker / treeBuilder / packer (may be null: host code) make the references, build the binary tree and pack the records — bdpt_set_scene passes the
// device implementations; when one fails *error says why (no nodes / no device records then).
void buildSceneBvh(const bdpt_scene_desc* d, int threads, float splitBudget, float splitBudgetAlpha, bool classify, SceneBvh& out,
                   BvhTreeBuilder treeBuilder = nullptr, void* treeBuilderUser = nullptr, std::string* error = nullptr, BvhPacker packer = nullptr,
                   BvhRefMaker refMaker = nullptr, bool collapseInPacker = false, bool prioritiesInRefMaker = false);

}  // namespace bdpt
