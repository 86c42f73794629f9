// Compile-and-link check for the multi-GPU host surface (host/Tiling.h, RenderingPipeline::setTiling): what a maintainer's
// program looks like when N GPUs of one node render one frame together — one pipeline per GPU on a thread of its own,
// RCCL communicators from ncclCommInitAll.  Written for this test; built, linked against libbdpt_amd.so + librccl and
// run by tests/test_cpu_oracle_and_host.py (without a GPU it reports that and exits 0 before touching RCCL).
#include <cstdio>
#include <thread>
#include <vector>

#include "ReferenceNames.h"

int main() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    std::printf("tiled host: no HIP device, nothing rendered\n");
    return 0;
  }
  std::vector<int> devs((size_t)n);
  for (int i = 0; i < n; i++) devs[(size_t)i] = i;
  std::vector<ncclComm_t> comms((size_t)n);
  if (ncclCommInitAll(comms.data(), n, devs.data()) != ncclSuccess) return 1;
  std::vector<std::vector<float>> frames((size_t)n);
  std::vector<std::thread> ranks;
  for (int r = 0; r < n; r++)
    ranks.emplace_back([&, r] {
      RenderingPipeline* pipeline = new RenderingPipeline();
      pipeline->setPass(0, LightProbeGBufferPass::create());
      pipeline->setPass(1, BDPTPass::create(ResourceManager::kOutputChannel));
      pipeline->setPass(2, SimpleAccumulationPass::create(ResourceManager::kOutputChannel));
      pipeline->setSize(256, 144, devs[(size_t)r]);
      pipeline->setFramesInFlight(2);
      if (pipeline->setTiling((uint32_t)r, (uint32_t)n, comms[(size_t)r]) && pipeline->initialize(Scene::createCornellBox())) {
        for (int f = 0; f < 4; f++) pipeline->renderFrame();
        frames[(size_t)r] = pipeline->readOutput();  // collective: every rank gets the whole frame
      }
      delete pipeline;
    });
  for (std::thread& t : ranks) t.join();
  for (int r = 0; r < n; r++) ncclCommDestroy(comms[(size_t)r]);
  for (int r = 0; r < n; r++)
    if (frames[(size_t)r].size() != 256u * 144u * 4u || frames[(size_t)r] != frames[0]) return 1;
  std::printf("tiled host: %d rank(s), frame gathered on every rank\n", n);
  return 0;
}
