// Compile check for host/ReferenceNames.h: a program written the way the reference's application entry point is
// (BidirectionalPathtracing/Main.cpp:9-29: construct the pipeline, set four passes by their global class names, fill
// a SampleConfig, hand both to RenderingPipeline::run) must compile against the host mirror without any
// `using namespace`.  Written for this test; built with -fsyntax-only by tests/test_cpu_oracle_and_host.py.
#include "ReferenceNames.h"

int main() {
  RenderingPipeline* pipeline = new RenderingPipeline();
  pipeline->setPass(0, LightProbeGBufferPass::create());
  pipeline->setPass(1, BDPTPass::create(ResourceManager::kOutputChannel));
  pipeline->setPass(2, SimpleAccumulationPass::create(ResourceManager::kOutputChannel));
  pipeline->setPass(3, BlockwiseMultiOrderFeatureRegression::create());
  SampleConfig config;
  config.windowDesc.resizableWindow = true;
  config.windowDesc.width = 1280;
  config.windowDesc.height = 720;
  config.windowDesc.title = "compile check";
  RenderingPipeline::run(pipeline, config);
  // the pass interface under its global name: a user-defined pass derives from ::RenderPass
  struct MyPass : public RenderPass {
    MyPass() : RenderPass("mine", "mine") {}
    bool initialize(RenderContext*, ResourceManager::SharedPtr pResManager) override {
      mpResManager = pResManager;
      return pResManager->requestTextureResource("Mine", ResourceFormat::RGBA16Float) >= 0;
    }
    void execute(RenderContext*) override {}
  };
  RenderPass::SharedPtr mine(new MyPass());
  return mine->getName() == "mine" ? 0 : 1;
}
