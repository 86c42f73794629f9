// main.cpp — headless counterpart of BidirectionalPathtracing/Main.cpp:9-29: the same pipeline
// (G-buffer pass -> BDPT pass -> accumulation pass -> BMFR denoiser, which is off unless --denoise /
// --denoise-regression tick its boxes), run for a number of frames, output written as a PFM image.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "ReferenceNames.h"  // the mirror's classes under the reference's global names: no `using namespace`

static void writePfm(const char* path, const std::vector<float>& rgba, uint32_t w, uint32_t h) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return;
  std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
  for (int y = (int)h - 1; y >= 0; y--)  // PFM rows go bottom to top
    for (uint32_t x = 0; x < w; x++) std::fwrite(&rgba[((size_t)y * w + x) * 4], 4, 3, f);
  std::fclose(f);
}

int main(int argc, char** argv) {
  std::string scene = "cornell", out = "bdpt_out.pfm", raw, checkpoint, resume, envFile;
  uint32_t W = 1280, H = 720;  // the reference's window, Main.cpp:23-24
  int frames = 8, depth = 3, mat = 0, device = 0, accumLimit = 100, inflight = 1, warmup = 0;
  bool denoise = false, denoiseRegression = false;
  for (int i = 1; i < argc; i++) {
    auto next = [&](const char* name) -> const char* {
      if (std::strcmp(argv[i], name) == 0 && i + 1 < argc) return argv[++i];
      return nullptr;
    };
    if (std::strcmp(argv[i], "--denoise") == 0) denoise = true;
    else if (std::strcmp(argv[i], "--denoise-regression") == 0) denoise = denoiseRegression = true;
    else if (const char* v = next("--scene")) scene = v;
    else if (const char* v = next("--width")) W = (uint32_t)std::atoi(v);
    else if (const char* v = next("--height")) H = (uint32_t)std::atoi(v);
    else if (const char* v = next("--frames")) frames = std::atoi(v);
    else if (const char* v = next("--depth")) depth = std::atoi(v);
    else if (const char* v = next("--mat")) mat = std::atoi(v);
    else if (const char* v = next("--device")) device = std::atoi(v);
    else if (const char* v = next("--accum-limit")) accumLimit = std::atoi(v);
    else if (const char* v = next("--out")) out = v;
    else if (const char* v = next("--raw")) raw = v;
    else if (const char* v = next("--checkpoint")) checkpoint = v;
    else if (const char* v = next("--resume")) resume = v;
    else if (const char* v = next("--env")) envFile = v;
    else if (const char* v = next("--inflight")) inflight = std::atoi(v);
    else if (const char* v = next("--warmup")) warmup = std::atoi(v);  // of --frames: rendered before the clock starts (first-use allocations)
    else {
      std::fprintf(stderr, "usage: bdpt_render [--scene cornell|atrium|FILE.fscene|FILE.obj] [--width W] [--height H] [--frames N] [--depth D] "
                           "[--mat 0|1] [--accum-limit N] [--denoise | --denoise-regression] [--out file.pfm] [--raw file.f32] "
                           "[--resume file.ckpt] [--checkpoint file.ckpt] [--env probe.hdr|image|Black] [--inflight N] [--warmup N]\n");
      return 2;
    }
  }
  // a path (.fscene / .obj) goes through the loader, as SceneLoaderWrapper::loadScene does for the file dialog's pick
  Scene::SharedPtr pScene;
  if (scene.find('.') != std::string::npos) {
    std::string err;
    pScene = Scene::loadFromFile(scene, &err);
    if (!pScene) {
      std::fprintf(stderr, "bdpt_render: %s\n", err.c_str());
      return 1;
    }
  } else {
    pScene = scene == "atrium" ? Scene::createAtrium(1, 262144) : Scene::createCornellBox();
  }

  // Create our rendering pipeline and add the passes, as Main.cpp:12-18 does
  RenderingPipeline* pipeline = new RenderingPipeline();
  pipeline->setPass(0, LightProbeGBufferPass::create());
  pipeline->setPass(1, BDPTPass::create(ResourceManager::kOutputChannel));
  pipeline->setPass(2, SimpleAccumulationPass::create(ResourceManager::kOutputChannel));
  pipeline->setPass(3, BlockwiseMultiOrderFeatureRegression::create());
  // the window parameters of Main.cpp:20-25 size the channels (there is no window)
  SampleConfig config;
  config.windowDesc.resizableWindow = true;
  config.windowDesc.width = W;
  config.windowDesc.height = H;
  config.windowDesc.title = "Bidirectional Path Tracing (headless)";
  pipeline->setSize(config.windowDesc.width, config.windowDesc.height, device);
  pipeline->setFramesInFlight((uint32_t)(inflight < 1 ? 1 : inflight));  // offline accumulation: frames overlap, same image
  if (!pipeline->initialize(pScene) || pipeline->getPassCount() != 4) {
    std::fprintf(stderr, "pipeline initialisation failed (no GPU?)\n");
    return 1;
  }
  // the light probe a user would pick in the file dialog (RenderingPipeline.cpp:229-243 -> ResourceManager::updateEnvironmentMap)
  if (!envFile.empty() && !pipeline->getResourceManager()->updateEnvironmentMap(envFile)) {
    std::fprintf(stderr, "bdpt_render: cannot load the environment map %s\n", envFile.c_str());
    return 1;
  }
  Gui gui;  // what a user would have set in the GUI windows
  gui.overrides["Max Ray Depth"] = depth;
  gui.overrides["Material"] = mat;
  gui.overrides["Max frames to accumulate"] = accumLimit;
  if (denoise) gui.overrides["Ignore the denoise stage"] = 1;  // the check box's label while it is off (DenoisePass.cpp:139)
  if (denoiseRegression) gui.overrides["Skip Regression"] = 1;
  pipeline->applyGui(&gui);
  if (!resume.empty() && !pipeline->loadCheckpoint(resume)) {  // continue an earlier run's frame sequence
    std::fprintf(stderr, "bdpt_render: cannot resume from %s (missing, or written for other passes / another frame size)\n", resume.c_str());
    return 1;
  }

  if (warmup < 0 || warmup >= frames) warmup = 0;
  for (int f = 0; f < warmup; f++) pipeline->renderFrame();
  (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int f = warmup; f < frames; f++) pipeline->renderFrame();
  (void)hipDeviceSynchronize();
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::vector<float> img = pipeline->readOutput();
  double mean = 0;
  for (size_t i = 0; i < img.size(); i += 4) mean += img[i] + img[i + 1] + img[i + 2];
  std::printf("%s %ux%u depth %d mat %d: %d frames in %.2f ms (%.2f ms/frame), mean radiance %.6f\n", scene.c_str(), W, H, depth, mat,
              frames - warmup, ms, ms / (frames - warmup), mean / (3.0 * (double)(img.size() / 4)));
  writePfm(out.c_str(), img, W, H);
  if (!checkpoint.empty() && !pipeline->saveCheckpoint(checkpoint)) {
    std::fprintf(stderr, "bdpt_render: cannot write %s\n", checkpoint.c_str());
    return 1;
  }
  if (!raw.empty()) {
    FILE* f = std::fopen(raw.c_str(), "wb");
    if (f) {
      std::fwrite(img.data(), 4, img.size(), f);
      std::fclose(f);
    }
  }
  delete pipeline;
  return 0;
}
