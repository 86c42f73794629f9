// main.cpp — headless counterpart of BidirectionalPathtracing/Main.cpp:9-29: the same pipeline
// (G-buffer pass -> BDPT pass -> accumulation pass -> BMFR denoiser, which is off unless --denoise /
// --denoise-regression tick its boxes), run for a number of frames, output written as a PFM image.
//
// Multi-GPU: `--gpus N` runs N such pipelines, one per GPU, each on a host thread of its own
// with its own RCCL communicator (ncclCommInitAll); `--rank R --world N --id-file F` is the same for one PROCESS per
// GPU (ncclCommInitRank; rank 0 hands the ncclUniqueId to the others through file F).  Every rank renders its
// interleaved stripes of the ONE frame (RenderingPipeline::setTiling), the splat accumulators are summed with one
// ncclReduceScatter per frame, the frame is assembled with ncclAllGather when it is written.  `--gpus 1` takes the
// same code path through a one-rank communicator.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include "RankSync.h"        // RankGroup: barrier + agreement point of the ranks of one process; the id-file protocol
#include "ReferenceNames.h"  // the mirror's classes under the reference's global names: no `using namespace`

static void writePfm(const char* path, const std::vector<float>& rgba, uint32_t w, uint32_t h) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return;
  std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
  for (int y = (int)h - 1; y >= 0; y--)  // PFM rows go bottom to top
    for (uint32_t x = 0; x < w; x++) std::fwrite(&rgba[((size_t)y * w + x) * 4], 4, 3, f);
  std::fclose(f);
}


namespace {
struct Options {
  std::string scene = "cornell", out = "bdpt_out.pfm", raw, checkpoint, resume, envFile, idFile, jobId;
  uint32_t W = 1280, H = 720;  // the reference's window, Main.cpp:23-24
  int frames = 8, depth = 3, mat = 0, device = 0, accumLimit = 100, inflight = 1, warmup = 0;
  int gpus = 0, rank = -1, world = 0;  // gpus: ranks as threads of this process; rank / world: this process is one rank
  bool denoise = false, denoiseRegression = false;
};

// How the ranks of a job stay together: a rank whose set-up fails must not leave its peers waiting in a
// barrier or a collective it never enters.  Ranks as threads: bdpt::RankGroup (agreement point after set-up; abort() from
// a rank that fails later, upon which main()'s watcher calls ncclCommAbort on every communicator so that collectives in
// flight return).  One process per rank: the agreement is an ncclAllReduce(min) of an ok flag, and a rank that fails
// later aborts its own communicator and exits non-zero.
struct RankEnv {
  bdpt::RankGroup* group = nullptr;  // ranks as threads of this process (null: one rank per process, or the plain run)
  ncclComm_t comm = nullptr;
  uint32_t rank = 0, world = 0;
  // BDPT_RENDER_INJECT_FAILURE="<rank>:setup" | "<rank>:readback" (tests): that rank behaves as if the step had failed
  bool injected(const char* stage) const {
    const char* e = std::getenv("BDPT_RENDER_INJECT_FAILURE");
    if (!e) return false;
    const std::string want = std::to_string(rank) + ":" + stage;
    return want == e;
  }
  // every rank reports how its set-up went; true only if all went well
  bool agree(bool ok) const {
    if (group) return group->agree(ok);
    if (world <= 1 || !comm) return ok;
    int* flag = nullptr;
    int mine = ok ? 1 : 0, all = 0;
    if (hipMalloc(reinterpret_cast<void**>(&flag), sizeof(int)) != hipSuccess) {
      ncclCommAbort(comm);  // cannot even take part: the peers' all-reduce must not wait for this rank
      return false;
    }
    bool done = hipMemcpy(flag, &mine, sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                ncclAllReduce(flag, flag, 1, ncclInt32, ncclMin, comm, nullptr) == ncclSuccess && hipStreamSynchronize(nullptr) == hipSuccess &&
                hipMemcpy(&all, flag, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(flag);
    return done && all == 1;
  }
  bool barrier() const { return group ? group->wait() : true; }
  // a failure after the agreement point
  void failLate() const {
    if (group) {
      group->abort();  // (idempotent; the watcher aborts every communicator once)
    } else if (comm && world > 1) {
      static std::atomic<bool> once{false};  // one rank per process: its one communicator is aborted once
      if (!once.exchange(true)) ncclCommAbort(comm);
    }
  }
};

struct RankResult {
  int rc = 1;
  double ms = 0;
};

Scene::SharedPtr makeScene(const Options& o, std::string* err) {
  // a path (.fscene / .obj) goes through the loader, as SceneLoaderWrapper::loadScene does for the file dialog's pick
  if (o.scene.find('.') != std::string::npos) return Scene::loadFromFile(o.scene, err);
  return o.scene == "atrium" ? Scene::createAtrium(1, 262144) : Scene::createCornellBox();
}

// set-up of one rank's pipeline; false (with a message) when any step fails
bool setUpRank(const Options& o, int device, uint32_t rank, uint32_t world, ncclComm_t comm, RenderingPipeline* pipeline) {
  const bool tiled = world > 0;
  std::string err;
  Scene::SharedPtr pScene = makeScene(o, &err);  // every rank holds the scene (and builds its BVH): replicated
  if (!pScene) {
    std::fprintf(stderr, "bdpt_render: %s\n", err.c_str());
    return false;
  }
  // add the passes to the rendering pipeline, as Main.cpp:12-18 does
  pipeline->setPass(0, LightProbeGBufferPass::create());
  pipeline->setPass(1, BDPTPass::create(ResourceManager::kOutputChannel));
  pipeline->setPass(2, SimpleAccumulationPass::create(ResourceManager::kOutputChannel));
  pipeline->setPass(3, BlockwiseMultiOrderFeatureRegression::create());
  // the window parameters of Main.cpp:20-25 size the channels (there is no window)
  SampleConfig config;
  config.windowDesc.resizableWindow = true;
  config.windowDesc.width = o.W;
  config.windowDesc.height = o.H;
  config.windowDesc.title = "Bidirectional Path Tracing (headless)";
  pipeline->setSize(config.windowDesc.width, config.windowDesc.height, device);
  pipeline->setFramesInFlight((uint32_t)(o.inflight < 1 ? 1 : o.inflight));  // offline accumulation: frames overlap, same image
  if (tiled && !pipeline->setTiling(rank, world, comm)) {
    std::fprintf(stderr, "bdpt_render: bad tiling (rank %u of %u)\n", rank, world);
    return false;
  }
  if (!pipeline->initialize(pScene) || pipeline->getPassCount() != 4) {
    std::fprintf(stderr, "pipeline initialisation failed (no GPU?)\n");
    return false;
  }
  // the light probe a user would pick in the file dialog (RenderingPipeline.cpp:229-243 -> ResourceManager::updateEnvironmentMap)
  if (!o.envFile.empty() && !pipeline->getResourceManager()->updateEnvironmentMap(o.envFile)) {
    std::fprintf(stderr, "bdpt_render: cannot load the environment map %s\n", o.envFile.c_str());
    return false;
  }
  Gui gui;  // what a user would have set in the GUI windows
  gui.overrides["Max Ray Depth"] = o.depth;
  gui.overrides["Material"] = o.mat;
  gui.overrides["Max frames to accumulate"] = o.accumLimit;
  if (o.denoise) gui.overrides["Ignore the denoise stage"] = 1;  // the check box's label while it is off (DenoisePass.cpp:139)
  if (o.denoiseRegression) gui.overrides["Skip Regression"] = 1;
  pipeline->applyGui(&gui);
  if (!o.resume.empty() && !pipeline->loadCheckpoint(o.resume)) {  // continue an earlier run's frame sequence
    std::fprintf(stderr, "bdpt_render: cannot resume from %s (missing, or written for other passes / another frame size)\n", o.resume.c_str());
    return false;
  }
  return true;
}

// One pipeline on one GPU.  world == 0: the plain single-GPU run.  Otherwise rank `rank` of `world` (comm may be null
// for world == 1).
RankResult runRank(const Options& o, int device, const RankEnv& env) {
  RankResult res;
  const uint32_t rank = env.rank, world = env.world;
  const bool tiled = world > 0, writer = !tiled || rank == 0;
  std::unique_ptr<RenderingPipeline> pipeline(new RenderingPipeline());
  // a rank that has to leave a collective later (TileExchange::abort) releases its peers the same way a failed read-back does
  pipeline->setTilingAbortHandler([env] { env.failLate(); });
  const bool setUp = !env.injected("setup") && setUpRank(o, device, rank, world, env.comm, pipeline.get());
  // agreement point: a rank whose set-up failed (scene, tiling, a hipMalloc in initialize, the environment map, the
  // checkpoint) says so HERE, and every rank leaves — none enters a barrier or a collective its peer will never reach
  if (!env.agree(setUp)) {
    if (setUp) std::fprintf(stderr, "bdpt_render: rank %u leaves: the set-up of another rank failed\n", rank);
    return res;
  }

  int warmup = o.warmup;
  if (warmup < 0 || warmup >= o.frames) warmup = 0;
  for (int f = 0; f < warmup; f++) pipeline->renderFrame();
  (void)hipDeviceSynchronize();
  if (!env.barrier()) return res;
  auto t0 = std::chrono::steady_clock::now();
  for (int f = warmup; f < o.frames; f++) pipeline->renderFrame();
  (void)hipDeviceSynchronize();
  if (!env.barrier()) return res;  // the job is done when its slowest rank is
  res.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  // tiled: a collective — every rank calls it, every rank gets the frame
  std::vector<float> img = env.injected("readback") ? std::vector<float>() : pipeline->readOutput();
  if (img.size() != (size_t)o.W * o.H * 4) {
    std::fprintf(stderr, "bdpt_render: rank %u: reading the output back failed\n", rank);
    env.failLate();  // the peers may be inside the all-gather this rank never entered (or left early): release them
    return res;
  }
  if (writer) {
    double mean = 0;
    for (size_t i = 0; i < img.size(); i += 4) mean += img[i] + img[i + 1] + img[i + 2];
    std::printf("%s %ux%u depth %d mat %d: %d frames in %.2f ms (%.2f ms/frame), mean radiance %.6f", o.scene.c_str(), o.W, o.H, o.depth, o.mat,
                o.frames - warmup, res.ms, res.ms / (o.frames - warmup), mean / (3.0 * (double)(img.size() / 4)));
    if (tiled) std::printf(", %u GPU%s (interleaved stripes of %u rows, RCCL reduce-scatter + all-gather)", world, world == 1 ? "" : "s",
                           bdpt_stripe_rows(o.H, world));
    std::printf("\n");
    writePfm(o.out.c_str(), img, o.W, o.H);
  }
  if (!o.checkpoint.empty() && !pipeline->saveCheckpoint(o.checkpoint)) {  // tiled: one file per rank (its rows)
    std::fprintf(stderr, "bdpt_render: cannot write %s\n", o.checkpoint.c_str());
    return res;
  }
  if (writer && !o.raw.empty()) {
    FILE* f = std::fopen(o.raw.c_str(), "wb");
    if (f) {
      std::fwrite(img.data(), 4, img.size(), f);
      std::fclose(f);
    }
  }
  res.rc = 0;
  return res;
}
}  // namespace

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; i++) {
    auto next = [&](const char* name) -> const char* {
      if (std::strcmp(argv[i], name) == 0 && i + 1 < argc) return argv[++i];
      return nullptr;
    };
    if (std::strcmp(argv[i], "--denoise") == 0) o.denoise = true;
    else if (std::strcmp(argv[i], "--denoise-regression") == 0) o.denoise = o.denoiseRegression = true;
    else if (const char* v = next("--scene")) o.scene = v;
    else if (const char* v = next("--width")) o.W = (uint32_t)std::atoi(v);
    else if (const char* v = next("--height")) o.H = (uint32_t)std::atoi(v);
    else if (const char* v = next("--frames")) o.frames = std::atoi(v);
    else if (const char* v = next("--depth")) o.depth = std::atoi(v);
    else if (const char* v = next("--mat")) o.mat = std::atoi(v);
    else if (const char* v = next("--device")) o.device = std::atoi(v);
    else if (const char* v = next("--accum-limit")) o.accumLimit = std::atoi(v);
    else if (const char* v = next("--out")) o.out = v;
    else if (const char* v = next("--raw")) o.raw = v;
    else if (const char* v = next("--checkpoint")) o.checkpoint = v;
    else if (const char* v = next("--resume")) o.resume = v;
    else if (const char* v = next("--env")) o.envFile = v;
    else if (const char* v = next("--inflight")) o.inflight = std::atoi(v);
    else if (const char* v = next("--warmup")) o.warmup = std::atoi(v);  // of --frames: rendered before the clock starts (first-use allocations)
    else if (const char* v = next("--gpus")) o.gpus = std::atoi(v);
    else if (const char* v = next("--rank")) o.rank = std::atoi(v);
    else if (const char* v = next("--world")) o.world = std::atoi(v);
    else if (const char* v = next("--id-file")) o.idFile = v;
    else if (const char* v = next("--job-id")) o.jobId = v;  // names THIS run of the job: the id file of another run is not accepted
    else {
      std::fprintf(stderr, "usage: bdpt_render [--scene cornell|atrium|FILE.fscene|FILE.obj] [--width W] [--height H] [--frames N] [--depth D] "
                           "[--mat 0|1] [--accum-limit N] [--denoise | --denoise-regression] [--out file.pfm] [--raw file.f32] "
                           "[--resume file.ckpt] [--checkpoint file.ckpt] [--env probe.hdr|image|Black] [--inflight N] [--warmup N] "
                           "[--gpus N | --rank R --world N --id-file F [--job-id J] [--device D]]\n");
      return 2;
    }
  }
  if (o.gpus < 0 || o.gpus > 64 || (o.gpus > 0 && o.world > 0)) {
    std::fprintf(stderr, "bdpt_render: --gpus N (ranks as threads) or --rank R --world N --id-file F (one process per rank), not both\n");
    return 2;
  }

  if (o.world > 0) {  // one process per GPU: this is rank --rank of --world
    if (o.rank < 0 || o.rank >= o.world || o.idFile.empty()) {
      std::fprintf(stderr, "bdpt_render: --world N needs --rank R in [0, N) and --id-file F\n");
      return 2;
    }
    if (hipSetDevice(o.device) != hipSuccess) {
      std::fprintf(stderr, "bdpt_render: no HIP device %d\n", o.device);
      return 1;
    }
    ncclUniqueId id;
    ncclComm_t comm = nullptr;
    if (!bdpt::exchangeUniqueIdThroughFile(o.idFile, (uint32_t)o.rank, &id, 120.0, bdpt::jobNonce(o.jobId)) ||
        ncclCommInitRank(&comm, o.world, id, o.rank) != ncclSuccess) {
      std::fprintf(stderr, "bdpt_render: rank %d could not join the communicator through %s\n", o.rank, o.idFile.c_str());
      return 1;
    }
    if (o.rank == 0) bdpt::retireUniqueIdFile(o.idFile);  // every rank has joined: the file has served (single use)
    RankEnv env;
    env.comm = comm;
    env.rank = (uint32_t)o.rank;
    env.world = (uint32_t)o.world;
    RankResult r = runRank(o, o.device, env);
    if (r.rc == 0) ncclCommDestroy(comm);  // (after a failure the communicator was aborted, or peers may be gone: no orderly teardown)
    return r.rc;
  }

  if (o.gpus > 0) {  // N ranks as threads of this process, GPUs device .. device + N - 1
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < o.device + o.gpus) {
      std::fprintf(stderr, "bdpt_render: --gpus %d from device %d, but %d HIP device(s) are visible\n", o.gpus, o.device, have);
      return 1;
    }
    std::vector<int> devs((size_t)o.gpus);
    for (int i = 0; i < o.gpus; i++) devs[(size_t)i] = o.device + i;
    std::vector<ncclComm_t> comms((size_t)o.gpus, nullptr);
    const ncclResult_t nr = ncclCommInitAll(comms.data(), o.gpus, devs.data());
    if (nr != ncclSuccess) {
      std::fprintf(stderr, "bdpt_render: ncclCommInitAll: %s\n", ncclGetErrorString(nr));
      return 1;
    }
    bdpt::RankGroup group(o.gpus);
    std::vector<RankResult> results((size_t)o.gpus);
    std::vector<std::thread> threads;
    auto envOf = [&](int r) {
      RankEnv env;
      env.group = &group;
      env.comm = comms[(size_t)r];
      env.rank = (uint32_t)r;
      env.world = (uint32_t)o.gpus;
      return env;
    };
    // the watcher: once a rank has called abort(), every communicator is aborted so that no peer stays inside a collective
    std::atomic<bool> ranksDone{false}, commsAborted{false};
    std::thread watcher([&] {
      while (!ranksDone.load()) {
        if (group.aborted()) {
          for (ncclComm_t c : comms) ncclCommAbort(c);
          commsAborted.store(true);
          return;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
    });
    for (int r = 1; r < o.gpus; r++) threads.emplace_back([&, r] { results[(size_t)r] = runRank(o, devs[(size_t)r], envOf(r)); });
    results[0] = runRank(o, devs[0], envOf(0));
    for (std::thread& t : threads) t.join();
    ranksDone.store(true);
    watcher.join();
    int rc = 0;
    for (int r = 0; r < o.gpus; r++) {
      if (results[(size_t)r].rc != 0) rc = 1;
      if (!commsAborted.load()) ncclCommDestroy(comms[(size_t)r]);
    }
    return rc;
  }

  return runRank(o, o.device, RankEnv{}).rc;
}
