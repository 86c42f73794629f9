// Passes.h — the three passes of the reference pipeline that sit on the hot path, with the
// reference's class names, factories and GUI-exposed state:
//   LightProbeGBufferPass   CommonPasses/LightProbeGBufferPass.{h,cpp}
//   BDPTPass                BidirectionalPathtracing/Passes/BDPTPass.{h,cpp}
//   SimpleAccumulationPass  CommonPasses/SimpleAccumulationPass.{h,cpp}
// plus RayLaunch, the thin launch wrapper they hold (SharedUtils/RayLaunch.h:85-135): here it owns
// one bdpt_ctx and forwards to the C ABI of include/bdpt.h instead of building a DXR state object.
#pragma once
#include "../../include/bdpt.h"
#include "RenderPass.h"
#include "Tiling.h"

namespace bdpt {

// One bdpt_ctx shared by the passes of a pipeline (they trace the same scene on the same device).
class RayLaunch {
 public:
  using SharedPtr = std::shared_ptr<RayLaunch>;
  // All passes created for one device share the same launcher (and so the same BVH).
  static SharedPtr create(RenderContext* ctx);
  ~RayLaunch();
  void setScene(Scene::SharedPtr pScene);           // RayLaunch::setScene -> bdpt_set_scene (BVH build)
  void setMaxRecursionDepth(uint32_t d) { mMaxDepth = d; }
  bool readyToRender() const { return mCtx && mSceneSet; }
  // (re)size the per-pixel path state; called by execute when the screen size changed
  bool ensureSize(uint32_t w, uint32_t h);
  bdpt_ctx* ctx() const { return mSlot == 0 ? mCtx : mMore[mSlot - 1]; }
  Scene::SharedPtr scene() const { return mpScene; }
  const char* lastError() const;
  // Frames in flight: one bdpt_ctx per frame slot (each owns its path state and traverses its own copy of the BVH)
  bool setSlotCount(uint32_t n);
  void setCurrentSlot(uint32_t s) { mSlot = s <= mMore.size() ? s : 0; }
  uint32_t currentSlot() const { return mSlot; }
  // Multi-GPU tiling (Tiling.h): every context of this launcher renders the stripes of rank
  // x->rank() of x->world() (bdpt_resize_stripes) and keeps its splat accumulators owner-major; the passes then run the
  // two-phase execute with the exchange in between.  Call before the first frame (RenderingPipeline::setTiling does).
  void setTiling(TileExchange::SharedPtr x);
  bool tiled() const { return mExchange != nullptr; }
  TileExchange::SharedPtr exchange() const { return mExchange; }
  const bdpt_tile_info& tileInfo() const { return mTileInfo; }
  uint32_t stripeRows() const { return mStripeRows; }
  uint64_t* tileSplat() const { return mTileSplat.empty() ? nullptr : mTileSplat[mSlot]; }  // the current slot's reduced chunk

 private:
  RayLaunch() = default;
  void freeTileSplat();
  TileExchange::SharedPtr mExchange;
  bdpt_tile_info mTileInfo{};
  uint32_t mStripeRows = 0;
  std::vector<uint64_t*> mTileSplat;  // per slot: chunkU64 words, receives this rank's chunk of the reduce-scatter
  bdpt_ctx* mCtx = nullptr;
  std::vector<bdpt_ctx*> mMore;  // slots 1..
  uint32_t mSlot = 0;
  int mDevice = 0;
  Scene::SharedPtr mpScene;
  bool mSceneSet = false;
  uint32_t mW = 0, mH = 0, mMaxDepth = 8, mSizedDepth = 0;
};

class LightProbeGBufferPass : public RenderPass {
 public:
  using SharedPtr = std::shared_ptr<LightProbeGBufferPass>;
  static SharedPtr create() { return SharedPtr(new LightProbeGBufferPass()); }

 protected:
  LightProbeGBufferPass() : RenderPass("G-Buf With Light Probe", "G-Buffer With Light Probe Options") {}
  bool initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) override;
  void execute(RenderContext* pRenderContext) override;
  void renderGui(Gui* pGui) override;
  void initScene(RenderContext* pRenderContext, Scene::SharedPtr pScene) override;
  bool requiresScene() override { return true; }
  bool usesRayTracing() override { return true; }
  bool usesEnvironmentMap() override { return true; }

  RayLaunch::SharedPtr mpRays;
  Scene::SharedPtr mpScene;
  bool mUseThinLens = false;
  float mFStop = 32.0f, mFocalLength = 1.0f, mLensRadius = 0.0f;
  bool mUseJitter = true, mUseRandomJitter = false;  // random jitter (std::mt19937 seeded by time) is not reproduced
  void saveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) override;
  bool loadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) override;
  uint32_t mFrameCount = 0xdeadbeefu;
};

class BDPTPass : public RenderPass {
 public:
  using SharedPtr = std::shared_ptr<BDPTPass>;
  static SharedPtr create(const std::string& outChannel) { return SharedPtr(new BDPTPass(outChannel)); }
  // build-only switches (BDPT_PARAM_*), default 0 = reference behaviour
  void setParamFlags(uint32_t f) { mParamFlags = f; }

 protected:
  BDPTPass(const std::string& outChannel) : RenderPass("Bidirectional Pathtracer", "BDPT Options"), mOutputTextureName(outChannel) {}
  bool initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) override;
  void initScene(RenderContext* pRenderContext, Scene::SharedPtr pScene) override;
  void execute(RenderContext* pRenderContext) override;
  void renderGui(Gui* pGui) override;
  // The frame's path state is sized as soon as the frame size is known — before initScene builds the acceleration
  // structure, whose freed scratch the driver would otherwise have to clear again under these allocations
  void resize(uint32_t width, uint32_t height) override;
  bool requiresScene() override { return true; }
  bool usesRayTracing() override { return true; }

  RayLaunch::SharedPtr mpRays;
  Scene::SharedPtr mpScene;
  int32_t mUserSpecifiedRayDepth = 3;
  const int32_t mMaxPossibleRayDepth = 8;
  int32_t mMaterialIndex = 0;
  const int32_t mNumOfMaterials = 2;
  float mClampUpper = 0.9f;
  float mRefractiveIndex = 1.0f;
  std::string mOutputTextureName;
  void saveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) override;
  bool loadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) override;
  uint32_t mFrameCount = 0x1337u;
  uint32_t mParamFlags = 0;
};

class SimpleAccumulationPass : public RenderPass {
 public:
  using SharedPtr = std::shared_ptr<SimpleAccumulationPass>;
  static SharedPtr create(const std::string& bufferToAccumulate) { return SharedPtr(new SimpleAccumulationPass(bufferToAccumulate)); }
  uint32_t getAccumCount() const { return mAccumCount; }

 protected:
  SimpleAccumulationPass(const std::string& bufferToAccumulate) : RenderPass("Accumulation Pass", "Accumulation Options") {
    mAccumChannel = bufferToAccumulate;
  }
  bool initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) override;
  void initScene(RenderContext* pRenderContext, Scene::SharedPtr pScene) override;
  void execute(RenderContext* pRenderContext) override;
  void renderGui(Gui* pGui) override;
  void resize(uint32_t width, uint32_t height) override;
  void stateRefreshed() override;
  bool appliesPostprocess() override { return true; }
  bool hasAnimation() override { return false; }
  bool needsFrameOrder() override { return true; }  // the running mean is applied in frame order
  bool hasCameraMoved();
  void saveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) override;
  bool loadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) override;

  std::string mAccumChannel;
  RayLaunch::SharedPtr mpRays;  // only for the context that runs bdpt_accumulate
  Texture::SharedPtr mpLastFrame;
  Scene::SharedPtr mpScene;
  uint64_t mLastCameraVersion = 0;
  bool mDoAccumulation = true;
  bool mResumed = false;  // state came from a checkpoint: the next frame's camera set-up is not a camera move
  uint32_t mAccumCount = 0;
  int32_t mCountLimit = 100;
  const int32_t mMaxCountLimit = 10000;
};

// BMFR denoiser pass (BidirectionalPathtracing/Passes/DenoisePass.{h,cpp}): same channels, switches and defaults
// (off until "Do BMFR Denoise" is ticked; regression off, pre/post-process on, DenoisePass.h:70-75).  The three
// shaders and their history textures live behind bdpt_bmfr_execute.
class BlockwiseMultiOrderFeatureRegression : public RenderPass {
 public:
  using SharedPtr = std::shared_ptr<BlockwiseMultiOrderFeatureRegression>;
  static SharedPtr create(const std::string& bufferToDenoise = ResourceManager::kOutputChannel) {
    return SharedPtr(new BlockwiseMultiOrderFeatureRegression(bufferToDenoise));
  }
  uint32_t getAccumCount() const { return mAccumCount; }
  ~BlockwiseMultiOrderFeatureRegression() override { freeGather(); }

 protected:
  BlockwiseMultiOrderFeatureRegression(const std::string& bufferToDenoise) : RenderPass("BMFR Denoise Pass", "BMFR Denoise Options") {
    mDenoiseChannel = bufferToDenoise;
  }
  bool initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) override;
  void initScene(RenderContext* pRenderContext, Scene::SharedPtr pScene) override;
  void execute(RenderContext* pRenderContext) override;
  void renderGui(Gui* pGui) override;
  void resize(uint32_t width, uint32_t height) override;
  bool appliesPostprocess() override { return true; }
  bool hasAnimation() override { return false; }
  // The temporal history (previous position / normal / noisy / filtered frames) lives inside the launcher context: frames
  // run one by one while the denoiser is on, and a checkpoint carries the history (bdpt_bmfr_save_history) beside the
  // frame number.
  bool holdsTemporalState() override { return mDoDenoise; }
  void saveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) override;
  bool loadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) override;
  // Tiled (RenderingPipeline::setTiling): the filter's blocks need their neighbours, so every rank gathers the WHOLE
  // noisy frame and the three feature channels (its rows packed, one all-gather, the rows put back) and filters the
  // same whole frame; the result replaces the rank's output channel, all rows of it.
  bool gatherWholeFrame(RenderContext* pRenderContext, Texture::SharedPtr noisy, Texture::SharedPtr pos, Texture::SharedPtr nrm, Texture::SharedPtr alb);
  void freeGather();

  std::string mDenoiseChannel;
  RayLaunch::SharedPtr mpRays;
  Scene::SharedPtr mpScene;
  bool mDoDenoise = false;
  bool mBMFR_preprocess = true;
  bool mBMFR_postprocess = true;
  bool mBMFR_regression = false;
  bool mBMFR_removeFeatures = true;
  uint32_t mAccumCount = 0;
  // tiled: this rank's packed rows of the four channels, all ranks' (after the all-gather), and the whole-frame copies
  uint8_t *mPackedMine = nullptr, *mPackedAll = nullptr;
  float *mFullNoisy = nullptr, *mFullPos = nullptr;
  uint16_t *mFullNorm = nullptr, *mFullAlb = nullptr;
  uint32_t mGatherW = 0, mGatherH = 0;
};

// Headless counterpart of SharedUtils/RenderingPipeline (setPass + per-frame loop,
// RenderingPipeline.cpp:421-471, 611-695); the window, GUI rendering and final blit are out of scope.
class RenderingPipeline {
 public:
  RenderingPipeline();  // as Main.cpp:12 constructs it; the size comes with run()'s SampleConfig or with setSize()
  RenderingPipeline(uint32_t width, uint32_t height, int device = 0);
  ~RenderingPipeline();
  void setSize(uint32_t width, uint32_t height, int device = 0);  // before initialize()
  // Offline accumulation runs (a 1024-sample image is a long job; the reference's loop is interactive,
  // RenderingPipeline.cpp:611-695): keep n frames in flight, each on its own stream with its own channels and
  // launcher context, the ordered passes (accumulation) chained by events.  Same image as n = 1, bit for bit.
  // Call before initialize(); ignored (n = 1) while a pass holds temporal state of its own (the denoiser switched on).
  void setFramesInFlight(uint32_t n) { mFramesInFlight = n < 1 ? 1 : (n > 8 ? 8 : n); }
  uint32_t getFramesInFlight() const { return mFramesInFlight; }
  // Multi-GPU: this pipeline is rank `rank` of `world` pipelines (one per GPU, in threads of one process or in
  // processes of their own) that render ONE frame together — interleaved stripes of rows, scene replicated, splat
  // accumulators summed with one ncclReduceScatter per frame, the frame assembled by ncclAllGather when it is read
  // back (readOutput).  `comm` is this rank's RCCL communicator, created and destroyed by the host program; nullptr is
  // allowed for world == 1 (the tiled code path without a collective).  Call before initialize().  The passes and
  // their per-frame order (SharedUtils/RenderingPipeline.cpp:611-695) are unchanged; what is replaced is the single
  // DispatchRays of the reference (Falcor API/D3D12/D3D12RenderContext.cpp:350-384).  Same image as one GPU, bit for bit.
  bool setTiling(uint32_t rank, uint32_t world, ncclComm_t comm);
  // what TileExchange::abort calls instead of ncclCommAbort when this rank has to leave a collective (before initialize())
  void setTilingAbortHandler(std::function<void()> f) { mTileAbort = std::move(f); }
  bool isTiled() const { return mTileWorld > 0; }
  // RenderingPipeline::run (RenderingPipeline.cpp:697-712) without a window: size the channels from the config,
  // load the scene named by BDPT_SCENE (a .fscene / .obj path, "atrium", default the Cornell box), render BDPT_FRAMES
  // frames (default 1), and delete the pipeline.
  static void run(RenderingPipeline* pipe, SampleConfig& config);
  void setPass(uint32_t passNum, RenderPass::SharedPtr pTargetPass);
  // onLoad (initialize every pass, drop those that fail), onFirstRun (scene), onResize
  bool initialize(Scene::SharedPtr pScene);
  void renderFrame();                 // onFrameRender: refresh notifications, then every pass in order
  void applyGui(Gui* pGui);           // renderGui of every pass (scripted widget values)
  ResourceManager::SharedPtr getResourceManager() { return mpResourceManager; }
  RenderContext* getRenderContext() { return &mContext; }
  std::vector<float> readOutput();    // "PipelineOutput" as RGBA32F
  // every pass's cross-frame state (frame counters, accumulated frame) to / from a file: a run resumed from a
  // checkpoint continues the frame sequence bit for bit.  false on I/O errors, when a pass holds cross-frame state it
  // cannot serialise (the BMFR denoiser when switched on), or for a file that does not match this pipeline: other
  // passes, frame size, scene, or pass settings (ray depth, material model, clamp, lens, accumulation limit).  After a
  // failed load the pipeline's state is unspecified: discard it.
  bool saveCheckpoint(const std::string& path);
  bool loadCheckpoint(const std::string& path);
  size_t getPassCount() const { return mActivePasses.size(); }

 private:
  uint64_t sceneIdentity() const;
  RenderContext mContext;
  uint32_t mWidth, mHeight;
  ResourceManager::SharedPtr mpResourceManager;
  std::vector<RenderPass::SharedPtr> mActivePasses;
  Scene::SharedPtr mpScene;
  uint32_t mFramesInFlight = 1;
  uint64_t mFrameIndex = 0;
  std::vector<hipStream_t> mSlotStreams;   // one per frame slot (slot 0 included)
  std::vector<hipEvent_t> mOrderEvents;    // per ordered pass: recorded after it ran for the latest frame
  RayLaunch::SharedPtr mpRays;             // kept to switch the launcher's slot
  bool inFlightActive();
  uint32_t mTileRank = 0, mTileWorld = 0;  // world 0: not tiled
  std::function<void()> mTileAbort;
  ncclComm_t mTileComm = nullptr;
  std::string rankPath(const std::string& path) const;  // checkpoints of a tiled pipeline are per rank
};

}  // namespace bdpt
