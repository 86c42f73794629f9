// RenderPass.h / ResourceManager — the pass-framework surface the BDPT pass plugs into.
//
// Mirrors, name for name, the global ::RenderPass of the reference (SharedUtils/RenderPass.h:25-220,
// RenderPass.cpp:26-62) and its ResourceManager (SharedUtils/ResourceManager.h:26-173,
// ResourceManager.cpp:22-241): protected virtual interface, public on* trampolines, capability
// queries, refresh / rebind flags; string-keyed channels where the first requester fixes the format
// and a conflicting request returns -1.  Falcor types are replaced by HostTypes.h.
#pragma once
#include "HostTypes.h"

namespace bdpt {

class ResourceManager : public std::enable_shared_from_this<ResourceManager> {
 public:
  using SharedPtr = std::shared_ptr<ResourceManager>;
  static const std::string kOutputChannel;   // "PipelineOutput"
  static const std::string kEnvironmentMap;  // "EnvironmentMap"
  static const BindFlags kDefaultFlags;

  static SharedPtr create(uint32_t width, uint32_t height, RenderContext* ctx) { return SharedPtr(new ResourceManager(width, height, ctx)); }

  int32_t requestTextureResource(const std::string& channelName, ResourceFormat channelFormat = ResourceFormat::RGBA32Float,
                                 BindFlags usageFlags = kDefaultFlags, int32_t channelWidth = -1, int32_t channelHeight = -1);
  void requestTextureResources(const std::vector<std::string>& channelNames, ResourceFormat channelFormat = ResourceFormat::RGBA32Float,
                               BindFlags usageFlags = kDefaultFlags, int32_t channelWidth = -1, int32_t channelHeight = -1);
  int32_t manageTextureResource(const std::string& channelName, Texture::SharedPtr sharedTex);

  Texture::SharedPtr getTexture(const std::string& channelName);
  Texture::SharedPtr getTexture(int32_t channelIdx);
  Texture::SharedPtr getClearedTexture(const std::string& channelName, const vec4& clearColor);
  Texture::SharedPtr getClearedTexture(int32_t channelIdx, const vec4& clearColor);
  void clearTexture(Texture::SharedPtr& tex, const vec4& clearColor);
  std::string getTextureName(int32_t channelIdx);
  int32_t getTextureIndex(const std::string& channelName) const;
  uint32_t getTextureCount() const { return uint32_t(mTextures.size()); }

  bool updateEnvironmentMap(const std::string& filename);  // "" = default (0.5,0.5,0.8), "Black" = black, else an image file (.hdr, .png, .jpg ...)
  Texture::SharedPtr getEnvironmentMap() { return getTexture(kEnvironmentMap); }
  uvec2 getEnvironmentMapSize() const;

  std::string getDefaultSceneName() { return mDefaultSceneName; }
  void setDefaultSceneName(const std::string& sceneFilename) { mDefaultSceneName = sceneFilename; mUserSetDefaultScene = true; }
  bool userSetDefaultScene() const { return mUserSetDefaultScene; }

  void resize(uint32_t width, uint32_t height);
  void initializeResources();
  bool isInitialized() const { return mIsInitialized; }
  uint32_t getWidth() const { return mWidth; }
  uint32_t getHeight() const { return mHeight; }
  uvec2 getScreenSize() const { return uvec2{mWidth, mHeight}; }
  bool haveResourcesChanged() const { return mUpdatedFlag; }
  void resetDirtyFlag() { mUpdatedFlag = false; }
  float getMinTDist() const { return mMinT; }
  void setMinTDist(float newMinT) { mMinT = newMinT; }
  RenderContext* getRenderContext() const { return mpContext; }
  // Frames in flight (RenderingPipeline::setFramesInFlight; not in the reference, whose loop is interactive): every
  // requested channel exists once per frame slot, and getTexture returns the current slot's.  Managed textures
  // (manageTextureResource: the environment map) are shared by all slots.
  void setSlotCount(uint32_t n);
  void setCurrentSlot(uint32_t s) { mSlot = s <= mSlotTextures.size() ? s : 0; }
  uint32_t getSlotCount() const { return (uint32_t)mSlotTextures.size() + 1u; }

 protected:
  ResourceManager(uint32_t width, uint32_t height, RenderContext* ctx) : mWidth(width), mHeight(height), mpContext(ctx) {}
  std::vector<Texture::SharedPtr>& slotTextures() { return mSlot == 0 ? mTextures : mSlotTextures[mSlot - 1]; }
  std::vector<std::vector<Texture::SharedPtr>> mSlotTextures;  // slots 1.. (slot 0 is mTextures)
  std::vector<bool> mManaged;                                   // per channel: shared by all slots
  uint32_t mSlot = 0;
  uint32_t mWidth = 0, mHeight = 0;
  bool mIsInitialized = false, mUpdatedFlag = true;
  float mMinT = 1.0e-4f;
  std::string mEnvMapFilename;
  std::string mDefaultSceneName = "Media/Arcade/Arcade.fscene";
  bool mUserSetDefaultScene = false;
  RenderContext* mpContext;
  std::vector<Texture::SharedPtr> mTextures;
  std::vector<std::string> mTextureNames;
  std::vector<ivec2> mTextureSizes;
  std::vector<BindFlags> mTextureFlags;
  std::vector<ResourceFormat> mTextureFormat;
};

class RenderPass : public std::enable_shared_from_this<RenderPass> {
 public:
  using SharedPtr = std::shared_ptr<RenderPass>;
  virtual ~RenderPass() = default;

 protected:
  virtual bool initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) = 0;
  virtual void initScene(RenderContext* pRenderContext, Scene::SharedPtr pScene) {}
  virtual void resize(uint32_t width, uint32_t height) {}
  virtual void pipelineUpdated(ResourceManager::SharedPtr pResManager) { mpResManager = pResManager; }
  virtual bool processKeyEvent(const KeyboardEvent& keyEvent) { return false; }
  virtual bool processMouseEvent(const MouseEvent& mouseEvent) { return false; }
  virtual void renderGui(Gui* pGui) {}
  virtual void execute(RenderContext* pRenderContext) = 0;
  virtual void shutdown() {}
  virtual void stateRefreshed() {}
  virtual void activatePass() {}
  virtual void deactivatePass() {}
  // Checkpoint / resume of a long accumulation (not in the reference, whose runs are interactive: SURVEY.md section 5):
  // what a pass must carry over for the frame sequence to continue bit for bit — frame counters, accumulated
  // texture — as raw bytes.  A pass without cross-frame state keeps the defaults.
  virtual void saveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) {}
  virtual bool loadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) { return size == 0; }
  // true when the pass carries state from frame to frame that saveState does NOT capture: a pipeline holding such a
  // pass refuses to write or read a checkpoint instead of resuming into a silently different sequence
  virtual bool hasUnsavedCrossFrameState() { return false; }
  // true when a frame of this pass reads what its previous frame left behind (the denoiser's history): the pipeline then
  // runs the frames one by one instead of keeping several in flight
  virtual bool holdsTemporalState() { return false; }
  // true when the pass must see the frames in their order (the running mean): with frames in flight the pipeline
  // makes this pass of frame f + 1 wait for the same pass of frame f; every other pass only depends on its own frame
  virtual bool needsFrameOrder() { return false; }

 public:
  virtual bool requiresScene() { return false; }
  virtual bool loadDefaultScene() { return false; }
  virtual bool usesRasterization() { return false; }
  virtual bool usesRayTracing() { return false; }
  virtual bool usesCompute() { return false; }
  virtual bool appliesPostprocess() { return false; }
  virtual bool usesEnvironmentMap() { return false; }
  virtual bool hasAnimation() { return true; }

  bool onInitialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager);
  void onInitScene(RenderContext* pRenderContext, Scene::SharedPtr pScene) { initScene(pRenderContext, pScene); }
  void onPipelineUpdate(ResourceManager::SharedPtr pResManager) { pipelineUpdated(pResManager); }
  void onStateRefresh() { stateRefreshed(); }
  void onResize(uint32_t width, uint32_t height) { resize(width, height); }
  bool onKeyEvent(const KeyboardEvent& keyEvent) { return processKeyEvent(keyEvent); }
  bool onMouseEvent(const MouseEvent& mouseEvent) { return processMouseEvent(mouseEvent); }
  void onRenderGui(Gui* pGui) { renderGui(pGui); }
  void onExecute(RenderContext* pRenderContext);
  void onShutdown();
  void onPassActivation() { activatePass(); }
  void onPassDeactivation() { deactivatePass(); }
  void onSaveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) { saveState(pRenderContext, out); }
  bool onLoadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) { return loadState(pRenderContext, data, size); }
  bool onHasUnsavedCrossFrameState() { return hasUnsavedCrossFrameState(); }
  bool onHoldsTemporalState() { return holdsTemporalState(); }
  bool onNeedsFrameOrder() { return needsFrameOrder(); }

  void setName(const std::string& name) { mName = name; }
  std::string getName() const { return mName; }
  void setGuiName(const std::string& guiName) { mGuiName = guiName; }
  std::string getGuiName() const { return mGuiName; }
  bool useGuiWindow() const { return true; }
  void setGuiPosition(const ivec2& p) { mGuiPosition = p; }
  ivec2 getGuiPosition() const { return mGuiPosition; }
  void setGuiSize(const ivec2& s) { mGuiSize = s; }
  ivec2 getGuiSize() const { return mGuiSize; }
  bool isInitialized() const { return mIsInitialized; }
  bool isRefreshFlagSet() const { return mRefreshFlag; }
  void resetRefreshFlag() { mRefreshFlag = false; }
  bool isRebindFlagSet() const { return mRebindFlag; }
  void resetRebindFlag() { mRebindFlag = false; }

 protected:
  RenderPass(const std::string name = "<Unknown render pass>", const std::string guiName = "<Unknown gui group>")
      : mName(name), mGuiName(guiName) {}
  RenderPass(const RenderPass&) = delete;
  RenderPass& operator=(const RenderPass&) = delete;
  void setRefreshFlag() { mRefreshFlag = true; }
  void setRebindFlag() { mRebindFlag = true; }

 private:
  std::string mName, mGuiName;
  ivec2 mGuiPosition{-270, 30}, mGuiSize{250, 160};
  bool mIsInitialized = false, mRefreshFlag = true, mRebindFlag = true;

 protected:
  ResourceManager::SharedPtr mpResManager;
};

}  // namespace bdpt
