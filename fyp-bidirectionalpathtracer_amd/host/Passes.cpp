// Passes.cpp — host mirror of the reference's pass framework and passes over the C ABI.
#include "Passes.h"
#include "RankSync.h"

#include "../../include/bdpt_scene.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace bdpt {

// ---- checkpoint byte helpers (RenderingPipeline::saveCheckpoint / the passes' saveState)
namespace {
constexpr uint32_t kCheckpointMagic = 0x42445054u;  // "BDPT"
constexpr uint32_t kCheckpointVersion = 2u;
void put32(std::vector<uint8_t>& o, uint32_t v) {
  for (int i = 0; i < 4; i++) o.push_back((uint8_t)(v >> (8 * i)));
}
void put64(std::vector<uint8_t>& o, uint64_t v) {
  for (int i = 0; i < 8; i++) o.push_back((uint8_t)(v >> (8 * i)));
}
uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t get64(const uint8_t* p) { return (uint64_t)get32(p) | ((uint64_t)get32(p + 4) << 32); }
}  // namespace

// ------------------------------------------------------------------------------------------------
// HostTypes
// ------------------------------------------------------------------------------------------------
static uint16_t floatToHalf(float f) {  // round to nearest even, as the device does for the G-buffer
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (ax < 0x33000001u) return (uint16_t)sign;
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t he = (e < -14) ? 0u : (uint32_t)(e + 15);
  uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  return (uint16_t)(sign | ((he == 0) ? q : (((he - 1) << 10) + q)));
}
static float halfToFloat(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
  if (e == 0) {
    float v = (float)m * 5.9604644775390625e-08f;
    std::memcpy(&x, &v, 4);
    x |= sign;
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

Texture::SharedPtr Texture::create2D(uint32_t w, uint32_t h, ResourceFormat fmt) {
  SharedPtr t(new Texture());
  t->mW = w;
  t->mH = h;
  t->mFormat = fmt;
  if (hipMalloc(&t->mData, std::max<size_t>(t->getSizeInBytes(), 16)) != hipSuccess) return nullptr;
  (void)hipMemset(t->mData, 0, t->getSizeInBytes());
  return t;
}
Texture::~Texture() {
  if (mData) (void)hipFree(mData);
}
void Texture::clear(const vec4& c, hipStream_t stream) {
  if (c.x == 0 && c.y == 0 && c.z == 0 && c.w == 0) {
    (void)hipMemsetAsync(mData, 0, getSizeInBytes(), stream);
    return;
  }
  const size_t n = (size_t)mW * mH;
  if (mFormat == ResourceFormat::RGBA32Float) {
    std::vector<float> host(n * 4);
    for (size_t i = 0; i < n; i++) {
      host[i * 4] = c.x;
      host[i * 4 + 1] = c.y;
      host[i * 4 + 2] = c.z;
      host[i * 4 + 3] = c.w;
    }
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(mData, host.data(), host.size() * 4, hipMemcpyHostToDevice);
  } else {
    std::vector<uint16_t> host(n * 4);
    const uint16_t q[4] = {floatToHalf(c.x), floatToHalf(c.y), floatToHalf(c.z), floatToHalf(c.w)};
    for (size_t i = 0; i < n * 4; i++) host[i] = q[i & 3];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(mData, host.data(), host.size() * 2, hipMemcpyHostToDevice);
  }
}
void Texture::downloadRaw(hipStream_t stream, std::vector<uint8_t>& out) const {
  (void)hipStreamSynchronize(stream);
  out.resize(getSizeInBytes());
  (void)hipMemcpy(out.data(), mData, out.size(), hipMemcpyDeviceToHost);
}
bool Texture::uploadRaw(hipStream_t stream, const uint8_t* data, size_t size) {
  if (size != getSizeInBytes()) return false;
  (void)hipStreamSynchronize(stream);
  return hipMemcpy(mData, data, size, hipMemcpyHostToDevice) == hipSuccess;
}
std::vector<float> Texture::download(hipStream_t stream) const {
  (void)hipStreamSynchronize(stream);
  const size_t n = (size_t)mW * mH * 4;
  std::vector<float> out(n);
  if (mFormat == ResourceFormat::RGBA32Float) {
    (void)hipMemcpy(out.data(), mData, n * 4, hipMemcpyDeviceToHost);
  } else {
    std::vector<uint16_t> raw(n);
    (void)hipMemcpy(raw.data(), mData, n * 2, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < n; i++) out[i] = halfToFloat(raw[i]);
  }
  return out;
}

bool Gui::addIntVar(const char* name, int32_t& v, int lo, int hi) {
  log.push_back(std::string("int:") + name);
  auto it = overrides.find(name);
  if (it == overrides.end()) return false;
  int32_t nv = (int32_t)it->second;
  nv = nv < lo ? lo : (nv > hi ? hi : nv);
  bool changed = nv != v;
  v = nv;
  return changed;
}
bool Gui::addFloatVar(const char* name, float& v, float lo, float hi, float, bool) {
  log.push_back(std::string("float:") + name);
  auto it = overrides.find(name);
  if (it == overrides.end()) return false;
  float nv = (float)it->second;
  nv = nv < lo ? lo : (nv > hi ? hi : nv);
  bool changed = nv != v;
  v = nv;
  return changed;
}
bool Gui::addCheckBox(const char* name, bool& v, bool) {
  log.push_back(std::string("check:") + name);
  auto it = overrides.find(name);
  if (it == overrides.end()) return false;
  bool nv = it->second != 0.0;
  bool changed = nv != v;
  v = nv;
  return changed;
}

// ------------------------------------------------------------------------------------------------
// ResourceManager (SharedUtils/ResourceManager.cpp:22-241)
// ------------------------------------------------------------------------------------------------
const std::string ResourceManager::kOutputChannel = "PipelineOutput";
const std::string ResourceManager::kEnvironmentMap = "EnvironmentMap";
const BindFlags ResourceManager::kDefaultFlags = BindFlags::ShaderResource | BindFlags::UnorderedAccess | BindFlags::RenderTarget;

int32_t ResourceManager::getTextureIndex(const std::string& channelName) const {
  for (size_t i = 0; i < mTextureNames.size(); i++)
    if (mTextureNames[i] == channelName) return (int32_t)i;
  return -1;
}
std::string ResourceManager::getTextureName(int32_t i) { return (i < 0 || i >= (int32_t)mTextureNames.size()) ? "" : mTextureNames[(size_t)i]; }

int32_t ResourceManager::requestTextureResource(const std::string& channelName, ResourceFormat channelFormat, BindFlags usageFlags,
                                                int32_t channelWidth, int32_t channelHeight) {
  int32_t existingIndex = getTextureIndex(channelName);
  if (existingIndex >= 0) {
    // conflicting needs: the first requester's format / size stands (ResourceManager.cpp:218-221)
    if (channelFormat != mTextureFormat[(size_t)existingIndex]) return -1;
    if (mTextureSizes[(size_t)existingIndex] != ivec2{channelWidth, channelHeight}) return -1;
    mTextureFlags[(size_t)existingIndex] |= usageFlags;
    return existingIndex;
  }
  existingIndex = int32_t(mTextures.size());
  mTextures.push_back(nullptr);  // created in initializeResources()
  mManaged.push_back(false);
  for (auto& v : mSlotTextures) v.push_back(nullptr);
  mTextureSizes.push_back(ivec2{channelWidth, channelHeight});
  mTextureNames.push_back(channelName);
  mTextureFlags.push_back(usageFlags);
  mTextureFormat.push_back(channelFormat);
  mUpdatedFlag = true;
  return existingIndex;
}
void ResourceManager::requestTextureResources(const std::vector<std::string>& names, ResourceFormat f, BindFlags fl, int32_t w, int32_t h) {
  for (const std::string& n : names) requestTextureResource(n, f, fl, w, h);
}
int32_t ResourceManager::manageTextureResource(const std::string& channelName, Texture::SharedPtr tex) {
  if (!tex) return -1;
  int32_t idx = getTextureIndex(channelName);
  if (idx < 0) {
    idx = int32_t(mTextures.size());
    mTextures.push_back(tex);
    mManaged.push_back(true);
    for (auto& v : mSlotTextures) v.push_back(nullptr);
    mTextureNames.push_back(channelName);
    mTextureSizes.push_back(ivec2{(int)tex->getWidth(), (int)tex->getHeight()});
    mTextureFlags.push_back(kDefaultFlags);
    mTextureFormat.push_back(tex->getFormat());
  } else {
    mTextures[(size_t)idx] = tex;
    if ((size_t)idx < mManaged.size()) mManaged[(size_t)idx] = true;
    mTextureSizes[(size_t)idx] = ivec2{(int)tex->getWidth(), (int)tex->getHeight()};
    mTextureFormat[(size_t)idx] = tex->getFormat();
  }
  mUpdatedFlag = true;
  return idx;
}
Texture::SharedPtr ResourceManager::getTexture(int32_t i) {
  if (i < 0 || i >= (int32_t)mTextures.size()) return nullptr;
  if (mSlot > 0 && !(i < (int32_t)mManaged.size() && mManaged[(size_t)i])) return mSlotTextures[mSlot - 1][(size_t)i];
  return mTextures[(size_t)i];
}
void ResourceManager::setSlotCount(uint32_t n) {
  mSlotTextures.assign(n > 1 ? n - 1 : 0, std::vector<Texture::SharedPtr>());
  for (auto& v : mSlotTextures) v.assign(mTextures.size(), nullptr);
  mSlot = 0;
  if (mIsInitialized) initializeResources();
}
Texture::SharedPtr ResourceManager::getTexture(const std::string& n) { return getTexture(getTextureIndex(n)); }
Texture::SharedPtr ResourceManager::getClearedTexture(const std::string& n, const vec4& c) { return getClearedTexture(getTextureIndex(n), c); }
Texture::SharedPtr ResourceManager::getClearedTexture(int32_t i, const vec4& c) {
  Texture::SharedPtr t = getTexture(i);
  if (!t) return nullptr;
  t->clear(c, mpContext->getStream());
  return t;
}
void ResourceManager::clearTexture(Texture::SharedPtr& tex, const vec4& c) {
  if (tex) tex->clear(c, mpContext->getStream());
}
bool ResourceManager::updateEnvironmentMap(const std::string& filename) {
  if (filename != "" && filename != "Black") {
    // createTextureFromFile(filename, false, false): a Radiance .hdr probe as it is, any other supported image as
    // 8-bit values / 255 without sRGB decoding (loadAsSrgb = false), both as RGBA32F
    uint32_t w = 0, h = 0;
    std::vector<float> px;
    char msg[256] = {0};
    const bool isHdr = filename.size() > 4 && (filename.substr(filename.size() - 4) == ".hdr" || filename.substr(filename.size() - 4) == ".HDR");
    if (isHdr) {
      if (bdpt_image_load_hdr(filename.c_str(), &w, &h, nullptr, 0, msg, sizeof(msg)) != BDPT_OK) return false;
      px.resize((size_t)w * h * 4);
      if (bdpt_image_load_hdr(filename.c_str(), &w, &h, px.data(), px.size(), msg, sizeof(msg)) != BDPT_OK) return false;
    } else {
      uint32_t alpha = 0;
      if (bdpt_image_load(filename.c_str(), &w, &h, &alpha, nullptr, 0, msg, sizeof(msg)) != BDPT_OK) return false;
      std::vector<uint8_t> b((size_t)w * h * 4);
      if (bdpt_image_load(filename.c_str(), &w, &h, &alpha, b.data(), b.size(), msg, sizeof(msg)) != BDPT_OK) return false;
      px.resize(b.size());
      for (size_t i = 0; i < b.size(); i++) px[i] = (float)b[i] / 255.0f;
    }
    Texture::SharedPtr envMap = Texture::create2D(w, h, ResourceFormat::RGBA32Float);
    if (!envMap || !envMap->uploadRaw(mpContext->getStream(), reinterpret_cast<const uint8_t*>(px.data()), px.size() * sizeof(float))) return false;
    const size_t found = filename.find_last_of("/\\");
    mEnvMapFilename = found == std::string::npos ? filename : filename.substr(found + 1);
    manageTextureResource(kEnvironmentMap, envMap);
    return true;
  }
  Texture::SharedPtr tmpEnv = Texture::create2D(128, 128, ResourceFormat::RGBA32Float);
  if (!tmpEnv) return false;
  tmpEnv->clear(filename == "" ? vec4{0.5f, 0.5f, 0.8f, 1.0f} : vec4{0.0f, 0.0f, 0.0f, 1.0f}, mpContext->getStream());
  // the map is read by frames on other (non-blocking) streams: the fill must have landed before anyone is told about it
  (void)hipStreamSynchronize(mpContext->getStream());
  manageTextureResource(kEnvironmentMap, tmpEnv);
  mEnvMapFilename = filename;
  return true;
}
uvec2 ResourceManager::getEnvironmentMapSize() const {
  int32_t i = getTextureIndex(kEnvironmentMap);
  if (i < 0) return uvec2{0, 0};
  return uvec2{(uint32_t)mTextureSizes[(size_t)i].x, (uint32_t)mTextureSizes[(size_t)i].y};
}
void ResourceManager::initializeResources() {
  for (size_t i = 0; i < mTextures.size(); i++) {
    uint32_t w = mTextureSizes[i].x <= 0 ? mWidth : (uint32_t)mTextureSizes[i].x;
    uint32_t h = mTextureSizes[i].y <= 0 ? mHeight : (uint32_t)mTextureSizes[i].y;
    if (!mTextures[i]) mTextures[i] = Texture::create2D(w, h, mTextureFormat[i]);
    if (!(i < mManaged.size() && mManaged[i]))
      for (auto& v : mSlotTextures) {
        if (v.size() < mTextures.size()) v.resize(mTextures.size());
        if (!v[i]) v[i] = Texture::create2D(w, h, mTextureFormat[i]);
      }
  }
  mIsInitialized = true;
  mUpdatedFlag = true;
}
void ResourceManager::resize(uint32_t width, uint32_t height) {
  if (width == mWidth && height == mHeight && mIsInitialized) return;
  mWidth = width;
  mHeight = height;
  if (mWidth == 0 || mHeight == 0) return;
  if (!mIsInitialized) initializeResources();
  for (size_t i = 0; i < mTextures.size(); i++) {
    if (mTextureSizes[i] != ivec2{-1, -1}) continue;
    if (mTextures[i] && mTextures[i]->getWidth() == mWidth && mTextures[i]->getHeight() == mHeight) continue;
    mTextures[i] = Texture::create2D(mWidth, mHeight, mTextureFormat[i]);
    for (auto& v : mSlotTextures)
      if (i < v.size()) v[i] = Texture::create2D(mWidth, mHeight, mTextureFormat[i]);
  }
  mUpdatedFlag = true;
}

// ------------------------------------------------------------------------------------------------
// RenderPass trampolines (SharedUtils/RenderPass.cpp:26-62)
// ------------------------------------------------------------------------------------------------
bool RenderPass::onInitialize(RenderContext* ctx, ResourceManager::SharedPtr rm) {
  mIsInitialized = initialize(ctx, rm);
  return mIsInitialized;
}
void RenderPass::onExecute(RenderContext* ctx) {
  mRefreshFlag = false;
  execute(ctx);
}
void RenderPass::onShutdown() {
  if (mIsInitialized) shutdown();
  mIsInitialized = false;
}

// ------------------------------------------------------------------------------------------------
// RayLaunch: one bdpt_ctx per device, shared by the passes
// ------------------------------------------------------------------------------------------------
RayLaunch::SharedPtr RayLaunch::create(RenderContext* ctx) {
  static std::map<int, std::weak_ptr<RayLaunch>> perDevice;
  static std::mutex perDeviceLock;  // a multi-GPU host runs one pipeline per device, each on a thread of its own
  std::lock_guard<std::mutex> guard(perDeviceLock);
  const int dev = ctx ? ctx->getDevice() : 0;
  if (auto sp = perDevice[dev].lock()) return sp;
  SharedPtr r(new RayLaunch());
  if (bdpt_create(dev, &r->mCtx) != BDPT_OK) {
    std::fprintf(stderr, "[RayLaunch] bdpt_create(%d) failed: no usable HIP device\n", dev);
    return nullptr;
  }
  r->mDevice = dev;
  perDevice[dev] = r;
  return r;
}
void RayLaunch::freeTileSplat() {
  for (uint64_t* p : mTileSplat)
    if (p) (void)hipFree(p);
  mTileSplat.clear();
}
void RayLaunch::setTiling(TileExchange::SharedPtr x) {
  mExchange = x;
  mW = mH = 0;  // the next ensureSize re-sizes every context as stripes
}
RayLaunch::~RayLaunch() {
  freeTileSplat();
  for (bdpt_ctx* c : mMore)
    if (c) bdpt_destroy(c);
  if (mCtx) bdpt_destroy(mCtx);
}
const char* RayLaunch::lastError() const { return bdpt_last_error(ctx()); }
bool RayLaunch::setSlotCount(uint32_t n) {
  while (mMore.size() + 1 > n && !mMore.empty()) {
    bdpt_destroy(mMore.back());
    mMore.pop_back();
  }
  while (mMore.size() + 1 < n) {
    bdpt_ctx* c = nullptr;
    if (bdpt_create(mDevice, &c) != BDPT_OK) return false;
    mMore.push_back(c);
  }
  mSlot = 0;
  mSceneSet = false;  // the new contexts have no scene yet
  mW = mH = 0;
  if (mpScene) {
    Scene::SharedPtr sc = mpScene;
    mpScene = nullptr;
    setScene(sc);
  }
  return true;
}
void RayLaunch::setScene(Scene::SharedPtr pScene) {
  if (!pScene || !mCtx) return;
  if (pScene == mpScene && mSceneSet) return;
  mpScene = pScene;
  pScene->addDefaultLightIfNone();  // SceneLoaderWrapper.cpp:71-78
  bdpt_scene_desc d;
  pScene->getDesc(&d);
  mSceneSet = bdpt_set_scene(mCtx, &d) == BDPT_OK;
  if (!mSceneSet) std::fprintf(stderr, "[RayLaunch] bdpt_set_scene failed: %s\n", bdpt_last_error(mCtx));
  for (bdpt_ctx* c : mMore)
    if (mSceneSet && bdpt_set_scene(c, &d) != BDPT_OK) {
      std::fprintf(stderr, "[RayLaunch] bdpt_set_scene failed: %s\n", bdpt_last_error(c));
      mSceneSet = false;
    }
}
bool RayLaunch::ensureSize(uint32_t w, uint32_t h) {
  if (!mCtx) return false;
  if (w == mW && h == mH && mSizedDepth == mMaxDepth) return true;
  bdpt_tile tile{0, h};
  std::vector<bdpt_ctx*> all{mCtx};
  all.insert(all.end(), mMore.begin(), mMore.end());
  if (mExchange) {
    // this rank's interleaved stripes; the reduced chunk of every frame slot lands in a buffer of its own
    mStripeRows = bdpt_stripe_rows(h, mExchange->world());
    const bdpt_stripes st{mStripeRows, mExchange->world(), mExchange->rank()};
    for (bdpt_ctx* c : all)
      if (bdpt_resize_stripes(c, w, h, st, mMaxDepth) != BDPT_OK) {
        std::fprintf(stderr, "[RayLaunch] bdpt_resize_stripes failed: %s\n", bdpt_last_error(c));
        return false;
      }
    if (bdpt_get_tile_info(mCtx, &mTileInfo) != BDPT_OK) return false;
    freeTileSplat();
    (void)hipSetDevice(mDevice);
    for (size_t i = 0; i < all.size(); i++) {
      uint64_t* p = nullptr;
      if (hipMalloc(&p, std::max<size_t>((size_t)mTileInfo.chunkU64 * 8, 16)) != hipSuccess) return false;
      mTileSplat.push_back(p);
    }
  } else
  for (bdpt_ctx* c : all)
    if (bdpt_resize(c, w, h, tile, mMaxDepth) != BDPT_OK) {
      std::fprintf(stderr, "[RayLaunch] bdpt_resize failed: %s\n", bdpt_last_error(c));
      return false;
    }
  mW = w;
  mH = h;
  mSizedDepth = mMaxDepth;
  return true;
}

static bool fillGBuffer(ResourceManager& rm, bdpt_gbuffer& gb) {
  Texture::SharedPtr p = rm.getTexture("WorldPosition"), n = rm.getTexture("WorldNormal"), d = rm.getTexture("MaterialDiffuse"),
                     s = rm.getTexture("MaterialSpecRough"), x = rm.getTexture("MaterialExtraParams"), e = rm.getTexture("Emissive");
  if (!p || !n || !d || !s || !x || !e) return false;
  // formats fixed by the first requester (LightProbeGBufferPass.cpp:46-51): position 32F, the rest 16F
  if (p->getFormat() != ResourceFormat::RGBA32Float) return false;
  for (Texture::SharedPtr t : {n, d, s, x, e})
    if (t->getFormat() != ResourceFormat::RGBA16Float) return false;
  gb.worldPosition = (float*)p->getDevicePointer();
  gb.worldNormal = (uint16_t*)n->getDevicePointer();
  gb.materialDiffuse = (uint16_t*)d->getDevicePointer();
  gb.materialSpecRough = (uint16_t*)s->getDevicePointer();
  gb.materialExtraParams = (uint16_t*)x->getDevicePointer();
  gb.emissive = (uint16_t*)e->getDevicePointer();
  return true;
}

// ------------------------------------------------------------------------------------------------
// LightProbeGBufferPass (CommonPasses/LightProbeGBufferPass.cpp:41-161)
// ------------------------------------------------------------------------------------------------
bool LightProbeGBufferPass::initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) {
  mpResManager = pResManager;
  mpResManager->requestTextureResource("WorldPosition");
  mpResManager->requestTextureResource("WorldNormal", ResourceFormat::RGBA16Float);
  mpResManager->requestTextureResource("MaterialDiffuse", ResourceFormat::RGBA16Float);
  mpResManager->requestTextureResource("MaterialSpecRough", ResourceFormat::RGBA16Float);
  mpResManager->requestTextureResource("MaterialExtraParams", ResourceFormat::RGBA16Float);
  mpResManager->requestTextureResource("Emissive", ResourceFormat::RGBA16Float);
  mpRays = RayLaunch::create(pRenderContext);
  if (!mpRays) return false;
  if (mpScene) mpRays->setScene(mpScene);
  setGuiSize(ivec2{250, 220});
  return true;
}
void LightProbeGBufferPass::initScene(RenderContext*, Scene::SharedPtr pScene) {
  mpScene = pScene;
  if (mpRays) mpRays->setScene(mpScene);
}
void LightProbeGBufferPass::renderGui(Gui* pGui) {
  int dirty = 0;
  dirty |= (int)pGui->addCheckBox(mUseThinLens ? "Using thin lens model" : "Using pinhole camera model", mUseThinLens);
  if (mUseThinLens) {
    dirty |= (int)pGui->addFloatVar("f stop", mFStop, 1.0f, 128.0f, 0.01f, true);
    dirty |= (int)pGui->addFloatVar("f plane", mFocalLength, 0.01f, 3.4e38f, 0.01f, true);
  }
  dirty |= (int)pGui->addCheckBox(mUseJitter ? "Using camera jitter" : "No camera jitter", mUseJitter);
  if (dirty) setRefreshFlag();
}
void LightProbeGBufferPass::execute(RenderContext* pRenderContext) {
  if (!mpRays || !mpRays->readyToRender()) return;
  const vec4 zero{0, 0, 0, 0};
  Texture::SharedPtr wsPos = mpResManager->getClearedTexture("WorldPosition", zero);
  mpResManager->getClearedTexture("WorldNormal", zero);
  mpResManager->getClearedTexture("MaterialDiffuse", zero);
  mpResManager->getClearedTexture("MaterialSpecRough", zero);
  mpResManager->getClearedTexture("MaterialExtraParams", zero);
  mpResManager->getClearedTexture("Emissive", zero);
  bdpt_gbuffer gb;
  if (!wsPos || !fillGBuffer(*mpResManager, gb)) return;
  if (!mpRays->ensureSize(wsPos->getWidth(), wsPos->getHeight())) return;
  Camera::SharedPtr cam = mpScene->getActiveCamera();
  cam->setAspectRatio((float)wsPos->getWidth() / (float)wsPos->getHeight());  // SceneLoaderWrapper.cpp:98
  bdpt_set_camera(mpRays->ctx(), &cam->getData());

  mLensRadius = mFocalLength / (2.0f * mFStop);
  bdpt_gbuffer_params gp;
  std::memset(&gp, 0, sizeof(gp));
  Texture::SharedPtr env = mpResManager->getTexture(ResourceManager::kEnvironmentMap);
  gp.envMap = env ? (const float*)env->getDevicePointer() : nullptr;
  gp.envWidth = env ? env->getWidth() : 0;
  gp.envHeight = env ? env->getHeight() : 0;
  gp.envColor[0] = 0.5f;
  gp.envColor[1] = 0.5f;
  gp.envColor[2] = 0.8f;
  gp.envColor[3] = 1.0f;
  gp.useThinLens = mUseThinLens ? 1u : 0u;
  gp.frameCount = mFrameCount;
  gp.lensRadius = mLensRadius;
  gp.focalLen = mFocalLength;
  if (mUseJitter) {
    bdpt_msaa_jitter(mFrameCount, gp.pixelJitter);  // kMSAA[(mFrameCount after ++) % 8], LightProbeGBufferPass.cpp:140-147
    cam->setJitter((gp.pixelJitter[0] - 0.5f) / (float)wsPos->getWidth(), (gp.pixelJitter[1] - 0.5f) / (float)wsPos->getHeight());
  } else {
    gp.pixelJitter[0] = gp.pixelJitter[1] = 0.5f;
    cam->setJitter(0, 0);
  }
  mFrameCount++;
  if (bdpt_gbuffer_execute(mpRays->ctx(), &gp, &gb, pRenderContext->getStream()) != BDPT_OK)
    std::fprintf(stderr, "[LightProbeGBufferPass] %s\n", mpRays->lastError());
}

// ------------------------------------------------------------------------------------------------
// BDPTPass (BidirectionalPathtracing/Passes/BDPTPass.cpp:23-107)
// ------------------------------------------------------------------------------------------------
bool BDPTPass::initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) {
  mpResManager = pResManager;
  // default-format requests lose against the G-buffer pass's RGBA16F ones and return -1, as in the reference
  mpResManager->requestTextureResources({"WorldPosition", "WorldNormal", "MaterialDiffuse", "MaterialSpecRough", "MaterialExtraParams", "Emissive"});
  mpResManager->requestTextureResource(mOutputTextureName);
  mpResManager->requestTextureResource(ResourceManager::kEnvironmentMap);
  mpResManager->setDefaultSceneName("Data/pink_room/pink_room.fscene");
  mpRays = RayLaunch::create(pRenderContext);
  if (!mpRays) return false;
  mpRays->setMaxRecursionDepth(uint32_t(mMaxPossibleRayDepth));
  if (mpScene) mpRays->setScene(mpScene);
  return true;
}
void BDPTPass::initScene(RenderContext*, Scene::SharedPtr pScene) {
  mpScene = pScene;
  if (mpRays) mpRays->setScene(mpScene);
}
void BDPTPass::resize(uint32_t width, uint32_t height) {
  if (mpRays && width && height) (void)mpRays->ensureSize(width, height);
}
void BDPTPass::renderGui(Gui* pGui) {
  int dirty = 0;
  dirty |= (int)pGui->addIntVar("Max Ray Depth", mUserSpecifiedRayDepth, 0, mMaxPossibleRayDepth);
  dirty |= (int)pGui->addIntVar("Material", mMaterialIndex, 0, mNumOfMaterials - 1);
  dirty |= (int)pGui->addFloatVar("Clamping Upper Bound", mClampUpper, 0.001f, 1.0f);
  dirty |= (int)pGui->addFloatVar("Refractive Index (only for dielectric material)", mRefractiveIndex, 0.1f, 5.0f);
  if (dirty) setRefreshFlag();
}
void BDPTPass::execute(RenderContext* pRenderContext) {
  Texture::SharedPtr pDstTex = mpResManager->getTexture(mOutputTextureName);  // cleared by bdpt_execute itself
  if (!pDstTex || !mpRays || !mpRays->readyToRender()) return;                 // silent no-op, BDPTPass.cpp:76
  bdpt_gbuffer gb;
  if (!fillGBuffer(*mpResManager, gb)) return;
  if (!mpRays->ensureSize(pDstTex->getWidth(), pDstTex->getHeight())) return;
  bdpt_params p;
  std::memset(&p, 0, sizeof(p));
  p.minT = mpResManager->getMinTDist();
  p.frameCount = mFrameCount;
  p.maxDepth = (uint32_t)mUserSpecifiedRayDepth;
  p.emitMult = 1.0f;
  p.matIndex = (uint32_t)mMaterialIndex;
  p.clampUpper = mClampUpper;
  p.refractiveIndex = mRefractiveIndex;
  p.flags = mParamFlags;
  bdpt_msaa_jitter(mFrameCount, p.pixelJitter);  // kMSAA[(mFrameCount after ++) % 8], BDPTPass.cpp:81,97-101
  mFrameCount++;
  Camera::SharedPtr cam = mpScene->getActiveCamera();
  cam->setJitter((p.pixelJitter[0] - 0.5f) / (float)pDstTex->getWidth(), (p.pixelJitter[1] - 0.5f) / (float)pDstTex->getHeight());
  bdpt_set_camera(mpRays->ctx(), &cam->getData());
  {  // the "EnvironmentMap" channel the pass requests (BDPTPass.cpp:29); only read with BDPT_PARAM_ENV_ON_MISS
    Texture::SharedPtr env = mpResManager->getTexture(ResourceManager::kEnvironmentMap);
    bdpt_environment e{};
    if (env && env->getFormat() == ResourceFormat::RGBA32Float) {
      e.envMap = (const float*)env->getDevicePointer();
      e.width = env->getWidth();
      e.height = env->getHeight();
    }
    bdpt_set_environment(mpRays->ctx(), &e);
  }
  float* out = (float*)pDstTex->getDevicePointer();
  hipStream_t stream = pRenderContext->getStream();
  if (!mpRays->tiled()) {
    if (bdpt_execute(mpRays->ctx(), &p, &gb, out, stream) != BDPT_OK) std::fprintf(stderr, "[BDPTPass] %s\n", mpRays->lastError());
    return;
  }
  // Tiled over several GPUs (RenderingPipeline::setTiling): this rank's stripes.  Phase 1 enqueues everything that
  // writes the splat accumulators; the reduce-scatter then runs on the exchange stream beside phase 2 (the zero-valued
  // connection rounds, which never touch them); the rank's own chunk, summed over all ranks, is folded into its rows.
  TileExchange::SharedPtr x = mpRays->exchange();
  const uint32_t slot = mpRays->currentSlot();
  uint64_t* full = nullptr;
  uint64_t fullWords = 0;
  p.flags |= BDPT_PARAM_DEFER_RESOLVE | BDPT_PARAM_DEFER_TAIL;
  bool ok = bdpt_execute(mpRays->ctx(), &p, &gb, out, stream) == BDPT_OK;
  ok = ok && bdpt_splat_buffer(mpRays->ctx(), &full, &fullWords) == BDPT_OK && fullWords == mpRays->tileInfo().splatU64;
  // this rank cannot enter the frame's collective: its peers must not wait in theirs (TileExchange::abort)
  if (!ok && !x->aborted()) x->abort(std::string("the BDPT pass failed before the splat exchange: ") + mpRays->lastError());
  if (ok && !x->reduceScatter(full, mpRays->tileSplat(), mpRays->tileInfo().chunkU64, stream, slot)) {
    std::fprintf(stderr, "[BDPTPass] splat exchange failed: %s\n", x->lastError().c_str());
    ok = false;
  }
  ok = ok && bdpt_execute_tail(mpRays->ctx(), &p, &gb, out, stream) == BDPT_OK;
  if (ok) {
    x->waitFor(stream, slot);
    ok = bdpt_resolve_tile(mpRays->ctx(), mpRays->tileSplat(), out, stream) == BDPT_OK;
  }
  if (!ok) std::fprintf(stderr, "[BDPTPass] %s\n", mpRays->lastError());
}

// ------------------------------------------------------------------------------------------------
// SimpleAccumulationPass (CommonPasses/SimpleAccumulationPass.cpp:24-140)
// ------------------------------------------------------------------------------------------------
bool SimpleAccumulationPass::initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) {
  if (!pResManager) return false;
  mpResManager = pResManager;
  mpResManager->requestTextureResource(mAccumChannel);
  mpRays = RayLaunch::create(pRenderContext);
  setGuiSize(ivec2{250, 135});
  return mpRays != nullptr;
}
void SimpleAccumulationPass::initScene(RenderContext*, Scene::SharedPtr pScene) {
  mAccumCount = 0;
  mpScene = pScene;
  if (mpScene && mpScene->getActiveCamera()) {
    mpScene->getActiveCamera()->getData();
    mLastCameraVersion = mpScene->getActiveCamera()->getViewVersion();
  }
}
void SimpleAccumulationPass::resize(uint32_t width, uint32_t height) {
  mpLastFrame = Texture::create2D(width, height, ResourceFormat::RGBA32Float);
  mAccumCount = 0;
}
void SimpleAccumulationPass::renderGui(Gui* pGui) {
  pGui->addText((std::string("Accumulating buffer:   ") + mAccumChannel).c_str());
  if (pGui->addCheckBox(mDoAccumulation ? "Accumulating samples temporally" : "No temporal accumulation", mDoAccumulation)) {
    mAccumCount = 0;
    setRefreshFlag();
  }
  (void)pGui->addIntVar("Max frames to accumulate", mCountLimit, 1, mMaxCountLimit);
  pGui->addText((std::string("Frames accumulated: ") + std::to_string(mAccumCount)).c_str());
}
bool SimpleAccumulationPass::hasCameraMoved() {
  if (!mpScene || !mpScene->getActiveCamera()) return false;
  mpScene->getActiveCamera()->getData();  // refresh the basis if a setter marked it dirty
  return mLastCameraVersion != mpScene->getActiveCamera()->getViewVersion();
}
void SimpleAccumulationPass::execute(RenderContext* pRenderContext) {
  Texture::SharedPtr inputTexture = mpResManager->getTexture(mAccumChannel);
  if (!inputTexture || !mDoAccumulation || !mpLastFrame || !mpRays) return;
  if (hasCameraMoved()) {
    if (!mResumed) mAccumCount = 0;
    mLastCameraVersion = mpScene->getActiveCamera()->getViewVersion();
  }
  mResumed = false;
  const uint32_t gAccumCount = (int32_t)mAccumCount < mCountLimit ? mAccumCount++ : (uint32_t)mCountLimit;
  if (mpRays->tiled()) {  // the running mean over this rank's rows only (the other rows belong to other GPUs)
    if (!mpRays->ensureSize(inputTexture->getWidth(), inputTexture->getHeight())) return;
    bdpt_accumulate_tile(mpRays->ctx(), (float*)mpLastFrame->getDevicePointer(), (float*)inputTexture->getDevicePointer(), gAccumCount,
                         (uint32_t)mCountLimit, pRenderContext->getStream());
    return;
  }
  bdpt_accumulate(mpRays->ctx(), (float*)mpLastFrame->getDevicePointer(), (float*)inputTexture->getDevicePointer(), gAccumCount,
                  (uint32_t)mCountLimit, (uint64_t)inputTexture->getWidth() * inputTexture->getHeight(), pRenderContext->getStream());
}
void SimpleAccumulationPass::stateRefreshed() { mAccumCount = 0; }

// ------------------------------------------------------------------------------------------------
// BlockwiseMultiOrderFeatureRegression (BidirectionalPathtracing/Passes/DenoisePass.cpp:12-279)
// ------------------------------------------------------------------------------------------------
bool BlockwiseMultiOrderFeatureRegression::initialize(RenderContext* pRenderContext, ResourceManager::SharedPtr pResManager) {
  if (!pResManager) return false;
  mpResManager = pResManager;
  mpResManager->requestTextureResource(mDenoiseChannel);
  mpResManager->requestTextureResources({"WorldPosition", "WorldNormal", "MaterialDiffuse"});  // the three feature buffers
  mpRays = RayLaunch::create(pRenderContext);
  setGuiSize(ivec2{250, 135});
  return mpRays != nullptr;
}
void BlockwiseMultiOrderFeatureRegression::initScene(RenderContext*, Scene::SharedPtr pScene) {
  mpScene = pScene;
  mAccumCount = 0;
  if (mpRays) bdpt_bmfr_reset(mpRays->ctx());
}
void BlockwiseMultiOrderFeatureRegression::resize(uint32_t, uint32_t) {
  mAccumCount = 0;  // the history itself goes with bdpt_resize
  if (mpRays) bdpt_bmfr_reset(mpRays->ctx());
}
void BlockwiseMultiOrderFeatureRegression::renderGui(Gui* pGui) {
  int dirty = 0;
  dirty |= (int)pGui->addCheckBox(mDoDenoise ? "Do BMFR Denoise" : "Ignore the denoise stage", mDoDenoise);
  dirty |= (int)pGui->addCheckBox(mBMFR_preprocess ? "Do Pre-Process" : "Skip Pre-process", mBMFR_preprocess);
  dirty |= (int)pGui->addCheckBox(mBMFR_regression ? "Do Regression" : "Skip Regression", mBMFR_regression);
  dirty |= (int)pGui->addCheckBox(mBMFR_postprocess ? "Do Post-Process" : "Skip Post-process", mBMFR_postprocess);
  dirty |= (int)pGui->addCheckBox(mBMFR_removeFeatures ? "Ignore Linearly Dependent Features" : "Add Noise", mBMFR_removeFeatures);
  if (dirty) setRefreshFlag();
}
void BlockwiseMultiOrderFeatureRegression::execute(RenderContext* pRenderContext) {
  if (!mpResManager || !mpRays) return;
  Texture::SharedPtr inputTexture = mpResManager->getTexture(mDenoiseChannel);
  if (!inputTexture || !mDoDenoise) return;
  Texture::SharedPtr pos = mpResManager->getTexture("WorldPosition"), nrm = mpResManager->getTexture("WorldNormal"),
                     alb = mpResManager->getTexture("MaterialDiffuse");
  if (!pos || !nrm || !alb || !mpScene || !mpScene->getActiveCamera()) return;
  if (!mpRays->ensureSize(inputTexture->getWidth(), inputTexture->getHeight())) return;
  bdpt_bmfr_params p{};
  p.frameNumber = mAccumCount;
  p.flags = (mBMFR_preprocess ? BDPT_BMFR_PREPROCESS : 0u) | (mBMFR_regression ? BDPT_BMFR_REGRESSION : 0u) |
            (mBMFR_postprocess ? BDPT_BMFR_POSTPROCESS : 0u) | (mBMFR_removeFeatures ? 0u : BDPT_BMFR_KEEP_LD_FEATURES);
  std::memcpy(p.prevViewProj, mpScene->getActiveCamera()->getPrevViewProjMat(), sizeof(p.prevViewProj));
  bdpt_gbuffer gb{};
  gb.worldPosition = (float*)pos->getDevicePointer();
  gb.worldNormal = (uint16_t*)nrm->getDevicePointer();
  gb.materialDiffuse = (uint16_t*)alb->getDevicePointer();
  float* noisy = (float*)inputTexture->getDevicePointer();
  if (mpRays->tiled()) {  // a rank holds its stripes: gather the whole frame of all four channels first
    if (!gatherWholeFrame(pRenderContext, inputTexture, pos, nrm, alb)) {
      std::fprintf(stderr, "[BMFR] gathering the frame failed: %s\n", mpRays->exchange()->lastError().c_str());
      return;
    }
    gb.worldPosition = mFullPos;
    gb.worldNormal = mFullNorm;
    gb.materialDiffuse = mFullAlb;
    noisy = mFullNoisy;
  }
  if (bdpt_bmfr_execute(mpRays->ctx(), &p, &gb, noisy, pRenderContext->getStream()) != BDPT_OK) {
    std::fprintf(stderr, "[BMFR] %s\n", mpRays->lastError());
    return;
  }
  if (mpRays->tiled())  // every rank filtered the same whole frame: its output channel now holds all rows of it
    (void)hipMemcpyAsync(inputTexture->getDevicePointer(), mFullNoisy, (size_t)mGatherW * mGatherH * 16, hipMemcpyDeviceToDevice, pRenderContext->getStream());
  mAccumCount++;
}
void BlockwiseMultiOrderFeatureRegression::freeGather() {
  for (void* q : {(void*)mPackedMine, (void*)mPackedAll, (void*)mFullNoisy, (void*)mFullPos, (void*)mFullNorm, (void*)mFullAlb})
    if (q) (void)hipFree(q);
  mPackedMine = mPackedAll = nullptr;
  mFullNoisy = mFullPos = nullptr;
  mFullNorm = mFullAlb = nullptr;
  mGatherW = mGatherH = 0;
}
// A rank's chunk of the all-gather: [noisy 16 B | position 16 B | normal 8 B | albedo 8 B] x (chunkRows x W) pixels, channel
// after channel; chunkRows is the same on every rank (bdpt_get_tile_info), rows past a rank's own are padding that
// bdpt_tile_unpack maps outside the frame and skips.  Collective: every rank's denoiser runs every frame.
bool BlockwiseMultiOrderFeatureRegression::gatherWholeFrame(RenderContext* pRenderContext, Texture::SharedPtr noisy, Texture::SharedPtr pos,
                                                            Texture::SharedPtr nrm, Texture::SharedPtr alb) {
  const uint32_t W = noisy->getWidth(), H = noisy->getHeight(), world = mpRays->exchange()->world();
  const size_t px = (size_t)mpRays->tileInfo().chunkRows * W, chunk = px * 48;
  static const uint32_t kBpp[4] = {16, 16, 8, 8};
  if (mGatherW != W || mGatherH != H) {
    freeGather();
    const size_t n = (size_t)W * H;
    if (hipMalloc((void**)&mPackedMine, std::max<size_t>(chunk, 16)) != hipSuccess || hipMalloc((void**)&mPackedAll, std::max<size_t>(chunk * world, 16)) != hipSuccess ||
        hipMalloc((void**)&mFullNoisy, n * 16) != hipSuccess || hipMalloc((void**)&mFullPos, n * 16) != hipSuccess ||
        hipMalloc((void**)&mFullNorm, n * 8) != hipSuccess || hipMalloc((void**)&mFullAlb, n * 8) != hipSuccess) {
      freeGather();
      mpRays->exchange()->abort("out of device memory for the denoiser's whole-frame buffers");  // peers: do not wait
      return false;
    }
    mGatherW = W;
    mGatherH = H;
  }
  hipStream_t st = pRenderContext->getStream();
  const void* src[4] = {noisy->getDevicePointer(), pos->getDevicePointer(), nrm->getDevicePointer(), alb->getDevicePointer()};
  void* full[4] = {mFullNoisy, mFullPos, mFullNorm, mFullAlb};
  size_t off = 0;
  for (int k = 0; k < 4; k++) {
    if (bdpt_tile_pack(mpRays->ctx(), src[k], mPackedMine + off, kBpp[k], st) != BDPT_OK) {
      mpRays->exchange()->abort(std::string("packing the denoiser's channels failed: ") + mpRays->lastError());  // peers: do not wait
      return false;
    }
    off += px * kBpp[k];
  }
  if (!mpRays->exchange()->allGather((const float*)mPackedMine, (float*)mPackedAll, chunk / 4, st)) return false;
  for (uint32_t r = 0; r < world; r++) {
    off = 0;
    for (int k = 0; k < 4; k++) {
      if (bdpt_tile_unpack(mpRays->ctx(), r, mPackedAll + (size_t)r * chunk + off, full[k], kBpp[k], st) != BDPT_OK) return false;
      off += px * kBpp[k];
    }
  }
  return true;
}
// [frame number][switches][history bytes][history]: the denoiser switched off, or not run yet, saves no history; a state
// written under other switches is refused (a run resumed with the denoiser on from frames rendered without it would
// filter against a history that does not exist)
static uint32_t bmfrSwitches(bool on, bool pre, bool reg, bool post, bool removeFeatures) {
  return (on ? 1u : 0u) | (pre ? 2u : 0u) | (reg ? 4u : 0u) | (post ? 8u : 0u) | (removeFeatures ? 16u : 0u);
}
void BlockwiseMultiOrderFeatureRegression::saveState(RenderContext*, std::vector<uint8_t>& out) {
  put32(out, mAccumCount);
  put32(out, bmfrSwitches(mDoDenoise, mBMFR_preprocess, mBMFR_regression, mBMFR_postprocess, mBMFR_removeFeatures));
  uint64_t bytes = 0;
  if (mDoDenoise && mpRays) (void)bdpt_bmfr_history_bytes(mpRays->ctx(), &bytes);
  std::vector<uint8_t> blob((size_t)bytes);
  if (bytes && bdpt_bmfr_save_history(mpRays->ctx(), blob.data(), bytes) != BDPT_OK) {
    std::fprintf(stderr, "[BMFR] %s\n", mpRays->lastError());
    blob.clear();
  }
  put64(out, blob.size());
  out.insert(out.end(), blob.begin(), blob.end());
}
bool BlockwiseMultiOrderFeatureRegression::loadState(RenderContext*, const uint8_t* data, size_t size) {
  if (size < 16) return false;
  if (get32(data + 4) != bmfrSwitches(mDoDenoise, mBMFR_preprocess, mBMFR_regression, mBMFR_postprocess, mBMFR_removeFeatures)) return false;
  const uint64_t bytes = get64(data + 8);
  if (bytes != size - 16) return false;
  if (bytes) {
    if (!mpRays || !mpResManager) return false;
    if (!mpRays->ensureSize(mpResManager->getScreenSize().x, mpResManager->getScreenSize().y)) return false;
    if (bdpt_bmfr_load_history(mpRays->ctx(), data + 16, bytes) != BDPT_OK) {
      std::fprintf(stderr, "[BMFR] %s\n", mpRays->lastError());
      return false;
    }
  }
  mAccumCount = get32(data);
  return true;
}

// ------------------------------------------------------------------------------------------------
// RenderingPipeline (headless subset of SharedUtils/RenderingPipeline.cpp)
// ------------------------------------------------------------------------------------------------
RenderingPipeline::RenderingPipeline() : mContext(0, nullptr), mWidth(0), mHeight(0) {}
RenderingPipeline::RenderingPipeline(uint32_t width, uint32_t height, int device) : mContext(0, nullptr), mWidth(0), mHeight(0) {
  setSize(width, height, device);
}
void RenderingPipeline::setSize(uint32_t width, uint32_t height, int device) {
  mContext = RenderContext(device, nullptr);
  mWidth = width;
  mHeight = height;
  (void)hipSetDevice(device);
  mpResourceManager = ResourceManager::create(width, height, &mContext);
}
void RenderingPipeline::run(RenderingPipeline* pipe, SampleConfig& config) {
  if (!pipe) return;
  pipe->setSize(config.windowDesc.width, config.windowDesc.height, 0);
  const char* sceneName = std::getenv("BDPT_SCENE");
  Scene::SharedPtr pScene;
  if (sceneName && std::strchr(sceneName, '.')) {
    std::string err;
    pScene = Scene::loadFromFile(sceneName, &err);
    if (!pScene) std::fprintf(stderr, "[RenderingPipeline] %s\n", err.c_str());
  } else if (sceneName && std::strcmp(sceneName, "atrium") == 0) {
    pScene = Scene::createAtrium(1, 262144);
  }
  if (!pScene) pScene = Scene::createCornellBox();
  if (pipe->initialize(pScene)) {
    const char* f = std::getenv("BDPT_FRAMES");
    const int frames = f ? std::max(1, std::atoi(f)) : 1;
    for (int i = 0; i < frames; i++) pipe->renderFrame();
    pipe->getRenderContext()->flush(true);
    std::printf("%s: %d frame(s) of %ux%u rendered\n", config.windowDesc.title.c_str(), frames, config.windowDesc.width, config.windowDesc.height);
  } else {
    std::fprintf(stderr, "[RenderingPipeline] initialisation failed (no GPU?)\n");
  }
  delete pipe;
}
RenderingPipeline::~RenderingPipeline() {
  for (hipStream_t st : mSlotStreams)
    if (st) {
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    }
  for (hipEvent_t e : mOrderEvents)
    if (e) (void)hipEventDestroy(e);
  for (auto& p : mActivePasses)
    if (p) p->onShutdown();
  mActivePasses.clear();
}
bool RenderingPipeline::setTiling(uint32_t rank, uint32_t world, ncclComm_t comm) {
  if (world == 0 || rank >= world || (world > 1 && !comm)) return false;
  mTileRank = rank;
  mTileWorld = world;
  mTileComm = comm;
  return true;
}
std::string RenderingPipeline::rankPath(const std::string& path) const {
  return mTileWorld > 0 ? path + ".rank" + std::to_string(mTileRank) + "of" + std::to_string(mTileWorld) : path;
}
bool RenderingPipeline::inFlightActive() {
  if (mFramesInFlight <= 1 || mSlotStreams.empty()) return false;
  for (auto& p : mActivePasses)
    if (p && (p->onHasUnsavedCrossFrameState() || p->onHoldsTemporalState())) return false;  // a pass with temporal state of its own needs the frames one by one
  return true;
}
void RenderingPipeline::setPass(uint32_t passNum, RenderPass::SharedPtr pTargetPass) {
  if (mActivePasses.size() <= passNum) mActivePasses.resize(passNum + 1);
  mActivePasses[passNum] = pTargetPass;
}
bool RenderingPipeline::initialize(Scene::SharedPtr pScene) {
  if (!mpResourceManager) return false;  // no size yet (default-constructed and neither run() nor setSize() called)
  mpScene = pScene;
  mpResourceManager->requestTextureResource(ResourceManager::kOutputChannel);
  mpResourceManager->updateEnvironmentMap("");  // the HDR probe blob is absent: default constant environment
  for (auto& p : mActivePasses) {
    if (!p) continue;
    if (!p->onInitialize(&mContext, mpResourceManager)) p = nullptr;  // a failing pass is dropped (RenderingPipeline.cpp:58-59)
  }
  mpResourceManager->initializeResources();
  if (mTileWorld > 0) {  // multi-GPU: the launcher's contexts render this rank's stripes
    mpRays = RayLaunch::create(&mContext);
    TileExchange::SharedPtr x = TileExchange::create(mContext.getDevice(), mTileRank, mTileWorld, mTileComm);
    if (!mpRays || !x) return false;
    if (mTileAbort) x->setAbortHandler(mTileAbort);
    mpRays->setTiling(x);
  }
  if (mFramesInFlight > 1) {  // frame slots: channels, launcher contexts and streams, one set per frame in flight
    mpRays = RayLaunch::create(&mContext);
    if (!mpRays || !mpRays->setSlotCount(mFramesInFlight)) return false;
    mpResourceManager->setSlotCount(mFramesInFlight);
    mSlotStreams.assign(mFramesInFlight, nullptr);
    for (hipStream_t& st : mSlotStreams)
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return false;
    mOrderEvents.assign(mActivePasses.size(), nullptr);
    for (size_t i = 0; i < mActivePasses.size(); i++)
      if (mActivePasses[i] && mActivePasses[i]->onNeedsFrameOrder())
        if (hipEventCreateWithFlags(&mOrderEvents[i], hipEventDisableTiming) != hipSuccess) return false;
  }
  bool any = false;
  for (auto& p : mActivePasses) {
    if (!p) continue;
    any = true;
    p->onResize(mWidth, mHeight);
    if (pScene) p->onInitScene(&mContext, pScene);
  }
  return any;
}
void RenderingPipeline::applyGui(Gui* pGui) {
  for (auto& p : mActivePasses)
    if (p) p->onRenderGui(pGui);
}
void RenderingPipeline::renderFrame() {
  if (mpScene && mpScene->getActiveCamera()) mpScene->getActiveCamera()->beginFrame();  // Scene::update, RenderingPipeline.cpp:630
  bool refresh = false;
  for (auto& p : mActivePasses) refresh |= (p && p->isRefreshFlagSet());
  if (refresh)
    for (auto& p : mActivePasses)
      if (p) p->onStateRefresh();  // RenderingPipeline.cpp:635-663
  if (!inFlightActive()) {
    for (auto& p : mActivePasses)
      if (p) p->onExecute(&mContext);
    mFrameIndex++;
    return;
  }
  // this frame's slot: its stream (in order behind the frame that used the slot before), channels and launcher context
  const uint32_t slot = (uint32_t)(mFrameIndex % mFramesInFlight);
  mContext.setStream(mSlotStreams[slot]);
  mpResourceManager->setCurrentSlot(slot);
  mpRays->setCurrentSlot(slot);
  for (size_t i = 0; i < mActivePasses.size(); i++) {
    RenderPass::SharedPtr& p = mActivePasses[i];
    if (!p) continue;
    const bool ordered = mOrderEvents[i] != nullptr;
    if (ordered && mFrameIndex > 0) (void)hipStreamWaitEvent(mSlotStreams[slot], mOrderEvents[i], 0);  // behind the previous frame's
    p->onExecute(&mContext);
    if (ordered) (void)hipEventRecord(mOrderEvents[i], mSlotStreams[slot]);
  }
  mFrameIndex++;
}
// ---- checkpoints (byte helpers: top of the file)

static uint32_t floatBits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
// Each pass writes the settings its frames depend on beside its counters and refuses a state written under others:
// a run resumed with another ray depth, material model or lens would blend unrelated frames into the restored mean.
void LightProbeGBufferPass::saveState(RenderContext*, std::vector<uint8_t>& out) {
  put32(out, mFrameCount);
  put32(out, (mUseThinLens ? 1u : 0u) | (mUseJitter ? 2u : 0u));
  put32(out, floatBits(mFStop));
  put32(out, floatBits(mFocalLength));
}
bool LightProbeGBufferPass::loadState(RenderContext*, const uint8_t* data, size_t size) {
  if (size != 16) return false;
  if (get32(data + 4) != ((mUseThinLens ? 1u : 0u) | (mUseJitter ? 2u : 0u))) return false;
  if (mUseThinLens && (get32(data + 8) != floatBits(mFStop) || get32(data + 12) != floatBits(mFocalLength))) return false;
  mFrameCount = get32(data);
  return true;
}
void BDPTPass::saveState(RenderContext*, std::vector<uint8_t>& out) {
  put32(out, mFrameCount);
  put32(out, (uint32_t)mUserSpecifiedRayDepth);
  put32(out, (uint32_t)mMaterialIndex);
  put32(out, floatBits(mClampUpper));
  put32(out, mParamFlags);
  put32(out, floatBits(mpResManager ? mpResManager->getMinTDist() : 0.0f));
}
bool BDPTPass::loadState(RenderContext*, const uint8_t* data, size_t size) {
  if (size != 24) return false;
  if (get32(data + 4) != (uint32_t)mUserSpecifiedRayDepth || get32(data + 8) != (uint32_t)mMaterialIndex ||
      get32(data + 12) != floatBits(mClampUpper) || get32(data + 16) != mParamFlags ||
      get32(data + 20) != floatBits(mpResManager ? mpResManager->getMinTDist() : 0.0f))
    return false;
  mFrameCount = get32(data);
  return true;
}
void SimpleAccumulationPass::saveState(RenderContext* pRenderContext, std::vector<uint8_t>& out) {
  put32(out, mAccumCount);
  std::vector<uint8_t> raw;
  if (mpLastFrame) mpLastFrame->downloadRaw(pRenderContext->getStream(), raw);
  out.insert(out.end(), raw.begin(), raw.end());
}
bool SimpleAccumulationPass::loadState(RenderContext* pRenderContext, const uint8_t* data, size_t size) {
  if (size < 4 || !mpLastFrame || size - 4 != mpLastFrame->getSizeInBytes()) return false;
  mAccumCount = get32(data);
  // resuming is not a camera move: the first frame's own set-up (the G-buffer pass sets the aspect ratio) will bump the
  // camera's view version once more
  mResumed = true;
  return mpLastFrame->uploadRaw(pRenderContext->getStream(), data + 4, size - 4);
}

// What a checkpoint belongs to besides its passes: FNV-1a over the scene's geometry, materials, lights and camera.
uint64_t RenderingPipeline::sceneIdentity() const {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  if (!mpScene) return h;
  bdpt_scene_desc d;
  mpScene->getDesc(&d);
  mix(&d.numVertices, 4);
  mix(&d.numTriangles, 4);
  if (d.positions) mix(d.positions, (size_t)d.numVertices * 12);
  if (d.indices) mix(d.indices, (size_t)d.numTriangles * 12);
  if (d.triMaterial) mix(d.triMaterial, (size_t)d.numTriangles * 4);
  if (d.materials) mix(d.materials, (size_t)d.numMaterials * sizeof(bdpt_material));
  if (d.lights) mix(d.lights, (size_t)d.numLights * sizeof(bdpt_light));
  if (mpScene->getActiveCamera()) {
    mpScene->getActiveCamera()->setAspectRatio((float)mWidth / (float)std::max(1u, mHeight));
    const bdpt_camera& c = mpScene->getActiveCamera()->getData();
    mix(&c, sizeof(c));
  }
  return h;
}

// [magic][version][width][height][scene identity][pass count] then per pass [name length][name][state length][state]
bool RenderingPipeline::saveCheckpoint(const std::string& path) {
  for (auto& pass : mActivePasses)
    if (pass && pass->onHasUnsavedCrossFrameState()) {
      std::fprintf(stderr, "[RenderingPipeline] pass '%s' holds cross-frame state that cannot be saved: no checkpoint written\n", pass->getName().c_str());
      return false;
    }
  for (hipStream_t st : mSlotStreams) (void)hipStreamSynchronize(st);  // frames in flight: everything submitted has landed
  std::vector<uint8_t> out;
  put32(out, kCheckpointMagic);
  put32(out, kCheckpointVersion);
  put32(out, mWidth);
  put32(out, mHeight);
  put64(out, sceneIdentity());
  put32(out, (uint32_t)mActivePasses.size());
  for (auto& pass : mActivePasses) {
    const std::string name = pass ? pass->getName() : std::string();
    std::vector<uint8_t> st;
    if (pass) pass->onSaveState(&mContext, st);
    put32(out, (uint32_t)name.size());
    out.insert(out.end(), name.begin(), name.end());
    put64(out, st.size());
    out.insert(out.end(), st.begin(), st.end());
  }
  FILE* f = std::fopen(rankPath(path).c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  return std::fclose(f) == 0 && ok;
}

bool RenderingPipeline::loadCheckpoint(const std::string& path) {
  for (auto& pass : mActivePasses)
    if (pass && pass->onHasUnsavedCrossFrameState()) {
      std::fprintf(stderr, "[RenderingPipeline] pass '%s' holds cross-frame state no checkpoint carries: not resuming\n", pass->getName().c_str());
      return false;
    }
  FILE* f = std::fopen(rankPath(path).c_str(), "rb");
  if (!f) return false;
  std::vector<uint8_t> d;
  uint8_t buf[65536];
  for (size_t n; (n = std::fread(buf, 1, sizeof(buf), f)) > 0;) d.insert(d.end(), buf, buf + n);
  std::fclose(f);
  constexpr size_t kHeader = 28;
  if (d.size() < kHeader || get32(&d[0]) != kCheckpointMagic || get32(&d[4]) != kCheckpointVersion) return false;
  if (get32(&d[8]) != mWidth || get32(&d[12]) != mHeight || get64(&d[16]) != sceneIdentity()) return false;
  if (get32(&d[24]) != mActivePasses.size()) return false;
  // first pass over the file: every section must be there, named after its pass, and inside the file (the lengths come
  // from the file: compare against what is LEFT, never add them to the cursor first)
  struct Section {
    size_t at, size;
  };
  std::vector<Section> sections;
  size_t p = kHeader;
  for (auto& pass : mActivePasses) {
    if (d.size() - p < 4) return false;
    const uint32_t nl = get32(&d[p]);
    p += 4;
    if ((size_t)nl > d.size() - p || d.size() - p - nl < 8) return false;
    if (std::string(d.begin() + (long)p, d.begin() + (long)(p + nl)) != (pass ? pass->getName() : std::string())) return false;
    p += nl;
    const uint64_t sl = get64(&d[p]);
    p += 8;
    if (sl > (uint64_t)(d.size() - p)) return false;
    sections.push_back(Section{p, (size_t)sl});
    p += (size_t)sl;
  }
  if (p != d.size()) return false;
  // deliver the refresh notification a first frame would (it resets the accumulation) BEFORE the state comes in
  for (auto& pass : mActivePasses)
    if (pass) {
      pass->onStateRefresh();
      pass->resetRefreshFlag();
    }
  for (size_t i = 0; i < mActivePasses.size(); i++) {
    if (!mActivePasses[i]) continue;
    if (!mActivePasses[i]->onLoadState(&mContext, sections[i].size ? &d[sections[i].at] : nullptr, sections[i].size)) return false;
  }
  return true;
}

std::vector<float> RenderingPipeline::readOutput() {
  for (hipStream_t st : mSlotStreams) (void)hipStreamSynchronize(st);
  Texture::SharedPtr t = mpResourceManager->getTexture(ResourceManager::kOutputChannel);  // the latest frame's slot
  if (!t) return std::vector<float>();
  if (mTileWorld == 0 || !mpRays || !mpRays->tiled()) return t->download(mContext.getStream());
  // Tiled: "tile framebuffers gathered".  Every rank packs its rows (its stripes in order, zero-padded to the chunk
  // size all ranks share), one ncclAllGather hands every rank all of them, and the rows go back to their places.
  // Collective: every rank of the group must call readOutput() at the same point of its frame sequence.
  hipStream_t stream = mContext.getStream();
  (void)hipStreamSynchronize(stream);
  const uint32_t W = t->getWidth(), H = t->getHeight(), world = mTileWorld, R = mpRays->stripeRows();
  const bdpt_tile_info& ti = mpRays->tileInfo();
  const size_t perRank = (size_t)ti.chunkRows * W * 4;  // (= stripeChunkRows(H, world, R) rows: bdpt_get_tile_info agrees, checked below)
  float *mine = nullptr, *all = nullptr;
  std::vector<float> out;
  if (ti.chunkRows != stripeChunkRows(H, world, R)) return out;
  if (hipMalloc(&mine, std::max<size_t>(perRank * 4, 16)) != hipSuccess || hipMalloc(&all, std::max<size_t>(perRank * 4 * world, 16)) != hipSuccess) {
    if (mine) (void)hipFree(mine);
    return out;
  }
  (void)hipMemsetAsync(mine, 0, perRank * 4, stream);
  const float* src = (const float*)t->getDevicePointer();
  const size_t rowFloats = (size_t)W * 4;
  for (const StripeSpan& s : stripeSpans(H, world, mTileRank, R))  // this rank's rows, in order, at the head of its chunk (RankSync.h)
    (void)hipMemcpyAsync(mine + (size_t)s.packedRow * rowFloats, src + (size_t)s.firstRow * rowFloats, (size_t)s.rows * rowFloats * 4,
                         hipMemcpyDeviceToDevice, stream);
  const bool ok = mpRays->exchange()->allGather(mine, all, perRank, stream);
  std::vector<float> packed(perRank * world);
  if (ok && hipStreamSynchronize(stream) == hipSuccess &&
      hipMemcpy(packed.data(), all, packed.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
    out.assign((size_t)W * H * 4, 0.0f);
    unpackStripes(reinterpret_cast<const uint8_t*>(packed.data()), reinterpret_cast<uint8_t*>(out.data()), H, world, R, rowFloats * 4);
  } else if (!ok) {
    std::fprintf(stderr, "[RenderingPipeline] gather failed: %s\n", mpRays->exchange()->lastError().c_str());
  }
  (void)hipFree(mine);
  (void)hipFree(all);
  return out;
}

}  // namespace bdpt
