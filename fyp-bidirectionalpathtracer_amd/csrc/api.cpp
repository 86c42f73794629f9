// api.cpp — the C ABI of include/bdpt.h: context, scene upload + BVH build, per-tile path
// buffers, stage launches.  Workspaces are sized by bdpt_set_scene / bdpt_resize / bdpt_prepare, so
// bdpt_gbuffer_execute, bdpt_execute and bdpt_bmfr_execute neither allocate nor synchronise and can be
// captured into a hipGraph.  The one exception is spelled out in bdpt.h: bdpt_execute(in = NULL) and
// bdpt_bmfr_execute allocate their optional buffers on first use when bdpt_prepare was not called, and
// refuse (BDPT_E_STATE) to do so while the stream is being captured.
// Every entry point that launches or allocates makes the context's device current first, so one host
// thread may drive contexts on several GPUs; per-context launch state (persistent grid sizes, counters
// read-back) lives in bdpt_ctx, never in statics.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <functional>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bdpt.h"
#include <new>

#include "bvh.h"
#include "kernels.h"
#include "scene_bvh.h"

using namespace bdpt;

namespace {
constexpr int kMaxStages = 64;
// path-queue counters, their fetch cursors, then the shadow sub-queue counters and cursors
// (every cursor is sharded: kNumSubQueues words, each on its own 128-byte line — atomics to one line serialise)
constexpr size_t kCountBlocks = 1;  // lengths of the valid-pixel lists
constexpr size_t kHeadBlocks = 2;   // fetch cursors of the walk kernel: pixel list x {eye, light}
constexpr size_t kLazyBlocks = kMaxLazyRounds + 2;
constexpr size_t kCursorWords = (kCountBlocks + kHeadBlocks + kLazyBlocks) * kCursorBlock + 4 * kRayCursorBlock;  // + ray count / head blocks of two classes
}

struct bdpt_ctx {
  int device = 0;
  int numCUs = 256;
  std::string err;
  // scene
  bool haveScene = false, haveCamera = false, haveSize = false;
  SceneDev S{};
  std::vector<void*> sceneAllocs;
  bdpt_bvh_info bvhInfo{};
  bdpt_camera cam{};
  bdpt_environment env{};  // bdpt_set_environment: what BDPT_PARAM_ENV_ON_MISS looks up (none: black)
  // frame
  uint32_t W = 0, H = 0, maxDepth = 0;
  // the tile: which frame rows this context renders (a contiguous band, or the stripes of one owner) and where
  // splat accumulators live (SplatLayout); rows in ascending order
  std::vector<std::pair<uint32_t, uint32_t>> rowRanges;  // [first, last) runs of rows
  uint32_t tileRows = 0;
  SplatLayout sl{1, 1, 0};
  bdpt_stripes stripes{0, 1, 0};  // stripeRows == 0: contiguous band (bdpt_resize)
  PathBuf P{};
  std::vector<void*> frameAllocs;
  unsigned long long* splat = nullptr;     // buffer in use (own or caller-provided)
  unsigned long long* ownSplat = nullptr;
  DevCounters* counters = nullptr;
  hipStream_t lastStream = nullptr;
  // stage timing
  bool timing = false;
  hipEvent_t ev[kMaxStages + 1]{};
  const char* stageNames[kMaxStages]{};
  int numStages = 0;
  bool evCreated = false;
  // kernels on the context's second stream (the splat and connection generators): their own event pairs, so that a
  // kernel's time is the kernel's and the caller's stream shows the WAIT for it as a stage of its own
  static constexpr int kMaxSideStages = 4;
  hipEvent_t sideEv[2 * kMaxSideStages]{};
  const char* sideNames[kMaxSideStages]{};
  int numSideStages = 0;
  int lazyRounds = 3;
  LaunchGrids grids{};  // persistent-grid sizes for this context's device
  int* stackOvf = nullptr;      // overflow rows of the persistent kernels' traversal stacks (kernels.h kStackLds)
  uint32_t stackOvfStride = 0;  // lanes per row: every wave a persistent grid can hold
  // channels of the built-in primary stage (bdpt_execute with in == NULL): bdpt_prepare or first use
  bdpt_gbuffer ownGb{};
  // BMFR history (bdpt_prepare or the first bdpt_bmfr_execute): [2] = ping-pong pair
  float4* bmfrPos[2] = {nullptr, nullptr};
  float4* bmfrNorm[2] = {nullptr, nullptr};
  float4* bmfrNoisy[2] = {nullptr, nullptr};
  float4* bmfrFiltered[2] = {nullptr, nullptr};
  uint8_t* bmfrAccept = nullptr;
  uint32_t* bmfrPrevPixel = nullptr;
  int bmfrRead = 0;  // which half holds the previous frame
  // the splat and NEE generators run beside the connection generator on this stream (fork/join with events; capture-safe)
  hipStream_t walkStream = nullptr;
  hipEvent_t evFork = nullptr, evJoin = nullptr, evSplat = nullptr;
  // occluder hints (kernels.hip "Occluder hints"): the primary-visibility triangle of every frame pixel, written by this
  // context's G-buffer pass and read by its light-tracing generator.  BDPT_NO_HINTS (environment) switches both kinds off.
  uint32_t* hintPix = nullptr;
  bool hints = true;
  // a context that renders only part of the frame fills the hints of ALL frame pixels when its camera changes (its
  // light-tracing rays aim anywhere); its own rows are refreshed by every G-buffer pass
  bdpt_camera hintCam{};
  bool hintCamValid = false;
};

namespace {

// [0, n) in contiguous chunks over the builder's host threads (bvhBuildThreads)
template <class F>
void hostParallelFor(size_t n, const F& f) {
  const int threads = bvhBuildThreads();
  if (threads <= 1 || n < 65536) {
    f((size_t)0, n);
    return;
  }
  WorkerScope pool;
  const size_t chunk = (n + (size_t)threads - 1) / (size_t)threads;
  for (int t = 1; t < threads; t++) {
    const size_t a = std::min(n, chunk * (size_t)t), b = std::min(n, a + chunk);
    if (a < b) pool.spawn([&f, a, b] { f(a, b); });
  }
  f((size_t)0, std::min(n, chunk));
  pool.join();
}

bool fail(bdpt_ctx* c, const std::string& m) {
  if (c) c->err = m;
  return false;
}
#define HIPCHK(ctx, expr)                                                                       \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      fail(ctx, std::string(#expr) + ": " + hipGetErrorString(e_));                             \
      return BDPT_E_HIP;                                                                        \
    }                                                                                           \
  } while (0)

template <class T>
int devAlloc(bdpt_ctx* c, std::vector<void*>& pool, T** out, size_t count) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    fail(c, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    return BDPT_E_NOMEM;
  }
  pool.push_back(p);
  *out = reinterpret_cast<T*>(p);
  return BDPT_OK;
}
template <class T>
int devUpload(bdpt_ctx* c, std::vector<void*>& pool, const T** out, const T* host, size_t count) {
  T* d = nullptr;
  int rc = devAlloc(c, pool, &d, count);
  if (rc) return rc;
  if (count) {  // (large arrays through pinned staging: bvhUploadStaged)
    std::string e;
    if (!bvhUploadStaged(d, host, count * sizeof(T), e)) {
      fail(c, e);
      return BDPT_E_HIP;
    }
  }
  *out = d;
  return BDPT_OK;
}
void freePool(std::vector<void*>& pool) {
  for (void* p : pool) (void)hipFree(p);
  pool.clear();
}

// Every entry point that launches or allocates starts here: the context's device becomes current, so one
// host thread can hold contexts on several GPUs.
#define ENTER(ctx) HIPCHK(ctx, hipSetDevice((ctx)->device))

bool streamIsCapturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) return false;
  return cs != hipStreamCaptureStatusNone;
}

// the six channels of the built-in primary stage; all or nothing
int allocOwnGbuffer(bdpt_ctx* c) {
  if (c->ownGb.worldPosition) return BDPT_OK;
  const size_t n = (size_t)c->W * c->H;
  const size_t mark = c->frameAllocs.size();
  bdpt_gbuffer g{};
  int rc;
  if ((rc = devAlloc(c, c->frameAllocs, &g.worldPosition, n * 4)) || (rc = devAlloc(c, c->frameAllocs, &g.worldNormal, n * 4)) ||
      (rc = devAlloc(c, c->frameAllocs, &g.materialDiffuse, n * 4)) || (rc = devAlloc(c, c->frameAllocs, &g.materialSpecRough, n * 4)) ||
      (rc = devAlloc(c, c->frameAllocs, &g.materialExtraParams, n * 4)) || (rc = devAlloc(c, c->frameAllocs, &g.emissive, n * 4))) {
    while (c->frameAllocs.size() > mark) {
      (void)hipFree(c->frameAllocs.back());
      c->frameAllocs.pop_back();
    }
    return rc;
  }
  c->ownGb = g;
  return BDPT_OK;
}

int allocBmfrHistory(bdpt_ctx* c) {
  if (c->bmfrAccept) return BDPT_OK;
  const size_t n = (size_t)c->W * c->H;
  const size_t mark = c->frameAllocs.size();
  float4 *pos[2]{}, *norm[2]{}, *noisy[2]{}, *filt[2]{};
  uint8_t* accept = nullptr;
  uint32_t* prevPixel = nullptr;
  int rc = BDPT_OK;
  for (int k = 0; k < 2 && !rc; k++) {
    if ((rc = devAlloc(c, c->frameAllocs, &pos[k], n)) || (rc = devAlloc(c, c->frameAllocs, &norm[k], n)) ||
        (rc = devAlloc(c, c->frameAllocs, &noisy[k], n)) || (rc = devAlloc(c, c->frameAllocs, &filt[k], n)))
      break;
  }
  if (!rc) rc = devAlloc(c, c->frameAllocs, &accept, n);
  if (!rc) rc = devAlloc(c, c->frameAllocs, &prevPixel, n);
  if (rc) {
    while (c->frameAllocs.size() > mark) {
      (void)hipFree(c->frameAllocs.back());
      c->frameAllocs.pop_back();
    }
    return rc;
  }
  for (int k = 0; k < 2; k++) {
    c->bmfrPos[k] = pos[k];
    c->bmfrNorm[k] = norm[k];
    c->bmfrNoisy[k] = noisy[k];
    c->bmfrFiltered[k] = filt[k];
  }
  c->bmfrAccept = accept;
  c->bmfrPrevPixel = prevPixel;
  return bdpt_bmfr_reset(c);
}

void stageMark(bdpt_ctx* c, hipStream_t st, const char* name) {
  if (!c->timing || c->numStages >= kMaxStages) return;
  c->stageNames[c->numStages] = name;
  c->numStages++;
  (void)hipEventRecord(c->ev[c->numStages], st);
}
// bracket of a kernel on the second stream: sideBegin before its launch, sideEnd after
void sideBegin(bdpt_ctx* c, hipStream_t side, const char* name) {
  if (!c->timing || c->numSideStages >= bdpt_ctx::kMaxSideStages) return;
  c->sideNames[c->numSideStages] = name;
  (void)hipEventRecord(c->sideEv[2 * c->numSideStages], side);
}
void sideEnd(bdpt_ctx* c, hipStream_t side) {
  if (!c->timing || c->numSideStages >= bdpt_ctx::kMaxSideStages) return;
  (void)hipEventRecord(c->sideEv[2 * c->numSideStages + 1], side);
  c->numSideStages++;
}

}  // namespace

extern "C" {

int bdpt_create(int device_ordinal, bdpt_ctx** out_ctx) {
  if (!out_ctx) return BDPT_E_INVALID;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n) return BDPT_E_HIP;
  if (hipSetDevice(device_ordinal) != hipSuccess) return BDPT_E_HIP;
  bdpt_ctx* c = new bdpt_ctx();
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0)
    c->numCUs = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->walkStream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->evJoin, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->evSplat, hipEventDisableTiming) != hipSuccess) {
    bdpt_destroy(c);
    return BDPT_E_HIP;
  }
  // traversal-stack overflow rows for the largest persistent grid (kMaxPersistentPerCU waves per CU)
  c->stackOvfStride = (uint32_t)c->numCUs * kMaxPersistentPerCU * kWave;
  if (kStackOvfRows > 0 &&
      hipMalloc(reinterpret_cast<void**>(&c->stackOvf), (size_t)kStackOvfRows * c->stackOvfStride * sizeof(int)) != hipSuccess) {
    bdpt_destroy(c);
    return BDPT_E_NOMEM;
  }
  bvhPrewarmStaging(device_ordinal);  // (the pinned staging buffers bdpt_set_scene's uploads go through)
  *out_ctx = c;
  return BDPT_OK;
}

void bdpt_destroy(bdpt_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->stackOvf) (void)hipFree(c->stackOvf);
  freePool(c->sceneAllocs);
  freePool(c->frameAllocs);
  if (c->evCreated)
    for (int i = 0; i <= kMaxStages; i++) (void)hipEventDestroy(c->ev[i]);
  if (c->evCreated)
    for (hipEvent_t e : c->sideEv) (void)hipEventDestroy(e);
  if (c->evFork) (void)hipEventDestroy(c->evFork);
  if (c->evJoin) (void)hipEventDestroy(c->evJoin);
  if (c->evSplat) (void)hipEventDestroy(c->evSplat);
  if (c->walkStream) (void)hipStreamDestroy(c->walkStream);
  delete c;
}

const char* bdpt_last_error(const bdpt_ctx* c) { return c ? c->err.c_str() : "null context"; }

static int setSceneImpl(bdpt_ctx* c, const bdpt_scene_desc* d) {
  if (!c || !d) return BDPT_E_INVALID;
  if (!d->positions || !d->normals || !d->indices || !d->triMaterial || !d->materials || !d->numMaterials) {
    fail(c, "scene: positions, normals, indices, triMaterial and materials are required");
    return BDPT_E_INVALID;
  }
  if (d->numLights == 0 || !d->lights) {
    fail(c, "scene: at least one light is required (SceneLoaderWrapper adds a directional light when a file has none)");
    return BDPT_E_INVALID;
  }
  if (d->numLights > BDPT_MAX_LIGHTS) {
    fail(c, "scene: more than BDPT_MAX_LIGHTS lights");
    return BDPT_E_LIMIT;
  }
  if (d->numTriangles >= (1u << 28)) {
    fail(c, "scene: triangle count exceeds the 2^28 leaf-reference limit");
    return BDPT_E_LIMIT;
  }
  {
    std::atomic<int> bad{0};  // 1 = material id, 2 = vertex index, 3 = vertex position
    hostParallelFor(d->numTriangles, [&](size_t t0, size_t t1) {
      for (size_t t = t0; t < t1; t++) {
        if (d->triMaterial[t] >= d->numMaterials) bad.store(1);
        for (int k = 0; k < 3; k++) {
          const uint32_t vi = d->indices[t * 3 + (size_t)k];
          if (vi >= d->numVertices) {
            bad.store(2);
            continue;
          }
          // (a box with a NaN or an infinity in it has no place in a tree built by comparing and sorting boxes)
          const float* p = d->positions + (size_t)vi * 3;
          if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])) && bad.load() == 0) bad.store(3);
        }
      }
    });
    if (bad.load() == 3) {
      fail(c, "scene: a triangle has a vertex position that is not finite");
      return BDPT_E_INVALID;
    }
    if (bad.load() == 1) {
      fail(c, "scene: triMaterial out of range");
      return BDPT_E_INVALID;
    }
    if (bad.load() == 2) {
      fail(c, "scene: vertex index out of range");
      return BDPT_E_INVALID;
    }
  }
  for (uint32_t m = 0; m < d->numMaterials; m++) {
    const bdpt_material& mm = d->materials[m];
    const int ids[4] = {mm.texBaseColor, mm.texSpecular, mm.texEmissive, mm.texNormal};
    for (int id : ids)
      if (id < -1 || id >= (int)d->numTextures) {
        fail(c, "scene: material texture index out of range");
        return BDPT_E_INVALID;
      }
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  freePool(c->sceneAllocs);
  c->haveScene = false;
  c->S = SceneDev{};
  c->S.stackOvf = c->stackOvf;
  c->S.stackOvfStride = c->stackOvfStride;

  // BDPT_BUILD_VERBOSE: where the set-up time goes (stderr), as bvh_build.cpp's own laps
  const bool verbose = std::getenv("BDPT_BUILD_VERBOSE") != nullptr;
  auto tLap = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[set_scene] %-18s %.3f s\n", what, std::chrono::duration<double>(t - tLap).count());
    tLap = t;
  };
  lap("validate");
  // Beside the build, on a thread of its own: the per-primitive shading records (3 x (position, normal, uv) + material id,
  // 112 B) and the uploads that depend on nothing the build produces — while the build's device stages run the host is
  // idle, and while its host stages run the copy engine is.
  struct SideUploads {
    std::vector<void*> pool;
    const float* shade = nullptr;
    const uint32_t* indices = nullptr;
    const float* bitangents = nullptr;
    hipError_t err = hipSuccess;
    std::string what;  // the staged upload's own message, when that is what failed
  };
  // (BDPT_SET_SCENE_SERIAL: measurement knob — the same work at the point where its results are needed, on this thread)
  std::future<SideUploads> side = std::async(std::getenv("BDPT_SET_SCENE_SERIAL") ? std::launch::deferred : std::launch::async, [c, d]() {
    SideUploads u;
    BigVec<float> shade((size_t)d->numTriangles * kShadeRecF4 * 4);  // (sized, not zeroed: the loop writes every float)
    hostParallelFor(d->numTriangles, [&](size_t t0, size_t t1) {
      for (size_t t = t0; t < t1; t++) {
        float* r = &shade[t * kShadeRecF4 * 4];
        for (int k = 0; k < 3; k++) {
          const uint32_t vi = d->indices[t * 3 + (size_t)k];
          const float* p = d->positions + (size_t)vi * 3;
          const float* nn = d->normals + (size_t)vi * 3;
          float* q = r + k * 8;
          q[0] = p[0];
          q[1] = p[1];
          q[2] = p[2];
          q[3] = nn[0];
          q[4] = nn[1];
          q[5] = nn[2];
          q[6] = d->texcoords ? d->texcoords[(size_t)vi * 3] : 0.0f;
          q[7] = d->texcoords ? d->texcoords[(size_t)vi * 3 + 1] : 0.0f;
        }
        uint32_t mid = d->triMaterial[t];
        std::memcpy(r + 24, &mid, 4);
        for (int k = 25; k < kShadeRecF4 * 4; k++) r[k] = 0.0f;
      }
    });
    auto up = [&](auto** dst, const auto* host, size_t count) {
      if (u.err != hipSuccess) return;
      void* q = nullptr;
      u.err = hipMalloc(&q, std::max<size_t>(count * sizeof(**dst), 16));
      if (u.err != hipSuccess) return;
      u.pool.push_back(q);
      if (count) {
        std::string e;
        if (!bvhUploadStaged(q, host, count * sizeof(**dst), e)) {
          u.err = hipErrorUnknown;
          u.what = e;
        }
      }
      *dst = static_cast<std::remove_reference_t<decltype(*dst)>>(q);
    };
    u.err = hipSetDevice(c->device);
    up(&u.shade, shade.data(), shade.size());
    up(&u.indices, d->indices, (size_t)d->numTriangles * 3);
    if (d->bitangents) up(&u.bitangents, d->bitangents, (size_t)d->numVertices * 3);
    return u;
  });
  struct SideJoin {  // whatever way this function is left, the thread is joined and what it allocated has an owner
    std::future<SideUploads>& f;
    ~SideJoin() {
      if (!f.valid()) return;
      try {
        SideUploads u = f.get();
        for (void* q : u.pool) (void)hipFree(q);
      } catch (...) {
      }
    }
  } sideJoin{side};
  // traversal flags, alpha classification, spatial pre-splitting and the tree itself: scene_bvh.cpp
  SceneBvh sb;
  try {
    // the binary tree is built, and the records are quantised and packed, on this context's device (bvh_device.hip):
    // the records the host code produces, bit for bit
    std::string treeError;
    struct Build {
      BvhDeviceBuild* b;
      ~Build() { bvhDeviceBuildEnd(b); }
    } build{bvhDeviceBuildBegin(c->device, std::getenv("BDPT_HOST_COLLAPSE") == nullptr)};  // (measurement knob: the collapse on the host)
    buildSceneBvh(d, 0, -1.0f, -1.0f, std::getenv("BDPT_NO_ALPHA_CLASSIFY") == nullptr, sb, buildBinaryTreeOnDevice, build.b, &treeError, packOnDevice, makeReferencesOnDevice,
                  std::getenv("BDPT_HOST_COLLAPSE") == nullptr, std::getenv("BDPT_HOST_PRIORITIES") == nullptr);
    if (sb.bvh.deviceRecs) c->sceneAllocs.push_back(sb.bvh.deviceRecs);  // (the context's from here on)
    if (!treeError.empty()) {
      fail(c, "scene: " + treeError);
      return treeError.find("2^31") != std::string::npos ? BDPT_E_LIMIT : (treeError.find("out of device memory") != std::string::npos ? BDPT_E_NOMEM : BDPT_E_HIP);
    }
  } catch (const std::bad_alloc&) {
    fail(c, "scene: out of host memory while building the acceleration structure");
    return BDPT_E_NOMEM;
  }
  lap("buildSceneBvh");
  Bvh& bvh = sb.bvh;
  const std::vector<uint32_t>& alphaTris = sb.alphaTris;
  if (bvh.maxStack > (uint32_t)kBvhMaxStack) {
    fail(c, "bvh needs a deeper traversal stack than the device provides");
    return BDPT_E_LIMIT;
  }
  if (!bvh.deviceRecs && !bvh.recs.empty()) {  // nothing to build a tree over (no triangle can be hit): the host's one empty node
    const BvhRec* dRecs = nullptr;
    if (int rc0 = devUpload(c, c->sceneAllocs, &dRecs, bvh.recs.data(), bvh.recs.size())) return rc0;
    bvh.deviceRecs = const_cast<BvhRec*>(dRecs);
    bvh.deviceNumRecs = bvh.recs.size();
  }
  if (!bvh.deviceRecs) {
    fail(c, "scene: the acceleration structure has no records");
    return BDPT_E_HIP;
  }
  c->bvhInfo.numNodes = bvh.numNodes;
  c->bvhInfo.numTriangles = d->numTriangles;
  c->bvhInfo.maxDepth = bvh.maxDepth;
  c->bvhInfo.nodeBytes = sizeof(BvhRec);
  c->bvhInfo.triBytes = sizeof(BvhTri);
  c->bvhInfo.sahCost = bvh.sahCost;
  c->bvhInfo.maxStack = bvh.maxStack;
  c->bvhInfo.numReferences = bvh.numRefs;
  c->bvhInfo.numDropped = bvh.numDropped;
  c->bvhInfo.numAlphaMode = sb.numAlphaMode;
  c->bvhInfo.numAlwaysPass = sb.numAlwaysPass;

  int rc;
  {
    SideUploads u = side.get();  // (a std::bad_alloc of that thread is rethrown here: bdpt_set_scene's catch)
    c->sceneAllocs.insert(c->sceneAllocs.end(), u.pool.begin(), u.pool.end());
    if (u.err != hipSuccess) {
      fail(c, std::string("scene upload: ") + (u.what.empty() ? std::string(hipGetErrorString(u.err)) : u.what));
      return u.err == hipErrorOutOfMemory ? BDPT_E_NOMEM : BDPT_E_HIP;
    }
    c->S.shade = reinterpret_cast<const float4*>(u.shade);
    c->S.indices = u.indices;
    if (d->bitangents) {
      c->S.bitangents = u.bitangents;
      c->S.hasBitangents = 1;
    }
  }
  lap("shading records + uploads (joined)");
  c->S.recs = reinterpret_cast<const uint4*>(bvh.deviceRecs);
  c->S.numRecs = (uint32_t)bvh.deviceNumRecs;
  if ((rc = devUpload(c, c->sceneAllocs, &c->S.materials, d->materials, d->numMaterials))) return rc;
  std::vector<TexDev> texs(d->numTextures);
  for (uint32_t i = 0; i < d->numTextures; i++) {
    const bdpt_texture& t = d->textures[i];
    if (!t.rgba8 || !t.width || !t.height) {
      fail(c, "scene: empty texture");
      return BDPT_E_INVALID;
    }
    const uint8_t* px;
    if ((rc = devUpload(c, c->sceneAllocs, &px, t.rgba8, (size_t)t.width * t.height * 4))) return rc;
    texs[i] = TexDev{px, t.width, t.height, t.srgb, 0};
  }
  if ((rc = devUpload(c, c->sceneAllocs, &c->S.textures, texs.data(), texs.size()))) return rc;
  {
    std::vector<TexDev> matTex((size_t)d->numMaterials * 4, TexDev{nullptr, 0, 0, 0, 0});
    for (uint32_t mi = 0; mi < d->numMaterials; mi++) {
      const bdpt_material& mm = d->materials[mi];
      const int ids[4] = {mm.texBaseColor, mm.texSpecular, mm.texEmissive, mm.texNormal};
      for (int k = 0; k < 4; k++)
        if (ids[k] >= 0) matTex[(size_t)mi * 4 + k] = texs[(size_t)ids[k]];
    }
    if ((rc = devUpload(c, c->sceneAllocs, &c->S.matTex, matTex.data(), matTex.size()))) return rc;
  }
  {
    // the alpha-test records of the non-opaque triangles, made on the device from the shading records and the material
    // tables already there (kernels.hip alpha_recs_kernel): only the list of those triangles crosses the bus
    float4* dAlpha = nullptr;
    if ((rc = devAlloc(c, c->sceneAllocs, &dAlpha, std::max<size_t>(alphaTris.size(), 1) * 4))) return rc;
    if (alphaTris.empty()) {
      HIPCHK(c, hipMemset(dAlpha, 0, 4 * sizeof(float4)));
    } else {
      const uint32_t* dList = nullptr;
      std::vector<void*> scratch;
      if ((rc = devUpload(c, scratch, &dList, alphaTris.data(), alphaTris.size()))) {
        freePool(scratch);
        return rc;
      }
      launchAlphaRecs(c->S, dList, (uint32_t)alphaTris.size(), dAlpha, nullptr);
      const hipError_t e = hipDeviceSynchronize();
      freePool(scratch);
      HIPCHK(c, e);
    }
    c->S.alphaRecs = dAlpha;
  }
  lap("indices, textures, alpha");
  SceneConst sc;
  std::memset(&sc, 0, sizeof(sc));
  std::memcpy(sc.lights, d->lights, sizeof(bdpt_light) * d->numLights);
  for (int i = 0; i < 256; i++) {
    double cc = (double)i / 255.0;
    double l = (cc <= 0.04045) ? cc / 12.92 : std::pow((cc + 0.055) / 1.055, 2.4);
    sc.srgbLut[i] = (float)l;
  }
  if ((rc = devUpload(c, c->sceneAllocs, &c->S.sc, &sc, 1))) return rc;
  c->S.numLights = d->numLights;
  // Occluder hints of next-event rays: one cube map of nearest triangles per light (directional lights: empty), traced
  // here once; the primary-visibility hints of an earlier scene index records that are gone.
  c->hints = std::getenv("BDPT_NO_HINTS") == nullptr;
  if (c->hints) {
    uint32_t res = 512;
    if (const char* e = std::getenv("BDPT_LIGHT_MAP_RES")) res = (uint32_t)std::max(0, std::min(2048, std::atoi(e)));
    if (res) {
      uint32_t* maps = nullptr;
      if ((rc = devAlloc(c, c->sceneAllocs, &maps, (size_t)6 * res * res * d->numLights))) return rc;
      launchLightMaps(c->S, maps, res, nullptr);
      HIPCHK(c, hipDeviceSynchronize());
      c->S.lightMap = maps;
      c->S.lightMapRes = res;
    }
  }
  if (c->hintPix) HIPCHK(c, hipMemset(c->hintPix, 0xFF, (size_t)c->W * c->H * sizeof(uint32_t)));
  c->hintCamValid = false;
  lap("light maps");
  c->haveScene = true;
  return BDPT_OK;
}
// Nothing thrown while the scene is copied and its acceleration structure built — on this thread or on a builder
// worker (bvh.h WorkerScope) — crosses the C boundary: the large allocations of a 10 M-triangle scene fail as BDPT_E_NOMEM.
int bdpt_set_scene(bdpt_ctx* c, const bdpt_scene_desc* d) {
  try {
    return setSceneImpl(c, d);
  } catch (const std::bad_alloc&) {
    fail(c, "scene: out of host memory while building the acceleration structure");
    return BDPT_E_NOMEM;
  } catch (const std::exception& e) {
    fail(c, std::string("scene: ") + e.what());
    return BDPT_E_INVALID;
  }
}


int bdpt_get_bvh_info(const bdpt_ctx* c, bdpt_bvh_info* out) {
  if (!c || !out) return BDPT_E_INVALID;
  if (!c->haveScene) return BDPT_E_STATE;
  *out = c->bvhInfo;
  return BDPT_OK;
}

int bdpt_set_camera(bdpt_ctx* c, const bdpt_camera* cam) {
  if (!c || !cam) return BDPT_E_INVALID;
  c->cam = *cam;
  c->haveCamera = true;
  return BDPT_OK;
}

int bdpt_set_environment(bdpt_ctx* c, const bdpt_environment* env) {
  if (!c) return BDPT_E_INVALID;
  if (env && env->envMap && (!env->width || !env->height)) {
    fail(c, "environment: a map needs a width and a height");
    return BDPT_E_INVALID;
  }
  c->env = env ? *env : bdpt_environment{};
  if (!c->env.envMap) c->env.width = c->env.height = 0;
  return BDPT_OK;
}

namespace {
int resizeRows(bdpt_ctx* c, uint32_t width, uint32_t height, uint32_t maxDepth);
}

int bdpt_resize(bdpt_ctx* c, uint32_t width, uint32_t height, bdpt_tile tile, uint32_t maxDepth) {
  if (!c) return BDPT_E_INVALID;
  if (!width || !height || tile.y0 > tile.y1 || tile.y1 > height) {
    fail(c, "resize: bad frame or tile");
    return BDPT_E_INVALID;
  }
  c->rowRanges.clear();
  if (tile.y1 > tile.y0) c->rowRanges.push_back({tile.y0, tile.y1});
  c->tileRows = tile.y1 - tile.y0;
  c->stripes = bdpt_stripes{0, 1, 0};
  c->sl = SplatLayout{1, 1, height};  // frame order
  return resizeRows(c, width, height, maxDepth);
}

// Interleaved stripes: rows are dealt to `numOwners` contexts in stripes of `stripeRows`; this one
// renders the stripes s with s % numOwners == owner.  The splat buffer becomes owner-major (bdpt_get_tile_info).
int bdpt_resize_stripes(bdpt_ctx* c, uint32_t width, uint32_t height, bdpt_stripes st, uint32_t maxDepth) {
  if (!c) return BDPT_E_INVALID;
  if (!width || !height || !st.stripeRows || !st.numOwners || st.owner >= st.numOwners) {
    fail(c, "resize_stripes: bad frame or stripe description");
    return BDPT_E_INVALID;
  }
  c->rowRanges.clear();
  c->tileRows = 0;
  const uint32_t numStripes = (height + st.stripeRows - 1) / st.stripeRows;
  for (uint32_t s = st.owner; s < numStripes; s += st.numOwners) {
    const uint32_t a = s * st.stripeRows, b = std::min(height, a + st.stripeRows);
    c->rowRanges.push_back({a, b});
    c->tileRows += b - a;
  }
  c->stripes = st;
  c->sl = SplatLayout{st.stripeRows, st.numOwners, ((numStripes + st.numOwners - 1) / st.numOwners) * st.stripeRows};
  return resizeRows(c, width, height, maxDepth);
}

uint32_t bdpt_stripe_rows(uint32_t height, uint32_t numOwners) {
  const uint32_t per = height / std::max(1u, numOwners * 4u);
  return std::max(1u, std::min(8u, per));
}
int bdpt_get_tile_info(const bdpt_ctx* c, bdpt_tile_info* out) {
  if (!c || !out) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  out->numRows = c->tileRows;
  out->numPixels = c->P.Np;
  out->chunkRows = c->sl.chunkRows;
  out->numRowRanges = (uint32_t)c->rowRanges.size();
  out->splatU64 = (uint64_t)c->sl.owners * c->sl.chunkRows * c->W * 4;
  out->chunkU64 = (uint64_t)c->sl.chunkRows * c->W * 4;
  return BDPT_OK;
}

int bdpt_tile_row_ranges(const bdpt_ctx* c, uint32_t* out_first_last, uint32_t cap) {
  if (!c || !out_first_last) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  const uint32_t n = std::min<uint32_t>(cap, (uint32_t)c->rowRanges.size());
  for (uint32_t i = 0; i < n; i++) {
    out_first_last[2 * i] = c->rowRanges[i].first;
    out_first_last[2 * i + 1] = c->rowRanges[i].second;
  }
  return (int)n;
}

namespace {
int resizeRows(bdpt_ctx* c, uint32_t width, uint32_t height, uint32_t maxDepth) {
  if (maxDepth > BDPT_MAX_DEPTH) {
    fail(c, "resize: maxDepth exceeds BDPT_MAX_DEPTH");
    return BDPT_E_LIMIT;
  }
  if ((uint64_t)width * height >= (1ull << 32) || (uint64_t)c->sl.owners * c->sl.chunkRows * width >= (1ull << 32)) {
    fail(c, "resize: frame too large");
    return BDPT_E_LIMIT;
  }
  if ((uint64_t)c->tileRows * width >= (1ull << 24)) {
    fail(c, "resize: a tile holds at most 2^24 - 1 pixels (path ids pack the pixel in 24 bits); render in smaller tiles");
    return BDPT_E_LIMIT;
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  freePool(c->frameAllocs);
  for (int k = 0; k < 2; k++) c->bmfrPos[k] = c->bmfrNorm[k] = c->bmfrNoisy[k] = c->bmfrFiltered[k] = nullptr;
  c->ownGb = bdpt_gbuffer{};
  c->bmfrAccept = nullptr;  // history goes with the frame (BlockwiseMultiOrderFeatureRegression::resize)
  c->bmfrPrevPixel = nullptr;
  c->haveSize = false;
  c->W = width;
  c->H = height;
  c->maxDepth = maxDepth;
  PathBuf P{};
  P.Np = c->tileRows * width;
  P.D1 = std::max<uint32_t>(maxDepth, 1) + 1;
  const size_t np = std::max<uint32_t>(P.Np, 1);
  int rc;
  {
    std::vector<uint32_t> pix;
    pix.reserve(np);
    for (const auto& rr : c->rowRanges)
      for (uint32_t y = rr.first; y < rr.second; y++)
        for (uint32_t x = 0; x < width; x++) pix.push_back(y * width + x);
    if ((rc = devUpload(c, c->frameAllocs, &P.pix, pix.data(), pix.size()))) return rc;
  }
  if ((rc = devAlloc(c, c->frameAllocs, &P.v, (size_t)2 * P.D1 * NF4 * 4 * np))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.rayDir, (size_t)6 * np))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.seedE, np))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.seedL, np))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.eyeLast, np))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.lightLast, np))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.lightReal, np))) return rc;
  // a path queue = kNumSubQueues lists; workgroup b appends to list b % kNumSubQueues
  // (+ one workgroup's worth of slack)
  P.pathSubCap = (uint32_t)((((np + kWave - 1) / kWave + kNumSubQueues - 1) / kNumSubQueues + 1) * kWave);
  const size_t qcap = (size_t)P.pathSubCap * kNumSubQueues;
  for (int q = 0; q < 3; q++)
    if ((rc = devAlloc(c, c->frameAllocs, &P.queue[q], qcap))) return rc;
  if ((rc = devAlloc(c, c->frameAllocs, &P.qcount, (size_t)kCursorWords))) return rc;
  P.qhead = P.qcount + kCountBlocks * kCursorBlock;
  P.lazyCount = P.qhead + kHeadBlocks * kCursorBlock;
  P.rayCount = P.lazyCount + kLazyBlocks * kCursorBlock;
  P.rayHead = P.rayCount + 2 * kRayCursorBlock;
  {
    // one shadow ray per NEE term, per splat term and per defined connection pair, at most
    const uint32_t D = std::max<uint32_t>(maxDepth, 1);
    const uint64_t slots = (uint64_t)2 * D + numConnectPairs(D);
    // workgroup b appends to sub-queue b % kNumSubQueues: size each for the workgroups it serves
    const uint64_t blocks = (np + kWave - 1) / kWave;
    // Producer workgroup b appends to ray sub-queue b % kNumRaySubQueues.  Generators run G lanes per pixel (8, or 16
    // when the context is sized for depth > 8: G x queueGrid workgroups of 64 / G pixels), lazy_gen one lane per pixel:
    // size every sub-queue for the workgroups it can serve under either launch shape.
    const uint64_t queueGridBlocks = (uint64_t)(P.pathSubCap / kWave) * kNumSubQueues;
    const uint64_t pairs = std::max<uint32_t>(numConnectPairs(D), 1);
    uint64_t subCapTerms = 0, subCapPairs = ((queueGridBlocks + kNumRaySubQueues - 1) / kNumRaySubQueues) * kWave * pairs;
    {
      const uint64_t G = D > 8 ? 16 : 8;
      const uint64_t perSub = (queueGridBlocks * G + kNumRaySubQueues - 1) / kNumRaySubQueues;  // workgroups per sub-queue
      subCapTerms = std::max<uint64_t>(subCapTerms, perSub * kWave * 2);                           // one NEE + one splat ray per lane
      subCapPairs = std::max<uint64_t>(subCapPairs, perSub * (kWave / G) * pairs);
    }
    (void)blocks;
    const uint64_t cap = (subCapTerms + subCapPairs) * kNumRaySubQueues;
    if (cap >= (1ull << 32) - 1) {
      fail(c, "resize: shadow-ray queue would exceed 2^32 entries; render in smaller tiles");
      return BDPT_E_LIMIT;
    }
    P.raySubCap[RAY_TERMS] = (uint32_t)subCapTerms;
    P.raySubCap[RAY_PAIRS] = (uint32_t)subCapPairs;
    P.rayBase[RAY_TERMS] = 0;
    P.rayBase[RAY_PAIRS] = (uint32_t)(subCapTerms * kNumRaySubQueues);
    P.rayCap = (uint32_t)cap;
    if ((rc = devAlloc(c, c->frameAllocs, &P.rayQ, (size_t)7 * cap))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.rayContrib, (size_t)3 * cap))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.rayVis, (size_t)cap))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.slotRay, (size_t)slots * np))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.splatPix, (size_t)D * np))) return rc;
    // Lazy rounds: each costs three small launches, so small tiles (multi-GPU bands) take fewer, larger ones.
    // Lazy rounds: each costs three small launches and cannot finish faster than its slowest ray, so there are few:
    // two or three rounds of kLazyBatchDiv-th shares, the last of which takes every candidate that is left.
    c->lazyRounds = np >= (1u << 18) ? 3 : 2;
    if (const char* e = std::getenv("BDPT_LAZY_ROUNDS")) {  // measurement knob (tools/prof_tile.sh): 1 .. kMaxLazyRounds
      const int v = std::atoi(e);
      if (v >= 1 && v <= kMaxLazyRounds) c->lazyRounds = v;
    }
    // the largest batch any round can ask for: the last round of the front-loaded schedule takes what is left
    const uint32_t batch = numConnectPairs(D);
    if ((rc = devAlloc(c, c->frameAllocs, &P.misE, (size_t)(D + 1) * np))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.misL, (size_t)(D + 1) * np))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.lazyCursor, np))) return rc;
    if ((rc = devAlloc(c, c->frameAllocs, &P.lazyRay, (size_t)std::max<uint32_t>(batch, 1) * np))) return rc;
  }
  c->hintPix = nullptr;
  c->hintCamValid = false;
  if (std::getenv("BDPT_NO_HINTS") == nullptr) {
    if ((rc = devAlloc(c, c->frameAllocs, &c->hintPix, (size_t)width * height))) return rc;
    HIPCHK(c, hipMemset(c->hintPix, 0xFF, (size_t)width * height * sizeof(uint32_t)));
  }
  const size_t splatU64 = (size_t)c->sl.owners * c->sl.chunkRows * width * 4;
  if ((rc = devAlloc(c, c->frameAllocs, &c->ownSplat, splatU64))) return rc;
  c->splat = c->ownSplat;
  if ((rc = devAlloc(c, c->frameAllocs, &c->counters, 1))) return rc;
  HIPCHK(c, hipMemset(c->splat, 0, splatU64 * sizeof(unsigned long long)));
  HIPCHK(c, hipMemset(c->counters, 0, sizeof(DevCounters)));
  c->P = P;
  if (!c->evCreated) {
    for (int i = 0; i <= kMaxStages; i++) HIPCHK(c, hipEventCreate(&c->ev[i]));
    for (hipEvent_t& e : c->sideEv) HIPCHK(c, hipEventCreate(&e));
    c->evCreated = true;
  }
  c->haveSize = true;
  return BDPT_OK;
}
}  // namespace

int bdpt_gbuffer_execute(bdpt_ctx* c, const bdpt_gbuffer_params* gp, const bdpt_gbuffer* out, void* stream) {
  if (!c || !gp || !out) return BDPT_E_INVALID;
  if (!c->haveScene || !c->haveCamera || !c->haveSize) {
    fail(c, "gbuffer_execute: scene, camera and size must be set first");
    return BDPT_E_STATE;
  }
  if (!out->worldPosition || !out->worldNormal || !out->materialDiffuse || !out->materialSpecRough ||
      !out->materialExtraParams || !out->emissive) {
    fail(c, "gbuffer_execute: all six channels are required");
    return BDPT_E_INVALID;
  }
  ENTER(c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GBufferDev G{};
  G.cam = c->cam;
  G.gp = *gp;
  G.W = c->W;
  G.H = c->H;
  G.Np = c->P.Np;
  G.pix = c->P.pix;
  G.gb = *out;
  G.counters = nullptr;  // primary rays are tallied analytically by bdpt_get_counters (one per tile pixel)
  G.hintPix = c->hintPix;
  if (c->hintPix && c->tileRows != c->H && (!c->hintCamValid || std::memcmp(&c->hintCam, &c->cam, sizeof(bdpt_camera)) != 0)) {
    GBufferDev A = G;
    A.Np = c->W * c->H;
    A.pix = nullptr;
    launchHintFill(c->S, A, st);
    c->hintCam = c->cam;
    c->hintCamValid = true;
  }
  launchGBuffer(c->S, G, st);
  HIPCHK(c, hipGetLastError());
  c->lastStream = st;
  return BDPT_OK;
}

namespace {
// argument checks shared by bdpt_execute and bdpt_execute_tail, and the per-frame constants the kernels take
int frameSetup(bdpt_ctx* c, const bdpt_params* p, const bdpt_gbuffer* in, float* out, hipStream_t st, FrameDev& F) {
  if (!c || !p || !out) return BDPT_E_INVALID;
  ENTER(c);
  if (!in) {  // the context's own channels: the primary stage runs inside bdpt_execute
    if (!c->haveSize) {
      fail(c, "execute: scene, camera and size must be set first");
      return BDPT_E_STATE;
    }
    if (!c->ownGb.worldPosition) {  // bdpt_prepare(BDPT_PREPARE_PRIMARY) was not called: allocate now, unless capturing
      if (streamIsCapturing(st)) {
        fail(c, "execute: the built-in primary stage needs bdpt_prepare(BDPT_PREPARE_PRIMARY) before stream capture");
        return BDPT_E_STATE;
      }
      if (int rc = allocOwnGbuffer(c)) return rc;
    }
    in = &c->ownGb;
  }
  if (!c->haveScene || !c->haveCamera || !c->haveSize) {
    fail(c, "execute: scene, camera and size must be set first");
    return BDPT_E_STATE;
  }
  if (p->maxDepth > c->maxDepth) {
    fail(c, "execute: params.maxDepth exceeds the depth given to bdpt_resize");
    return BDPT_E_LIMIT;
  }
  if (p->matIndex > 1) {
    fail(c, "execute: matIndex must be 0 (GGX) or 1 (Lambertian)");
    return BDPT_E_INVALID;
  }
  if (!in->worldPosition || !in->worldNormal || !in->materialDiffuse || !in->materialSpecRough || !in->emissive) {
    fail(c, "execute: G-buffer channels missing");
    return BDPT_E_INVALID;
  }
  F = FrameDev{};
  F.cam = c->cam;
  F.p = *p;
  F.W = c->W;
  F.H = c->H;
  F.sl = c->sl;
  F.out = out;
  F.splat = c->splat;
  F.gb = *in;
  F.counters = c->counters;  // ray tallies are always on; node/triangle visits need BDPT_PARAM_COUNTERS
  F.envMap = c->env.envMap;
  F.envW = c->env.width;
  F.envH = c->env.height;
  for (int k = 0; k < 3; k++) F.envColor[k] = c->env.color[k];
  F.hintPix = c->hintPix;
  return BDPT_OK;
}

// Connection pairs whose contribution is exactly zero, for the pixels no visible connection has saturated
// yet (DESIGN.md "Lazy connection rounds"), then the splat fold-in unless the caller defers it.
int connectionTail(bdpt_ctx* c, const FrameDev& F, hipStream_t st) {
  const PathBuf& P = c->P;
  const bdpt_params* p = &F.p;
  const int D = (int)p->maxDepth;
  if (!(p->flags & BDPT_PARAM_NO_CONNECT) && D >= 2) {
    const int nPairs = (int)numConnectPairs((uint32_t)D);
    // Schedule: rounds of ceil(pairs / kLazyBatchDiv) candidates; the last round takes everything that is left.  A launch
    // cannot finish faster than its slowest ray (~0.1 ms), so once few pixels are pending one big round beats several.
    // (Front rounds of 1,2,4 / 2,4 / 1,4 / 2,6 / 3 / 1,2,4,8 / 2 candidates were measured within 5 % of this one:
    // most lazy rays belong to pixels whose candidates are all occluded.  profiles/README.md)
    const int b0 = (nPairs + kLazyBatchDiv - 1) / kLazyBatchDiv;
    int left = nPairs;
    for (int r = 0; r < c->lazyRounds && left > 0; r++) {
      const int batch = (r + 1 == c->lazyRounds) ? left : (b0 < left ? b0 : left);
      left -= batch;
      uint32_t* list = P.queue[1 + (r & 1)];
      uint32_t* next = P.queue[1 + ((r + 1) & 1)];
      HIPCHK(c, hipMemsetAsync(P.rayCount + (size_t)RAY_PAIRS * kRayCursorBlock, 0, kRayCursorBlock * sizeof(uint32_t), st));
      HIPCHK(c, hipMemsetAsync(P.rayHead + (size_t)RAY_PAIRS * kRayCursorBlock, 0, kRayCursorBlock * sizeof(uint32_t), st));
      launchLazyGen(F, P, list, P.lazyCount + (size_t)r * kCursorBlock, batch, st);
      stageMark(c, st, "lazy_gen");
      launchTraceShadow(c->S, F, P, RAY_PAIRS, c->grids, c->numCUs, st);
      stageMark(c, st, "lazy_trace");
      launchLazyCheck(F, P, list, P.lazyCount + (size_t)r * kCursorBlock, batch, next, P.lazyCount + (size_t)(r + 1) * kCursorBlock, st);
    }
    stageMark(c, st, "lazy_check");
  }
  if (!(p->flags & BDPT_PARAM_DEFER_RESOLVE)) {
    launchResolve(c->splat, false, 0, c->sl, F.out, c->W, c->P.pix, c->P.Np, st);
    stageMark(c, st, "resolve");
  }
  HIPCHK(c, hipGetLastError());
  c->lastStream = st;
  return BDPT_OK;
}
}  // namespace

int bdpt_execute(bdpt_ctx* c, const bdpt_params* p, const bdpt_gbuffer* in, float* out, void* stream) {
  FrameDev F;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (int rc = frameSetup(c, p, in, out, st, F)) return rc;
  const PathBuf& P = c->P;
  if (!in) {
    // Built-in primary stage: pinhole camera, this frame's jitter and counter, the default constant
    // environment (SharedUtils/ResourceManager.cpp:77-87) — what LightProbeGBufferPass does with its defaults.
    bdpt_gbuffer_params gp{};
    gp.pixelJitter[0] = p->pixelJitter[0];
    gp.pixelJitter[1] = p->pixelJitter[1];
    gp.focalLen = 1.0f;
    gp.frameCount = p->frameCount;
    gp.envColor[0] = gp.envColor[1] = 0.5f;
    gp.envColor[2] = 0.8f;
    gp.envColor[3] = 1.0f;
    if (int rc = bdpt_gbuffer_execute(c, &gp, &c->ownGb, stream)) return rc;
  }

  c->numStages = 0;
  c->numSideStages = 0;
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[0], st));
  HIPCHK(c, hipMemsetAsync(P.qcount, 0, (size_t)kCursorWords * sizeof(uint32_t), st));
  HIPCHK(c, hipMemsetAsync(c->splat, 0, (size_t)c->sl.owners * c->sl.chunkRows * c->W * 4 * sizeof(unsigned long long), st));
  if (!(p->flags & BDPT_PARAM_KEEP_COUNTERS)) HIPCHK(c, hipMemsetAsync(c->counters, 0, sizeof(DevCounters), st));
  stageMark(c, st, "clear");

  launchInitPaths(c->S, F, P, st);
  stageMark(c, st, "init_paths");

  // Both walks (eye vertices 2..D, BDPTMain.rt.hlsl:106-112; light vertices 1..D, :138-145) in one persistent
  // launch: traversal and hit/miss shading alternate inside the kernel, lanes re-arm themselves per bounce.
  launchWalk(c->S, F, P, c->grids, c->numCUs, st);
  stageMark(c, st, "walk");

  // NEE terms (caller's stream) and splat terms (second stream) are generated side by side and traced at once (ray
  // class RAY_TERMS) while the connection generator — the long one — fills the RAY_PAIRS queue on the second stream;
  // the connection rays are traced when both are done.  (The MIS weights read both paths: everything sequential then.)
  const bool mis = (p->flags & (BDPT_PARAM_MIS_POWER | BDPT_PARAM_MIS_LINEAR)) != 0;
  if (mis) {
    launchMisPrefix(F, P, st);
    stageMark(c, st, "mis_prefix");
    launchGenNee(c->S, F, P, st);
    stageMark(c, st, "gen_nee");
    launchGenSplat(c->S, F, P, st);
    stageMark(c, st, "gen_splat");
    launchGenConnect(c->S, F, P, st);
    stageMark(c, st, "gen_connect");
    launchTraceShadow(c->S, F, P, RAY_TERMS, c->grids, c->numCUs, st);
    stageMark(c, st, "trace_terms");
  } else {
    // Stage names: a stage of the caller's stream is ONE kernel (or one memset group) — "gen_nee", "trace_terms",
    // "trace_pairs" ... — or a wait for the second stream ("splat_wait", "connect_wait": what of gen_splat / gen_connect
    // the caller's stream did not cover); the second stream's kernels are timed by event pairs of their own and reported
    // behind the caller's stages as "side:gen_splat", "side:gen_connect" (they overlap the stages above: not part of the
    // critical-path sum).
    HIPCHK(c, hipEventRecord(c->evFork, st));
    HIPCHK(c, hipStreamWaitEvent(c->walkStream, c->evFork, 0));
    sideBegin(c, c->walkStream, "side:gen_splat");
    launchGenSplat(c->S, F, P, c->walkStream);  // beside the NEE generator; both are short and latency-bound
    sideEnd(c, c->walkStream);
    HIPCHK(c, hipEventRecord(c->evSplat, c->walkStream));
    sideBegin(c, c->walkStream, "side:gen_connect");
    launchGenConnect(c->S, F, P, c->walkStream);
    sideEnd(c, c->walkStream);
    HIPCHK(c, hipEventRecord(c->evJoin, c->walkStream));
    launchGenNee(c->S, F, P, st);
    stageMark(c, st, "gen_nee");
    HIPCHK(c, hipStreamWaitEvent(st, c->evSplat, 0));
    stageMark(c, st, "splat_wait");
    launchTraceShadow(c->S, F, P, RAY_TERMS, c->grids, c->numCUs, st);
    stageMark(c, st, "trace_terms");
    HIPCHK(c, hipStreamWaitEvent(st, c->evJoin, 0));
    stageMark(c, st, "connect_wait");
  }
  launchTraceShadow(c->S, F, P, RAY_PAIRS, c->grids, c->numCUs, st);
  stageMark(c, st, "trace_pairs");
  launchGather(F, P, P.queue[1], P.lazyCount, st);
  stageMark(c, st, "gather");
  // Everything that touches the splat buffer is enqueued by now: a tiled host may start its exchange here
  // and run the connection tail beside it (BDPT_PARAM_DEFER_TAIL + bdpt_execute_tail).
  if (p->flags & BDPT_PARAM_DEFER_TAIL) {
    HIPCHK(c, hipGetLastError());
    c->lastStream = st;
    return BDPT_OK;
  }
  return connectionTail(c, F, st);
}

int bdpt_execute_tail(bdpt_ctx* c, const bdpt_params* p, const bdpt_gbuffer* in, float* out, void* stream) {
  FrameDev F;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (int rc = frameSetup(c, p, in, out, st, F)) return rc;
  stageMark(c, st, "tail_wait");  // stream time between the end of phase 1 and this call (the host's exchange set-up), not a kernel
  return connectionTail(c, F, st);
}

// Allocate the optional buffers up front so that no later execute allocates (hipGraph capture, latency).
int bdpt_prepare(bdpt_ctx* c, uint32_t what) {
  if (!c) return BDPT_E_INVALID;
  if (!c->haveSize) {
    fail(c, "prepare: bdpt_resize must be called first");
    return BDPT_E_STATE;
  }
  ENTER(c);
  if (what & BDPT_PREPARE_PRIMARY)
    if (int rc = allocOwnGbuffer(c)) return rc;
  if (what & BDPT_PREPARE_BMFR)  // (whole-frame history also on a band / stripes context: bdpt_bmfr_execute takes whole-frame buffers)
    if (int rc = allocBmfrHistory(c)) return rc;
  return BDPT_OK;
}

// BlockwiseMultiOrderFeatureRegression::execute (DenoisePass.cpp:146-204)
int bdpt_bmfr_execute(bdpt_ctx* c, const bdpt_bmfr_params* p, const bdpt_gbuffer* g, float* noisy, void* stream) {
  if (!c || !p || !g || !noisy) return BDPT_E_INVALID;
  if (!c->haveSize) {
    fail(c, "bmfr: bdpt_resize must be called first");
    return BDPT_E_STATE;
  }
  if (!g->worldPosition || !g->worldNormal || !g->materialDiffuse) {
    fail(c, "bmfr: WorldPosition, WorldNormal and MaterialDiffuse are required");
    return BDPT_E_INVALID;
  }
  ENTER(c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)c->W * c->H;
  if (!c->bmfrAccept) {  // bdpt_prepare(BDPT_PREPARE_BMFR) was not called: allocate now, unless capturing
    if (streamIsCapturing(st)) {
      fail(c, "bmfr: the history needs bdpt_prepare(BDPT_PREPARE_BMFR) before stream capture");
      return BDPT_E_STATE;
    }
    if (int rc = allocBmfrHistory(c)) return rc;
  }
  BmfrDev A{};
  A.W = c->W;
  A.H = c->H;
  A.frame = p->frameNumber;
  A.full = (p->flags & BDPT_BMFR_FULL_FRAME) ? 1u : 0u;
  A.doPre = (p->flags & BDPT_BMFR_PREPROCESS) ? 1u : 0u;
  for (int k = 0; k < 16; k++) A.m[k] = p->prevViewProj[k];
  A.curPos = reinterpret_cast<const float4*>(g->worldPosition);
  A.curNorm = g->worldNormal;
  A.albedo = g->materialDiffuse;
  A.noisy = reinterpret_cast<float4*>(noisy);
  const int r = c->bmfrRead, w = 1 - r;
  A.prevPosR = c->bmfrPos[r];
  A.prevNormR = c->bmfrNorm[r];
  A.prevNoisyR = c->bmfrNoisy[r];
  A.prevFilteredR = c->bmfrFiltered[r];
  A.prevPosW = c->bmfrPos[w];
  A.prevNormW = c->bmfrNorm[w];
  A.prevNoisyW = c->bmfrNoisy[w];
  A.prevFilteredW = c->bmfrFiltered[w];
  A.accept = c->bmfrAccept;
  A.prevPixel = c->bmfrPrevPixel;
  if (!(p->flags & BDPT_BMFR_POSTPROCESS))  // no new filtered frame this time: keep the old one on the read side next frame
    HIPCHK(c, hipMemcpyAsync(c->bmfrFiltered[w], c->bmfrFiltered[r], n * sizeof(float4), hipMemcpyDeviceToDevice, st));
  launchBmfr(A, p->flags, st);
  HIPCHK(c, hipGetLastError());
  c->bmfrRead = w;
  c->lastStream = st;
  return BDPT_OK;
}

int bdpt_bmfr_reset(bdpt_ctx* c) {
  if (!c) return BDPT_E_INVALID;
  if (!c->bmfrAccept) return BDPT_OK;  // nothing allocated yet
  ENTER(c);
  const size_t n = (size_t)c->W * c->H;
  for (int k = 0; k < 2; k++) {
    HIPCHK(c, hipMemset(c->bmfrPos[k], 0, n * sizeof(float4)));
    HIPCHK(c, hipMemset(c->bmfrNorm[k], 0, n * sizeof(float4)));
    HIPCHK(c, hipMemset(c->bmfrNoisy[k], 0, n * sizeof(float4)));
    HIPCHK(c, hipMemset(c->bmfrFiltered[k], 0, n * sizeof(float4)));
  }
  HIPCHK(c, hipMemset(c->bmfrAccept, 0, n));
  HIPCHK(c, hipMemset(c->bmfrPrevPixel, 0, n * sizeof(uint32_t)));
  c->bmfrRead = 0;
  return BDPT_OK;
}

// [pos | norm | noisy | filtered] of the read side, W * H float4 each
int bdpt_bmfr_history_bytes(const bdpt_ctx* c, uint64_t* out_bytes) {
  if (!c || !out_bytes) return BDPT_E_INVALID;
  *out_bytes = (c->haveSize && c->bmfrAccept) ? (uint64_t)c->W * c->H * sizeof(float4) * 4 : 0;
  return BDPT_OK;
}
int bdpt_bmfr_save_history(bdpt_ctx* c, void* blob, uint64_t bytes) {
  if (!c || !blob) return BDPT_E_INVALID;
  if (!c->haveSize || !c->bmfrAccept) {
    fail(c, "bmfr_save_history: no history (the denoiser has not run on this context)");
    return BDPT_E_STATE;
  }
  const size_t plane = (size_t)c->W * c->H * sizeof(float4);
  if (bytes < 4 * plane) return BDPT_E_INVALID;
  ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->lastStream));
  const int r = c->bmfrRead;
  const float4* src[4] = {c->bmfrPos[r], c->bmfrNorm[r], c->bmfrNoisy[r], c->bmfrFiltered[r]};
  for (int k = 0; k < 4; k++) HIPCHK(c, hipMemcpy(static_cast<uint8_t*>(blob) + k * plane, src[k], plane, hipMemcpyDeviceToHost));
  return BDPT_OK;
}
int bdpt_bmfr_load_history(bdpt_ctx* c, const void* blob, uint64_t bytes) {
  if (!c || !blob) return BDPT_E_INVALID;
  if (!c->haveSize) {
    fail(c, "bmfr_load_history: bdpt_resize must be called first");
    return BDPT_E_STATE;
  }
  const size_t plane = (size_t)c->W * c->H * sizeof(float4);
  if (bytes != 4 * plane) {
    fail(c, "bmfr_load_history: the blob was written for another frame size");
    return BDPT_E_INVALID;
  }
  ENTER(c);
  if (int rc = allocBmfrHistory(c)) return rc;  // (resets: read side 0)
  HIPCHK(c, hipStreamSynchronize(c->lastStream));
  const int r = c->bmfrRead;
  float4* dst[4] = {c->bmfrPos[r], c->bmfrNorm[r], c->bmfrNoisy[r], c->bmfrFiltered[r]};
  for (int k = 0; k < 4; k++) HIPCHK(c, hipMemcpy(dst[k], static_cast<const uint8_t*>(blob) + k * plane, plane, hipMemcpyHostToDevice));
  return BDPT_OK;
}

int bdpt_tile_pack(bdpt_ctx* c, const void* frame, void* packed, uint32_t bytesPerPixel, void* stream) {
  if (!c || !frame || !packed || (bytesPerPixel != 4 && bytesPerPixel != 8 && bytesPerPixel != 16)) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  ENTER(c);
  launchTilePack(frame, packed, bytesPerPixel, c->P.pix, c->P.Np, reinterpret_cast<hipStream_t>(stream));
  HIPCHK(c, hipGetLastError());
  return BDPT_OK;
}
int bdpt_tile_unpack(bdpt_ctx* c, uint32_t owner, const void* packed, void* frame, uint32_t bytesPerPixel, void* stream) {
  if (!c || !frame || !packed || (bytesPerPixel != 4 && bytesPerPixel != 8 && bytesPerPixel != 16)) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  ENTER(c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (c->stripes.stripeRows == 0) {  // a band: its rows are one run
    if (owner != 0) return BDPT_E_INVALID;
    if (c->tileRows)
      HIPCHK(c, hipMemcpyAsync(static_cast<uint8_t*>(frame) + (size_t)c->rowRanges.front().first * c->W * bytesPerPixel, packed,
                               (size_t)c->tileRows * c->W * bytesPerPixel, hipMemcpyDeviceToDevice, st));
    return BDPT_OK;
  }
  if (owner >= c->stripes.numOwners) return BDPT_E_INVALID;
  launchTileUnpack(packed, frame, bytesPerPixel, c->W, c->H, c->stripes.stripeRows, c->stripes.numOwners, owner, c->sl.chunkRows, st);
  HIPCHK(c, hipGetLastError());
  return BDPT_OK;
}

int bdpt_splat_buffer(bdpt_ctx* c, uint64_t** out_ptr, uint64_t* out_n) {
  if (!c || !out_ptr) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  *out_ptr = reinterpret_cast<uint64_t*>(c->splat);
  if (out_n) *out_n = (uint64_t)c->sl.owners * c->sl.chunkRows * c->W * 4;
  return BDPT_OK;
}

int bdpt_set_splat_buffer(bdpt_ctx* c, uint64_t* device_ptr, uint64_t num_u64) {
  if (!c) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  if (!device_ptr) {
    c->splat = c->ownSplat;
    return BDPT_OK;
  }
  if (num_u64 < (uint64_t)c->sl.owners * c->sl.chunkRows * c->W * 4) {
    fail(c, "set_splat_buffer: buffer smaller than bdpt_get_tile_info's splatU64");
    return BDPT_E_INVALID;
  }
  c->splat = reinterpret_cast<unsigned long long*>(device_ptr);
  return BDPT_OK;
}

int bdpt_resolve(bdpt_ctx* c, const uint64_t* splat, uint32_t splat_row0, float* out, void* stream) {
  if (!c || !splat || !out) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  const uint32_t firstRow = c->rowRanges.empty() ? 0 : c->rowRanges.front().first;
  if (splat_row0 > firstRow || (splat_row0 != 0 && c->sl.owners != 1)) {
    fail(c, "resolve: splat buffer does not cover the tile");
    return BDPT_E_INVALID;
  }
  ENTER(c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  launchResolve(reinterpret_cast<const unsigned long long*>(splat), false, splat_row0, c->sl, out, c->W, c->P.pix, c->P.Np, st);
  HIPCHK(c, hipGetLastError());
  return BDPT_OK;
}

int bdpt_resolve_tile(bdpt_ctx* c, const uint64_t* tile_splat, float* out, void* stream) {
  if (!c || !tile_splat || !out) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  ENTER(c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  launchResolve(reinterpret_cast<const unsigned long long*>(tile_splat), true, 0, c->sl, out, c->W, c->P.pix, c->P.Np, st);
  HIPCHK(c, hipGetLastError());
  return BDPT_OK;
}

int bdpt_accumulate(bdpt_ctx* c, float* lastFrame, float* curFrame, uint32_t accumCount, uint32_t maxAccumCount,
                    uint64_t numTexels, void* stream) {
  if (!c || !lastFrame || !curFrame) return BDPT_E_INVALID;
  ENTER(c);
  launchAccumulate(lastFrame, curFrame, accumCount, maxAccumCount, numTexels, reinterpret_cast<hipStream_t>(stream));
  HIPCHK(c, hipGetLastError());
  return BDPT_OK;
}

int bdpt_accumulate_tile(bdpt_ctx* c, float* lastFrame, float* curFrame, uint32_t accumCount, uint32_t maxAccumCount, void* stream) {
  if (!c || !lastFrame || !curFrame) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  ENTER(c);
  launchAccumulateTile(lastFrame, curFrame, accumCount, maxAccumCount, c->P.pix, c->P.Np, reinterpret_cast<hipStream_t>(stream));
  HIPCHK(c, hipGetLastError());
  return BDPT_OK;
}

// Test hook: which builder the host-only hash / check hooks (bdpt_bvh_build_hash, bdpt_bvh_build_check, bdpt_host_bvh_*)
// use for the binary tree: device >= 0 the device implementation on that device, < 0 the host code (the default).
int bdpt_test_tree_builder(int device) {
  static BvhDeviceBuild* sBuild = nullptr;
  bvhSetDefaultTreeBuilder(nullptr, nullptr);
  if (sBuild) bvhDeviceBuildEnd(sBuild);
  sBuild = nullptr;
  if (device < 0) return BDPT_OK;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device >= count) return BDPT_E_HIP;
  sBuild = bvhDeviceBuildBegin(device);
  bvhSetDefaultTreeBuilder(buildBinaryTreeOnDevice, sBuild);
  return BDPT_OK;
}

// Test hook: FNV-1a over the packed records (every node, every leaf triangle, the pad) and the summary of the
// acceleration structure of a scene — device < 0: built, quantised and packed by the host code; device >= 0: as
// bdpt_set_scene does it, tree + quantisation + packing on that device, the records read back.
int bdpt_bvh_recs_hash(const bdpt_scene_desc* d, int device, uint64_t* out_hash, bdpt_bvh_info* out_info) try {
  if (!d || !out_hash || !d->positions || !d->indices || !d->materials || !d->triMaterial || !d->numMaterials) return BDPT_E_INVALID;
  SceneBvh sb;
  BigVec<BvhRec> fromDevice;
  const BvhRec* recs = nullptr;
  size_t numRecs = 0;
  if (device < 0) {
    buildSceneBvh(d, 0, -1.0f, -1.0f, true, sb);
    recs = sb.bvh.recs.data();
    numRecs = sb.bvh.recs.size();
  } else {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count) return BDPT_E_HIP;
    std::string error;
    struct Build {
      BvhDeviceBuild* b;
      ~Build() { bvhDeviceBuildEnd(b); }
    } build{bvhDeviceBuildBegin(device, std::getenv("BDPT_HOST_COLLAPSE") == nullptr)};
    buildSceneBvh(d, 0, -1.0f, -1.0f, true, sb, buildBinaryTreeOnDevice, build.b, &error, packOnDevice, makeReferencesOnDevice, std::getenv("BDPT_HOST_COLLAPSE") == nullptr,
                  std::getenv("BDPT_HOST_PRIORITIES") == nullptr);
    if (!sb.bvh.deviceRecs && !sb.bvh.recs.empty() && error.empty()) {  // (nothing to build a tree over: the host's one empty node)
      recs = sb.bvh.recs.data();
      numRecs = sb.bvh.recs.size();
    } else {
    if (!sb.bvh.deviceRecs) return BDPT_E_HIP;
    fromDevice.resize(sb.bvh.deviceNumRecs);
    const hipError_t e = hipMemcpy(fromDevice.data(), sb.bvh.deviceRecs, fromDevice.size() * sizeof(BvhRec), hipMemcpyDeviceToHost);
    (void)hipFree(sb.bvh.deviceRecs);
    if (e != hipSuccess || !error.empty()) return BDPT_E_HIP;
    recs = fromDevice.data();
    numRecs = fromDevice.size();
    }
  }
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  mix(recs, numRecs * sizeof(BvhRec));
  mix(&sb.bvh.maxDepth, sizeof(sb.bvh.maxDepth));
  mix(&sb.bvh.maxStack, sizeof(sb.bvh.maxStack));
  mix(&sb.bvh.sahCost, sizeof(sb.bvh.sahCost));
  *out_hash = h;
  if (out_info) {
    *out_info = bdpt_bvh_info{};
    out_info->numNodes = sb.bvh.numNodes;
    out_info->numTriangles = d->numTriangles;
    out_info->maxDepth = sb.bvh.maxDepth;
    out_info->nodeBytes = sizeof(BvhRec);
    out_info->triBytes = sizeof(BvhTri);
    out_info->sahCost = sb.bvh.sahCost;
    out_info->maxStack = sb.bvh.maxStack;
    out_info->reserved = (uint32_t)numRecs;
    out_info->numReferences = sb.bvh.numRefs;
    out_info->numDropped = sb.bvh.numDropped;
    out_info->numAlphaMode = sb.numAlphaMode;
    out_info->numAlwaysPass = sb.numAlwaysPass;
  }
  return BDPT_OK;
} catch (const std::bad_alloc&) {
  return BDPT_E_NOMEM;
} catch (...) {
  return BDPT_E_INVALID;
}

int bdpt_get_counters(bdpt_ctx* c, bdpt_counters* out) {
  if (!c || !out) return BDPT_E_INVALID;
  if (!c->haveSize) return BDPT_E_STATE;
  ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->lastStream));
  DevCounters h;
  HIPCHK(c, hipMemcpy(&h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
  static_assert(sizeof(bdpt_counters) == 17 * sizeof(uint64_t), "counter fields");
  uint64_t* o = reinterpret_cast<uint64_t*>(out);
  for (int f = 0; f < 13; f++) {
    o[f] = 0;
    for (uint32_t sh = 0; sh < kCounterShards; sh++) o[f] += h.v[sh][f];
  }
  out->alphaTestsClosest = out->alphaTestsShadow = out->hintedNee = out->hintedSplat = 0;
  for (uint32_t sh = 0; sh < kCounterShards; sh++) {
    out->alphaTestsClosest += h.v[sh][C_ALPHA_CLOSEST];
    out->alphaTestsShadow += h.v[sh][C_ALPHA_SHADOW];
    out->hintedNee += h.v[sh][C_HINT_NEE];
    out->hintedSplat += h.v[sh][C_HINT_SPLAT];
  }
  out->raysPrimary = (uint64_t)c->P.Np;  // GBufferRayGen traces exactly one ray per tile pixel
  return BDPT_OK;
}

int bdpt_enable_stage_timing(bdpt_ctx* c, int enable) {
  if (!c) return BDPT_E_INVALID;
  c->timing = enable != 0;
  return BDPT_OK;
}

int bdpt_get_stage_times(bdpt_ctx* c, const char** names, float* ms, int cap) {
  if (!c || !names || !ms) return BDPT_E_INVALID;
  if (!c->timing || c->numStages == 0) return 0;
  ENTER(c);
  HIPCHK(c, hipEventSynchronize(c->ev[c->numStages]));
  int n = std::min(cap, c->numStages);
  for (int i = 0; i < n; i++) {
    names[i] = c->stageNames[i];
    float t = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
    ms[i] = t;
  }
  // the second stream's kernels, behind the caller's stages ("side:" names; the caller's stream joined them before its end)
  for (int i = 0; i < c->numSideStages && n < cap; i++, n++) {
    names[n] = c->sideNames[i];
    float t = 0.0f;
    HIPCHK(c, hipEventElapsedTime(&t, c->sideEv[2 * i], c->sideEv[2 * i + 1]));
    ms[n] = t;
  }
  return n;
}

int bdpt_sync(bdpt_ctx* c, void* stream) {
  if (!c) return BDPT_E_INVALID;
  ENTER(c);
  HIPCHK(c, hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  return BDPT_OK;
}

// ---- test hooks --------------------------------------------------------------------------------
int bdpt_test_rng(bdpt_ctx* c, const uint32_t* val0, const uint32_t* val1, uint32_t n, uint32_t draws, uint32_t* out_states,
                  float* out_floats) {
  if (!c || !val0 || !val1 || !out_states || !out_floats) return BDPT_E_INVALID;
  ENTER(c);
  std::vector<void*> pool;
  const uint32_t *d0, *d1;
  uint32_t* ds;
  float* df;
  int rc;
  if ((rc = devUpload(c, pool, &d0, val0, n)) || (rc = devUpload(c, pool, &d1, val1, n)) ||
      (rc = devAlloc(c, pool, &ds, (size_t)n * draws)) || (rc = devAlloc(c, pool, &df, (size_t)n * draws))) {
    freePool(pool);
    return rc;
  }
  launchTestRng(d0, d1, n, draws, ds, df, nullptr);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out_states, ds, (size_t)n * draws * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_floats, df, (size_t)n * draws * 4, hipMemcpyDeviceToHost);
  freePool(pool);
  if (e != hipSuccess) {
    fail(c, hipGetErrorString(e));
    return BDPT_E_HIP;
  }
  return BDPT_OK;
}

int bdpt_test_trace(bdpt_ctx* c, const float* rays, uint32_t n, int mode, int32_t* out_prim, float* out_tuv) {
  if (!c || !rays || !out_prim || !out_tuv || mode < 0 || mode > 2) return BDPT_E_INVALID;
  if (!c->haveScene) return BDPT_E_STATE;
  ENTER(c);
  std::vector<void*> pool;
  const float* dr;
  int32_t* dp;
  float* dt;
  int rc;
  if ((rc = devUpload(c, pool, &dr, rays, (size_t)n * 8)) || (rc = devAlloc(c, pool, &dp, n)) ||
      (rc = devAlloc(c, pool, &dt, (size_t)n * 3))) {
    freePool(pool);
    return rc;
  }
  launchTestTrace(c->S, dr, n, mode, dp, dt, nullptr);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out_prim, dp, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(out_tuv, dt, (size_t)n * 12, hipMemcpyDeviceToHost);
  freePool(pool);
  if (e != hipSuccess) {
    fail(c, hipGetErrorString(e));
    return BDPT_E_HIP;
  }
  return BDPT_OK;
}

int bdpt_test_trace_shadow(bdpt_ctx* c, const float* rays, uint32_t n, uint8_t* out_vis, uint32_t* out_max_stack) {
  if (!c || !rays || !out_vis || !n) return BDPT_E_INVALID;
  if (!c->haveScene) return BDPT_E_STATE;
  ENTER(c);
  // SoA planes as the ray queue holds them; tmin comes from ray 0 (the kernel takes one tmin per launch)
  std::vector<float> planes((size_t)7 * n);
  for (uint32_t i = 0; i < n; i++) {
    const float* r = rays + (size_t)i * 8;
    for (int k = 0; k < 6; k++) planes[(size_t)k * n + i] = r[k];
    planes[(size_t)6 * n + i] = r[7];
  }
  std::vector<uint32_t> cursors(2 * kCursorStride, 0u);
  cursors[0] = n;  // count; head = cursors[kCursorStride] = 0
  std::vector<void*> pool;
  const float* dp;
  const uint32_t* dc;
  uint8_t* dv;
  DevCounters* dcnt;
  int rc;
  if ((rc = devUpload(c, pool, &dp, planes.data(), planes.size())) || (rc = devUpload(c, pool, &dc, cursors.data(), cursors.size())) ||
      (rc = devAlloc(c, pool, &dv, (size_t)n)) || (rc = devAlloc(c, pool, &dcnt, 1))) {
    freePool(pool);
    return rc;
  }
  hipError_t e = hipMemset(dcnt, 0, sizeof(DevCounters));
  if (e == hipSuccess) {
    launchTestTraceShadow(c->S, dp, n, dc, const_cast<uint32_t*>(dc) + kCursorStride, dv, dcnt, rays[6], c->numCUs, nullptr);
    e = hipDeviceSynchronize();
  }
  DevCounters h;
  if (e == hipSuccess) e = hipMemcpy(out_vis, dv, (size_t)n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&h, dcnt, sizeof(h), hipMemcpyDeviceToHost);
  freePool(pool);
  if (e != hipSuccess) {
    fail(c, hipGetErrorString(e));
    return BDPT_E_HIP;
  }
  if (out_max_stack) {
    unsigned long long m = 0;
    for (uint32_t sh = 0; sh < kCounterShards; sh++) m = std::max(m, h.v[sh][C_STACK_MAX]);
    *out_max_stack = (uint32_t)m;
  }
  return BDPT_OK;
}

int bdpt_test_bsdf(bdpt_ctx* c, const float* in, uint32_t n, uint32_t matIndex, float* out) {
  if (!c || !in || !out) return BDPT_E_INVALID;
  ENTER(c);
  std::vector<void*> pool;
  const float* di;
  float* dout;
  int rc;
  if ((rc = devUpload(c, pool, &di, in, (size_t)n * 20)) || (rc = devAlloc(c, pool, &dout, (size_t)n * 16))) {
    freePool(pool);
    return rc;
  }
  launchTestBsdf(di, n, matIndex, dout, nullptr);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 64, hipMemcpyDeviceToHost);
  freePool(pool);
  if (e != hipSuccess) {
    fail(c, hipGetErrorString(e));
    return BDPT_E_HIP;
  }
  return BDPT_OK;
}

}  // extern "C"
