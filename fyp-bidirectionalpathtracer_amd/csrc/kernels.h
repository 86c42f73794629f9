// kernels.h — device-side views and launch entry points shared by kernels.hip and api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bdpt.h"

namespace bdpt {

constexpr int kWave = 64;          // gfx950 wavefront
#ifndef KSTACK
#define KSTACK 32
#endif
constexpr int kStackEntries = KSTACK;  // per-lane traversal stack capacity (the builder's bound is kBvhMaxStack = KSTACK - 1)
// The any-hit kernel keeps only the first BDPT_STACK_LDS entries of a lane's stack in LDS (6 KiB per wave); the few
// deeper entries of very deep descents go to a per-context overflow area in device memory (SceneDev::stackOvf).
// Measured on the bench frame: rays never go beyond 12 entries there, and the smaller LDS footprint (more resident
// any-hit waves, room for the connection generator beside them) is worth 3 % of the frame (profiles/README.md).
#ifndef BDPT_STACK_LDS
#define BDPT_STACK_LDS 24
#endif
constexpr int kStackLds = BDPT_STACK_LDS < KSTACK ? BDPT_STACK_LDS : KSTACK;
constexpr int kMaxPersistentPerCU = 32;  // resident one-wave workgroups per CU a persistent grid may use (8 per SIMD)
// Rows of the per-context stack overflow area (SceneDev::stackOvf).  INVARIANT the area relies on: it is indexed by
// workgroup and lane only, so no two persistent traversal launches of one context may be resident at once — every
// launchWalk / launchTraceShadow of a context goes to the caller's stream, the context's second stream only runs
// generators (api.cpp bdpt_execute, evFork / evJoin), and the test hooks synchronise the device first.
constexpr int kStackOvfRows = KSTACK - kStackLds;
constexpr int kShadeRecF4 = 7;  // float4s per triangle shading record (7 used = 112 B)
constexpr uint32_t kNoRay = 0xFFFFFFFFu;
// Hot single-word atomics top out near 90 M/s on this chip (MI355X_MICROARCH.md "dequeue"), so
// producer and consumer cursors of every queue are sharded over sub-queues, the any-hit trace kernel
// takes up to kFetchChunk rays per atomic, and the statistics counters are sharded per workgroup.
constexpr uint32_t kNumSubQueues = 32;
#ifndef BDPT_FETCH_CHUNK
#define BDPT_FETCH_CHUNK 256
#endif
constexpr uint32_t kFetchChunk = BDPT_FETCH_CHUNK;
constexpr uint32_t kCursorStride = 32;                            // uint32 words between two sub-queue cursors: one 128-byte line each
constexpr uint32_t kCursorBlock = kNumSubQueues * kCursorStride;  // words of one sharded cursor
// The ray queues have their own cursor count (BDPT_RAY_SUBQUEUES).  Measured on the bench frame: 128 cursors instead
// of 32 change nothing (20.9 vs 20.5 ms: the appends of the generators are not what bounds them).
#ifndef BDPT_RAY_SUBQUEUES
#define BDPT_RAY_SUBQUEUES 32
#endif
constexpr uint32_t kNumRaySubQueues = BDPT_RAY_SUBQUEUES;
constexpr uint32_t kRayCursorBlock = kNumRaySubQueues * kCursorStride;  // words of one sharded ray cursor
constexpr uint32_t kCounterShards = 64;

// Path-vertex field ids (PathVertex, BDPT/RayPathData.hlsli:1-45); kernels.hip "Path vertices" maps them to the 96-byte record.  pdfForward is only read by the
// MIS weights (BDPT_PARAM_MIS_*), which the reference defines but never calls.
enum : int { F_COL = 0, F_POS = 3, F_N = 6, F_V = 9, F_DIF = 12, F_SPEC = 15, F_ROUGH = 18, F_ISSPEC = 19, F_PDF = 20, NF = 21 };
constexpr int NF4 = 6;  // float4s per stored vertex record (96 B; layout in kernels.hip "Path vertices")
enum : int { PATH_EYE = 0, PATH_LIGHT = 1 };
enum : int { RAY_TERMS = 0, RAY_PAIRS = 1 };

struct TexDev {
  const uint8_t* px;
  uint32_t w, h, srgb, pad;
};

struct SceneConst {
  bdpt_light lights[BDPT_MAX_LIGHTS];
  float srgbLut[256];
};

struct SceneDev {
  const uint4* recs;        // 3 per 48-byte record: interior nodes and leaf triangles in one array (bvh.h BvhRec)
  const float4* shade;      // kShadeRecF4 per primitive, primitive order
  const float* bitangents;  // 3 per vertex (normal-mapped primary hits only)
  const uint32_t* indices;  // 3 per primitive
  const bdpt_material* materials;
  const TexDev* textures;
  const TexDev* matTex;     // 4 per material: descriptors of its base-colour, specular, emissive and normal textures (zero when absent)
  const SceneConst* sc;
  uint32_t numLights;
  uint32_t hasBitangents;
  uint32_t numRecs;
  const float4* alphaRecs;  // 4 per non-opaque triangle (device_scene.hpp alphaTestFails), indexed by BvhTri::aux
  int* stackOvf;            // overflow rows of the persistent kernels' stacks: [entry - kStackLds][workgroup * 64 + lane]
  uint32_t stackOvfStride;  // lanes per row
  // occluder hints for next-event rays (kernels.hip "Occluder hints"): per point / spot light a cube map of the record
  // index of the nearest triangle in each direction from the light; [light][face][v][u], kNoHint where nothing is seen
  const uint32_t* lightMap;
  uint32_t lightMapRes;     // texels per cube-face edge (0: no maps)
};

struct DevCounters {  // [shard][field]; fields mirror bdpt_counters; two 128-byte lines per shard
  unsigned long long v[kCounterShards][32];
};
enum : int {
  C_RAYS_PRIMARY = 0, C_RAYS_EYE, C_RAYS_LIGHT, C_RAYS_NEE, C_RAYS_SPLAT, C_RAYS_CONNECT,
  C_NODE_CLOSEST, C_TRI_CLOSEST, C_NODE_SHADOW, C_TRI_SHADOW, C_PIX_VALID, C_SPLATS, C_RAYS_LAZY,
  C_STACK_MAX,  // deepest any-hit traversal stack seen (a maximum, not a sum; with BDPT_PARAM_COUNTERS)
  C_ALPHA_CLOSEST, C_ALPHA_SHADOW,  // any-hit alpha tests run by closest-hit / any-hit queries (with BDPT_PARAM_COUNTERS)
  C_HINT_NEE, C_HINT_SPLAT          // any-hit queries answered "occluded" by their occluder hint (not part of C_RAYS_*)
};

// Number of connection pairs the reference defines for depth D (cameraLength <= totalLength,
// BDPTMain.rt.hlsl:212-216): sum_{t=2..D} min(t, D-1).
inline __host__ __device__ uint32_t numConnectPairs(uint32_t D) {
  uint32_t n = 0;
  for (uint32_t t = 2; t <= D; t++) n += (t < D - 1) ? t : (D - 1);
  return n;
}

// Per-tile path state, SoA by tile-local pixel index p in [0, Np).
struct PathBuf {
  float* v;            // vertex records: ((path*D1 + k)*Np + p) * NF4 float4s
  float* rayDir;       // planes: (path*3 + axis)*Np + p
  const uint32_t* pix; // tile-local pixel p -> full-frame pixel index y*W + x (the tile's rows, row-major)
  uint32_t* seedE;     // RNG state the eye walk draws from at every bounce (initRand of the pixel, quirk 1)
  uint32_t* seedL;     // RNG state after sampleLight
  uint8_t* eyeLast;    // last stored eye vertex (ghost included); 0 = pixel has no geometry
  uint8_t* lightLast;  // last stored light vertex (ghost included)
  uint8_t* lightReal;  // number of light vertices produced by hits (takeContribution, BDPTMain.rt.hlsl:144)
  uint32_t* queue[3];  // sharded pixel queues (kernels.hip "Path queues"): [0] valid pixels; [1],[2] lazy-round ping-pong
  uint32_t pathSubCap; // capacity of one list of a path queue (multiple of 64)
  // cursor blocks: each is kNumSubQueues cursors, one per 128-byte line (kCursorBlock words)
  uint32_t* qcount;    // block 0 = valid-pixel list lengths
  uint32_t* qhead;     // blocks 0-1 = fetch cursors of the walk kernel (one per virtual list: pixel list x path)
  // shadow-ray queue (NEE + splat + connection rays of one frame)
  float* rayQ;           // 7 planes, stride rayCap
  float* rayContrib;     // 3 planes, stride rayCap: the clamped contribution the ray gates
  uint8_t* rayVis;       // visibility by ray id
  // Two ray classes share the planes: RAY_TERMS (NEE + splat rays) and RAY_PAIRS (connection rays, lazy rounds), each
  // with its own sub-queues and cursors, so the first can be traced while the second is still being generated.
  uint32_t* rayCount;    // cursor blocks [class]: rays queued per sub-queue
  uint32_t* rayHead;     // cursor blocks [class]: fetch cursors
  uint32_t raySubCap[2]; // capacity of one sub-queue of the class
  uint32_t rayBase[2];   // first ray id of the class; ray id = rayBase[c] + subQueue*raySubCap[c] + offset
  uint32_t* slotRay;     // planes: slot*Np + p -> ray id or kNoRay.  slots: [0,D) NEE, [D,2D) splat, [2D,..) pairs
  uint32_t* splatPix;    // planes: t*Np + p -> full-frame pixel index of splat t
  float* misE;           // planes: k*Np + p, k in [0, D]: eye-side prefix product of getWeightPower/Linear
  float* misL;           // same, light side
  uint8_t* lazyCursor;   // next connection-pair ordinal a pending pixel has not examined yet
  uint32_t* lazyRay;     // planes: b*Np + p -> ray id of the b-th lazy ray of the current round
  uint32_t* lazyCount;   // cursor blocks: pending-list lengths, one block per round
  uint32_t rayCap;
  uint32_t Np, D1;
};

// Where a frame pixel's splat accumulator lives: owner-major, so that one reduce-scatter over ranks hands every
// rank the accumulators of its own rows as one contiguous chunk.  Rows are dealt to `owners` ranks in stripes of
// `stripeRows`; owner o's chunk holds its rows in order, padded to chunkRows.  owners == 1 is plain frame order.
struct SplatLayout {
  uint32_t stripeRows, owners, chunkRows;
};
inline __host__ __device__ size_t splatIndex(const SplatLayout& L, uint32_t W, uint32_t x, uint32_t y) {
  const uint32_t s = y / L.stripeRows;
  const uint32_t owner = s % L.owners, row = (s / L.owners) * L.stripeRows + (y - s * L.stripeRows);
  return ((size_t)owner * L.chunkRows + row) * W + x;
}

struct FrameDev {
  bdpt_camera cam;
  bdpt_params p;
  uint32_t W, H;
  SplatLayout sl;
  float* out;                 // full-frame RGBA32F
  unsigned long long* splat;  // 4 x u64 per pixel, SplatLayout order
  bdpt_gbuffer gb;
  DevCounters* counters;
  // environment of the BDPT pass (BDPT_PARAM_ENV_ON_MISS; bdpt_set_environment): RGBA32F lat-long map or constant colour
  const float* envMap;
  uint32_t envW, envH;
  float envColor[3];
  // occluder hints for light-tracing rays: record index of the triangle the primary ray of each FRAME pixel hit
  // (written by the context's own G-buffer pass; kNoHint where it has not run or saw nothing); NULL = no hints
  const uint32_t* hintPix;
};

struct GBufferDev {
  bdpt_camera cam;
  bdpt_gbuffer_params gp;
  uint32_t W, H, Np;
  const uint32_t* pix;  // the tile's pixels (PathBuf::pix)
  bdpt_gbuffer gb;
  DevCounters* counters;
  uint32_t* hintPix;    // FrameDev::hintPix, written here (may be NULL)
};

// BMFR denoise pass (bmfr.hip).  History buffers come in ping-pong pairs: R = previous frame (read), W = this
// frame (written), so the reference's post-pass blits (DenoisePass.cpp:180-182, 194) cost no extra copy.
struct BmfrDev {
  uint32_t W, H, frame;
  uint32_t full, doPre;
  float m[16];                 // prevViewProj, row-major
  const float4* curPos;        // WorldPosition
  const uint16_t* curNorm;     // WorldNormal (half4)
  const uint16_t* albedo;      // MaterialDiffuse (half4)
  float4* noisy;               // channel being denoised, in/out
  const float4 *prevPosR, *prevNormR, *prevNoisyR, *prevFilteredR;
  float4 *prevPosW, *prevNormW, *prevNoisyW, *prevFilteredW;
  uint8_t* accept;             // BMFR_AcceptedBools
  uint32_t* prevPixel;         // BMFR_PrevFramePixel, RG16Float
};
void launchBmfr(const BmfrDev& A, uint32_t flags, hipStream_t st);

// launchers (kernels.hip)
void launchGBuffer(const SceneDev& S, const GBufferDev& G, hipStream_t st);
// SceneDev::alphaRecs (4 float4 per non-opaque triangle, in the order of alphaTris) from the shading records and material tables
void launchAlphaRecs(const SceneDev& S, const uint32_t* alphaTris, uint32_t n, float4* out, hipStream_t st);
// FrameDev::hintPix of EVERY frame pixel (G.Np = W * H; nothing else is written): partial-tile contexts, on a camera change
void launchHintFill(const SceneDev& S, const GBufferDev& G, hipStream_t st);
// SceneDev::lightMap of every point / spot light of the scene (res texels per face edge), closest-hit rays from the light
void launchLightMaps(const SceneDev& S, uint32_t* maps, uint32_t res, hipStream_t st);
void launchInitPaths(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st);
// Persistent-grid sizes of one context's device, filled on first use (occupancy query per kernel variant).
struct LaunchGrids {
  uint32_t walk[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [EXT][GGX][COUNT]
  uint32_t shadow[2] = {0, 0};        // [COUNT]
};
// both random walks of the frame: one persistent launch (trace + hit/miss shading in place)
void launchWalk(const SceneDev& S, const FrameDev& F, const PathBuf& P, LaunchGrids& G, int numCUs, hipStream_t st);
void launchMisPrefix(const FrameDev& F, const PathBuf& P, hipStream_t st);
void launchGenNee(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st);
void launchGenSplat(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st);
void launchGenConnect(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st);
void launchTraceShadow(const SceneDev& S, const FrameDev& F, const PathBuf& P, int rayClass, LaunchGrids& G, int numCUs, hipStream_t st);
void launchGather(const FrameDev& F, const PathBuf& P, uint32_t* lazyList, uint32_t* lazyCount, hipStream_t st);
void launchLazyGen(const FrameDev& F, const PathBuf& P, const uint32_t* list, const uint32_t* listCount, int batch, hipStream_t st);
void launchLazyCheck(const FrameDev& F, const PathBuf& P, const uint32_t* list, const uint32_t* listCount, int batch,
                     uint32_t* nextList, uint32_t* nextCount, hipStream_t st);
constexpr int kLazyBatchDiv = 8;    // a front round examines ceil(pairs / 8) candidates per pending pixel
constexpr int kMaxLazyRounds = 8;   // cursor blocks reserved for lazy rounds  // rounds per frame; batch = ceil(pairs / rounds)
// out[pix] = saturate(out[pix] + splat) for the tile's pixels.  tileLocal: `splat` holds the tile's accumulators in
// tile-local order (a reduce-scattered chunk); otherwise SplatLayout order, starting at frame row splatRow0 (owners == 1).
void launchResolve(const unsigned long long* splat, bool tileLocal, uint32_t splatRow0, const SplatLayout& L, float* out, uint32_t W,
                   const uint32_t* pix, uint32_t Np, hipStream_t st);
void launchAccumulate(float* last, float* cur, uint32_t accumCount, uint32_t maxAccum, uint64_t numTexels, hipStream_t st);
void launchAccumulateTile(float* last, float* cur, uint32_t accumCount, uint32_t maxAccum, const uint32_t* pix, uint32_t Np,
                          hipStream_t st);
// rows of a tile <-> a contiguous run (bdpt_tile_pack / bdpt_tile_unpack): elements of 4, 8 or 16 bytes
void launchTilePack(const void* frame, void* packed, uint32_t bytesPerPixel, const uint32_t* pix, uint32_t Np, hipStream_t st);
void launchTileUnpack(const void* packed, void* frame, uint32_t bytesPerPixel, uint32_t W, uint32_t H, uint32_t stripeRows, uint32_t owners,
                      uint32_t owner, uint32_t packedRows, hipStream_t st);
void launchTestRng(const uint32_t* v0, const uint32_t* v1, uint32_t n, uint32_t draws, uint32_t* states, float* floats,
                   hipStream_t st);
void launchTestTrace(const SceneDev& S, const float* rays, uint32_t n, int mode, int32_t* prim, float* tuv, hipStream_t st);
void launchTestTraceShadow(const SceneDev& S, const float* planes, uint32_t cap, const uint32_t* count, uint32_t* head, uint8_t* vis,
                           DevCounters* counters, float tmin, int numCUs, hipStream_t st);
void launchTestBsdf(const float* in, uint32_t n, uint32_t matIndex, float* out, hipStream_t st);

}  // namespace bdpt
