/*
 * bdpt.h — C ABI of the MI355X-native bidirectional path-tracing render pass.
 *
 * This is the drop-in boundary.  Every entry point names the
 * reference interface it replaces (paths relative to /root/reference/src):
 *
 *   bdpt_create / bdpt_destroy      RayLaunch::create + compileRayProgram
 *                                   (SharedUtils/RayLaunch.h:85-101,
 *                                    BidirectionalPathtracing/Passes/BDPTPass.cpp:34-47)
 *   bdpt_set_scene                  RayLaunch::setScene -> RtProgramVars/RtScene::createTlas,
 *                                   RtModel::buildAccelerationStructure
 *                                   (SharedUtils/RayLaunch.cpp:105-167,
 *                                    Falcor/Framework/Source/Raytracing/RtModel.cpp:181-254,
 *                                    Falcor/Framework/Source/Raytracing/RtScene.cpp:220-308)
 *   bdpt_set_environment            the "EnvironmentMap" channel BDPTPass requests (BDPTPass.cpp:29; bound by no shader
 *                                   of the pass in the reference): read only with BDPT_PARAM_ENV_ON_MISS
 *   bdpt_bvh_build_check / _hash, bdpt_host_bvh_*
 *                                   (none in the reference: the DXR driver's acceleration structure is opaque) —
 *                                   host-only checks of the builder and a CPU walk of its tree for tests and tools
 *   bdpt_set_camera                 gCamera constant buffer (CameraData,
 *                                   Falcor/Framework/Source/Data/HostDeviceSharedCode.h:69-99;
 *                                   Camera::calculateCameraParameters, Graphics/Camera/Camera.cpp:129-136)
 *   bdpt_gbuffer_execute            LightProbeGBufferPass::execute -> DispatchRays of GBufferRayGen
 *                                   (CommonPasses/LightProbeGBufferPass.cpp:104-161,
 *                                    CommonPasses/Data/CommonPasses/lightProbeGBuffer.rt.hlsl:63-159)
 *   bdpt_execute                    BDPTPass::execute -> RayLaunch::execute -> DispatchRays of
 *                                   SimpleDiffuseGIRayGen (BidirectionalPathtracing/Passes/BDPTPass.cpp:70-107,
 *                                   SharedUtils/RayLaunch.cpp:200-223,
 *                                   BidirectionalPathtracing/Data/BDPTMain.rt.hlsl:42-234)
 *   bdpt_splat_buffer / bdpt_set_splat_buffer / bdpt_resolve
 *                                   the cross-pixel gOutput[id] read-modify-write of the
 *                                   light-tracing loop (BDPTMain.rt.hlsl:186-204), made
 *                                   deterministic: fixed-point atomics into a separate buffer,
 *                                   one final saturate
 *   bdpt_resize / bdpt_resize_stripes / bdpt_prepare
 *                                   RenderPass::resize + the lazy texture creation of
 *                                   ResourceManager::requestTextureResource (SharedUtils/RenderPass.h:42,
 *                                   SharedUtils/ResourceManager.cpp:210-241): per-tile path state; the striped form
 *                                   and bdpt_resolve_tile / bdpt_accumulate_tile / bdpt_get_tile_info have no
 *                                   counterpart (the reference renders on one GPU: a single DispatchRays,
 *                                   Falcor API/D3D12/D3D12RenderContext.cpp:350-384) — they are the multi-GPU
 *                                   tiling of SURVEY.md §8e
 *   bdpt_accumulate                 SimpleAccumulationPass::execute + accumulate.ps.hlsl
 *                                   (CommonPasses/SimpleAccumulationPass.cpp:104-134,
 *                                    CommonPasses/Data/CommonPasses/accumulate.ps.hlsl:28-42)
 *   bdpt_bmfr_execute / bdpt_bmfr_reset
 *                                   BlockwiseMultiOrderFeatureRegression::execute / resize
 *                                   (BidirectionalPathtracing/Passes/DenoisePass.cpp:146-279 with
 *                                    Data/preprocess.ps.hlsl:33-165, regressionCP.hlsl:100-500,
 *                                    postprocess.ps.hlsl:22-91); the pass is off by default
 *                                   (DenoisePass.h:71) and outside radiance parity
 *   bdpt_camera_view_proj           Camera::calculateCameraParameters' viewProjMat without jitter
 *                                   (Falcor Graphics/Camera/Camera.cpp:60-109), for prevViewProjMat
 *   bdpt_get_counters               (none in the reference; replaces nothing — ray tallies
 *                                    needed by the Mrays/s metric)
 *   bdpt_last_error                 Falcor logError / silent no-op conventions
 *                                   (BidirectionalPathtracing/Passes/BDPTPass.cpp:76)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success and a negative BDPT_E_* code on failure; nothing throws across the
 * boundary; `stream` is a hipStream_t passed as void* (NULL = default stream).
 * All image buffers are device pointers, row-major, pitch = width.
 */
#ifndef BDPT_H_
#define BDPT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDPT_OK 0
#define BDPT_E_INVALID (-1)  /* bad argument */
#define BDPT_E_STATE (-2)    /* scene / camera / size not set */
#define BDPT_E_HIP (-3)      /* HIP runtime error, see bdpt_last_error */
#define BDPT_E_NOMEM (-4)
#define BDPT_E_LIMIT (-5)    /* exceeds a documented limit (lights, depth) */

#define BDPT_MAX_LIGHTS 16   /* MAX_LIGHT_SOURCES, Falcor Data/HostDeviceSharedMacros.h:152 */
#define BDPT_MAX_DEPTH 16    /* reference caps at 8 (BDPTPass.h:33); storage here is depth-parametric */

/* Light types, Falcor Data/HostDeviceSharedMacros.h:145-150 */
#define BDPT_LIGHT_POINT 0
#define BDPT_LIGHT_DIRECTIONAL 1

/* Material flag bit layout is Falcor's (Data/HostDeviceSharedMacros.h:69-123). */
#define BDPT_SHADING_MODEL_METAL_ROUGH 0u
#define BDPT_SHADING_MODEL_SPEC_GLOSS 2u
#define BDPT_CHANNEL_UNUSED 0u
#define BDPT_CHANNEL_CONST 1u
#define BDPT_CHANNEL_TEXTURE 2u
#define BDPT_NORMAL_MAP_UNUSED 0u
#define BDPT_NORMAL_MAP_RGB 1u
#define BDPT_NORMAL_MAP_RG 2u
#define BDPT_ALPHA_MODE_OPAQUE 0u
#define BDPT_ALPHA_MODE_MASK 1u
#define BDPT_FLAG_SHADING_MODEL(f) (((f) >> 0) & 7u)
#define BDPT_FLAG_DIFFUSE_TYPE(f) (((f) >> 3) & 7u)
#define BDPT_FLAG_SPECULAR_TYPE(f) (((f) >> 6) & 7u)
#define BDPT_FLAG_EMISSIVE_TYPE(f) (((f) >> 9) & 7u)
#define BDPT_FLAG_NORMAL_MAP_TYPE(f) (((f) >> 12) & 3u)
#define BDPT_FLAG_ALPHA_MODE(f) (((f) >> 17) & 3u)
#define BDPT_FLAG_DOUBLE_SIDED(f) (((f) >> 19) & 1u)
#define BDPT_MAKE_FLAGS(model, dif, spec, emis, nmap, alpha, dbl)                              \
  ((uint32_t)(((model) & 7u) | (((dif) & 7u) << 3) | (((spec) & 7u) << 6) | (((emis) & 7u) << 9) | \
              (((nmap) & 3u) << 12) | (((alpha) & 3u) << 17) | (((dbl) & 1u) << 19)))

/* MaterialData (Falcor Data/HostDeviceSharedCode.h:119-135) without the
 * resource handles; textures are indices into bdpt_scene_desc.textures. 64 B. */
typedef struct bdpt_material {
  float baseColor[4];
  float specular[4];
  float emissive[3];
  float alphaThreshold;
  float IoR;
  uint32_t flags;
  int16_t texBaseColor; /* -1 = none */
  int16_t texSpecular;
  int16_t texEmissive;
  int16_t texNormal;
} bdpt_material;

/* One mip-0 RGBA8 image.  The reference binds a linear/wrap sampler to every
 * scene texture (SharedUtils/SceneLoaderWrapper.cpp:65-68) and only samples
 * at explicit LOD 0 on this path (BDPTUtils.hlsli:6, lightProbeGBuffer.rt.hlsl:102). */
typedef struct bdpt_texture {
  const uint8_t* rgba8; /* host pointer, width*height*4 bytes, row 0 first */
  uint32_t width;
  uint32_t height;
  uint32_t srgb; /* 1: rgb channels are sRGB-encoded (Falcor default for colour textures) */
  uint32_t reserved;
} bdpt_texture;

/* The LightData fields this path reads (Falcor Data/HostDeviceSharedCode.h:199-217). 64 B. */
typedef struct bdpt_light {
  float posW[3];
  uint32_t type;
  float dirW[3];
  float openingAngle;
  float intensity[3];
  float cosOpeningAngle;
  float penumbraAngle;
  float reserved[3];
} bdpt_light;

/* World-space, instancing flattened (the reference loads with
 * Model::LoadFlags::RemoveInstancing, SharedUtils/SceneLoaderWrapper.cpp:58).
 * Every vertex stream has 12-byte stride, including texcoords (.xy used), as in
 * Falcor's ray-tracing vertex fetch (ShadingUtils/Raytracing.slang:79-85). */
typedef struct bdpt_scene_desc {
  uint32_t numVertices;
  uint32_t numTriangles;
  uint32_t numMaterials;
  uint32_t numTextures;
  uint32_t numLights;
  uint32_t reserved;
  const float* positions;
  const float* normals;
  const float* bitangents; /* may be NULL when no material has a normal map */
  const float* texcoords;  /* may be NULL when no material samples a texture */
  const uint32_t* indices;     /* 3 per triangle */
  const uint32_t* triMaterial; /* 1 per triangle */
  const bdpt_material* materials;
  const bdpt_texture* textures;
  const bdpt_light* lights;
} bdpt_scene_desc;

/* posW + cameraU/V/W of CameraData (Data/HostDeviceSharedCode.h:86-92). */
typedef struct bdpt_camera {
  float posW[3];
  float cameraU[3];
  float cameraV[3];
  float cameraW[3];
} bdpt_camera;

/* GlobalCB of BDPTMain.rt.hlsl:12-22 plus build-only switches. */
typedef struct bdpt_params {
  float minT;            /* gMinT, default 1e-4 (SharedUtils/ResourceManager.h:150) */
  uint32_t frameCount;   /* gFrameCount, starts at 0x1337 (BDPTPass.h:44) */
  uint32_t matIndex;     /* gMatIndex 0 = GGX, 1 = Lambertian */
  float refractiveIndex; /* gRefractiveIndex — read by no shader, kept for layout */
  uint32_t maxDepth;     /* gMaxDepth */
  float emitMult;        /* gEmitMult — read by no shader */
  float clampUpper;      /* gClampUpper, default 0.9 */
  float pixelJitter[2];  /* gPixelJitter = kMSAA[(frame+1)%8]/16 + 0.5 */
  uint32_t flags;        /* BDPT_PARAM_* */
} bdpt_params;

#define BDPT_PARAM_COUNTERS 1u      /* also tally node / triangle visits (slower; not for timing).  Ray tallies are always on. */
#define BDPT_PARAM_DEFER_RESOLVE 2u /* leave splats in the splat buffer; caller runs bdpt_resolve */
#define BDPT_PARAM_NO_NEE 4u        /* partial images for stage-wise parity (SURVEY §8c iv) */
#define BDPT_PARAM_NO_SPLAT 8u
#define BDPT_PARAM_NO_CONNECT 16u
/* sampleGGXBRDF never writes its `out bool isSpecular` (MaterialUtils.hlsli:209-252): undefined in
 * HLSL.  Default build definition: false.  With this flag: true iff the GGX lobe was sampled. */
#define BDPT_PARAM_SPECULAR_FROM_LOBE 32u
/* Weight every strategy with getWeightPower / getWeightLinear (BDPTUtils.hlsli:226-278) instead of the
 * uniform 1/k the reference applies (it defines these functions but never calls them). */
#define BDPT_PARAM_MIS_POWER 64u
#define BDPT_PARAM_MIS_LINEAR 128u
/* Two-phase execute for tiled multi-GPU hosts: bdpt_execute returns once every launch that writes the
 * splat buffer is enqueued (walks, NEE, splats, non-zero connections); bdpt_execute_tail enqueues the rest
 * (zero-valued connection rounds, resolve unless deferred).  The host starts its splat exchange between
 * the two so it overlaps the tail.  Same image as the one-call form, bit for bit. */
#define BDPT_PARAM_DEFER_TAIL 256u
/* Do not zero the counters at the start of this execute: tallies add up over frames until an execute
 * without the flag (lets a host keep several frames in flight without a per-frame read-back;
 * raysPrimary in bdpt_get_counters stays one frame's worth). */
#define BDPT_PARAM_KEEP_COUNTERS 512u
/* Lighting the reference leaves out of its walks (RayMiss returns black, globalIlluminationRay.hlsli:14-19; emissive is
 * only added for the primary hit, BDPTMain.rt.hlsl:155-158).  Off by default = the
 * reference's image.  Build definitions, both for the EYE walk only and both weighted like the path-tracing strategy
 * of the same length (uniform 1/edges, as the reference weights its NEE terms), clamped with gClampUpper, added
 * without saturate after the pixel's own emissive term and before the NEE terms, in bounce order:
 *   ENV_ON_MISS    a ray leaving eye vertex k that misses adds cameraPath[k].color * environment(dir) / (k + 1); the
 *                  environment is the lat-long lookup of the G-buffer pass's miss shader
 *                  (lightProbeGBuffer.rt.hlsl:63-74) on the map given to bdpt_set_environment, else its constant colour
 *   EMISSIVE_HITS  a ray leaving eye vertex k that hits an emissive surface adds cameraPath[k].color * emissive / (k + 1) */
#define BDPT_PARAM_ENV_ON_MISS 1024u
#define BDPT_PARAM_EMISSIVE_HITS 2048u

/* RayGenCB of lightProbeGBuffer.rt.hlsl:45-52 + the miss shader's env map. */
typedef struct bdpt_gbuffer_params {
  float pixelJitter[2]; /* (0.5,0.5) when jitter is off */
  float lensRadius;
  float focalLen;
  uint32_t frameCount;  /* starts at 0xdeadbeef (CommonPasses/LightProbeGBufferPass.h:79) */
  uint32_t useThinLens;
  uint32_t envWidth;    /* gEnvMapRes */
  uint32_t envHeight;
  const float* envMap;  /* device RGBA32F, envWidth*envHeight*4 floats; NULL = constant envColor */
  float envColor[4];    /* default (0.5,0.5,0.8,1) (SharedUtils/ResourceManager.cpp:77-87) */
} bdpt_gbuffer_params;

/* The six G-buffer channels in the formats the first requester fixes
 * (CommonPasses/LightProbeGBufferPass.cpp:46-51): position RGBA32F, the rest
 * RGBA16F.  Device pointers, width*height texels each (full frame). */
typedef struct bdpt_gbuffer {
  float* worldPosition;          /* float4 */
  uint16_t* worldNormal;         /* half4: N.xyz, distance */
  uint16_t* materialDiffuse;     /* half4: diffuse.rgb, opacity | env colour on miss */
  uint16_t* materialSpecRough;   /* half4: specular.rgb, linearRoughness */
  uint16_t* materialExtraParams; /* half4: IoR, lightMap.rgb */
  uint16_t* emissive;            /* half4: emissive.rgb, 0 */
} bdpt_gbuffer;

/* Rows [y0,y1) of the full width×height frame are this context's tile.  The
 * light sub-path of every tile pixel still splats anywhere in the frame. */
typedef struct bdpt_tile {
  uint32_t y0;
  uint32_t y1;
} bdpt_tile;

/* Interleaved stripes for multi-GPU hosts: the frame's rows are dealt to numOwners contexts in
 * stripes of stripeRows rows; context `owner` renders stripes owner, owner + numOwners, ...  Cost that varies by
 * row (sky above, geometry below) then spreads over all owners, which contiguous bands do not give. */
typedef struct bdpt_stripes {
  uint32_t stripeRows;
  uint32_t numOwners;
  uint32_t owner;
} bdpt_stripes;

typedef struct bdpt_tile_info {
  uint32_t numRows;      /* rows this context renders */
  uint32_t numPixels;    /* numRows * width */
  uint32_t chunkRows;    /* rows of one owner's chunk of the splat buffer (its rows in order, zero-padded) */
  uint32_t numRowRanges; /* contiguous row runs (stripes) of this tile */
  uint64_t splatU64;     /* uint64 words of the whole splat buffer = numOwners * chunkU64 */
  uint64_t chunkU64;     /* uint64 words of one owner's chunk = chunkRows * width * 4 */
} bdpt_tile_info;

typedef struct bdpt_counters {
  uint64_t raysPrimary;
  uint64_t raysEyeExtend;
  uint64_t raysLightExtend;
  uint64_t raysNee;
  uint64_t raysSplat;
  uint64_t raysConnect;
  uint64_t nodeVisitsClosest; /* interior-node visits, closest-hit queries */
  uint64_t triTestsClosest;
  uint64_t nodeVisitsShadow;  /* interior-node visits, any-hit queries */
  uint64_t triTestsShadow;
  uint64_t pixelsValid;       /* G-buffer pixels with geometry */
  uint64_t splatsLanded;
  uint64_t raysConnectLazy;   /* subset of raysConnect traced by the gather stage for zero-valued pairs */
  uint64_t alphaTestsClosest; /* any-hit alpha tests (alphaTestFails, BDPTUtils.hlsli:115-127) run by closest-hit queries, */
  uint64_t alphaTestsShadow;  /* ... and by any-hit queries; with BDPT_PARAM_COUNTERS (the G-buffer ray is not tallied) */
  /* Any-hit queries answered "occluded" by their occluder hint — one triangle tried before the traversal: the nearest
   * triangle the light sees towards the vertex (next-event rays), the triangle the camera sees through the target pixel
   * (light-tracing rays).  Such a query never enters the ray queue and is NOT part of raysNee / raysSplat. */
  uint64_t hintedNee;
  uint64_t hintedSplat;
} bdpt_counters;

/* The environment secondary misses see (BDPT_PARAM_ENV_ON_MISS): an RGBA32F lat-long map (device pointer for
 * bdpt_set_environment, host pointer for the oracle) or, when envMap is NULL, a constant colour. */
typedef struct bdpt_environment {
  const float* envMap;
  uint32_t width, height;
  float color[4];
} bdpt_environment;

typedef struct bdpt_bvh_info {
  uint32_t numNodes;
  uint32_t numTriangles;
  uint32_t maxDepth;
  uint32_t nodeBytes; /* bytes per interior node */
  uint32_t triBytes;  /* bytes per leaf triangle */
  float sahCost;
  uint32_t maxStack;  /* worst-case traversal stack entries this tree needs (the device holds 32 per lane) */
  uint32_t reserved;
  uint32_t numReferences; /* leaf entries: a triangle cut by spatial pre-splitting has one per piece */
  uint32_t numDropped;    /* alpha-mode triangles without any reference: every texel they can sample fails the alpha test */
  uint32_t numAlphaMode;  /* triangles whose material is not AlphaModeOpaque (the reference runs the any-hit test on all of them) */
  uint32_t numAlwaysPass; /* of those: triangles whose texels all pass, traversed as opaque */
} bdpt_bvh_info;

typedef struct bdpt_ctx bdpt_ctx;

int bdpt_create(int device_ordinal, bdpt_ctx** out_ctx);
void bdpt_destroy(bdpt_ctx* ctx);
const char* bdpt_last_error(const bdpt_ctx* ctx);

/* Copies the scene and builds its acceleration structure on the context's device (what the reference leaves to the DXR
 * driver: Falcor RtModel.cpp:181-254, RtScene.cpp:220-308).  The caller keeps its arrays.  BDPT_E_INVALID for indices or
 * material ids out of range and for a triangle with a vertex position that is not finite; BDPT_E_NOMEM when host or
 * device memory runs out during the build; BDPT_E_LIMIT past 2^28 triangles / 2^31 records. */
int bdpt_set_scene(bdpt_ctx* ctx, const bdpt_scene_desc* scene);
int bdpt_get_bvh_info(const bdpt_ctx* ctx, bdpt_bvh_info* out);
int bdpt_set_camera(bdpt_ctx* ctx, const bdpt_camera* cam);
/* ResourceManager's "EnvironmentMap" channel as the BDPT pass would bind it (BDPTPass.cpp:29 requests it; no shader of
 * the pass reads it in the reference).  NULL = none (black).  Only read with BDPT_PARAM_ENV_ON_MISS. */
int bdpt_set_environment(bdpt_ctx* ctx, const bdpt_environment* env);

/* Host-only (no GPU, no context): run the acceleration-structure builder on a scene (geometry only: every triangle
 * opaque) and check its invariants — every triangle referenced, every leaf entry in exactly one leaf, the pieces of
 * a split triangle covering it, every child box containing its subtree's pieces, depth within the traversal stack.
 * Returns BDPT_OK and fills *out, or BDPT_E_INVALID with the first violated invariant in msg (msgCap bytes, may be NULL). */
int bdpt_bvh_build_check(const bdpt_scene_desc* scene, bdpt_bvh_info* out, char* msg, uint32_t msgCap);

/* Host-only test hook: build with `threads` host threads (0 = default: BDPT_BUILD_THREADS, else the CPUs this
 * process may use) and return a 64-bit FNV-1a hash of the node array, the leaf-ordered triangle list and the
 * summary; the tree must not depend on the thread count.  A scene with materials is built exactly as bdpt_set_scene
 * builds it (traversal flags, alpha classification, pre-splitting), one without as plain geometry.  out_info may be
 * NULL (its `reserved` field returns the default thread count). */
int bdpt_bvh_build_hash(const bdpt_scene_desc* scene, int threads, uint64_t* out_hash, bdpt_bvh_info* out_info);

/* Test hook: the builder the host-only hooks above and below use for the BINARY TREE stage — device >= 0: the device
 * implementation bdpt_set_scene uses (csrc/bvh_device.hip) on that device; device < 0: the host code (the default).  The
 * two build the same tree bit for bit; the GPU tests compare bdpt_bvh_build_hash under both settings. */
int bdpt_test_tree_builder(int device);
/* Test hook: FNV-1a over the packed records (every node, every leaf triangle) and the summary of a scene's acceleration
 * structure.  device < 0: everything by the host code; device >= 0: as bdpt_set_scene builds it — classification and split
 * priorities, references, binary tree, four-wide collapse, quantisation and packing on that device — with the records read back.  The two must agree; info->reserved = number of records. */
int bdpt_bvh_recs_hash(const bdpt_scene_desc* scene, int device, uint64_t* out_hash, bdpt_bvh_info* out_info);

/* Host-only test hooks: the acceleration structure exactly as bdpt_set_scene builds it (traversal flags, alpha
 * classification, spatial pre-splitting; negative budgets = build defaults, classify = 0 keeps the reference's
 * per-material opacity) walked on the CPU with the device's query semantics (bdpt_test_trace), or — brute != 0 —
 * the linear scan over every input triangle with the reference's per-material flags, which defines the right
 * answer.  rays: n x 8 floats (origin, direction, tmin, tmax); out_visits[2] = interior-node visits and triangle
 * tests summed over the rays (tree walk only).  The scene description must outlive the handle. */
void* bdpt_host_bvh_create(const bdpt_scene_desc* scene, int threads, float splitBudget, float splitBudgetAlpha, int classify,
                           bdpt_bvh_info* out_info);
void bdpt_host_bvh_destroy(void* handle);
int bdpt_host_bvh_trace(void* handle, const float* rays, uint32_t n, int mode, int brute, int threads, int32_t* out_prim,
                        float* out_tuv, uint64_t* out_visits);

/* Camera::calculateCameraParameters (Graphics/Camera/Camera.cpp:129-136) with
 * fovY = focalLengthToFovY (Utils/Math/FalcorMath.h:148-151).  Host-only helper. */
int bdpt_camera_look_at(const float pos[3], const float target[3], const float up[3], float focalLengthMm,
                        float frameHeightMm, float aspect, float focalDistance, bdpt_camera* out);

/* kMSAA jitter table shared by both passes (BDPTPass.cpp:20, LightProbeGBufferPass.cpp:36):
 * out = kMSAA[(frameCounterBeforeIncrement + 1) % 8] / 16 + 0.5. */
void bdpt_msaa_jitter(uint32_t frameCounterBeforeIncrement, float out[2]);

/* Size the per-pixel path state for a width×height frame of which rows
 * [tile.y0, tile.y1) are rendered here, at up to maxDepth. */
int bdpt_resize(bdpt_ctx* ctx, uint32_t width, uint32_t height, bdpt_tile tile, uint32_t maxDepth);

/* bdpt_resize for a tile made of interleaved stripes.  The splat buffer of such a context is OWNER-MAJOR: chunk o
 * (chunkU64 words, bdpt_tile_info) holds the accumulators of owner o's rows in row order, so a reduce-scatter over
 * the numOwners ranks leaves every rank with the summed accumulators of exactly its own pixels, in tile-local
 * order, ready for bdpt_resolve_tile.  (With bdpt_resize the buffer is in plain frame order.) */
int bdpt_resize_stripes(bdpt_ctx* ctx, uint32_t width, uint32_t height, bdpt_stripes stripes, uint32_t maxDepth);
/* The stripe height the tiled hosts of this build use for a frame of `height` rows dealt to `numOwners` ranks (C++
 * RenderingPipeline::setTiling, Python tiling.stripe_rows): small enough that every rank gets at least four stripes,
 * at most 8 rows — a stripe is only a run of rows in the tile's pixel list, so its size costs nothing, and small ones
 * spread the rows that cost most.  Host-only helper; any stripeRows >= 1 is valid for
 * bdpt_resize_stripes as long as all ranks agree. */
uint32_t bdpt_stripe_rows(uint32_t height, uint32_t numOwners);
int bdpt_get_tile_info(const bdpt_ctx* ctx, bdpt_tile_info* out);
/* The tile's rows as [first, last) pairs in ascending order; writes up to cap pairs, returns how many. */
int bdpt_tile_row_ranges(const bdpt_ctx* ctx, uint32_t* out_first_last, uint32_t cap);

/* Allocate optional per-frame buffers ahead of time, so that no execute call allocates (hipGraph capture,
 * first-frame latency): the built-in primary stage's channels and/or the BMFR history.  Call after
 * bdpt_resize (a resize frees them again).  Replaces the lazy texture creation of
 * ResourceManager::requestTextureResource / DenoisePass::initialize (SharedUtils/ResourceManager.cpp:210-241,
 * BidirectionalPathtracing/Passes/DenoisePass.cpp:60-96). */
#define BDPT_PREPARE_PRIMARY 1u
#define BDPT_PREPARE_BMFR 2u
int bdpt_prepare(bdpt_ctx* ctx, uint32_t what);

/* Primary-visibility pass.  Writes the tile rows of all six channels. */
int bdpt_gbuffer_execute(bdpt_ctx* ctx, const bdpt_gbuffer_params* p, const bdpt_gbuffer* out, void* stream);

/* The BDPT pass.  `out` is the full-frame RGBA32F "PipelineOutput" channel;
 * only the tile rows are written (cleared to 0 first, as getClearedTexture does,
 * BDPTPass.cpp:73).  Without BDPT_PARAM_DEFER_RESOLVE the splats of this call are
 * folded in before returning control to the stream.
 * `in` may be NULL: the primary stage then runs inside the call into channels the context owns
 * (pinhole, p->pixelJitter, p->frameCount, default constant environment).  They are allocated by
 * bdpt_prepare(BDPT_PREPARE_PRIMARY), or else by the first such call — which therefore must not be
 * inside a stream capture (BDPT_E_STATE). */
int bdpt_execute(bdpt_ctx* ctx, const bdpt_params* p, const bdpt_gbuffer* in, float* out, void* stream);

/* Second phase of a bdpt_execute issued with BDPT_PARAM_DEFER_TAIL (same params, channels and out). */
int bdpt_execute_tail(bdpt_ctx* ctx, const bdpt_params* p, const bdpt_gbuffer* in, float* out, void* stream);

/* Full-frame fixed-point splat accumulator: uint64[4] per pixel (r,g,b in
 * 2^-32 units, splat count) in frame order (bdpt_resize) or owner-major order (bdpt_resize_stripes),
 * zeroed by every bdpt_execute before its splat stage.
 * Exposed so a multi-GPU host can sum it across ranks (integer sum: exact and
 * order-independent) before bdpt_resolve. */
int bdpt_splat_buffer(bdpt_ctx* ctx, uint64_t** out_device_ptr, uint64_t* out_num_u64);

/* Make the pass accumulate splats into caller-owned device memory (e.g. a tensor that a
 * collective library can reduce in place); NULL restores the context's own buffer. */
int bdpt_set_splat_buffer(bdpt_ctx* ctx, uint64_t* device_ptr, uint64_t num_u64);

/* out[tile rows] = saturate(out + splat) where the splat count is non-zero.
 * `splat` may be the context's own buffer or a reduced copy holding at least the
 * tile rows at the same full-frame indexing (splat_row0 = first row it holds). */
int bdpt_resolve(bdpt_ctx* ctx, const uint64_t* splat, uint32_t splat_row0, float* out, void* stream);

/* The same for a buffer that holds the tile's accumulators in tile-local order (4 uint64 per tile pixel): one
 * owner's chunk after the reduce-scatter of an owner-major splat buffer. */
int bdpt_resolve_tile(bdpt_ctx* ctx, const uint64_t* tile_splat, float* out, void* stream);

/* Running mean of accumulate.ps.hlsl:28-42 over `numTexels` RGBA32F texels:
 *   cur = accumCount < maxAccumCount ? (accumCount*last + cur)/(accumCount+1) : last;  last = cur. */
int bdpt_accumulate(bdpt_ctx* ctx, float* lastFrame, float* curFrame, uint32_t accumCount, uint32_t maxAccumCount,
                    uint64_t numTexels, void* stream);

/* The same running mean restricted to this context's tile: lastFrame / curFrame are full-frame RGBA32F buffers, only
 * the tile's pixels are read and written (one launch, whatever the number of stripes). */
int bdpt_accumulate_tile(bdpt_ctx* ctx, float* lastFrame, float* curFrame, uint32_t accumCount, uint32_t maxAccumCount, void* stream);

/* ---- BMFR denoiser (DenoisePass.*).  Works on the full frame: the context's tile must cover it. ---- */
#define BDPT_BMFR_PREPROCESS 1u        /* mBMFR_preprocess, default on (DenoisePass.h:72) */
#define BDPT_BMFR_REGRESSION 2u        /* mBMFR_regression, default off (DenoisePass.h:74) */
#define BDPT_BMFR_POSTPROCESS 4u       /* mBMFR_postprocess, default on (DenoisePass.h:73) */
#define BDPT_BMFR_KEEP_LD_FEATURES 8u  /* mBMFR_removeFeatures == false: plain Householder QR with feature noise */
#define BDPT_BMFR_FULL_FRAME 16u       /* deviation switch: the reference filters only texC.x <= 0.5 (its comparison split) */

typedef struct bdpt_bmfr_params {
  uint32_t frameNumber;   /* mAccumCount: 0 on the first frame after bdpt_bmfr_reset */
  uint32_t flags;         /* BDPT_BMFR_* */
  float prevViewProj[16]; /* gCamera.prevViewProjMat, row-major: clip[r] = sum_c m[4r+c] * (posW,1)[c] */
} bdpt_bmfr_params;

/* One execute() of the denoise pass on `noisy` (full-frame RGBA32F, in/out: the channel being
 * denoised) with the G-buffer's WorldPosition / WorldNormal / MaterialDiffuse as features.  History
 * (previous position, normal, noisy and filtered frames, accept masks) lives in the context; it is allocated
 * by bdpt_prepare(BDPT_PREPARE_BMFR) or by the first call (not inside a stream capture: BDPT_E_STATE).
 * The filter's blocks need their neighbours, so ALL FOUR buffers must hold the WHOLE frame, also when the context
 * renders only a band or the stripes of one rank: a tiled host gathers them first (bdpt_tile_pack -> all-gather ->
 * bdpt_tile_unpack; host/Passes.cpp BlockwiseMultiOrderFeatureRegression::execute) and every rank filters the same
 * whole frame. */
int bdpt_bmfr_execute(bdpt_ctx* ctx, const bdpt_bmfr_params* p, const bdpt_gbuffer* features, float* noisy, void* stream);

/* Forget the history (BlockwiseMultiOrderFeatureRegression::resize / initScene set mAccumCount = 0). */
int bdpt_bmfr_reset(bdpt_ctx* ctx);

/* The denoiser's state that crosses frames — the previous frame's position, normal, noisy and filtered images, what the
 * next bdpt_bmfr_execute reads — as one host blob, so that a host can checkpoint a pipeline with the denoiser on
 * (the reference keeps these in textures of the pass, DenoisePass.h:40-66, and has no checkpoints at all).
 * bdpt_bmfr_history_bytes: the blob's size for this context's frame (0 before the history exists).
 * save: synchronises the last stream used; `bytes` must be at least that size.  load: allocates the history if need
 * be; `bytes` must be exactly that size (BDPT_E_INVALID otherwise: a blob of another frame size). */
int bdpt_bmfr_history_bytes(const bdpt_ctx* ctx, uint64_t* out_bytes);
int bdpt_bmfr_save_history(bdpt_ctx* ctx, void* host_blob, uint64_t bytes);
int bdpt_bmfr_load_history(bdpt_ctx* ctx, const void* host_blob, uint64_t bytes);

/* Rows of a tile as one contiguous run, and back: `frame` is a whole-frame buffer of `bytesPerPixel` (4, 8 or 16) bytes
 * per pixel, `packed` holds rows of `width` pixels.
 * bdpt_tile_pack:   packed[i] = frame[pixel i of this context's tile] (its rows in ascending order: the order of a
 *                   stripes context's chunk of an all-gather).
 * bdpt_tile_unpack: the rows of owner `owner` of this context's stripe layout (bdpt_resize_stripes; a band context has
 *                   the one owner 0) go from `packed` (numRows of that owner, in order; rows the padding of a chunk
 *                   stands for lie outside the frame and are skipped) back to their places in `frame`.
 * What a tiled host needs around an all-gather of tile framebuffers when the gathered frame
 * stays on the device (the denoiser); RenderingPipeline::readOutput does the same on the host. */
int bdpt_tile_pack(bdpt_ctx* ctx, const void* frame, void* packed, uint32_t bytesPerPixel, void* stream);
int bdpt_tile_unpack(bdpt_ctx* ctx, uint32_t owner, const void* packed, void* frame, uint32_t bytesPerPixel, void* stream);

/* projMat * viewMat of Falcor's camera without the jitter matrix (glm::perspective * glm::lookAt,
 * Camera.cpp:77-105), row-major as bdpt_bmfr_params::prevViewProj wants it.  Host only. */
int bdpt_camera_view_proj(const float pos[3], const float target[3], const float up[3], float focalLengthMm,
                          float frameHeightMm, float aspect, float nearZ, float farZ, float out16[16]);

/* Counters of the most recent bdpt_execute (ray tallies always; node/triangle visits when it ran
 * with BDPT_PARAM_COUNTERS).  Synchronises the stream it was launched on. */
int bdpt_get_counters(bdpt_ctx* ctx, bdpt_counters* out);

/* Per-stage device time (ms) of the most recent bdpt_execute, measured with
 * HIP events on the launch stream.  names/ms hold up to `cap` entries; returns
 * the number written (>= 0) or a negative error.  A stage of the launch stream is one kernel ("walk", "gen_nee",
 * "trace_terms", "trace_pairs", "gather", "lazy_gen", "lazy_trace" ...), one group of memsets ("clear") or a wait for
 * the context's second stream ("splat_wait", "connect_wait"); their sum is the frame's critical path.  The kernels of
 * the second stream follow with names that start with "side:" ("side:gen_splat", "side:gen_connect"), timed by event
 * pairs on that stream: they run beside the stages above and are not part of that sum. */
int bdpt_get_stage_times(bdpt_ctx* ctx, const char** names, float* ms, int cap);
int bdpt_enable_stage_timing(bdpt_ctx* ctx, int enable);

int bdpt_sync(bdpt_ctx* ctx, void* stream);

/* Test hooks through the same library (used by the parity tests; they launch
 * the device functions of the hot path on caller-supplied inputs). */
/* initRand/nextRand stream: out[i*draws + k] = k-th nextRand seed state for (val0[i], val1[i]). */
int bdpt_test_rng(bdpt_ctx* ctx, const uint32_t* val0, const uint32_t* val1, uint32_t n, uint32_t draws,
                  uint32_t* out_states, float* out_floats);
/* Closest-hit / any-hit queries on host ray arrays (origin xyz, dir xyz, tmin, tmax per ray).
 * hit: prim (int32, -1 miss), t, u, v per ray.  mode 0 closest, 1 closest+cull-back, 2 any-hit (prim = 0/-1). */
int bdpt_test_trace(bdpt_ctx* ctx, const float* rays, uint32_t n, int mode, int32_t* out_prim, float* out_tuv);
/* The persistent any-hit kernel (the one the pass uses for NEE, splat and connection rays) over host rays of the same
 * layout; tmin is taken from ray 0.  out_vis: 1 = unoccluded.  out_max_stack (may be NULL): deepest traversal stack any
 * ray reached — entries beyond the LDS rows live in the context's overflow area (kernels.h kStackLds). */
int bdpt_test_trace_shadow(bdpt_ctx* ctx, const float* rays, uint32_t n, uint8_t* out_vis, uint32_t* out_max_stack);
/* BSDF known-answer hook: inputs are n records of 20 floats
 * (N3 V3 L3 dif3 spec3 rough isSpecular seed(bits) pad2); outputs n records of 16 floats
 * (sampleBRDF: weight3 L3 pdf isSpec | evalBRDF: f3 | pad).  matIndex bit 1 = BDPT_PARAM_SPECULAR_FROM_LOBE. */
int bdpt_test_bsdf(bdpt_ctx* ctx, const float* in, uint32_t n, uint32_t matIndex, float* out);

#ifdef __cplusplus
}
#endif
#endif /* BDPT_H_ */
