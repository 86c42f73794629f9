/*
 * bdpt_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar C++17 restatement of the reference's bidirectional path-tracing pass
 * (SunBangjie/FYP-BidirectionalPathTracer).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; nothing under
 * fyp-bidirectionalpathtracer_amd/ links, imports or calls it.
 *
 * Parity pinning status: the reference holds no golden vectors,
 * known-answer tests or rendered images for this path and cannot be built or run
 * here (HLSL/DXR + D3D12, Windows-only; toolchain and dependencies absent), so
 *   - the integrator arithmetic (RNG, BSDF sampling/eval, NEE, splat, connection,
 *     accumulation) is a line-by-line restatement of the cited shader lines and is
 *     self-pinned by the fixtures under tests/golden/ that this oracle generated
 *     (tests/golden/make_golden.py);
 *   - BVH traversal, ray/triangle intersection, texture filtering and the
 *     float->half G-buffer rounding live in the DXR driver / sampler hardware with
 *     no source in the reference: for those, PARITY IS UNPINNED and the oracle's
 *     definition (Moeller-Trumbore, brute-force cross-check, bilinear/wrap,
 *     round-to-nearest-even) is the build's own.
 *
 * Scene / camera / parameter structs are the product ABI's (include/bdpt.h) so
 * both sides consume byte-identical inputs.
 */
#ifndef BDPT_ORACLE_H_
#define BDPT_ORACLE_H_

#include <stdint.h>

#include "../include/bdpt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

/* Copies the scene, builds the oracle's own (median-split) BVH. */
oracle_scene* oracle_scene_create(const bdpt_scene_desc* desc);
void oracle_scene_destroy(oracle_scene* s);
/* Environment for BDPT_PARAM_ENV_ON_MISS (include/bdpt.h); envMap is a HOST pointer here and is copied. NULL = black. */
void oracle_set_environment(oracle_scene* s, const bdpt_environment* env);

#define ORACLE_BRUTE_FORCE 1u /* intersect every triangle instead of using the BVH */
/* Cross-check hook (tests/test_oracle_cross_check.py): treat every vertex-connection shadow ray as unoccluded.
 * The reference ends those rays exactly ON the far surface (tmax = |B-A|, BDPTMain.rt.hlsl:222-223), so whether
 * the far surface itself occludes them is decided by the last bit of the arithmetic; with this flag the
 * connection sums can be compared with an implementation in another precision. */
#define ORACLE_CONNECT_ALL_VISIBLE 2u

typedef struct oracle_frame {
  uint32_t width, height;
  uint32_t y0, y1; /* rows rendered */
  /* Full-frame host buffers (width*height texels).  G-buffer channels hold the
   * values the BDPT pass would read back: position fp32, others fp16-rounded. */
  float* worldPosition;     /* 4 floats/texel */
  float* worldNormal;       /* 4 */
  float* materialDiffuse;   /* 4 */
  float* materialSpecRough; /* 4 */
  float* materialExtra;     /* 4 */
  float* emissive;          /* 4 */
  float* out;               /* 4: own-pixel terms, then resolved */
  uint64_t* splat;          /* 4 u64/texel, full frame; caller zeroes */
} oracle_frame;

/* G-buffer pass for rows [y0,y1). */
int oracle_gbuffer(const oracle_scene* s, const bdpt_camera* cam, const bdpt_gbuffer_params* gp,
                   const float* envMapHost, oracle_frame* f, uint32_t flags, int threads);

/* BDPT pass for rows [y0,y1): own-pixel terms into f->out, splats into f->splat.
 * Honors BDPT_PARAM_NO_* flags.  counters may be NULL. */
int oracle_bdpt(const oracle_scene* s, const bdpt_camera* cam, const bdpt_params* p, oracle_frame* f,
                uint32_t flags, int threads, bdpt_counters* counters);

/* out = saturate(out + splat) where count != 0, rows [y0,y1). */
int oracle_resolve(oracle_frame* f);

void oracle_accumulate(float* lastFrame, float* curFrame, uint32_t accumCount, uint32_t maxAccumCount,
                       uint64_t numTexels);

/* Known-answer hooks mirroring bdpt_test_* of include/bdpt.h. */
void oracle_rng(const uint32_t* val0, const uint32_t* val1, uint32_t n, uint32_t draws, uint32_t* out_states,
                float* out_floats);
void oracle_trace(const oracle_scene* s, const float* rays, uint32_t n, int mode, uint32_t flags,
                  int32_t* out_prim, float* out_tuv);
void oracle_bsdf(const float* in, uint32_t n, uint32_t matIndex, float* out);
void oracle_sincos2pi(const float* u, uint32_t n, float* s, float* c);
void oracle_half_round(const float* in, uint32_t n, float* out);

/* BMFR denoise pass (bmfr_oracle.cpp; DenoisePass.cpp:146-279 + its three shaders).  Channels are
 * float4 per pixel; curNorm / albedo hold the half-precision G-buffer values widened to float. */
typedef struct oracle_bmfr oracle_bmfr;
oracle_bmfr* oracle_bmfr_create(uint32_t width, uint32_t height);
void oracle_bmfr_destroy(oracle_bmfr* b);
void oracle_bmfr_reset(oracle_bmfr* b);
int oracle_bmfr_execute(oracle_bmfr* b, const bdpt_bmfr_params* p, const float* curPos, const float* curNorm,
                        const float* albedo, float* noisy);

#ifdef __cplusplus
}
#endif
#endif
