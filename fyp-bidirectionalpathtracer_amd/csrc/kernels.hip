// kernels.hip — the wavefront BDPT pipeline for gfx950 (CDNA4, wave64).
//
// Per frame (the caller's stream plus one stream the context owns for two of the generators; no host
// synchronisation in between):
//   gbuffer_kernel      primary visibility                      CP lightProbeGBuffer.rt.hlsl:63-159
//   init_paths_kernel   eye vertex 1 + light vertex 0           BDPTMain.rt.hlsl:51-103, 124-135
//   walk_kernel         both random walks, one persistent launch:
//                       closest-hit traversal                   globalIlluminationRay.hlsli:1-12 (TraceRay)
//                       + hit/miss shading in place             globalIlluminationRay.hlsli:14-45
//   gen_{nee,splat,connect}_kernel  terms -> shadow-ray queue   BDPTMain.rt.hlsl:161-233
//   trace_shadow_kernel persistent any-hit traversal            standardShadowRay.hlsli:7-49
//   gather_kernel       ordered sums, splat atomics             BDPTMain.rt.hlsl:166, 199, 230
//   resolve_kernel      fold the splat buffer in                (build definition)
//   accumulate_kernel   running mean                            CP accumulate.ps.hlsl:28-42
//
// Layout: path vertices are 96-byte records by pixel (a lane owns a sub-path and moves whole records), shadow
// rays SoA planes by ray id (a wave reads 64 consecutive floats per field); valid pixels and generated rays are
// compacted with __ballot + popcount prefix + one atomic per wave; BVH traversal happens in two persistent
// kernels — the walk kernel and the any-hit trace kernel (device_trace.hpp) — whose lanes are refilled as they
// retire, with each lane's stack in LDS interleaved by lane (entry e of lane l at word e*64+l: pushes and pops
// never bank-conflict).  One workgroup = one wave (64 threads): __syncthreads only orders a wave's own LDS
// traffic, and a finished wave frees its slot immediately.
#include "kernels.h"

#include <algorithm>

#include "device_math.hpp"
#include "device_scene.hpp"
#include "device_trace.hpp"

namespace bdpt {

#define BD __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// Pixel queues (valid pixels, lazy-round lists).  A queue is kNumSubQueues dense lists (workgroup b appends to and, in the dense
// kernels, reads list b % kNumSubQueues) with one cursor per list on its own 128-byte line, because
// every wave of a launch hitting one atomic word caps the chip near 90 M appends/s.
//   item of list q at offset i lives at items[q*subCap + i]; count[q*kCursorStride] = list length
// ------------------------------------------------------------------------------------------------
// q: the list the wave appends to (default: workgroup b -> list b % kNumSubQueues)
BD void wavePush(bool active, uint32_t value, uint32_t* items, uint32_t* count, uint32_t subCap, uint32_t q = 0xffffffffu) {
  unsigned long long mask = __ballot(active);
  if (mask == 0ull) return;
  const int lane = (int)(threadIdx.x & 63u);
  const int leader = __ffsll((long long)mask) - 1;
  if (q == 0xffffffffu) q = blockIdx.x % kNumSubQueues;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(&count[q * kCursorStride], (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, leader);
  const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  if (active) items[q * subCap + base + prefix] = value;
}
// Dense consumer: workgroup b handles entries [64*(b / NQ), +64) of list b % NQ.  Returns false for
// the whole wave when the chunk lies beyond the list (wave-uniform), else sets act / idx per lane.
BD bool queueChunk(const uint32_t* count, uint32_t subCap, bool& act, uint32_t& idx) {
  const uint32_t q = blockIdx.x % kNumSubQueues, c0 = (blockIdx.x / kNumSubQueues) * kWave;
  const uint32_t n = count[q * kCursorStride];
  if (c0 >= n) return false;
  const uint32_t i = c0 + threadIdx.x;
  act = i < n;
  idx = q * subCap + i;
  return true;
}

// ------------------------------------------------------------------------------------------------
// Path vertices: one 96-byte record (six float4) per (path, k, pixel), records of one (path, k) contiguous
// by pixel.  A lane owns a sub-path, so it reads and writes whole records: six 16-byte accesses that use
// every byte of the lines they touch, whatever pixel the lane holds (the walk kernel's lanes hold unrelated
// pixels; the dense kernels' lanes hold consecutive ones and so cover 6 KiB contiguously per wave).
//   q0 = pos.xyz, roughness   q1 = N.xyz, isSpecular   q2 = diffuse.xyz, pdfForward
//   q3 = specular.xyz, -      q4 = colour.xyz, -       q5 = V.xyz, -
// ------------------------------------------------------------------------------------------------
constexpr int kVtxQ = NF4;
BD float4* vtxPtr(const PathBuf& P, int path, int k, uint32_t p) {
  return reinterpret_cast<float4*>(P.v) + ((size_t)(path * (int)P.D1 + k) * P.Np + p) * kVtxQ;
}
BD constexpr int fieldQ(int f) { return f == F_POS ? 0 : f == F_N ? 1 : f == F_DIF ? 2 : f == F_SPEC ? 3 : f == F_COL ? 4 : 5; }
BD f3 ldPlane3(const PathBuf& P, int path, int k, int f, uint32_t p) {
  const float4 q = vtxPtr(P, path, k, p)[fieldQ(f)];
  return f3{q.x, q.y, q.z};
}
BD float ldPlane1(const PathBuf& P, int path, int k, int f, uint32_t p) {  // F_ROUGH, F_ISSPEC, F_PDF: .w of q0, q1, q2
  const int q = (f == F_ROUGH) ? 0 : (f == F_ISSPEC ? 1 : 2);
  return reinterpret_cast<const float*>(vtxPtr(P, path, k, p))[q * 4 + 3];
}

struct Vtx {
  f3 color, pos, N, V, dif, spec;
  float rough;
  bool isSpec;
  float pdf;
};
BD Vtx zeroVtx() {
  Vtx v;
  v.color = v.pos = v.N = v.V = v.dif = v.spec = mk(0);
  v.rough = 0.0f;
  v.isSpec = false;
  v.pdf = 0.0f;
  return v;
}
BD void storeVtx(const PathBuf& P, int path, int k, uint32_t p, const Vtx& v) {
  float4* r = vtxPtr(P, path, k, p);
  const float4 q0 = make_float4(v.pos.x, v.pos.y, v.pos.z, v.rough), q1 = make_float4(v.N.x, v.N.y, v.N.z, v.isSpec ? 1.0f : 0.0f),
               q2 = make_float4(v.dif.x, v.dif.y, v.dif.z, v.pdf), q3 = make_float4(v.spec.x, v.spec.y, v.spec.z, 0.0f),
               q4 = make_float4(v.color.x, v.color.y, v.color.z, 0.0f), q5 = make_float4(v.V.x, v.V.y, v.V.z, 0.0f);
  r[0] = q0;
  r[1] = q1;
  r[2] = q2;
  r[3] = q3;
  r[4] = q4;
  r[5] = q5;
}
// geometry + material of a stored vertex (no colour, no V); GGX = false skips the fields Lambert never reads
template <bool GGX = true>
BD void loadSurf(const PathBuf& P, int path, int k, uint32_t p, Vtx& v) {
  const float4* r = vtxPtr(P, path, k, p);
  const float4 q0 = r[0], q1 = r[1], q2 = r[2];
  v.pos = mk(q0.x, q0.y, q0.z);
  v.N = mk(q1.x, q1.y, q1.z);
  v.dif = mk(q2.x, q2.y, q2.z);
  v.pdf = q2.w;
  if (GGX) {
    const float4 q3 = r[3];
    v.spec = mk(q3.x, q3.y, q3.z);
    v.rough = q0.w;
    v.isSpec = q1.w != 0.0f;
  } else {
    v.spec = mk(0);
    v.rough = 0.0f;
    v.isSpec = false;
  }
}

BD void unpackHalf4(const uint16_t* base, size_t idx, float& a, float& b, float& c, float& d) {
  const uint2 raw = reinterpret_cast<const uint2*>(base)[idx];
  a = f16_to_f32((uint16_t)(raw.x & 0xffffu));
  b = f16_to_f32((uint16_t)(raw.x >> 16));
  c = f16_to_f32((uint16_t)(raw.y & 0xffffu));
  d = f16_to_f32((uint16_t)(raw.y >> 16));
}
BD void packHalf4(uint16_t* base, size_t idx, float a, float b, float c, float d) {
  uint2 raw;
  raw.x = (uint32_t)f32_to_f16(a) | ((uint32_t)f32_to_f16(b) << 16);
  raw.y = (uint32_t)f32_to_f16(c) | ((uint32_t)f32_to_f16(d) << 16);
  reinterpret_cast<uint2*>(base)[idx] = raw;
}

// ------------------------------------------------------------------------------------------------
// G-buffer pass: GBufferRayGen + PrimaryClosestHit/AnyHit/Miss (CP lightProbeGBuffer.rt.hlsl:63-159)
// ------------------------------------------------------------------------------------------------
BD float atan2_WAR(float y, float x) {  // CP lightProbeGBufferUtils.hlsli:45-58
  if (x > 0.f)
    return det_atan(y / x);
  else if (x < 0.f && y >= 0.f)
    return det_atan(y / x) + kPi;
  else if (x < 0.f && y < 0.f)
    return det_atan(y / x) - kPi;
  else if (x == 0.f && y > 0.f)
    return kPi / 2.f;
  else if (x == 0.f && y < 0.f)
    return -kPi / 2.f;
  return 0.f;
}

// lat-long lookup of PrimaryMiss (CP lightProbeGBuffer.rt.hlsl:63-74): texel of the RGBA32F map, 0 outside
BD f3 envLookup(const float* envMap, uint32_t envW, uint32_t envH, f3 d) {
  const f3 pd = normalize(d);
  const float u = (1.f + atan2_WAR(pd.x, -pd.z) * kInvPi) * 0.5f;
  const float v = det_acos(pd.y) * kInvPi;
  const uint32_t ex = (uint32_t)(u * (float)envW), ey = (uint32_t)(v * (float)envH);
  f3 c = mk(0);
  if (ex < envW && ey < envH) c = ld3(envMap + ((size_t)ey * envW + ex) * 4);
  return c;
}

// HINT_ONLY: the primary ray of EVERY frame pixel (G.Np = W * H, no pixel list), and only its occluder-hint word is
// written — what a context that renders a band or one rank's stripes runs when the camera has moved, so that its
// light-tracing rays, which may aim at any pixel of the frame, find a hint outside its own rows too.
template <bool COUNT, bool HINT_ONLY = false>
__global__ __launch_bounds__(kWave) void gbuffer_kernel(SceneDev S, GBufferDev G) {
  BDPT_ONE_WAVE_PER_GROUP();
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t p = blockIdx.x * kWave + threadIdx.x;
  if (p >= G.Np) return;
  const size_t pix = HINT_ONLY ? (size_t)p : (size_t)G.pix[p];
  const uint32_t y = (uint32_t)(pix / G.W), x = (uint32_t)(pix - (size_t)y * G.W);

  const f3 U = ld3(G.cam.cameraU), V = ld3(G.cam.cameraV), Wv = ld3(G.cam.cameraW), camPos = ld3(G.cam.posW);
  float pcx = ((float)x + G.gp.pixelJitter[0]) / (float)G.W;
  float pcy = ((float)y + G.gp.pixelJitter[1]) / (float)G.H;
  float ndx = 2.0f * pcx + -1.0f;
  float ndy = -2.0f * pcy + 1.0f;
  f3 rayDir = U * ndx + V * ndy + Wv;
  rayDir = rayDir / length(Wv);
  f3 focalPoint = camPos + rayDir * G.gp.focalLen;
  uint32_t randSeed = initRand(x + y * G.W, G.gp.frameCount);
  float r0 = nextRand(randSeed);
  float r1 = nextRand(randSeed);
  float sn, cs;
  det_sincos2pi(r0, sn, cs);
  float lr = G.gp.lensRadius * r1;
  float lu = cs * lr, lv = sn * lr;
  f3 randomOrig = camPos + normalize(U) * lu + normalize(V) * lv;
  f3 o = G.gp.useThinLens ? randomOrig : camPos;
  f3 d = normalize(G.gp.useThinLens ? (focalPoint - randomOrig) : rayDir);

  uint32_t nNodes = 0, nTris = 0;
  Hit h = traverse<1, COUNT>(S, o, d, 0.0f, 1e+38f, s_stack + threadIdx.x, nNodes, nTris);
  if (COUNT) {
    addCount(G.counters, C_RAYS_PRIMARY, 1);
    addCount(G.counters, C_NODE_CLOSEST, nNodes);
    addCount(G.counters, C_TRI_CLOSEST, nTris);
  }
  float4* oP = reinterpret_cast<float4*>(G.gb.worldPosition);
  if (G.hintPix) G.hintPix[pix] = (h.prim < 0) ? kNoHint : h.rec;  // occluder hint of light-tracing rays aimed at this pixel
  if (HINT_ONLY) return;
  if (h.prim < 0) {
    const f3 c = G.gp.envMap ? envLookup(G.gp.envMap, G.gp.envWidth, G.gp.envHeight, d) : ld3(G.gp.envColor);
    oP[pix] = make_float4(0, 0, 0, 0);
    packHalf4(G.gb.worldNormal, pix, 0, 0, 0, 0);
    packHalf4(G.gb.materialDiffuse, pix, c.x, c.y, c.z, 1.0f);
    packHalf4(G.gb.materialSpecRough, pix, 0, 0, 0, 0);
    packHalf4(G.gb.materialExtraParams, pix, 0, 0, 0, 0);
    packHalf4(G.gb.emissive, pix, 0, 0, 0, 0);
    return;
  }
  Shading sd = shadeHit<true>(S, (uint32_t)h.prim, h.u, h.v, camPos);
  oP[pix] = make_float4(sd.posW.x, sd.posW.y, sd.posW.z, 1.0f);
  packHalf4(G.gb.worldNormal, pix, sd.N.x, sd.N.y, sd.N.z, length(sd.posW - camPos));
  packHalf4(G.gb.materialDiffuse, pix, sd.diffuse.x, sd.diffuse.y, sd.diffuse.z, sd.opacity);
  packHalf4(G.gb.materialSpecRough, pix, sd.specular.x, sd.specular.y, sd.specular.z, sd.linearRoughness);
  packHalf4(G.gb.materialExtraParams, pix, sd.IoR, 0, 0, 0);
  packHalf4(G.gb.emissive, pix, sd.emissive.x, sd.emissive.y, sd.emissive.z, 0);
}

// ------------------------------------------------------------------------------------------------
// Occluder hints (device_trace.hpp recOccludes).  Two kinds of any-hit rays end at a point the whole frame shares:
//   * a light-tracing ray ends at the camera: whatever lies beyond the surface the camera sees through the target pixel
//     is hidden by that surface, so the triangle the pixel's PRIMARY ray hit is tried first (FrameDev::hintPix, written
//     by the G-buffer pass of the same context);
//   * a next-event ray ends at a point / spot light: the nearest triangle the light sees in that direction hides
//     everything behind it, so a cube map of nearest triangles per light (SceneDev::lightMap, built once per scene by
//     closest-hit rays from the light) supplies the triangle to try.
// A hint only ever picks WHICH triangle is tested first; a ray its hint does not occlude is traced as before, so the
// image does not depend on the map's resolution, on a stale G-buffer, or on there being hints at all.
// ------------------------------------------------------------------------------------------------
// cube-map texel of direction v (from the light): face = 2 * major axis + (negative ? 1 : 0), the two other components
// over the major one mapped from [-1, 1] to [0, res)
BD uint32_t cubeTexel(f3 v, uint32_t res) {
  const float ax = fabsf(v.x), ay = fabsf(v.y), az = fabsf(v.z);
  uint32_t face;
  float ma, uc, vc;
  if (ax >= ay && ax >= az) {
    face = v.x < 0.0f ? 1u : 0u;
    ma = ax;
    uc = v.y;
    vc = v.z;
  } else if (ay >= az) {
    face = v.y < 0.0f ? 3u : 2u;
    ma = ay;
    uc = v.x;
    vc = v.z;
  } else {
    face = v.z < 0.0f ? 5u : 4u;
    ma = az;
    uc = v.x;
    vc = v.y;
  }
  if (!(ma > 0.0f)) return kNoHint;  // zero or NaN direction
  const float fr = (float)res;
  const float fu = (uc / ma * 0.5f + 0.5f) * fr, fv = (vc / ma * 0.5f + 0.5f) * fr;
  const uint32_t iu = fu >= fr ? res - 1u : (uint32_t)(fu < 0.0f ? 0.0f : fu), iv = fv >= fr ? res - 1u : (uint32_t)(fv < 0.0f ? 0.0f : fv);
  return (face * res + iv) * res + iu;
}
BD uint32_t lightHint(const SceneDev& S, int light, f3 lightPos, f3 pos) {
  if (!S.lightMap) return kNoHint;
  const uint32_t t = cubeTexel(pos - lightPos, S.lightMapRes);
  if (t == kNoHint) return kNoHint;
  return S.lightMap[(size_t)light * 6u * S.lightMapRes * S.lightMapRes + t];
}

__global__ __launch_bounds__(kWave) void light_map_kernel(SceneDev S, uint32_t* __restrict__ maps, uint32_t res) {
  BDPT_ONE_WAVE_PER_GROUP();
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t perLight = 6u * res * res;
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  if (i >= perLight * S.numLights) return;
  const uint32_t light = i / perLight, t = i - light * perLight;
  const bdpt_light& l = S.sc->lights[light];
  uint32_t out = kNoHint;
  if (l.type != BDPT_LIGHT_DIRECTIONAL) {
    const uint32_t face = t / (res * res), iv = (t / res) % res, iu = t % res;
    const float uc = (((float)iu + 0.5f) / (float)res) * 2.0f - 1.0f, vc = (((float)iv + 0.5f) / (float)res) * 2.0f - 1.0f;
    const float m = (face & 1u) ? -1.0f : 1.0f;
    const f3 d = (face < 2u) ? mk(m, uc, vc) : ((face < 4u) ? mk(uc, m, vc) : mk(uc, vc, m));
    uint32_t nNodes = 0, nTris = 0;
    const Hit h = traverse<0, false>(S, ld3(l.posW), d, 0.0f, 1.0e38f, s_stack + threadIdx.x, nNodes, nTris);
    if (h.prim >= 0) out = h.rec;
  }
  maps[i] = out;
}

// ------------------------------------------------------------------------------------------------
// init_paths: eye vertex 1 from the G-buffer, light vertex 0 from sampleLight, valid-pixel queue
// (BDPTMain.rt.hlsl:51-103, 124-135; sampleLight BDPTUtils.hlsli:140-152)
// ------------------------------------------------------------------------------------------------
template <bool GGX>
__global__ __launch_bounds__(kWave) void init_paths_kernel(SceneDev S, FrameDev F, PathBuf P) {
  BDPT_ONE_WAVE_PER_GROUP();
  const uint32_t p = blockIdx.x * kWave + threadIdx.x;
  const bool inTile = p < P.Np;
  const size_t pix = inTile ? P.pix[p] : 0;
  bool geom = false;
  float4* out4 = reinterpret_cast<float4*>(F.out);
  if (inTile) {
    const float4 wp = reinterpret_cast<const float4*>(F.gb.worldPosition)[pix];
    float dr, dg, db, da;
    unpackHalf4(F.gb.materialDiffuse, pix, dr, dg, db, da);
    geom = (wp.w != 0.0f);
    if (!geom) {
      out4[pix] = make_float4(dr, dg, db, 1.0f);  // :62-66
      P.eyeLast[p] = 0;
    } else {
      float nx, ny, nz, nw, sr, sg, sb, sa, er, eg, eb, ea;
      unpackHalf4(F.gb.worldNormal, pix, nx, ny, nz, nw);
      unpackHalf4(F.gb.materialSpecRough, pix, sr, sg, sb, sa);
      unpackHalf4(F.gb.emissive, pix, er, eg, eb, ea);
      const f3 camPos = ld3(F.cam.posW);
      const f3 worldPos = mk(wp.x, wp.y, wp.z), worldNorm = mk(nx, ny, nz), dif = mk(dr, dg, db), spec = mk(sr, sg, sb);
      const float roughness = sa * sa;
      const f3 V = normalize(camPos - worldPos);
      uint32_t seed = initRand((uint32_t)pix, F.p.frameCount);
      P.seedE[p] = seed;  // every eye bounce draws from this state by value (quirk 1)
      f3 outDir;
      float pdf;
      bool isSpec;
      f3 thr = sampleBRDF<GGX>(seed, worldNorm, worldNorm, V, dif, spec, roughness,
                               (F.p.flags & BDPT_PARAM_SPECULAR_FROM_LOBE) != 0, outDir, pdf, isSpec);
      Vtx v;
      v.color = thr;
      v.pos = worldPos;
      v.N = worldNorm;
      v.V = V;
      v.dif = dif;
      v.spec = spec;
      v.rough = roughness;
      v.isSpec = isSpec;
      v.pdf = pdf;
      storeVtx(P, PATH_EYE, 1, p, v);
      float* rd = P.rayDir + (size_t)(PATH_EYE * 3) * P.Np + p;
      rd[0] = outDir.x;
      rd[P.Np] = outDir.y;
      rd[2 * (size_t)P.Np] = outDir.z;

      // sampleLight (seed continues from seed0: the eye walk never advanced it, quirk 1)
      const int lightsCount = (int)S.numLights;
      int index = (int)(nextRand(seed) * (float)lightsCount);
      if (index > lightsCount - 1) index = lightsCount - 1;
      const bdpt_light& l = S.sc->lights[index];
      f3 lightDir;
      if (l.type == BDPT_LIGHT_DIRECTIONAL)
        lightDir = ld3(l.dirW);
      else
        lightDir = sampleUnitSphere(seed);
      lightDir = getCosHemisphereSample(seed, lightDir);
      Vtx lv = zeroVtx();
      lv.pos = ld3(l.posW);
      lv.color = ld3(l.intensity);
      lv.pdf = 1.0f / (float)lightsCount;  // lightPath[0].pdfForward, BDPTMain.rt.hlsl:132
      storeVtx(P, PATH_LIGHT, 0, p, lv);
      float* rl = P.rayDir + (size_t)(PATH_LIGHT * 3) * P.Np + p;
      rl[0] = lightDir.x;
      rl[P.Np] = lightDir.y;
      rl[2 * (size_t)P.Np] = lightDir.z;
      P.seedL[p] = seed;

      const uint8_t D = (uint8_t)F.p.maxDepth;
      P.eyeLast[p] = D < 1 ? 1 : D;
      P.lightLast[p] = D;
      P.lightReal[p] = D;
      // gOutput cleared (BDPTPass.cpp:73) then += emissive (:155-158)
      const bool em = (er > 0.0f) || (eg > 0.0f) || (eb > 0.0f);
      out4[pix] = em ? make_float4(0.0f + er, 0.0f + eg, 0.0f + eb, 0.0f + ea) : make_float4(0, 0, 0, 0);
    }
  }
  wavePush(geom, p, P.queue[0], P.qcount, P.pathSubCap);
}

// ------------------------------------------------------------------------------------------------
// walk: both random walks of the frame in ONE persistent launch (BDPTMain.rt.hlsl:106-112 eye,
// :138-145 light; shootRay globalIlluminationRay.hlsli:1-12; RayClosestHit / RayMiss :14-45;
// updateRayData RayPathData.hlsli:88-109).
//
// A wave owns up to 127 sub-paths at a time: 64 traversal slots (one per lane, state in registers,
// stack in LDS) plus a 128-entry LDS pool holding PARKED hit records (filled from the bottom) and
// READY next-bounce rays (filled from the top).  Per iteration of the wave:
//   1. 64 or more parked records -> the hit/miss shader runs for 64 of them with every lane busy
//      (vertex fetch, material + texture decode, sampleBRDF, vertex k+1 stored to its SoA planes);
//      survivors become ready rays.  Which lane shades which record has nothing to do with the ray the
//      lane is traversing: its traversal registers simply stay put meanwhile.
//   2. empty lanes (kRefillIdle or more) take ready rays, then new sub-paths from the valid-pixel lists;
//   3. one while-while traversal round; lanes that found their closest hit park the record and are empty.
// Hit records, per-bounce path queues and the 2(2D-1) launch boundaries of a bounce-synchronous
// wavefront never exist; shading is dense although traversal lengths diverge.
//
// Pool bound: new sub-paths are only fetched when no ray is ready, at which point fewer than 64 records
// are parked, so traversing + parked + ready <= 127 always.
// Work list: virtual list vq in [0, 2*kNumSubQueues): pixel list vq % kNumSubQueues of the valid-pixel
// queue, path vq / kNumSubQueues (eye first).  `head` holds one fetch cursor per virtual list.
// A miss stores the reference's "ghost" vertex (quirk 2): colour 0, previous payload geometry.
// A path id packs pixel (24 bits), path (bit 24) and vertex index k (bits 25-29): tiles are < 2^24 pixels; bit 30 of
// a PARKED record's id says "miss" (the record then carries the ray's direction instead of primitive + barycentrics).
// EXT (BDPT_PARAM_ENV_ON_MISS / _EMISSIVE_HITS; build definitions, include/bdpt.h) is its own instantiation, so the
// reference's path pays nothing for it: the eye walk's shading pass adds what the ray found where it ended —
// environment radiance on a miss, emissive on a hit — to the path's own pixel, in bounce order (a sub-path is shaded
// by one lane at a time, and init_paths has written the pixel before this launch starts).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kPoolEntries = 128;
#ifndef BDPT_WALK_ORDER
#define BDPT_WALK_ORDER 1  // child order of the walk's closest-hit traversal (device_trace.hpp nodeStep)
#endif
#ifndef BDPT_WALK_SHADE_MIN
#define BDPT_WALK_SHADE_MIN 64  // parked hit records that trigger a shading pass (64 = every lane shades)
#endif
#ifndef BDPT_WALK_LEAF_WAIT
#define BDPT_WALK_LEAF_WAIT 1  // deferred leaf phase (device_trace.hpp); 0 = plain while-while
#endif
#ifndef BDPT_WALK_NODE_BURST
#define BDPT_WALK_NODE_BURST 3
#endif
#ifndef BDPT_WALK_CHUNK
#define BDPT_WALK_CHUNK 64  // sub-paths per fetch from the valid-pixel lists
#endif
#ifndef BDPT_WALK_REFILL
#define BDPT_WALK_REFILL BDPT_REFILL_IDLE  // empty lanes before the walk kernel refills
#endif
BD uint32_t packPath(uint32_t p, int path, int k) { return p | ((uint32_t)path << 24) | ((uint32_t)k << 25); }

#ifndef BDPT_WALK_WAVES_PER_EU
#define BDPT_WALK_WAVES_PER_EU 4
#endif
#ifndef BDPT_WALK_STACK_LDS
#define BDPT_WALK_STACK_LDS kStackEntries
#endif
constexpr int kWalkStackLds = BDPT_WALK_STACK_LDS;
constexpr uint32_t kParkedMiss = 1u << 30;
template <bool GGX, bool COUNT, bool EXT>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(BDPT_WALK_WAVES_PER_EU, 8))) void walk_kernel(SceneDev S, FrameDev F, PathBuf P, uint32_t* __restrict__ head) {
  BDPT_ONE_WAVE_PER_GROUP();
  __shared__ int s_stack[kWalkStackLds * kWave];
  __shared__ uint4 s_pool[kPoolEntries];
  int* stk = s_stack + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned long long laneBelow = (1ull << lane) - 1ull;
  const int D = (int)F.p.maxDepth;
  const bool fromLobe = (F.p.flags & BDPT_PARAM_SPECULAR_FROM_LOBE) != 0;
  // the eye walk extends vertices 1..D-1 (none when D < 2), the light walk vertices 0..D-1
  const uint32_t firstV = (D >= 2) ? 0u : kNumSubQueues, numV = 2u * kNumSubQueues - firstV;
  bool trav = false;   // this lane holds a ray
  uint32_t id = 0;     // its path id
  TravState T;
  travInit(T, mk(0), mk(0), F.p.minT, 1.0e38f);
  T.cur = kDone;
  uint32_t nNodes = 0, nTris = 0, nAlpha = 0;
  uint32_t nEye = 0, nLight = 0, nParked = 0, nReady = 0;  // wave-uniform
  uint32_t vq = firstV + blockIdx.x % numV, tried = 0, chunkPos = 0, chunkEnd = 0, chunk = BDPT_WALK_CHUNK;
  bool exhausted = false;
  const uint32_t wavesPerList = (gridDim.x + numV - 1) / numV;
  for (;;) {
    unsigned long long travMask = __ballot(trav);
    // ---- 1. hit / miss shaders, one parked record per lane ------------------------------------------
    // (also when nothing else can make progress: the last records of the wave are shaded short-handed)
    const bool flush = (travMask == 0ull) && nReady == 0 && exhausted && nParked > 0;
    if (nParked >= (uint32_t)BDPT_WALK_SHADE_MIN || flush) {
      const uint32_t n = nParked < (uint32_t)kWave ? nParked : (uint32_t)kWave;
      nParked -= n;
      const bool act = (uint32_t)lane < n;
      uint4 rec = make_uint4(0, 0, 0, 0);
      if (act) rec = s_pool[nParked + (uint32_t)lane];
      __syncthreads();  // the slots may be overwritten by ready rays below
      const uint32_t p = rec.x & 0xffffffu;
      const int path = (int)((rec.x >> 24) & 1u), k = (int)((rec.x >> 25) & 31u);
      nEye += (uint32_t)__popcll(__ballot(act && path == PATH_EYE));
      nLight += (uint32_t)__popcll(__ballot(act && path == PATH_LIGHT));
      bool survive = false;
      f3 L = mk(0);
      if (act) {
        const bool miss = EXT ? (rec.x & kParkedMiss) != 0u : (int)rec.y < 0;
        const int prim = miss ? -1 : (int)rec.y;
        const f3 o = ldPlane3(P, path, k, F_POS, p);
        // what the eye ray that left vertex k found where it ended (EXT): added to the path's pixel below
        f3 found = mk(0);
        bool haveFound = false;
        if (prim >= 0) {
          const uint32_t seed = (path == PATH_EYE) ? P.seedE[p] : P.seedL[p];
          const f3 thr = ldPlane3(P, path, k, F_COL, p);
          Shading sd = shadeHit<false>(S, (uint32_t)prim, __uint_as_float(rec.z), __uint_as_float(rec.w), o);  // V points at WorldRayOrigin()
          float pdf;
          bool isSpec;
          f3 w = sampleBRDF<GGX>(seed, sd.N, sd.N, sd.V, sd.diffuse, sd.specular, sd.roughness, fromLobe, L, pdf, isSpec);
          Vtx v;
          v.color = thr * w;
          v.pos = sd.posW;
          v.N = sd.N;
          v.V = sd.V;
          v.dif = sd.diffuse;
          v.spec = sd.specular;
          v.rough = sd.roughness;
          v.isSpec = isSpec;
          v.pdf = pdf;
          storeVtx(P, path, k + 1, p, v);
          survive = (k + 2 <= D);  // k + 1 < maxK
          if (EXT && path == PATH_EYE && (F.p.flags & BDPT_PARAM_EMISSIVE_HITS) &&
              (sd.emissive.x > 0.0f || sd.emissive.y > 0.0f || sd.emissive.z > 0.0f)) {
            found = thr * sd.emissive;
            haveFound = true;
          }
        } else {
          if (EXT && path == PATH_EYE && (F.p.flags & BDPT_PARAM_ENV_ON_MISS)) {
            const f3 dir = mk(__uint_as_float(rec.y), __uint_as_float(rec.z), __uint_as_float(rec.w));
            const f3 env = F.envMap ? envLookup(F.envMap, F.envW, F.envH, dir) : ld3(F.envColor);
            found = ldPlane3(P, path, k, F_COL, p) * env;
            haveFound = true;
          }
          Vtx g = zeroVtx();
          if (path == PATH_EYE && k == 1) {
            g.pos = o;  // payload still holds initPayload's values (RayPathData.hlsli:69-86)
          } else {
            loadSurf(P, path, k, p, g);
            g.V = ldPlane3(P, path, k, F_V, p);
            if (path == PATH_LIGHT && k == 0) g.pdf = 0.0f;  // initPayload: pdfForward = 0
          }
          g.color = mk(0);
          storeVtx(P, path, k + 1, p, g);
          if (path == PATH_EYE) {
            P.eyeLast[p] = (uint8_t)(k + 1);
          } else {
            P.lightLast[p] = (uint8_t)(k + 1);
            P.lightReal[p] = (uint8_t)k;
          }
        }
        if (EXT && haveFound) {  // path-tracing strategy of k + 1 edges: uniform 1/edges, clamped, no saturate (as NEE terms)
          f3 term = clampVec(found / (float)(k + 1), F.p.clampUpper);
          if (isnan3(term)) term = mk(0);
          float4* out4 = reinterpret_cast<float4*>(F.out);
          const size_t pix = P.pix[p];
          float4 acc = out4[pix];
          acc.x = acc.x + term.x;
          acc.y = acc.y + term.y;
          acc.z = acc.z + term.z;
          acc.w = acc.w + 1.0f;
          out4[pix] = acc;
        }
      }
      // survivors -> ready rays (origin = the stored vertex k+1, re-read at pick-up)
      const unsigned long long sm = __ballot(survive);
      if (survive) {
        const uint32_t slot = kPoolEntries - 1u - nReady - (uint32_t)__popcll(sm & laneBelow);
        s_pool[slot] = make_uint4(packPath(p, path, k + 1), __float_as_uint(L.x), __float_as_uint(L.y), __float_as_uint(L.z));
      }
      nReady += (uint32_t)__popcll(sm);
      // this wave's later loads of the vertices it just stored must see them (same CU: ordering is enough)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    }
    // ---- 2. empty lanes take ready rays, then new sub-paths -------------------------------------------
    const int empty = 64 - __popcll(travMask);
    if ((empty >= BDPT_WALK_REFILL || travMask == 0ull) && (nReady > 0 || !exhausted)) {
      const unsigned long long emptyMask = ~travMask;
      const uint32_t rank = (uint32_t)__popcll(emptyMask & laneBelow);
      const uint32_t fromReady = ((uint32_t)empty < nReady) ? (uint32_t)empty : nReady;
      bool got = false;
      uint32_t nid = 0;
      f3 dir = mk(0);
      if (!trav && rank < fromReady) {
        const uint4 r = s_pool[kPoolEntries - nReady + rank];
        nid = r.x;
        dir = mk(__uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
        got = true;
      }
      __syncthreads();  // the pool slots just read may be reused by parked records
      nReady -= fromReady;
      uint32_t want = (uint32_t)empty - fromReady;  // lanes still empty: new sub-paths (only reached with nReady == 0)
      uint32_t taken = fromReady;
      while (want > 0 && !exhausted) {
        while (chunkPos >= chunkEnd && !exhausted) {  // wave-uniform loop: take a new chunk
          const uint32_t nq = P.qcount[(vq % kNumSubQueues) * kCursorStride];
          uint32_t base = nq;
          if (__hip_atomic_load(&head[vq * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nq) {
            // A sub-path keeps its lane for up to D rays, so the list is handed out in small pieces (BDPT_WALK_CHUNK):
            // with 256 per fetch the last pieces kept single waves busy long after the rest of the grid had drained.
            uint32_t share = (nq / wavesPerList + 15u) & ~15u;
            chunk = share < 16u ? 16u : (share > (uint32_t)BDPT_WALK_CHUNK ? (uint32_t)BDPT_WALK_CHUNK : share);
            if (lane == 0) base = atomicAdd(&head[vq * kCursorStride], chunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          }
          if (base < nq) {
            chunkPos = base;
            chunkEnd = (base + chunk < nq) ? base + chunk : nq;
            tried = 0;
          } else {
            vq = (vq + 1 == 2u * kNumSubQueues) ? firstV : vq + 1;
            if (++tried >= numV) exhausted = true;
          }
        }
        if (exhausted) break;
        const uint32_t avail = chunkEnd - chunkPos;
        const uint32_t take = (want < avail) ? want : avail;
        if (!trav && !got && rank >= taken && rank < taken + take) {
          const uint32_t p = P.queue[0][(vq % kNumSubQueues) * P.pathSubCap + chunkPos + (rank - taken)];
          const int path = (int)(vq / kNumSubQueues);
          const float* rd = P.rayDir + (size_t)(path * 3) * P.Np + p;
          nid = packPath(p, path, (path == PATH_EYE) ? 1 : 0);
          dir = mk(rd[0], rd[P.Np], rd[2 * (size_t)P.Np]);
          got = true;
        }
        chunkPos += take;
        taken += take;
        want -= take;
      }
      if (got) {
        id = nid;
        travInit(T, ldPlane3(P, (int)((nid >> 24) & 1u), (int)(nid >> 25), F_POS, nid & 0xffffffu), dir, F.p.minT, 1.0e38f);
        trav = true;
      }
      travMask = __ballot(trav);
    }
    if (travMask == 0ull) {
      if (nParked == 0 && nReady == 0 && exhausted) break;
      continue;  // parked records are flushed (or ready rays picked up) at the top
    }
    // ---- 3. traversal for the lanes that hold a ray
    bool finished = false;
#if BDPT_WALK_LEAF_WAIT > 0
    // node visits in short bursts; a lane that reaches a leaf (or runs out of stack) waits, and the leaves are intersected
    // once half of the lanes that hold a ray are waiting or no lane can take a node visit (device_trace.hpp, trace_shadow_kernel)
    if (trav) {
#pragma unroll 1
      for (int kk = 0; kk < BDPT_WALK_NODE_BURST && T.cur >= 0; kk++) {
        if (COUNT) nNodes++;
        nodeStep<BDPT_WALK_ORDER, kWalkStackLds>(S, T, stk);
      }
    }
    {
      const unsigned long long waitMask = __ballot(trav && T.cur < 0), nodeMask = __ballot(trav && T.cur >= 0);
      const int waitNeed = (__popcll(waitMask | nodeMask) * BDPT_LEAF_WAIT_FRAC8 + 7) >> 3;
      if ((int)__popcll(waitMask) >= waitNeed || nodeMask == 0ull) {
        if (trav && T.cur < 0) {
          finished = (T.cur == kDone);
          if (!finished) {
            finished = leafStep<0, COUNT>(S, T, nTris, nAlpha);
            if (!finished) {
              T.cur = travPop<kWalkStackLds>(S, T, stk);
              finished = (T.cur == kDone);
            }
          }
        }
      }
    }
#else
    if (trav) {
      while (T.cur >= 0) {
        if (COUNT) nNodes++;
        nodeStep<BDPT_WALK_ORDER, kWalkStackLds>(S, T, stk);
      }
      finished = (T.cur == kDone);
      if (!finished) {
        finished = leafStep<0, COUNT>(S, T, nTris, nAlpha);
        if (!finished) {
          T.cur = travPop<kWalkStackLds>(S, T, stk);
          finished = (T.cur == kDone);
        }
      }
    }
#endif
    const unsigned long long finMask = __ballot(finished);
    if (finMask) {
      if (finished) {
        const bool miss = T.best.prim < 0;
        s_pool[nParked + (uint32_t)__popcll(finMask & laneBelow)] =
            (EXT && miss) ? make_uint4(id | kParkedMiss, __float_as_uint(T.d.x), __float_as_uint(T.d.y), __float_as_uint(T.d.z))
                          : make_uint4(id, (uint32_t)T.best.prim, __float_as_uint(T.best.u), __float_as_uint(T.best.v));
        trav = false;
      }
      nParked += (uint32_t)__popcll(finMask);
      __syncthreads();
    }
  }
  if (lane == 0) {
    if (nEye) atomicAdd(&F.counters->v[blockIdx.x % kCounterShards][C_RAYS_EYE], (unsigned long long)nEye);
    if (nLight) atomicAdd(&F.counters->v[blockIdx.x % kCounterShards][C_RAYS_LIGHT], (unsigned long long)nLight);
  }
  if (COUNT) {
    waveAddCount(F.counters, C_NODE_CLOSEST, nNodes);
    waveAddCount(F.counters, C_TRI_CLOSEST, nTris);
    waveAddCount(F.counters, C_ALPHA_CLOSEST, nAlpha);
  }
}

// ------------------------------------------------------------------------------------------------
// gen_shadow: every shadow ray of the frame, generated per valid pixel in the reference's order
// and appended (wave-compacted) to one SoA ray queue together with the clamped contribution it
// gates.  slotRay[slot][p] remembers which ray (if any) belongs to which term so that the gather
// stage can add the terms in the reference's order.
//   NEE        BDPTMain.rt.hlsl:161-167, evalDirect MaterialUtils.hlsli:93-103/149-184/288-307
//   splat      BDPTMain.rt.hlsl:171-208, getLaunchIndexFromDirection BDPTUtils.hlsli:129-138
//   connection BDPTMain.rt.hlsl:212-233, evalGWithoutV / getUnweightedContribution BDPTUtils.hlsli:172-224
//              (uniform 1/totalLength weights; aL indexes the light path with cameraIndex-1, sic :198)
// A term whose value is already exactly 0 after clampVec gets no ray: NEE terms then add 0 either
// way; connection terms only matter through the per-write saturate, which the gather stage
// reproduces with at most a few lazily traced rays.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// MIS weights — getWeightPower / getWeightLinear (BDPT/BDPTUtils.hlsli:226-278), which the reference
// defines but never calls (it weights every strategy 1/k).  Behind BDPT_PARAM_MIS_POWER / _LINEAR:
//   pE(i) = cameraPath[0].pdfForward * prod_{x=1..i} cameraPath[x].pdfForward * evalGWithoutV(x-1, x)
//   pL(j) likewise on the light path;  w(c,l) = f(pE(c) pL(l)) / sum_{i+j=c+l} f(pE(i) pL(j)),  f = x or x^2
// (`if (i == cameraIndex, j == lightIndex)` at :245/:272 is a comma expression; since i + j is fixed it
// selects the same term as the intended `&&`.)  Faithful consequence worth knowing: lightPath[0].N = 0
// makes evalGWithoutV(light[0], light[1]) = 0, so pL(j >= 1) = 0 and every strategy that uses a light
// vertex gets weight 0 while NEE gets 1 — the estimator degenerates to path tracing with NEE.
// ------------------------------------------------------------------------------------------------
BD float evalGWithoutVPos(f3 posA, f3 nA, f3 posB, f3 nB) {  // BDPTUtils.hlsli:172-184
  const f3 vecAB = posB - posA;
  const float invLengthAB = 1.0f / length(vecAB);
  const f3 dirAB = vecAB * invLengthAB;
  const float cosA = fabsf(dot(nA, dirAB));
  const float cosB = fabsf(dot(nB, dirAB));
  return cosA * cosB * invLengthAB * invLengthAB;
}

__global__ __launch_bounds__(kWave) void mis_prefix_kernel(FrameDev F, PathBuf P) {
  BDPT_ONE_WAVE_PER_GROUP();
  bool act = false;
  uint32_t i = 0;
  if (!queueChunk(P.qcount, P.pathSubCap, act, i)) return;
  if (!act) return;
  const uint32_t p = P.queue[0][i];
  const int D = (int)F.p.maxDepth;
  const int eyeLast = P.eyeLast[p], lightLast = P.lightLast[p];
  // eye side: vertex 0 is the camera (pos, N = normalize(cameraW), pdfForward 1), BDPTMain.rt.hlsl:90-93
  f3 prevPos = ld3(F.cam.posW), prevN = normalize(ld3(F.cam.cameraW));
  float pE = 1.0f;
  P.misE[p] = pE;
  for (int x = 1; x <= D; x++) {
    f3 pos = mk(0), N = mk(0);
    float pdf = 0.0f;
    if (x <= eyeLast) {
      pos = ldPlane3(P, PATH_EYE, x, F_POS, p);
      N = ldPlane3(P, PATH_EYE, x, F_N, p);
      pdf = ldPlane1(P, PATH_EYE, x, F_PDF, p);
    }
    pE *= pdf * evalGWithoutVPos(prevPos, prevN, pos, N);
    P.misE[(size_t)x * P.Np + p] = pE;
    prevPos = pos;
    prevN = N;
  }
  prevPos = ldPlane3(P, PATH_LIGHT, 0, F_POS, p);
  prevN = ldPlane3(P, PATH_LIGHT, 0, F_N, p);
  float pL = ldPlane1(P, PATH_LIGHT, 0, F_PDF, p);
  P.misL[p] = pL;
  for (int x = 1; x <= D; x++) {
    f3 pos = mk(0), N = mk(0);
    float pdf = 0.0f;
    if (x <= lightLast) {
      pos = ldPlane3(P, PATH_LIGHT, x, F_POS, p);
      N = ldPlane3(P, PATH_LIGHT, x, F_N, p);
      pdf = ldPlane1(P, PATH_LIGHT, x, F_PDF, p);
    }
    pL *= pdf * evalGWithoutVPos(prevPos, prevN, pos, N);
    P.misL[(size_t)x * P.Np + p] = pL;
    prevPos = pos;
    prevN = N;
  }
}

BD float misWeight(const PathBuf& P, uint32_t p, int cameraIndex, int lightIndex, bool power) {
  const int totalLength = cameraIndex + lightIndex;
  float totalPdf = 0.0f, currentPdf = 1.0f;
  for (int i = 0; i <= totalLength; i++) {
    const int j = totalLength - i;
    const float pE = P.misE[(size_t)i * P.Np + p], pL = P.misL[(size_t)j * P.Np + p];
    const float term = power ? (pE * pE * pL * pL) : (pE * pL);
    totalPdf += term;
    if (j == lightIndex) currentPdf = term;
  }
  return currentPdf / totalPdf;
}
// the factor a term is scaled by: the reference's uniform 1/k, or the MIS weight when switched on
BD f3 applyStrategyWeight(const FrameDev& F, const PathBuf& P, uint32_t p, f3 v, int k, int cameraIndex, int lightIndex) {
  if (F.p.flags & (BDPT_PARAM_MIS_POWER | BDPT_PARAM_MIS_LINEAR))
    return v * misWeight(P, p, cameraIndex, lightIndex, (F.p.flags & BDPT_PARAM_MIS_POWER) != 0);
  return v / (float)k;
}

BD uint32_t emitRay(const PathBuf& P, int cls, bool active, f3 o, f3 d, float tmax, f3 contrib) {
  const unsigned long long mask = __ballot(active);
  uint32_t id = kNoRay;
  if (mask == 0ull) return id;
  const int lane = (int)(threadIdx.x & 63u);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  const uint32_t q = blockIdx.x % kNumRaySubQueues;
  if (lane == leader) base = atomicAdd(&P.rayCount[(size_t)cls * kRayCursorBlock + q * kCursorStride], (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, leader);
  if (active) {
    id = P.rayBase[cls] + q * P.raySubCap[cls] + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    const size_t c = P.rayCap;
    float* r = P.rayQ + id;
    r[0] = o.x;
    r[c] = o.y;
    r[2 * c] = o.z;
    r[3 * c] = d.x;
    r[4 * c] = d.y;
    r[5 * c] = d.z;
    r[6 * c] = tmax;
    float* q = P.rayContrib + id;
    q[0] = contrib.x;
    q[c] = contrib.y;
    q[2 * c] = contrib.z;
  }
  return id;
}

template <bool GGX>
BD void loadConnVtx(const PathBuf& P, int path, int k, int last, uint32_t p, Vtx& v) {
  if (k > last) {
    v = zeroVtx();
    return;
  }
  loadSurf<GGX>(P, path, k, p, v);
}

// Three kernels (NEE, splat, connection).  A lane owns one (pixel, vertex) pair: G = 8 (depth <= 8) or 16 lanes
// per pixel, lane g of the group handling term / camera length g.  A lane reads whole 96-byte vertex records, every
// record is fetched once per kernel, and no lane runs a loop of dependent (load -> evaluate -> append) rounds over a
// pixel's vertices: with one lane per pixel the three kernels took 2.2 / 0.6 / 2.6 ms on the bench frame
// (profiles/README.md), bound by those serial rounds and by re-reading the light vertices once per camera vertex.
template <int G>
BD bool queueGroup(const uint32_t* count, uint32_t subCap, bool& act, uint32_t& idx, int& g) {
  constexpr uint32_t kPix = kWave / G;  // pixels per wave
  const uint32_t q = blockIdx.x % kNumSubQueues, c0 = (blockIdx.x / kNumSubQueues) * kPix;
  const uint32_t n = count[q * kCursorStride];
  if (c0 >= n) return false;
  const uint32_t i = c0 + threadIdx.x / G;
  g = (int)(threadIdx.x % G);
  act = i < n;
  idx = q * subCap + i;
  return true;
}

template <bool GGX, int G>
__global__ __launch_bounds__(kWave) void gen_nee_kernel(SceneDev S, FrameDev F, PathBuf P) {
  BDPT_ONE_WAVE_PER_GROUP();
  bool act = false;  // inactive lanes still take part in the wave-collective emitRay
  uint32_t i = 0;
  int t = 0;
  if (!queueGroup<G>(P.qcount, P.pathSubCap, act, i, t)) return;
  const int D = (int)F.p.maxDepth;
  const bool firstOfPixel = act && t == 0;
  act = act && t < D;
  const uint32_t p = act ? P.queue[0][i] : 0u;
  const int eyeLast = act ? (int)P.eyeLast[p] : 0;
  const int lightsCount = (int)S.numLights;
  // term t uses the (t+1)-th draw after sampleLight: one draw per term, also for vertices that do not exist (App. A item 8)
  uint32_t seed = act ? P.seedL[p] : 0u;
  float r = 0.0f;
  for (int k = 0; k <= t; k++) r = nextRand(seed);
  bool emit = false, hinted = false;
  f3 pos = mk(0), L = mk(0), shade = mk(0);
  float distToLight = 0.0f;
  if (act && (t + 1) <= eyeLast) {
    int lightToSample = (int)(r * (float)lightsCount);
    if (lightToSample > lightsCount - 1) lightToSample = lightsCount - 1;
    Vtx v;
    loadSurf<GGX>(P, PATH_EYE, t + 1, p, v);
    pos = v.pos;
    f3 V = mk(0);
    if (GGX) V = ldPlane3(P, PATH_EYE, t + 1, F_V, p);
    const f3 prevColor = (t == 0) ? mk(1.0f) : ldPlane3(P, PATH_EYE, t, F_COL, p);  // cameraPath[t].color; [0] = 1
    f3 lightIntensity;
    getLightData(S.sc->lights[lightToSample], pos, L, lightIntensity, distToLight);
    f3 direct = directIfVisible<GGX>((float)lightsCount, L, lightIntensity, v.N, V, v.dif, v.spec, v.rough);
    shade = clampVec(applyStrategyWeight(F, P, p, prevColor * direct, t + 2, t + 1, 0), F.p.clampUpper);
    emit = !allZero(shade);
    // the nearest triangle the light sees towards this vertex is tried first: an occluded term adds nothing and needs no ray
    if (emit && S.sc->lights[lightToSample].type != BDPT_LIGHT_DIRECTIONAL &&
        recOccludes(S, lightHint(S, lightToSample, ld3(S.sc->lights[lightToSample].posW), pos), pos, L, F.p.minT, distToLight)) {
      emit = false;
      hinted = true;
    }
  }
  const uint32_t id = emitRay(P, RAY_TERMS, emit, pos, L, distToLight, shade);
  if (act) P.slotRay[(size_t)t * P.Np + p] = id;
  waveAddCount(F.counters, C_RAYS_NEE, emit ? 1u : 0u);
  waveAddCount(F.counters, C_HINT_NEE, hinted ? 1u : 0u);
  waveAddCount(F.counters, C_PIX_VALID, firstOfPixel ? 1u : 0u);
}

template <bool GGX, int G>
__global__ __launch_bounds__(kWave) void gen_splat_kernel(SceneDev S, FrameDev F, PathBuf P) {
  BDPT_ONE_WAVE_PER_GROUP();
  bool act = false;
  uint32_t i = 0;
  int t = 0;
  if (!queueGroup<G>(P.qcount, P.pathSubCap, act, i, t)) return;
  const int D = (int)F.p.maxDepth;
  act = act && t < D;
  const uint32_t p = act ? P.queue[0][i] : 0u;
  const int real = act ? (int)P.lightReal[p] : 0;
  const f3 camPos = ld3(F.cam.posW);
  const f3 U = ld3(F.cam.cameraU), Vc = ld3(F.cam.cameraV), Wc = ld3(F.cam.cameraW);
  const f3 cameraN = normalize(Wc);
  bool emit = false, hinted = false;
  f3 pos = mk(0), dirToCamera = mk(0), shade = mk(0);
  float disToCamera = 0.0f;
  uint32_t target = kNoRay;
  if (act && t < real) {
    Vtx lv;
    loadSurf<GGX>(P, PATH_LIGHT, t + 1, p, lv);
    pos = lv.pos;
    dirToCamera = normalize(camPos - lv.pos);
    disToCamera = length(camPos - lv.pos);
    if (dot(cameraN, dirToCamera) < 0) {
      float d1 = dot(dirToCamera, U) / dot(U, U);
      float d2 = dot(dirToCamera, Vc) / dot(Vc, Vc);
      float d3 = dot(dirToCamera, Wc) / dot(Wc, Wc);
      float nx = d1 / d3, ny = -d2 / d3;
      float px = nx * 0.5f + 0.5f, py = ny * 0.5f + 0.5f;
      float fx = rintf(px * (float)F.W - F.p.pixelJitter[0]);
      float fy = rintf(py * (float)F.H - F.p.pixelJitter[1]);
      const bool inside = (fx >= 0.0f && fx < (float)F.W && fy >= 0.0f && fy < (float)F.H);
      // The reference traces this ray whatever the value (its write saturates the target pixel) — but a target outside
      // the frame has no pixel to write (quirk 8: the out-of-range UAV write is dropped), so such a ray decides nothing
      // and is not traced.
      emit = inside;
      if (inside) {
        target = (uint32_t)splatIndex(F.sl, F.W, (uint32_t)(int)fx, (uint32_t)(int)fy);
        float theta1 = saturate(fabsf(dot(dirToCamera, cameraN)));
        float theta2 = saturate(fabsf(dot(dirToCamera, lv.N)));
        float invDisToCamera = 1.0f / disToCamera;
        float Gt = theta1 * theta2 * invDisToCamera * invDisToCamera;
        f3 vV = mk(0);
        if (GGX) vV = ldPlane3(P, PATH_LIGHT, t + 1, F_V, p);
        f3 fr = evalBRDF<GGX>(vV, normalize(camPos - lv.pos), lv.N, lv.N, lv.dif, lv.spec, lv.rough, lv.isSpec);
        f3 prevColor = ldPlane3(P, PATH_LIGHT, t, F_COL, p);
        shade = clampVec(applyStrategyWeight(F, P, p, (prevColor * fr) * Gt, t + 2, 0, t + 1), F.p.clampUpper);
        if (isnan3(shade)) shade = mk(0);
        // the surface the camera sees through the target pixel is tried first: a hidden vertex lands no splat and needs no ray
        if (F.hintPix && recOccludes(S, F.hintPix[(size_t)(uint32_t)(int)fy * F.W + (uint32_t)(int)fx], pos, dirToCamera, F.p.minT, disToCamera)) {
          emit = false;
          hinted = true;
        }
      }
    }
  }
  const uint32_t id = emitRay(P, RAY_TERMS, emit, pos, dirToCamera, disToCamera, shade);
  if (act) {
    P.slotRay[(size_t)(D + t) * P.Np + p] = id;
    P.splatPix[(size_t)t * P.Np + p] = target;
  }
  waveAddCount(F.counters, C_RAYS_SPLAT, emit ? 1u : 0u);
  waveAddCount(F.counters, C_HINT_SPLAT, hinted ? 1u : 0u);
}

// Connections.  Lane g of a pixel's group LOADS camera length g + 1 (its vertex record, the two path colours and the
// direction to its predecessor) and light vertex g (into LDS), so every vertex record leaves memory once.  The pairs
// themselves — camera length c in 1..D-1 with light length l in 1..D-c, D(D-1)/2 of them — form a triangle; handed out by
// camera length (round 2) the lane of c = 1 evaluated seven pairs while the lane of c = 7 evaluated one and every lane
// sat through all eight rounds: lane utilisation 0.50 in a kernel bound by VALU issue.  Now they are dealt to the group's
// lanes DENSELY: pair ordinal it * G + g goes to lane g in round it (ceil(D(D-1)/2G) rounds: 4 instead of 8 at depth 8),
// and the lane fetches the eye side of its pair from the lane that holds it (__shfl, 23 values) and the light side
// from LDS.  The arithmetic of a pair is what it was; only the order in which the rays are appended changes, and the
// slot index keeps the reference's (totalLength, cameraLength) numbering, which is all the gather stage needs.
// The l = 0 pairs (c in 2..D-1) never carry a ray: their slot is written as empty by the lane of c.
template <bool GGX, int G>
__global__ __launch_bounds__(kWave) void gen_connect_kernel(SceneDev S, FrameDev F, PathBuf P) {
  BDPT_ONE_WAVE_PER_GROUP();
  constexpr int kPix = kWave / G;
  __shared__ float4 s_light[kPix][G][4];       // q0..q3 of light vertex g (zeros past the end of the sub-path)
  __shared__ float4 s_eyePos[kPix][G];         // position of eye vertex g + 1 (zero past the end)
  __shared__ uint8_t s_pair[G * (G - 1) / 2];  // pair ordinal -> (cameraLength << 4) | lightLength, for this launch's depth
  bool act = false;
  uint32_t i = 0;
  int g = 0;
  if (!queueGroup<G>(P.qcount, P.pathSubCap, act, i, g)) return;
  const int e = (int)(threadIdx.x / G);
  const uint32_t p = act ? P.queue[0][i] : 0u;
  const int D = (int)F.p.maxDepth;  // <= G: the context is sized for it (launcher)
  const int eyeLast = act ? (int)P.eyeLast[p] : 0;
  const int lightLast = act ? (int)P.lightLast[p] : -1;
  const f3 camPos = ld3(F.cam.posW);
  const int cameraLength = g + 1;
  const bool camAct = act && cameraLength <= D - 1;
  const int nEval = (D * (D - 1)) / 2;
  for (int q = (int)threadIdx.x; q < nEval; q += kWave) {  // ordinal q: camera lengths in order, light lengths 1..D-c within
    int c = 1, r = q;
    while (r >= D - c) {
      r -= D - c;
      c++;
    }
    s_pair[q] = (uint8_t)((c << 4) | (r + 1));
  }
  {
    float4 l0 = make_float4(0, 0, 0, 0), l1 = l0, l2 = l0, l3 = l0;
    if (act && g <= lightLast && g < D) {
      const float4* r = vtxPtr(P, PATH_LIGHT, g, p);
      l0 = r[0];
      l1 = r[1];
      l2 = r[2];
      if (GGX) l3 = r[3];
    }
    s_light[e][g][0] = l0;
    s_light[e][g][1] = l1;
    s_light[e][g][2] = l2;
    s_light[e][g][3] = l3;
  }
  Vtx ce = zeroVtx();
  f3 aE = mk(0), aL = mk(0);
  if (camAct) {
    loadConnVtx<GGX>(P, PATH_EYE, cameraLength, eyeLast, p, ce);
    if (cameraLength - 1 == 0)
      aE = mk(1.0f);
    else if (cameraLength - 1 <= eyeLast)
      aE = ldPlane3(P, PATH_EYE, cameraLength - 1, F_COL, p);
    aL = (cameraLength - 1 <= lightLast) ? ldPlane3(P, PATH_LIGHT, cameraLength - 1, F_COL, p) : mk(0);  // sic, BDPTUtils.hlsli:198
  }
  s_eyePos[e][g] = make_float4(ce.pos.x, ce.pos.y, ce.pos.z, 0.0f);
  __syncthreads();
  f3 cprevPos = camPos;
  if (g > 0) {
    const float4 q = s_eyePos[e][g - 1];  // eye vertex cameraLength - 1, zero when it does not exist
    cprevPos = mk(q.x, q.y, q.z);
  }
  const f3 woE = normalize(cprevPos - ce.pos);
  if (camAct && cameraLength >= 2)  // (c, l = 0): total length c, never evaluated (BDPTMain.rt.hlsl:217 skips lightLength 0 below)
    P.slotRay[(size_t)(2 * D + ((cameraLength - 1) * cameraLength) / 2 - 1 + (cameraLength - 1)) * P.Np + p] = kNoRay;
  const int groupBase = (int)threadIdx.x - g;
  const int rounds = (nEval + G - 1) / G;
  uint32_t nConn = 0;
  for (int it = 0; it < rounds; it++) {
    const int ord = it * G + g;
    const bool pairAct = act && ord < nEval;
    const int cl = (ord < nEval) ? (int)s_pair[ord] : ((1 << 4) | 1);
    const int c = cl >> 4, lightLength = cl & 15, totalLength = c + lightLength;
    const int slot = 2 * D + ((totalLength - 1) * totalLength) / 2 - 1 + (c - 1);
    // the eye side of the pair, from the lane that loaded camera length c
    const int src = groupBase + c - 1;
    Vtx ev;
    ev.pos = mk(__shfl(ce.pos.x, src), __shfl(ce.pos.y, src), __shfl(ce.pos.z, src));
    ev.N = mk(__shfl(ce.N.x, src), __shfl(ce.N.y, src), __shfl(ce.N.z, src));
    ev.dif = mk(__shfl(ce.dif.x, src), __shfl(ce.dif.y, src), __shfl(ce.dif.z, src));
    if (GGX) {
      ev.spec = mk(__shfl(ce.spec.x, src), __shfl(ce.spec.y, src), __shfl(ce.spec.z, src));
      ev.rough = __shfl(ce.rough, src);
      ev.isSpec = __shfl(ce.isSpec ? 1 : 0, src) != 0;
    } else {
      ev.spec = mk(0);
      ev.rough = 0.0f;
      ev.isSpec = false;
    }
    const f3 aEs = mk(__shfl(aE.x, src), __shfl(aE.y, src), __shfl(aE.z, src));
    const f3 aLs = mk(__shfl(aL.x, src), __shfl(aL.y, src), __shfl(aL.z, src));
    const f3 woEs = mk(__shfl(woE.x, src), __shfl(woE.y, src), __shfl(woE.z, src));
    bool emit = false;
    f3 dirAB = mk(0), shade = mk(0);
    float lengthAB = 0.0f;
    if (pairAct) {
      const float4 q0 = s_light[e][lightLength][0], q1 = s_light[e][lightLength][1], q2 = s_light[e][lightLength][2];
      Vtx le;
      le.pos = mk(q0.x, q0.y, q0.z);
      le.N = mk(q1.x, q1.y, q1.z);
      le.dif = mk(q2.x, q2.y, q2.z);
      if (GGX) {
        const float4 q3 = s_light[e][lightLength][3];
        le.spec = mk(q3.x, q3.y, q3.z);
        le.rough = q0.w;
        le.isSpec = q1.w != 0.0f;
      } else {
        le.spec = mk(0);
        le.rough = 0.0f;
        le.isSpec = false;
      }
      const float4 qp = s_light[e][lightLength - 1][0];  // light vertex lightLength - 1 (zero past the end)
      const f3 lprev = mk(qp.x, qp.y, qp.z);
      const f3 vecAB = le.pos - ev.pos;
      const float invLengthAB = 1.0f / length(vecAB);
      const f3 dirG = vecAB * invLengthAB;
      const float cosA = fabsf(dot(ev.N, dirG));
      const float cosB = fabsf(dot(le.N, dirG));
      const float Gt = cosA * cosB * invLengthAB * invLengthAB;
      const f3 connectDir = normalize(ev.pos - le.pos);
      const f3 woL = normalize(lprev - le.pos);
      f3 cst;
      const f3 fsL = evalBRDF<GGX>(connectDir, woL, le.N, le.N, le.dif, le.spec, le.rough, le.isSpec);
      if (allZero(fsL)) {
        cst = fsL;
      } else {
        const f3 fsE = evalBRDF<GGX>(-connectDir, woEs, ev.N, ev.N, ev.dif, ev.spec, ev.rough, ev.isSpec);
        if (allZero(fsE)) {
          cst = fsE;
        } else {
          const f3 k = (fsL * Gt) * fsE;
          cst = (aLs * k) * aEs;
        }
      }
      shade = clampVec(applyStrategyWeight(F, P, p, cst, totalLength, c, lightLength), F.p.clampUpper);
      if (isnan3(shade)) shade = mk(0);
      if (!allZero(shade)) {
        emit = true;
        lengthAB = length(le.pos - ev.pos);
        dirAB = (le.pos - ev.pos) / lengthAB;
      }
    }
    const uint32_t id = emitRay(P, RAY_PAIRS, emit, ev.pos, dirAB, lengthAB, shade);
    if (pairAct) P.slotRay[(size_t)slot * P.Np + p] = id;
    nConn += emit ? 1u : 0u;
  }
  waveAddCount(F.counters, C_RAYS_CONNECT, nConn);
}

// ------------------------------------------------------------------------------------------------
// gather: per valid pixel, add the terms in the reference's order using the visibility the trace
// kernel wrote; NEE terms add without saturate (:166), connection terms saturate per write (:230).
// If no non-zero connection turned out visible the pixel is handed to the lazy rounds: the
// zero-valued pairs still matter, because the first visible one would have saturated the pixel
// (rgb clamp + alpha = 1).  Visible splats are added to the fixed-point splat buffer (quirk 6).
// ------------------------------------------------------------------------------------------------
// The slots of a pixel are read eight at a time: first the eight ray ids, then their visibility bytes, then the
// contributions of the visible ones — three rounds of independent loads instead of a (id -> visibility -> contribution)
// chain per slot; the sums below still run in the reference's order.
constexpr int kGatherChunk = 8;
struct SlotChunk {
  uint32_t id[kGatherChunk];
  bool vis[kGatherChunk];
  float cx[kGatherChunk], cy[kGatherChunk], cz[kGatherChunk];
};
BD void loadSlotChunk(const PathBuf& P, uint32_t p, int slot0, int n, SlotChunk& c) {
  const size_t cap = P.rayCap;
#pragma unroll
  for (int j = 0; j < kGatherChunk; j++) c.id[j] = (j < n) ? P.slotRay[(size_t)(slot0 + j) * P.Np + p] : kNoRay;
#pragma unroll
  for (int j = 0; j < kGatherChunk; j++) c.vis[j] = (c.id[j] != kNoRay) && (P.rayVis[c.id[j]] != 0);
#pragma unroll
  for (int j = 0; j < kGatherChunk; j++) {
    c.cx[j] = c.cy[j] = c.cz[j] = 0.0f;
    if (c.vis[j]) {
      c.cx[j] = P.rayContrib[c.id[j]];
      c.cy[j] = P.rayContrib[cap + c.id[j]];
      c.cz[j] = P.rayContrib[2 * cap + c.id[j]];
    }
  }
}

BD bool gatherLane(const FrameDev& F, const PathBuf& P, uint32_t p, uint32_t& nSplat) {
  bool pending = false;
  const size_t pix = P.pix[p];
  float4* out4 = reinterpret_cast<float4*>(F.out);
  float4 acc = out4[pix];
  const int D = (int)F.p.maxDepth;
  SlotChunk c;
  if (!(F.p.flags & BDPT_PARAM_NO_NEE)) {
    for (int t0 = 0; t0 < D; t0 += kGatherChunk) {
      const int n = (D - t0 < kGatherChunk) ? D - t0 : kGatherChunk;
      loadSlotChunk(P, p, t0, n, c);
#pragma unroll
      for (int j = 0; j < kGatherChunk; j++)
        if (j < n) {  // an occluded or absent term adds 0 (the sum's rounding is the reference's either way)
          acc.x = acc.x + c.cx[j];
          acc.y = acc.y + c.cy[j];
          acc.z = acc.z + c.cz[j];
          acc.w = acc.w + 1.0f;
        }
    }
  }
  if (!(F.p.flags & BDPT_PARAM_NO_CONNECT)) {
    const int nPairs = (int)numConnectPairs((uint32_t)D);
    bool sat = false;
    for (int s0 = 0; s0 < nPairs; s0 += kGatherChunk) {
      const int n = (nPairs - s0 < kGatherChunk) ? nPairs - s0 : kGatherChunk;
      loadSlotChunk(P, p, 2 * D + s0, n, c);
#pragma unroll
      for (int j = 0; j < kGatherChunk; j++)
        if (c.vis[j]) {
          acc.x = saturate(acc.x + c.cx[j]);
          acc.y = saturate(acc.y + c.cy[j]);
          acc.z = saturate(acc.z + c.cz[j]);
          acc.w = saturate(acc.w + 1.0f);
          sat = true;
        }
    }
    pending = (!sat && nPairs > 0);  // settled by the lazy rounds below
  }
  out4[pix] = acc;
  if (!(F.p.flags & BDPT_PARAM_NO_SPLAT)) {
    for (int t0 = 0; t0 < D; t0 += kGatherChunk) {
      const int n = (D - t0 < kGatherChunk) ? D - t0 : kGatherChunk;
      loadSlotChunk(P, p, D + t0, n, c);
      uint32_t target[kGatherChunk];
#pragma unroll
      for (int j = 0; j < kGatherChunk; j++) target[j] = c.vis[j] ? P.splatPix[(size_t)(t0 + j) * P.Np + p] : kNoRay;
#pragma unroll
      for (int j = 0; j < kGatherChunk; j++) {
        if (!c.vis[j] || target[j] == kNoRay) continue;  // kNoRay: outside the frame (quirk 8)
        unsigned long long* sp = F.splat + (size_t)target[j] * 4;
        const unsigned long long qx = toFixed(c.cx[j]), qy = toFixed(c.cy[j]), qz = toFixed(c.cz[j]);
        if (qx) atomicAdd(&sp[0], qx);
        if (qy) atomicAdd(&sp[1], qy);
        if (qz) atomicAdd(&sp[2], qz);
        atomicAdd(&sp[3], 1ull);
        nSplat++;
      }
    }
  }
  return pending;
}

__global__ __launch_bounds__(kWave) void gather_kernel(FrameDev F, PathBuf P, uint32_t* __restrict__ lazyList,
                                                       uint32_t* __restrict__ lazyCount) {
  BDPT_ONE_WAVE_PER_GROUP();
  bool act = false;
  uint32_t i = 0;
  if (!queueChunk(P.qcount, P.pathSubCap, act, i)) return;
  uint32_t nSplat = 0, p = 0;
  bool pending = false;
  if (act) {
    p = P.queue[0][i];
    pending = gatherLane(F, P, p, nSplat);
    if (pending) P.lazyCursor[p] = 0;
  }
  waveAddCount(F.counters, C_SPLATS, nSplat);
  wavePush(pending, p, lazyList, lazyCount, P.pathSubCap);
}

// ------------------------------------------------------------------------------------------------
// Lazy rounds.  A pending pixel has no visible non-zero connection; the reference would still have
// traced its zero-valued pairs (BDPTMain.rt.hlsl:223) and the first visible one saturates the pixel
// (:230).  Only the OR of their visibilities matters, so each round queues the next `batch`
// zero-valued pairs of every still-pending pixel, traces them in the persistent kernel, and
// settles the pixels that found a visible one.
// ------------------------------------------------------------------------------------------------
BD void pairFromOrdinal(int D, int ord, int& totalLength, int& cameraLength) {
  int t = 2;
  for (; t <= D; t++) {
    const int cnt = (t < D - 1) ? t : (D - 1);
    if (ord < cnt) break;
    ord -= cnt;
  }
  totalLength = t;
  cameraLength = ord + 1;
}

// G lanes per pending pixel (8, or 16 for contexts sized beyond depth 8, as the generators): lane g examines pair ordinal
// base + g of the current chunk, the group's candidates are counted with one __ballot, and the first `batch` of them in
// ordinal order get a ray — the very rays, in the very slots of lazyRay, one lane per pixel found by walking the
// ordinals one by one (round 3: up to 90 serial steps per lane at depth 16, lane utilisation 0.14,
// profiles/r4/config5/pmc_summary.txt).
template <int G>
__global__ __launch_bounds__(kWave) void lazy_gen_kernel(FrameDev F, PathBuf P, const uint32_t* __restrict__ list,
                                                         const uint32_t* __restrict__ listCount, int batch) {
  BDPT_ONE_WAVE_PER_GROUP();
  bool act = false;
  uint32_t i = 0;
  int g = 0;
  if (!queueGroup<G>(listCount, P.pathSubCap, act, i, g)) return;
  const uint32_t p = act ? list[i] : 0u;
  const int D = (int)F.p.maxDepth;
  const int nPairs = (int)numConnectPairs((uint32_t)D);
  const int eyeLast = act ? (int)P.eyeLast[p] : 0, lightLast = act ? (int)P.lightLast[p] : -1;
  const bool lightGhost = act && (int)P.lightReal[p] < lightLast;  // last light vertex is a copy of its predecessor
  const int lane = (int)(threadIdx.x & 63u), groupShift = lane - g;
  const unsigned long long groupMask = ((1ull << G) - 1ull) << groupShift, below = (1ull << lane) - 1ull;
  int base = act ? (int)P.lazyCursor[p] : nPairs;  // first ordinal of the chunk the group looks at (group-uniform)
  int cursor = base;                               // what the pixel's cursor becomes
  int taken = 0;                                   // rays this round has given the pixel so far (group-uniform)
  uint32_t nRays = 0;
  for (;;) {
    const bool live = act && taken < batch && base < nPairs;
    if (__ballot(live) == 0ull) break;  // (wave-uniform: emitRay below is a wave collective)
    const int ord = base + g;
    bool cand = false;
    int cameraLength = 0, lightLength = 0;
    if (live && ord < nPairs && P.slotRay[(size_t)(2 * D + ord) * P.Np + p] == kNoRay) {  // (a slot with a ray: it was occluded)
      int totalLength = 0;
      pairFromOrdinal(D, ord, totalLength, cameraLength);
      lightLength = totalLength - cameraLength;
      // Vertices past the end of a sub-path are all the zero vertex and a ghost repeats its predecessor,
      // so many pairs are the SAME ray (bitwise equal end points).  Only the first pair of each such
      // class in the reference's order needs tracing: it comes earlier and, the pixel still being
      // pending, was occluded.
      const int cmin = (cameraLength <= eyeLast) ? cameraLength : eyeLast + 1;
      int lmin = (lightLength > lightLast) ? lightLast + 1 : ((lightGhost && lightLength == lightLast) ? lightLast - 1 : lightLength);
      if (cmin == 1 && lmin == 0) lmin = 1;  // (cameraLength 1, lightLength 0) has total length 1: not a pair
      cand = (cameraLength == cmin && lightLength == lmin);
    }
    const unsigned long long cm = __ballot(cand) & groupMask;
    const int rank = __popcll(cm & below), n = __popcll(cm);
    const bool emit = cand && taken + rank < batch;
    const unsigned long long lastM = __ballot(cand && taken + rank == batch - 1) & groupMask;  // the candidate that fills the round
    f3 posA = mk(0), dirAB = mk(0);
    float lengthAB = 0.0f;
    if (emit) {
      posA = (cameraLength <= eyeLast) ? ldPlane3(P, PATH_EYE, cameraLength, F_POS, p) : mk(0);
      const f3 posB = (lightLength <= lightLast) ? ldPlane3(P, PATH_LIGHT, lightLength, F_POS, p) : mk(0);
      lengthAB = length(posB - posA);
      dirAB = (posB - posA) / lengthAB;
    }
    const uint32_t id = emitRay(P, RAY_PAIRS, emit, posA, dirAB, lengthAB, mk(0));
    if (emit) P.lazyRay[(size_t)(taken + rank) * P.Np + p] = id;
    nRays += emit ? 1u : 0u;
    if (live) {
      if (lastM != 0ull) {  // the round is full: the cursor stops behind the candidate that filled it
        cursor = base + (__ffsll((long long)lastM) - 1 - groupShift) + 1;
        taken = batch;
      } else {
        taken += n;
        base += G;
        cursor = base < nPairs ? base : nPairs;
      }
    }
  }
  if (act) {
    for (int b = taken + g; b < batch; b += G) P.lazyRay[(size_t)b * P.Np + p] = kNoRay;  // slots the round did not fill
    if (g == 0) P.lazyCursor[p] = (uint8_t)cursor;
  }
  waveAddCount(F.counters, C_RAYS_CONNECT, nRays);
  waveAddCount(F.counters, C_RAYS_LAZY, nRays);
}

__global__ __launch_bounds__(kWave) void lazy_check_kernel(FrameDev F, PathBuf P, const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ listCount, int batch,
                                                           uint32_t* __restrict__ nextList, uint32_t* __restrict__ nextCount) {
  BDPT_ONE_WAVE_PER_GROUP();
  bool act = false;
  uint32_t i = 0;
  if (!queueChunk(listCount, P.pathSubCap, act, i)) return;
  bool again = false;
  uint32_t p = 0;
  if (act) {
    p = list[i];
    bool vis = false;
    for (int b = 0; b < batch; b++) {
      const uint32_t id = P.lazyRay[(size_t)b * P.Np + p];
      if (id != kNoRay && P.rayVis[id]) vis = true;
    }
    if (vis) {
      const size_t pix = P.pix[p];
      float4* out4 = reinterpret_cast<float4*>(F.out);
      float4 acc = out4[pix];
      acc.x = saturate(acc.x + 0.0f);
      acc.y = saturate(acc.y + 0.0f);
      acc.z = saturate(acc.z + 0.0f);
      acc.w = saturate(acc.w + 1.0f);
      out4[pix] = acc;
    } else {
      const int nPairs = (int)numConnectPairs(F.p.maxDepth);
      again = (int)P.lazyCursor[p] < nPairs;
    }
  }
  wavePush(again, p, nextList, nextCount, P.pathSubCap);
}

// out = saturate(out + splat) where at least one splat landed
__global__ void resolve_kernel(const unsigned long long* __restrict__ splat, bool tileLocal, uint32_t splatRow0, SplatLayout L,
                               float4* __restrict__ out, uint32_t W, const uint32_t* __restrict__ pixOf, uint32_t Np) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < Np; p += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = pixOf[p];
    size_t sidx = p;
    if (!tileLocal) {
      const uint32_t y = (uint32_t)(pix / W), x = (uint32_t)(pix - (size_t)y * W);
      sidx = splatIndex(L, W, x, y) - (size_t)splatRow0 * W;
    }
    const ulonglong2 a = reinterpret_cast<const ulonglong2*>(splat + sidx * 4)[0];
    const ulonglong2 b = reinterpret_cast<const ulonglong2*>(splat + sidx * 4)[1];
    if (b.y == 0ull) continue;
    float4 o = out[pix];
    o.x = saturate(o.x + (float)a.x * 2.3283064365386963e-10f);
    o.y = saturate(o.y + (float)a.y * 2.3283064365386963e-10f);
    o.z = saturate(o.z + (float)b.x * 2.3283064365386963e-10f);
    o.w = saturate(o.w + (float)b.y);
    out[pix] = o;
  }
}

// CP accumulate.ps.hlsl:28-42 followed by the two blits of SimpleAccumulationPass.cpp:127-133
__global__ void accumulate_kernel(float4* __restrict__ last, float4* __restrict__ cur, uint32_t accumCount, uint32_t maxAccum,
                                  uint64_t numTexels) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numTexels; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 c = cur[i], pv = last[i];
    float4 r;
    if (accumCount < maxAccum) {
      const float a = (float)accumCount, b = (float)(accumCount + 1);
      r.x = (a * pv.x + c.x) / b;
      r.y = (a * pv.y + c.y) / b;
      r.z = (a * pv.z + c.z) / b;
      r.w = (a * pv.w + c.w) / b;
    } else {
      r = pv;
    }
    cur[i] = r;
    last[i] = r;
  }
}

// the same running mean over a tile's pixels only (full-frame buffers, tile-local pixel list)
__global__ void accumulate_tile_kernel(float4* __restrict__ last, float4* __restrict__ cur, uint32_t accumCount, uint32_t maxAccum,
                                       const uint32_t* __restrict__ pixOf, uint32_t Np) {
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < Np; p += (size_t)gridDim.x * blockDim.x) {
    const size_t i = pixOf[p];
    const float4 c = cur[i], pv = last[i];
    float4 r;
    if (accumCount < maxAccum) {
      const float a = (float)accumCount, b = (float)(accumCount + 1);
      r.x = (a * pv.x + c.x) / b;
      r.y = (a * pv.y + c.y) / b;
      r.z = (a * pv.z + c.z) / b;
      r.w = (a * pv.w + c.w) / b;
    } else {
      r = pv;
    }
    cur[i] = r;
    last[i] = r;
  }
}

// ---- test hooks ---------------------------------------------------------------------------------
__global__ void test_rng_kernel(const uint32_t* v0, const uint32_t* v1, uint32_t n, uint32_t draws, uint32_t* states, float* floats) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s = initRand(v0[i], v1[i]);
  for (uint32_t k = 0; k < draws; k++) {
    float r = nextRand(s);
    states[(size_t)i * draws + k] = s;
    floats[(size_t)i * draws + k] = r;
  }
}
template <int MODE>
__global__ __launch_bounds__(kWave) void test_trace_kernel(SceneDev S, const float* rays, uint32_t n, int32_t* prim, float* tuv) {
  BDPT_ONE_WAVE_PER_GROUP();
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + (size_t)i * 8;
  uint32_t a = 0, b = 0;
  Hit h = traverse<MODE, false>(S, ld3(r), ld3(r + 3), r[6], r[7], s_stack + threadIdx.x, a, b);
  // Hit::rec (what the occluder hints of light-tracing rays are made of) must name the record of the primitive hit: a
  // query whose record word says otherwise reports primitive -2, which no oracle answer equals
  if (MODE != 2 && h.prim >= 0) {
    const bool named = h.rec < S.numRecs && (int)__float_as_uint(reinterpret_cast<const float4*>(S.recs)[(size_t)h.rec * kRecF4].w) == h.prim;
    if (!named) h.prim = -2;
  }
  prim[i] = h.prim;
  const bool rec = (MODE != 2) && h.prim >= 0;
  tuv[(size_t)i * 3] = rec ? h.t : 0.0f;
  tuv[(size_t)i * 3 + 1] = rec ? h.u : 0.0f;
  tuv[(size_t)i * 3 + 2] = rec ? h.v : 0.0f;
}
template <bool GGX>
__global__ void test_bsdf_kernel(const float* in, uint32_t n, bool fromLobe, float* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * 20;
  f3 N = ld3(p), V = ld3(p + 3), Lq = ld3(p + 6), dif = ld3(p + 9), spec = ld3(p + 12);
  float rough = p[15];
  bool isSpec = p[16] != 0.0f;
  uint32_t seed = __float_as_uint(p[17]);
  f3 L;
  float pdf;
  bool sSpec;
  f3 w = sampleBRDF<GGX>(seed, N, N, V, dif, spec, rough, fromLobe, L, pdf, sSpec);
  f3 fr = evalBRDF<GGX>(V, Lq, N, N, dif, spec, rough, isSpec);
  float* o = out + (size_t)i * 16;
  o[0] = w.x;
  o[1] = w.y;
  o[2] = w.z;
  o[3] = L.x;
  o[4] = L.y;
  o[5] = L.z;
  o[6] = pdf;
  o[7] = sSpec ? 1.0f : 0.0f;
  o[8] = fr.x;
  o[9] = fr.y;
  o[10] = fr.z;
  o[11] = o[12] = o[13] = o[14] = o[15] = 0.0f;
}


// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#ifndef PERCU
#define PERCU 24
#endif
static inline uint32_t blocksFor(uint64_t n) { return (uint32_t)((n + kWave - 1) / kWave); }
// grid of a dense kernel over a sharded path queue: every (list, chunk) pair gets a workgroup
static inline uint32_t queueGrid(const PathBuf& P) { return (P.pathSubCap / kWave) * kNumSubQueues; }

// ONE WORKGROUP = ONE WAVE.  Every kernel above this line that is declared __launch_bounds__(kWave) relies on it: LDS arrays
// are sized for one wave and indexed by threadIdx.x (s_stack + threadIdx.x with kWave-strided rows, s_pool, s_light[e][g] with
// e = threadIdx.x / G, s_eyePos, s_pair), ovfSlot() addresses blockIdx.x * kWave + lane, queueChunk / queueGroup deal
// list entries by threadIdx.x, __syncthreads() is used as a wave barrier, and emitRay / wavePush / waveAddCount are wave
// collectives whose atomics assume one leader per workgroup.  With a second wave in the workgroup threadIdx.x runs to 127:
// the rows alias or run past the arrays (round 4's multi-wave gen_connect variant: s_pair overwritten by the other wave
// -> light lengths up to 15 -> slot indices past slotRay's planes -> the wrong terms and the memory fault recorded in
// profiles/README.md).  So these kernels are launched through launchWave() only — there is no block size to get wrong —
// and each starts with oneWavePerGroup(), which makes a launch of any other shape do nothing instead of corrupting.
template <class K, class... Args>
static void launchWave(K kernel, uint32_t grid, hipStream_t st, Args... args) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kWave), 0, st, args...);
}

// Persistent grids: as many one-wave workgroups as can be resident (LDS 8 KiB/wave, VGPRs).
template <class K>
static uint32_t persistentGrid(K kernel, int numCUs) {
  int perCU = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, kernel, kWave, 0) != hipSuccess || perCU <= 0) perCU = 8;
  if (perCU > PERCU) perCU = PERCU;
  if (perCU > kMaxPersistentPerCU) perCU = kMaxPersistentPerCU;  // the stack overflow area is sized for that many
  return (uint32_t)(perCU * numCUs);
}

void launchGBuffer(const SceneDev& S, const GBufferDev& G, hipStream_t st) {
  const uint32_t Np = G.Np;
  if (!Np) return;
  if (G.counters)
    launchWave(gbuffer_kernel<true>, (uint32_t)(blocksFor(Np)), st, S, G);
  else
    launchWave(gbuffer_kernel<false>, (uint32_t)(blocksFor(Np)), st, S, G);
}

// The 64-byte alpha-test record of every non-opaque triangle (device_scene.hpp alphaTestFails: the triangle's three texture
// coordinates, the material's threshold and constant alpha, how its base colour is given, the texture's size and address)
// from what is on the device already — the triangle's shading record and the material tables — instead of 64 B per
// triangle built by the host and copied over (0.32 GB for the 10 M-triangle courtyard).
__global__ void alpha_recs_kernel(SceneDev S, const uint32_t* __restrict__ alphaTris, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = alphaTris[i];
  const float4* sr = S.shade + (size_t)t * kShadeRecF4;  // 3 x (pos, normal, uv) + material id: floats 6, 7 of each vertex = uv
  const float4 a1 = sr[1], b1 = sr[3], c1 = sr[5], m6 = sr[6];
  const uint32_t mid = __float_as_uint(m6.x);
  const bdpt_material& m = S.materials[mid];
  const uint32_t type = BDPT_FLAG_DIFFUSE_TYPE(m.flags);
  uint32_t mode = 0, tw = 0, th = 0;
  unsigned long long px = 0;
  if (type == BDPT_CHANNEL_UNUSED) {
    mode = 0;
  } else if (type == BDPT_CHANNEL_CONST || m.texBaseColor < 0) {
    mode = 1;
  } else {
    mode = 2;
    const TexDev td = S.matTex[(size_t)mid * 4];
    tw = td.w;
    th = td.h;
    px = (unsigned long long)reinterpret_cast<uintptr_t>(td.px);
  }
  float4* r = out + (size_t)i * 4;
  r[0] = make_float4(a1.z, a1.w, b1.z, b1.w);
  r[1] = make_float4(c1.z, c1.w, m.alphaThreshold, m.baseColor[3]);
  r[2] = make_float4(__uint_as_float(mode), __uint_as_float(tw), __uint_as_float(th), 0.0f);
  r[3] = make_float4(__uint_as_float((uint32_t)(px & 0xffffffffull)), __uint_as_float((uint32_t)(px >> 32)), 0.0f, 0.0f);
}
void launchAlphaRecs(const SceneDev& S, const uint32_t* alphaTris, uint32_t n, float4* out, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(alpha_recs_kernel, dim3((n + 255) / 256), dim3(256), 0, st, S, alphaTris, n, out);
}

void launchHintFill(const SceneDev& S, const GBufferDev& G, hipStream_t st) {  // G.Np = W * H, G.pix unused
  if (!G.Np || !G.hintPix) return;
  launchWave(gbuffer_kernel<false, true>, (uint32_t)(blocksFor(G.Np)), st, S, G);
}

void launchLightMaps(const SceneDev& S, uint32_t* maps, uint32_t res, hipStream_t st) {
  const uint64_t n = (uint64_t)6 * res * res * S.numLights;
  if (!n) return;
  launchWave(light_map_kernel, (uint32_t)((n + kWave - 1) / kWave), st, S, maps, res);
}

void launchInitPaths(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  if (F.p.matIndex == 0)
    launchWave(init_paths_kernel<true>, (uint32_t)(blocksFor(P.Np)), st, S, F, P);
  else
    launchWave(init_paths_kernel<false>, (uint32_t)(blocksFor(P.Np)), st, S, F, P);
}

void launchWalk(const SceneDev& S, const FrameDev& F, const PathBuf& P, LaunchGrids& G, int numCUs, hipStream_t st) {
  if (!P.Np || F.p.maxDepth < 1) return;
  const bool cnt = (F.p.flags & BDPT_PARAM_COUNTERS) != 0, ggx = F.p.matIndex == 0;
  const bool ext = (F.p.flags & (BDPT_PARAM_ENV_ON_MISS | BDPT_PARAM_EMISSIVE_HITS)) != 0;
  uint32_t& g = G.walk[(ext ? 4 : 0) + (ggx ? 2 : 0) + (cnt ? 1 : 0)];
  // at most two sub-paths per pixel: a small tile does not need the whole persistent grid
  const uint32_t need = blocksFor((uint64_t)2 * P.Np);
#define BDPT_LAUNCH_WALK(GGX, CNT, EXT)                                                                        \
  {                                                                                                           \
    if (!g) g = persistentGrid(walk_kernel<GGX, CNT, EXT>, numCUs);                                            \
    launchWave((walk_kernel<GGX, CNT, EXT>), (uint32_t)(std::min(g, need)), st, S, F, P, P.qhead); \
  }
  if (ext) {
    if (ggx && cnt) BDPT_LAUNCH_WALK(true, true, true)
    else if (ggx) BDPT_LAUNCH_WALK(true, false, true)
    else if (cnt) BDPT_LAUNCH_WALK(false, true, true)
    else BDPT_LAUNCH_WALK(false, false, true)
  } else {
    if (ggx && cnt) BDPT_LAUNCH_WALK(true, true, false)
    else if (ggx) BDPT_LAUNCH_WALK(true, false, false)
    else if (cnt) BDPT_LAUNCH_WALK(false, true, false)
    else BDPT_LAUNCH_WALK(false, false, false)
  }
#undef BDPT_LAUNCH_WALK
}

void launchMisPrefix(const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  launchWave(mis_prefix_kernel, (uint32_t)(queueGrid(P)), st, F, P);
}
// The three generators only share the ray queues (atomic appends), so the host may launch them on different streams.
// G lanes per pixel: 8 for contexts sized up to depth 8, else 16 (the ray queues are sized for that shape).
#define BDPT_LAUNCH_GEN(KERNEL)                                                                   \
  {                                                                                               \
    const bool ggx = F.p.matIndex == 0, wide = P.D1 > 9; /* the depth the context is sized for */  \
    if (ggx && !wide) launchWave(KERNEL<true, 8>, queueGrid(P) * 8, st, S, F, P);     \
    else if (ggx) launchWave(KERNEL<true, 16>, queueGrid(P) * 16, st, S, F, P);       \
    else if (!wide) launchWave(KERNEL<false, 8>, queueGrid(P) * 8, st, S, F, P);      \
    else launchWave(KERNEL<false, 16>, queueGrid(P) * 16, st, S, F, P);               \
  }
void launchGenNee(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  BDPT_LAUNCH_GEN(gen_nee_kernel)
}
void launchGenSplat(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np || (F.p.flags & BDPT_PARAM_NO_SPLAT)) return;
  BDPT_LAUNCH_GEN(gen_splat_kernel)
}
void launchGenConnect(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np || (F.p.flags & BDPT_PARAM_NO_CONNECT) || F.p.maxDepth < 2) return;
  BDPT_LAUNCH_GEN(gen_connect_kernel)
}
#undef BDPT_LAUNCH_GEN

void launchTraceShadow(const SceneDev& S, const FrameDev& F, const PathBuf& P, int cls, LaunchGrids& G, int numCUs, hipStream_t st) {
  if (!P.Np) return;
  const bool cnt = (F.p.flags & BDPT_PARAM_COUNTERS) != 0;
  // the class's rays are ids [rayBase, rayBase + kNumRaySubQueues * raySubCap): the kernel sees planes and visibility bytes from rayBase on
  RayQueue Q{P.rayQ + P.rayBase[cls], P.rayCap, P.raySubCap[cls], kNumRaySubQueues, P.rayCount + (size_t)cls * kRayCursorBlock,
             P.rayHead + (size_t)cls * kRayCursorBlock};
  uint8_t* vis = P.rayVis + P.rayBase[cls];
  uint32_t& g = G.shadow[cnt ? 1 : 0];
  if (cnt) {
    if (!g) g = persistentGrid(trace_shadow_kernel<true>, numCUs);
    launchWave(trace_shadow_kernel<true>, (uint32_t)(g), st, S, Q, vis, F.counters, F.p.minT);
  } else {
    if (!g) g = persistentGrid(trace_shadow_kernel<false>, numCUs);
    launchWave(trace_shadow_kernel<false>, (uint32_t)(g), st, S, Q, vis, F.counters, F.p.minT);
  }
}

void launchGather(const FrameDev& F, const PathBuf& P, uint32_t* lazyList, uint32_t* lazyCount, hipStream_t st) {
  if (!P.Np) return;
  launchWave(gather_kernel, (uint32_t)(queueGrid(P)), st, F, P, lazyList, lazyCount);
}
void launchLazyGen(const FrameDev& F, const PathBuf& P, const uint32_t* list, const uint32_t* listCount, int batch, hipStream_t st) {
  if (!P.Np) return;
  if (P.D1 > 9)  // the depth the context is sized for: 16 lanes per pixel, as the generators
    launchWave(lazy_gen_kernel<16>, (uint32_t)(queueGrid(P) * 16), st, F, P, list, listCount, batch);
  else
    launchWave(lazy_gen_kernel<8>, (uint32_t)(queueGrid(P) * 8), st, F, P, list, listCount, batch);
}
void launchLazyCheck(const FrameDev& F, const PathBuf& P, const uint32_t* list, const uint32_t* listCount, int batch,
                     uint32_t* nextList, uint32_t* nextCount, hipStream_t st) {
  if (!P.Np) return;
  launchWave(lazy_check_kernel, (uint32_t)(queueGrid(P)), st, F, P, list, listCount, batch, nextList, nextCount);
}

void launchResolve(const unsigned long long* splat, bool tileLocal, uint32_t splatRow0, const SplatLayout& L, float* out, uint32_t W,
                   const uint32_t* pix, uint32_t Np, hipStream_t st) {
  if (!Np) return;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)Np + 255) / 256, 2048);
  hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, st, splat, tileLocal, splatRow0, L, reinterpret_cast<float4*>(out), W, pix,
                     Np);
}
void launchAccumulate(float* last, float* cur, uint32_t accumCount, uint32_t maxAccum, uint64_t numTexels, hipStream_t st) {
  if (!numTexels) return;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((numTexels + 255) / 256, 2048);
  hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<float4*>(last),
                     reinterpret_cast<float4*>(cur), accumCount, maxAccum, numTexels);
}
void launchAccumulateTile(float* last, float* cur, uint32_t accumCount, uint32_t maxAccum, const uint32_t* pix, uint32_t Np,
                          hipStream_t st) {
  if (!Np) return;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)Np + 255) / 256, 2048);
  hipLaunchKernelGGL(accumulate_tile_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<float4*>(last), reinterpret_cast<float4*>(cur),
                     accumCount, maxAccum, pix, Np);
}
template <class T>
__global__ void tile_pack_kernel(const T* __restrict__ frame, T* __restrict__ packed, const uint32_t* __restrict__ pix, uint32_t Np) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Np) packed[i] = frame[pix[i]];
}
// packed row r of owner o: stripe k = r / R of that owner = frame stripe k * owners + o, row r % R within it
template <class T>
__global__ void tile_unpack_kernel(const T* __restrict__ packed, T* __restrict__ frame, uint32_t W, uint32_t H, uint32_t R, uint32_t owners,
                                   uint32_t owner, uint32_t packedRows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)packedRows * W) return;
  const uint32_t r = (uint32_t)(i / W), x = (uint32_t)(i - (size_t)r * W);
  const uint32_t y = ((r / R) * owners + owner) * R + r % R;
  if (y < H) frame[(size_t)y * W + x] = packed[i];
}
void launchTilePack(const void* frame, void* packed, uint32_t bpp, const uint32_t* pix, uint32_t Np, hipStream_t st) {
  if (!Np) return;
  const dim3 g((Np + 255) / 256), b(256);
  if (bpp == 16)
    hipLaunchKernelGGL(tile_pack_kernel<uint4>, g, b, 0, st, (const uint4*)frame, (uint4*)packed, pix, Np);
  else if (bpp == 8)
    hipLaunchKernelGGL(tile_pack_kernel<uint2>, g, b, 0, st, (const uint2*)frame, (uint2*)packed, pix, Np);
  else
    hipLaunchKernelGGL(tile_pack_kernel<uint32_t>, g, b, 0, st, (const uint32_t*)frame, (uint32_t*)packed, pix, Np);
}
void launchTileUnpack(const void* packed, void* frame, uint32_t bpp, uint32_t W, uint32_t H, uint32_t R, uint32_t owners, uint32_t owner,
                      uint32_t packedRows, hipStream_t st) {
  const size_t n = (size_t)packedRows * W;
  if (!n) return;
  const dim3 g((uint32_t)((n + 255) / 256)), b(256);
  if (bpp == 16)
    hipLaunchKernelGGL(tile_unpack_kernel<uint4>, g, b, 0, st, (const uint4*)packed, (uint4*)frame, W, H, R, owners, owner, packedRows);
  else if (bpp == 8)
    hipLaunchKernelGGL(tile_unpack_kernel<uint2>, g, b, 0, st, (const uint2*)packed, (uint2*)frame, W, H, R, owners, owner, packedRows);
  else
    hipLaunchKernelGGL(tile_unpack_kernel<uint32_t>, g, b, 0, st, (const uint32_t*)packed, (uint32_t*)frame, W, H, R, owners, owner, packedRows);
}

void launchTestRng(const uint32_t* v0, const uint32_t* v1, uint32_t n, uint32_t draws, uint32_t* states, float* floats,
                   hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(test_rng_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v0, v1, n, draws, states, floats);
}
void launchTestTrace(const SceneDev& S, const float* rays, uint32_t n, int mode, int32_t* prim, float* tuv, hipStream_t st) {
  if (!n) return;
  if (mode == 0)
    launchWave(test_trace_kernel<0>, (uint32_t)(blocksFor(n)), st, S, rays, n, prim, tuv);
  else if (mode == 1)
    launchWave(test_trace_kernel<1>, (uint32_t)(blocksFor(n)), st, S, rays, n, prim, tuv);
  else
    launchWave(test_trace_kernel<2>, (uint32_t)(blocksFor(n)), st, S, rays, n, prim, tuv);
}
// The persistent any-hit kernel over a caller's ray list (planes ox oy oz dx dy dz tmax of stride `cap`, one sub-queue):
// visibility bytes and, through `counters`, the visit tallies and the deepest stack.
void launchTestTraceShadow(const SceneDev& S, const float* planes, uint32_t cap, const uint32_t* count, uint32_t* head, uint8_t* vis,
                           DevCounters* counters, float tmin, int numCUs, hipStream_t st) {
  RayQueue Q{planes, cap, cap, 1u, count, head};
  const uint32_t g = persistentGrid(trace_shadow_kernel<true>, numCUs);
  launchWave(trace_shadow_kernel<true>, (uint32_t)(g), st, S, Q, vis, counters, tmin);
}
void launchTestBsdf(const float* in, uint32_t n, uint32_t matIndex, float* out, hipStream_t st) {
  if (!n) return;
  const bool fromLobe = (matIndex & 2u) != 0;
  if ((matIndex & 1u) == 0)
    hipLaunchKernelGGL(test_bsdf_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, st, in, n, fromLobe, out);
  else
    hipLaunchKernelGGL(test_bsdf_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, st, in, n, fromLobe, out);
}

}  // namespace bdpt
