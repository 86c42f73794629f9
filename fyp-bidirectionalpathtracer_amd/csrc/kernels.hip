// kernels.hip — the wavefront BDPT pipeline for gfx950 (CDNA4, wave64).
//
// Stage kernels (one launch each per frame, per bounce for the two walks):
//   gbuffer_kernel    primary visibility            CP lightProbeGBuffer.rt.hlsl:63-159
//   init_paths_kernel eye vertex 1 + light vertex 0 BDPTMain.rt.hlsl:51-103, 124-135
//   extend_kernel     one bounce of a sub-path      globalIlluminationRay.hlsli:1-45 (closest hit + shade)
//   nee_kernel        next-event estimation         BDPTMain.rt.hlsl:155-167
//   splat_kernel      light tracing to the camera   BDPTMain.rt.hlsl:171-208
//   connect_kernel    s x t vertex connections      BDPTMain.rt.hlsl:212-233
//   resolve_kernel    fold the splat buffer in      (build definition)
//   accumulate_kernel running mean                  CP accumulate.ps.hlsl:28-42
//
// Layout: path vertices are SoA planes indexed by tile-local pixel, so a wave reads 64
// consecutive floats per field; live paths are compacted between bounces with
// __ballot + popcount prefix + one atomic per wave; each lane's BVH traversal stack lives in
// LDS, interleaved by lane (entry e of lane l at word e*64+l) so pushes and pops never
// bank-conflict.  One workgroup = one wave (64 threads): no __syncthreads anywhere, and a
// finished wave frees its slot immediately.
#include "kernels.h"

#include <algorithm>

#include "device_math.hpp"

namespace bdpt {

#define BD __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// queue compaction: append `value` of every lane with `active` to queue[]; one atomic per wave
// ------------------------------------------------------------------------------------------------
BD void wavePush(bool active, uint32_t value, uint32_t* queue, uint32_t* counter) {
  unsigned long long mask = __ballot(active);
  if (mask == 0ull) return;
  const int lane = (int)(threadIdx.x & 63u);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, leader);
  const uint32_t prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  if (active) queue[base + prefix] = value;
}

// ------------------------------------------------------------------------------------------------
// textures, vertex fetch, shading data
// ------------------------------------------------------------------------------------------------
struct f4 {
  float x, y, z, w;
};
BD f4 lerp4(f4 a, f4 b, float s) {
  return f4{a.x + (b.x - a.x) * s, a.y + (b.y - a.y) * s, a.z + (b.z - a.z) * s, a.w + (b.w - a.w) * s};
}
BD int wrapi(int i, int n) {
  int m = i % n;
  return (m < 0) ? m + n : m;
}
BD f4 texel(const SceneDev& S, const TexDev& t, int ix, int iy) {
  const uchar4 p = *reinterpret_cast<const uchar4*>(t.px + ((size_t)iy * t.w + (size_t)ix) * 4);
  f4 r;
  if (t.srgb) {
    r.x = S.sc->srgbLut[p.x];
    r.y = S.sc->srgbLut[p.y];
    r.z = S.sc->srgbLut[p.z];
  } else {
    r.x = (float)p.x / 255.0f;
    r.y = (float)p.y / 255.0f;
    r.z = (float)p.z / 255.0f;
  }
  r.w = (float)p.w / 255.0f;
  return r;
}
// linear filter, wrap addressing, mip 0 (sampler: SharedUtils/SceneLoaderWrapper.cpp:65-68)
BD f4 sampleBilinear(const SceneDev& S, int texId, float u, float v) {
  const TexDev t = S.textures[texId];
  float x = u * (float)t.w - 0.5f;
  float y = v * (float)t.h - 0.5f;
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = wrapi((int)x0, (int)t.w), iy0 = wrapi((int)y0, (int)t.h);
  int ix1 = wrapi(ix0 + 1, (int)t.w), iy1 = wrapi(iy0 + 1, (int)t.h);
  f4 t00 = texel(S, t, ix0, iy0), t10 = texel(S, t, ix1, iy0);
  f4 t01 = texel(S, t, ix0, iy1), t11 = texel(S, t, ix1, iy1);
  return lerp4(lerp4(t00, t10, fx), lerp4(t01, t11, fx), fy);
}
// Falcor ShadingUtils/Shading.slang:88-94
BD f4 sampleTexture(const SceneDev& S, int texId, float u, float v, f4 factor, uint32_t mode) {
  if (mode == BDPT_CHANNEL_UNUSED) return f4{0, 0, 0, 0};
  if (mode == BDPT_CHANNEL_CONST || texId < 0) return factor;
  return sampleBilinear(S, texId, u, v);
}

struct MatDev {
  f4 baseColor, specular;
  f3 emissive;
  float alphaThreshold, IoR;
  uint32_t flags;
  int texBase, texSpec, texEmis, texNorm;
};
BD MatDev loadMaterial(const SceneDev& S, uint32_t id) {
  const float4* m = reinterpret_cast<const float4*>(S.materials + id);
  float4 a = m[0], b = m[1], c = m[2], d = m[3];
  MatDev r;
  r.baseColor = f4{a.x, a.y, a.z, a.w};
  r.specular = f4{b.x, b.y, b.z, b.w};
  r.emissive = mk(c.x, c.y, c.z);
  r.alphaThreshold = c.w;
  r.IoR = d.x;
  r.flags = __float_as_uint(d.y);
  uint32_t t0 = __float_as_uint(d.z), t1 = __float_as_uint(d.w);
  r.texBase = (int)(int16_t)(t0 & 0xffffu);
  r.texSpec = (int)(int16_t)(t0 >> 16);
  r.texEmis = (int)(int16_t)(t1 & 0xffffu);
  r.texNorm = (int)(int16_t)(t1 >> 16);
  return r;
}

// BDPT/BDPTUtils.hlsli:115-127
BD bool alphaTestFails(const SceneDev& S, uint32_t prim, float bu, float bv) {
  const float4* r = S.shade + (size_t)prim * kShadeRecF4;
  const MatDev m = loadMaterial(S, __float_as_uint(r[6].x));
  float u = 0, v = 0;
  const uint32_t mode = BDPT_FLAG_DIFFUSE_TYPE(m.flags);
  if (mode == BDPT_CHANNEL_TEXTURE && m.texBase >= 0) {
    float b0 = 1.0f - bu - bv;
    float4 r1 = r[1], r3 = r[3], r5 = r[5];
    u += r1.z * b0;
    v += r1.w * b0;
    u += r3.z * bu;
    v += r3.w * bu;
    u += r5.z * bv;
    v += r5.w * bv;
  }
  f4 base = sampleTexture(S, m.texBase, u, v, m.baseColor, mode);
  return base.w < m.alphaThreshold;
}

struct Shading {
  f3 posW, V, N, diffuse, specular, emissive;
  float opacity, linearRoughness, roughness, IoR;
};
// getVertexAttributes (Falcor ShadingUtils/Raytracing.slang:60-106) + simplePrepareShadingData
// (BDPT/BDPTUtils.hlsli:2-52); NMAP adds applyNormalMap for the primary hit
// (Falcor ShadingUtils/Shading.slang:135-157, 189-259).
template <bool NMAP>
BD Shading shadeHit(const SceneDev& S, uint32_t prim, float bu, float bv, f3 camPosW) {
  const float4* r = S.shade + (size_t)prim * kShadeRecF4;
  const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6];
  const float b0 = 1.0f - bu - bv;
  float u = 0, v = 0;
  f3 normalW = mk(0), posW = mk(0);
  u += r1.z * b0;
  v += r1.w * b0;
  normalW = normalW + mk(r0.w, r1.x, r1.y) * b0;
  posW = posW + mk(r0.x, r0.y, r0.z) * b0;
  u += r3.z * bu;
  v += r3.w * bu;
  normalW = normalW + mk(r2.w, r3.x, r3.y) * bu;
  posW = posW + mk(r2.x, r2.y, r2.z) * bu;
  u += r5.z * bv;
  v += r5.w * bv;
  normalW = normalW + mk(r4.w, r5.x, r5.y) * bv;
  posW = posW + mk(r4.x, r4.y, r4.z) * bv;
  normalW = normalize(normalW);

  const MatDev m = loadMaterial(S, __float_as_uint(r6.x));
  Shading sd;
  f4 base = sampleTexture(S, m.texBase, u, v, m.baseColor, BDPT_FLAG_DIFFUSE_TYPE(m.flags));
  sd.opacity = m.baseColor.w;
  sd.posW = posW;
  sd.V = normalize(camPosW - posW);
  sd.N = normalize(normalW);
  f4 spec = sampleTexture(S, m.texSpec, u, v, m.specular, BDPT_FLAG_SPECULAR_TYPE(m.flags));
  f3 baseRgb = mk(base.x, base.y, base.z);
  if (BDPT_FLAG_SHADING_MODEL(m.flags) == BDPT_SHADING_MODEL_METAL_ROUGH) {
    sd.diffuse = lerp3(baseRgb, mk(0), spec.z);
    sd.specular = lerp3(mk(0.04f), baseRgb, spec.z);
    sd.linearRoughness = spec.y;
  } else {
    sd.diffuse = baseRgb;
    sd.specular = mk(spec.x, spec.y, spec.z);
    sd.linearRoughness = 1.0f - spec.w;
  }
  sd.linearRoughness = maxf(0.08f, sd.linearRoughness);
  sd.roughness = sd.linearRoughness * sd.linearRoughness;
  f4 em = sampleTexture(S, m.texEmis, u, v, f4{m.emissive.x, m.emissive.y, m.emissive.z, 1.0f}, BDPT_FLAG_EMISSIVE_TYPE(m.flags));
  sd.emissive = mk(em.x, em.y, em.z);
  sd.IoR = m.IoR;
  const bool doubleSided = BDPT_FLAG_DOUBLE_SIDED(m.flags) != 0;
  if (NMAP) {
    const uint32_t mapType = BDPT_FLAG_NORMAL_MAP_TYPE(m.flags);
    if (mapType != BDPT_NORMAL_MAP_UNUSED && m.texNorm >= 0 && S.hasBitangents) {
      f3 bitW = mk(0);
      const float bw[3] = {b0, bu, bv};
#pragma unroll
      for (int i = 0; i < 3; i++) {
        uint32_t vi = S.indices[(size_t)prim * 3 + i];
        bitW = bitW + ld3(S.bitangents + (size_t)vi * 3) * bw[i];
      }
      bitW = normalize(bitW);
      f3 B = normalize(bitW - sd.N * dot(bitW, sd.N));
      f3 T = normalize(cross(B, sd.N));
      f4 mp = sampleBilinear(S, m.texNorm, u, v);
      f3 mapN;
      if (mapType == BDPT_NORMAL_MAP_RGB) {
        mapN = normalize(mk(mp.x, mp.y, mp.z) * 2.0f - mk(1.0f));
      } else {
        float nx = mp.x * 2.0f - 1.0f, ny = mp.y * 2.0f - 1.0f;
        float nz = saturate(mp.x * mp.x + mp.y * mp.y);
        nz = sqrtf(1.0f - nz);
        mapN = normalize(mk(nx, ny, nz));
      }
      sd.N = T * mapN.x + B * mapN.y + sd.N * mapN.z;
    }
  }
  float NdotV = dot(sd.N, sd.V);
  if (NdotV <= 0.0f && doubleSided) sd.N = -sd.N;
  return sd;
}

// ------------------------------------------------------------------------------------------------
// BVH traversal.  MODE 0 closest hit, 1 closest hit with back-face culling (primary rays,
// RAY_FLAG_CULL_BACK_FACING_TRIANGLES), 2 any hit (ACCEPT_FIRST_HIT_AND_END_SEARCH).
// Hit iff tmin < t < tmax; closest-hit ties resolve to the lowest primitive index so the result
// does not depend on traversal order (and equals a brute-force scan).
// ------------------------------------------------------------------------------------------------
struct Hit {
  int prim;
  float t, u, v;
};

template <int MODE, bool COUNT>
BD Hit traverse(const SceneDev& S, f3 o, f3 d, float tmin, float tmax, int* stk, uint32_t& nNodes, uint32_t& nTris) {
  Hit best;
  best.prim = -1;
  best.t = tmax;
  best.u = 0.0f;
  best.v = 0.0f;
  const f3 idir = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  int sp = 0;
  int cur = 0;
  for (;;) {
    if (cur >= 0) {
      const float4* np = S.nodes + (size_t)cur * 4;
      const float4 q0 = np[0], q1 = np[1], q2 = np[2];
      const int4 q3 = reinterpret_cast<const int4*>(np)[3];
      if (COUNT) nNodes++;
      float ax0 = (q0.x - o.x) * idir.x, ax1 = (q0.w - o.x) * idir.x;
      float ay0 = (q0.y - o.y) * idir.y, ay1 = (q1.x - o.y) * idir.y;
      float az0 = (q0.z - o.z) * idir.z, az1 = (q1.y - o.z) * idir.z;
      float tn0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), tmin));
      float tf0 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fminf(fmaxf(az0, az1), best.t));
      float bx0 = (q1.z - o.x) * idir.x, bx1 = (q2.y - o.x) * idir.x;
      float by0 = (q1.w - o.y) * idir.y, by1 = (q2.z - o.y) * idir.y;
      float bz0 = (q2.x - o.z) * idir.z, bz1 = (q2.w - o.z) * idir.z;
      float tn1 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fmaxf(fminf(bz0, bz1), tmin));
      float tf1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fminf(fmaxf(bz0, bz1), best.t));
      const bool h0 = tn0 <= tf0, h1 = tn1 <= tf1;
      if (h0 && h1) {
        const bool swap = tn1 < tn0;
        const int nearC = swap ? q3.y : q3.x;
        const int farC = swap ? q3.x : q3.y;
        stk[sp * kWave] = farC;
        sp++;
        cur = nearC;
      } else if (h0) {
        cur = q3.x;
      } else if (h1) {
        cur = q3.y;
      } else {
        if (sp == 0) break;
        sp--;
        cur = stk[sp * kWave];
      }
    } else {
      const uint32_t enc = (uint32_t)(-1 - cur);
      const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
      for (uint32_t k = 0; k < cnt; k++) {
        const float4* tp = S.tris + (size_t)(first + k) * 3;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        if (COUNT) nTris++;
        const f3 v0 = mk(a.x, a.y, a.z), e1 = mk(b.x, b.y, b.z), e2 = mk(c.x, c.y, c.z);
        const uint32_t prim = __float_as_uint(a.w), flags = __float_as_uint(b.w);
        const f3 pvec = cross(d, e2);
        const float det = dot(e1, pvec);
        if (MODE == 1 && !(flags & 2u)) {
          if (!(det > 0.0f)) continue;
        } else {
          if (det == 0.0f) continue;
        }
        const float inv = 1.0f / det;
        const f3 tvec = o - v0;
        const float u = dot(tvec, pvec) * inv;
        if (u < 0.0f || u > 1.0f) continue;
        const f3 qvec = cross(tvec, e1);
        const float v = dot(d, qvec) * inv;
        if (v < 0.0f || u + v > 1.0f) continue;
        const float t = dot(e2, qvec) * inv;
        if (!((t > tmin) && (t < tmax))) continue;
        if ((flags & 1u) && alphaTestFails(S, prim, u, v)) continue;  // any-hit shader: IgnoreHit()
        if (MODE == 2) {
          best.prim = 0;
          best.t = t;
          return best;
        }
        if (t < best.t || (t == best.t && best.prim >= 0 && (int)prim < best.prim)) {
          best.prim = (int)prim;
          best.t = t;
          best.u = u;
          best.v = v;
        }
      }
      if (sp == 0) break;
      sp--;
      cur = stk[sp * kWave];
    }
  }
  return best;
}

BD void addCount(DevCounters* c, int idx, uint32_t n) {
  if (n) atomicAdd(&c->v[idx], (unsigned long long)n);
}
// Sum over the active lanes of the wave, one atomic per wave (always-on ray tallies).
BD void waveAddCount(DevCounters* c, int idx, uint32_t n) {
  uint32_t v = n;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
  if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&c->v[idx], (unsigned long long)v);
}

// ------------------------------------------------------------------------------------------------
// vertex planes
// ------------------------------------------------------------------------------------------------
BD float* planePtr(const PathBuf& P, int path, int k, int f) { return P.v + ((size_t)(path * (int)P.D1 + k) * NF + (size_t)f) * P.Np; }
BD f3 ldPlane3(const PathBuf& P, int path, int k, int f, uint32_t p) {
  const float* b = planePtr(P, path, k, f) + p;
  return f3{b[0], b[P.Np], b[2 * (size_t)P.Np]};
}
BD void stPlane3(const PathBuf& P, int path, int k, int f, uint32_t p, f3 v) {
  float* b = planePtr(P, path, k, f) + p;
  b[0] = v.x;
  b[P.Np] = v.y;
  b[2 * (size_t)P.Np] = v.z;
}
BD float ldPlane1(const PathBuf& P, int path, int k, int f, uint32_t p) { return planePtr(P, path, k, f)[p]; }
BD void stPlane1(const PathBuf& P, int path, int k, int f, uint32_t p, float v) { planePtr(P, path, k, f)[p] = v; }

struct Vtx {
  f3 color, pos, N, V, dif, spec;
  float rough;
  bool isSpec;
};
BD Vtx zeroVtx() {
  Vtx v;
  v.color = v.pos = v.N = v.V = v.dif = v.spec = mk(0);
  v.rough = 0.0f;
  v.isSpec = false;
  return v;
}
BD void storeVtx(const PathBuf& P, int path, int k, uint32_t p, const Vtx& v) {
  stPlane3(P, path, k, F_COL, p, v.color);
  stPlane3(P, path, k, F_POS, p, v.pos);
  stPlane3(P, path, k, F_N, p, v.N);
  stPlane3(P, path, k, F_V, p, v.V);
  stPlane3(P, path, k, F_DIF, p, v.dif);
  stPlane3(P, path, k, F_SPEC, p, v.spec);
  stPlane1(P, path, k, F_ROUGH, p, v.rough);
  stPlane1(P, path, k, F_ISSPEC, p, v.isSpec ? 1.0f : 0.0f);
}
// geometry + material of a stored vertex (no colour, no V)
BD void loadSurf(const PathBuf& P, int path, int k, uint32_t p, Vtx& v) {
  v.pos = ldPlane3(P, path, k, F_POS, p);
  v.N = ldPlane3(P, path, k, F_N, p);
  v.dif = ldPlane3(P, path, k, F_DIF, p);
  v.spec = ldPlane3(P, path, k, F_SPEC, p);
  v.rough = ldPlane1(P, path, k, F_ROUGH, p);
  v.isSpec = ldPlane1(P, path, k, F_ISSPEC, p) != 0.0f;
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

template <bool COUNT>
__global__ __launch_bounds__(kWave) void gbuffer_kernel(SceneDev S, GBufferDev G) {
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t Np = (G.y1 - G.y0) * G.W;
  const uint32_t p = blockIdx.x * kWave + threadIdx.x;
  if (p >= Np) return;
  const uint32_t x = p % G.W, y = G.y0 + p / G.W;
  const size_t pix = (size_t)y * G.W + x;

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
  if (h.prim < 0) {
    f3 pd = normalize(d);
    float u = (1.f + atan2_WAR(pd.x, -pd.z) * kInvPi) * 0.5f;
    float v = det_acos(pd.y) * kInvPi;
    f3 c = mk(0);
    if (G.gp.envMap) {
      uint32_t ex = (uint32_t)(u * (float)G.gp.envWidth), ey = (uint32_t)(v * (float)G.gp.envHeight);
      if (ex < G.gp.envWidth && ey < G.gp.envHeight) c = ld3(G.gp.envMap + ((size_t)ey * G.gp.envWidth + ex) * 4);
    } else {
      c = ld3(G.gp.envColor);
    }
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
// init_paths: eye vertex 1 from the G-buffer, light vertex 0 from sampleLight, valid-pixel queue
// (BDPTMain.rt.hlsl:51-103, 124-135; sampleLight BDPTUtils.hlsli:140-152)
// ------------------------------------------------------------------------------------------------
template <bool GGX>
__global__ __launch_bounds__(kWave) void init_paths_kernel(SceneDev S, FrameDev F, PathBuf P) {
  const uint32_t p = blockIdx.x * kWave + threadIdx.x;
  const bool inTile = p < P.Np;
  const size_t pix = (size_t)F.y0 * F.W + p;
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
  wavePush(geom, p, P.queue[0], &P.qcount[0]);
}

// ------------------------------------------------------------------------------------------------
// extend: one bounce of one sub-path type for every queued path.  shootRay + RayClosestHit /
// RayMiss (+ RayAnyHit inside traverse) — globalIlluminationRay.hlsli:1-45, updateRayData
// RayPathData.hlsli:88-109.  A miss stores the reference's "ghost" vertex (quirk 2): colour 0 and
// the previous payload geometry.
// ------------------------------------------------------------------------------------------------
template <bool GGX, bool COUNT>
__global__ __launch_bounds__(kWave) void extend_kernel(SceneDev S, FrameDev F, PathBuf P, int path, int k, int maxK,
                                                       const uint32_t* __restrict__ qin, const uint32_t* __restrict__ countIn,
                                                       uint32_t* __restrict__ qout, uint32_t* __restrict__ countOut) {
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t n = *countIn;
  if (blockIdx.x * kWave >= n) return;
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  const bool active = i < n;
  bool survive = false;
  uint32_t p = 0;
  if (active) {
    p = qin[i];
    const f3 o = ldPlane3(P, path, k, F_POS, p);
    const float* rd = P.rayDir + (size_t)(path * 3) * P.Np + p;
    const f3 d = mk(rd[0], rd[P.Np], rd[2 * (size_t)P.Np]);
    uint32_t nNodes = 0, nTris = 0;
    Hit h = traverse<0, COUNT>(S, o, d, F.p.minT, 1.0e38f, s_stack + threadIdx.x, nNodes, nTris);
    if (COUNT) {
      addCount(F.counters, C_NODE_CLOSEST, nNodes);
      addCount(F.counters, C_TRI_CLOSEST, nTris);
    }
    if (h.prim >= 0) {
      Shading sd = shadeHit<false>(S, (uint32_t)h.prim, h.u, h.v, o);  // V points at WorldRayOrigin()
      const size_t pix = (size_t)F.y0 * F.W + p;
      const uint32_t seed = (path == PATH_EYE) ? initRand((uint32_t)pix, F.p.frameCount) : P.seedL[p];
      f3 L;
      float pdf;
      bool isSpec;
      f3 w = sampleBRDF<GGX>(seed, sd.N, sd.N, sd.V, sd.diffuse, sd.specular, sd.roughness,
                             (F.p.flags & BDPT_PARAM_SPECULAR_FROM_LOBE) != 0, L, pdf, isSpec);
      Vtx v;
      v.color = ldPlane3(P, path, k, F_COL, p) * w;
      v.pos = sd.posW;
      v.N = sd.N;
      v.V = sd.V;
      v.dif = sd.diffuse;
      v.spec = sd.specular;
      v.rough = sd.roughness;
      v.isSpec = isSpec;
      storeVtx(P, path, k + 1, p, v);
      float* wr = P.rayDir + (size_t)(path * 3) * P.Np + p;
      wr[0] = L.x;
      wr[P.Np] = L.y;
      wr[2 * (size_t)P.Np] = L.z;
      survive = (k + 1 < maxK);
    } else {
      Vtx g = zeroVtx();
      if (path == PATH_EYE && k == 1) {
        g.pos = o;  // payload still holds initPayload's values (RayPathData.hlsli:69-86)
      } else {
        loadSurf(P, path, k, p, g);
        g.V = ldPlane3(P, path, k, F_V, p);
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
  }
  waveAddCount(F.counters, path == PATH_EYE ? C_RAYS_EYE : C_RAYS_LIGHT, active ? 1u : 0u);
  wavePush(survive, p, qout, countOut);
}

// ------------------------------------------------------------------------------------------------
// NEE: BDPTMain.rt.hlsl:161-167 with evalDirect (MaterialUtils.hlsli:93-103, 149-184, 288-307).
// One rand per term, drawn also for vertices that do not exist (App. A item 8).  The shadow ray
// is skipped when the visible-light value is already 0 after clampVec (identical output).
// ------------------------------------------------------------------------------------------------
template <bool GGX, bool COUNT>
BD void neeLane(const SceneDev& S, const FrameDev& F, const PathBuf& P, uint32_t i, int* stk, uint32_t& nRays, uint32_t& nNodes,
                uint32_t& nTris) {
  const uint32_t p = P.queue[0][i];
  const size_t pix = (size_t)F.y0 * F.W + p;
  float4* out4 = reinterpret_cast<float4*>(F.out);
  float4 acc = out4[pix];
  uint32_t seed = P.seedL[p];
  const uint32_t D = F.p.maxDepth;
  const int eyeLast = P.eyeLast[p];
  const int lightsCount = (int)S.numLights;
  f3 prevColor = mk(1.0f);  // cameraPath[0].color
  for (uint32_t t = 0; t < D; t++) {
    const float r = nextRand(seed);
    f3 add = mk(0);
    if ((int)(t + 1) <= eyeLast) {
      int lightToSample = (int)(r * (float)lightsCount);
      if (lightToSample > lightsCount - 1) lightToSample = lightsCount - 1;
      const f3 pos = ldPlane3(P, PATH_EYE, (int)t + 1, F_POS, p);
      const f3 N = ldPlane3(P, PATH_EYE, (int)t + 1, F_N, p);
      const f3 dif = ldPlane3(P, PATH_EYE, (int)t + 1, F_DIF, p);
      f3 V = mk(0), spec = mk(0);
      float rough = 0.0f;
      if (GGX) {
        V = ldPlane3(P, PATH_EYE, (int)t + 1, F_V, p);
        spec = ldPlane3(P, PATH_EYE, (int)t + 1, F_SPEC, p);
        rough = ldPlane1(P, PATH_EYE, (int)t + 1, F_ROUGH, p);
      }
      f3 L, lightIntensity;
      float distToLight;
      getLightData(S.sc->lights[lightToSample], pos, L, lightIntensity, distToLight);
      f3 direct = directIfVisible<GGX>((float)lightsCount, L, lightIntensity, N, V, dif, spec, rough);
      f3 shade = prevColor * direct;
      shade = clampVec(shade / (float)(t + 2), F.p.clampUpper);
      if (!allZero(shade)) {
        Hit h = traverse<2, COUNT>(S, pos, L, F.p.minT, distToLight, stk, nNodes, nTris);
        nRays++;
        if (h.prim < 0) add = shade;
      }
      prevColor = ldPlane3(P, PATH_EYE, (int)t + 1, F_COL, p);
    } else {
      prevColor = mk(0);
    }
    if (!(F.p.flags & BDPT_PARAM_NO_NEE)) {
      acc.x = acc.x + add.x;
      acc.y = acc.y + add.y;
      acc.z = acc.z + add.z;
      acc.w = acc.w + 1.0f;
    }
  }
  out4[pix] = acc;
}
template <bool GGX, bool COUNT>
__global__ __launch_bounds__(kWave) void nee_kernel(SceneDev S, FrameDev F, PathBuf P) {
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t n = P.qcount[0];
  if (blockIdx.x * kWave >= n) return;
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  uint32_t nRays = 0, nNodes = 0, nTris = 0;
  if (i < n) neeLane<GGX, COUNT>(S, F, P, i, s_stack + threadIdx.x, nRays, nNodes, nTris);
  waveAddCount(F.counters, C_RAYS_NEE, nRays);
  waveAddCount(F.counters, C_PIX_VALID, i < n ? 1u : 0u);
  if (COUNT) {
    waveAddCount(F.counters, C_NODE_SHADOW, nNodes);
    waveAddCount(F.counters, C_TRI_SHADOW, nTris);
  }
}

// ------------------------------------------------------------------------------------------------
// Light tracing: BDPTMain.rt.hlsl:171-208, getLaunchIndexFromDirection BDPTUtils.hlsli:129-138,
// connectToCamera MaterialUtils.hlsli:10-13.  Splats go to the fixed-point buffer (quirk 6);
// out-of-frame indices are discarded (quirk 8).
// ------------------------------------------------------------------------------------------------
template <bool GGX, bool COUNT>
BD void splatLane(const SceneDev& S, const FrameDev& F, const PathBuf& P, uint32_t i, int* stk, uint32_t& nRays, uint32_t& nNodes,
                  uint32_t& nTris, uint32_t& nSplat) {
  const uint32_t p = P.queue[0][i];
  const int real = P.lightReal[p];
  const f3 camPos = ld3(F.cam.posW);
  const f3 U = ld3(F.cam.cameraU), Vc = ld3(F.cam.cameraV), Wc = ld3(F.cam.cameraW);
  const f3 cameraN = normalize(Wc);
  for (int t = 0; t < real; t++) {
    Vtx lv;
    loadSurf(P, PATH_LIGHT, t + 1, p, lv);
    const f3 dirToCamera = normalize(camPos - lv.pos);
    const float disToCamera = length(camPos - lv.pos);
    if (!(dot(cameraN, dirToCamera) < 0)) continue;
    Hit h = traverse<2, COUNT>(S, lv.pos, dirToCamera, F.p.minT, disToCamera, stk, nNodes, nTris);
    nRays++;
    if (h.prim >= 0) continue;
    // pixel index
    float d1 = dot(dirToCamera, U) / dot(U, U);
    float d2 = dot(dirToCamera, Vc) / dot(Vc, Vc);
    float d3 = dot(dirToCamera, Wc) / dot(Wc, Wc);
    float nx = d1 / d3, ny = -d2 / d3;
    float px = nx * 0.5f + 0.5f, py = ny * 0.5f + 0.5f;
    float fx = rintf(px * (float)F.W - F.p.pixelJitter[0]);
    float fy = rintf(py * (float)F.H - F.p.pixelJitter[1]);
    const bool inside = (fx >= 0.0f && fx < (float)F.W && fy >= 0.0f && fy < (float)F.H);
    if (!inside) continue;
    float theta1 = saturate(fabsf(dot(dirToCamera, cameraN)));
    float theta2 = saturate(fabsf(dot(dirToCamera, lv.N)));
    float invDisToCamera = 1.0f / disToCamera;
    float G = theta1 * theta2 * invDisToCamera * invDisToCamera;
    f3 vV = mk(0);
    if (GGX) vV = ldPlane3(P, PATH_LIGHT, t + 1, F_V, p);
    f3 fr = evalBRDF<GGX>(vV, normalize(camPos - lv.pos), lv.N, lv.N, lv.dif, lv.spec, lv.rough, lv.isSpec);
    f3 prevColor = ldPlane3(P, PATH_LIGHT, t, F_COL, p);
    f3 shade = (prevColor * fr) * G;
    shade = clampVec(shade / (float)(t + 2), F.p.clampUpper);
    if (isnan3(shade)) shade = mk(0);
    unsigned long long* sp = F.splat + ((size_t)(int)fy * F.W + (size_t)(int)fx) * 4;
    const unsigned long long qx = toFixed(shade.x), qy = toFixed(shade.y), qz = toFixed(shade.z);
    if (qx) atomicAdd(&sp[0], qx);
    if (qy) atomicAdd(&sp[1], qy);
    if (qz) atomicAdd(&sp[2], qz);
    atomicAdd(&sp[3], 1ull);
    nSplat++;
  }
}
template <bool GGX, bool COUNT>
__global__ __launch_bounds__(kWave) void splat_kernel(SceneDev S, FrameDev F, PathBuf P) {
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t n = P.qcount[0];
  if (blockIdx.x * kWave >= n) return;
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  uint32_t nRays = 0, nNodes = 0, nTris = 0, nSplat = 0;
  if (i < n) splatLane<GGX, COUNT>(S, F, P, i, s_stack + threadIdx.x, nRays, nNodes, nTris, nSplat);
  waveAddCount(F.counters, C_RAYS_SPLAT, nRays);
  waveAddCount(F.counters, C_SPLATS, nSplat);
  if (COUNT) {
    waveAddCount(F.counters, C_NODE_SHADOW, nNodes);
    waveAddCount(F.counters, C_TRI_SHADOW, nTris);
  }
}

// ------------------------------------------------------------------------------------------------
// Vertex connection: BDPTMain.rt.hlsl:212-233, evalGWithoutV / getUnweightedContribution
// BDPTUtils.hlsli:172-224 (uniform 1/totalLength weights; aL indexes the light path with
// cameraIndex-1, sic :198).  Pairs are visited in the reference's order because every write
// saturates.  A pair whose contribution is exactly 0 only matters through that saturate, so its
// shadow ray is traced only while the pixel has not been saturated yet.
// ------------------------------------------------------------------------------------------------
template <bool GGX>
BD void loadConnVtx(const PathBuf& P, int path, int k, int last, uint32_t p, Vtx& v) {
  if (k > last) {
    v = zeroVtx();
    return;
  }
  v.pos = ldPlane3(P, path, k, F_POS, p);
  v.N = ldPlane3(P, path, k, F_N, p);
  v.dif = ldPlane3(P, path, k, F_DIF, p);
  if (GGX) {
    v.spec = ldPlane3(P, path, k, F_SPEC, p);
    v.rough = ldPlane1(P, path, k, F_ROUGH, p);
    v.isSpec = ldPlane1(P, path, k, F_ISSPEC, p) != 0.0f;
  } else {
    v.spec = mk(0);
    v.rough = 0.0f;
    v.isSpec = false;
  }
}

template <bool GGX, bool COUNT>
BD void connectLane(const SceneDev& S, const FrameDev& F, const PathBuf& P, uint32_t i, int* stk, uint32_t& nRays, uint32_t& nNodes,
                    uint32_t& nTris) {
  const uint32_t p = P.queue[0][i];
  const size_t pix = (size_t)F.y0 * F.W + p;
  float4* out4 = reinterpret_cast<float4*>(F.out);
  float4 acc = out4[pix];
  const int D = (int)F.p.maxDepth;
  const int eyeLast = P.eyeLast[p], lightLast = P.lightLast[p];
  const f3 camPos = ld3(F.cam.posW);
  bool sat = false;
  for (int totalLength = 2; totalLength <= D; totalLength++) {
    for (int cameraLength = 1; cameraLength <= D - 1; cameraLength++) {
      if (cameraLength > totalLength) continue;  // undefined in the reference (uint underflow, quirk 3)
      const int lightLength = totalLength - cameraLength;
      Vtx ce, le;
      loadConnVtx<GGX>(P, PATH_EYE, cameraLength, eyeLast, p, ce);
      loadConnVtx<GGX>(P, PATH_LIGHT, lightLength, lightLast, p, le);
      // evalGWithoutV
      const f3 vecAB = le.pos - ce.pos;
      const float invLengthAB = 1.0f / length(vecAB);
      const f3 dirG = vecAB * invLengthAB;
      const float cosA = fabsf(dot(ce.N, dirG));
      const float cosB = fabsf(dot(le.N, dirG));
      const float G = cosA * cosB * invLengthAB * invLengthAB;
      // getUnweightedContribution
      f3 c = mk(0);
      if (lightLength != 0) {
        const f3 connectDir = normalize(ce.pos - le.pos);
        const f3 lprev = (lightLength - 1 <= lightLast) ? ldPlane3(P, PATH_LIGHT, lightLength - 1, F_POS, p) : mk(0);
        f3 wo = normalize(lprev - le.pos);
        f3 fsL = evalBRDF<GGX>(connectDir, wo, le.N, le.N, le.dif, le.spec, le.rough, le.isSpec);
        if (allZero(fsL)) {
          c = fsL;
        } else {
          f3 cprevPos, aE;
          if (cameraLength - 1 == 0) {
            cprevPos = camPos;
            aE = mk(1.0f);
          } else if (cameraLength - 1 <= eyeLast) {
            cprevPos = ldPlane3(P, PATH_EYE, cameraLength - 1, F_POS, p);
            aE = ldPlane3(P, PATH_EYE, cameraLength - 1, F_COL, p);
          } else {
            cprevPos = mk(0);
            aE = mk(0);
          }
          wo = normalize(cprevPos - ce.pos);
          f3 fsE = evalBRDF<GGX>(-connectDir, wo, ce.N, ce.N, ce.dif, ce.spec, ce.rough, ce.isSpec);
          if (allZero(fsE)) {
            c = fsE;
          } else {
            const f3 aL = (cameraLength - 1 <= lightLast) ? ldPlane3(P, PATH_LIGHT, cameraLength - 1, F_COL, p) : mk(0);
            f3 cst = (fsL * G) * fsE;
            c = (aL * cst) * aE;
          }
        }
      }
      f3 shade = clampVec(c / (float)totalLength, F.p.clampUpper);
      if (isnan3(shade)) shade = mk(0);
      if (allZero(shade) && sat) continue;
      const float lengthAB = length(le.pos - ce.pos);
      const f3 dirAB = (le.pos - ce.pos) / lengthAB;
      Hit h = traverse<2, COUNT>(S, ce.pos, dirAB, F.p.minT, lengthAB, stk, nNodes, nTris);
      nRays++;
      if (h.prim < 0) {
        acc.x = saturate(acc.x + shade.x);
        acc.y = saturate(acc.y + shade.y);
        acc.z = saturate(acc.z + shade.z);
        acc.w = saturate(acc.w + 1.0f);
        sat = true;
      }
    }
  }
  out4[pix] = acc;
}
template <bool GGX, bool COUNT>
__global__ __launch_bounds__(kWave) void connect_kernel(SceneDev S, FrameDev F, PathBuf P) {
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t n = P.qcount[0];
  if (blockIdx.x * kWave >= n) return;
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  uint32_t nRays = 0, nNodes = 0, nTris = 0;
  if (i < n) connectLane<GGX, COUNT>(S, F, P, i, s_stack + threadIdx.x, nRays, nNodes, nTris);
  waveAddCount(F.counters, C_RAYS_CONNECT, nRays);
  if (COUNT) {
    waveAddCount(F.counters, C_NODE_SHADOW, nNodes);
    waveAddCount(F.counters, C_TRI_SHADOW, nTris);
  }
}

// out = saturate(out + splat) where at least one splat landed
__global__ void resolve_kernel(const unsigned long long* __restrict__ splat, uint32_t splatRow0, float4* __restrict__ out, uint32_t W,
                               uint32_t y0, uint32_t y1) {
  const size_t nTile = (size_t)(y1 - y0) * W;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < nTile; p += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = (size_t)y0 * W + p;
    const size_t sidx = (pix - (size_t)splatRow0 * W) * 4;
    const ulonglong2 a = reinterpret_cast<const ulonglong2*>(splat + sidx)[0];
    const ulonglong2 b = reinterpret_cast<const ulonglong2*>(splat + sidx)[1];
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
  __shared__ int s_stack[kStackEntries * kWave];
  const uint32_t i = blockIdx.x * kWave + threadIdx.x;
  if (i >= n) return;
  const float* r = rays + (size_t)i * 8;
  uint32_t a = 0, b = 0;
  Hit h = traverse<MODE, false>(S, ld3(r), ld3(r + 3), r[6], r[7], s_stack + threadIdx.x, a, b);
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
static inline uint32_t blocksFor(uint64_t n) { return (uint32_t)((n + kWave - 1) / kWave); }

void launchGBuffer(const SceneDev& S, const GBufferDev& G, hipStream_t st) {
  const uint32_t Np = (G.y1 - G.y0) * G.W;
  if (!Np) return;
  if (G.counters)
    hipLaunchKernelGGL(gbuffer_kernel<true>, dim3(blocksFor(Np)), dim3(kWave), 0, st, S, G);
  else
    hipLaunchKernelGGL(gbuffer_kernel<false>, dim3(blocksFor(Np)), dim3(kWave), 0, st, S, G);
}

void launchInitPaths(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  if (F.p.matIndex == 0)
    hipLaunchKernelGGL(init_paths_kernel<true>, dim3(blocksFor(P.Np)), dim3(kWave), 0, st, S, F, P);
  else
    hipLaunchKernelGGL(init_paths_kernel<false>, dim3(blocksFor(P.Np)), dim3(kWave), 0, st, S, F, P);
}

#define BDPT_DISPATCH(KERNEL, GRID, ...)                                                         \
  do {                                                                                           \
    const bool ggx_ = (F.p.matIndex == 0), cnt_ = (F.p.flags & BDPT_PARAM_COUNTERS) != 0;                       \
    if (ggx_ && cnt_)                                                                            \
      hipLaunchKernelGGL((KERNEL<true, true>), dim3(GRID), dim3(kWave), 0, st, __VA_ARGS__);     \
    else if (ggx_)                                                                               \
      hipLaunchKernelGGL((KERNEL<true, false>), dim3(GRID), dim3(kWave), 0, st, __VA_ARGS__);    \
    else if (cnt_)                                                                               \
      hipLaunchKernelGGL((KERNEL<false, true>), dim3(GRID), dim3(kWave), 0, st, __VA_ARGS__);    \
    else                                                                                         \
      hipLaunchKernelGGL((KERNEL<false, false>), dim3(GRID), dim3(kWave), 0, st, __VA_ARGS__);   \
  } while (0)

void launchExtend(const SceneDev& S, const FrameDev& F, const PathBuf& P, int path, int k, int maxK, const uint32_t* qin,
                  const uint32_t* countIn, uint32_t* qout, uint32_t* countOut, hipStream_t st) {
  if (!P.Np) return;
  BDPT_DISPATCH(extend_kernel, blocksFor(P.Np), S, F, P, path, k, maxK, qin, countIn, qout, countOut);
}
void launchNee(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  BDPT_DISPATCH(nee_kernel, blocksFor(P.Np), S, F, P);
}
void launchSplat(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  BDPT_DISPATCH(splat_kernel, blocksFor(P.Np), S, F, P);
}
void launchConnect(const SceneDev& S, const FrameDev& F, const PathBuf& P, hipStream_t st) {
  if (!P.Np) return;
  BDPT_DISPATCH(connect_kernel, blocksFor(P.Np), S, F, P);
}
void launchResolve(const unsigned long long* splat, uint32_t splatRow0, float* out, uint32_t W, uint32_t y0, uint32_t y1,
                   hipStream_t st) {
  const uint64_t n = (uint64_t)(y1 - y0) * W;
  if (!n) return;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(resolve_kernel, dim3(grid), dim3(256), 0, st, splat, splatRow0, reinterpret_cast<float4*>(out), W, y0, y1);
}
void launchAccumulate(float* last, float* cur, uint32_t accumCount, uint32_t maxAccum, uint64_t numTexels, hipStream_t st) {
  if (!numTexels) return;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((numTexels + 255) / 256, 2048);
  hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<float4*>(last),
                     reinterpret_cast<float4*>(cur), accumCount, maxAccum, numTexels);
}
void launchTestRng(const uint32_t* v0, const uint32_t* v1, uint32_t n, uint32_t draws, uint32_t* states, float* floats,
                   hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(test_rng_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v0, v1, n, draws, states, floats);
}
void launchTestTrace(const SceneDev& S, const float* rays, uint32_t n, int mode, int32_t* prim, float* tuv, hipStream_t st) {
  if (!n) return;
  if (mode == 0)
    hipLaunchKernelGGL(test_trace_kernel<0>, dim3(blocksFor(n)), dim3(kWave), 0, st, S, rays, n, prim, tuv);
  else if (mode == 1)
    hipLaunchKernelGGL(test_trace_kernel<1>, dim3(blocksFor(n)), dim3(kWave), 0, st, S, rays, n, prim, tuv);
  else
    hipLaunchKernelGGL(test_trace_kernel<2>, dim3(blocksFor(n)), dim3(kWave), 0, st, S, rays, n, prim, tuv);
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
