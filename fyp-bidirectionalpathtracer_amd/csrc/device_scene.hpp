// device_scene.hpp — texture sampling, material decode, vertex fetch and shading data on the device.
#pragma once
#include "device_math.hpp"
#include "kernels.h"

namespace bdpt {
#define BD __device__ __forceinline__

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
BD f4 sampleBilinearT(const SceneDev& S, const TexDev& t, float u, float v) {
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
BD f4 sampleBilinear(const SceneDev& S, int texId, float u, float v) { return sampleBilinearT(S, S.textures[texId], u, v); }
// Falcor ShadingUtils/Shading.slang:88-94
BD f4 sampleTexture(const SceneDev& S, int texId, float u, float v, f4 factor, uint32_t mode) {
  if (mode == BDPT_CHANNEL_UNUSED) return f4{0, 0, 0, 0};
  if (mode == BDPT_CHANNEL_CONST || texId < 0) return factor;
  return sampleBilinear(S, texId, u, v);
}
// the same with the texture's descriptor already fetched (SceneDev::matTex: fetched beside the material record, not after it)
BD f4 sampleTextureT(const SceneDev& S, const TexDev& t, int texId, float u, float v, f4 factor, uint32_t mode) {
  if (mode == BDPT_CHANNEL_UNUSED) return f4{0, 0, 0, 0};
  if (mode == BDPT_CHANNEL_CONST || texId < 0) return factor;
  return sampleBilinearT(S, t, u, v);
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

// BDPT/BDPTUtils.hlsli:115-127.  Everything the any-hit alpha test of a non-opaque triangle reads sits in ONE 64-byte
// record (built by bdpt_set_scene: the three texture coordinates, the material's threshold and constant alpha, how the base
// colour is sampled, and the base-colour texture's address and size), so a candidate hit costs one record fetch and four
// texel fetches instead of the chain shading record -> material -> texture descriptor -> texels.  Same arithmetic as
// shadeHit + sampleTexture on the alpha channel (texel alpha = byte / 255, bilinear, wrap).
//   f4[0] = uv0, uv1   f4[1] = uv2, alphaThreshold, baseColor.a   f4[2] = mode (0 unused, 1 constant, 2 texture), width, height
//   f4[3] = texel address (two words)
BD bool alphaTestFails(const SceneDev& S, uint32_t rec, float bu, float bv) {
  const float4* r = S.alphaRecs + (size_t)rec * 4;
  const float4 a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3];
  const uint32_t mode = __float_as_uint(a2.x);
  float alpha = 0.0f;  // BDPT_CHANNEL_UNUSED: sampleTexture returns 0
  if (mode == 1u) {
    alpha = a1.w;
  } else if (mode == 2u) {
    float u = 0, v = 0;
    const float b0 = 1.0f - bu - bv;
    u += a0.x * b0;
    v += a0.y * b0;
    u += a0.z * bu;
    v += a0.w * bu;
    u += a1.x * bv;
    v += a1.y * bv;
    const int tw = (int)__float_as_uint(a2.y), th = (int)__float_as_uint(a2.z);
    const uint8_t* px = reinterpret_cast<const uint8_t*>(((unsigned long long)__float_as_uint(a3.y) << 32) | (unsigned long long)__float_as_uint(a3.x));
    const float x = u * (float)tw - 0.5f, y = v * (float)th - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int ix0 = wrapi((int)x0, tw), iy0 = wrapi((int)y0, th);
    const int ix1 = wrapi(ix0 + 1, tw), iy1 = wrapi(iy0 + 1, th);
    const float t00 = (float)px[((size_t)iy0 * tw + (size_t)ix0) * 4 + 3] / 255.0f, t10 = (float)px[((size_t)iy0 * tw + (size_t)ix1) * 4 + 3] / 255.0f;
    const float t01 = (float)px[((size_t)iy1 * tw + (size_t)ix0) * 4 + 3] / 255.0f, t11 = (float)px[((size_t)iy1 * tw + (size_t)ix1) * 4 + 3] / 255.0f;
    const float top = t00 + (t10 - t00) * fx, bot = t01 + (t11 - t01) * fx;
    alpha = top + (bot - top) * fy;
  }
  return alpha < a1.z;
}

// Two alpha tests side by side (a leaf's two triangles): both records are fetched before either texel quad, both quads
// before either result is used.  need0 / need1 say which to run; same arithmetic as alphaTestFails.
struct AlphaIn {
  float4 a0, a1, a2, a3;
};
BD void alphaTexels(const AlphaIn& r, float bu, float bv, bool run, float& fx, float& fy, float& t00, float& t10, float& t01, float& t11) {
  fx = fy = t00 = t10 = t01 = t11 = 0.0f;
  if (!run || __float_as_uint(r.a2.x) != 2u) return;
  float u = 0, v = 0;
  const float b0 = 1.0f - bu - bv;
  u += r.a0.x * b0;
  v += r.a0.y * b0;
  u += r.a0.z * bu;
  v += r.a0.w * bu;
  u += r.a1.x * bv;
  v += r.a1.y * bv;
  const int tw = (int)__float_as_uint(r.a2.y), th = (int)__float_as_uint(r.a2.z);
  const uint8_t* px = reinterpret_cast<const uint8_t*>(((unsigned long long)__float_as_uint(r.a3.y) << 32) | (unsigned long long)__float_as_uint(r.a3.x));
  const float x = u * (float)tw - 0.5f, y = v * (float)th - 0.5f;
  const float x0 = floorf(x), y0 = floorf(y);
  fx = x - x0;
  fy = y - y0;
  const int ix0 = wrapi((int)x0, tw), iy0 = wrapi((int)y0, th);
  const int ix1 = wrapi(ix0 + 1, tw), iy1 = wrapi(iy0 + 1, th);
  t00 = (float)px[((size_t)iy0 * tw + (size_t)ix0) * 4 + 3];
  t10 = (float)px[((size_t)iy0 * tw + (size_t)ix1) * 4 + 3];
  t01 = (float)px[((size_t)iy1 * tw + (size_t)ix0) * 4 + 3];
  t11 = (float)px[((size_t)iy1 * tw + (size_t)ix1) * 4 + 3];
}
BD bool alphaVerdict(const AlphaIn& r, float fx, float fy, float t00, float t10, float t01, float t11) {
  const uint32_t mode = __float_as_uint(r.a2.x);
  float alpha = 0.0f;
  if (mode == 1u) {
    alpha = r.a1.w;
  } else if (mode == 2u) {
    const float w00 = t00 / 255.0f, w10 = t10 / 255.0f, w01 = t01 / 255.0f, w11 = t11 / 255.0f;
    const float top = w00 + (w10 - w00) * fx, bot = w01 + (w11 - w01) * fx;
    alpha = top + (bot - top) * fy;
  }
  return alpha < r.a1.z;
}
BD void alphaTestFails2(const SceneDev& S, bool need0, uint32_t rec0, float u0, float v0, bool need1, uint32_t rec1, float u1, float v1,
                        bool& fail0, bool& fail1) {
  AlphaIn r0, r1;
  r0.a0 = r0.a1 = r0.a2 = r0.a3 = r1.a0 = r1.a1 = r1.a2 = r1.a3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (need0) {
    const float4* p = S.alphaRecs + (size_t)rec0 * 4;
    r0.a0 = p[0];
    r0.a1 = p[1];
    r0.a2 = p[2];
    r0.a3 = p[3];
  }
  if (need1) {
    const float4* p = S.alphaRecs + (size_t)rec1 * 4;
    r1.a0 = p[0];
    r1.a1 = p[1];
    r1.a2 = p[2];
    r1.a3 = p[3];
  }
  float fx0, fy0, a00, a10, a01, a11, fx1, fy1, b00, b10, b01, b11;
  alphaTexels(r0, u0, v0, need0, fx0, fy0, a00, a10, a01, a11);
  alphaTexels(r1, u1, v1, need1, fx1, fy1, b00, b10, b01, b11);
  fail0 = need0 && alphaVerdict(r0, fx0, fy0, a00, a10, a01, a11);
  fail1 = need1 && alphaVerdict(r1, fx1, fy1, b00, b10, b01, b11);
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

  const uint32_t matId = __float_as_uint(r6.x);
  const MatDev m = loadMaterial(S, matId);
  // the descriptors of the material's base, specular and emissive textures, fetched beside the material record
  const TexDev tBase = S.matTex[(size_t)matId * 4], tSpec = S.matTex[(size_t)matId * 4 + 1], tEmis = S.matTex[(size_t)matId * 4 + 2];
  Shading sd;
  f4 base = sampleTextureT(S, tBase, m.texBase, u, v, m.baseColor, BDPT_FLAG_DIFFUSE_TYPE(m.flags));
  sd.opacity = m.baseColor.w;
  sd.posW = posW;
  sd.V = normalize(camPosW - posW);
  sd.N = normalize(normalW);
  f4 spec = sampleTextureT(S, tSpec, m.texSpec, u, v, m.specular, BDPT_FLAG_SPECULAR_TYPE(m.flags));
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
  f4 em = sampleTextureT(S, tEmis, m.texEmis, u, v, f4{m.emissive.x, m.emissive.y, m.emissive.z, 1.0f}, BDPT_FLAG_EMISSIVE_TYPE(m.flags));
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

#undef BD
}  // namespace bdpt
