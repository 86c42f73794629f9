// device_math.hpp — device-side arithmetic of the BDPT pass for gfx950.
//
// Arithmetic contract (DESIGN.md §"Numerics"): IEEE fp32, no FMA contraction
// (-ffp-contract=off), operations evaluated in the order written, correctly rounded
// divide and sqrt, and fixed polynomial forms for the transcendental functions the
// reference shaders call.  Each function cites the reference shader lines it implements
// (paths relative to /root/reference/src; BDPT/ = BidirectionalPathtracing/Data/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bdpt.h"

namespace bdpt {

#define BD __device__ __forceinline__

struct f3 {
  float x, y, z;
};
BD f3 mk(float x, float y, float z) { return f3{x, y, z}; }
BD f3 mk(float s) { return f3{s, s, s}; }
BD f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
BD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
BD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
BD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
BD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
BD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
BD f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
BD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
BD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
BD f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
BD float length(f3 a) { return sqrtf(dot(a, a)); }
BD f3 normalize(f3 a) {
  float inv = 1.0f / sqrtf(dot(a, a));
  return a * inv;
}
BD f3 absv(f3 a) { return f3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// HLSL max/min/saturate: the non-NaN operand wins; saturate(NaN) = 0.
BD float maxf(float a, float b) { return (b > a) ? b : ((a == a) ? a : b); }
BD float minf(float a, float b) { return (b < a) ? b : ((a == a) ? a : b); }
BD float saturate(float x) {
  float y = (x > 0.0f) ? x : 0.0f;
  return (y < 1.0f) ? y : 1.0f;
}
BD float clampUp(float x, float hi) {
  float y = (x > 0.0f) ? x : 0.0f;
  return (y < hi) ? y : hi;
}
BD bool isnan3(f3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }
BD bool allZero(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
BD f3 lerp3(f3 a, f3 b, float s) { return a + (b - a) * s; }

constexpr float kPi = 3.14159265358979323846f;      // M_PI, Falcor Data/HostDeviceSharedMacros.h:170
constexpr float kInvPi = 0.318309886183790671538f;  // M_1_PI, BDPT/MaterialUtils.hlsli:2

// ---- fixed-form transcendental functions (HLSL leaves sin/cos/acos/atan/pow precision to the driver)
BD void det_sincos2pi(float u, float& s, float& c) {
  float t = u * 4.0f;
  float q = floorf(t + 0.5f);
  float r = t - q;
  float a = r * 1.57079632679489661923f;
  float a2 = a * a;
  float sp = -1.0f / 5040.0f + a2 * (1.0f / 362880.0f);
  sp = 1.0f / 120.0f + a2 * sp;
  sp = -1.0f / 6.0f + a2 * sp;
  sp = 1.0f + a2 * sp;
  float sa = a * sp;
  float cp = 1.0f / 40320.0f + a2 * (-1.0f / 3628800.0f);
  cp = -1.0f / 720.0f + a2 * cp;
  cp = 1.0f / 24.0f + a2 * cp;
  cp = -0.5f + a2 * cp;
  float ca = 1.0f + a2 * cp;
  int qi = ((int)q) & 3;
  float s0 = (qi & 1) ? ca : sa;
  float c0 = (qi & 1) ? sa : ca;
  s = (qi & 2) ? -s0 : s0;
  c = ((qi == 1) || (qi == 2)) ? -c0 : c0;
}
BD float det_acos(float x) {  // Abramowitz & Stegun 4.4.46
  float ax = fabsf(x);
  if (ax > 1.0f) ax = 1.0f;
  float p = -0.0012624911f;
  p = 0.0066700901f + ax * p;
  p = -0.0170881256f + ax * p;
  p = 0.0308918810f + ax * p;
  p = -0.0501743046f + ax * p;
  p = 0.0889789874f + ax * p;
  p = -0.2145988016f + ax * p;
  p = 1.5707963050f + ax * p;
  float r = sqrtf(1.0f - ax) * p;
  return (x < 0.0f) ? (kPi - r) : r;
}
BD float det_atan(float z) {  // Abramowitz & Stegun 4.4.49
  float az = fabsf(z);
  bool inv = az > 1.0f;
  float w = inv ? (1.0f / az) : az;
  float w2 = w * w;
  float p = 0.0028662257f;
  p = -0.0161657367f + w2 * p;
  p = 0.0429096138f + w2 * p;
  p = -0.0752896400f + w2 * p;
  p = 0.1065626393f + w2 * p;
  p = -0.1420889944f + w2 * p;
  p = 0.1999355085f + w2 * p;
  p = -0.3333314528f + w2 * p;
  p = 1.0f + w2 * p;
  float r = w * p;
  if (inv) r = 1.57079632679489661923f - r;
  return (z < 0.0f) ? -r : r;
}
BD float det_pow5(float x) {
  float x2 = x * x;
  return x2 * x2 * x;
}

// float -> half (round to nearest even) and back, in integer arithmetic so host and device agree.
BD uint16_t f32_to_f16(float f) {
  uint32_t x = __float_as_uint(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (ax < 0x33000001u) return (uint16_t)sign;
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t he = (e < -14) ? 0u : (uint32_t)(e + 15);
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t h = (he == 0) ? q : (((he - 1) << 10) + q);
  return (uint16_t)(sign | h);
}
BD float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    x = (m == 0) ? sign : (__float_as_uint((float)m * 5.9604644775390625e-08f) | sign);
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  return __uint_as_float(x);
}

// ---- RNG: BDPT/BDPTUtils.hlsli:91-110 -------------------------------------------------------
BD uint32_t initRand(uint32_t val0, uint32_t val1) {
  uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
  for (uint32_t n = 0; n < 16; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
BD float nextRand(uint32_t& s) {
  s = 1664525u * s + 1013904223u;
  return (float)(s & 0x00FFFFFFu) / (float)0x01000000;
}

// ---- BRDF utilities: BDPT/BRDFUtils.hlsli, BDPT/MaterialUtils.hlsli --------------------------
BD float luminance(f3 rgb) { return dot(rgb, mk(0.2126f, 0.7152f, 0.0722f)); }  // Falcor HostDeviceSharedCode.h:256-259
BD float probabilityToSampleDiffuse(f3 dif, f3 spec) {                          // MaterialUtils.hlsli:22-27
  float lumDiffuse = maxf(0.01f, luminance(dif));
  float lumSpecular = maxf(0.01f, luminance(spec));
  return lumDiffuse / (lumDiffuse + lumSpecular);
}
BD f3 getPerpendicularVector(f3 u) {  // MaterialUtils.hlsli:31-38
  f3 a = absv(u);
  uint32_t xm = ((a.x - a.y) < 0 && (a.x - a.z) < 0) ? 1 : 0;
  uint32_t ym = (a.y - a.z) < 0 ? (1 ^ xm) : 0;
  uint32_t zm = 1 ^ (xm | ym);
  return cross(u, mk((float)xm, (float)ym, (float)zm));
}
BD f3 getCosHemisphereSample(uint32_t& seed, f3 hitNorm) {  // MaterialUtils.hlsli:41-54
  float r0 = nextRand(seed);
  float r1 = nextRand(seed);
  f3 bitangent = getPerpendicularVector(hitNorm);
  f3 tangent = cross(bitangent, hitNorm);
  float r = sqrtf(r0);
  float sn, cs;
  det_sincos2pi(r1, sn, cs);
  return tangent * (r * cs) + bitangent * (r * sn) + hitNorm * sqrtf(maxf(0.0f, 1.0f - r0));
}
BD f3 sampleUnitSphere(uint32_t& seed) {  // MaterialUtils.hlsli:56-63
  f3 p = mk(2.0f, 2.0f, 2.0f);
  while (length(p) > 1.0f) {
    float a = nextRand(seed) * 2.0f - 1.0f;
    float b = nextRand(seed) * 2.0f - 1.0f;
    float c = nextRand(seed) * 2.0f - 1.0f;
    p = mk(a, b, c);
  }
  return p;
}
BD float ggxNormalDistribution(float NdotH, float roughness) {  // BRDFUtils.hlsli:5-10
  float a2 = roughness * roughness;
  float d = ((NdotH * a2 - NdotH) * NdotH + 1);
  return a2 / maxf(0.001f, (d * d * kPi));
}
BD float ggxSchlickMaskingTerm(float NdotL, float NdotV, float roughness) {  // BRDFUtils.hlsli:15-30
  float k = roughness * roughness / 2;
  float g_v = NdotV / (NdotV * (1 - k) + k);
  float g_l = NdotL / (NdotL * (1 - k) + k);
  return g_v * g_l;
}
BD f3 schlickFresnel(f3 f0, float u) { return f0 + (mk(1.0f) - f0) * det_pow5(1.0f - u); }  // BRDFUtils.hlsli:35-38
BD f3 getGGXMicrofacet(uint32_t& seed, float roughness, f3 hitNorm) {                       // BRDFUtils.hlsli:44-61
  float r0 = nextRand(seed);
  float r1 = nextRand(seed);
  f3 B = getPerpendicularVector(hitNorm);
  f3 T = cross(B, hitNorm);
  float a2 = roughness * roughness;
  float cosThetaH = sqrtf(maxf(0.0f, (1.0f - r0) / ((a2 - 1.0f) * r0 + 1)));
  float sinThetaH = sqrtf(maxf(0.0f, 1.0f - cosThetaH * cosThetaH));
  float sn, cs;
  det_sincos2pi(r1, sn, cs);
  return T * (sinThetaH * cs) + B * (sinThetaH * sn) + hitNorm * cosThetaH;
}
BD f3 ggxLighting(f3 H, f3 L, f3 N, float NdotL, float NdotV, float rough, f3 spec, float& ggxProb) {  // BRDFUtils.hlsli:63-73
  float NdotH = saturate(dot(N, H));
  float LdotH = saturate(dot(L, H));
  float D = ggxNormalDistribution(NdotH, rough);
  float G = ggxSchlickMaskingTerm(NdotL, NdotV, rough);
  f3 F = schlickFresnel(spec, LdotH);
  ggxProb = D * NdotH / (4 * LdotH);
  return (D * G) * F / (4 * NdotL * NdotV);
}

// MaterialUtils.hlsli:130-141, 209-252, 321-329.  The seed is taken BY VALUE.
// isSpecular: the GGX branch never writes its `out bool` (undefined in HLSL) -> false unless fromLobe.
template <bool GGX>
BD f3 sampleBRDF(uint32_t seed, f3 N, f3 noNormalN, f3 V, f3 dif, f3 spec, float rough, bool fromLobe, f3& L, float& pdf,
                 bool& isSpecular) {
  if (GGX) {
    float probDiffuse = probabilityToSampleDiffuse(dif, spec);
    bool chooseDiffuse = (nextRand(seed) < probDiffuse);
    float NdotV = saturate(dot(N, V));
    isSpecular = fromLobe ? !chooseDiffuse : false;
    if (chooseDiffuse) {
      L = getCosHemisphereSample(seed, N);
      if (dot(noNormalN, L) <= 0.0f) {
        pdf = 0;
        return mk(0);
      }
      float NdotL = saturate(dot(N, L));
      pdf = (NdotL * kInvPi) * probDiffuse;
      return dif / probDiffuse;
    } else {
      f3 H = getGGXMicrofacet(seed, rough, N);
      L = normalize(H * (2.f * dot(V, H)) - V);
      if (dot(noNormalN, L) <= 0.0f) {
        pdf = 0;
        return mk(0);
      }
      float NdotL = saturate(dot(N, L));
      float ggxProb;
      f3 ggxTerm = ggxLighting(H, L, N, NdotL, NdotV, rough, spec, ggxProb);
      pdf = ggxProb * (1.0f - probDiffuse);
      return ggxTerm * NdotL / (ggxProb * (1.0f - probDiffuse));
    }
  } else {
    isSpecular = false;
    L = getCosHemisphereSample(seed, N);
    pdf = saturate(dot(N, L)) * kInvPi;
    return dif;
  }
}

// MaterialUtils.hlsli:105-115, 186-207, 309-314 (Lambertian: dif, no 1/pi, no cosine test — sic)
template <bool GGX>
BD f3 evalBRDF(f3 V, f3 L, f3 N, f3 noNormalN, f3 dif, f3 spec, float rough, bool isSpecular) {
  if (!GGX) return dif;
  if (!isSpecular) {
    if (dot(noNormalN, L) <= 0.0f) return mk(0);
    return dif * kInvPi;
  } else {
    f3 H = normalize(L + V);
    if (dot(noNormalN, L) <= 0.0f) return mk(0);
    float NdotL = saturate(dot(N, L));
    float NdotV = saturate(dot(N, V));
    float ggxProb;
    return ggxLighting(H, L, N, NdotL, NdotV, rough, spec, ggxProb);
  }
}

// Falcor ShadingUtils/Lights.slang:54-102 + BDPT/MaterialUtils.hlsli:67-85
BD void getLightData(const bdpt_light& l, f3 hitPos, f3& toLight, f3& lightIntensity, float& distToLight) {
  f3 lpos = ld3(l.posW), ldir = ld3(l.dirW), lint = ld3(l.intensity);
  f3 lsL, lsPos, lsDiffuse;
  if (l.type == BDPT_LIGHT_DIRECTIONAL) {
    lsDiffuse = lint;
    lsL = -normalize(ldir);
    float dist = length(hitPos - lpos);
    lsPos = hitPos - ldir * dist;
  } else {
    lsPos = lpos;
    lsL = lpos - hitPos;
    float distSquared = dot(lsL, lsL);
    lsL = (distSquared > 1e-5f) ? normalize(lsL) : mk(0);
    float falloff = 1 / ((0.01f * 0.01f) + distSquared);
    float cosTheta = -dot(lsL, ldir);
    if (cosTheta < l.cosOpeningAngle) {
      falloff = 0;
    } else if (l.penumbraAngle > 0) {
      float deltaAngle = l.openingAngle - det_acos(cosTheta);
      falloff *= saturate((deltaAngle - l.penumbraAngle) / l.penumbraAngle);
    }
    lsDiffuse = lint * falloff;
  }
  toLight = normalize(lsL);
  lightIntensity = lsDiffuse;
  distToLight = length(lsPos - hitPos);
}

// ggxDirect / lambertianDirect (MaterialUtils.hlsli:149-184, 288-307) with the shadow term factored
// out: returns the value for a VISIBLE light; an occluded one contributes exactly 0 after clampVec.
template <bool GGX>
BD f3 directIfVisible(float lightsCount, f3 L, f3 lightIntensity, f3 N, f3 V, f3 dif, f3 spec, float rough) {
  if (GGX) {
    float NdotL = saturate(dot(N, L));
    float shadowMult = lightsCount;
    f3 H = normalize(V + L);
    float NdotH = saturate(dot(N, H));
    float LdotH = saturate(dot(L, H));
    float NdotV = saturate(dot(N, V));
    float D = ggxNormalDistribution(NdotH, rough);
    float G = ggxSchlickMaskingTerm(NdotL, NdotV, rough);
    f3 F = schlickFresnel(spec, LdotH);
    f3 ggxTerm = (D * G) * F / (4 * NdotV);
    return (shadowMult * lightIntensity) * (ggxTerm + (NdotL * dif) / kPi);
  } else {
    float LdotN = saturate(dot(N, L));
    float shadowMult = lightsCount * 1.0f;
    return (((shadowMult * LdotN) * lightIntensity) * dif) / kPi;
  }
}

BD f3 clampVec(f3 v, float hi) { return mk(clampUp(v.x, hi), clampUp(v.y, hi), clampUp(v.z, hi)); }  // MaterialUtils.hlsli:15-18

BD uint64_t toFixed(float c) { return (uint64_t)(c * 4294967296.0f); }

#undef BD
}  // namespace bdpt
