// bdpt_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT (see bdpt_oracle.h).
//
// Scalar restatement of the reference's BDPT pass.  Every function cites the
// reference lines it follows (paths relative to /root/reference/src):
//   BDPT/  = BidirectionalPathtracing/Data/
//   CP/    = CommonPasses/Data/CommonPasses/
//   F/     = Falcor/Framework/Source/
//
// Arithmetic contract shared with the HIP path so that results can be compared
// bit for bit: IEEE fp32, no FMA contraction (-ffp-contract=off), left-to-right
// evaluation as written, correctly rounded / and sqrt, and the few
// transcendental functions the shaders call (sin/cos of 2*pi*u, acos, atan,
// pow(x,5)) restated as fixed polynomial / product forms (det_* below) because
// HLSL leaves their precision to the driver.  PARITY UNPINNED for those and for
// traversal / intersection / texture filtering / fp16 rounding (header).
#include "bdpt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ----------------------------------------------------------------------------------------------
// float3 algebra in HLSL's componentwise sense
// ----------------------------------------------------------------------------------------------
struct f3 {
  float x, y, z;
};
inline f3 mk(float x, float y, float z) { return f3{x, y, z}; }
inline f3 mk(float s) { return f3{s, s, s}; }
inline f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
inline f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
inline f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
inline f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
inline f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(f3 a) { return sqrtf(dot(a, a)); }
inline f3 normalize(f3 a) {
  float inv = 1.0f / sqrtf(dot(a, a));
  return a * inv;
}
inline f3 absv(f3 a) { return f3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// HLSL max/min/clamp/saturate return the non-NaN operand; written as selects so
// both back ends agree on NaN: saturate(NaN) = 0.
inline float maxf(float a, float b) { return (b > a) ? b : ((a == a) ? a : b); }
inline float minf(float a, float b) { return (b < a) ? b : ((a == a) ? a : b); }
inline float saturate(float x) {
  float y = (x > 0.0f) ? x : 0.0f;
  return (y < 1.0f) ? y : 1.0f;
}
inline float clampUp(float x, float hi) {
  float y = (x > 0.0f) ? x : 0.0f;
  return (y < hi) ? y : hi;
}
inline bool isnan3(f3 a) { return (a.x != a.x) || (a.y != a.y) || (a.z != a.z); }
inline f3 lerp3(f3 a, f3 b, float s) { return a + (b - a) * s; }

constexpr float kPi = 3.14159265358979323846f;      // M_PI, F/Data/HostDeviceSharedMacros.h:170
constexpr float kInvPi = 0.318309886183790671538f;  // M_1_PI, BDPT/MaterialUtils.hlsli:2

// ----------------------------------------------------------------------------------------------
// Deterministic stand-ins for the driver's transcendental functions
// ----------------------------------------------------------------------------------------------
// sin/cos(2*pi*u), u in [0,1): quadrant split + Taylor on [-pi/4, pi/4].
inline void det_sincos2pi(float u, float& s, float& c) {
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
  if (qi == 0) {
    s = sa;
    c = ca;
  } else if (qi == 1) {
    s = ca;
    c = -sa;
  } else if (qi == 2) {
    s = -sa;
    c = -ca;
  } else {
    s = -ca;
    c = sa;
  }
}
// acos on [-1,1]: Abramowitz & Stegun 4.4.46 (|err| <= 2e-8).
inline float det_acos(float x) {
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
// atan: A&S 4.4.49 on [-1,1] + reciprocal reduction.
inline float det_atan(float z) {
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
inline float det_pow5(float x) {
  float x2 = x * x;
  return x2 * x2 * x;
}

// float -> half -> float with round-to-nearest-even (G-buffer RGBA16F channels,
// CP/../LightProbeGBufferPass.cpp:46-51).  PARITY UNPINNED: D3D12 UAV store rounding.
inline uint16_t f32_to_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to >= 65520 -> inf
  if (ax < 0x33000001u) return (uint16_t)sign;               // <= 2^-25 -> 0 (tie to even)
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t he;
  if (e < -14) {  // subnormal half
    shift = 13 + (-14 - e);
    he = 0;
  } else {
    shift = 13;
    he = (uint32_t)(e + 15);
  }
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t h;
  if (he == 0)
    h = q;  // q may carry into exponent 1: correct
  else
    h = ((he - 1) << 10) + q;  // q includes the implicit bit (0x400)
  return (uint16_t)(sign | h);
}
inline float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {
      float v = (float)m * 5.9604644775390625e-08f;  // m * 2^-24
      memcpy(&x, &v, 4);
      x |= sign;
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}
inline float halfRound(float f) { return f16_to_f32(f32_to_f16(f)); }

// ----------------------------------------------------------------------------------------------
// RNG — BDPT/BDPTUtils.hlsli:91-110
// ----------------------------------------------------------------------------------------------
inline uint32_t initRand(uint32_t val0, uint32_t val1, uint32_t backoff = 16) {
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for (uint32_t n = 0; n < backoff; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
inline float nextRand(uint32_t& s) {
  s = 1664525u * s + 1013904223u;
  return (float)(s & 0x00FFFFFFu) / (float)0x01000000;
}

// ----------------------------------------------------------------------------------------------
// Scene storage
// ----------------------------------------------------------------------------------------------
struct Tex {
  std::vector<uint8_t> px;
  uint32_t w = 0, h = 0, srgb = 0;
};
struct BNode {
  float lo[3], hi[3];
  int32_t left, right;  // inner: child node indices; leaf: left = -1-first, right = count
};

}  // namespace

struct oracle_scene {
  std::vector<f3> pos, nrm, bit, uv;
  std::vector<uint32_t> idx, triMat;
  std::vector<bdpt_material> mats;
  std::vector<Tex> tex;
  std::vector<bdpt_light> lights;
  float srgbLut[256];
  // triangle records as intersected: v0, e1 = v1-v0, e2 = v2-v0
  std::vector<f3> tv0, te1, te2;
  std::vector<uint8_t> triNonOpaque, triDoubleSided;
  std::vector<BNode> nodes;
  std::vector<uint32_t> order;  // BVH leaf order -> triangle id
  bool hasUv = false, hasBit = false;
  // environment of the BDPT pass (BDPT_PARAM_ENV_ON_MISS: a build definition, include/bdpt.h)
  std::vector<float> envMap;
  uint32_t envW = 0, envH = 0;
  float envColor[4] = {0, 0, 0, 0};
};

namespace {

// ----------------------------------------------------------------------------------------------
// Texture sampling: linear filter, wrap addressing, mip 0, sRGB decode before filtering.
// (Sampler state: SharedUtils/SceneLoaderWrapper.cpp:65-68; sampleTexture F/ShadingUtils/Shading.slang:88-94.)
// PARITY UNPINNED: sampler hardware arithmetic is not in the reference.
// ----------------------------------------------------------------------------------------------
struct f4 {
  float x, y, z, w;
};
inline f4 texel(const oracle_scene& s, const Tex& t, int ix, int iy) {
  const uint8_t* p = &t.px[((size_t)iy * t.w + (size_t)ix) * 4];
  f4 r;
  if (t.srgb) {
    r.x = s.srgbLut[p[0]];
    r.y = s.srgbLut[p[1]];
    r.z = s.srgbLut[p[2]];
  } else {
    r.x = (float)p[0] / 255.0f;
    r.y = (float)p[1] / 255.0f;
    r.z = (float)p[2] / 255.0f;
  }
  r.w = (float)p[3] / 255.0f;
  return r;
}
inline int wrapi(int i, int n) {
  int m = i % n;
  return (m < 0) ? m + n : m;
}
inline f4 lerp4(f4 a, f4 b, float s) {
  return f4{a.x + (b.x - a.x) * s, a.y + (b.y - a.y) * s, a.z + (b.z - a.z) * s, a.w + (b.w - a.w) * s};
}
f4 sampleBilinear(const oracle_scene& s, int texId, float u, float v) {
  const Tex& t = s.tex[(size_t)texId];
  float x = u * (float)t.w - 0.5f;
  float y = v * (float)t.h - 0.5f;
  float x0 = floorf(x), y0 = floorf(y);
  float fx = x - x0, fy = y - y0;
  int ix0 = wrapi((int)x0, (int)t.w), iy0 = wrapi((int)y0, (int)t.h);
  int ix1 = wrapi(ix0 + 1, (int)t.w), iy1 = wrapi(iy0 + 1, (int)t.h);
  f4 t00 = texel(s, t, ix0, iy0), t10 = texel(s, t, ix1, iy0);
  f4 t01 = texel(s, t, ix0, iy1), t11 = texel(s, t, ix1, iy1);
  return lerp4(lerp4(t00, t10, fx), lerp4(t01, t11, fx), fy);
}
// F/ShadingUtils/Shading.slang:88-94
inline f4 sampleTexture(const oracle_scene& s, int texId, float u, float v, f4 factor, uint32_t mode) {
  if (mode == BDPT_CHANNEL_UNUSED) return f4{0, 0, 0, 0};
  if (mode == BDPT_CHANNEL_CONST || texId < 0) return factor;
  return sampleBilinear(s, texId, u, v);
}

// ----------------------------------------------------------------------------------------------
// Vertex fetch — F/ShadingUtils/Raytracing.slang:53-106 (instance transform is identity:
// the ABI hands over world-space, instancing-flattened streams).
// ----------------------------------------------------------------------------------------------
struct VertexOut {
  f3 posW, normalW, bitangentW;
  float u, v;
};
VertexOut getVertexAttributes(const oracle_scene& s, uint32_t tri, float bu, float bv) {
  float b[3] = {1.0f - bu - bv, bu, bv};
  VertexOut o;
  o.posW = mk(0);
  o.normalW = mk(0);
  o.bitangentW = mk(0);
  o.u = 0;
  o.v = 0;
  for (int i = 0; i < 3; i++) {
    uint32_t vi = s.idx[(size_t)tri * 3 + i];
    if (s.hasUv) {
      o.u += s.uv[vi].x * b[i];
      o.v += s.uv[vi].y * b[i];
    }
    o.normalW = o.normalW + s.nrm[vi] * b[i];
    if (s.hasBit) o.bitangentW = o.bitangentW + s.bit[vi] * b[i];
    o.posW = o.posW + s.pos[vi] * b[i];
  }
  o.normalW = normalize(o.normalW);
  if (s.hasBit) o.bitangentW = normalize(o.bitangentW);
  return o;
}

// BDPT/BDPTUtils.hlsli:115-127, CP/lightProbeGBuffer.rt.hlsl:76-90
bool alphaTestFails(const oracle_scene& s, uint32_t tri, float bu, float bv) {
  const bdpt_material& m = s.mats[s.triMat[tri]];
  float u = 0, v = 0;
  uint32_t mode = BDPT_FLAG_DIFFUSE_TYPE(m.flags);
  if (mode == BDPT_CHANNEL_TEXTURE && m.texBaseColor >= 0 && s.hasUv) {
    float b[3] = {1.0f - bu - bv, bu, bv};
    for (int i = 0; i < 3; i++) {
      uint32_t vi = s.idx[(size_t)tri * 3 + i];
      u += s.uv[vi].x * b[i];
      v += s.uv[vi].y * b[i];
    }
  }
  f4 base = sampleTexture(s, m.texBaseColor, u, v, f4{m.baseColor[0], m.baseColor[1], m.baseColor[2], m.baseColor[3]}, mode);
  return base.w < m.alphaThreshold;
}

// ----------------------------------------------------------------------------------------------
// Intersection.  DXR conventions relied upon: hit iff TMin < t < TMax,
// direction need not be unit length, no culling unless the ray asks, front face = clockwise
// seen from the ray origin in D3D's convention <=> det > 0 below.  Closest hit ties resolve to
// the lowest triangle index so the answer is independent of traversal order.
// ----------------------------------------------------------------------------------------------
struct Hit {
  int32_t prim;
  float t, u, v;
};
struct Ray {
  f3 o, d;
  float tmin, tmax;
};
inline bool triTest(const oracle_scene& s, uint32_t tri, const Ray& r, bool cullBack, float& t, float& u, float& v) {
  f3 e1 = s.te1[tri], e2 = s.te2[tri];
  f3 pvec = cross(r.d, e2);
  float det = dot(e1, pvec);
  if (cullBack && !s.triDoubleSided[tri]) {
    if (!(det > 0.0f)) return false;
  } else {
    if (det == 0.0f) return false;
  }
  float inv = 1.0f / det;
  f3 tvec = r.o - s.tv0[tri];
  u = dot(tvec, pvec) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  f3 qvec = cross(tvec, e1);
  v = dot(r.d, qvec) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot(e2, qvec) * inv;
  return (t > r.tmin) && (t < r.tmax);
}
inline bool boxTest(const BNode& n, const Ray& r, f3 idir, float tbest) {
  float tx0 = (n.lo[0] - r.o.x) * idir.x, tx1 = (n.hi[0] - r.o.x) * idir.x;
  float ty0 = (n.lo[1] - r.o.y) * idir.y, ty1 = (n.hi[1] - r.o.y) * idir.y;
  float tz0 = (n.lo[2] - r.o.z) * idir.z, tz1 = (n.hi[2] - r.o.z) * idir.z;
  float tn = maxf(maxf(minf(tx0, tx1), minf(ty0, ty1)), maxf(minf(tz0, tz1), r.tmin));
  float tf = minf(minf(maxf(tx0, tx1), maxf(ty0, ty1)), minf(maxf(tz0, tz1), tbest));
  return tn <= tf;
}

// mode: 0 closest, 1 closest + cull back faces, 2 any hit (accept first)
Hit traceRay(const oracle_scene& s, const Ray& r, int mode, uint32_t flags, uint64_t* nodeVisits, uint64_t* triTests) {
  Hit best{-1, r.tmax, 0, 0};
  bool cull = (mode == 1);
  auto consider = [&](uint32_t tri) -> bool {
    float t, u, v;
    if (triTests) (*triTests)++;
    if (!triTest(s, tri, r, cull, t, u, v)) return false;
    if (s.triNonOpaque[tri] && alphaTestFails(s, tri, u, v)) return false;  // any-hit shader: IgnoreHit
    if (mode == 2) {
      best = Hit{0, t, u, v};
      return true;
    }
    if (t < best.t || (t == best.t && best.prim >= 0 && (int32_t)tri < best.prim)) best = Hit{(int32_t)tri, t, u, v};
    return false;
  };
  if ((flags & ORACLE_BRUTE_FORCE) || s.nodes.empty()) {
    for (uint32_t tri = 0; tri < (uint32_t)s.tv0.size(); tri++)
      if (consider(tri)) return best;
    return best;
  }
  f3 idir = mk(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
  int32_t stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const BNode& n = s.nodes[(size_t)stack[--sp]];
    if (nodeVisits) (*nodeVisits)++;
    // closest: only boxes that can still hold a hit at t <= best.t (ties matter -> <=)
    if (!boxTest(n, r, idir, best.t)) continue;
    if (n.left < 0) {
      uint32_t first = (uint32_t)(-1 - n.left), cnt = (uint32_t)n.right;
      for (uint32_t k = 0; k < cnt; k++)
        if (consider(s.order[first + k])) return best;
    } else {
      stack[sp++] = n.left;
      stack[sp++] = n.right;
    }
  }
  return best;
}

// Oracle's own BVH: top-down median split on the longest centroid axis, <= 4 triangles per leaf.
void buildBvh(oracle_scene& s) {
  uint32_t n = (uint32_t)s.tv0.size();
  s.order.resize(n);
  std::vector<f3> lo(n), hi(n), cen(n);
  f3 slo = mk(1e30f), shi = mk(-1e30f);
  for (uint32_t i = 0; i < n; i++) {
    s.order[i] = i;
    f3 a = s.tv0[i], b = a + s.te1[i], c = a + s.te2[i];
    lo[i] = mk(minf(a.x, minf(b.x, c.x)), minf(a.y, minf(b.y, c.y)), minf(a.z, minf(b.z, c.z)));
    hi[i] = mk(maxf(a.x, maxf(b.x, c.x)), maxf(a.y, maxf(b.y, c.y)), maxf(a.z, maxf(b.z, c.z)));
    cen[i] = (lo[i] + hi[i]) * 0.5f;
    slo = mk(minf(slo.x, lo[i].x), minf(slo.y, lo[i].y), minf(slo.z, lo[i].z));
    shi = mk(maxf(shi.x, hi[i].x), maxf(shi.y, hi[i].y), maxf(shi.z, hi[i].z));
  }
  float pad = 2e-5f * length(shi - slo) + 1e-30f;
  s.nodes.clear();
  if (n == 0) return;
  struct Item {
    uint32_t node, first, count;
  };
  std::vector<Item> todo;
  s.nodes.push_back(BNode{});
  todo.push_back(Item{0, 0, n});
  while (!todo.empty()) {
    Item it = todo.back();
    todo.pop_back();
    f3 blo = mk(1e30f), bhi = mk(-1e30f), clo = mk(1e30f), chi = mk(-1e30f);
    for (uint32_t k = 0; k < it.count; k++) {
      uint32_t t = s.order[it.first + k];
      blo = mk(minf(blo.x, lo[t].x), minf(blo.y, lo[t].y), minf(blo.z, lo[t].z));
      bhi = mk(maxf(bhi.x, hi[t].x), maxf(bhi.y, hi[t].y), maxf(bhi.z, hi[t].z));
      clo = mk(minf(clo.x, cen[t].x), minf(clo.y, cen[t].y), minf(clo.z, cen[t].z));
      chi = mk(maxf(chi.x, cen[t].x), maxf(chi.y, cen[t].y), maxf(chi.z, cen[t].z));
    }
    BNode nd;
    nd.lo[0] = blo.x - pad;
    nd.lo[1] = blo.y - pad;
    nd.lo[2] = blo.z - pad;
    nd.hi[0] = bhi.x + pad;
    nd.hi[1] = bhi.y + pad;
    nd.hi[2] = bhi.z + pad;
    if (it.count <= 4) {
      nd.left = -1 - (int32_t)it.first;
      nd.right = (int32_t)it.count;
      s.nodes[it.node] = nd;
      continue;
    }
    f3 ext = chi - clo;
    int axis = (ext.x >= ext.y && ext.x >= ext.z) ? 0 : (ext.y >= ext.z ? 1 : 2);
    uint32_t mid = it.count / 2;
    auto key = [&](uint32_t t) { return axis == 0 ? cen[t].x : (axis == 1 ? cen[t].y : cen[t].z); };
    std::nth_element(s.order.begin() + it.first, s.order.begin() + it.first + mid,
                     s.order.begin() + it.first + it.count,
                     [&](uint32_t a, uint32_t b) { return key(a) < key(b) || (key(a) == key(b) && a < b); });
    uint32_t l = (uint32_t)s.nodes.size();
    s.nodes.push_back(BNode{});
    s.nodes.push_back(BNode{});
    nd.left = (int32_t)l;
    nd.right = (int32_t)l + 1;
    s.nodes[it.node] = nd;
    todo.push_back(Item{l, it.first, mid});
    todo.push_back(Item{l + 1, it.first + mid, it.count - mid});
  }
}

// ----------------------------------------------------------------------------------------------
// Shading data — BDPT/BDPTUtils.hlsli:2-61 (secondary hits: no normal map) and
// F/ShadingUtils/Shading.slang:189-259 (primary hit: with normal map, :135-157)
// ----------------------------------------------------------------------------------------------
struct ShadingData {
  f3 posW, V, N, T, B;
  float u, v;
  float NdotV;
  f3 diffuse;
  float opacity;
  f3 specular;
  float linearRoughness, roughness;
  f3 emissive;
  float IoR;
  bool doubleSided;
};
ShadingData prepareShadingData(const oracle_scene& s, uint32_t tri, float bu, float bv, f3 camPosW, bool withNormalMap) {
  VertexOut vo = getVertexAttributes(s, tri, bu, bv);
  const bdpt_material& m = s.mats[s.triMat[tri]];
  ShadingData sd;
  f4 base = sampleTexture(s, m.texBaseColor, vo.u, vo.v, f4{m.baseColor[0], m.baseColor[1], m.baseColor[2], m.baseColor[3]},
                          BDPT_FLAG_DIFFUSE_TYPE(m.flags));
  sd.opacity = m.baseColor[3];
  sd.posW = vo.posW;
  sd.u = vo.u;
  sd.v = vo.v;
  sd.V = normalize(camPosW - vo.posW);
  sd.N = normalize(vo.normalW);
  sd.B = mk(0);
  sd.T = mk(0);
  f4 spec = sampleTexture(s, m.texSpecular, vo.u, vo.v, f4{m.specular[0], m.specular[1], m.specular[2], m.specular[3]},
                          BDPT_FLAG_SPECULAR_TYPE(m.flags));
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
  f4 em = sampleTexture(s, m.texEmissive, vo.u, vo.v, f4{m.emissive[0], m.emissive[1], m.emissive[2], 1.0f},
                        BDPT_FLAG_EMISSIVE_TYPE(m.flags));
  sd.emissive = mk(em.x, em.y, em.z);
  sd.IoR = m.IoR;
  sd.doubleSided = BDPT_FLAG_DOUBLE_SIDED(m.flags) != 0;
  uint32_t mapType = BDPT_FLAG_NORMAL_MAP_TYPE(m.flags);
  if (withNormalMap && mapType != BDPT_NORMAL_MAP_UNUSED && m.texNormal >= 0 && s.hasBit) {
    // applyNormalMap, F/ShadingUtils/Shading.slang:135-157
    sd.B = normalize(vo.bitangentW - sd.N * dot(vo.bitangentW, sd.N));
    sd.T = normalize(cross(sd.B, sd.N));
    f4 mp = sampleBilinear(s, m.texNormal, vo.u, vo.v);
    f3 mapN;
    if (mapType == BDPT_NORMAL_MAP_RGB) {
      mapN = normalize(mk(mp.x, mp.y, mp.z) * 2.0f - mk(1.0f));
    } else {
      float nx = mp.x * 2.0f - 1.0f, ny = mp.y * 2.0f - 1.0f;
      float nz = saturate(mp.x * mp.x + mp.y * mp.y);
      nz = sqrtf(1.0f - nz);
      mapN = normalize(mk(nx, ny, nz));
    }
    sd.N = sd.T * mapN.x + sd.B * mapN.y + sd.N * mapN.z;
  }
  sd.NdotV = dot(sd.N, sd.V);
  if (sd.NdotV <= 0.0f && sd.doubleSided) {
    sd.N = -sd.N;
    sd.NdotV = -sd.NdotV;
  }
  return sd;
}

// ----------------------------------------------------------------------------------------------
// BRDF utilities — BDPT/BRDFUtils.hlsli, BDPT/MaterialUtils.hlsli
// ----------------------------------------------------------------------------------------------
inline float luminance(f3 rgb) { return dot(rgb, mk(0.2126f, 0.7152f, 0.0722f)); }  // F/Data/HostDeviceSharedCode.h:256-259
// MaterialUtils.hlsli:22-27
inline float probabilityToSampleDiffuse(f3 dif, f3 spec) {
  float lumDiffuse = maxf(0.01f, luminance(dif));
  float lumSpecular = maxf(0.01f, luminance(spec));
  return lumDiffuse / (lumDiffuse + lumSpecular);
}
// MaterialUtils.hlsli:31-38
inline f3 getPerpendicularVector(f3 u) {
  f3 a = absv(u);
  uint32_t xm = ((a.x - a.y) < 0 && (a.x - a.z) < 0) ? 1 : 0;
  uint32_t ym = (a.y - a.z) < 0 ? (1 ^ xm) : 0;
  uint32_t zm = 1 ^ (xm | ym);
  return cross(u, mk((float)xm, (float)ym, (float)zm));
}
// MaterialUtils.hlsli:41-54.  float2(nextRand, nextRand): .x is the first draw.
inline f3 getCosHemisphereSample(uint32_t& seed, f3 hitNorm) {
  float r0 = nextRand(seed);
  float r1 = nextRand(seed);
  f3 bitangent = getPerpendicularVector(hitNorm);
  f3 tangent = cross(bitangent, hitNorm);
  float r = sqrtf(r0);
  float sn, cs;
  det_sincos2pi(r1, sn, cs);
  return tangent * (r * cs) + bitangent * (r * sn) + hitNorm * sqrtf(maxf(0.0f, 1.0f - r0));
}
// MaterialUtils.hlsli:56-63
inline f3 sampleUnitSphere(uint32_t& seed) {
  f3 p = mk(2.0f, 2.0f, 2.0f);
  while (length(p) > 1.0f) {
    float a = nextRand(seed) * 2.0f - 1.0f;
    float b = nextRand(seed) * 2.0f - 1.0f;
    float c = nextRand(seed) * 2.0f - 1.0f;
    p = mk(a, b, c);
  }
  return p;
}
// BRDFUtils.hlsli:5-10
inline float ggxNormalDistribution(float NdotH, float roughness) {
  float a2 = roughness * roughness;
  float d = ((NdotH * a2 - NdotH) * NdotH + 1);
  return a2 / maxf(0.001f, (d * d * kPi));
}
// BRDFUtils.hlsli:15-30
inline float ggxSchlickMaskingTerm(float NdotL, float NdotV, float roughness) {
  float k = roughness * roughness / 2;
  float g_v = NdotV / (NdotV * (1 - k) + k);
  float g_l = NdotL / (NdotL * (1 - k) + k);
  return g_v * g_l;
}
// BRDFUtils.hlsli:35-38
inline f3 schlickFresnel(f3 f0, float u) { return f0 + (mk(1.0f) - f0) * det_pow5(1.0f - u); }
// BRDFUtils.hlsli:44-61
inline f3 getGGXMicrofacet(uint32_t& seed, float roughness, f3 hitNorm) {
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
// BRDFUtils.hlsli:63-73
inline f3 ggxLighting(f3 H, f3 L, f3 N, float NdotL, float NdotV, float rough, f3 spec, float& ggxProb) {
  float NdotH = saturate(dot(N, H));
  float LdotH = saturate(dot(L, H));
  float D = ggxNormalDistribution(NdotH, rough);
  float G = ggxSchlickMaskingTerm(NdotL, NdotV, rough);
  f3 F = schlickFresnel(spec, LdotH);
  ggxProb = D * NdotH / (4 * LdotH);
  return (D * G) * F / (4 * NdotL * NdotV);
}

struct Globals {
  const oracle_scene* s;
  bdpt_camera cam;
  bdpt_params p;
  uint32_t flags;
  uint32_t W, H;
  int lightsCount;
};

// MaterialUtils.hlsli:209-252
f3 sampleGGXBRDF(uint32_t seed, f3 N, f3 noNormalN, f3 V, f3 dif, f3 spec, float rough, f3& L, float& pdf, bool& isSpecular,
                 bool fromLobe) {
  float probDiffuse = probabilityToSampleDiffuse(dif, spec);
  bool chooseDiffuse = (nextRand(seed) < probDiffuse);
  float NdotV = saturate(dot(N, V));
  // The shader declares `out bool isSpecular` and never writes it on this branch of sampleBRDF
  // (MaterialUtils.hlsli:209-252): undefined in HLSL.  Build definition: false (a DXIL undef
  // reads as 0), or the lobe pick when BDPT_PARAM_SPECULAR_FROM_LOBE is set.
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
}
// MaterialUtils.hlsli:321-329
f3 sampleLambertianBRDF(uint32_t& seed, f3 norm, f3 dif, f3& L, float& pdf) {
  L = getCosHemisphereSample(seed, norm);
  pdf = saturate(dot(norm, L)) * kInvPi;
  return dif;
}
// MaterialUtils.hlsli:130-141 — randSeed is taken BY VALUE (quirk 1)
f3 sampleBRDF(const Globals& g, uint32_t seed, f3 N, f3 noNormalN, f3 V, f3 dif, f3 spec, float rough, f3& L, float& pdf,
              bool& isSpecular) {
  if (g.p.matIndex == 0)
    return sampleGGXBRDF(seed, N, noNormalN, V, dif, spec, rough, L, pdf, isSpecular, (g.p.flags & BDPT_PARAM_SPECULAR_FROM_LOBE) != 0);
  isSpecular = false;
  return sampleLambertianBRDF(seed, N, dif, L, pdf);
}
// MaterialUtils.hlsli:186-207
f3 evalGGXBRDF(f3 V, f3 L, f3 N, f3 noNormalN, f3 dif, f3 spec, float rough, bool isSpecular) {
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
// MaterialUtils.hlsli:105-115, 309-314 (Lambertian returns dif with no 1/pi and no cosine test, sic)
f3 evalBRDF(const Globals& g, f3 V, f3 L, f3 N, f3 noNormalN, f3 dif, f3 spec, float rough, bool isSpecular) {
  if (g.p.matIndex == 0) return evalGGXBRDF(V, L, N, noNormalN, dif, spec, rough, isSpecular);
  return dif;
}

// F/ShadingUtils/Lights.slang:54-102 + BDPT/MaterialUtils.hlsli:67-85
void getLightData(const Globals& g, int index, f3 hitPos, f3& toLight, f3& lightIntensity, float& distToLight) {
  const bdpt_light& l = g.s->lights[(size_t)index];
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

struct Tally {
  uint64_t rays[6] = {0, 0, 0, 0, 0, 0};  // primary, eye, light, nee, splat, connect
  uint64_t nodeC = 0, triC = 0, nodeS = 0, triS = 0, valid = 0, splats = 0;
};

// BDPT/standardShadowRay.hlsli:7-49
bool shadowRayVisibility(const Globals& g, Tally& tl, int stage, f3 origin, f3 direction, float minT, float maxT) {
  Ray r{origin, direction, minT, maxT};
  tl.rays[stage]++;
  Hit h = traceRay(*g.s, r, 2, g.flags, &tl.nodeS, &tl.triS);
  return h.prim < 0;
}

// MaterialUtils.hlsli:149-184
f3 ggxDirect(const Globals& g, Tally& tl, uint32_t& seed, f3 hit, f3 N, f3 V, f3 dif, f3 spec, float rough) {
  int lightToSample = (int)(nextRand(seed) * (float)g.lightsCount);
  if (lightToSample > g.lightsCount - 1) lightToSample = g.lightsCount - 1;
  float distToLight;
  f3 lightIntensity, L;
  getLightData(g, lightToSample, hit, L, lightIntensity, distToLight);
  float NdotL = saturate(dot(N, L));
  bool vis = shadowRayVisibility(g, tl, 3, hit, L, g.p.minT, distToLight);
  float shadowMult = vis ? (float)g.lightsCount : 0.f;
  f3 H = normalize(V + L);
  float NdotH = saturate(dot(N, H));
  float LdotH = saturate(dot(L, H));
  float NdotV = saturate(dot(N, V));
  float D = ggxNormalDistribution(NdotH, rough);
  float G = ggxSchlickMaskingTerm(NdotL, NdotV, rough);
  f3 F = schlickFresnel(spec, LdotH);
  f3 ggxTerm = (D * G) * F / (4 * NdotV);
  return (shadowMult * lightIntensity) * (ggxTerm + (NdotL * dif) / kPi);
}
// MaterialUtils.hlsli:288-307
f3 lambertianDirect(const Globals& g, Tally& tl, uint32_t& seed, f3 hit, f3 norm, f3 difColor) {
  int lightToSample = (int)(nextRand(seed) * (float)g.lightsCount);
  if (lightToSample > g.lightsCount - 1) lightToSample = g.lightsCount - 1;
  float distToLight;
  f3 lightIntensity, toLight;
  getLightData(g, lightToSample, hit, toLight, lightIntensity, distToLight);
  float LdotN = saturate(dot(norm, toLight));
  bool vis = shadowRayVisibility(g, tl, 3, hit, toLight, g.p.minT, distToLight);
  float shadowMult = (float)g.lightsCount * (vis ? 1.0f : 0.0f);
  return (((shadowMult * LdotN) * lightIntensity) * difColor) / kPi;
}

// BDPT/RayPathData.hlsli:1-45
struct PathVertex {
  f3 color, posW, N, V, dif, spec;
  float rough;
  bool isSpecular;
  float pdfForward;
};
inline PathVertex vinit() {
  PathVertex v;
  v.color = v.posW = v.N = v.V = v.dif = v.spec = mk(0);
  v.rough = 0;
  v.isSpecular = false;
  v.pdfForward = 0;
  return v;
}
// BDPT/RayPathData.hlsli:48-86
struct RayPayload {
  f3 color;
  uint32_t rndSeed;
  f3 posW, N, V, dif, spec;
  float rough;
  bool isSpecular;
  float pdfForward;
  f3 rayOrigin, rayDir;
  bool terminated;
  f3 extra;       // not in the reference: radiance the ray picked up where it ended (environment on a miss, emissive
  bool hasExtra;  // on a hit), for BDPT_PARAM_ENV_ON_MISS / BDPT_PARAM_EMISSIVE_HITS
};
inline RayPayload initPayload(f3 rayOrigin, f3 rayDir, f3 color, uint32_t seed) {
  RayPayload p;
  p.rayOrigin = rayOrigin;
  p.rayDir = rayDir;
  p.rndSeed = seed;
  p.color = color;
  p.posW = rayOrigin;
  p.N = p.V = p.dif = p.spec = mk(0);
  p.rough = 0;
  p.isSpecular = false;
  p.pdfForward = 0;
  p.terminated = false;
  p.extra = mk(0);
  p.hasExtra = false;
  return p;
}
inline PathVertex fromPayload(const RayPayload& p) {
  PathVertex v;
  v.color = p.color;
  v.posW = p.posW;
  v.N = p.N;
  v.V = p.V;
  v.dif = p.dif;
  v.spec = p.spec;
  v.rough = p.rough;
  v.isSpecular = p.isSpecular;
  v.pdfForward = p.pdfForward;
  return v;
}

inline float atan2_WAR(float y, float x);
// The environment a ray that leaves the scene sees: the lat-long lookup of the G-buffer pass's miss shader
// (CP/lightProbeGBuffer.rt.hlsl:63-74) on the BDPT pass's own environment (BDPT_PARAM_ENV_ON_MISS, a build definition).
inline f3 environmentOf(const oracle_scene& s, f3 dir) {
  if (s.envMap.empty()) return mk(s.envColor[0], s.envColor[1], s.envColor[2]);
  f3 p = normalize(dir);
  float u = (1.f + atan2_WAR(p.x, -p.z) * kInvPi) * 0.5f;
  float v = det_acos(p.y) * kInvPi;
  uint32_t ex = (uint32_t)(u * (float)s.envW), ey = (uint32_t)(v * (float)s.envH);
  if (ex < s.envW && ey < s.envH) return ld3(s.envMap.data() + ((size_t)ey * s.envW + ex) * 4);
  return mk(0);
}

// BDPT/globalIlluminationRay.hlsli:1-45 (shootRay + RayMiss + RayAnyHit + RayClosestHit + handleIndirectRayHit)
void shootRay(const Globals& g, Tally& tl, int stage, RayPayload& pl) {
  Ray r{pl.rayOrigin, pl.rayDir, g.p.minT, 1.0e38f};
  tl.rays[stage]++;
  pl.hasExtra = false;
  Hit h = traceRay(*g.s, r, 0, g.flags, &tl.nodeC, &tl.triC);
  if (h.prim < 0) {  // RayMiss
    if (stage == 1 && (g.p.flags & BDPT_PARAM_ENV_ON_MISS)) {  // not in the reference
      pl.extra = environmentOf(*g.s, pl.rayDir);
      pl.hasExtra = true;
    }
    pl.color = mk(0);
    pl.terminated = true;
    return;
  }
  // RayClosestHit: getHitShadingData(attribs, WorldRayOrigin()) — V points at the ray origin
  ShadingData sd = prepareShadingData(*g.s, (uint32_t)h.prim, h.u, h.v, pl.rayOrigin, false);
  if (stage == 1 && (g.p.flags & BDPT_PARAM_EMISSIVE_HITS) && (sd.emissive.x > 0.0f || sd.emissive.y > 0.0f || sd.emissive.z > 0.0f)) {
    pl.extra = sd.emissive;  // not in the reference
    pl.hasExtra = true;
  }
  f3 L;
  float pdf;
  bool isSpecular;
  f3 color = sampleBRDF(g, pl.rndSeed, sd.N, sd.N, sd.V, sd.diffuse, sd.specular, sd.roughness, L, pdf, isSpecular);
  // updateRayData, BDPT/RayPathData.hlsli:88-109
  pl.color = pl.color * color;
  pl.rayOrigin = sd.posW;
  pl.rayDir = L;
  pl.posW = sd.posW;
  pl.N = sd.N;
  pl.V = sd.V;
  pl.dif = sd.diffuse;
  pl.spec = sd.specular;
  pl.rough = sd.roughness;
  pl.isSpecular = isSpecular;
  pl.pdfForward = pdf;
}

inline f3 clampVec(const Globals& g, f3 v) {  // MaterialUtils.hlsli:15-18
  return mk(clampUp(v.x, g.p.clampUpper), clampUp(v.y, g.p.clampUpper), clampUp(v.z, g.p.clampUpper));
}
// BDPT/BDPTUtils.hlsli:172-184
inline float evalGWithoutV(const PathVertex& a, const PathVertex& b) {
  f3 vecAB = b.posW - a.posW;
  float invLengthAB = 1.0f / length(vecAB);
  f3 dirAB = vecAB * invLengthAB;
  float cosA = fabsf(dot(a.N, dirAB));
  float cosB = fabsf(dot(b.N, dirAB));
  return cosA * cosB * invLengthAB * invLengthAB;
}
// BDPT/BDPTUtils.hlsli:186-224 (aL indexes the LIGHT path with cameraIndex-1, sic :198)
f3 getUnweightedContribution(const Globals& g, const PathVertex* cameraPath, const PathVertex* lightPath, uint32_t cameraIndex,
                             uint32_t lightIndex, float G) {
  if (cameraIndex == 0 || lightIndex == 0) return mk(0);
  const PathVertex& cameraEndV = cameraPath[cameraIndex];
  const PathVertex& lightEndV = lightPath[lightIndex];
  f3 aE = cameraPath[cameraIndex - 1].color;
  f3 aL = lightPath[cameraIndex - 1].color;
  f3 connectDir = normalize(cameraEndV.posW - lightEndV.posW);
  f3 wi = connectDir;
  f3 wo = normalize(lightPath[lightIndex - 1].posW - lightEndV.posW);
  f3 fsL = evalBRDF(g, wi, wo, lightEndV.N, lightEndV.N, lightEndV.dif, lightEndV.spec, lightEndV.rough, lightEndV.isSpecular);
  if (fsL.x == 0 && fsL.y == 0 && fsL.z == 0) return fsL;
  wi = -connectDir;
  wo = normalize(cameraPath[cameraIndex - 1].posW - cameraEndV.posW);
  f3 fsE = evalBRDF(g, wi, wo, cameraEndV.N, cameraEndV.N, cameraEndV.dif, cameraEndV.spec, cameraEndV.rough, cameraEndV.isSpecular);
  if (fsE.x == 0 && fsE.y == 0 && fsE.z == 0) return fsE;
  f3 cst = (fsL * G) * fsE;
  return (aL * cst) * aE;
}

// BDPT/BDPTUtils.hlsli:226-251 (power = true) and :253-278 (power = false).  The reference defines both
// and never calls them; the build wires them in behind BDPT_PARAM_MIS_POWER / BDPT_PARAM_MIS_LINEAR.
// `if (i == cameraIndex, j == lightIndex)` is a comma expression: it tests j == lightIndex, which,
// i + j being fixed, selects the intended term.
float getWeight(const PathVertex* cameraPath, const PathVertex* lightPath, uint32_t cameraIndex, uint32_t lightIndex, bool power) {
  uint32_t totalLength = cameraIndex + lightIndex;
  float totalPdf = 0;
  float currentPdf = 1;
  for (uint32_t i = 0; i <= totalLength; i++) {
    uint32_t j = totalLength - i;
    float pE = cameraPath[0].pdfForward;
    for (uint32_t x = 1; x <= i; x++) pE *= cameraPath[x].pdfForward * evalGWithoutV(cameraPath[x - 1], cameraPath[x]);
    float pL = lightPath[0].pdfForward;
    for (uint32_t x = 1; x <= j; x++) pL *= lightPath[x].pdfForward * evalGWithoutV(lightPath[x - 1], lightPath[x]);
    const float term = power ? (pE * pE * pL * pL) : (pE * pL);
    totalPdf += term;
    if (j == lightIndex) currentPdf = term;
  }
  return currentPdf / totalPdf;
}
// uniform 1/k of the reference, or the MIS weight when switched on
inline f3 applyStrategyWeight(const Globals& g, const PathVertex* cameraPath, const PathVertex* lightPath, f3 v, uint32_t k,
                              uint32_t cameraIndex, uint32_t lightIndex) {
  if (g.p.flags & (BDPT_PARAM_MIS_POWER | BDPT_PARAM_MIS_LINEAR))
    return v * getWeight(cameraPath, lightPath, cameraIndex, lightIndex, (g.p.flags & BDPT_PARAM_MIS_POWER) != 0);
  return v / (float)k;
}

// BDPT/BDPTUtils.hlsli:129-138; signed index + range check is the build's definition (quirk 8)
inline bool getLaunchIndexFromDirection(const Globals& g, f3 dir, int& ix, int& iy) {
  f3 U = ld3(g.cam.cameraU), V = ld3(g.cam.cameraV), Wv = ld3(g.cam.cameraW);
  float d1 = dot(dir, U) / dot(U, U);
  float d2 = dot(dir, V) / dot(V, V);
  float d3 = dot(dir, Wv) / dot(Wv, Wv);
  float nx = d1 / d3, ny = -d2 / d3;
  float px = nx * 0.5f + 0.5f, py = ny * 0.5f + 0.5f;
  float fx = rintf(px * (float)g.W - g.p.pixelJitter[0]);
  float fy = rintf(py * (float)g.H - g.p.pixelJitter[1]);
  if (!(fx >= 0.0f && fx < (float)g.W && fy >= 0.0f && fy < (float)g.H)) return false;
  ix = (int)fx;
  iy = (int)fy;
  return true;
}

inline uint64_t toFixed(float c) {  // c in [0, clampUpper<=1]; 2^-32 units
  return (uint64_t)(c * 4294967296.0f);
}

// ----------------------------------------------------------------------------------------------
// SimpleDiffuseGIRayGen — BDPT/BDPTMain.rt.hlsl:42-234 for one pixel
// ----------------------------------------------------------------------------------------------
void bdptPixel(const Globals& g, Tally& tl, oracle_frame* f, uint32_t x, uint32_t y) {
  const uint32_t W = g.W;
  const size_t pix = (size_t)y * W + x;
  const uint32_t D = g.p.maxDepth;
  const float* gp = f->worldPosition + pix * 4;
  const float* gn = f->worldNormal + pix * 4;
  const float* gd = f->materialDiffuse + pix * 4;
  const float* gs = f->materialSpecRough + pix * 4;
  const float* ge = f->emissive + pix * 4;
  float* out = f->out + pix * 4;
  out[0] = out[1] = out[2] = out[3] = 0.0f;  // getClearedTexture, BDPTPass.cpp:73

  if (!(gp[3] != 0.0f)) {  // :59-66 background
    out[0] = gd[0];
    out[1] = gd[1];
    out[2] = gd[2];
    out[3] = 1.0f;
    return;
  }
  tl.valid++;
  f3 camPos = ld3(g.cam.posW);
  f3 worldPos = ld3(gp), worldNorm = ld3(gn), difMatl = ld3(gd), specMatl = ld3(gs);
  float roughness = gs[3] * gs[3];
  f3 V = normalize(camPos - worldPos);
  uint32_t randSeed = initRand(x + y * W, g.p.frameCount, 16);

  PathVertex cameraPath[BDPT_MAX_DEPTH + 2], lightPath[BDPT_MAX_DEPTH + 2];  // [9] in the reference, :76-82
  bool takeContribution[BDPT_MAX_DEPTH + 2];
  for (uint32_t i = 0; i < BDPT_MAX_DEPTH + 2; i++) {
    cameraPath[i] = vinit();
    lightPath[i] = vinit();
    takeContribution[i] = true;
  }

  cameraPath[0].posW = camPos;
  cameraPath[0].N = normalize(ld3(g.cam.cameraW));
  cameraPath[0].color = mk(1.0f);
  cameraPath[0].pdfForward = 1.0f;

  f3 outDir;
  float pdf;
  bool isHitSpecular;
  f3 hitThroughput = sampleBRDF(g, randSeed, worldNorm, worldNorm, V, difMatl, specMatl, roughness, outDir, pdf, isHitSpecular);
  {
    PathVertex v;
    v.color = hitThroughput;
    v.posW = worldPos;
    v.N = worldNorm;
    v.V = V;
    v.dif = difMatl;
    v.spec = specMatl;
    v.rough = roughness;
    v.isSpecular = isHitSpecular;
    v.pdfForward = pdf;
    cameraPath[1] = v;
  }
  RayPayload payload = initPayload(worldPos, outDir, hitThroughput, randSeed);
  f3 walkTerms[BDPT_MAX_DEPTH + 2];  // BDPT_PARAM_ENV_ON_MISS / _EMISSIVE_HITS: what the ray that left eye vertex `depth` found
  uint32_t numWalkTerms = 0;
  for (uint32_t depth = 1; depth < D && !payload.terminated; depth++) {  // :106-112
    shootRay(g, tl, 1, payload);
    if (payload.hasExtra) {  // path-tracing strategy of depth + 1 edges, weighted like the NEE terms (uniform 1/edges)
      f3 term = clampVec(g, (cameraPath[depth].color * payload.extra) / (float)(depth + 1));
      walkTerms[numWalkTerms++] = isnan3(term) ? mk(0) : term;
    }
    cameraPath[depth + 1] = fromPayload(payload);
  }
  randSeed = payload.rndSeed;  // :115 (never advanced: quirk 1)

  // sampleLight, BDPT/BDPTUtils.hlsli:140-152
  f3 lightOrigin, lightDir, lightIntensity;
  {
    int index = (int)(nextRand(randSeed) * (float)g.lightsCount);
    if (index > g.lightsCount - 1) index = g.lightsCount - 1;
    const bdpt_light& l = g.s->lights[(size_t)index];
    lightOrigin = ld3(l.posW);
    lightIntensity = ld3(l.intensity);
    if (l.type == BDPT_LIGHT_DIRECTIONAL)
      lightDir = ld3(l.dirW);
    else
      lightDir = sampleUnitSphere(randSeed);
    lightDir = getCosHemisphereSample(randSeed, lightDir);
  }
  lightPath[0].posW = lightOrigin;
  lightPath[0].color = lightIntensity;
  lightPath[0].pdfForward = 1.0f / (float)g.lightsCount;
  RayPayload lightPayload = initPayload(lightOrigin, lightDir, lightIntensity, randSeed);
  for (uint32_t depth = 0; depth < D && !lightPayload.terminated; depth++) {  // :138-145
    shootRay(g, tl, 2, lightPayload);
    lightPath[depth + 1] = fromPayload(lightPayload);
    takeContribution[depth + 1] = !lightPayload.terminated;
  }
  randSeed = lightPayload.rndSeed;  // :148

  if (ge[0] > 0.0f || ge[1] > 0.0f || ge[2] > 0.0f) {  // :155-158
    out[0] += ge[0];
    out[1] += ge[1];
    out[2] += ge[2];
    out[3] += ge[3];
  }
  for (uint32_t i = 0; i < numWalkTerms; i++) {  // after the pixel's own emissive, before the NEE terms, bounce order
    out[0] = out[0] + walkTerms[i].x;
    out[1] = out[1] + walkTerms[i].y;
    out[2] = out[2] + walkTerms[i].z;
    out[3] = out[3] + 1.0f;
  }
  const bool doNee = !(g.p.flags & BDPT_PARAM_NO_NEE);
  const bool doSplat = !(g.p.flags & BDPT_PARAM_NO_SPLAT);
  const bool doConnect = !(g.p.flags & BDPT_PARAM_NO_CONNECT);

  for (uint32_t i = 0; i < D; i++) {  // :161-167
    const PathVertex& v = cameraPath[i + 1];
    f3 direct = (g.p.matIndex == 0) ? ggxDirect(g, tl, randSeed, v.posW, v.N, v.V, v.dif, v.spec, v.rough)
                                    : lambertianDirect(g, tl, randSeed, v.posW, v.N, v.dif);
    f3 shade = cameraPath[i].color * direct;
    shade = clampVec(g, applyStrategyWeight(g, cameraPath, lightPath, shade, i + 2, i + 1, 0));
    bool colorsNan = isnan3(shade);
    if (doNee) {
      out[0] = out[0] + (colorsNan ? 0.0f : shade.x);
      out[1] = out[1] + (colorsNan ? 0.0f : shade.y);
      out[2] = out[2] + (colorsNan ? 0.0f : shade.z);
      out[3] = out[3] + 1.0f;
    }
  }

  // light tracing, :171-208.  Build definition (quirk 6): fixed-point sums in a separate buffer.
  for (uint32_t i = 0; doSplat && i < D && takeContribution[i + 1]; i++) {
    f3 lastHitPos = lightPath[i + 1].posW;
    f3 lastHitN = lightPath[i + 1].N;
    f3 cameraN = normalize(ld3(g.cam.cameraW));
    f3 dirToCamera = normalize(camPos - lastHitPos);
    float disToCamera = length(camPos - lastHitPos);
    if (dot(cameraN, dirToCamera) < 0 && takeContribution[i + 1]) {
      bool vis = shadowRayVisibility(g, tl, 4, lastHitPos, dirToCamera, g.p.minT, disToCamera);
      if (vis) {
        int ix, iy;
        bool inside = getLaunchIndexFromDirection(g, dirToCamera, ix, iy);
        float theta1 = saturate(fabsf(dot(dirToCamera, cameraN)));
        float theta2 = saturate(fabsf(dot(dirToCamera, lastHitN)));
        float invDisToCamera = 1.0f / disToCamera;
        float G = theta1 * theta2 * invDisToCamera * invDisToCamera;
        const PathVertex& lv = lightPath[i + 1];
        // connectToCamera, MaterialUtils.hlsli:10-13
        f3 fr = evalBRDF(g, lv.V, normalize(camPos - lv.posW), lv.N, lv.N, lv.dif, lv.spec, lv.rough, lv.isSpecular);
        f3 shade = (lightPath[i].color * fr) * G;
        shade = clampVec(g, applyStrategyWeight(g, cameraPath, lightPath, shade, i + 2, 0, i + 1));
        bool colorsNan = isnan3(shade);
        if (colorsNan) shade = mk(0);
        if (inside) {
          uint64_t* sp = f->splat + ((size_t)iy * W + (size_t)ix) * 4;
          __atomic_fetch_add(&sp[0], toFixed(shade.x), __ATOMIC_RELAXED);
          __atomic_fetch_add(&sp[1], toFixed(shade.y), __ATOMIC_RELAXED);
          __atomic_fetch_add(&sp[2], toFixed(shade.z), __ATOMIC_RELAXED);
          __atomic_fetch_add(&sp[3], (uint64_t)1, __ATOMIC_RELAXED);
          tl.splats++;
        }
      }
    }
  }

  // vertex connection, :212-233
  for (uint32_t totalLength = 2; doConnect && totalLength <= D; totalLength++) {
    for (uint32_t cameraLength = 1; cameraLength + 1 <= D; cameraLength++) {
      if (cameraLength > totalLength) continue;  // uint underflow -> OOB read, undefined in HLSL: skipped (quirk 3)
      uint32_t lightLength = totalLength - cameraLength;
      float G = evalGWithoutV(cameraPath[cameraLength], lightPath[lightLength]);
      f3 posA = cameraPath[cameraLength].posW;
      f3 posB = lightPath[lightLength].posW;
      float lengthAB = length(posB - posA);
      f3 dirAB = (posB - posA) / lengthAB;
      bool vis = shadowRayVisibility(g, tl, 5, posA, dirAB, g.p.minT, lengthAB);
      if (g.flags & ORACLE_CONNECT_ALL_VISIBLE) vis = true;  // cross-check hook, see bdpt_oracle.h
      if (vis) {
        f3 shade = getUnweightedContribution(g, cameraPath, lightPath, cameraLength, lightLength, G);
        shade = clampVec(g, applyStrategyWeight(g, cameraPath, lightPath, shade, totalLength, cameraLength, lightLength));
        bool colorsNan = isnan3(shade);
        out[0] = saturate(out[0] + (colorsNan ? 0.0f : shade.x));
        out[1] = saturate(out[1] + (colorsNan ? 0.0f : shade.y));
        out[2] = saturate(out[2] + (colorsNan ? 0.0f : shade.z));
        out[3] = saturate(out[3] + 1.0f);
      }
    }
  }
}

// BDPT/BDPTUtils.hlsli:64-88 == CP/lightProbeGBufferUtils.hlsli:45-69
inline float atan2_WAR(float y, float x) {
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

// GBufferRayGen + PrimaryClosestHit/AnyHit/Miss — CP/lightProbeGBuffer.rt.hlsl:63-159
void gbufferPixel(const oracle_scene& s, const bdpt_camera& cam, const bdpt_gbuffer_params& gp, const float* env, uint32_t flags,
                  Tally& tl, oracle_frame* f, uint32_t x, uint32_t y) {
  const uint32_t W = f->width, H = f->height;
  size_t pix = (size_t)y * W + x;
  float* oP = f->worldPosition + pix * 4;
  float* oN = f->worldNormal + pix * 4;
  float* oD = f->materialDiffuse + pix * 4;
  float* oS = f->materialSpecRough + pix * 4;
  float* oX = f->materialExtra + pix * 4;
  float* oE = f->emissive + pix * 4;
  for (int k = 0; k < 4; k++) oP[k] = oN[k] = oD[k] = oS[k] = oX[k] = oE[k] = 0.0f;  // getClearedTexture, :109-114 of the pass

  f3 U = ld3(cam.cameraU), V = ld3(cam.cameraV), Wv = ld3(cam.cameraW), camPos = ld3(cam.posW);
  float pcx = ((float)x + gp.pixelJitter[0]) / (float)W;
  float pcy = ((float)y + gp.pixelJitter[1]) / (float)H;
  float ndx = 2.0f * pcx + -1.0f;
  float ndy = -2.0f * pcy + 1.0f;
  f3 rayDir = U * ndx + V * ndy + Wv;
  rayDir = rayDir / length(Wv);
  f3 focalPoint = camPos + rayDir * gp.focalLen;
  uint32_t randSeed = initRand(x + y * W, gp.frameCount, 16);
  float r0 = nextRand(randSeed);
  float r1 = nextRand(randSeed);
  float sn, cs;
  det_sincos2pi(r0, sn, cs);
  float lr = gp.lensRadius * r1;
  float lu = cs * lr, lv = sn * lr;
  f3 randomOrig = camPos + normalize(U) * lu + normalize(V) * lv;
  Ray r;
  r.o = gp.useThinLens ? randomOrig : camPos;
  r.d = normalize(gp.useThinLens ? (focalPoint - randomOrig) : rayDir);
  r.tmin = 0.0f;
  r.tmax = 1e+38f;
  tl.rays[0]++;
  Hit h = traceRay(s, r, 1, flags, &tl.nodeC, &tl.triC);
  if (h.prim < 0) {  // PrimaryMiss :63-74
    f3 p = normalize(r.d);
    float u = (1.f + atan2_WAR(p.x, -p.z) * kInvPi) * 0.5f;
    float v = det_acos(p.y) * kInvPi;
    f3 c = mk(0);
    if (env) {
      uint32_t ex = (uint32_t)(u * (float)gp.envWidth), ey = (uint32_t)(v * (float)gp.envHeight);
      if (ex < gp.envWidth && ey < gp.envHeight) c = ld3(env + ((size_t)ey * gp.envWidth + ex) * 4);
    } else {
      c = ld3(gp.envColor);
    }
    oD[0] = halfRound(c.x);
    oD[1] = halfRound(c.y);
    oD[2] = halfRound(c.z);
    oD[3] = 1.0f;
    return;
  }
  ShadingData sd = prepareShadingData(s, (uint32_t)h.prim, h.u, h.v, camPos, true);
  oP[0] = sd.posW.x;
  oP[1] = sd.posW.y;
  oP[2] = sd.posW.z;
  oP[3] = 1.0f;
  oN[0] = halfRound(sd.N.x);
  oN[1] = halfRound(sd.N.y);
  oN[2] = halfRound(sd.N.z);
  oN[3] = halfRound(length(sd.posW - camPos));
  oD[0] = halfRound(sd.diffuse.x);
  oD[1] = halfRound(sd.diffuse.y);
  oD[2] = halfRound(sd.diffuse.z);
  oD[3] = halfRound(sd.opacity);
  oS[0] = halfRound(sd.specular.x);
  oS[1] = halfRound(sd.specular.y);
  oS[2] = halfRound(sd.specular.z);
  oS[3] = halfRound(sd.linearRoughness);
  oX[0] = halfRound(sd.IoR);
  oE[0] = halfRound(sd.emissive.x);
  oE[1] = halfRound(sd.emissive.y);
  oE[2] = halfRound(sd.emissive.z);
}

template <class Fn>
void parallelRows(uint32_t y0, uint32_t y1, int threads, Fn fn) {
  if (threads <= 1 || y1 - y0 < 2) {
    Tally tl;
    for (uint32_t y = y0; y < y1; y++) fn(y, tl, 0);
    fn(UINT32_MAX, tl, 0);
    return;
  }
  std::atomic<uint32_t> next{y0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++) {
    pool.emplace_back([&, t]() {
      Tally tl;
      for (;;) {
        uint32_t y = next.fetch_add(1);
        if (y >= y1) break;
        fn(y, tl, t);
      }
      fn(UINT32_MAX, tl, t);
    });
  }
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

oracle_scene* oracle_scene_create(const bdpt_scene_desc* d) {
  if (!d || !d->positions || !d->normals || !d->indices || !d->triMaterial || !d->materials) return nullptr;
  oracle_scene* s = new oracle_scene();
  s->pos.resize(d->numVertices);
  s->nrm.resize(d->numVertices);
  for (uint32_t i = 0; i < d->numVertices; i++) {
    s->pos[i] = ld3(d->positions + (size_t)i * 3);
    s->nrm[i] = ld3(d->normals + (size_t)i * 3);
  }
  if (d->bitangents) {
    s->hasBit = true;
    s->bit.resize(d->numVertices);
    for (uint32_t i = 0; i < d->numVertices; i++) s->bit[i] = ld3(d->bitangents + (size_t)i * 3);
  }
  if (d->texcoords) {
    s->hasUv = true;
    s->uv.resize(d->numVertices);
    for (uint32_t i = 0; i < d->numVertices; i++) s->uv[i] = ld3(d->texcoords + (size_t)i * 3);
  }
  s->idx.assign(d->indices, d->indices + (size_t)d->numTriangles * 3);
  s->triMat.assign(d->triMaterial, d->triMaterial + d->numTriangles);
  s->mats.assign(d->materials, d->materials + d->numMaterials);
  for (uint32_t i = 0; i < d->numTextures; i++) {
    Tex t;
    t.w = d->textures[i].width;
    t.h = d->textures[i].height;
    t.srgb = d->textures[i].srgb;
    t.px.assign(d->textures[i].rgba8, d->textures[i].rgba8 + (size_t)t.w * t.h * 4);
    s->tex.push_back(std::move(t));
  }
  if (d->lights) s->lights.assign(d->lights, d->lights + d->numLights);
  for (int i = 0; i < 256; i++) {
    double c = (double)i / 255.0;
    double l = (c <= 0.04045) ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
    s->srgbLut[i] = (float)l;
  }
  uint32_t n = d->numTriangles;
  s->tv0.resize(n);
  s->te1.resize(n);
  s->te2.resize(n);
  s->triNonOpaque.resize(n);
  s->triDoubleSided.resize(n);
  for (uint32_t t = 0; t < n; t++) {
    f3 a = s->pos[s->idx[(size_t)t * 3]], b = s->pos[s->idx[(size_t)t * 3 + 1]], c = s->pos[s->idx[(size_t)t * 3 + 2]];
    s->tv0[t] = a;
    s->te1[t] = b - a;
    s->te2[t] = c - a;
    const bdpt_material& m = s->mats[s->triMat[t]];
    // BLAS OPAQUE flag iff AlphaModeOpaque (F/Raytracing/RtModel.cpp:221-224);
    // TRIANGLE_CULL_DISABLE iff double-sided (F/Raytracing/RtScene.cpp:175-178)
    s->triNonOpaque[t] = BDPT_FLAG_ALPHA_MODE(m.flags) != BDPT_ALPHA_MODE_OPAQUE;
    s->triDoubleSided[t] = BDPT_FLAG_DOUBLE_SIDED(m.flags) != 0;
  }
  buildBvh(*s);
  return s;
}

void oracle_scene_destroy(oracle_scene* s) { delete s; }

void oracle_set_environment(oracle_scene* s, const bdpt_environment* env) {
  if (!s) return;
  s->envMap.clear();
  s->envW = s->envH = 0;
  for (int k = 0; k < 4; k++) s->envColor[k] = env ? env->color[k] : 0.0f;
  if (env && env->envMap && env->width && env->height) {
    s->envW = env->width;
    s->envH = env->height;
    s->envMap.assign(env->envMap, env->envMap + (size_t)env->width * env->height * 4);
  }
}

int oracle_gbuffer(const oracle_scene* s, const bdpt_camera* cam, const bdpt_gbuffer_params* gp, const float* envMapHost,
                   oracle_frame* f, uint32_t flags, int threads) {
  if (!s || !cam || !gp || !f) return -1;
  parallelRows(f->y0, f->y1, threads, [&](uint32_t y, Tally& tl, int) {
    if (y == UINT32_MAX) return;
    for (uint32_t x = 0; x < f->width; x++) gbufferPixel(*s, *cam, *gp, envMapHost, flags, tl, f, x, y);
  });
  return 0;
}

int oracle_bdpt(const oracle_scene* s, const bdpt_camera* cam, const bdpt_params* p, oracle_frame* f, uint32_t flags, int threads,
                bdpt_counters* counters) {
  if (!s || !cam || !p || !f) return -1;
  if (s->lights.empty() || p->maxDepth > BDPT_MAX_DEPTH) return -2;
  Globals g;
  g.s = s;
  g.cam = *cam;
  g.p = *p;
  g.flags = flags;
  g.W = f->width;
  g.H = f->height;
  g.lightsCount = (int)s->lights.size();
  std::vector<Tally> tallies((size_t)(threads > 1 ? threads : 1));
  parallelRows(f->y0, f->y1, threads, [&](uint32_t y, Tally& tl, int t) {
    if (y == UINT32_MAX) {
      tallies[(size_t)t] = tl;
      return;
    }
    for (uint32_t x = 0; x < f->width; x++) bdptPixel(g, tl, f, x, y);
  });
  if (counters) {
    memset(counters, 0, sizeof(*counters));
    for (const Tally& tl : tallies) {
      counters->raysEyeExtend += tl.rays[1];
      counters->raysLightExtend += tl.rays[2];
      counters->raysNee += tl.rays[3];
      counters->raysSplat += tl.rays[4];
      counters->raysConnect += tl.rays[5];
      counters->nodeVisitsClosest += tl.nodeC;
      counters->triTestsClosest += tl.triC;
      counters->nodeVisitsShadow += tl.nodeS;
      counters->triTestsShadow += tl.triS;
      counters->pixelsValid += tl.valid;
      counters->splatsLanded += tl.splats;
    }
  }
  return 0;
}

int oracle_resolve(oracle_frame* f) {
  if (!f) return -1;
  for (uint32_t y = f->y0; y < f->y1; y++)
    for (uint32_t x = 0; x < f->width; x++) {
      size_t pix = (size_t)y * f->width + x;
      const uint64_t* sp = f->splat + pix * 4;
      if (sp[3] == 0) continue;
      float* o = f->out + pix * 4;
      o[0] = saturate(o[0] + (float)sp[0] * 2.3283064365386963e-10f);
      o[1] = saturate(o[1] + (float)sp[1] * 2.3283064365386963e-10f);
      o[2] = saturate(o[2] + (float)sp[2] * 2.3283064365386963e-10f);
      o[3] = saturate(o[3] + (float)sp[3]);
    }
  return 0;
}

// CP/accumulate.ps.hlsl:28-42 + the two blits of SimpleAccumulationPass.cpp:127-133
void oracle_accumulate(float* lastFrame, float* curFrame, uint32_t accumCount, uint32_t maxAccumCount, uint64_t numTexels) {
  for (uint64_t i = 0; i < numTexels * 4; i++) {
    float cur = curFrame[i], prev = lastFrame[i];
    float r = (accumCount < maxAccumCount) ? ((float)accumCount * prev + cur) / (float)(accumCount + 1) : prev;
    curFrame[i] = r;
    lastFrame[i] = r;
  }
}

void oracle_rng(const uint32_t* val0, const uint32_t* val1, uint32_t n, uint32_t draws, uint32_t* out_states, float* out_floats) {
  for (uint32_t i = 0; i < n; i++) {
    uint32_t s = initRand(val0[i], val1[i], 16);
    for (uint32_t k = 0; k < draws; k++) {
      float r = nextRand(s);
      out_states[(size_t)i * draws + k] = s;
      out_floats[(size_t)i * draws + k] = r;
    }
  }
}

void oracle_trace(const oracle_scene* s, const float* rays, uint32_t n, int mode, uint32_t flags, int32_t* out_prim, float* out_tuv) {
  for (uint32_t i = 0; i < n; i++) {
    const float* p = rays + (size_t)i * 8;
    Ray r{ld3(p), ld3(p + 3), p[6], p[7]};
    Hit h = traceRay(*s, r, mode, flags, nullptr, nullptr);
    out_prim[i] = h.prim;
    if (mode == 2) {
      out_tuv[(size_t)i * 3] = out_tuv[(size_t)i * 3 + 1] = out_tuv[(size_t)i * 3 + 2] = 0.0f;
    } else {
      out_tuv[(size_t)i * 3] = h.prim < 0 ? 0.0f : h.t;
      out_tuv[(size_t)i * 3 + 1] = h.prim < 0 ? 0.0f : h.u;
      out_tuv[(size_t)i * 3 + 2] = h.prim < 0 ? 0.0f : h.v;
    }
  }
}

void oracle_bsdf(const float* in, uint32_t n, uint32_t matIndex, float* out) {
  Globals g;
  memset(&g, 0, sizeof(g));
  g.p.matIndex = matIndex & 1u;
  if (matIndex & 2u) g.p.flags |= BDPT_PARAM_SPECULAR_FROM_LOBE;
  for (uint32_t i = 0; i < n; i++) {
    const float* p = in + (size_t)i * 20;
    f3 N = ld3(p), V = ld3(p + 3), Lq = ld3(p + 6), dif = ld3(p + 9), spec = ld3(p + 12);
    float rough = p[15];
    bool isSpec = p[16] != 0.0f;
    uint32_t seed;
    memcpy(&seed, p + 17, 4);
    f3 L;
    float pdf;
    bool sSpec;
    f3 w = sampleBRDF(g, seed, N, N, V, dif, spec, rough, L, pdf, sSpec);
    f3 fr = evalBRDF(g, V, Lq, N, N, dif, spec, rough, isSpec);
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
}

void oracle_sincos2pi(const float* u, uint32_t n, float* s, float* c) {
  for (uint32_t i = 0; i < n; i++) det_sincos2pi(u[i], s[i], c[i]);
}
void oracle_half_round(const float* in, uint32_t n, float* out) {
  for (uint32_t i = 0; i < n; i++) out[i] = halfRound(in[i]);
}

}  // extern "C"
