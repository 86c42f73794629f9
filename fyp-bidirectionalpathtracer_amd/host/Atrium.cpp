// Atrium.cpp — seeded procedural stand-in for Crytek Sponza (BASELINE config 3).
//
// Neither the reference tree nor this image holds Sponza (or Bistro / San Miguel), and there is no
// network, so the bench scene is generated: a colonnaded two-storey hall, open roof over the
// nave, arches, hanging drapes, relief walls, gilt urns, torus-knot ornaments and alpha-masked
// lattice railings; sRGB base-colour textures, a roughness/metal map, a tangent-space normal map,
// MetalRough and SpecGloss materials, two point lights and one spot light with penumbra.  The
// triangle count is padded to exactly `targetTriangles` with pebbles so it equals Sponza's 262,144
// when asked.  Everything derives from `seed`; the same seed gives the same bytes.
#include <cmath>
#include <functional>

#include "Scene.h"

namespace bdpt {
namespace {

struct Rng {
  uint32_t s;
  explicit Rng(uint32_t seed) : s(seed * 747796405u + 2891336453u) {}
  uint32_t next() {
    s = 1664525u * s + 1013904223u;
    return s;
  }
  float uni() { return (float)(next() >> 8) / 16777216.0f; }
  float range(float a, float b) { return a + (b - a) * uni(); }
};

inline float3 add(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float3 sub(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 mul(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 crs(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dt(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 nrm(float3 a, float3 fallback = {0, 1, 0}) {
  float l = std::sqrt(dt(a, a));
  return l > 1e-20f ? mul(a, 1.0f / l) : fallback;
}

// integer lattice value noise (deterministic, no libm beyond floor)
inline float hash2(int x, int y, uint32_t seed) {
  uint32_t h = (uint32_t)x * 374761393u + (uint32_t)y * 668265263u + seed * 2246822519u;
  h = (h ^ (h >> 13)) * 1274126177u;
  h ^= h >> 16;
  return (float)(h & 0xFFFFFFu) / 16777216.0f;
}
inline float vnoise(float x, float y, uint32_t seed) {
  float fx = std::floor(x), fy = std::floor(y);
  int ix = (int)fx, iy = (int)fy;
  float tx = x - fx, ty = y - fy;
  tx = tx * tx * (3.0f - 2.0f * tx);
  ty = ty * ty * (3.0f - 2.0f * ty);
  float a = hash2(ix, iy, seed), b = hash2(ix + 1, iy, seed), c = hash2(ix, iy + 1, seed), d = hash2(ix + 1, iy + 1, seed);
  return (a + (b - a) * tx) + ((c + (d - c) * tx) - (a + (b - a) * tx)) * ty;
}
inline float fbm(float x, float y, uint32_t seed) {
  return 0.5f * vnoise(x, y, seed) + 0.25f * vnoise(x * 2.03f, y * 2.03f, seed + 1) + 0.125f * vnoise(x * 4.01f, y * 4.01f, seed + 2) +
         0.125f * vnoise(x * 8.05f, y * 8.05f, seed + 3);
}

using Surf = std::function<float3(float, float)>;

// (nu x nv) quads of a parametric surface; normals from central differences, bitangent = d/dv.
// Winding: normal = dP/du x dP/dv (flip reverses).  Returns triangles added.
uint32_t addParametric(Scene& s, int nu, int nv, const Surf& f, uint32_t mat, float uvU, float uvV, bool flip) {
  if (nu < 1) nu = 1;
  if (nv < 1) nv = 1;
  const uint32_t base = s.getVertexCount();
  const float eu = 0.25f / (float)nu, ev = 0.25f / (float)nv;
  for (int j = 0; j <= nv; j++)
    for (int i = 0; i <= nu; i++) {
      float u = (float)i / (float)nu, v = (float)j / (float)nv;
      float3 p = f(u, v);
      float3 du = sub(f(u + eu, v), f(u - eu, v));
      float3 dv = sub(f(u, v + ev), f(u, v - ev));
      float3 n = nrm(crs(du, dv));
      if (flip) n = mul(n, -1.0f);
      s.addVertex(p, n, nrm(dv, {1, 0, 0}), u * uvU, v * uvV);
    }
  uint32_t added = 0;
  for (int j = 0; j < nv; j++)
    for (int i = 0; i < nu; i++) {
      uint32_t a = base + (uint32_t)(j * (nu + 1) + i), b = a + 1, c = a + (uint32_t)(nu + 1) + 1, d = a + (uint32_t)(nu + 1);
      if (!flip) {
        s.addTriangle(a, b, c, mat);
        s.addTriangle(a, c, d, mat);
      } else {
        s.addTriangle(a, c, b, mat);
        s.addTriangle(a, d, c, mat);
      }
      added += 2;
    }
  return added;
}

Scene::Texture makeTexture(uint32_t w, uint32_t h, uint32_t srgb, const std::function<void(float, float, uint8_t*)>& fn) {
  Scene::Texture t;
  t.width = w;
  t.height = h;
  t.srgb = srgb;
  t.rgba8.resize((size_t)w * h * 4);
  for (uint32_t y = 0; y < h; y++)
    for (uint32_t x = 0; x < w; x++) fn(((float)x + 0.5f) / (float)w, ((float)y + 0.5f) / (float)h, &t.rgba8[((size_t)y * w + x) * 4]);
  return t;
}
inline uint8_t u8(float v) {
  v = v < 0 ? 0 : (v > 1 ? 1 : v);
  return (uint8_t)(v * 255.0f + 0.5f);
}

bdpt_material texMaterial(uint32_t model, int texBase, const float base[4], int texSpec, const float spec[4], int texNormal,
                          bool alphaMask, bool doubleSided) {
  bdpt_material m{};
  for (int i = 0; i < 4; i++) {
    m.baseColor[i] = base[i];
    m.specular[i] = spec[i];
  }
  m.alphaThreshold = 0.5f;
  m.IoR = 1.5f;
  m.texBaseColor = (int16_t)texBase;
  m.texSpecular = (int16_t)texSpec;
  m.texEmissive = -1;
  m.texNormal = (int16_t)texNormal;
  m.flags = BDPT_MAKE_FLAGS(model, texBase >= 0 ? BDPT_CHANNEL_TEXTURE : BDPT_CHANNEL_CONST,
                            texSpec >= 0 ? BDPT_CHANNEL_TEXTURE : BDPT_CHANNEL_CONST, BDPT_CHANNEL_UNUSED,
                            texNormal >= 0 ? BDPT_NORMAL_MAP_RGB : BDPT_NORMAL_MAP_UNUSED,
                            alphaMask ? BDPT_ALPHA_MODE_MASK : BDPT_ALPHA_MODE_OPAQUE, doubleSided ? 1u : 0u);
  return m;
}

constexpr float kTwoPi = 6.28318530717958647692f;

struct Builder {
  Scene& s;
  float q;  // tessellation scale
  uint32_t seed;
  bool uneven = false;  // heavy-tailed variant: the large flat surfaces are two triangles each, the budget goes to the ornaments
  int seg(float n) const {
    int v = (int)(n * q + 0.5f);
    return v < 2 ? 2 : v;
  }
  // segments of a LARGE surface (floors, slabs, walls): one in the uneven variant — a real asset's walls are not
  // tessellated to the density of its drapes, and triangle areas spanning decades are what strains a SAH tree
  int big(float n) const { return uneven ? 1 : seg(n); }
};

enum Mat : uint32_t { M_FLOOR = 0, M_BRICK, M_STONE, M_DRAPE, M_GOLD, M_LATTICE, M_RELIEF, M_GLOSS, M_LAMP, M_PEBBLE, M_LEAF };

void buildGeometry(Builder& b) {
  Scene& s = b.s;
  const float X0 = -15, X1 = 15, Z0 = -6, Z1 = 6, YG = 5.0f, YT = 11.0f, ZG = 3.5f;
  // ground floor (gentle undulation so it is not two triangles)
  addParametric(s, b.big(96), b.big(40),
                [&](float u, float v) { return float3{X0 + (X1 - X0) * u, b.uneven ? 0.0f : 0.02f * fbm(u * 40, v * 16, b.seed + 11), Z1 - (Z1 - Z0) * v}; },
                M_FLOOR, 15, 6, false);
  // gallery floors (top faces up, underside faces down) on both sides
  for (int side = 0; side < 2; side++) {
    float za = side ? ZG : Z0, zb = side ? Z1 : -ZG;
    addParametric(s, b.big(60), b.big(6), [&](float u, float v) { return float3{X0 + (X1 - X0) * u, YG, zb - (zb - za) * v}; }, M_FLOOR, 15,
                  1.5f, false);
    addParametric(s, b.big(60), b.big(6), [&](float u, float v) { return float3{X0 + (X1 - X0) * u, YG - 0.3f, za + (zb - za) * v}; },
                  M_STONE, 15, 1.5f, false);
    // gallery edge fascia facing the nave
    float ze = side ? ZG : -ZG;
    float sgn = side ? -1.0f : 1.0f;
    addParametric(s, b.big(60), b.uneven ? 1 : 2, [&](float u, float v) { return float3{X0 + (X1 - X0) * u, YG - 0.3f + 0.3f * v, ze}; }, M_STONE, 15, 0.2f,
                  sgn < 0);
    // roof slab over the gallery
    addParametric(s, b.big(40), b.big(4), [&](float u, float v) { return float3{X0 + (X1 - X0) * u, YT, za + (zb - za) * v}; }, M_STONE, 15,
                  1.5f, false);
  }
  // side walls (relief: displaced bricks, normal mapped), facing inward
  for (int side = 0; side < 2; side++) {
    float z = side ? Z1 : Z0;
    float sgn = side ? -1.0f : 1.0f;
    addParametric(s, b.big(150), b.big(56),
                  [&](float u, float v) {
                    float d = b.uneven ? 0.0f : 0.06f * fbm(u * 60, v * 22, b.seed + 21 + (uint32_t)side);
                    return float3{X0 + (X1 - X0) * u, YT * v, z + sgn * d};
                  },
                  M_BRICK, 15, 5.5f, side == 1);
  }
  // end walls (deep relief panels)
  for (int side = 0; side < 2; side++) {
    float x = side ? X1 : X0;
    float sgn = side ? -1.0f : 1.0f;
    addParametric(s, b.big(72), b.big(64),
                  [&](float u, float v) {
                    float d = b.uneven ? 0.0f : 0.25f * fbm(u * 9, v * 8, b.seed + 31 + (uint32_t)side) + 0.05f * fbm(u * 50, v * 44, b.seed + 33);
                    return float3{x + sgn * d, YT * v, Z0 + (Z1 - Z0) * u};
                  },
                  M_RELIEF, 6, 5.5f, side == 0);
  }
  // columns: two rows, two storeys
  const int nCols = 7;
  for (int row = 0; row < 2; row++)
    for (int storey = 0; storey < 2; storey++)
      for (int c = 0; c < nCols; c++) {
        float cx = -12.0f + 4.0f * (float)c, cz = row ? ZG : -ZG;
        float y0 = storey ? YG : 0.0f, hgt = storey ? (YT - YG) : (YG - 0.3f);
        float r0 = storey ? 0.26f : 0.34f;
        // fluted shaft with entasis
        addParametric(s, b.seg(40), b.seg(26),
                      [&](float u, float v) {
                        float ang = kTwoPi * u;
                        float r = r0 * (1.0f - 0.12f * v * v) * (1.0f + 0.035f * std::cos(ang * 12.0f));
                        return float3{cx + r * std::cos(ang), y0 + 0.25f + (hgt - 0.5f) * v, cz - r * std::sin(ang)};
                      },
                      M_STONE, 4, 6, false);
        // torus base and capital
        for (int k = 0; k < 2; k++) {
          float yc = k ? (y0 + hgt - 0.14f) : (y0 + 0.13f);
          float R = r0 * (k ? 1.15f : 1.25f), rr = 0.12f;
          addParametric(s, b.seg(32), b.seg(10),
                        [&](float u, float v) {
                          float a = kTwoPi * u, t = kTwoPi * v;
                          float rad = R + rr * std::cos(t);
                          return float3{cx + rad * std::cos(a), yc + rr * std::sin(t), cz - rad * std::sin(a)};
                        },
                        M_GLOSS, 4, 1, false);
        }
      }
  // arches between ground-floor columns (half tori in the x-y plane)
  for (int row = 0; row < 2; row++)
    for (int c = 0; c + 1 < nCols; c++) {
      float cx = -10.0f + 4.0f * (float)c, cz = row ? ZG : -ZG;
      float R = 1.66f, rr = 0.17f, yc = YG - 0.3f - R - rr;
      addParametric(s, b.seg(36), b.seg(10),
                    [&](float u, float v) {
                      float a = 3.14159265f * u, t = kTwoPi * v;
                      float rad = R + rr * std::cos(t);
                      return float3{cx - rad * std::cos(a), yc + rad * std::sin(a), cz + rr * std::sin(t)};
                    },
                    M_STONE, 6, 1, false);
    }
  // drapes hanging from the upper gallery into the nave (double-sided cloth)
  Rng rng(b.seed + 101);
  for (int d = 0; d < 8; d++) {
    float cx = -11.0f + 3.1f * (float)d, cz = rng.range(-2.6f, 2.6f);
    float wdt = rng.range(1.6f, 2.4f), len = rng.range(3.0f, 5.5f), ph = rng.range(0, 6.28f), yaw = rng.range(-0.5f, 0.5f);
    float cyw = std::cos(yaw), syw = std::sin(yaw);
    uint32_t ns = b.seed + 200 + (uint32_t)d;
    addParametric(s, b.seg(44), b.seg(56),
                  [&](float u, float v) {
                    float lx = (u - 0.5f) * wdt;
                    float fold = (0.10f + 0.18f * v) * std::sin(u * 22.0f + ph + 2.0f * v) + 0.05f * fbm(u * 6, v * 6, ns);
                    return float3{cx + lx * cyw + fold * syw, 10.4f - len * v, cz - lx * syw + fold * cyw};
                  },
                  M_DRAPE, 3, 4, false);
  }
  // gilt urns on the nave floor
  for (int k = 0; k < 6; k++) {
    float cx = -10.0f + 4.0f * (float)k, cz = (k & 1) ? 1.9f : -1.9f;
    addParametric(s, b.seg(44), b.seg(36),
                  [&](float u, float v) {
                    float a = kTwoPi * u, t = v;  // profile: foot, belly, neck, lip
                    float r = 0.18f + 0.42f * std::sin(3.14159265f * std::pow(t, 0.8f)) * (1.0f - 0.35f * t) + 0.10f * t * t * t;
                    r *= 1.0f + 0.03f * std::cos(a * 10.0f) * std::sin(3.14159265f * t);
                    return float3{cx + r * std::cos(a), 0.02f + 1.25f * t, cz - r * std::sin(a)};
                  },
                  M_GOLD, 3, 2, false);
  }
  // torus-knot ornaments hanging in the nave
  for (int k = 0; k < 4; k++) {
    float cx = -9.0f + 6.0f * (float)k, cy = 7.2f + 0.5f * (float)(k & 1), cz = (k & 1) ? -0.8f : 0.8f;
    int P = 2 + (k & 1), Q = 3 + (k >> 1) * 2;
    auto center = [=](float t) {
      float a = kTwoPi * t;
      float r = 0.55f + 0.22f * std::cos((float)Q * a);
      return float3{cx + r * std::cos((float)P * a), cy + 0.22f * std::sin((float)Q * a), cz + r * std::sin((float)P * a)};
    };
    addParametric(s, b.seg(220), b.seg(12),
                  [&](float u, float v) {
                    float3 c0 = center(u), c1 = center(u + 0.002f);
                    float3 T = nrm(sub(c1, c0));
                    float3 N = nrm(crs(T, float3{0.31f, 0.9f, 0.29f}));
                    float3 B = crs(T, N);
                    float t = kTwoPi * v;
                    return add(c0, add(mul(N, 0.085f * std::cos(t)), mul(B, 0.085f * std::sin(t))));
                  },
                  (k & 1) ? M_GOLD : M_GLOSS, 24, 1, false);
  }
  // alpha-masked lattice railings along both gallery edges (any-hit work)
  for (int side = 0; side < 2; side++) {
    float z = side ? (ZG - 0.02f) : (-ZG + 0.02f);
    addParametric(s, b.seg(90), b.seg(3), [&](float u, float v) { return float3{X0 + 1 + (X1 - X0 - 2) * u, YG + 1.1f * v, z}; }, M_LATTICE, 60,
                  2.2f, false);
  }
  // lamp bodies next to the two point lights (emissive, tiny)
  const float lamps[2][3] = {{0.0f, 9.2f, 0.0f}, {-8.0f, 3.9f, -4.9f}};
  for (int k = 0; k < 2; k++) {
    float cx = lamps[k][0], cy = lamps[k][1] + 0.35f, cz = lamps[k][2];
    addParametric(s, b.seg(16), b.seg(10),
                  [&](float u, float v) {
                    float a = kTwoPi * u, t = 3.14159265f * v;
                    return float3{cx + 0.12f * std::sin(t) * std::cos(a), cy + 0.12f * std::cos(t), cz - 0.12f * std::sin(t) * std::sin(a)};
                  },
                  M_LAMP, 1, 1, true);
  }
}

void addPebbles(Scene& s, uint32_t count, uint32_t seed) {
  // tetrahedra (4 triangles) scattered on the nave floor, then single leaf triangles for the remainder
  Rng rng(seed + 777);
  auto tri = [&](float3 a, float3 b, float3 c) {
    float3 n = nrm(crs(sub(b, a), sub(c, a)));
    float3 bt = nrm(sub(c, a), {1, 0, 0});
    uint32_t i0 = s.addVertex(a, n, bt, 0, 0), i1 = s.addVertex(b, n, bt, 1, 0), i2 = s.addVertex(c, n, bt, 0, 1);
    s.addTriangle(i0, i1, i2, M_PEBBLE);
  };
  while (count >= 4) {
    float cx = rng.range(-14.0f, 14.0f), cz = rng.range(-3.0f, 3.0f), r = rng.range(0.02f, 0.06f), y = 0.02f;
    float a0 = rng.range(0, 6.28f);
    float3 p[3];
    for (int k = 0; k < 3; k++) {
      float a = a0 + 2.0943951f * (float)k;
      p[k] = {cx + r * std::cos(a), y, cz - r * std::sin(a)};
    }
    float3 top{cx, y + r * 1.2f, cz};
    tri(p[0], p[1], top);
    tri(p[1], p[2], top);
    tri(p[2], p[0], top);
    tri(p[0], p[2], p[1]);
    count -= 4;
  }
  while (count > 0) {
    float cx = rng.range(-14.0f, 14.0f), cz = rng.range(-3.0f, 3.0f);
    tri({cx, 0.05f, cz}, {cx + 0.05f, 0.05f, cz}, {cx, 0.05f, cz - 0.05f});
    count--;
  }
}

// Foliage for the courtyard variant (San Miguel stand-in, BASELINE config 5): `count` triangles as
// alpha-masked, double-sided leaf cards (two triangles each) in tree crowns standing in the nave, so that
// shadow and extension rays crossing the hall run the any-hit alpha test many times per query.
void addFoliage(Scene& s, uint32_t count, uint32_t seed) {
  Rng rng(seed + 4242);
  const int nTrees = 9;
  uint32_t cards = count / 2;
  for (uint32_t i = 0; i < cards; i++) {
    const int tree = (int)(i % (uint32_t)nTrees);
    const float tx = -12.0f + 3.0f * (float)tree, tz = (tree & 1) ? 1.3f : -1.3f, ty = 3.4f + 0.9f * (float)(tree % 3);
    // point in an ellipsoidal crown, denser towards the shell
    float3 d;
    do {
      d = {rng.range(-1, 1), rng.range(-1, 1), rng.range(-1, 1)};
    } while (dt(d, d) > 1.0f || dt(d, d) < 0.05f);
    const float sh = 0.55f + 0.45f * rng.uni();
    d = mul(nrm(d), sh);
    const float3 c{tx + 1.35f * d.x, ty + 1.9f * d.y, tz + 1.35f * d.z};
    // leaf card: random orientation, slightly drooping
    float3 ax = nrm(float3{rng.range(-1, 1), rng.range(-0.4f, 0.4f), rng.range(-1, 1)}, {1, 0, 0});
    float3 up = nrm(float3{rng.range(-0.5f, 0.5f), 1.0f, rng.range(-0.5f, 0.5f)});
    float3 ay = nrm(crs(crs(ax, up), ax), {0, 1, 0});
    const float hw = rng.range(0.05f, 0.11f), hh = rng.range(0.08f, 0.17f);
    const float3 n = nrm(crs(ax, ay));
    const float3 p0 = add(c, add(mul(ax, -hw), mul(ay, -hh))), p1 = add(c, add(mul(ax, hw), mul(ay, -hh)));
    const float3 p2 = add(c, add(mul(ax, hw), mul(ay, hh))), p3 = add(c, add(mul(ax, -hw), mul(ay, hh)));
    const uint32_t i0 = s.addVertex(p0, n, ay, 0, 0), i1 = s.addVertex(p1, n, ay, 1, 0), i2 = s.addVertex(p2, n, ay, 1, 1),
                   i3 = s.addVertex(p3, n, ay, 0, 1);
    s.addTriangle(i0, i1, i2, M_LEAF);
    s.addTriangle(i0, i2, i3, M_LEAF);
  }
  if (count & 1u) addPebbles(s, 1, seed + 1);
}

void addTexturesAndMaterials(Scene& s, uint32_t seed) {
  // 0 marble checker
  s.textures.push_back(makeTexture(512, 512, 1, [&](float u, float v, uint8_t* px) {
    int cx = (int)(u * 8), cy = (int)(v * 8);
    float vein = fbm(u * 14 + 3.0f * vnoise(u * 5, v * 5, seed), v * 14, seed + 1);
    float base = ((cx + cy) & 1) ? 0.82f : 0.28f;
    float c = base * (0.75f + 0.35f * vein);
    px[0] = u8(c);
    px[1] = u8(c * 0.96f);
    px[2] = u8(c * 0.90f);
    px[3] = 255;
  }));
  // 1 brick
  s.textures.push_back(makeTexture(512, 512, 1, [&](float u, float v, uint8_t* px) {
    float row = v * 16.0f;
    int r = (int)row;
    float bu = u * 8.0f + ((r & 1) ? 0.5f : 0.0f);
    float fu = bu - std::floor(bu), fv = row - (float)r;
    bool mortar = fu < 0.04f || fv < 0.08f;
    float n = fbm(u * 40, v * 40, seed + 5);
    float tone = 0.85f + 0.3f * hash2((int)std::floor(bu), r, seed + 6);
    if (mortar) {
      px[0] = u8(0.55f + 0.1f * n);
      px[1] = u8(0.53f + 0.1f * n);
      px[2] = u8(0.50f + 0.1f * n);
    } else {
      px[0] = u8((0.52f + 0.18f * n) * tone);
      px[1] = u8((0.25f + 0.10f * n) * tone);
      px[2] = u8((0.18f + 0.08f * n) * tone);
    }
    px[3] = 255;
  }));
  // 2 fabric stripes
  s.textures.push_back(makeTexture(256, 256, 1, [&](float u, float v, uint8_t* px) {
    float st = u * 12.0f - std::floor(u * 12.0f);
    float weave = 0.9f + 0.1f * (float)((((int)(u * 256) ^ (int)(v * 256)) & 1));
    bool a = st < 0.5f;
    px[0] = u8((a ? 0.70f : 0.85f) * weave);
    px[1] = u8((a ? 0.08f : 0.78f) * weave);
    px[2] = u8((a ? 0.10f : 0.55f) * weave);
    px[3] = 255;
  }));
  // 3 stone
  s.textures.push_back(makeTexture(512, 512, 1, [&](float u, float v, uint8_t* px) {
    float n = fbm(u * 24, v * 24, seed + 9), m = fbm(u * 90, v * 90, seed + 10);
    float c = 0.55f + 0.3f * n + 0.1f * m;
    px[0] = u8(c);
    px[1] = u8(c * 0.95f);
    px[2] = u8(c * 0.85f);
    px[3] = 255;
  }));
  // 4 occlusion / roughness / metal (linear)
  s.textures.push_back(makeTexture(256, 256, 0, [&](float u, float v, uint8_t* px) {
    float n = fbm(u * 18, v * 18, seed + 13);
    int cx = (int)(u * 8), cy = (int)(v * 8);
    px[0] = 255;
    px[1] = u8(((cx + cy) & 1) ? 0.18f + 0.2f * n : 0.45f + 0.4f * n);
    px[2] = 0;
    px[3] = 255;
  }));
  // 5 tangent-space normal map (RGB, linear): brick bevels + grain
  s.textures.push_back(makeTexture(512, 512, 0, [&](float u, float v, uint8_t* px) {
    auto hgt = [&](float uu, float vv) {
      float row = vv * 16.0f;
      int r = (int)std::floor(row);
      float bu = uu * 8.0f + ((r & 1) ? 0.5f : 0.0f);
      float fu = bu - std::floor(bu), fv = row - std::floor(row);
      float e = std::fmin(std::fmin(fu, 1 - fu) * 6.0f, std::fmin(fv, 1 - fv) * 3.0f);
      e = e > 1 ? 1 : e;
      return e + 0.15f * fbm(uu * 60, vv * 60, seed + 15);
    };
    float e = 1.0f / 512.0f;
    float dx = (hgt(u + e, v) - hgt(u - e, v)) * 1.5f, dy = (hgt(u, v + e) - hgt(u, v - e)) * 1.5f;
    float3 n = nrm({-dx, -dy, 1.0f});
    px[0] = u8(n.x * 0.5f + 0.5f);
    px[1] = u8(n.y * 0.5f + 0.5f);
    px[2] = u8(n.z * 0.5f + 0.5f);
    px[3] = 255;
  }));
  // 6 lattice with alpha holes
  s.textures.push_back(makeTexture(256, 256, 1, [&](float u, float v, uint8_t* px) {
    float a = u + v, b = u - v;
    float fa = a * 4.0f - std::floor(a * 4.0f), fb = b * 4.0f - std::floor(b * 4.0f);
    bool bar = fa < 0.22f || fb < 0.22f || v < 0.06f || v > 0.94f;
    px[0] = u8(0.30f);
    px[1] = u8(0.22f);
    px[2] = u8(0.12f);
    px[3] = bar ? 255 : 0;
  }));

  const float white[4] = {1, 1, 1, 1};
  const float specRough6[4] = {0, 0.6f, 0, 0}, specRough8[4] = {0, 0.8f, 0, 0}, specRough9[4] = {0, 0.92f, 0, 0};
  // M_FLOOR: marble, roughness from texture 4
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, 0, white, 4, specRough6, -1, false, false));
  // M_BRICK: normal mapped
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, 1, white, -1, specRough8, 5, false, false));
  // M_STONE
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, 3, white, -1, specRough6, -1, false, false));
  // M_DRAPE: double-sided cloth
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, 2, white, -1, specRough9, -1, false, true));
  // M_GOLD: metallic
  const float gold[4] = {0.95f, 0.72f, 0.28f, 1};
  const float specGold[4] = {0, 0.28f, 1.0f, 0};
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, -1, gold, -1, specGold, -1, false, false));
  // M_LATTICE: alpha mask, double-sided
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, 6, white, -1, specRough8, -1, true, true));
  // M_RELIEF: stone + normal map
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, 3, white, -1, specRough8, 5, false, false));
  // M_GLOSS: SpecGloss model, constant
  const float glossBase[4] = {0.55f, 0.57f, 0.62f, 1};
  const float glossSpec[4] = {0.22f, 0.22f, 0.24f, 0.72f};
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_SPEC_GLOSS, -1, glossBase, -1, glossSpec, -1, false, false));
  // M_LAMP: emissive
  bdpt_material lamp = texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, -1, white, -1, specRough8, -1, false, true);
  lamp.emissive[0] = 4.0f;
  lamp.emissive[1] = 3.4f;
  lamp.emissive[2] = 2.4f;
  lamp.flags = BDPT_MAKE_FLAGS(BDPT_SHADING_MODEL_METAL_ROUGH, BDPT_CHANNEL_CONST, BDPT_CHANNEL_CONST, BDPT_CHANNEL_CONST,
                               BDPT_NORMAL_MAP_UNUSED, BDPT_ALPHA_MODE_OPAQUE, 1u);
  s.materials.push_back(lamp);
  // M_PEBBLE
  const float peb[4] = {0.35f, 0.33f, 0.30f, 1};
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, -1, peb, -1, specRough8, -1, false, false));
}

// texture 7 + M_LEAF: a leaf silhouette in the alpha channel (about half of the card is cut away)
void addLeafMaterial(Scene& s, uint32_t seed) {
  s.textures.push_back(makeTexture(128, 128, 1, [&](float u, float v, uint8_t* px) {
    const float x = (u - 0.5f) * 2.0f, y = v;
    const float halfWidth = 0.95f * std::sqrt(std::fmax(0.0f, y * (1.0f - y))) * (1.0f + 0.15f * std::sin(y * 37.0f));
    const bool leaf = std::fabs(x) < halfWidth || (std::fabs(x) < 0.05f && y < 0.1f);
    const float vein = 0.85f + 0.15f * fbm(u * 20, v * 20, seed + 71);
    px[0] = u8(0.16f * vein);
    px[1] = u8(0.42f * vein);
    px[2] = u8(0.10f * vein);
    px[3] = leaf ? 255 : 0;
  }));
  const float white[4] = {1, 1, 1, 1};
  const float specRough7[4] = {0, 0.7f, 0, 0};
  s.materials.push_back(texMaterial(BDPT_SHADING_MODEL_METAL_ROUGH, (int)s.textures.size() - 1, white, -1, specRough7, -1, true, true));
}

}  // namespace

Scene::SharedPtr Scene::createAtrium(uint32_t seed, uint32_t targetTriangles) { return createAtrium(seed, targetTriangles, 0.0f, false); }
Scene::SharedPtr Scene::createAtrium(uint32_t seed, uint32_t targetTriangles, float foliageFraction) {
  return createAtrium(seed, targetTriangles, foliageFraction, false);
}

// foliageFraction > 0: the courtyard variant — that share of the triangles are alpha-masked leaf cards.
// uneven: the heavy-tailed variant — floors, slabs and walls are two flat triangles each (hundreds of square metres)
// and the whole triangle budget goes to columns, drapes, urns and ornaments (square millimetres): the same hall, the
// same count, triangle areas spread over six decades, as in assets modelled by hand.
Scene::SharedPtr Scene::createAtrium(uint32_t seed, uint32_t targetTriangles, float foliageFraction, bool uneven) {
  if (targetTriangles < 4096) targetTriangles = 4096;
  if (!(foliageFraction > 0.0f)) foliageFraction = 0.0f;
  if (foliageFraction > 0.9f) foliageFraction = 0.9f;
  const uint32_t foliage = (uint32_t)((double)targetTriangles * (double)foliageFraction) & ~1u;
  const uint32_t total = targetTriangles;
  targetTriangles -= foliage;
  // find the largest tessellation scale whose triangle count stays at or below the target
  float lo = 0.02f, hi = 8.0f;
  if (uneven) {  // the ornaments take the whole budget: bracket the scale by doubling first (a probe costs scale^2)
    hi = 1.0f;
    for (int it = 0; it < 8; it++) {
      SharedPtr probe = create();
      Builder b{*probe, hi, seed, true};
      buildGeometry(b);
      if (probe->getTriangleCount() > targetTriangles) break;
      lo = hi;
      hi *= 2.0f;
    }
  }
  for (int it = 0; it < 22; it++) {
    float mid = 0.5f * (lo + hi);
    SharedPtr probe = create();
    Builder b{*probe, mid, seed, uneven};
    buildGeometry(b);
    if (probe->getTriangleCount() <= targetTriangles)
      lo = mid;
    else
      hi = mid;
  }
  SharedPtr s = create();
  Builder b{*s, lo, seed, uneven};
  buildGeometry(b);
  if (s->getTriangleCount() < targetTriangles) addPebbles(*s, targetTriangles - s->getTriangleCount(), seed);
  if (foliage) addFoliage(*s, total > s->getTriangleCount() ? total - s->getTriangleCount() : 0u, seed);  // (never below zero: the base may overshoot a small target)
  addTexturesAndMaterials(*s, seed);
  if (foliage) addLeafMaterial(*s, seed);

  auto point = [&](float x, float y, float z, float r, float g, float bl) {
    bdpt_light l{};
    l.type = BDPT_LIGHT_POINT;
    l.posW[0] = x;
    l.posW[1] = y;
    l.posW[2] = z;
    l.dirW[1] = -1.0f;
    l.intensity[0] = r;
    l.intensity[1] = g;
    l.intensity[2] = bl;
    l.openingAngle = 3.14159265f;
    l.cosOpeningAngle = -1.0f;
    l.penumbraAngle = 0.0f;
    return l;
  };
  s->lights.push_back(point(0.0f, 9.2f, 0.0f, 95.0f, 88.0f, 74.0f));
  s->lights.push_back(point(-8.0f, 3.9f, -4.9f, 14.0f, 11.0f, 7.0f));
  bdpt_light spot = point(6.0f, 10.2f, 1.5f, 150.0f, 150.0f, 165.0f);
  float3 d = nrm({-0.25f, -1.0f, 0.1f});
  spot.dirW[0] = d.x;
  spot.dirW[1] = d.y;
  spot.dirW[2] = d.z;
  spot.openingAngle = 0.62f;
  spot.cosOpeningAngle = std::cos(0.62f);
  spot.penumbraAngle = 0.2f;
  s->lights.push_back(spot);

  Camera::SharedPtr cam = Camera::create();
  cam->setPosition({-13.2f, 2.1f, 0.6f});
  cam->setTarget({8.0f, 3.6f, -0.4f});
  cam->setUpVector({0, 1, 0});
  cam->setFocalLength(21.0f);
  cam->setFrameHeight(24.0f);
  cam->setFocalDistance(1.0f);
  s->setActiveCamera(cam);
  return s;
}

}  // namespace bdpt
