// Scene.cpp — camera basis, scene container, Cornell box and triangle-soup factories, and the
// C entry points of include/bdpt_scene.h.  CPU-only.
#include "Scene.h"

#include <cmath>
#include <cstring>

#include "../../include/bdpt_scene.h"

namespace bdpt {

namespace {
inline float3 sub(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float3 mul(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float3 crs(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dt(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 nrm(float3 a) {
  float l = std::sqrt(dt(a, a));
  return {a.x / l, a.y / l, a.z / l};
}
}  // namespace

const bdpt_camera& Camera::getData() {
  if (mDirty) {
    float pos[3] = {mPos.x, mPos.y, mPos.z}, tgt[3] = {mTarget.x, mTarget.y, mTarget.z}, up[3] = {mUp.x, mUp.y, mUp.z};
    bdpt_camera_look_at(pos, tgt, up, mFocalLength, mFrameHeight, mAspect, mFocalDistance, &mData);
    mDirty = false;
    mVersion++;
  }
  return mData;
}

void Camera::beginFrame() {
  float pos[3] = {mPos.x, mPos.y, mPos.z}, tgt[3] = {mTarget.x, mTarget.y, mTarget.z}, up[3] = {mUp.x, mUp.y, mUp.z};
  bdpt_camera_view_proj(pos, tgt, up, mFocalLength, mFrameHeight, mAspect, mNearZ, mFarZ, mPrevViewProj);
}

uint32_t Scene::addVertex(float3 p, float3 n, float3 b, float u, float v) {
  uint32_t id = getVertexCount();
  positions.insert(positions.end(), {p.x, p.y, p.z});
  normals.insert(normals.end(), {n.x, n.y, n.z});
  bitangents.insert(bitangents.end(), {b.x, b.y, b.z});
  texcoords.insert(texcoords.end(), {u, v, 0.0f});
  return id;
}

void Scene::addTriangle(uint32_t a, uint32_t b, uint32_t c, uint32_t material) {
  indices.insert(indices.end(), {a, b, c});
  triMaterial.push_back(material);
}

// Flat-shaded quad a,b,c,d (counter-clockwise seen from the side its normal faces).
void Scene::addQuad(float3 a, float3 b, float3 c, float3 d, uint32_t material) {
  float3 n = nrm(crs(sub(b, a), sub(d, a)));
  float3 bt = nrm(sub(d, a));
  uint32_t i0 = addVertex(a, n, bt, 0, 0), i1 = addVertex(b, n, bt, 1, 0), i2 = addVertex(c, n, bt, 1, 1),
           i3 = addVertex(d, n, bt, 0, 1);
  addTriangle(i0, i1, i2, material);
  addTriangle(i0, i2, i3, material);
}

void Scene::addDefaultLightIfNone() {
  if (!lights.empty()) return;
  bdpt_light l{};
  l.type = BDPT_LIGHT_DIRECTIONAL;
  float3 d = nrm({-0.189f, -0.861f, -0.471f});  // setWorldDirection normalises
  l.dirW[0] = d.x;
  l.dirW[1] = d.y;
  l.dirW[2] = d.z;
  l.intensity[0] = 10.0f;
  l.intensity[1] = 10.0f;
  l.intensity[2] = 9.85f;
  l.openingAngle = 3.14159265f;
  l.cosOpeningAngle = -1.0f;
  lights.push_back(l);
}

void Scene::getDesc(bdpt_scene_desc* out) {
  mTexDescs.clear();
  for (const Texture& t : textures) {
    bdpt_texture d{};
    d.rgba8 = t.rgba8.data();
    d.width = t.width;
    d.height = t.height;
    d.srgb = t.srgb;
    mTexDescs.push_back(d);
  }
  std::memset(out, 0, sizeof(*out));
  out->numVertices = getVertexCount();
  out->numTriangles = getTriangleCount();
  out->numMaterials = (uint32_t)materials.size();
  out->numTextures = (uint32_t)textures.size();
  out->numLights = (uint32_t)lights.size();
  out->positions = positions.data();
  out->normals = normals.data();
  out->bitangents = bitangents.empty() ? nullptr : bitangents.data();
  out->texcoords = texcoords.empty() ? nullptr : texcoords.data();
  out->indices = indices.data();
  out->triMaterial = triMaterial.data();
  out->materials = materials.data();
  out->textures = mTexDescs.empty() ? nullptr : mTexDescs.data();
  out->lights = lights.data();
}

static bdpt_material constMaterial(float r, float g, float b, float roughness, float metal, bool doubleSided = false) {
  bdpt_material m{};
  m.baseColor[0] = r;
  m.baseColor[1] = g;
  m.baseColor[2] = b;
  m.baseColor[3] = 1.0f;
  m.specular[0] = 0.0f;       // occlusion (unused)
  m.specular[1] = roughness;  // MetalRough: G = linear roughness
  m.specular[2] = metal;      //             B = metalness
  m.specular[3] = 0.0f;
  m.alphaThreshold = 0.5f;
  m.IoR = 1.0f;
  m.flags = BDPT_MAKE_FLAGS(BDPT_SHADING_MODEL_METAL_ROUGH, BDPT_CHANNEL_CONST, BDPT_CHANNEL_CONST, BDPT_CHANNEL_UNUSED,
                            BDPT_NORMAL_MAP_UNUSED, BDPT_ALPHA_MODE_OPAQUE, doubleSided ? 1u : 0u);
  m.texBaseColor = m.texSpecular = m.texEmissive = m.texNormal = -1;
  return m;
}

Scene::SharedPtr Scene::createCornellBox() {
  SharedPtr s = create();
  // 0 white, 1 red, 2 green, 3 emissive patch
  s->materials.push_back(constMaterial(0.73f, 0.73f, 0.73f, 1.0f, 0.0f));
  s->materials.push_back(constMaterial(0.65f, 0.05f, 0.05f, 1.0f, 0.0f));
  s->materials.push_back(constMaterial(0.12f, 0.45f, 0.15f, 1.0f, 0.0f));
  bdpt_material em = constMaterial(0.78f, 0.78f, 0.78f, 1.0f, 0.0f);
  em.emissive[0] = 0.85f;
  em.emissive[1] = 0.8f;
  em.emissive[2] = 0.6f;
  em.flags = BDPT_MAKE_FLAGS(BDPT_SHADING_MODEL_METAL_ROUGH, BDPT_CHANNEL_CONST, BDPT_CHANNEL_CONST, BDPT_CHANNEL_CONST,
                             BDPT_NORMAL_MAP_UNUSED, BDPT_ALPHA_MODE_OPAQUE, 0u);
  s->materials.push_back(em);
  auto Q = [&](float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz, float dx, float dy,
               float dz, uint32_t m) { s->addQuad({ax, ay, az}, {bx, by, bz}, {cx, cy, cz}, {dx, dy, dz}, m); };
  // Cornell box data (Cornell University Program of Computer Graphics); normals face the room.
  Q(552.8f, 0, 0, 0, 0, 0, 0, 0, 559.2f, 549.6f, 0, 559.2f, 0);                      // floor
  Q(556, 548.8f, 0, 556, 548.8f, 559.2f, 0, 548.8f, 559.2f, 0, 548.8f, 0, 0);        // ceiling
  Q(549.6f, 0, 559.2f, 0, 0, 559.2f, 0, 548.8f, 559.2f, 556, 548.8f, 559.2f, 0);     // back wall
  Q(0, 0, 559.2f, 0, 0, 0, 0, 548.8f, 0, 0, 548.8f, 559.2f, 2);                      // right wall (green)
  Q(552.8f, 0, 0, 549.6f, 0, 559.2f, 556, 548.8f, 559.2f, 556, 548.8f, 0, 1);        // left wall (red)
  Q(343, 548.7f, 227, 343, 548.7f, 332, 213, 548.7f, 332, 213, 548.7f, 227, 3);      // emissive patch
  // short block
  Q(130, 165, 65, 82, 165, 225, 240, 165, 272, 290, 165, 114, 0);
  Q(290, 0, 114, 290, 165, 114, 240, 165, 272, 240, 0, 272, 0);
  Q(130, 0, 65, 130, 165, 65, 290, 165, 114, 290, 0, 114, 0);
  Q(82, 0, 225, 82, 165, 225, 130, 165, 65, 130, 0, 65, 0);
  Q(240, 0, 272, 240, 165, 272, 82, 165, 225, 82, 0, 225, 0);
  // tall block
  Q(423, 330, 247, 265, 330, 296, 314, 330, 456, 472, 330, 406, 0);
  Q(423, 0, 247, 423, 330, 247, 472, 330, 406, 472, 0, 406, 0);
  Q(472, 0, 406, 472, 330, 406, 314, 330, 456, 314, 0, 456, 0);
  Q(314, 0, 456, 314, 330, 456, 265, 330, 296, 265, 0, 296, 0);
  Q(265, 0, 296, 265, 330, 296, 423, 330, 247, 423, 0, 247, 0);

  bdpt_light l{};
  l.type = BDPT_LIGHT_POINT;
  l.posW[0] = 278.0f;
  l.posW[1] = 530.0f;
  l.posW[2] = 279.6f;
  l.dirW[1] = -1.0f;
  l.intensity[0] = 170000.0f;
  l.intensity[1] = 150000.0f;
  l.intensity[2] = 115000.0f;
  l.openingAngle = 3.14159265f;
  l.cosOpeningAngle = -1.0f;
  l.penumbraAngle = 0.0f;
  s->lights.push_back(l);

  Camera::SharedPtr cam = Camera::create();
  cam->setPosition({278, 273, -800});
  cam->setTarget({278, 273, 0});
  cam->setUpVector({0, 1, 0});
  cam->setFrameHeight(24.0f);
  cam->setFocalLength(33.6f);  // fovY = 2 atan(12/33.6) = 39.3 deg
  cam->setFocalDistance(1.0f);
  s->setActiveCamera(cam);
  return s;
}

static inline uint32_t lcg(uint32_t& s) {
  s = 1664525u * s + 1013904223u;
  return s;
}
static inline float urand(uint32_t& s) { return (float)(lcg(s) >> 8) / 16777216.0f; }

Scene::SharedPtr Scene::createTriangleSoup(uint32_t seed, uint32_t numTriangles, float maxEdge) {
  SharedPtr s = create();
  s->materials.push_back(constMaterial(0.7f, 0.7f, 0.7f, 0.5f, 0.0f, false));
  s->materials.push_back(constMaterial(0.3f, 0.6f, 0.9f, 0.3f, 0.0f, true));  // double-sided
  uint32_t st = seed * 2654435761u + 12345u;
  for (uint32_t t = 0; t < numTriangles; t++) {
    float3 c{urand(st), urand(st), urand(st)};
    float3 p[3];
    for (int k = 0; k < 3; k++)
      p[k] = {c.x + (urand(st) - 0.5f) * maxEdge, c.y + (urand(st) - 0.5f) * maxEdge, c.z + (urand(st) - 0.5f) * maxEdge};
    float3 n = crs(sub(p[1], p[0]), sub(p[2], p[0]));
    float l = std::sqrt(dt(n, n));
    n = l > 0 ? mul(n, 1.0f / l) : float3{0, 1, 0};
    float3 e = sub(p[1], p[0]);
    float le = std::sqrt(dt(e, e));
    float3 bt = le > 0 ? mul(e, 1.0f / le) : float3{1, 0, 0};
    uint32_t i0 = s->addVertex(p[0], n, bt, 0, 0), i1 = s->addVertex(p[1], n, bt, 1, 0), i2 = s->addVertex(p[2], n, bt, 0, 1);
    s->addTriangle(i0, i1, i2, (lcg(st) >> 16) & 1u);
  }
  bdpt_light l{};
  l.type = BDPT_LIGHT_POINT;
  l.posW[0] = 0.5f;
  l.posW[1] = 1.5f;
  l.posW[2] = 0.5f;
  l.dirW[1] = -1.0f;
  l.intensity[0] = l.intensity[1] = l.intensity[2] = 3.0f;
  l.openingAngle = 3.14159265f;
  l.cosOpeningAngle = -1.0f;
  s->lights.push_back(l);
  Camera::SharedPtr cam = Camera::create();
  cam->setPosition({0.5f, 0.5f, -2.0f});
  cam->setTarget({0.5f, 0.5f, 0.5f});
  cam->setFocalDistance(1.0f);
  s->setActiveCamera(cam);
  return s;
}

}  // namespace bdpt

// ------------------------------------------------------------------------------------------------
// C entry points
// ------------------------------------------------------------------------------------------------
struct bdpt_scene {
  bdpt::Scene::SharedPtr scene;
};

extern "C" {

int bdpt_camera_look_at(const float pos[3], const float target[3], const float up[3], float focalLengthMm, float frameHeightMm,
                        float aspect, float focalDistance, bdpt_camera* out) {
  if (!pos || !target || !up || !out) return BDPT_E_INVALID;
  using namespace bdpt;
  float3 p{pos[0], pos[1], pos[2]}, t{target[0], target[1], target[2]}, u{up[0], up[1], up[2]};
  // Falcor Utils/Math/FalcorMath.h:148-151 and Graphics/Camera/Camera.cpp:129-136
  const float fovY = 2.0f * std::atan(0.5f * frameHeightMm / focalLengthMm);
  float3 W = mul(nrm(sub(t, p)), focalDistance);
  float3 U = nrm(crs(W, u));
  float3 V = nrm(crs(U, W));
  const float ulen = focalDistance * std::tan(fovY * 0.5f) * aspect;
  U = mul(U, ulen);
  const float vlen = focalDistance * std::tan(fovY * 0.5f);
  V = mul(V, vlen);
  out->posW[0] = p.x;
  out->posW[1] = p.y;
  out->posW[2] = p.z;
  out->cameraU[0] = U.x;
  out->cameraU[1] = U.y;
  out->cameraU[2] = U.z;
  out->cameraV[0] = V.x;
  out->cameraV[1] = V.y;
  out->cameraV[2] = V.z;
  out->cameraW[0] = W.x;
  out->cameraW[1] = W.y;
  out->cameraW[2] = W.z;
  return BDPT_OK;
}

int bdpt_camera_view_proj(const float pos[3], const float target[3], const float up[3], float focalLengthMm, float frameHeightMm,
                          float aspect, float nearZ, float farZ, float out16[16]) {
  if (!pos || !target || !up || !out16) return BDPT_E_INVALID;
  using namespace bdpt;
  const float3 eye{pos[0], pos[1], pos[2]};
  // glm::lookAt (right-handed), Camera.cpp:77
  const float3 f = nrm(sub(float3{target[0], target[1], target[2]}, eye));
  const float3 sv = nrm(crs(f, float3{up[0], up[1], up[2]}));
  const float3 uv = crs(sv, f);
  const float V[4][4] = {{sv.x, sv.y, sv.z, -dot(sv, eye)}, {uv.x, uv.y, uv.z, -dot(uv, eye)}, {-f.x, -f.y, -f.z, dot(f, eye)}, {0, 0, 0, 1}};
  // glm::perspective (right-handed, depth 0..1), Camera.cpp:89; only x, y and w matter to the reprojection
  const float fovY = 2.0f * std::atan(0.5f * frameHeightMm / focalLengthMm);
  const float th = std::tan(fovY * 0.5f);
  const float P[4][4] = {{1.0f / (aspect * th), 0, 0, 0}, {0, 1.0f / th, 0, 0}, {0, 0, farZ / (nearZ - farZ), -(farZ * nearZ) / (farZ - nearZ)}, {0, 0, -1, 0}};
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      float a = 0;
      for (int k = 0; k < 4; k++) a += P[r][k] * V[k][c];
      out16[4 * r + c] = a;
    }
  return BDPT_OK;
}

void bdpt_msaa_jitter(uint32_t frameCounterBeforeIncrement, float out[2]) {
  // kMSAA, BDPTPass.cpp:20 == LightProbeGBufferPass.cpp:36; both passes index it with the
  // counter AFTER the post-increment that fed gFrameCount (BDPTPass.cpp:81,97-98).
  static const float kMSAA[8][2] = {{1, -3}, {-1, 3}, {5, 1}, {-3, -5}, {-5, 5}, {-7, -1}, {3, 7}, {7, -7}};
  uint32_t i = (frameCounterBeforeIncrement + 1u) % 8u;
  out[0] = kMSAA[i][0] * 0.0625f + 0.5f;
  out[1] = kMSAA[i][1] * 0.0625f + 0.5f;
}

bdpt_scene* bdpt_scene_create_cornell(void) {
  bdpt_scene* h = new bdpt_scene();
  h->scene = bdpt::Scene::createCornellBox();
  return h;
}
bdpt_scene* bdpt_scene_create_atrium(uint32_t seed, uint32_t targetTriangles) {
  bdpt_scene* h = new bdpt_scene();
  h->scene = bdpt::Scene::createAtrium(seed, targetTriangles);
  return h;
}
bdpt_scene* bdpt_scene_create_courtyard(uint32_t seed, uint32_t targetTriangles, float foliageFraction) {
  bdpt_scene* h = new bdpt_scene();
  h->scene = bdpt::Scene::createAtrium(seed, targetTriangles, foliageFraction);
  return h;
}
bdpt_scene* bdpt_scene_create_atrium_uneven(uint32_t seed, uint32_t targetTriangles) {
  bdpt_scene* h = new bdpt_scene();
  h->scene = bdpt::Scene::createAtrium(seed, targetTriangles, 0.0f, true);
  return h;
}
bdpt_scene* bdpt_scene_create_soup(uint32_t seed, uint32_t numTriangles, float maxEdge) {
  bdpt_scene* h = new bdpt_scene();
  h->scene = bdpt::Scene::createTriangleSoup(seed, numTriangles, maxEdge);
  return h;
}
void bdpt_scene_destroy(bdpt_scene* s) { delete s; }
int bdpt_scene_get_desc(const bdpt_scene* s, bdpt_scene_desc* out) {
  if (!s || !out || !s->scene) return BDPT_E_INVALID;
  s->scene->getDesc(out);
  return BDPT_OK;
}
int bdpt_scene_get_camera(const bdpt_scene* s, float aspect, bdpt_camera* out) {
  if (!s || !out || !s->scene || !s->scene->getActiveCamera()) return BDPT_E_INVALID;
  bdpt::Camera::SharedPtr c = s->scene->getActiveCamera();
  c->setAspectRatio(aspect);
  *out = c->getData();
  return BDPT_OK;
}

}  // extern "C"
