// Scene.h — host scene container (geometry streams, materials, textures, lights, camera).
//
// Plays the role of Falcor's RtScene for this pass: what BDPTPass::initScene receives
// (BidirectionalPathtracing/Passes/BDPTPass.cpp:52-57) and what SceneLoaderWrapper
// post-processes (SharedUtils/SceneLoaderWrapper.cpp:56-103: linear sampler on every
// texture, default directional light when the file has none, camera aspect := W/H).
// Geometry is world-space and instancing-flattened (Model::LoadFlags::RemoveInstancing, :58).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/bdpt.h"

namespace bdpt {

struct float3 {
  float x, y, z;
};

// Camera::calculateCameraParameters inputs (Falcor Graphics/Camera/Camera.cpp:129-136,
// defaults Data/HostDeviceSharedCode.h:69-99).
class Camera {
 public:
  using SharedPtr = std::shared_ptr<Camera>;
  static SharedPtr create() { return SharedPtr(new Camera()); }
  void setPosition(float3 p) { mPos = p; mDirty = true; }
  void setTarget(float3 t) { mTarget = t; mDirty = true; }
  void setUpVector(float3 u) { mUp = u; mDirty = true; }
  void setFocalLength(float mm) { mFocalLength = mm; mDirty = true; }
  void setFrameHeight(float mm) { mFrameHeight = mm; mDirty = true; }
  void setAspectRatio(float a) { if (a != mAspect) { mAspect = a; mDirty = true; } }
  void setFocalDistance(float d) { mFocalDistance = d; mDirty = true; }
  void setJitter(float jx, float jy) { mJitterX = jx; mJitterY = jy; }
  float3 getPosition() const { return mPos; }
  float3 getTarget() const { return mTarget; }
  float3 getUpVector() const { return mUp; }
  float getAspectRatio() const { return mAspect; }
  // Increments whenever a view parameter changes; SimpleAccumulationPass compares it the way
  // the reference compares view matrices (CommonPasses/SimpleAccumulationPass.cpp:96-102).
  uint64_t getViewVersion() const { return mVersion; }
  const bdpt_camera& getData();
  // Camera::beginFrame (Falcor Camera.cpp:51-62, called by Scene::update, Scene.cpp:129): the jitter-free
  // view-projection of the state before this frame's camera changes becomes prevViewProjMat.
  void beginFrame();
  const float* getPrevViewProjMat() const { return mPrevViewProj; }  // row-major, see bdpt_bmfr_params

 private:
  Camera() = default;
  float3 mPos{0, 0, 0}, mTarget{0, 0, -1}, mUp{0, 1, 0};
  float mFocalLength = 21.0f, mFrameHeight = 24.0f, mAspect = 1.7777f, mFocalDistance = 10000.0f;
  float mJitterX = 0, mJitterY = 0;
  bool mDirty = true;
  uint64_t mVersion = 0;
  bdpt_camera mData{};
  float mNearZ = 0.1f, mFarZ = 1000.0f;  // HostDeviceSharedCode.h:82-84
  float mPrevViewProj[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
};

class Scene {
 public:
  using SharedPtr = std::shared_ptr<Scene>;
  static SharedPtr create() { return SharedPtr(new Scene()); }
  static SharedPtr createCornellBox();
  static SharedPtr createAtrium(uint32_t seed, uint32_t targetTriangles);
  static SharedPtr createAtrium(uint32_t seed, uint32_t targetTriangles, float foliageFraction);
  static SharedPtr createAtrium(uint32_t seed, uint32_t targetTriangles, float foliageFraction, bool uneven);  // Atrium.cpp: heavy-tailed triangle areas
  static SharedPtr createTriangleSoup(uint32_t seed, uint32_t numTriangles, float maxEdge);
  // `.fscene` (Falcor scene JSON) or `.obj` (+ .mtl, PPM/PGM/TGA textures); the role of
  // RtScene::loadFromFile behind SharedUtils/SceneLoaderWrapper.cpp:56-60.  nullptr + *error on failure.
  static SharedPtr loadFromFile(const std::string& path, std::string* error = nullptr);

  // Streams (12-byte stride each, Falcor ShadingUtils/Raytracing.slang:79-85)
  std::vector<float> positions, normals, bitangents, texcoords;
  std::vector<uint32_t> indices, triMaterial;
  std::vector<bdpt_material> materials;
  std::vector<bdpt_light> lights;
  struct Texture {
    uint32_t width = 0, height = 0, srgb = 0;
    std::vector<uint8_t> rgba8;
  };
  std::vector<Texture> textures;

  uint32_t getTriangleCount() const { return (uint32_t)(indices.size() / 3); }
  uint32_t getVertexCount() const { return (uint32_t)(positions.size() / 3); }
  uint32_t getLightCount() const { return (uint32_t)lights.size(); }
  Camera::SharedPtr getActiveCamera() const { return mCamera; }
  void setActiveCamera(Camera::SharedPtr c) { mCamera = c; }
  // SceneLoaderWrapper.cpp:71-78: a scene without lights gets this directional light.
  void addDefaultLightIfNone();
  // Fills a descriptor whose pointers alias this object's storage.
  void getDesc(bdpt_scene_desc* out);

  // mesh-building helpers used by the procedural factories
  uint32_t addVertex(float3 p, float3 n, float3 b, float u, float v);
  void addTriangle(uint32_t a, uint32_t b, uint32_t c, uint32_t material);
  void addQuad(float3 a, float3 b, float3 c, float3 d, uint32_t material);

 private:
  Scene() = default;
  Camera::SharedPtr mCamera;
  std::vector<bdpt_texture> mTexDescs;
};

}  // namespace bdpt
