// HostTypes.h — the few Falcor types the pass interface names, as headless stand-ins.
//
// The reference's passes are written against Falcor (RenderContext, Texture, Gui, Scene, ...).
// The drop-in boundary keeps the method names and argument meaning of ::RenderPass /
// ResourceManager (SharedUtils/RenderPass.h:25-220, SharedUtils/ResourceManager.h:26-173) and
// replaces the Falcor types by these: a RenderContext is a HIP device + stream, a Texture is a
// device buffer with a format, a Gui records widget calls (there is no window).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "Scene.h"

namespace bdpt {

struct vec2 { float x, y; };
struct vec4 { float x, y, z, w; };
struct ivec2 { int x, y; bool operator!=(const ivec2& o) const { return x != o.x || y != o.y; } };
struct uvec2 { uint32_t x, y; };

enum class ResourceFormat { RGBA32Float, RGBA16Float };
inline uint32_t getFormatBytesPerTexel(ResourceFormat f) { return f == ResourceFormat::RGBA32Float ? 16u : 8u; }

// Falcor::Resource::BindFlags — kept so requestTextureResource has the reference's signature.
enum class BindFlags : uint32_t { None = 0, ShaderResource = 1, UnorderedAccess = 2, RenderTarget = 4, DepthStencil = 8 };
inline BindFlags operator|(BindFlags a, BindFlags b) { return BindFlags(uint32_t(a) | uint32_t(b)); }
inline BindFlags& operator|=(BindFlags& a, BindFlags b) { a = a | b; return a; }

class RenderContext {
 public:
  explicit RenderContext(int device = 0, hipStream_t stream = nullptr) : mDevice(device), mStream(stream) {}
  int getDevice() const { return mDevice; }
  hipStream_t getStream() const { return mStream; }
  void setStream(hipStream_t s) { mStream = s; }  // frames in flight: the pipeline points the context at the frame's stream
  void flush(bool wait = true) { if (wait) (void)hipStreamSynchronize(mStream); }
 private:
  int mDevice;
  hipStream_t mStream;
};

class Texture {
 public:
  using SharedPtr = std::shared_ptr<Texture>;
  static SharedPtr create2D(uint32_t w, uint32_t h, ResourceFormat fmt);
  ~Texture();
  uint32_t getWidth() const { return mW; }
  uint32_t getHeight() const { return mH; }
  ResourceFormat getFormat() const { return mFormat; }
  void* getDevicePointer() const { return mData; }
  size_t getSizeInBytes() const { return (size_t)mW * mH * getFormatBytesPerTexel(mFormat); }
  // RenderContext::clearUAV: every texel := colour (converted to the texture's format)
  void clear(const vec4& c, hipStream_t stream);
  // read back as RGBA32F regardless of format (tests / image output)
  std::vector<float> download(hipStream_t stream) const;
  // raw bytes in the texture's own format (checkpoints)
  void downloadRaw(hipStream_t stream, std::vector<uint8_t>& out) const;
  bool uploadRaw(hipStream_t stream, const uint8_t* data, size_t size);
 private:
  Texture() = default;
  uint32_t mW = 0, mH = 0;
  ResourceFormat mFormat = ResourceFormat::RGBA32Float;
  void* mData = nullptr;
};

// Headless Gui: passes call the same add*Var functions; values can be scripted by name.
class Gui {
 public:
  std::map<std::string, double> overrides;   // name -> value to apply on the next renderGui
  std::vector<std::string> log;              // widgets seen, in order
  bool addIntVar(const char* name, int32_t& v, int lo, int hi);
  bool addFloatVar(const char* name, float& v, float lo, float hi, float step = 0.0f, bool sameLine = false);
  bool addCheckBox(const char* name, bool& v, bool sameLine = false);
  void addText(const char* text) { log.push_back(std::string("text:") + text); }
};

// Falcor::SampleConfig as far as Main.cpp:20-25 fills it in.  There is no window: width / height size the channels.
struct SampleConfig {
  struct WindowDesc {
    bool resizableWindow = false;
    uint32_t width = 1920, height = 1080;
    std::string title;
  } windowDesc;
};

struct KeyboardEvent { int key = 0; };
struct MouseEvent { int button = 0; float x = 0, y = 0; };

}  // namespace bdpt
