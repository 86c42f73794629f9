// SceneLoader.cpp — scene ingestion: Falcor `.fscene` JSON + Wavefront OBJ/MTL
// models + PPM/PGM/TGA textures, following the reference's loader semantics:
//   .fscene keys, degrees->radians, instances      Falcor Graphics/Scene/SceneImporter.cpp:106-175, 310-460
//   instance matrix = T * yawPitchRoll * S          Falcor Graphics/Model/ObjectInstance.h:271-278
//   wrapper defaults (default light, aspect)       SharedUtils/SceneLoaderWrapper.cpp:56-103
//   material rules                                 Falcor Graphics/Model/Loaders/AssimpModelImporter.cpp:326-417
//     (MetalRough unless "shading_model": "spec_gloss"; Kd->baseColor, Ks->specular rgb, Ns->specular.a,
//      Ke->emissive, d->baseColor.a, ".doublesided" name suffix; OBJ bump maps go to the normal-map slot)
//   channel type: texture / const / unused(lum==0)  Falcor Graphics/Material/Material.cpp:162-184
//   alpha mode = mask iff the base-colour texture has an alpha channel   Material.cpp:119-126
//   specular TEXTURES are dropped: Material::setSpecularTexture never stores its argument (Material.cpp:128-132)
//   colour textures are sRGB, normal maps linear     AssimpModelImporter.cpp:241-270
// What the reference gets from Assimp/FreeImage and this loader does not: FBX and other model formats,
// PNG/JPG/HDR images, smoothing groups (normals are taken from the file or set per face).
#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/bdpt_scene.h"
#include "Scene.h"

namespace bdpt {
namespace {

// ---------------------------------------------------------------------------------------------
// Host threads of the model loader (round 5).  A 10 M-triangle OBJ is 1.2 GB of text: read line by line through
// iostreams by one thread (round 4: 0.87 M triangles per second) it took ten times as long as the device builds the
// acceleration structure over it.  Everything below splits its input into `threads` contiguous ranges and is written so
// that the RESULT DOES NOT DEPEND ON THE THREAD COUNT, bit for bit: per-range results are concatenated in range order,
// and every order-dependent sum (smooth normals, bitangents) and every "first occurrence" (joined vertices) is taken in
// the file's face order by the ONE thread that owns the target (orderedForEach).
// BDPT_LOADER_THREADS (environment) or bdpt_scene_load_threads (C ABI) set the count; default: the host's cores, at most 32.
// ---------------------------------------------------------------------------------------------
int& loaderThreadsSetting() {
  static int t = 0;
  return t;
}
int loaderThreads() {
  const int g_loaderThreads = loaderThreadsSetting();
  int t = g_loaderThreads;
  if (t <= 0)
    if (const char* e = std::getenv("BDPT_LOADER_THREADS")) t = std::atoi(e);
  if (t <= 0) t = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  return std::max(1, std::min(t, 256));
}
// A big array WITHOUT the one-thread zero fill std::vector does (the loops below write every element they read): for a
// 10 M-triangle model the joins' scratch is over 1 GB, and page-faulting it in by one thread costs as much as the work.
template <class T>
struct Buf {
  T* p = nullptr;
  size_t n = 0;
  explicit Buf(size_t count) : p(static_cast<T*>(std::malloc(std::max<size_t>(count, 1) * sizeof(T)))), n(count) {
    if (!p) throw std::bad_alloc();
  }
  ~Buf() { std::free(p); }
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
};
// f(range index, begin, end) over [0, n) split into `threads` contiguous ranges
template <class F>
void parallelRanges(size_t n, int threads, const F& f) {
  threads = std::max(1, threads);
  if (threads == 1 || n < 4096) {
    f(0, 0, n);
    return;
  }
  std::vector<std::thread> pool;
  const size_t chunk = (n + (size_t)threads - 1) / (size_t)threads;
  for (int t = 1; t < threads; t++) {
    const size_t a = std::min(n, chunk * (size_t)t), b = std::min(n, a + chunk);
    if (a < b) pool.emplace_back([&f, t, a, b] { f(t, a, b); });
  }
  f(0, 0, std::min(n, chunk));
  for (std::thread& th : pool) th.join();
}
// f(task) for task in [0, count), handed out to `threads` workers (order of execution is free: tasks must be independent)
template <class F>
void parallelTasks(int count, int threads, const F& f) {
  threads = std::max(1, std::min(threads, count));
  if (threads == 1) {
    for (int k = 0; k < count; k++) f(k);
    return;
  }
  std::atomic<int> next{0};
  auto work = [&] {
    for (int k; (k = next.fetch_add(1)) < count;) f(k);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; t++) pool.emplace_back(work);
  work();
  for (std::thread& th : pool) th.join();
}
// For every item i in [0, n) with target(i) < numTargets (others are skipped): visit(i) is called by the one thread that
// owns target(i), and the items of one target are visited in INCREASING i — so `out[target(i)] += value(i)` gives the sum a
// serial loop over i gives, and "first item of a target" is well defined, whatever the thread count.
// (Items are first dealt to (producer range, owner) lists in order; an owner then walks the producers' lists in order.)
template <class Target, class Visit>
void orderedForEach(size_t n, size_t numTargets, int threads, const Target& target, const Visit& visit) {
  threads = std::max(1, threads);
  if (threads == 1 || n < 65536) {
    for (size_t i = 0; i < n; i++)
      if (target(i) < numTargets) visit(i);
    return;
  }
  const size_t T = (size_t)threads, per = (numTargets + T - 1) / T;
  std::vector<std::vector<std::vector<uint32_t>>> lists(T, std::vector<std::vector<uint32_t>>(T));
  parallelRanges(n, threads, [&](int t, size_t a, size_t b) {
    for (auto& l : lists[(size_t)t]) l.reserve((b - a) / T + 16);
    for (size_t i = a; i < b; i++) {
      const size_t k = target(i);
      if (k < numTargets) lists[(size_t)t][k / per].push_back((uint32_t)i);
    }
  });
  parallelTasks(threads, threads, [&](int o) {
    for (size_t t = 0; t < T; t++)
      for (uint32_t i : lists[t][(size_t)o]) visit(i);
  });
}

// ---------------------------------------------------------------------------------------------
// minimal JSON (objects, arrays, strings, numbers, true/false/null)
// ---------------------------------------------------------------------------------------------
struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
};
struct JsonParser {
  const std::string& s;
  size_t i = 0;
  bool ok = true;
  explicit JsonParser(const std::string& src) : s(src) {}
  void ws() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) i++;
  }
  Json value() {
    ws();
    Json j;
    if (i >= s.size()) {
      ok = false;
      return j;
    }
    char c = s[i];
    if (c == '{') {
      j.type = Json::Obj;
      i++;
      ws();
      if (i < s.size() && s[i] == '}') {
        i++;
        return j;
      }
      while (ok) {
        ws();
        Json k = value();
        if (k.type != Json::Str) ok = false;
        ws();
        if (i >= s.size() || s[i] != ':') ok = false;
        i++;
        Json v = value();
        j.obj.emplace_back(k.str, std::move(v));
        ws();
        if (i < s.size() && s[i] == ',') {
          i++;
          continue;
        }
        if (i < s.size() && s[i] == '}') {
          i++;
          break;
        }
        ok = false;
      }
    } else if (c == '[') {
      j.type = Json::Arr;
      i++;
      ws();
      if (i < s.size() && s[i] == ']') {
        i++;
        return j;
      }
      while (ok) {
        j.arr.push_back(value());
        ws();
        if (i < s.size() && s[i] == ',') {
          i++;
          continue;
        }
        if (i < s.size() && s[i] == ']') {
          i++;
          break;
        }
        ok = false;
      }
    } else if (c == '"') {
      j.type = Json::Str;
      i++;
      while (i < s.size() && s[i] != '"') {
        if (s[i] == '\\' && i + 1 < s.size()) {
          i++;
          char e = s[i];
          j.str += (e == 'n') ? '\n' : (e == 't') ? '\t' : e;
        } else {
          j.str += s[i];
        }
        i++;
      }
      i++;
    } else if (!s.compare(i, 4, "true")) {
      j.type = Json::Bool;
      j.b = true;
      i += 4;
    } else if (!s.compare(i, 5, "false")) {
      j.type = Json::Bool;
      i += 5;
    } else if (!s.compare(i, 4, "null")) {
      i += 4;
    } else {
      j.type = Json::Num;
      size_t st = i;
      while (i < s.size() && (std::isdigit((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.' || s[i] == 'e' || s[i] == 'E')) i++;
      if (st == i) {
        ok = false;
        return j;
      }
      j.num = std::strtod(s.substr(st, i - st).c_str(), nullptr);
    }
    return j;
  }
};
bool vec3Of(const Json* j, float v[3]) {
  if (!j || j->type != Json::Arr || j->arr.size() != 3) return false;
  for (int k = 0; k < 3; k++) v[k] = (float)j->arr[(size_t)k].num;
  return true;
}

std::string dirOf(const std::string& path) {
  size_t p = path.find_last_of("/\\");
  return p == std::string::npos ? "." : path.substr(0, p);
}
std::string lower(std::string s) {
  std::transform(s.begin(), s.end(), s.begin(), ::tolower);
  return s;
}
bool endsWith(const std::string& s, const std::string& suf) { return s.size() >= suf.size() && lower(s).compare(s.size() - suf.size(), suf.size(), suf) == 0; }

}  // namespace
int& loaderThreadsSettingPublic() { return loaderThreadsSetting(); }  // (bdpt_scene_load_threads, below)
// ImageDecode.cpp
bool decodePng(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba8, int& channels, std::string& err);
bool decodeJpeg(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba8, int& channels, std::string& err);
bool decodeHdr(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, std::vector<float>& rgba32f, std::string& err);
namespace {

// ---------------------------------------------------------------------------------------------
// images: PNG and baseline JPEG (ImageDecode.cpp), binary PPM (P6) / PGM (P5), TGA (uncompressed and RLE, 24/32-bit
// colour, 8-bit grey).  hasAlpha = the image is a 32-bit one for Falcor (Utils/Bitmap.cpp:104-126), which is what
// makes a material's alpha mode Mask (Graphics/Material/Material.cpp:120-126).  Grey images become (g, g, g, 255).
// ---------------------------------------------------------------------------------------------
bool loadImage(const std::string& path, Scene::Texture& out, bool& hasAlpha, std::string* why = nullptr) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  hasAlpha = false;
  const bool png = d.size() > 8 && d[0] == 0x89 && d[1] == 'P' && d[2] == 'N' && d[3] == 'G';
  const bool jpg = d.size() > 3 && d[0] == 0xff && d[1] == 0xd8;
  if (png || jpg) {
    int channels = 0;
    std::string err;
    const bool ok = png ? decodePng(d.data(), d.size(), out.width, out.height, out.rgba8, channels, err)
                        : decodeJpeg(d.data(), d.size(), out.width, out.height, out.rgba8, channels, err);
    if (!ok) {
      if (why) *why = err;
      return false;
    }
    hasAlpha = channels == 4;
    return true;
  }
  if (d.size() > 2 && d[0] == 'P' && (d[1] == '6' || d[1] == '5')) {
    size_t p = 2;
    auto num = [&]() -> long {
      for (;;) {
        while (p < d.size() && std::isspace(d[p])) p++;
        if (p < d.size() && d[p] == '#') {
          while (p < d.size() && d[p] != '\n') p++;
          continue;
        }
        break;
      }
      long v = 0;
      while (p < d.size() && std::isdigit(d[p])) v = v * 10 + (d[p++] - '0');
      return v;
    };
    long w = num(), h = num(), mx = num();
    p++;  // single whitespace after maxval
    const int ch = d[1] == '6' ? 3 : 1;
    if (w <= 0 || h <= 0 || mx != 255 || d.size() < p + (size_t)w * h * ch) return false;
    out.width = (uint32_t)w;
    out.height = (uint32_t)h;
    out.rgba8.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
      const uint8_t* s = &d[p + i * ch];
      out.rgba8[i * 4] = s[0];
      out.rgba8[i * 4 + 1] = s[ch == 3 ? 1 : 0];
      out.rgba8[i * 4 + 2] = s[ch == 3 ? 2 : 0];
      out.rgba8[i * 4 + 3] = 255;
    }
    return true;
  }
  if (endsWith(path, ".tga") && d.size() >= 18) {
    const int idLen = d[0], type = d[2], w = d[12] | (d[13] << 8), h = d[14] | (d[15] << 8), bpp = d[16], desc = d[17];
    if ((type != 2 && type != 3 && type != 10 && type != 11) || (bpp != 8 && bpp != 24 && bpp != 32) || w <= 0 || h <= 0) return false;
    const int ch = bpp / 8;
    size_t p = 18 + (size_t)idLen;
    std::vector<uint8_t> px((size_t)w * h * ch);
    if (type == 2 || type == 3) {
      if (d.size() < p + px.size()) return false;
      std::memcpy(px.data(), &d[p], px.size());
    } else {
      size_t o = 0;
      while (o < px.size() && p < d.size()) {
        int hdr = d[p++], cnt = (hdr & 127) + 1;
        if (hdr & 128) {
          if (p + ch > d.size()) return false;
          for (int k = 0; k < cnt && o + ch <= px.size(); k++, o += ch) std::memcpy(&px[o], &d[p], ch);
          p += ch;
        } else {
          size_t nb = (size_t)cnt * ch;
          if (p + nb > d.size() || o + nb > px.size()) return false;
          std::memcpy(&px[o], &d[p], nb);
          p += nb;
          o += nb;
        }
      }
    }
    out.width = (uint32_t)w;
    out.height = (uint32_t)h;
    out.rgba8.resize((size_t)w * h * 4);
    const bool topDown = (desc & 0x20) != 0;
    for (int y = 0; y < h; y++) {
      const int sy = topDown ? y : (h - 1 - y);
      for (int x = 0; x < w; x++) {
        const uint8_t* s = &px[((size_t)sy * w + x) * ch];
        uint8_t* o = &out.rgba8[((size_t)y * w + x) * 4];
        if (ch == 1) {
          o[0] = o[1] = o[2] = s[0];
          o[3] = 255;
        } else {  // BGR(A)
          o[0] = s[2];
          o[1] = s[1];
          o[2] = s[0];
          o[3] = ch == 4 ? s[3] : 255;
        }
      }
    }
    hasAlpha = (ch == 4);
    return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// OBJ / MTL
// ---------------------------------------------------------------------------------------------
struct MtlInfo {
  bdpt_material m;
  std::string name;
};
inline float lum(const float* c) { return 0.2126f * c[0] + 0.7152f * c[1] + 0.0722f * c[2]; }

struct ModelLoader {
  Scene& s;
  std::string dir;
  bool specGloss;
  std::map<std::string, int> texCache;
  std::string err;
  std::map<std::string, uint32_t> matIndex;  // per model file: instances share materials and textures
  std::map<std::string, bool> mtlLoaded;
  ModelLoader(Scene& sc, const std::string& d, bool sg) : s(sc), dir(d), specGloss(sg) {}

  // Image files decoded ahead of their use, all at once on the loader's threads (a real asset's 40 textures of 1024 x 1024
  // take longer to inflate than its geometry to parse); texture() then takes them from here in the order the material
  // library names them, so ids, messages and results are those of decoding one by one.
  struct Decoded {
    Scene::Texture tex;
    bool alpha = false, ok = false;
    std::string why;
  };
  std::map<std::string, Decoded> decoded;
  static std::string texturePath(const std::string& dir, const std::string& file) {
    std::string p = dir + "/" + file;
    std::replace(p.begin(), p.end(), '\\', '/');
    return p;
  }
  void predecode(const std::vector<std::string>& files) {
    std::vector<std::string> todo;
    for (const std::string& f : files) {
      const std::string p = texturePath(dir, f);
      if (!decoded.count(p) && std::find(todo.begin(), todo.end(), p) == todo.end()) todo.push_back(p);
    }
    std::vector<Decoded> out(todo.size());
    parallelTasks((int)todo.size(), loaderThreads(), [&](int k) {
      Decoded& d = out[(size_t)k];
      d.ok = loadImage(todo[(size_t)k], d.tex, d.alpha, &d.why);
    });
    for (size_t k = 0; k < todo.size(); k++) decoded[todo[k]] = std::move(out[k]);
  }

  int texture(const std::string& file, bool srgb, bool* hasAlpha) {
    std::string key = file + (srgb ? "|s" : "|l");
    auto it = texCache.find(key);
    if (it != texCache.end()) {
      if (hasAlpha) *hasAlpha = alphaOf[it->second];
      return it->second;
    }
    Scene::Texture t;
    bool a = false;
    const std::string p = texturePath(dir, file);
    int id = -1;
    std::string why;
    bool ok;
    auto pre = decoded.find(p);
    if (pre != decoded.end()) {  // (a file used both as colour and as linear data is decoded once and copied)
      ok = pre->second.ok;
      t = pre->second.tex;
      a = pre->second.alpha;
      why = pre->second.why;
    } else {
      ok = loadImage(p, t, a, &why);
    }
    if (ok) {
      t.srgb = srgb ? 1u : 0u;
      id = (int)s.textures.size();
      s.textures.push_back(std::move(t));
    } else {
      std::fprintf(stderr, "[SceneLoader] cannot load texture %s (supported: PNG, baseline JPEG, PPM/PGM, TGA)%s%s\n", p.c_str(),
                   why.empty() ? "" : ": ", why.c_str());
    }
    texCache[key] = id;
    alphaOf[id] = a;
    if (hasAlpha) *hasAlpha = a;
    return id;
  }
  std::map<int, bool> alphaOf;

  void finishMaterial(bdpt_material& m, bool doubleSided) {
    // Material.cpp:162-184: texture > const (luminance != 0) > unused; emissive likewise
    const uint32_t dif = m.texBaseColor >= 0 ? BDPT_CHANNEL_TEXTURE : (lum(m.baseColor) == 0 ? BDPT_CHANNEL_UNUSED : BDPT_CHANNEL_CONST);
    const uint32_t spc = lum(m.specular) == 0 ? BDPT_CHANNEL_UNUSED : BDPT_CHANNEL_CONST;  // specular textures are dropped (see header)
    const uint32_t emi = m.texEmissive >= 0 ? BDPT_CHANNEL_TEXTURE : (lum(m.emissive) == 0 ? BDPT_CHANNEL_UNUSED : BDPT_CHANNEL_CONST);
    const bool mask = m.texBaseColor >= 0 && alphaOf[m.texBaseColor];
    m.texSpecular = -1;
    m.flags = BDPT_MAKE_FLAGS(specGloss ? BDPT_SHADING_MODEL_SPEC_GLOSS : BDPT_SHADING_MODEL_METAL_ROUGH, dif, spc, emi,
                              m.texNormal >= 0 ? BDPT_NORMAL_MAP_RGB : BDPT_NORMAL_MAP_UNUSED, mask ? BDPT_ALPHA_MODE_MASK : BDPT_ALPHA_MODE_OPAQUE,
                              doubleSided ? 1u : 0u);
  }

  bool loadMtl(const std::string& file, std::map<std::string, uint32_t>& matIndex) {
    std::ifstream f(dir + "/" + file);
    if (!f) return false;
    std::string line;
    std::vector<std::string> lines;
    while (std::getline(f, line)) lines.push_back(line);
    {  // every image the library names, decoded side by side before the materials are read (predecode)
      std::vector<std::string> files;
      bool any = false;
      for (const std::string& ln : lines) {
        std::istringstream ss(ln);
        std::string k;
        if (!(ss >> k) || k[0] == '#') continue;
        if (k == "newmtl") any = true;
        if (!any || k.compare(0, 4, "map_") != 0) {
          if (!(any && (k == "bump" || k == "norm" || k == "disp"))) continue;
        }
        std::string t, last;
        while (ss >> t) last = t;
        if (!last.empty()) files.push_back(last);
      }
      predecode(files);
    }
    bdpt_material cur{};
    std::string curName;
    bool have = false, dbl = false;
    int nrmBump = -1, nrmNorm = -1, nrmDisp = -1;  // HEIGHT < NORMALS < DISPLACEMENT in aiTextureType order: the last one set wins
    auto flush = [&]() {
      if (!have) return;
      cur.texNormal = (int16_t)(nrmDisp >= 0 ? nrmDisp : nrmNorm >= 0 ? nrmNorm : nrmBump);
      // AssimpModelImporter.cpp:390-398: an OBJ material with a non-black Ke gets its BASE-COLOUR texture as emissive texture
      if (lum(cur.emissive) > 0) cur.texEmissive = cur.texBaseColor;
      finishMaterial(cur, dbl);
      matIndex[curName] = (uint32_t)s.materials.size();
      s.materials.push_back(cur);
      nrmBump = nrmNorm = nrmDisp = -1;
    };
    for (const std::string& mtlLine : lines) {
      std::istringstream ss(mtlLine);
      std::string k;
      if (!(ss >> k) || k[0] == '#') continue;
      if (k == "newmtl") {
        flush();
        have = true;
        std::string nm;
        ss >> nm;
        curName = nm;
        std::string ln = lower(nm);
        dbl = ln.find(".doublesided") != std::string::npos;  // AssimpModelImporter.cpp:405-413
        cur = bdpt_material{};
        cur.baseColor[0] = cur.baseColor[1] = cur.baseColor[2] = 0.6f;  // Assimp's OBJ material default (ObjFileData.h); Falcor copies it
        cur.baseColor[3] = 1.0f;
        cur.alphaThreshold = 0.5f;
        cur.IoR = 1.0f;
        cur.texBaseColor = cur.texSpecular = cur.texEmissive = cur.texNormal = -1;
      } else if (!have) {
        continue;
      } else if (k == "Kd") {
        ss >> cur.baseColor[0] >> cur.baseColor[1] >> cur.baseColor[2];
      } else if (k == "Ks") {
        ss >> cur.specular[0] >> cur.specular[1] >> cur.specular[2];
      } else if (k == "Ns") {
        ss >> cur.specular[3];
      } else if (k == "Ke") {
        ss >> cur.emissive[0] >> cur.emissive[1] >> cur.emissive[2];
      } else if (k == "d") {
        ss >> cur.baseColor[3];
      } else if (k == "Ni") {
        ss >> cur.IoR;
      } else if (k == "map_Kd" || k == "map_kd" || k == "map_Ke" || k == "map_emissive" || k == "map_bump" || k == "map_Bump" || k == "bump" ||
                 k == "norm" || k == "map_Kn" || k == "disp" || k == "map_disp") {
        std::string t, last;
        while (ss >> t) last = t;  // options such as `-bm 1.0` precede the file name
        if (last.empty()) continue;
        if (k == "map_Kd" || k == "map_kd")
          cur.texBaseColor = (int16_t)texture(last, true, nullptr);
        else if (k == "map_Ke" || k == "map_emissive")
          cur.texEmissive = (int16_t)texture(last, true, nullptr);
        else if (k == "norm" || k == "map_Kn")
          nrmNorm = texture(last, false, nullptr);
        else if (k == "disp" || k == "map_disp")
          nrmDisp = texture(last, false, nullptr);
        else
          nrmBump = texture(last, false, nullptr);
      }
    }
    flush();
    return true;
  }

  // BDPT_LOADER_VERBOSE: where the load time goes (stderr)
  std::chrono::steady_clock::time_point lapT = std::chrono::steady_clock::now();
  void lap(const char* what) {
    static const bool verbose = std::getenv("BDPT_LOADER_VERBOSE") != nullptr;
    if (!verbose) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[loader] %-22s %.3f s\n", what, std::chrono::duration<double>(t - lapT).count());
    lapT = t;
  }

  // ---- OBJ text, read in two parallel passes over line-aligned pieces of the file held in memory -----------------------
  // pass 1 counts what a piece holds (v / vt / vn lines, triangles after fan triangulation, mtllib / usemtl lines): the
  // prefix sums are every piece's place in the global arrays; pass 2 parses the numbers straight into those places.
  // Relative (negative) indices and the "not yet defined" checks use the counts SO FAR, as the line-by-line reader did.
  static bool isSpace(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }
  struct Cursor {  // one line
    const char *p, *end;
    void skip() {
      while (p < end && isSpace(*p)) p++;
    }
    bool token(const char*& a, const char*& b) {
      skip();
      if (p >= end) return false;
      a = p;
      while (p < end && !isSpace(*p)) p++;
      b = p;
      return true;
    }
    bool number(float& out) {  // what `stream >> float` reads: an optional sign, then a decimal floating-point number
      const char *a, *b;
      if (!token(a, b)) return false;
      if (a < b && *a == '+') a++;
      float v = 0.0f;
      const auto r = std::from_chars(a, b, v);
      if (r.ec != std::errc() && r.ec != std::errc::result_out_of_range) return false;
      if (r.ec == std::errc::result_out_of_range) v = (float)std::strtod(std::string(a, b).c_str(), nullptr);
      out = v;
      return true;
    }
  };
  static int atoiRange(const char* a, const char* b) {  // atoi on [a, b): optional sign, leading digits, rest ignored
    bool neg = false;
    if (a < b && (*a == '-' || *a == '+')) neg = *a++ == '-';
    long v = 0;
    while (a < b && *a >= '0' && *a <= '9' && v < (1l << 40)) v = v * 10 + (*a++ - '0');
    v = neg ? -v : v;
    return (int)std::max<long>(std::min<long>(v, 2147483647l), -2147483647l);
  }
  struct PieceCounts {
    size_t v = 0, vt = 0, vn = 0, tris = 0, events = 0;
  };
  struct Event {  // a mtllib / usemtl line: applies from triangle `tri` (global index) on
    bool lib;
    std::string name;
    size_t tri;
  };
  template <class Line>
  static void forEachLine(const char* a, const char* e, const Line& fn) {
    while (a < e) {
      const char* nl = static_cast<const char*>(std::memchr(a, '\n', (size_t)(e - a)));
      const char* le = nl ? nl : e;
      fn(Cursor{a, le});
      a = nl ? nl + 1 : e;
    }
  }
  static int keyword(Cursor& c) {  // 1 v, 2 vn, 3 vt, 4 f, 5 mtllib, 6 usemtl, 0 anything else (comments included)
    const char *a, *b;
    if (!c.token(a, b) || *a == '#') return 0;
    const size_t n = (size_t)(b - a);
    if (n == 1) return *a == 'v' ? 1 : (*a == 'f' ? 4 : 0);
    if (n == 2 && a[0] == 'v') return a[1] == 'n' ? 2 : (a[1] == 't' ? 3 : 0);
    if (n == 6 && std::memcmp(a, "mtllib", 6) == 0) return 5;
    if (n == 6 && std::memcmp(a, "usemtl", 6) == 0) return 6;
    return 0;
  }

  // appends the model transformed by `mat` (row-major 3x4) to the scene
  bool loadObj(const std::string& path, const float mat[12], const float nmat[9]) {
    std::string text;
    {
      std::ifstream f(path, std::ios::binary);
      if (!f) {
        err = "cannot open " + path;
        return false;
      }
      f.seekg(0, std::ios::end);
      const std::streamoff n = f.tellg();
      f.seekg(0, std::ios::beg);
      text.resize((size_t)std::max<std::streamoff>(n, 0));
      if (n > 0) f.read(&text[0], n);
    }
    dir = dirOf(path);
    const int threads = loaderThreads();
    lapT = std::chrono::steady_clock::now();
    if (std::getenv("BDPT_LOADER_VERBOSE")) std::fprintf(stderr, "[loader] %s: %zu bytes, %d threads\n", path.c_str(), text.size(), threads);
    // line-aligned pieces: a fixed piece SIZE (not a piece count) so that nothing below depends on the thread count
    constexpr size_t kPiece = 4u << 20;
    std::vector<size_t> cut{0};
    while (cut.back() < text.size()) {
      size_t e = std::min(text.size(), cut.back() + kPiece);
      if (e < text.size()) {
        const void* nl = std::memchr(text.data() + e, '\n', text.size() - e);
        e = nl ? (size_t)(static_cast<const char*>(nl) - text.data()) + 1 : text.size();
      }
      cut.push_back(e);
    }
    const int pieces = (int)cut.size() - 1;
    std::vector<PieceCounts> cnt((size_t)std::max(pieces, 0)), base((size_t)std::max(pieces, 0) + 1);
    parallelTasks(pieces, threads, [&](int k) {
      PieceCounts c;
      forEachLine(text.data() + cut[(size_t)k], text.data() + cut[(size_t)k + 1], [&](Cursor ln) {
        switch (keyword(ln)) {
          case 1: c.v++; break;
          case 2: c.vn++; break;
          case 3: c.vt++; break;
          case 4: {
            size_t corners = 0;
            const char *a, *b;
            while (ln.token(a, b)) corners++;
            if (corners >= 3) c.tris += corners - 2;
            break;
          }
          case 5: case 6: c.events++; break;
          default: break;
        }
      });
      cnt[(size_t)k] = c;
    });
    lap("pass 1 (count)");
    for (int k = 0; k < pieces; k++) {
      base[(size_t)k + 1].v = base[(size_t)k].v + cnt[(size_t)k].v;
      base[(size_t)k + 1].vt = base[(size_t)k].vt + cnt[(size_t)k].vt;
      base[(size_t)k + 1].vn = base[(size_t)k].vn + cnt[(size_t)k].vn;
      base[(size_t)k + 1].tris = base[(size_t)k].tris + cnt[(size_t)k].tris;
      base[(size_t)k + 1].events = base[(size_t)k].events + cnt[(size_t)k].events;
    }
    const PieceCounts total = base[(size_t)std::max(pieces, 0)];
    if (total.v > 0x7fffffffull || total.vt > 0x7fffffffull || total.vn > 0x7fffffffull || total.tris * 3 > 0xffffffffull) {
      err = "model too large in " + path;
      return false;
    }
    std::vector<float> P(total.v * 3), N(total.vn * 3), T(total.vt * 2);
    faces.assign(total.tris, Face{});
    std::vector<Event> events(total.events);
    std::atomic<bool> badIndex{false};
    parallelTasks(pieces, threads, [&](int k) {
      size_t iv = base[(size_t)k].v, ivt = base[(size_t)k].vt, ivn = base[(size_t)k].vn, itri = base[(size_t)k].tris, iev = base[(size_t)k].events;
      std::vector<Corner> cs;
      forEachLine(text.data() + cut[(size_t)k], text.data() + cut[(size_t)k + 1], [&](Cursor ln) {
        const int kw = keyword(ln);
        switch (kw) {
          case 1: {
            float a = 0, b = 0, c = 0;
            (void)(ln.number(a) && ln.number(b) && ln.number(c));
            P[iv * 3] = a;
            P[iv * 3 + 1] = b;
            P[iv * 3 + 2] = c;
            iv++;
            break;
          }
          case 2: {
            float a = 0, b = 0, c = 0;
            (void)(ln.number(a) && ln.number(b) && ln.number(c));
            N[ivn * 3] = a;
            N[ivn * 3 + 1] = b;
            N[ivn * 3 + 2] = c;
            ivn++;
            break;
          }
          case 3: {
            float a = 0, b = 0;
            (void)(ln.number(a) && ln.number(b));
            T[ivt * 2] = a;
            T[ivt * 2 + 1] = b;
            ivt++;
            break;
          }
          case 4: {
            cs.clear();
            const char *a, *b;
            const int nv = (int)iv, nt = (int)ivt, nn = (int)ivn;  // defined SO FAR
            while (ln.token(a, b)) {
              Corner c{0, 0, 0};
              int* dst[3] = {&c.v, &c.t, &c.n};
              const char* st = a;
              for (int part = 0; part < 3 && st <= b; part++) {
                const char* e = static_cast<const char*>(std::memchr(st, '/', (size_t)(b - st)));
                const char* pe = e ? e : b;
                if (pe > st) *dst[part] = atoiRange(st, pe);
                if (!e) break;
                st = e + 1;
              }
              if (c.v < 0) c.v = nv + c.v + 1;
              if (c.t < 0) c.t = nt + c.t + 1;
              if (c.n < 0) c.n = nn + c.n + 1;
              if (c.v < 1 || c.v > nv) badIndex.store(true);
              if (c.t > nt) c.t = 0;
              if (c.n > nn) c.n = 0;
              cs.push_back(c);
            }
            if (cs.size() < 3) break;
            for (size_t k2 = 1; k2 + 1 < cs.size(); k2++) faces[itri++] = Face{{cs[0], cs[k2], cs[k2 + 1]}, 0xFFFFFFFFu};  // fan triangulation
            break;
          }
          case 5: case 6: {
            const char *a, *b;
            std::string nm;
            if (ln.token(a, b)) nm.assign(a, b);
            events[iev++] = Event{kw == 5, nm, itri};
            break;
          }
          default: break;
        }
      });
    });
    lap("pass 2 (parse)");
    if (badIndex.load()) {
      err = "face index out of range in " + path;
      return false;
    }
    // materials: the few mtllib / usemtl lines, in file order, exactly as the line-by-line reader met them (a library is
    // read when its line is met; the default material is created when the first face without one is met)
    auto defaultMat = [&]() -> uint32_t {
      auto it = matIndex.find("\x01" "default");
      if (it != matIndex.end()) return it->second;
      bdpt_material m{};
      m.baseColor[0] = m.baseColor[1] = m.baseColor[2] = 0.6f;  // Assimp's "DefaultMaterial"
      m.baseColor[3] = 1.0f;
      m.alphaThreshold = 0.5f;
      m.IoR = 1.0f;
      m.texBaseColor = m.texSpecular = m.texEmissive = m.texNormal = -1;
      finishMaterial(m, false);
      matIndex["\x01" "default"] = (uint32_t)s.materials.size();
      s.materials.push_back(m);
      return matIndex["\x01" "default"];
    };
    struct Span {
      size_t first, last;
      uint32_t mat;
    };
    std::vector<Span> spans;
    {
      uint32_t curMat = 0xFFFFFFFFu;
      size_t from = 0;
      auto close = [&](size_t upTo) {  // triangles [from, upTo) were read under curMat
        if (upTo > from) {
          if (curMat == 0xFFFFFFFFu) curMat = defaultMat();
          spans.push_back(Span{from, upTo, curMat});
        }
        from = upTo;
      };
      for (const Event& ev : events) {
        close(ev.tri);
        if (ev.lib) {
          if (!mtlLoaded[ev.name]) loadMtl(ev.name, matIndex);
          mtlLoaded[ev.name] = true;
        } else {
          auto it = matIndex.find(ev.name);
          curMat = it != matIndex.end() ? it->second : defaultMat();
        }
      }
      close(faces.size());
    }
    parallelTasks((int)spans.size(), threads, [&](int k) {
      for (size_t f = spans[(size_t)k].first; f < spans[(size_t)k].last; f++) faces[f].mat = spans[(size_t)k].mat;
    });
    lap("materials + textures");
    emit(P, N, T, faces, mat, nmat);
    return true;
  }

  struct Corner {
    int v, t, n;
  };
  struct Face {
    Corner c[3];
    uint32_t mat;
  };
  std::vector<Face> faces;

  static float3 sub(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
  static float3 add(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
  static float3 mul(float3 a, float k) { return {a.x * k, a.y * k, a.z * k}; }
  static float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
  static float3 cross(float3 a, float3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
  static float3 nrm(float3 a) {
    float l = std::sqrt(dot(a, a));
    return {a.x / l, a.y / l, a.z / l};  // 0-length -> NaN, caught by invalid() like the reference's isInvalidVec
  }
  static bool invalid(float3 a) { return !std::isfinite(a.x) || !std::isfinite(a.y) || !std::isfinite(a.z); }
  // BinaryModelImporter.cpp:64-76
  static float3 projectNormalToBitangent(float3 n) {
    float3 b;
    if (std::fabs(n.x) > std::fabs(n.y))
      b = mul(float3{n.z, 0.f, -n.x}, 1.0f / std::sqrt(n.x * n.x + n.z * n.z));
    else
      b = mul(float3{0.f, n.z, -n.y}, 1.0f / std::sqrt(n.y * n.y + n.z * n.z));
    return nrm(b);
  }

  // Joins identical (position, texcoord, normal) corners per material (what Assimp's JoinIdenticalVertices leaves
  // Falcor with, one aiMesh per material), generates smooth normals when the file has none, and builds the
  // bitangent stream the way Falcor does (AssimpModelImporter.cpp:150-176 -> BinaryModelImporter.cpp:84-173),
  // on the pre-transformed (world-space) vertices.  Threaded; every "first occurrence" and every order-dependent float
  // sum is taken in the file's face order by the thread that owns the target (orderedForEach), so the vertex and
  // triangle lists are those of a one-thread run, bit for bit.
  void emit(const std::vector<float>& P, const std::vector<float>& N, const std::vector<float>& T, const std::vector<Face>& fs, const float mat[12],
            const float nmat[9]) {
    const int threads = loaderThreads();
    const size_t nF = fs.size(), nC = nF * 3;
    auto xf = [&](const float* p) -> float3 {
      return {mat[0] * p[0] + mat[1] * p[1] + mat[2] * p[2] + mat[3], mat[4] * p[0] + mat[5] * p[1] + mat[6] * p[2] + mat[7],
              mat[8] * p[0] + mat[9] * p[1] + mat[10] * p[2] + mat[11]};
    };
    auto xn = [&](float3 n) -> float3 {
      float3 r{nmat[0] * n.x + nmat[1] * n.y + nmat[2] * n.z, nmat[3] * n.x + nmat[4] * n.y + nmat[5] * n.z,
               nmat[6] * n.x + nmat[7] * n.y + nmat[8] * n.z};
      r = nrm(r);
      return invalid(r) ? float3{0, 1, 0} : r;
    };
    // smooth normals per position index (area-weighted face normals, added in face order), used for corners without vn
    std::vector<float3> smooth;
    std::atomic<bool> needSmooth{false};
    parallelRanges(nF, threads, [&](int, size_t f0, size_t f1) {
      bool need = false;
      for (size_t f = f0; f < f1 && !need; f++)
        for (int c = 0; c < 3; c++) need |= fs[f].c[c].n <= 0;
      if (need) needSmooth.store(true);
    });
    if (needSmooth.load()) {
      smooth.assign(P.size() / 3, float3{0, 0, 0});
      Buf<float3> fn(nF);
      parallelRanges(nF, threads, [&](int, size_t f0, size_t f1) {
        for (size_t f = f0; f < f1; f++) {
          float3 p[3];
          for (int c = 0; c < 3; c++) p[c] = {P[(size_t)(fs[f].c[c].v - 1) * 3], P[(size_t)(fs[f].c[c].v - 1) * 3 + 1], P[(size_t)(fs[f].c[c].v - 1) * 3 + 2]};
          fn[f] = cross(sub(p[1], p[0]), sub(p[2], p[0]));
        }
      });
      orderedForEach(nC, smooth.size(), threads, [&](size_t i) { return (size_t)fs[i / 3].c[i % 3].v - 1; },
                     [&](size_t i) {
                       float3& d = smooth[(size_t)fs[i / 3].c[i % 3].v - 1];
                       d = add(d, fn[i / 3]);
                     });
    }
    lap("smooth normals");
    // joined vertices: corner i -> the FIRST corner with the same (material, v, vt, vn); a corner that is its own first
    // becomes a vertex, numbered in corner order
    struct Key {
      uint32_t mat;
      int v, t, n;
      bool operator==(const Key& o) const { return mat == o.mat && v == o.v && t == o.t && n == o.n; }
    };
    auto keyOf = [&](size_t i) {
      const Corner& cr = fs[i / 3].c[i % 3];
      return Key{fs[i / 3].mat, cr.v, cr.t, cr.n};
    };
    auto hashOf = [](const Key& k) {
      uint64_t h = (uint64_t)k.mat * 0x9E3779B97F4A7C15ull;
      h ^= (uint64_t)(uint32_t)k.v * 0xC2B2AE3D27D4EB4Full;
      h = (h << 31) | (h >> 33);
      h ^= (uint64_t)(uint32_t)k.t * 0x165667B19E3779F9ull;
      h = (h << 29) | (h >> 35);
      h ^= (uint64_t)(uint32_t)k.n * 0x27D4EB2F165667C5ull;
      h ^= h >> 32;
      return h * 0x9E3779B97F4A7C15ull;
    };
    Buf<uint32_t> rep(nC);  // first corner of corner i's key
    {
      // owner of a key = a slice of the hash space; each owner keeps an open-addressing table of (first corner) slots
      constexpr size_t kOwnersBits = 8;  // 256 owners whatever the thread count (only the assignment of owners to threads varies)
      const size_t owners = (size_t)1 << kOwnersBits;
      Buf<uint64_t> hs(nC);
      parallelRanges(nC, threads, [&](int, size_t a0, size_t a1) {
        for (size_t i = a0; i < a1; i++) hs[i] = hashOf(keyOf(i));
      });
      std::vector<uint32_t> perOwner(owners + 1, 0);
      {
        std::vector<std::vector<uint32_t>> local((size_t)threads, std::vector<uint32_t>(owners, 0));
        parallelRanges(nC, threads, [&](int t, size_t a0, size_t a1) {
          for (size_t i = a0; i < a1; i++) local[(size_t)t][hs[i] >> (64 - kOwnersBits)]++;
        });
        for (size_t o = 0; o < owners; o++)
          for (int t = 0; t < threads; t++) perOwner[o + 1] += local[(size_t)t][o];
      }
      std::vector<std::vector<uint32_t>> table(owners);
      parallelTasks((int)owners, threads, [&](int o) {
        size_t cap = 16;
        while (cap < (size_t)perOwner[(size_t)o + 1] * 2) cap <<= 1;
        table[(size_t)o].assign(cap, 0xFFFFFFFFu);
      });
      orderedForEach(nC, owners, threads, [&](size_t i) { return (size_t)(hs[i] >> (64 - kOwnersBits)); },
                     [&](size_t i) {
                       std::vector<uint32_t>& tb = table[hs[i] >> (64 - kOwnersBits)];
                       const size_t mask = tb.size() - 1;
                       const Key k = keyOf(i);
                       for (size_t at = (size_t)hs[i] & mask;; at = (at + 1) & mask) {
                         if (tb[at] == 0xFFFFFFFFu) {
                           tb[at] = (uint32_t)i;  // corners of one key arrive in increasing i: this is the first
                           rep[i] = (uint32_t)i;
                           return;
                         }
                         if (keyOf(tb[at]) == k) {
                           rep[i] = tb[at];
                           return;
                         }
                       }
                     });
    }
    lap("join (first corners)");
    // vertex number of a first corner = how many first corners precede it (two-level prefix sum)
    Buf<uint32_t> vid(nC);
    size_t nV = 0;
    {
      const size_t blocks = (nC + 65535) / 65536;
      std::vector<uint32_t> blockCount(blocks + 1, 0);
      parallelTasks((int)blocks, threads, [&](int b) {
        uint32_t c = 0;
        for (size_t i = (size_t)b * 65536, e = std::min(nC, i + 65536); i < e; i++) c += rep[i] == i ? 1u : 0u;
        blockCount[(size_t)b + 1] = c;
      });
      for (size_t b = 0; b < blocks; b++) blockCount[b + 1] += blockCount[b];
      nV = blockCount[blocks];
      parallelTasks((int)blocks, threads, [&](int b) {
        uint32_t c = blockCount[(size_t)b];
        for (size_t i = (size_t)b * 65536, e = std::min(nC, i + 65536); i < e; i++)
          if (rep[i] == i) vid[i] = c++;
      });
    }
    Buf<float3> pos(nV), nor(nV), bit(nV);
    Buf<float> uvs(nV * 2);
    Buf<uint32_t> idx(nC);
    parallelRanges(nV, threads, [&](int, size_t a0, size_t a1) {
      for (size_t v = a0; v < a1; v++) bit[v] = float3{0, 0, 0};
    });
    parallelRanges(nC, threads, [&](int, size_t a0, size_t a1) {
      for (size_t i = a0; i < a1; i++) {
        idx[i] = vid[rep[i]];
        if (rep[i] != i) continue;
        const Corner& cr = fs[i / 3].c[i % 3];
        const uint32_t v = vid[i];
        pos[v] = xf(&P[(size_t)(cr.v - 1) * 3]);
        float3 n = cr.n > 0 ? float3{N[(size_t)(cr.n - 1) * 3], N[(size_t)(cr.n - 1) * 3 + 1], N[(size_t)(cr.n - 1) * 3 + 2]} : smooth[(size_t)cr.v - 1];
        nor[v] = xn(n);
        // aiProcess_FlipUVs (AssimpModelImporter.cpp:516): v -> 1 - v
        uvs[(size_t)v * 2] = cr.t > 0 ? T[(size_t)(cr.t - 1) * 2] : 0.0f;
        uvs[(size_t)v * 2 + 1] = cr.t > 0 ? 1.0f - T[(size_t)(cr.t - 1) * 2 + 1] : 0.0f;
      }
    });
    lap("vertices");
    // bitangents: every face's contribution to its three corners (parallel), then added per vertex in face order
    const bool haveUv = !T.empty();
    Buf<float3> contrib(nC);
    Buf<uint8_t> contribOk(nF);
    parallelRanges(nF, threads, [&](int, size_t f0, size_t f1) {
      for (size_t f = f0; f < f1; f++) {
        const uint32_t* i3 = &idx[f * 3];
        const float3 d0 = sub(pos[i3[1]], pos[i3[0]]), d1 = sub(pos[i3[2]], pos[i3[0]]);
        float sx = 0, sy = 0, tx = 0, ty = 0;
        if (haveUv) {
          sx = uvs[(size_t)i3[1] * 2] - uvs[(size_t)i3[0] * 2];
          sy = uvs[(size_t)i3[1] * 2 + 1] - uvs[(size_t)i3[0] * 2 + 1];
          tx = uvs[(size_t)i3[2] * 2] - uvs[(size_t)i3[0] * 2];
          ty = uvs[(size_t)i3[2] * 2 + 1] - uvs[(size_t)i3[0] * 2 + 1];
        }
        float3 tangent, bitangent;
        if ((sx == 0 && sy == 0) || (tx == 0 && ty == 0)) {
          bitangent = projectNormalToBitangent(nor[i3[0]]);
          tangent = cross(bitangent, nor[i3[0]]);
        } else {
          const float dc = 1.0f / (sx * ty - sy * tx);
          tangent = mul(sub(mul(d0, ty), mul(d1, tx)), dc);
          bitangent = mul(sub(mul(d1, sx), mul(d0, sy)), dc);  // sic: the reference uses s.y here, not t.x
        }
        contribOk[f] = invalid(bitangent) ? 0 : 1;
        for (int c = 0; c < 3; c++) {
          const float3 n = nor[i3[c]];
          float3 lt = nrm(sub(tangent, mul(n, dot(tangent, n))));
          float3 lb = nrm(sub(bitangent, mul(n, dot(bitangent, n))));
          lb = nrm(sub(lb, mul(lt, dot(lb, lt))));
          contrib[f * 3 + (size_t)c] = nrm(lb);
        }
      }
    });
    orderedForEach(nC, nV, threads, [&](size_t i) { return contribOk[i / 3] ? (size_t)idx[i] : (size_t)-1; },
                   [&](size_t i) { bit[idx[i]] = add(bit[idx[i]], contrib[i]); });
    lap("bitangents");
    const uint32_t base = s.getVertexCount();
    const size_t v0 = s.positions.size() / 3, t0 = s.indices.size() / 3;
    s.positions.resize((v0 + nV) * 3);
    s.normals.resize((v0 + nV) * 3);
    s.bitangents.resize((v0 + nV) * 3);
    s.texcoords.resize((v0 + nV) * 3);
    s.indices.resize((t0 + nF) * 3);
    s.triMaterial.resize(t0 + nF);
    parallelRanges(nV, threads, [&](int, size_t a0, size_t a1) {
      for (size_t v = a0; v < a1; v++) {
        float3 b = nrm(bit[v]);
        if (invalid(b)) b = projectNormalToBitangent(nor[v]);
        if (invalid(b)) b = float3{1, 0, 0};
        float* p = &s.positions[(v0 + v) * 3];
        p[0] = pos[v].x, p[1] = pos[v].y, p[2] = pos[v].z;
        float* n = &s.normals[(v0 + v) * 3];
        n[0] = nor[v].x, n[1] = nor[v].y, n[2] = nor[v].z;
        float* bb = &s.bitangents[(v0 + v) * 3];
        bb[0] = b.x, bb[1] = b.y, bb[2] = b.z;
        float* uv = &s.texcoords[(v0 + v) * 3];
        uv[0] = uvs[v * 2], uv[1] = uvs[v * 2 + 1], uv[2] = 0.0f;
      }
    });
    parallelRanges(nF, threads, [&](int, size_t f0, size_t f1) {
      for (size_t f = f0; f < f1; f++) {
        for (int c = 0; c < 3; c++) s.indices[(t0 + f) * 3 + (size_t)c] = base + idx[f * 3 + (size_t)c];
        s.triMaterial[t0 + f] = fs[f].mat;
      }
    });
    lap("append to scene");
  }
};

void instanceMatrix(const float t[3], const float yprDeg[3], const float sc[3], float m[12], float nm[9]) {
  // glm::yawPitchRoll(yaw, pitch, roll) = Ry(yaw) * Rx(pitch) * Rz(roll); M = T * R * S
  const float d2r = 3.14159265358979323846f / 180.0f;
  const float y = yprDeg[0] * d2r, p = yprDeg[1] * d2r, r = yprDeg[2] * d2r;
  const float cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
  const float R[9] = {cy * cr + sy * sp * sr, -cy * sr + sy * sp * cr, sy * cp,  //
                      cp * sr,                cp * cr,                 -sp,      //
                      -sy * cr + cy * sp * sr, sy * sr + cy * sp * cr, cy * cp};
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      m[i * 4 + j] = R[i * 3 + j] * sc[j];
      nm[i * 3 + j] = sc[j] != 0 ? R[i * 3 + j] / sc[j] : R[i * 3 + j];  // inverse transpose of R*S
    }
    m[i * 4 + 3] = t[i];
  }
}

}  // namespace

Scene::SharedPtr Scene::loadFromFile(const std::string& path, std::string* error) {
  auto fail = [&](const std::string& e) -> SharedPtr {
    if (error) *error = e;
    return nullptr;
  };
  SharedPtr s = create();
  const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, identN[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (endsWith(path, ".obj")) {
    ModelLoader ml(*s, dirOf(path), false);
    if (!ml.loadObj(path, ident, identN)) return fail(ml.err);
  } else if (endsWith(path, ".fscene")) {
    std::ifstream f(path);
    if (!f) return fail("cannot open " + path);
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string src = buf.str();
    JsonParser jp(src);
    Json root = jp.value();
    if (!jp.ok || root.type != Json::Obj) return fail("malformed JSON in " + path);
    const std::string dir = dirOf(path);
    if (const Json* models = root.get("models")) {
      for (const Json& jm : models->arr) {
        const Json* file = jm.get("file");
        if (!file || file->type != Json::Str) return fail("Model must have a filename");
        bool specGloss = false;
        if (const Json* mat = jm.get("material"))
          if (const Json* sm = mat->get("shading_model")) specGloss = sm->str == "spec_gloss";
        if (!endsWith(file->str, ".obj")) return fail("Could not load model: " + file->str + " (only OBJ is supported)");
        const Json* inst = jm.get("instances");
        std::vector<Json> one(1);
        const std::vector<Json>& list = (inst && inst->type == Json::Arr && !inst->arr.empty()) ? inst->arr : one;
        ModelLoader ml(*s, dir, specGloss);
        for (const Json& ji : list) {
          float t[3] = {0, 0, 0}, sc[3] = {1, 1, 1}, rot[3] = {0, 0, 0};
          vec3Of(ji.get("translation"), t);
          vec3Of(ji.get("scaling"), sc);
          vec3Of(ji.get("rotation"), rot);
          float m[12], nm[9];
          instanceMatrix(t, rot, sc, m, nm);
          if (!ml.loadObj(dir + "/" + file->str, m, nm)) return fail(ml.err);
        }
      }
    }
    // scene extents for the directional lights' far-away position (Light.cpp:199-210)
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i + 2 < s->positions.size(); i += 3)
      for (int k = 0; k < 3; k++) {
        lo[k] = std::min(lo[k], s->positions[i + k]);
        hi[k] = std::max(hi[k], s->positions[i + k]);
      }
    float center[3], radius = 0;
    for (int k = 0; k < 3; k++) {
      center[k] = 0.5f * (lo[k] + hi[k]);
      radius += (hi[k] - center[k]) * (hi[k] - center[k]);
    }
    radius = std::sqrt(radius);
    if (const Json* lights = root.get("lights")) {
      for (const Json& jl : lights->arr) {
        const Json* type = jl.get("type");
        bdpt_light l{};
        l.openingAngle = 3.14159265f;
        l.cosOpeningAngle = -1.0f;
        l.intensity[0] = l.intensity[1] = l.intensity[2] = 1.0f;
        l.dirW[1] = -1.0f;
        vec3Of(jl.get("intensity"), l.intensity);
        if (type && type->str == "dir_light") {
          l.type = BDPT_LIGHT_DIRECTIONAL;
          float d[3] = {0, -1, 0};
          vec3Of(jl.get("direction"), d);
          float n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
          for (int k = 0; k < 3; k++) {
            l.dirW[k] = d[k] / n;  // setWorldDirection normalises
            l.posW[k] = center[k] - l.dirW[k] * radius;
          }
        } else if (type && type->str == "point_light") {
          l.type = BDPT_LIGHT_POINT;
          vec3Of(jl.get("pos"), l.posW);
          vec3Of(jl.get("direction"), l.dirW);  // PointLight::setWorldDirection does not normalise
          if (const Json* a = jl.get("opening_angle")) {
            float ang = (float)a->num * 3.14159265358979323846f / 180.0f;
            ang = std::min(std::max(ang, 0.0f), 3.14159265358979323846f);
            l.openingAngle = ang;
            l.cosOpeningAngle = std::cos(ang);
          }
          if (const Json* a = jl.get("penumbra_angle")) l.penumbraAngle = (float)a->num * 3.14159265358979323846f / 180.0f;
        } else {
          continue;  // area lights: not on this path
        }
        if (s->lights.size() < BDPT_MAX_LIGHTS) s->lights.push_back(l);
      }
    }
    const Json* cams = root.get("cameras");
    const Json* active = root.get("active_camera");
    if (cams && cams->type == Json::Arr) {
      for (const Json& jc : cams->arr) {
        const Json* nm = jc.get("name");
        if (active && nm && nm->str != active->str && cams->arr.size() > 1) continue;
        Camera::SharedPtr cam = Camera::create();
        float v[3];
        if (vec3Of(jc.get("pos"), v)) cam->setPosition({v[0], v[1], v[2]});
        if (vec3Of(jc.get("target"), v)) cam->setTarget({v[0], v[1], v[2]});
        if (vec3Of(jc.get("up"), v)) cam->setUpVector({v[0], v[1], v[2]});
        if (const Json* fl = jc.get("focal_length")) cam->setFocalLength((float)fl->num);
        if (const Json* ar = jc.get("aspect_ratio")) cam->setAspectRatio((float)ar->num);
        cam->setFocalDistance(1.0f);
        s->setActiveCamera(cam);
        break;
      }
    }
  } else {
    return fail("unsupported scene file type: " + path);
  }
  if (s->getTriangleCount() == 0) return fail("scene has no triangles: " + path);
  s->addDefaultLightIfNone();  // SceneLoaderWrapper.cpp:71-78
  if (!s->getActiveCamera()) {  // SceneLoaderWrapper.cpp:81-95: look at the scene centre from 3 radii along +z
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i + 2 < s->positions.size(); i += 3)
      for (int k = 0; k < 3; k++) {
        lo[k] = std::min(lo[k], s->positions[i + k]);
        hi[k] = std::max(hi[k], s->positions[i + k]);
      }
    float c[3], r = 0;
    for (int k = 0; k < 3; k++) {
      c[k] = 0.5f * (lo[k] + hi[k]);
      r += (hi[k] - c[k]) * (hi[k] - c[k]);
    }
    r = std::sqrt(r);
    Camera::SharedPtr cam = Camera::create();
    cam->setPosition({c[0], c[1], c[2] + 3.0f * r});
    cam->setTarget({c[0], c[1], c[2]});
    cam->setUpVector({0, 1, 0});
    cam->setFocalDistance(1.0f);
    s->setActiveCamera(cam);
  }
  bool anyTex = false;
  for (const bdpt_material& m : s->materials) anyTex |= (m.texBaseColor >= 0 || m.texEmissive >= 0 || m.texNormal >= 0);
  (void)anyTex;
  return s;
}

}  // namespace bdpt

struct bdpt_scene {
  bdpt::Scene::SharedPtr scene;
};

// Allocation failures and anything else thrown while decoding a file stay on this side of the C boundary.
extern "C" int bdpt_image_load(const char* path, uint32_t* width, uint32_t* height, uint32_t* hasAlpha, uint8_t* rgba8, uint64_t cap, char* msg,
                               uint32_t msgCap) {
  if (!path || !width || !height) return BDPT_E_INVALID;
  try {
    bdpt::Scene::Texture t;
    bool a = false;
    std::string why;
    if (!bdpt::loadImage(path, t, a, &why)) {
      if (msg && msgCap) std::snprintf(msg, msgCap, "%s", why.empty() ? "cannot read or recognise the image" : why.c_str());
      return BDPT_E_INVALID;
    }
    *width = t.width;
    *height = t.height;
    if (hasAlpha) *hasAlpha = a ? 1u : 0u;
    if (rgba8) {
      if (cap < t.rgba8.size()) return BDPT_E_LIMIT;
      std::memcpy(rgba8, t.rgba8.data(), t.rgba8.size());
    }
    return BDPT_OK;
  } catch (const std::bad_alloc&) {
    if (msg && msgCap) std::snprintf(msg, msgCap, "out of memory while decoding the image");
    return BDPT_E_NOMEM;
  } catch (...) {
    if (msg && msgCap) std::snprintf(msg, msgCap, "internal error while decoding the image");
    return BDPT_E_INVALID;
  }
}

extern "C" int bdpt_image_load_hdr(const char* path, uint32_t* width, uint32_t* height, float* rgba32f, uint64_t capFloats, char* msg,
                                   uint32_t msgCap) {
  if (!path || !width || !height) return BDPT_E_INVALID;
  try {
    std::vector<uint8_t> d;
    FILE* f = std::fopen(path, "rb");
    if (!f) {
      if (msg && msgCap) std::snprintf(msg, msgCap, "cannot open %s", path);
      return BDPT_E_INVALID;
    }
    uint8_t buf[65536];
    for (size_t n; (n = std::fread(buf, 1, sizeof(buf), f)) > 0;) d.insert(d.end(), buf, buf + n);
    std::fclose(f);
    std::vector<float> px;
    std::string why;
    uint32_t w = 0, h = 0;
    if (!bdpt::decodeHdr(d.data(), d.size(), w, h, px, why)) {
      if (msg && msgCap) std::snprintf(msg, msgCap, "%s", why.c_str());
      return BDPT_E_INVALID;
    }
    *width = w;
    *height = h;
    if (rgba32f) {
      if (capFloats < px.size()) return BDPT_E_LIMIT;
      std::memcpy(rgba32f, px.data(), px.size() * sizeof(float));
    }
    return BDPT_OK;
  } catch (const std::bad_alloc&) {
    if (msg && msgCap) std::snprintf(msg, msgCap, "out of memory while decoding the image");
    return BDPT_E_NOMEM;
  } catch (...) {
    if (msg && msgCap) std::snprintf(msg, msgCap, "internal error while decoding the image");
    return BDPT_E_INVALID;
  }
}

extern "C" int bdpt_scene_load_threads(int threads) {
  const int before = bdpt::loaderThreadsSettingPublic();
  bdpt::loaderThreadsSettingPublic() = threads < 0 ? 0 : threads;
  return before;
}

extern "C" bdpt_scene* bdpt_scene_load(const char* path, char* msg, uint32_t msgCap) {
  if (!path) return nullptr;
  try {
    std::string err;
    bdpt::Scene::SharedPtr s = bdpt::Scene::loadFromFile(path, &err);
    if (!s) {
      if (msg && msgCap) std::snprintf(msg, msgCap, "%s", err.c_str());
      return nullptr;
    }
    bdpt_scene* h = new bdpt_scene();
    h->scene = s;
    return h;
  } catch (const std::bad_alloc&) {
    if (msg && msgCap) std::snprintf(msg, msgCap, "out of memory while loading the scene");
    return nullptr;
  } catch (...) {
    if (msg && msgCap) std::snprintf(msg, msgCap, "internal error while loading the scene");
    return nullptr;
  }
}
