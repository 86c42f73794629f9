// CPU-only sanitizer run (AddressSanitizer + UBSan; GPU sanitizers are not available on the pool): the host-side
// C++ that never touches the device — scene factories, scene loader, BVH builder — and the oracle, driven the way
// the tests drive them.  Built and run by tests/test_sanitizers.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../fyp-bidirectionalpathtracer_amd/csrc/bvh.h"
#include "../../include/bdpt.h"
#include "../../include/bdpt_scene.h"
#include "../../oracle/bdpt_oracle.h"

static int renderWith(const bdpt_scene_desc& d, const bdpt_camera& cam, uint32_t W, uint32_t H, uint32_t depth, uint32_t mat, uint32_t flags) {
  oracle_scene* s = oracle_scene_create(&d);
  if (!s) return 1;
  const size_t n = (size_t)W * H;
  std::vector<float> pos(n * 4), nrm(n * 4), dif(n * 4), spec(n * 4), extra(n * 4), emis(n * 4), out(n * 4);
  std::vector<uint64_t> splat(n * 4);
  oracle_frame f{};
  f.width = W;
  f.height = H;
  f.y0 = 0;
  f.y1 = H;
  f.worldPosition = pos.data();
  f.worldNormal = nrm.data();
  f.materialDiffuse = dif.data();
  f.materialSpecRough = spec.data();
  f.materialExtra = extra.data();
  f.emissive = emis.data();
  f.out = out.data();
  f.splat = splat.data();
  bdpt_gbuffer_params gp{};
  gp.pixelJitter[0] = gp.pixelJitter[1] = 0.5f;
  gp.focalLen = 1.0f;
  gp.frameCount = 0xdeadbeefu;
  gp.envColor[0] = gp.envColor[1] = 0.5f;
  gp.envColor[2] = 0.8f;
  gp.envColor[3] = 1.0f;
  bdpt_params p{};
  p.minT = 1e-4f;
  p.frameCount = 0x1337;
  p.matIndex = mat;
  p.refractiveIndex = 1.0f;
  p.maxDepth = depth;
  p.emitMult = 1.0f;
  p.clampUpper = 0.9f;
  p.pixelJitter[0] = p.pixelJitter[1] = 0.5f;
  p.flags = flags;
  bdpt_counters cnt{};
  int rc = oracle_gbuffer(s, &cam, &gp, nullptr, &f, 0, 2);
  rc |= oracle_bdpt(s, &cam, &p, &f, 0, 2, &cnt);
  rc |= oracle_resolve(&f);
  // BMFR on top, three frames, regression on
  oracle_bmfr* b = oracle_bmfr_create(W, H);
  for (uint32_t k = 0; k < 3; k++) {
    bdpt_bmfr_params bp{};
    bp.frameNumber = k;
    bp.flags = BDPT_BMFR_PREPROCESS | BDPT_BMFR_REGRESSION | BDPT_BMFR_POSTPROCESS | (k == 2 ? BDPT_BMFR_KEEP_LD_FEATURES | BDPT_BMFR_FULL_FRAME : 0u);
    for (int i = 0; i < 4; i++) bp.prevViewProj[i * 5] = 1.0f;
    rc |= oracle_bmfr_execute(b, &bp, pos.data(), nrm.data(), dif.data(), out.data());
  }
  oracle_bmfr_destroy(b);
  oracle_scene_destroy(s);
  double sum = 0;
  for (float v : out) sum += v;
  std::printf("  %ux%u depth %u mat %u: rc=%d rays=%llu sum=%.4f\n", W, H, depth, mat, rc, (unsigned long long)(cnt.raysNee + cnt.raysConnect + cnt.raysSplat), sum);
  return rc;
}

int main(int argc, char** argv) {
  const std::string tmp = argc > 1 ? argv[1] : "/tmp";
  int rc = 0;
  for (int which = 0; which < 3; which++) {
    bdpt_scene* sc = which == 0 ? bdpt_scene_create_cornell() : which == 1 ? bdpt_scene_create_atrium(3, 6000) : bdpt_scene_create_soup(5, 700, 0.2f);
    bdpt_scene_desc d{};
    bdpt_camera cam{};
    rc |= bdpt_scene_get_desc(sc, &d);
    rc |= bdpt_scene_get_camera(sc, 1.5f, &cam);
    bdpt::Bvh bvh;
    bdpt::buildBvh(d.positions, d.indices, d.numTriangles, nullptr, bvh);
    std::printf("scene %d: %u triangles, %zu nodes, stack %u\n", which, d.numTriangles, bvh.nodes.size(), bvh.maxStack);
    if (which < 2) {
      rc |= renderWith(d, cam, 36, 24, 4, 0, 0);
      rc |= renderWith(d, cam, 24, 16, 8, 1, BDPT_PARAM_MIS_POWER);
    }
    bdpt_scene_destroy(sc);
  }
  {  // the full scene build (traversal flags, alpha classification, pre-splitting with clipped pieces) and the host trace hook
    bdpt_scene* sc = bdpt_scene_create_courtyard(2, 9000, 0.6f);
    bdpt_scene_desc d{};
    rc |= bdpt_scene_get_desc(sc, &d);
    for (int variant = 0; variant < 3; variant++) {
      bdpt_bvh_info info{};
      void* h = bdpt_host_bvh_create(&d, 2, variant == 0 ? 0.0f : (variant == 1 ? -1.0f : 2.0f), variant == 0 ? 0.0f : (variant == 1 ? -1.0f : 12.0f),
                                     variant != 0, &info);
      if (!h) {
        rc |= 1;
        continue;
      }
      std::vector<float> rays;
      uint32_t st = 99u;
      auto u01 = [&]() { return (float)((st = st * 1664525u + 1013904223u) >> 8) / 16777216.0f; };
      for (int i = 0; i < 600; i++) {
        const float o[3] = {-13.0f + 26.0f * u01(), 0.2f + 7.0f * u01(), -3.0f + 6.0f * u01()};
        float dd[3] = {u01() - 0.5f, u01() - 0.5f, u01() - 0.5f};
        const float l = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]) + 1e-9f;
        rays.insert(rays.end(), {o[0], o[1], o[2], dd[0] / l, dd[1] / l, dd[2] / l, 1e-4f, 1e38f});
      }
      std::vector<int32_t> prim(600), primScan(600);
      std::vector<float> tuv(1800), tuvScan(1800);
      uint64_t visits[2] = {0, 0};
      for (int mode = 0; mode < 3; mode++) {
        rc |= bdpt_host_bvh_trace(h, rays.data(), 600, mode, 0, 2, prim.data(), tuv.data(), visits);
        rc |= bdpt_host_bvh_trace(h, rays.data(), 600, mode, 1, 2, primScan.data(), tuvScan.data(), nullptr);
        for (int i = 0; i < 600; i++)
          if ((mode == 2 ? (prim[i] >= 0) != (primScan[i] >= 0) : prim[i] != primScan[i]) || tuv[(size_t)i * 3] != tuvScan[(size_t)i * 3]) rc |= 1;
      }
      std::printf("courtyard variant %d: %u references for %u triangles, %u dropped, %u always pass; %llu node visits\n", variant, info.numReferences,
                  info.numTriangles, info.numDropped, info.numAlwaysPass, (unsigned long long)visits[0]);
      bdpt_host_bvh_destroy(h);
    }
    char msg[256] = {0};
    bdpt_bvh_info info{};
    if (bdpt_bvh_build_check(&d, &info, msg, sizeof(msg)) != BDPT_OK) {
      std::printf("build check: %s\n", msg);
      rc |= 1;
    }
    bdpt_scene_destroy(sc);
  }
  {  // scene loader on a small OBJ + MTL + PPM
    std::ofstream(tmp + "/san.mtl") << "newmtl a\nKd 0.5 0.6 0.7\nKs 0.1 0.1 0.1\nNs 0.4\nmap_Kd san.ppm\nnewmtl b.DoubleSided\nKe 1 1 1\n";
    {
      std::ofstream ppm(tmp + "/san.ppm", std::ios::binary);
      ppm << "P6\n2 2\n255\n";
      const unsigned char px[12] = {255, 0, 0, 0, 255, 0, 0, 0, 255, 255, 255, 255};
      ppm.write(reinterpret_cast<const char*>(px), 12);
    }
    std::ofstream(tmp + "/san.obj") << "mtllib san.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nusemtl a\nf 1/1 2/2 3/3 4/4\n"
                                      "usemtl b.DoubleSided\nf -1 -4 -3\nusemtl nope\nf 1 5 2\n";
    std::ofstream(tmp + "/san.fscene") << "{ \"models\": [ { \"file\": \"san.obj\", \"instances\": [ { \"translation\": [1,2,3], \"rotation\": [10,20,30], \"scaling\": [1,2,1] } ] } ],"
                                         " \"lights\": [ { \"type\": \"point_light\", \"pos\": [0,3,0], \"direction\": [0,-1,0], \"intensity\": [5,5,5], \"opening_angle\": 45.0, \"penumbra_angle\": 3.0 } ],"
                                         " \"cameras\": [ { \"name\": \"c\", \"pos\": [0,1,5], \"target\": [0,0,0], \"up\": [0,1,0], \"focal_length\": 30.0 } ] }";
    char msg[256] = {0};
    bdpt_scene* sc = bdpt_scene_load((tmp + "/san.fscene").c_str(), msg, sizeof(msg));
    if (!sc) {
      std::printf("loader failed: %s\n", msg);
      rc |= 1;
    } else {
      bdpt_scene_desc d{};
      bdpt_camera cam{};
      bdpt_scene_get_desc(sc, &d);
      bdpt_scene_get_camera(sc, 1.0f, &cam);
      std::printf("loaded: %u triangles %u materials %u textures %u lights\n", d.numTriangles, d.numMaterials, d.numTextures, d.numLights);
      rc |= renderWith(d, cam, 16, 16, 3, 0, 0);
      bdpt_scene_destroy(sc);
    }
    bdpt_scene* bad = bdpt_scene_load((tmp + "/missing.fscene").c_str(), msg, sizeof(msg));
    if (bad) rc |= 1;
  }
  {  // the threaded OBJ reader on mutated and truncated model text (host/SceneLoader.cpp parses raw memory with memchr /
     // from_chars): a corrupt file may be refused or load as something else, never read or write out of bounds; and an
     // intact one gives the same scene for 1, 3 and 8 threads
    std::string good = "mtllib san.mtl\n# comment\nv 0 0 0\nv 1 0 0\r\nv 1 1 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nusemtl a\n"
                       "f 1/1/1 2/2/1 3/3/1 4/4/1\nusemtl b.DoubleSided\nf -1 -4 -3\nf 1//1 5//1 2//1\ng grp\ns off\nusemtl nope\nf 1 5 2 3 4\nf +1 2 3\n";
    for (int k = 0; k < 400; k++) good += "v " + std::to_string(k % 7) + ".5 " + std::to_string(k % 5) + " -" + std::to_string(k % 3) + "e-1\nf -1 -2 -3\n";
    uint32_t state = 777u;
    auto rnd = [&]() { return state = state * 1664525u + 1013904223u; };
    int loaded = 0, refused = 0;
    uint32_t trisGood[3] = {0, 0, 0};
    for (int k = 0; k < 600; k++) {
      std::string d = good;
      if (k >= 3) {
        const int flips = 1 + (int)(rnd() % 6);
        static const char pool[] = "0123456789-+/. \n\rvftnemusl#e";
        for (int i = 0; i < flips; i++) d[rnd() % d.size()] = (rnd() % 3) ? pool[rnd() % (sizeof(pool) - 1)] : (char)(rnd() >> 24);
        if (k % 4 == 0) d.resize(1 + rnd() % d.size());
      }
      std::ofstream(tmp + "/mut.obj", std::ios::binary).write(d.data(), (std::streamsize)d.size());
      bdpt_scene_load_threads(k % 3 == 0 ? 1 : (k % 3 == 1 ? 3 : 8));
      char msg[256] = {0};
      bdpt_scene* sc = bdpt_scene_load((tmp + "/mut.obj").c_str(), msg, sizeof(msg));
      if (sc) {
        bdpt_scene_desc dd{};
        bdpt_scene_get_desc(sc, &dd);
        for (uint32_t t = 0; t < dd.numTriangles * 3; t++)
          if (dd.indices[t] >= dd.numVertices) rc |= 1;  // whatever was loaded is a consistent scene
        if (k < 3) trisGood[k] = dd.numTriangles;
        loaded++;
        bdpt_scene_destroy(sc);
      } else {
        refused++;
        if (k < 3) rc |= 1;  // the unmodified file must load
      }
    }
    bdpt_scene_load_threads(0);
    if (trisGood[0] == 0 || trisGood[0] != trisGood[1] || trisGood[0] != trisGood[2]) rc |= 1;
    std::printf("obj reader: %d loaded, %d refused, %u triangles in the intact file\n", loaded, refused, trisGood[0]);
  }
  {  // image decoders: the files tests/test_sanitizers.py wrote, then byte-mutated and truncated copies of them (a corrupt
     // file may be refused or decoded to garbage, never read or written out of bounds)
    uint32_t state = 12345u;
    auto rnd = [&]() { return state = state * 1664525u + 1013904223u; };
    int decoded = 0, refused = 0;
    for (const char* name : {"san_rgba.png", "san_pal.png", "san_420.jpg", "san_grey.jpg", "san_rst.jpg", "san_probe.hdr"}) {
      const bool hdr = std::strstr(name, ".hdr") != nullptr;
      std::ifstream f(tmp + "/" + name, std::ios::binary);
      if (!f) continue;
      std::vector<char> good((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      for (int k = 0; k < 400; k++) {
        std::vector<char> d = good;
        if (k > 0) {
          const int flips = 1 + (int)(rnd() % 4);
          for (int i = 0; i < flips; i++) d[rnd() % d.size()] = (char)(rnd() >> 24);
          if (k % 5 == 0) d.resize(1 + rnd() % d.size());
        }
        const std::string path = tmp + "/mut.bin";
        std::ofstream(path, std::ios::binary).write(d.data(), (std::streamsize)d.size());
        uint32_t w = 0, h = 0, a = 0;
        char msg[128];
        if (hdr) {  // the light-probe reader (Radiance RGBE)
          if (bdpt_image_load_hdr(path.c_str(), &w, &h, nullptr, 0, msg, sizeof(msg)) == BDPT_OK) {
            std::vector<float> px((size_t)w * h * 4);
            if (bdpt_image_load_hdr(path.c_str(), &w, &h, px.data(), px.size(), msg, sizeof(msg)) != BDPT_OK) rc |= 1;
            decoded++;
          } else {
            refused++;
            if (k == 0) rc |= 1;
          }
          continue;
        }
        if (bdpt_image_load(path.c_str(), &w, &h, &a, nullptr, 0, msg, sizeof(msg)) == BDPT_OK) {
          std::vector<uint8_t> px((size_t)w * h * 4);
          if (bdpt_image_load(path.c_str(), &w, &h, &a, px.data(), px.size(), msg, sizeof(msg)) != BDPT_OK) rc |= 1;
          decoded++;
        } else {
          refused++;
          if (k == 0) rc |= 1;  // the unmodified file must decode
        }
      }
    }
    std::printf("image decoders: %d decoded, %d refused\n", decoded, refused);
  }
  std::printf("sanitizer run finished rc=%d\n", rc);
  return rc;
}
