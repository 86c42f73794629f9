// bmfr_oracle.cpp — CPU restatement of the reference's BMFR denoise pass.  TEST INFRASTRUCTURE ONLY
// (same rules as bdpt_oracle.cpp: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
// load it; the product never does).
//
// Follows, line for line where the arithmetic matters:
//   BidirectionalPathtracing/Passes/DenoisePass.cpp:146-279   execute(): stage order, blits, constants
//   BidirectionalPathtracing/Data/preprocess.ps.hlsl:33-165   temporal reprojection of the noisy frame
//   BidirectionalPathtracing/Data/regressionCP.hlsl:100-500   blockwise feature regression (Householder QR)
//   BidirectionalPathtracing/Data/postprocess.ps.hlsl:22-91   temporal accumulation of the filtered frame
// PARITY UNPINNED: the reference ships no images or vectors for this pass and its shaders cannot be
// compiled here; the compute shader is simulated "thread" by thread between its group barriers, with
// its reductions in the shader's own pairing order, under the arithmetic contract of bdpt_oracle.cpp
// (IEEE fp32, no FMA contraction, left-to-right).  Defined-away undefined behaviour:
//   * regressionCP.hlsl reads gCurNoisy while other groups write it (mirrored border pixels): reads
//     come from the copy DenoisePass.cpp:180 blits to BMFR_PrevNoisy just before the dispatch;
//   * out-of-range texel reads return 0 (D3D semantics), out-of-range writes are dropped;
//   * mul(float4(p,1), prevViewProjMat) is evaluated as ((m0*x + m1*y) + m2*z) + m3.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/bdpt.h"

extern "C" void oracle_half_round(const float* in, uint32_t n, float* out);

namespace {

constexpr int kBufferCount = 13, kFeatures = 10, kFeaturesNotScaled = 4, kBlockPixels = 1024, kLocal = 256, kBlockEdge = 32;
constexpr int kSub = kBlockPixels / kLocal;
const int kBlockOffsets[16][2] = {{-30, -30}, {-12, -22}, {-24, -2}, {-8, -16}, {-26, -24}, {-14, -4}, {-4, -28}, {-26, -16},
                                  {-4, -2},   {-24, -32}, {-10, -10}, {-18, -18}, {-12, -30}, {-32, -4}, {-2, -20}, {-22, -12}};

inline int mirror(int index, int size) {
  if (index < 0)
    index = std::abs(index) - 1;
  else if (index >= size)
    index = 2 * size - index - 1;
  return index;
}
inline float hashRandom(uint32_t a) {  // regressionCP.hlsl:75-84
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return (float)a / 4294967296.0f;
}
inline float addRandom(float value, int id, int sub, int featureBuffer, int frame) {  // :86-95
  return value + 0.01f * 2 *
                     (hashRandom((uint32_t)(id + sub * kLocal + featureBuffer * kBlockEdge * kBlockEdge +
                                            frame * kBufferCount * kBlockEdge * kBlockEdge)) -
                      0.5f);
}
// the shader's parallel reduction: v[i] (op)= v[i+128], +64, ... +2, then v[0] (op) v[1]
template <class Op>
inline float treeReduce(float* v, Op op) {
  for (int stride = 128; stride >= 2; stride >>= 1)
    for (int i = 0; i < stride; i++) v[i] = op(v[i], v[i + stride]);
  return op(v[0], v[1]);
}

}  // namespace

struct oracle_bmfr {
  uint32_t W = 0, H = 0;
  std::vector<float> prevPos, prevNorm, prevNoisy, prevFiltered, accumulated, prevPixel;  // float4 x4, float4, float2
  std::vector<uint32_t> accept;
};

extern "C" {

oracle_bmfr* oracle_bmfr_create(uint32_t w, uint32_t h) {
  oracle_bmfr* b = new oracle_bmfr();
  b->W = w;
  b->H = h;
  const size_t n = (size_t)w * h;
  b->prevPos.assign(n * 4, 0.0f);
  b->prevNorm.assign(n * 4, 0.0f);
  b->prevNoisy.assign(n * 4, 0.0f);
  b->prevFiltered.assign(n * 4, 0.0f);
  b->accumulated.assign(n * 4, 0.0f);
  b->prevPixel.assign(n * 2, 0.0f);
  b->accept.assign(n, 0u);
  return b;
}
void oracle_bmfr_destroy(oracle_bmfr* b) { delete b; }
void oracle_bmfr_reset(oracle_bmfr* b) {
  if (!b) return;
  for (auto* v : {&b->prevPos, &b->prevNorm, &b->prevNoisy, &b->prevFiltered, &b->accumulated, &b->prevPixel}) std::fill(v->begin(), v->end(), 0.0f);
  std::fill(b->accept.begin(), b->accept.end(), 0u);
}

// preprocess.ps.hlsl:33-165
static void preprocess(oracle_bmfr& B, const bdpt_bmfr_params& P, const float* curPos, const float* curNorm, float* noisy) {
  const int W = (int)B.W, H = (int)B.H;
  const bool full = (P.flags & BDPT_BMFR_FULL_FRAME) != 0;
  const float* m = P.prevViewProj;
  std::vector<float> out((size_t)W * H * 4);
  std::memcpy(out.data(), noisy, out.size() * 4);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t i = (size_t)y * W + x;
      const float posx = (float)x + 0.5f, posy = (float)y + 0.5f;
      const float texCx = posx / (float)W;
      if (!full && texCx > 0.5f) continue;  // "Denoise only half image for comparison"
      const float wx = curPos[i * 4], wy = curPos[i * 4 + 1], wz = curPos[i * 4 + 2];
      const float nx = curNorm[i * 4], ny = curNorm[i * 4 + 1], nz = curNorm[i * 4 + 2];
      const float cr = noisy[i * 4], cg = noisy[i * 4 + 1], cb = noisy[i * 4 + 2];
      float pfx = posx, pfy = posy;
      uint32_t storeAccept = 0;
      float blendAlpha = 1.0f;
      float pr = 0, pg = 0, pb = 0, sampleSpp = 0, totalWeight = 0;
      if (P.frameNumber > 0) {
        float c[4];
        for (int r = 0; r < 4; r++) c[r] = ((m[4 * r] * wx + m[4 * r + 1] * wy) + m[4 * r + 2] * wz) + m[4 * r + 3];
        float ux = c[0] / c[3], uy = c[1] / c[3];
        ux = (ux + 1.0f) / 2.0f;
        uy = (1 - uy) / 2.0f;
        if (ux > 1.0f || ux < 0.0f || uy > 1.0f || uy < 0.0f) {
          out[i * 4 + 3] = 1.0f;
          B.accept[i] = 0;
          continue;
        }
        pfx = ux * (float)(uint32_t)W - 0.5f;
        pfy = uy * (float)(uint32_t)H - 0.5f;
        const int ipx = (int)pfx, ipy = (int)pfy;
        const float fx = pfx - (float)ipx, fy = pfy - (float)ipy;
        const float ox = 1.0f - fx, oy = 1.0f - fy;
        const float wts[4] = {ox * oy, fx * oy, ox * fy, fx * fy};
        const int offs[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}};
        for (int k = 0; k < 4; k++) {
          const int sx = ipx + offs[k][0], sy = ipy + offs[k][1];
          if (sx >= 0 && sy >= 0 && sx < W && sy < H) {
            const size_t j = (size_t)sy * W + sx;
            const float dx = B.prevPos[j * 4] - wx, dy = B.prevPos[j * 4 + 1] - wy, dz = B.prevPos[j * 4 + 2] - wz;
            const float pd = (dx * dx + dy * dy) + dz * dz;
            if (pd < 0.01f) {
              const float ex = B.prevNorm[j * 4] - nx, ey = B.prevNorm[j * 4 + 1] - ny, ez = B.prevNorm[j * 4 + 2] - nz;
              const float nd = (ex * ex + ey * ey) + ez * ez;
              if (nd < 1.0f) {
                storeAccept |= 1u << k;
                const float* pd4 = &B.prevNoisy[j * 4];
                sampleSpp += wts[k] * pd4[3];
                pr += wts[k] * pd4[0];
                pg += wts[k] * pd4[1];
                pb += wts[k] * pd4[2];
                totalWeight += wts[k];
              }
            }
          }
        }
        if (totalWeight > 0.0f) {
          pr /= totalWeight;
          pg /= totalWeight;
          pb /= totalWeight;
          sampleSpp /= totalWeight;
          blendAlpha = 1.0f / (sampleSpp + 1.0f);
          blendAlpha = std::max(blendAlpha, 0.2f);
        }
      }
      float newSpp = 1.0f;
      if (blendAlpha < 1.0f) newSpp += sampleSpp;
      out[i * 4] = blendAlpha * cr + (1.0f - blendAlpha) * pr;
      out[i * 4 + 1] = blendAlpha * cg + (1.0f - blendAlpha) * pg;
      out[i * 4 + 2] = blendAlpha * cb + (1.0f - blendAlpha) * pb;
      out[i * 4 + 3] = newSpp;
      B.accept[i] = storeAccept;
      const float pf[2] = {pfx, pfy};
      oracle_half_round(pf, 2, &B.prevPixel[i * 2]);  // BMFR_PrevFramePixel is RG16Float (DenoisePass.cpp:94)
    }
  std::memcpy(noisy, out.data(), out.size() * 4);
}

// regressionCP.hlsl:100-500 for one work group
static void fitBlock(const oracle_bmfr& B, const bdpt_bmfr_params& P, int group, int horizontalBlocks, const float* curPos, const float* curNorm,
                     const float* albedo, const float* noisyIn, float* noisyOut) {
  const int W = (int)B.W, H = (int)B.H;
  const int frame = (int)P.frameNumber;
  const bool ignoreLD = !(P.flags & BDPT_BMFR_KEEP_LD_FEATURES);
  static thread_local std::vector<float> tmpV, outV;
  tmpV.assign((size_t)kBufferCount * kBlockPixels, 0.0f);
  outV.assign((size_t)kBufferCount * kBlockPixels, 0.0f);
  auto tmp = [&](int index, int buf) -> float& { return tmpV[(size_t)buf * kBlockPixels + index]; };
  auto out = [&](int index, int buf) -> float& { return outV[(size_t)buf * kBlockPixels + index]; };
  auto pixelOf = [&](int index, int& ux, int& uy) {
    ux = (group % horizontalBlocks) * kBlockEdge + index % kBlockEdge + kBlockOffsets[frame % 16][0];
    uy = (group / horizontalBlocks) * kBlockEdge + index / kBlockEdge + kBlockOffsets[frame % 16][1];
  };
  for (int index = 0; index < kBlockPixels; index++) {
    int ux, uy;
    pixelOf(index, ux, uy);
    ux = mirror(ux, W);
    uy = mirror(uy, H);
    tmp(index, 0) = 1.0f;
    if (ux < 0 || uy < 0 || ux >= W || uy >= H) {
      // one reflection is not enough when the frame is narrower than the block offset: the shader then loads
      // outside the texture, which returns 0 in D3D
      for (int k = 1; k < kBufferCount; k++) tmp(index, k) = 0.0f;
      continue;
    }
    const size_t i = (size_t)uy * W + ux;
    tmp(index, 1) = curNorm[i * 4];
    tmp(index, 2) = curNorm[i * 4 + 1];
    tmp(index, 3) = curNorm[i * 4 + 2];
    tmp(index, 4) = curPos[i * 4];
    tmp(index, 5) = curPos[i * 4 + 1];
    tmp(index, 6) = curPos[i * 4 + 2];
    tmp(index, 7) = curPos[i * 4] * curPos[i * 4];
    tmp(index, 8) = curPos[i * 4 + 1] * curPos[i * 4 + 1];
    tmp(index, 9) = curPos[i * 4 + 2] * curPos[i * 4 + 2];
    for (int c = 0; c < 3; c++) tmp(index, 10 + c) = albedo[i * 4 + c] < 0.01f ? 0.0f : noisyIn[i * 4 + c] / albedo[i * 4 + c];
  }
  float sumVec[kLocal];
  for (int fb = kFeaturesNotScaled; fb < kFeatures; fb++) {
    for (int t = 0; t < kLocal; t++) {
      float mx = tmp(t, fb);
      for (int s = 1; s < kSub; s++) mx = std::max(tmp(s * kLocal + t, fb), mx);
      sumVec[t] = mx;
    }
    const float blockMax = treeReduce(sumVec, [](float a, float b) { return std::max(a, b); });
    for (int t = 0; t < kLocal; t++) {
      float mn = tmp(t, fb);
      for (int s = 1; s < kSub; s++) mn = std::min(tmp(s * kLocal + t, fb), mn);
      sumVec[t] = mn;
    }
    const float blockMin = treeReduce(sumVec, [](float a, float b) { return std::min(a, b); });
    for (int index = 0; index < kBlockPixels; index++) {
      const float v = (blockMax - blockMin > 1.0f) ? (tmp(index, fb) - blockMin) / (blockMax - blockMin) : tmp(index, fb) - blockMin;
      out(index, fb) = v;
      tmp(index, fb) = v;
    }
  }
  for (int fb = kFeatures; fb < kBufferCount; fb++)
    for (int index = 0; index < kBlockPixels; index++) out(index, fb) = tmp(index, fb);
  for (int fb = 0; fb < kFeaturesNotScaled; fb++)
    for (int index = 0; index < kBlockPixels; index++) out(index, fb) = tmp(index, fb);

  float rmat[kFeatures][kBufferCount];
  std::memset(rmat, 0, sizeof(rmat));
  float uVec[kBlockPixels];
  float uLengthSquared = 0.0f, vecLength = 0.0f, dotV = 0.0f;
  auto sumReduce = [&]() { return treeReduce(sumVec, [](float a, float b) { return a + b; }); };
  int limit = 0;
  if (ignoreLD) {
    for (int col = 0; col < kFeatures; col++) {
      for (int t = 0; t < kLocal; t++) {
        float acc = 0;
        for (int s = 0; s < kSub; s++) {
          const int index = s * kLocal + t;
          const float v = out(index, col);
          uVec[index] = v;
          if (index >= limit + 1) acc += v * v;
        }
        sumVec[t] = acc;
      }
      vecLength = sumReduce();
      float rValue[kLocal];
      for (int t = 0; t < kLocal; t++) {
        if (t < limit) {
          rValue[t] = uVec[t];
        } else if (t == limit) {
          uLengthSquared = vecLength;
          vecLength = std::sqrt(vecLength + uVec[limit] * uVec[limit]);
          uVec[limit] -= vecLength;
          uLengthSquared += uVec[limit] * uVec[limit];
          rValue[t] = vecLength;
        } else {
          rValue[t] = 0;
        }
      }
      if (vecLength > 0.01f) {
        limit++;
        for (int t = 0; t < kFeatures; t++) rmat[t][col] = rValue[t];
      } else {
        for (int t = 0; t < kFeatures; t++) rmat[t][col] = 0.0f;
        continue;
      }
      if (uLengthSquared < 0.001f) continue;
      for (int fb = col + 1; fb < kBufferCount; fb++) {
        for (int t = 0; t < kLocal; t++) {
          float acc = 0.0f;
          for (int s = 0; s < kSub; s++) {
            const int index = s * kLocal + t;
            if (index >= limit - 1) acc += out(index, fb) * uVec[index];
          }
          sumVec[t] = acc;
        }
        dotV = sumReduce();
        for (int index = limit - 1 < 0 ? 0 : limit - 1; index < kBlockPixels; index++)
          out(index, fb) = out(index, fb) - 2.0f * uVec[index] * dotV / uLengthSquared;
      }
    }
    for (int t = 0; t < kFeatures; t++) {
      rmat[t][kFeatures] = out(t, kFeatures);
      rmat[t][kBufferCount - 2] = out(t, kBufferCount - 2);
      rmat[t][kBufferCount - 1] = out(t, kBufferCount - 1);
    }
    limit--;
    for (int i = kBufferCount - 4; i >= 0; i--) {
      if (rmat[limit][i] != 0.0f) {
        for (int t = 0; t < 3; t++) rmat[i][kBufferCount - t - 1] = rmat[limit][kBufferCount - t - 1] / rmat[limit][i];
        limit--;
      } else {
        for (int t = 0; t < 3; t++) rmat[i][kBufferCount - t - 1] = 0.0f;
      }
      for (int t = 0; t < 3 * limit + 3; t++) {
        const int rowId = limit - t / 3;
        const int channel = kBufferCount - (t % 3) - 1;
        rmat[rowId][channel] -= rmat[i][channel] * rmat[rowId][i];
      }
    }
  } else {
    for (int col = 0; col < kFeatures; col++) {
      for (int t = 0; t < kLocal; t++) {
        float acc = 0;
        for (int s = 0; s < kSub; s++) {
          const int index = s * kLocal + t;
          const float v = out(index, col);
          uVec[index] = v;
          if (index >= col + 1) acc += v * v;
        }
        sumVec[t] = acc;
      }
      vecLength = sumReduce();
      for (int t = 0; t < kFeatures; t++) {
        float r;
        if (t < col) {
          r = uVec[t];
        } else if (t == col) {
          uLengthSquared = vecLength;
          vecLength = std::sqrt(vecLength + uVec[col] * uVec[col]);
          uVec[col] -= vecLength;
          uLengthSquared += uVec[col] * uVec[col];
          r = vecLength;
        } else {
          r = 0;
        }
        rmat[t][col] = r;
      }
      for (int fb = col + 1; fb < kBufferCount; fb++) {
        static thread_local std::vector<float> cache;
        cache.assign(kBlockPixels, 0.0f);
        for (int t = 0; t < kLocal; t++) {
          float acc = 0.0f;
          for (int s = 0; s < kSub; s++) {
            const int index = s * kLocal + t;
            if (index >= col) {
              float v = out(index, fb);
              if (col == 0 && fb < kFeatures) v = addRandom(v, t, s, fb, frame);
              cache[index] = v;
              acc += v * uVec[index];
            }
          }
          sumVec[t] = acc;
        }
        dotV = sumReduce();
        for (int index = col; index < kBlockPixels; index++) out(index, fb) = cache[index] - 2.0f * uVec[index] * dotV / uLengthSquared;
      }
    }
    for (int t = 0; t < kFeatures; t++) {
      rmat[t][kFeatures] = out(t, kFeatures);
      rmat[t][kBufferCount - 2] = out(t, kBufferCount - 2);
      rmat[t][kBufferCount - 1] = out(t, kBufferCount - 1);
    }
    for (int i = kBufferCount - 4; i >= 0; i--) {
      for (int t = 0; t < 3; t++) rmat[i][kBufferCount - t - 1] /= rmat[i][i];
      for (int t = 0; t < 3 * i; t++) {
        const int rowId = i - t / 3 - 1;
        const int channel = kBufferCount - (t % 3) - 1;
        rmat[rowId][channel] -= rmat[i][channel] * rmat[rowId][i];
      }
    }
  }
  for (int index = 0; index < kBlockPixels; index++) {
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (int col = 0; col < kFeatures; col++) {
      const float t = tmp(index, col);
      r += rmat[col][kFeatures] * t;
      g += rmat[col][kFeatures + 1] * t;
      b += rmat[col][kFeatures + 2] * t;
    }
    int ux, uy;
    pixelOf(index, ux, uy);
    if (ux < 0 || uy < 0 || ux >= W || uy >= H) continue;
    const size_t i = (size_t)uy * W + ux;
    noisyOut[i * 4] = albedo[i * 4] * (r < 0.0f ? 0.0f : r);
    noisyOut[i * 4 + 1] = albedo[i * 4 + 1] * (g < 0.0f ? 0.0f : g);
    noisyOut[i * 4 + 2] = albedo[i * 4 + 2] * (b < 0.0f ? 0.0f : b);
    noisyOut[i * 4 + 3] = albedo[i * 4 + 3] * noisyIn[i * 4 + 3];
  }
}

// postprocess.ps.hlsl:22-91
static void postprocess(oracle_bmfr& B, const bdpt_bmfr_params& P, const float* filtered) {
  const int W = (int)B.W, H = (int)B.H;
  const bool full = (P.flags & BDPT_BMFR_FULL_FRAME) != 0;
  auto prevAt = [&](int x, int y, int c) -> float { return (x >= 0 && y >= 0 && x < W && y < H) ? B.prevFiltered[((size_t)y * W + x) * 4 + c] : 0.0f; };
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      const size_t i = (size_t)y * W + x;
      const float texCx = ((float)x + 0.5f) / (float)W;
      if (!full && texCx > 0.5f) {
        for (int c = 0; c < 4; c++) B.accumulated[i * 4 + c] = filtered[i * 4 + c];
        continue;
      }
      const float spp = filtered[i * 4 + 3];
      float prev[3] = {0, 0, 0};
      float blendAlpha = 1.0f;
      if (P.frameNumber > 0) {
        const uint32_t accept = B.accept[i];
        if (accept > 0) {
          const float pfx = B.prevPixel[i * 2], pfy = B.prevPixel[i * 2 + 1];
          const int ipx = (int)pfx, ipy = (int)pfy;
          const float fx = pfx - (float)ipx, fy = pfy - (float)ipy;
          const float ox = 1.0f - fx, oy = 1.0f - fy;
          float totalWeight = 0.0f;
          const float wts[4] = {ox * oy, fx * oy, ox * fy, fx * fy};
          const int offs[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}};
          for (int k = 0; k < 4; k++)
            if (accept & (1u << k)) {
              totalWeight += wts[k];
              for (int c = 0; c < 3; c++) prev[c] += wts[k] * prevAt(ipx + offs[k][0], ipy + offs[k][1], c);
            }
          if (totalWeight > 0.0f) {
            blendAlpha = 1.0f / spp;
            blendAlpha = std::max(blendAlpha, 0.1f);
            for (int c = 0; c < 3; c++) prev[c] /= totalWeight;
          }
        }
      }
      for (int c = 0; c < 3; c++) B.accumulated[i * 4 + c] = blendAlpha * filtered[i * 4 + c] + (1.0f - blendAlpha) * prev[c];
      B.accumulated[i * 4 + 3] = 1.0f;
    }
}

// DenoisePass.cpp:146-204
int oracle_bmfr_execute(oracle_bmfr* b, const bdpt_bmfr_params* p, const float* curPos, const float* curNorm, const float* albedo, float* noisy) {
  if (!b || !p || !curPos || !curNorm || !albedo || !noisy) return -1;
  const size_t n4 = (size_t)b->W * b->H * 4;
  if (p->flags & BDPT_BMFR_PREPROCESS) preprocess(*b, *p, curPos, curNorm, noisy);
  std::memcpy(b->prevNoisy.data(), noisy, n4 * 4);
  std::memcpy(b->prevNorm.data(), curNorm, n4 * 4);
  std::memcpy(b->prevPos.data(), curPos, n4 * 4);
  if (p->flags & BDPT_BMFR_REGRESSION) {
    const int bw = ((int)b->W + 31) / 32, bh = ((int)b->H + 31) / 32;
    int w = bw + 1;
    const int h = bh + 1;
    if (!(p->flags & BDPT_BMFR_FULL_FRAME)) w /= 2;  // DenoisePass.cpp:262 (half image)
    for (int g = 0; g < w * h; g++) fitBlock(*b, *p, g, w, curPos, curNorm, albedo, b->prevNoisy.data(), noisy);
  }
  if (p->flags & BDPT_BMFR_POSTPROCESS) {
    postprocess(*b, *p, noisy);
    std::memcpy(noisy, b->accumulated.data(), n4 * 4);
    std::memcpy(b->prevFiltered.data(), b->accumulated.data(), n4 * 4);
  }
  return 0;
}

}  // extern "C"
