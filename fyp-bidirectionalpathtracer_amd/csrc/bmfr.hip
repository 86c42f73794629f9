// bmfr.hip — the reference's BMFR denoise pass as three HIP kernels for gfx950.
//
//   bmfr_preprocess_kernel   Data/preprocess.ps.hlsl:33-165 (+ the three history blits of DenoisePass.cpp:180-182,
//                            written into the other half of a ping-pong pair instead of copied afterwards)
//   bmfr_fit_kernel          Data/regressionCP.hlsl:100-500: one 256-thread workgroup per 32x32 block; the 13x1024
//                            working matrix (out_data) lives in LDS (52 KiB, column-major so a thread's four pixels
//                            are conflict-free), the normalised features (tmp_data) in registers; the reference
//                            keeps both in R32Float textures in device memory
//   bmfr_postprocess_kernel  Data/postprocess.ps.hlsl:22-91 (+ the two blits of DenoisePass.cpp:193-194)
//
// Arithmetic contract as everywhere else (DESIGN.md "Numerics"): the shader's own reduction pairing, no FMA
// contraction, correctly rounded divide/sqrt, so the parity tests can compare with the CPU checker bit for bit.
#include <hip/hip_runtime.h>

#include "device_math.hpp"
#include "kernels.h"

namespace bdpt {

#define BD __device__ __forceinline__

namespace {

constexpr int kBufferCount = 13, kFeatures = 10, kFeaturesNotScaled = 4, kBlockPixels = 1024, kLocal = 256, kBlockEdge = 32;
constexpr int kSub = kBlockPixels / kLocal;
__constant__ int kBlockOffsets[16][2] = {{-30, -30}, {-12, -22}, {-24, -2}, {-8, -16}, {-26, -24}, {-14, -4}, {-4, -28}, {-26, -16},
                                         {-4, -2},   {-24, -32}, {-10, -10}, {-18, -18}, {-12, -30}, {-32, -4}, {-2, -20}, {-22, -12}};

BD float4 loadHalf4(const uint16_t* p, size_t i) {
  const uint2 w = reinterpret_cast<const uint2*>(p)[i];
  return make_float4(f16_to_f32((uint16_t)(w.x & 0xffffu)), f16_to_f32((uint16_t)(w.x >> 16)), f16_to_f32((uint16_t)(w.y & 0xffffu)),
                     f16_to_f32((uint16_t)(w.y >> 16)));
}
BD int mirror(int index, int size) {
  if (index < 0)
    index = (index < 0 ? -index : index) - 1;
  else if (index >= size)
    index = 2 * size - index - 1;
  return index;
}
BD float hashRandom(uint32_t a) {
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return (float)a / 4294967296.0f;
}
BD float addRandom(float value, int id, int sub, int featureBuffer, int frame) {
  return value + 0.01f * 2 *
                     (hashRandom((uint32_t)(id + sub * kLocal + featureBuffer * kBlockEdge * kBlockEdge +
                                            frame * kBufferCount * kBlockEdge * kBlockEdge)) -
                      0.5f);
}

}  // namespace

__global__ __launch_bounds__(256) void bmfr_preprocess_kernel(BmfrDev A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t n = A.W * A.H;
  if (i >= n) return;
  const int W = (int)A.W, H = (int)A.H;
  const int x = (int)(i % A.W), y = (int)(i / A.W);
  const float4 cp = A.curPos[i];
  const float4 cn = loadHalf4(A.curNorm, i);
  float4 cur = A.noisy[i];
  const float posx = (float)x + 0.5f, posy = (float)y + 0.5f;
  const float texCx = posx / (float)W;
  const bool process = A.doPre && (A.full || !(texCx > 0.5f));
  if (process) {
    float pfx = posx, pfy = posy;
    uint32_t storeAccept = 0;
    float blendAlpha = 1.0f;
    float pr = 0, pg = 0, pb = 0, sampleSpp = 0, totalWeight = 0;
    bool outside = false;
    if (A.frame > 0) {
      float c[4];
#pragma unroll
      for (int r = 0; r < 4; r++) c[r] = ((A.m[4 * r] * cp.x + A.m[4 * r + 1] * cp.y) + A.m[4 * r + 2] * cp.z) + A.m[4 * r + 3];
      float ux = c[0] / c[3], uy = c[1] / c[3];
      ux = (ux + 1.0f) / 2.0f;
      uy = (1 - uy) / 2.0f;
      if (ux > 1.0f || ux < 0.0f || uy > 1.0f || uy < 0.0f) {
        outside = true;
      } else {
        pfx = ux * (float)A.W - 0.5f;
        pfy = uy * (float)A.H - 0.5f;
        const int ipx = (int)pfx, ipy = (int)pfy;
        const float fx = pfx - (float)ipx, fy = pfy - (float)ipy;
        const float ox = 1.0f - fx, oy = 1.0f - fy;
        const float wts[4] = {ox * oy, fx * oy, ox * fy, fx * fy};
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int sx = ipx + (k & 1), sy = ipy + (k >> 1);
          if (sx >= 0 && sy >= 0 && sx < W && sy < H) {
            const size_t j = (size_t)sy * W + sx;
            const float4 pp = A.prevPosR[j];
            const float dx = pp.x - cp.x, dy = pp.y - cp.y, dz = pp.z - cp.z;
            const float pd = (dx * dx + dy * dy) + dz * dz;
            if (pd < 0.01f) {
              const float4 pn = A.prevNormR[j];
              const float ex = pn.x - cn.x, ey = pn.y - cn.y, ez = pn.z - cn.z;
              const float nd = (ex * ex + ey * ey) + ez * ez;
              if (nd < 1.0f) {
                storeAccept |= 1u << k;
                const float4 pd4 = A.prevNoisyR[j];
                sampleSpp += wts[k] * pd4.w;
                pr += wts[k] * pd4.x;
                pg += wts[k] * pd4.y;
                pb += wts[k] * pd4.z;
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
          blendAlpha = blendAlpha > 0.2f ? blendAlpha : 0.2f;
        }
      }
    }
    if (outside) {
      cur.w = 1.0f;
      A.accept[i] = 0;
    } else {
      float newSpp = 1.0f;
      if (blendAlpha < 1.0f) newSpp += sampleSpp;
      cur = make_float4(blendAlpha * cur.x + (1.0f - blendAlpha) * pr, blendAlpha * cur.y + (1.0f - blendAlpha) * pg,
                        blendAlpha * cur.z + (1.0f - blendAlpha) * pb, newSpp);
      A.accept[i] = (uint8_t)storeAccept;
      A.prevPixel[i] = (uint32_t)f32_to_f16(pfx) | ((uint32_t)f32_to_f16(pfy) << 16);  // RG16Float
    }
    A.noisy[i] = cur;
  }
  // history for the next frame (DenoisePass.cpp:180-182), other half of the ping-pong pair
  A.prevNoisyW[i] = cur;
  A.prevNormW[i] = cn;
  A.prevPosW[i] = cp;
}

// v[i] (op)= v[i+128], +64, ... +2, then v[0] (op) v[1]: the shader's reduction, pairing preserved
template <int OP>
BD float blockTree(float v, float* sumVec, int tid) {
  sumVec[tid] = v;
  __syncthreads();
#pragma unroll
  for (int stride = 128; stride >= 2; stride >>= 1) {
    if (tid < stride) {
      const float a = sumVec[tid], b = sumVec[tid + stride];
      sumVec[tid] = OP == 0 ? a + b : (OP == 1 ? fmaxf(a, b) : fminf(a, b));
    }
    __syncthreads();
  }
  const float a = sumVec[0], b = sumVec[1];
  const float r = OP == 0 ? a + b : (OP == 1 ? fmaxf(a, b) : fminf(a, b));
  __syncthreads();
  return r;
}

template <bool IGNORE_LD>
__global__ __launch_bounds__(256) void bmfr_fit_kernel(BmfrDev A, int horizontalBlocks) {
  __shared__ float outS[kBufferCount * kBlockPixels];  // out_data of this block: [buffer][pixel]
  __shared__ float sumVec[kLocal];
  __shared__ float rmat[kFeatures][kBufferCount];
  __shared__ float bcast[2];
  const int tid = (int)threadIdx.x, group = (int)blockIdx.x;
  const int W = (int)A.W, H = (int)A.H, frame = (int)A.frame;
  const int offx = kBlockOffsets[frame % 16][0], offy = kBlockOffsets[frame % 16][1];
  const int bx = (group % horizontalBlocks) * kBlockEdge + offx, by = (group / horizontalBlocks) * kBlockEdge + offy;
  float tmp[kSub][kFeatures];
  float spp[kSub];
#define OUT(index, buf) outS[(buf) * kBlockPixels + (index)]
#pragma unroll
  for (int s = 0; s < kSub; s++) {
    const int index = s * kLocal + tid;
    const int ux = mirror(bx + index % kBlockEdge, W), uy = mirror(by + index / kBlockEdge, H);
    // a frame narrower than the block offset is not covered by one reflection: such loads fall outside the
    // texture and return 0 in D3D
    const bool inside = ux >= 0 && uy >= 0 && ux < W && uy < H;
    const size_t i = inside ? (size_t)uy * W + ux : 0;
    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float4 p = inside ? A.curPos[i] : zero;
    const float4 nrm = inside ? loadHalf4(A.curNorm, i) : zero;
    const float4 alb = inside ? loadHalf4(A.albedo, i) : zero;
    const float4 c = inside ? A.prevNoisyW[i] : zero;  // the copy of gCurNoisy made just before the dispatch (DenoisePass.cpp:180)
    tmp[s][0] = 1.0f;
    tmp[s][1] = nrm.x;
    tmp[s][2] = nrm.y;
    tmp[s][3] = nrm.z;
    tmp[s][4] = p.x;
    tmp[s][5] = p.y;
    tmp[s][6] = p.z;
    tmp[s][7] = p.x * p.x;
    tmp[s][8] = p.y * p.y;
    tmp[s][9] = p.z * p.z;
    OUT(index, 10) = alb.x < 0.01f ? 0.0f : c.x / alb.x;
    OUT(index, 11) = alb.y < 0.01f ? 0.0f : c.y / alb.y;
    OUT(index, 12) = alb.z < 0.01f ? 0.0f : c.z / alb.z;
    spp[s] = c.w;
  }
  // scale features 4..9 to the block's range (regressionCP.hlsl:122-182)
#pragma unroll
  for (int fb = kFeaturesNotScaled; fb < kFeatures; fb++) {
    float mx = tmp[0][fb], mn = tmp[0][fb];
#pragma unroll
    for (int s = 1; s < kSub; s++) {
      mx = fmaxf(tmp[s][fb], mx);
      mn = fminf(tmp[s][fb], mn);
    }
    const float blockMax = blockTree<1>(mx, sumVec, tid);
    const float blockMin = blockTree<2>(mn, sumVec, tid);
    const bool wide = blockMax - blockMin > 1.0f;
#pragma unroll
    for (int s = 0; s < kSub; s++) tmp[s][fb] = wide ? (tmp[s][fb] - blockMin) / (blockMax - blockMin) : tmp[s][fb] - blockMin;
  }
#pragma unroll
  for (int fb = 0; fb < kFeatures; fb++)
#pragma unroll
    for (int s = 0; s < kSub; s++) OUT(s * kLocal + tid, fb) = tmp[s][fb];
  __syncthreads();

  // Householder QR over the 10 feature columns (regressionCP.hlsl:200-330 / 331-440)
  float u[kSub];
  float uLengthSquared = 0.0f;
  int limit = 0;
  for (int col = 0; col < kFeatures; col++) {
    const int firstRow = IGNORE_LD ? limit + 1 : col + 1;
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < kSub; s++) {
      const int index = s * kLocal + tid;
      const float v = OUT(index, col);
      u[s] = v;
      if (index >= firstRow) acc += v * v;
    }
    float vecLength = blockTree<0>(acc, sumVec, tid);
    const int pivot = IGNORE_LD ? limit : col;
    float rValue = 0.0f;
    if (tid < pivot) {
      rValue = u[0];
    } else if (tid == pivot) {
      float uls = vecLength;
      vecLength = sqrtf(vecLength + u[0] * u[0]);
      u[0] -= vecLength;
      uls += u[0] * u[0];
      rValue = vecLength;
      bcast[0] = vecLength;
      bcast[1] = uls;
    }
    __syncthreads();
    vecLength = bcast[0];
    uLengthSquared = bcast[1];
    if (IGNORE_LD) {
      if (vecLength > 0.01f) {
        limit++;
        if (tid < kFeatures) rmat[tid][col] = rValue;
      } else {
        if (tid < kFeatures) rmat[tid][col] = 0.0f;
        __syncthreads();  // bcast is rewritten by the next column's pivot thread
        continue;
      }
      if (uLengthSquared < 0.001f) {
        __syncthreads();
        continue;
      }
    } else {
      if (tid < kFeatures) rmat[tid][col] = rValue;
    }
    const int firstUpd = IGNORE_LD ? limit - 1 : col;
    for (int fb = col + 1; fb < kBufferCount; fb++) {
      float cache[kSub];
      float dot = 0.0f;
#pragma unroll
      for (int s = 0; s < kSub; s++) {
        const int index = s * kLocal + tid;
        if (index >= firstUpd) {
          float v = OUT(index, fb);
          if (!IGNORE_LD && col == 0 && fb < kFeatures) v = addRandom(v, tid, s, fb, frame);
          cache[s] = v;
          dot += v * u[s];
        }
      }
      const float dotV = blockTree<0>(dot, sumVec, tid);
#pragma unroll
      for (int s = 0; s < kSub; s++) {
        const int index = s * kLocal + tid;
        if (index >= firstUpd) OUT(index, fb) = cache[s] - 2.0f * u[s] * dotV / uLengthSquared;
      }
    }
    __syncthreads();
  }
  if (tid < kFeatures) {
    rmat[tid][kFeatures] = OUT(tid, kFeatures);
    rmat[tid][kBufferCount - 2] = OUT(tid, kBufferCount - 2);
    rmat[tid][kBufferCount - 1] = OUT(tid, kBufferCount - 1);
  }
  __syncthreads();
  // back substitution (regressionCP.hlsl:332-352 / 441-455): 10x3 unknowns, done by one lane in the order the
  // shader's barriers impose
  if (tid == 0) {
    if (IGNORE_LD) {
      int lim = limit - 1;
      for (int i = kBufferCount - 4; i >= 0; i--) {
        if (rmat[lim][i] != 0.0f) {
          for (int t = 0; t < 3; t++) rmat[i][kBufferCount - t - 1] = rmat[lim][kBufferCount - t - 1] / rmat[lim][i];
          lim--;
        } else {
          for (int t = 0; t < 3; t++) rmat[i][kBufferCount - t - 1] = 0.0f;
        }
        for (int t = 0; t < 3 * lim + 3; t++) {
          const int rowId = lim - t / 3;
          const int channel = kBufferCount - (t % 3) - 1;
          rmat[rowId][channel] -= rmat[i][channel] * rmat[rowId][i];
        }
      }
    } else {
      for (int i = kBufferCount - 4; i >= 0; i--) {
        for (int t = 0; t < 3; t++) rmat[i][kBufferCount - t - 1] /= rmat[i][i];
        for (int t = 0; t < 3 * i; t++) {
          const int rowId = i - t / 3 - 1;
          const int channel = kBufferCount - (t % 3) - 1;
          rmat[rowId][channel] -= rmat[i][channel] * rmat[rowId][i];
        }
      }
    }
  }
  __syncthreads();
  // filtered colour = features . weights, re-modulated by albedo (regressionCP.hlsl:458-500)
#pragma unroll
  for (int s = 0; s < kSub; s++) {
    const int index = s * kLocal + tid;
    float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
    for (int col = 0; col < kFeatures; col++) {
      const float t = tmp[s][col];
      r += rmat[col][kFeatures] * t;
      g += rmat[col][kFeatures + 1] * t;
      b += rmat[col][kFeatures + 2] * t;
    }
    const int ux = bx + index % kBlockEdge, uy = by + index / kBlockEdge;
    if (ux < 0 || uy < 0 || ux >= W || uy >= H) continue;
    const size_t i = (size_t)uy * W + ux;
    const float4 alb = loadHalf4(A.albedo, i);
    A.noisy[i] = make_float4(alb.x * (r < 0.0f ? 0.0f : r), alb.y * (g < 0.0f ? 0.0f : g), alb.z * (b < 0.0f ? 0.0f : b), alb.w * spp[s]);
  }
#undef OUT
}

__global__ __launch_bounds__(256) void bmfr_postprocess_kernel(BmfrDev A) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t n = A.W * A.H;
  if (i >= n) return;
  const int W = (int)A.W, H = (int)A.H;
  const int x = (int)(i % A.W);
  const float4 f = A.noisy[i];
  const float texCx = ((float)x + 0.5f) / (float)W;
  float4 res;
  if (!A.full && texCx > 0.5f) {
    res = f;
  } else {
    float prev[3] = {0, 0, 0};
    float blendAlpha = 1.0f;
    if (A.frame > 0) {
      const uint32_t accept = A.accept[i];
      if (accept > 0) {
        const uint32_t pw = A.prevPixel[i];
        const float pfx = f16_to_f32((uint16_t)(pw & 0xffffu)), pfy = f16_to_f32((uint16_t)(pw >> 16));
        const int ipx = (int)pfx, ipy = (int)pfy;
        const float fx = pfx - (float)ipx, fy = pfy - (float)ipy;
        const float ox = 1.0f - fx, oy = 1.0f - fy;
        const float wts[4] = {ox * oy, fx * oy, ox * fy, fx * fy};
        float totalWeight = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (accept & (1u << k)) {
            totalWeight += wts[k];
            const int sx = ipx + (k & 1), sy = ipy + (k >> 1);
            float4 pv = make_float4(0, 0, 0, 0);
            if (sx >= 0 && sy >= 0 && sx < W && sy < H) pv = A.prevFilteredR[(size_t)sy * W + sx];
            prev[0] += wts[k] * pv.x;
            prev[1] += wts[k] * pv.y;
            prev[2] += wts[k] * pv.z;
          }
        if (totalWeight > 0.0f) {
          blendAlpha = 1.0f / f.w;
          blendAlpha = blendAlpha > 0.1f ? blendAlpha : 0.1f;
          prev[0] /= totalWeight;
          prev[1] /= totalWeight;
          prev[2] /= totalWeight;
        }
      }
    }
    res = make_float4(blendAlpha * f.x + (1.0f - blendAlpha) * prev[0], blendAlpha * f.y + (1.0f - blendAlpha) * prev[1],
                      blendAlpha * f.z + (1.0f - blendAlpha) * prev[2], 1.0f);
  }
  A.noisy[i] = res;           // "only curNoisy will be displayed" (DenoisePass.cpp:193)
  A.prevFilteredW[i] = res;   // DenoisePass.cpp:194
}

void launchBmfr(const BmfrDev& A, uint32_t flags, hipStream_t st) {
  const uint32_t n = A.W * A.H;
  if (!n) return;
  const dim3 grid((n + 255u) / 256u), block(256);
  hipLaunchKernelGGL(bmfr_preprocess_kernel, grid, block, 0, st, A);
  if (flags & BDPT_BMFR_REGRESSION) {
    const int bw = ((int)A.W + 31) / 32, bh = ((int)A.H + 31) / 32;
    int w = bw + 1;
    const int h = bh + 1;
    if (!A.full) w /= 2;  // DenoisePass.cpp:262: the reference fits the left half only
    if (w * h > 0) {
      if (flags & BDPT_BMFR_KEEP_LD_FEATURES)
        hipLaunchKernelGGL(bmfr_fit_kernel<false>, dim3((uint32_t)(w * h)), block, 0, st, A, w);
      else
        hipLaunchKernelGGL(bmfr_fit_kernel<true>, dim3((uint32_t)(w * h)), block, 0, st, A, w);
    }
  }
  if (flags & BDPT_BMFR_POSTPROCESS) hipLaunchKernelGGL(bmfr_postprocess_kernel, grid, block, 0, st, A);
}

#undef BD
}  // namespace bdpt
