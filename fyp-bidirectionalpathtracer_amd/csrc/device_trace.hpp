// device_trace.hpp — BVH traversal for gfx950: per-lane LDS stack, while-while loop structure,
// and the persistent any-hit trace kernel whose idle lanes are refilled from a ray queue.
//
// Replaces DXR TraceRay (BDPT/globalIlluminationRay.hlsli:11, BDPT/standardShadowRay.hlsli:20-22,
// CP lightProbeGBuffer.rt.hlsl:151-158) — fixed-function in the reference's driver.
// MODE 0 closest hit, 1 closest hit with back-face culling (RAY_FLAG_CULL_BACK_FACING_TRIANGLES),
// 2 any hit (ACCEPT_FIRST_HIT_AND_END_SEARCH | SKIP_CLOSEST_HIT_SHADER).
// Hit iff tmin < t < tmax; closest-hit ties resolve to the lowest primitive index so the result
// does not depend on traversal order (and equals a brute-force scan).
#pragma once
#include "bvh.h"  // record slot size (kRecF4)
#include "device_scene.hpp"

namespace bdpt {
#define BD __device__ __forceinline__

struct Hit {
  int prim;
  float t, u, v;
};

#ifndef BDPT_ORDERED_ANYHIT
#define BDPT_ORDERED_ANYHIT 0
#endif
constexpr int kDone = (int)0x80000000;  // traversal cursor value: stack exhausted

struct TravState {
  f3 o, d, idir;  // idir clamped to +-1e30 so slab arithmetic never produces inf - inf
  uint32_t neg;   // bit a set when d[a] < 0: near plane of axis a is the box's hi plane
  float tmin, tmax;
  int cur, sp;
  Hit best;
};

BD float clampedRcp(float d) {
  float r = 1.0f / d;
  if (!(fabsf(r) <= 1.0e30f)) r = copysignf(1.0e30f, d);
  return r;
}

BD void travInit(TravState& T, f3 o, f3 d, float tmin, float tmax) {
  T.o = o;
  T.d = d;
  T.idir = mk(clampedRcp(d.x), clampedRcp(d.y), clampedRcp(d.z));
  T.neg = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  T.tmin = tmin;
  T.tmax = tmax;
  T.sp = 0;
  T.best.prim = -1;
  T.best.t = tmax;
  T.best.u = 0.0f;
  T.best.v = 0.0f;
  // A ray with a NaN anywhere, or with an empty (tmin, tmax) interval, cannot satisfy
  // tmin < t < tmax for any triangle (the brute-force scan agrees): it misses without traversal.
  const bool finite = (o.x == o.x) && (o.y == o.y) && (o.z == o.z) && (d.x == d.x) && (d.y == d.y) && (d.z == d.z);
  T.cur = (finite && (tmax > tmin)) ? 0 : kDone;
}

// Stack entry e of this lane: LDS row e below KL, the context's overflow area from KL on (KL = kStackEntries: all in LDS).
template <int KL>
BD int* ovfSlot(const SceneDev& S, int e) {
  return S.stackOvf + (size_t)(e - KL) * S.stackOvfStride + (size_t)blockIdx.x * kWave + (threadIdx.x & 63u);
}
template <int KL>
BD void stackStore(const SceneDev& S, int* stk, int e, int ref) {
  if (KL < kStackEntries && e >= KL)
    *ovfSlot<KL>(S, e) = ref;
  else
    stk[e * kWave] = ref;
}
template <int KL>
BD int travPop(const SceneDev& S, TravState& T, const int* stk) {
  if (T.sp == 0) return kDone;
  T.sp--;
  if (KL < kStackEntries && T.sp >= KL) return *ovfSlot<KL>(S, T.sp);
  return stk[T.sp * kWave];
}
template <int KL>
BD void travPush(const SceneDev& S, TravState& T, int* stk, int ref) {
  stackStore<KL>(S, stk, T.sp, ref);
  T.sp++;
}

BD float ubyte(uint32_t w, int c) { return (float)((w >> (8 * c)) & 0xffu); }  // v_cvt_f32_ubyteN

// One visit of a four-wide node with 8-bit quantised child boxes (bvh.h): plane = origin + q*scale (scale stored as
// a float), so t = fma(q, scale*idir, (origin - o)*idir) — one convert and one fma per plane.  Near/far plane
// bytes are picked per axis by the ray's direction sign (one select per axis for all four children).
// ORDER 1 (closest hit): children are entered nearest first, the rest stacked far to near; ORDER 2: the nearest is
// entered, the rest stacked in slot order; ORDER 0 (any hit): slot order.
template <int ORDER, int KL = kStackEntries>
BD void nodeStep(const SceneDev& S, TravState& T, int* stk) {
  const uint4* np = S.recs + (size_t)T.cur * kRecF4;
  // origin.xyz, exponents + leaf bits | lo.x lo.y lo.z hi.x | hi.y hi.z childBase childOffsets   (bvh.h BvhRec)
  const uint4 q0 = np[0], q1 = np[1], q2 = np[2];
  const float sx = __uint_as_float((q0.w << 23) & 0x7f800000u), sy = __uint_as_float((q0.w << 15) & 0x7f800000u),
              sz = __uint_as_float((q0.w << 7) & 0x7f800000u);
  const float ax = sx * T.idir.x, ay = sy * T.idir.y, az = sz * T.idir.z;
  const float bx = (__uint_as_float(q0.x) - T.o.x) * T.idir.x, by = (__uint_as_float(q0.y) - T.o.y) * T.idir.y,
              bz = (__uint_as_float(q0.z) - T.o.z) * T.idir.z;
  const bool nx = (T.neg & 1u) != 0, ny = (T.neg & 2u) != 0, nz = (T.neg & 4u) != 0;
  const uint32_t nearX = nx ? q1.w : q1.x, farX = nx ? q1.x : q1.w;
  const uint32_t nearY = ny ? q2.x : q1.y, farY = ny ? q1.y : q2.x;
  const uint32_t nearZ = nz ? q2.y : q1.z, farZ = nz ? q1.z : q2.y;
  float tn[4];
  bool hit[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const float tnx = fmaf(ubyte(nearX, c), ax, bx), tfx = fmaf(ubyte(farX, c), ax, bx);
    const float tny = fmaf(ubyte(nearY, c), ay, by), tfy = fmaf(ubyte(farY, c), ay, by);
    const float tnz = fmaf(ubyte(nearZ, c), az, bz), tfz = fmaf(ubyte(farZ, c), az, bz);
    const float n = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, T.tmin));
    const float f = fminf(fminf(tfx, tfy), fminf(tfz, T.best.t));
    hit[c] = n <= f;
    tn[c] = n;
  }
  // child c: record childBase + offset byte c; a leaf reference is the complement of its first triangle's record index
  int r0 = (int)q2.z ^ ((int)(q0.w << 7) >> 31);
  int r1 = (int)(q2.z + ((q2.w >> 8) & 0xffu)) ^ ((int)(q0.w << 6) >> 31);
  int r2 = (int)(q2.z + ((q2.w >> 16) & 0xffu)) ^ ((int)(q0.w << 5) >> 31);
  int r3 = (int)(q2.z + (q2.w >> 24)) ^ ((int)(q0.w << 4) >> 31);
  if (ORDER == 2) {
    // nearest hit child by a 3-comparator min tree over (t, slot); every other hit child goes on the stack in slot
    // order with unconditional stores (a slot above sp is scratch) and a conditional stack-pointer bump
    const float t0 = hit[0] ? tn[0] : 3.0e38f, t1 = hit[1] ? tn[1] : 3.0e38f, t2 = hit[2] ? tn[2] : 3.0e38f, t3 = hit[3] ? tn[3] : 3.0e38f;
    const bool s01 = t1 < t0, s23 = t3 < t2;
    const float ta = s01 ? t1 : t0, tb = s23 ? t3 : t2;
    const int ia = s01 ? 1 : 0, ib = s23 ? 3 : 2;
    const bool sab = tb < ta;
    const int near = sab ? ib : ia;
    const bool any = (sab ? tb : ta) < 3.0e38f;
    int sp = T.sp;
    if (KL >= kStackEntries || sp <= KL - 4) {
#pragma unroll
      for (int c = 3; c >= 0; c--) {
        const int rc = (c == 3) ? r3 : ((c == 2) ? r2 : ((c == 1) ? r1 : r0));
        stk[sp * kWave] = rc;
        sp += (hit[c] && c != near) ? 1 : 0;
      }
    } else {  // near the end of the LDS rows: every store picks its place
#pragma unroll
      for (int c = 3; c >= 0; c--) {
        const int rc = (c == 3) ? r3 : ((c == 2) ? r2 : ((c == 1) ? r1 : r0));
        if (hit[c] && c != near) stackStore<KL>(S, stk, sp++, rc);
      }
    }
    T.sp = sp;
    const int rn = (near == 3) ? r3 : ((near == 2) ? r2 : ((near == 1) ? r1 : r0));
    T.cur = any ? rn : travPop<KL>(S, T, stk);
  } else if (ORDER == 1) {
    float t0 = hit[0] ? tn[0] : 3.0e38f, t1 = hit[1] ? tn[1] : 3.0e38f, t2 = hit[2] ? tn[2] : 3.0e38f, t3 = hit[3] ? tn[3] : 3.0e38f;
    r0 = hit[0] ? r0 : kDone;
    r1 = hit[1] ? r1 : kDone;
    r2 = hit[2] ? r2 : kDone;
    r3 = hit[3] ? r3 : kDone;
#define BDPT_CSWAP(ta, ra, tb, rb) \
  {                                \
    const bool sw = tb < ta;       \
    const float tt = sw ? tb : ta; \
    const int rr = sw ? rb : ra;   \
    tb = sw ? ta : tb;             \
    rb = sw ? ra : rb;             \
    ta = tt;                       \
    ra = rr;                       \
  }
    BDPT_CSWAP(t0, r0, t1, r1)
    BDPT_CSWAP(t2, r2, t3, r3)
    BDPT_CSWAP(t0, r0, t2, r2)
    BDPT_CSWAP(t1, r1, t3, r3)
    BDPT_CSWAP(t1, r1, t2, r2)
#undef BDPT_CSWAP
    // misses sorted to the back (t = 3e38, ref = kDone): push far to near, enter the nearest
    if (KL >= kStackEntries || T.sp <= KL - 3) {
      if (r3 != kDone) travPush<kStackEntries>(S, T, stk, r3);
      if (r2 != kDone) travPush<kStackEntries>(S, T, stk, r2);
      if (r1 != kDone) travPush<kStackEntries>(S, T, stk, r1);
    } else {  // near the end of the LDS rows
      if (r3 != kDone) travPush<KL>(S, T, stk, r3);
      if (r2 != kDone) travPush<KL>(S, T, stk, r2);
      if (r1 != kDone) travPush<KL>(S, T, stk, r1);
    }
    T.cur = (r0 != kDone) ? r0 : travPop<KL>(S, T, stk);
  } else {
    // Branch-free: walk the slots from 3 down to 0 keeping the last hit in `next`; a newly found hit
    // pushes the previous one.  The LDS store is unconditional (a slot above sp is scratch), only the
    // stack pointer moves conditionally, so the wave never splits here.
    int next = hit[3] ? r3 : kDone;
    int sp = T.sp;
    if (KL >= kStackEntries || sp <= KL - 3) {
#pragma unroll
      for (int c = 2; c >= 0; c--) {
        const int rc = (c == 2) ? r2 : ((c == 1) ? r1 : r0);
        stk[sp * kWave] = next;
        sp += (hit[c] && next != kDone) ? 1 : 0;
        next = hit[c] ? rc : next;
      }
    } else {  // near the end of the LDS rows: every store picks its place
#pragma unroll
      for (int c = 2; c >= 0; c--) {
        const int rc = (c == 2) ? r2 : ((c == 1) ? r1 : r0);
        if (hit[c] && next != kDone) stackStore<KL>(S, stk, sp++, next);
        next = hit[c] ? rc : next;
      }
    }
    T.sp = sp;
    if (next == kDone) next = travPop<KL>(S, T, stk);
    T.cur = next;
  }
}

// One triangle of a leaf in three parts, so that a leaf's two triangles can share the fetch rounds of their alpha tests:
// triGeom — Moeller-Trumbore exactly as the oracle evaluates it (this is the part that must match bit for bit) — says
// whether the triangle is a candidate hit; the any-hit alpha test (IgnoreHit) may still reject it; triCommit records it.
struct TriCand {
  float t, u, v;
  uint32_t prim, flags, aux;
  bool ok, last;
};
template <int MODE>
BD TriCand triGeom(const TravState& T, const float4 a, const float4 b, const float4 c) {
  TriCand k;
  const f3 v0 = mk(a.x, a.y, a.z), e1 = mk(b.x, b.y, b.z), e2 = mk(c.x, c.y, c.z);
  k.prim = __float_as_uint(a.w);
  k.flags = __float_as_uint(b.w);
  k.aux = __float_as_uint(c.w);
  k.last = (k.flags & 4u) != 0;  // kTriLastOfLeaf
  k.ok = false;
  k.t = k.u = k.v = 0.0f;
  const f3 pvec = cross(T.d, e2);
  const float det = dot(e1, pvec);
  if (MODE == 1 && !(k.flags & 2u)) {
    if (!(det > 0.0f)) return k;
  } else {
    if (det == 0.0f) return k;
  }
  const float inv = 1.0f / det;
  const f3 tvec = T.o - v0;
  const float u = dot(tvec, pvec) * inv;
  if (u < 0.0f || u > 1.0f) return k;
  const f3 qvec = cross(tvec, e1);
  const float v = dot(T.d, qvec) * inv;
  if (v < 0.0f || u + v > 1.0f) return k;
  const float t = dot(e2, qvec) * inv;
  if (!((t > T.tmin) && (t < T.tmax))) return k;
  k.t = t;
  k.u = u;
  k.v = v;
  k.ok = true;
  return k;
}
// returns true when an any-hit query is finished
template <int MODE>
BD bool triCommit(TravState& T, const TriCand& k) {
  if (MODE == 2) {
    T.best.prim = 0;
    T.best.t = k.t;
    return true;
  }
  if (k.t < T.best.t || (k.t == T.best.t && T.best.prim >= 0 && (int)k.prim < T.best.prim)) {
    T.best.prim = (int)k.prim;
    T.best.t = k.t;
    T.best.u = k.u;
    T.best.v = k.v;
  }
  return false;
}
template <int MODE>
BD bool triStep(const SceneDev& S, TravState& T, const float4 a, const float4 b, const float4 c, bool& last) {
  const TriCand k = triGeom<MODE>(T, a, b, c);
  last = k.last;
  if (!k.ok) return false;
  // any-hit shader: IgnoreHit().  A closest-hit candidate beyond the hit already held cannot be committed whatever the
  // test says (DXR does not report such candidates either), so the test and its fetches are skipped for it.
  if ((k.flags & 1u) && (MODE == 2 || k.t <= T.best.t) && alphaTestFails(S, k.aux, k.u, k.v)) return false;
  return triCommit<MODE>(T, k);
}

// All triangles of the leaf in T.cur; returns true when an any-hit query is finished.
template <int MODE, bool COUNT>
BD bool leafStep(const SceneDev& S, TravState& T, uint32_t& nTris) {
  // The first TWO records are fetched together (six loads in flight): half of the leaves hold two triangles and their
  // second test would otherwise wait for a second, dependent fetch; a one-triangle leaf fetches the record behind it
  // for nothing (the array ends with a pad record).  Likewise the alpha tests of the two: both records, then both texel
  // quads, fetched side by side (the test has no side effect, so running the second one although the first triangle may
  // end an any-hit query changes nothing but a little traffic).
  const float4* tp = reinterpret_cast<const float4*>(S.recs) + (size_t)(uint32_t)~T.cur * kRecF4;
  const float4 a0 = ldMaybeNt4<2>(tp), b0 = ldMaybeNt4<2>(tp + 1), c0 = ldMaybeNt4<2>(tp + 2), a1 = ldMaybeNt4<2>(tp + kRecF4),
               b1 = ldMaybeNt4<2>(tp + kRecF4 + 1), c1 = ldMaybeNt4<2>(tp + kRecF4 + 2);
  if (COUNT) nTris++;
  TriCand k0 = triGeom<MODE>(T, a0, b0, c0), k1;
  k1.ok = false;
  k1.last = true;
  k1.t = k1.u = k1.v = 0.0f;
  k1.prim = k1.flags = k1.aux = 0u;
  if (!k0.last) {
    if (COUNT) nTris++;
    k1 = triGeom<MODE>(T, a1, b1, c1);
  }
  // (closest hit: a candidate beyond the hit already held cannot be committed whatever its alpha test says: not run)
  const bool n0 = k0.ok && (k0.flags & 1u) && (MODE == 2 || k0.t <= T.best.t);
  const bool n1 = k1.ok && (k1.flags & 1u) && (MODE == 2 || k1.t <= T.best.t);
  if (n0 || n1) {
    bool f0 = false, f1 = false;
    alphaTestFails2(S, n0, k0.aux, k0.u, k0.v, n1, k1.aux, k1.u, k1.v, f0, f1);
    k0.ok = k0.ok && !f0;
    k1.ok = k1.ok && !f1;
  }
  if (k0.ok && triCommit<MODE>(T, k0)) return true;
  if (k1.ok && triCommit<MODE>(T, k1)) return true;
  bool last = k1.last;
  tp += 2 * kRecF4;
  while (!last) {  // leaves of more than two triangles (builder knob BDPT_LEAF_MAX)
    const float4 a = tp[0], b = tp[1], c = tp[2];
    if (COUNT) nTris++;
    if (triStep<MODE>(S, T, a, b, c, last)) return true;
    tp += kRecF4;
  }
  return false;
}

// Whole query by one lane (coherent primary rays, test hooks, the few lazy rays of the gather stage).
template <int MODE, bool COUNT>
BD Hit traverse(const SceneDev& S, f3 o, f3 d, float tmin, float tmax, int* stk, uint32_t& nNodes, uint32_t& nTris) {
  TravState T;
  travInit(T, o, d, tmin, tmax);
  while (T.cur != kDone) {
    while (T.cur >= 0) {
      if (COUNT) nNodes++;
      nodeStep<((MODE != 2) || BDPT_ORDERED_ANYHIT) ? 1 : 0>(S, T, stk);
    }
    if (T.cur == kDone) break;
    if (leafStep<MODE, COUNT>(S, T, nTris)) break;
    T.cur = travPop<kStackEntries>(S, T, stk);
  }
  return T.best;
}

BD void addCount(DevCounters* c, int idx, uint32_t n) {
  if (n) atomicAdd(&c->v[blockIdx.x % kCounterShards][idx], (unsigned long long)n);
}
// Sum over the wave (every lane must call it), one atomic per wave into the workgroup's shard.
BD void waveAddCount(DevCounters* c, int idx, uint32_t n) {
  uint32_t v = n;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
  if ((threadIdx.x & 63u) == 0u && v) atomicAdd(&c->v[blockIdx.x % kCounterShards][idx], (unsigned long long)v);
}

// ------------------------------------------------------------------------------------------------
// Persistent trace kernel.  Each wave owns 64 traversal slots; whenever kRefillIdle or more lanes
// have retired their ray the wave takes that many new rays from the queue with ONE atomic
// (ballot + popcount prefix), so lanes do not idle behind the longest ray of a static batch.
// Inside, the loop is while-while: all lanes descend interior nodes together, then all lanes
// that reached a leaf intersect triangles together.
// ------------------------------------------------------------------------------------------------
// Loop structure of the persistent kernels (BDPT_LEAF_WAIT 0 = plain while-while): node visits in bursts of
// BDPT_NODE_BURST, leaves intersected once BDPT_LEAF_WAIT_FRAC8 eighths of the lanes that hold a ray are waiting at one.
#ifndef BDPT_LEAF_WAIT
#define BDPT_LEAF_WAIT 1
#endif
#ifndef BDPT_LEAF_WAIT_FRAC8
#define BDPT_LEAF_WAIT_FRAC8 4
#endif
#ifndef BDPT_NODE_BURST
#define BDPT_NODE_BURST 3
#endif
#ifndef BDPT_REFILL_IDLE
#define BDPT_REFILL_IDLE 16
#endif
constexpr int kRefillIdle = BDPT_REFILL_IDLE;

struct RayQueue {         // SoA planes, stride = cap: ox oy oz dx dy dz tmax
  const float* rays;
  uint32_t cap;           // plane stride
  uint32_t subCap;        // capacity of one sub-queue (ray id = q*subCap + offset)
  uint32_t numSub;        // number of sub-queues
  const uint32_t* count;  // [numSub] queued rays (device counters written by the producer)
  uint32_t* head;         // [numSub] next unfetched offset
};

// Any-hit rays (NEE, splat, connection, lazy rounds): visibility byte by ray id, 1 = unoccluded.
// (Closest-hit rays of the walks never pass through a queue: kernels.hip walk_kernel.)
template <bool COUNT>
__global__ __launch_bounds__(kWave) void trace_shadow_kernel(SceneDev S, RayQueue Q, uint8_t* __restrict__ vis, DevCounters* counters,
                                                             float shadowTmin) {
  __shared__ int s_stack[kStackLds * kWave];
  int* stk = s_stack + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63u);
  bool has = false, exhausted = false;
  uint32_t rid = 0;
  TravState T;
  T.cur = kDone;
  uint32_t nNodes = 0, nTris = 0;
  int maxSp = 0;
  // wave-uniform fetch state: current sub-queue and the chunk [chunkPos, chunkEnd) taken from it
  uint32_t q = blockIdx.x % Q.numSub, tried = 0, chunkPos = 0, chunkEnd = 0, chunk = kFetchChunk;
  const uint32_t wavesPerList = (gridDim.x + Q.numSub - 1) / Q.numSub;
  for (;;) {
    const unsigned long long idleMask = __ballot(!has);
    const int idle = __popcll(idleMask);
    if (!exhausted && idle >= kRefillIdle) {
      while (chunkPos >= chunkEnd && !exhausted) {  // wave-uniform loop: take a new chunk
        const uint32_t nq = Q.count[q * kCursorStride];
        uint32_t base = nq;
        if (__hip_atomic_load(&Q.head[q * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nq) {
          // rays per atomic: up to kFetchChunk, but no more than this list's fair share per wave, so
          // that short queues (lazy rounds) still spread over every resident wave
          uint32_t share = (nq / wavesPerList + kWave - 1) & ~(uint32_t)(kWave - 1);
          chunk = share < (uint32_t)kWave ? (uint32_t)kWave : (share > kFetchChunk ? kFetchChunk : share);
          if (lane == 0) base = atomicAdd(&Q.head[q * kCursorStride], chunk);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        }
        if (base < nq) {
          chunkPos = base;
          chunkEnd = (base + chunk < nq) ? base + chunk : nq;
          tried = 0;
        } else {
          q = (q + 1 == Q.numSub) ? 0u : q + 1;
          if (++tried >= Q.numSub) exhausted = true;
        }
      }
      if (!exhausted) {
        const uint32_t avail = chunkEnd - chunkPos;
        const uint32_t take = ((uint32_t)idle < avail) ? (uint32_t)idle : avail;
        const uint32_t rank = (uint32_t)__popcll(idleMask & ((1ull << lane) - 1ull));
        if (!has && rank < take) {
          const uint32_t idx = q * Q.subCap + chunkPos + rank;
          rid = idx;
          const float* r = Q.rays + idx;
          const size_t c = Q.cap;
          travInit(T, mk(ldMaybeNt<1>(r), ldMaybeNt<1>(r + c), ldMaybeNt<1>(r + 2 * c)),
                   mk(ldMaybeNt<1>(r + 3 * c), ldMaybeNt<1>(r + 4 * c), ldMaybeNt<1>(r + 5 * c)), shadowTmin, ldMaybeNt<1>(r + 6 * c));
          has = true;
        }
        chunkPos += take;
      }
    }
    if (__ballot(has) == 0ull) break;  // exhausted and every lane retired
    // while-while: every occupied lane descends interior nodes until it sits on a leaf (or is done),
    // then every lane on a leaf intersects its triangles.  (A majority-vote schedule — one node step
    // OR one triangle test per iteration, whichever group is larger — raised lane utilisation from
    // 0.33 to 0.52 but executed as many wave instructions because of its per-iteration bookkeeping,
    // so the simpler loop stays: profiles/r1.)
#if BDPT_LEAF_WAIT > 0
    // Deferred leaves: every lane with a ray takes up to BDPT_NODE_BURST node visits; a lane that reaches a leaf (or runs
    // out of stack) waits, and the leaves are intersected once half of the lanes that hold a ray are waiting or no lane
    // can take a node visit.  The plain while-while loop (#else) descends until EVERY lane sits on a leaf, so the node visits —
    // nine tenths of the work — ran with a third of the lanes; a leaf phase after every node visit (if-if) or by
    // majority vote (round 1) paid a 65-instruction triangle test for a handful of lanes per iteration.  Measured on
    // the bench frame: any-hit tracing 7.9 -> 6.7 ms, walk 7.4 -> 6.5 ms (profiles/README.md).
    if (has) {
#pragma unroll 1
      for (int k = 0; k < BDPT_NODE_BURST && T.cur >= 0; k++) {
        if (COUNT) nNodes++;
        nodeStep<BDPT_ORDERED_ANYHIT ? 1 : 0, kStackLds>(S, T, stk);
        if (COUNT) maxSp = T.sp > maxSp ? T.sp : maxSp;
      }
    }
    const unsigned long long waitMask = __ballot(has && T.cur < 0), nodeMask = __ballot(has && T.cur >= 0);
    const int waitNeed = (__popcll(waitMask | nodeMask) * BDPT_LEAF_WAIT_FRAC8 + 7) >> 3;
    if ((int)__popcll(waitMask) >= waitNeed || nodeMask == 0ull) {
      if (has && T.cur < 0) {
        bool finished = (T.cur == kDone);
        if (!finished) {
          finished = leafStep<2, COUNT>(S, T, nTris);
          if (!finished) {
            T.cur = travPop<kStackLds>(S, T, stk);
            finished = (T.cur == kDone);
          }
        }
        if (finished) {
          vis[rid] = (T.best.prim < 0) ? (uint8_t)1 : (uint8_t)0;
          has = false;
          T.cur = kDone;
        }
      }
    }
  }
#else
    if (has) {
      while (T.cur >= 0) {
        if (COUNT) nNodes++;
        nodeStep<BDPT_ORDERED_ANYHIT ? 1 : 0, kStackLds>(S, T, stk);
      }
      bool finished = (T.cur == kDone);
      if (!finished) {
        finished = leafStep<2, COUNT>(S, T, nTris);
        if (!finished) {
          T.cur = travPop<kStackLds>(S, T, stk);
          finished = (T.cur == kDone);
        }
      }
      if (finished) {
        vis[rid] = (T.best.prim < 0) ? (uint8_t)1 : (uint8_t)0;
        has = false;
        T.cur = kDone;
      }
    }
  }
#endif
  if (COUNT) {
    waveAddCount(counters, C_NODE_SHADOW, nNodes);
    waveAddCount(counters, C_TRI_SHADOW, nTris);
    if (maxSp > 0) atomicMax(&counters->v[blockIdx.x % kCounterShards][C_STACK_MAX], (unsigned long long)maxSp);
  }
}


// ------------------------------------------------------------------------------------------------
// EXPERIMENT (round 3, profiles/README.md): four lanes per ray.  A wave holds 16 any-hit rays; lane c of a ray's quad
// decodes and slab-tests child c of the four-wide node, the quad's hit bits combine through __ballot, one stack per
// quad (2 KiB of LDS per wave for all 32 rows: no overflow area), a leaf's triangles are tested by the quad's lanes
// side by side.  Any-hit answers do not depend on the order of traversal, so the visibility bytes are those of
// trace_shadow_kernel bit for bit.  Built only with -DBDPT_QUAD_ANYHIT=1.
// ------------------------------------------------------------------------------------------------
#ifndef BDPT_QUAD_ANYHIT
#define BDPT_QUAD_ANYHIT 0
#endif
#if BDPT_QUAD_ANYHIT
template <bool COUNT>
__global__ __launch_bounds__(kWave) void trace_shadow_quad_kernel(SceneDev S, RayQueue Q, uint8_t* __restrict__ vis, DevCounters* counters,
                                                                  float shadowTmin) {
  constexpr int kQuads = kWave / 4;
  __shared__ int s_stack[kStackEntries * kQuads];
  const int lane = (int)(threadIdx.x & 63u), sub = lane & 3, slot = lane >> 2, quadBase = lane & ~3;
  int* stk = s_stack + slot;  // entry e of this quad's ray at stk[e * kQuads]
  bool has = false, exhausted = false;
  uint32_t rid = 0;
  f3 o = mk(0), idir = mk(0), d = mk(0);
  float tmin = shadowTmin, tmax = 0.0f;
  int cur = kDone, sp = 0;
  uint32_t nNodes = 0, nTris = 0;
  uint32_t q = blockIdx.x % Q.numSub, tried = 0, chunkPos = 0, chunkEnd = 0, chunk = kFetchChunk;
  const uint32_t wavesPerList = (gridDim.x + Q.numSub - 1) / Q.numSub;
  const unsigned long long leaders = 0x1111111111111111ull;  // lane 0 of every quad
  for (;;) {
    const unsigned long long idleMask = __ballot(!has) & leaders;
    const int idle = __popcll(idleMask);
    if (!exhausted && idle >= 4) {
      while (chunkPos >= chunkEnd && !exhausted) {
        const uint32_t nq = Q.count[q * kCursorStride];
        uint32_t base = nq;
        if (__hip_atomic_load(&Q.head[q * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nq) {
          uint32_t share = (nq / wavesPerList + kQuads - 1) & ~(uint32_t)(kQuads - 1);
          chunk = share < (uint32_t)kQuads ? (uint32_t)kQuads : (share > kFetchChunk ? kFetchChunk : share);
          if (lane == 0) base = atomicAdd(&Q.head[q * kCursorStride], chunk);
          base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        }
        if (base < nq) {
          chunkPos = base;
          chunkEnd = (base + chunk < nq) ? base + chunk : nq;
          tried = 0;
        } else {
          q = (q + 1 == Q.numSub) ? 0u : q + 1;
          if (++tried >= Q.numSub) exhausted = true;
        }
      }
      if (!exhausted) {
        const uint32_t avail = chunkEnd - chunkPos;
        const uint32_t take = ((uint32_t)idle < avail) ? (uint32_t)idle : avail;
        const uint32_t rank = (uint32_t)__popcll(idleMask & ((1ull << quadBase) - 1ull));  // idle quads below this one
        if (!has && rank < take) {
          const uint32_t idx = q * Q.subCap + chunkPos + rank;
          rid = idx;
          const float* r = Q.rays + idx;
          const size_t c = Q.cap;
          o = mk(r[0], r[c], r[2 * c]);
          d = mk(r[3 * c], r[4 * c], r[5 * c]);
          tmax = r[6 * c];
          idir = mk(clampedRcp(d.x), clampedRcp(d.y), clampedRcp(d.z));
          const bool finite = (o.x == o.x) && (o.y == o.y) && (o.z == o.z) && (d.x == d.x) && (d.y == d.y) && (d.z == d.z);
          cur = (finite && (tmax > shadowTmin)) ? 0 : kDone;
          sp = 0;
          has = true;
        }
        chunkPos += take;
      }
    }
    if (__ballot(has) == 0ull) break;
    // ---- node visits: every quad whose ray stands on an interior node, up to BDPT_NODE_BURST times
#pragma unroll 1
    for (int k = 0; k < BDPT_NODE_BURST; k++) {
      const bool at = has && cur >= 0;
      if (__ballot(at) == 0ull) break;
      bool hit = false;
      int ref = kDone;
      if (at) {
        if (COUNT && sub == 0) nNodes++;
        const uint4* np = S.recs + (size_t)cur * kRecF4;
        const uint4 q0 = np[0], q1 = np[1], q2 = np[2];
        const uint32_t sh = 8u * (uint32_t)sub;
        const float sx = __uint_as_float((q0.w << 23) & 0x7f800000u), sy = __uint_as_float((q0.w << 15) & 0x7f800000u),
                    sz = __uint_as_float((q0.w << 7) & 0x7f800000u);
        const float ax = sx * idir.x, ay = sy * idir.y, az = sz * idir.z;
        const float bx = (__uint_as_float(q0.x) - o.x) * idir.x, by = (__uint_as_float(q0.y) - o.y) * idir.y,
                    bz = (__uint_as_float(q0.z) - o.z) * idir.z;
        const uint32_t lx = (q1.x >> sh) & 0xffu, ly = (q1.y >> sh) & 0xffu, lz = (q1.z >> sh) & 0xffu;
        const uint32_t hx = (q1.w >> sh) & 0xffu, hy = (q2.x >> sh) & 0xffu, hz = (q2.y >> sh) & 0xffu;
        const float t0x = fmaf((float)lx, ax, bx), t1x = fmaf((float)hx, ax, bx);
        const float t0y = fmaf((float)ly, ay, by), t1y = fmaf((float)hy, ay, by);
        const float t0z = fmaf((float)lz, az, bz), t1z = fmaf((float)hz, az, bz);
        const float n = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
        const float f = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
        hit = (n <= f) && (lx <= hx);  // an unused slot has lo = 255, hi = 0 on every axis
        const int idx = (int)(q2.z + ((q2.w >> sh) & 0xffu));
        ref = idx ^ ((int)(q0.w << (7 - sub)) >> 31);
      }
      const unsigned long long hm = __ballot(hit);
      if (at) {
        const uint32_t m = (uint32_t)(hm >> quadBase) & 15u;
        const int pos = __popc(m & ((1u << sub) - 1u)), nh = __popc(m);
        if (hit) stk[(sp + pos) * kQuads] = ref;
        sp += nh;
        if (sp == 0) {
          cur = kDone;
        } else {
          sp--;
          cur = stk[sp * kQuads];
        }
      }
    }
    // ---- leaves (and retirement) once half of the rays wait there or none can take a node visit
    const unsigned long long waitMask = __ballot(has && cur < 0) & leaders, nodeMask = __ballot(has && cur >= 0) & leaders;
    const int waitNeed = (__popcll(waitMask | nodeMask) * BDPT_LEAF_WAIT_FRAC8 + 7) >> 3;
    if ((int)__popcll(waitMask) >= waitNeed || nodeMask == 0ull) {
      const bool atLeaf = has && cur < 0 && cur != kDone;
      bool found = false;
      // a leaf's triangles, four at a time: lane `sub` takes triangle `sub` unless an earlier one ends the leaf
      uint32_t base = atLeaf ? (uint32_t)~cur : 0u;
      bool more = atLeaf;
      while (__ballot(more) != 0ull) {
        // fetch four records; those behind the leaf's last triangle are not triangles of this leaf (another leaf's, a
        // node's, or the zero pad): their flags are read to find the end, nothing else of them is used
        bool lastHere = false, hitHere = false;
        float4 a = make_float4(0, 0, 0, 0), b = a, c = a;
        if (more) {
          const float4* tp = reinterpret_cast<const float4*>(S.recs) + (size_t)(base + (uint32_t)sub) * kRecF4;
          a = tp[0];
          b = tp[1];
          c = tp[2];
          lastHere = (__float_as_uint(b.w) & 4u) != 0u;
        }
        const unsigned long long lm = __ballot(lastHere);
        const uint32_t lq = (uint32_t)(lm >> quadBase) & 15u;
        // triangles up to and including the first "last of leaf" belong to the leaf
        const uint32_t validBits = lq ? ((2u << (uint32_t)(__ffs((int)lq) - 1)) - 1u) : 15u;
        if (more && ((validBits >> sub) & 1u)) {
          TravState T;
          T.o = o;
          T.d = d;
          T.tmin = tmin;
          T.tmax = tmax;
          const TriCand kc = triGeom<2>(T, a, b, c);
          hitHere = kc.ok;
          if (hitHere && (kc.flags & 1u)) hitHere = !alphaTestFails(S, kc.aux, kc.u, kc.v);
        }
        const unsigned long long hmask = __ballot(hitHere);
        if (more) {
          const uint32_t hq = (uint32_t)(hmask >> quadBase) & 15u;
          if (COUNT && sub == 0) nTris += (uint32_t)__popc(validBits);
          if (hq) found = true;
          more = !found && lq == 0u;
          base += 4u;
        }
      }
      if (has && cur < 0) {
        bool finished = (cur == kDone) || found;
        if (!finished) {
          if (sp == 0) {
            finished = true;
          } else {
            sp--;
            cur = stk[sp * kQuads];
          }
        }
        if (finished) {
          if (sub == 0) vis[rid] = found ? (uint8_t)0 : (uint8_t)1;
          has = false;
          cur = kDone;
        }
      }
    }
  }
  if (COUNT) {
    waveAddCount(counters, C_NODE_SHADOW, nNodes);
    waveAddCount(counters, C_TRI_SHADOW, nTris);
  }
}
#endif  // BDPT_QUAD_ANYHIT

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (round 3, profiles/README.md): two rays per lane.  Every lane owns a LIVE ray (registers) and a PARKED
// one (14 dwords in LDS); when the live ray reaches a leaf and the parked one can take node visits the two swap, so
// the node bursts — nine tenths of the work — run with the lanes the deferred-leaf loop leaves idle.  Each ray keeps
// only KL rows of its stack in LDS (the rest in the context's overflow area), so that the wave holds twice the rays
// in about the LDS the one-ray kernel uses.  Any-hit answers are independent of the order of traversal: visibility
// bytes identical to trace_shadow_kernel.  Built only with -DBDPT_TWO_RAYS=1.
// ------------------------------------------------------------------------------------------------
#if BDPT_TWO_RAYS
constexpr int kParkFields = 14;  // o d idir (9) neg tmax cur sp rid
template <bool COUNT, int KL>
__global__ __launch_bounds__(kWave) void trace_shadow2_kernel(SceneDev S0, RayQueue Q, uint8_t* __restrict__ vis, DevCounters* counters,
                                                              float shadowTmin) {
  __shared__ int s_stack[2 * KL * kWave];
  __shared__ uint32_t s_park[kParkFields * kWave];
  const int lane = (int)(threadIdx.x & 63u);
  uint32_t* park = s_park + lane;  // field f at park[f * 64]
  bool has = false, hasParked = false, exhausted = false;
  uint32_t rid = 0, which = 0;  // which half of the stack (and of the overflow rows) the live ray uses
  TravState T;
  T.cur = kDone;
  T.tmin = shadowTmin;
  uint32_t nNodes = 0, nTris = 0;
  int maxSp = 0;
  SceneDev S = S0;  // stackOvf is offset per lane by the live ray's half
  auto bindStack = [&]() { S.stackOvf = S0.stackOvf + (size_t)which * (size_t)(kStackEntries - KL) * S0.stackOvfStride; };
  bindStack();
  auto initRay = [&](TravState& R, uint32_t idx) {
    const float* r = Q.rays + idx;
    const size_t c = Q.cap;
    travInit(R, mk(r[0], r[c], r[2 * c]), mk(r[3 * c], r[4 * c], r[5 * c]), shadowTmin, r[6 * c]);
  };
  auto storeParked = [&](const TravState& R, uint32_t id) {
    park[0 * 64] = __float_as_uint(R.o.x);
    park[1 * 64] = __float_as_uint(R.o.y);
    park[2 * 64] = __float_as_uint(R.o.z);
    park[3 * 64] = __float_as_uint(R.d.x);
    park[4 * 64] = __float_as_uint(R.d.y);
    park[5 * 64] = __float_as_uint(R.d.z);
    park[6 * 64] = __float_as_uint(R.idir.x);
    park[7 * 64] = __float_as_uint(R.idir.y);
    park[8 * 64] = __float_as_uint(R.idir.z);
    park[9 * 64] = R.neg;
    park[10 * 64] = __float_as_uint(R.tmax);
    park[11 * 64] = (uint32_t)R.cur;
    park[12 * 64] = (uint32_t)R.sp;
    park[13 * 64] = id;
  };
  auto loadParked = [&](TravState& R, uint32_t& id) {
    R.o = mk(__uint_as_float(park[0 * 64]), __uint_as_float(park[1 * 64]), __uint_as_float(park[2 * 64]));
    R.d = mk(__uint_as_float(park[3 * 64]), __uint_as_float(park[4 * 64]), __uint_as_float(park[5 * 64]));
    R.idir = mk(__uint_as_float(park[6 * 64]), __uint_as_float(park[7 * 64]), __uint_as_float(park[8 * 64]));
    R.neg = park[9 * 64];
    R.tmin = shadowTmin;
    R.tmax = __uint_as_float(park[10 * 64]);
    R.cur = (int)park[11 * 64];
    R.sp = (int)park[12 * 64];
    R.best.prim = -1;
    R.best.t = R.tmax;
    R.best.u = R.best.v = 0.0f;
    id = park[13 * 64];
  };
  uint32_t q = blockIdx.x % Q.numSub, tried = 0, chunkPos = 0, chunkEnd = 0, chunk = kFetchChunk;
  const uint32_t wavesPerList = (gridDim.x + Q.numSub - 1) / Q.numSub;
  for (;;) {
    // ---- refill: empty live slots first, then empty parked slots
    const unsigned long long liveEmpty = __ballot(!has), parkEmpty = __ballot(!hasParked);
    const int nLE = __popcll(liveEmpty), nPE = __popcll(parkEmpty);
    if (!exhausted && (nLE >= kRefillIdle || nLE + nPE >= 2 * kRefillIdle)) {
      uint32_t want = (uint32_t)(nLE + nPE), taken = 0;
      const uint32_t rankL = (uint32_t)__popcll(liveEmpty & ((1ull << lane) - 1ull));
      const uint32_t rankP = (uint32_t)nLE + (uint32_t)__popcll(parkEmpty & ((1ull << lane) - 1ull));
      bool gotL = false, gotP = false;
      uint32_t idL = 0, idP = 0;
      while (want > 0 && !exhausted) {
        while (chunkPos >= chunkEnd && !exhausted) {
          const uint32_t nq = Q.count[q * kCursorStride];
          uint32_t base = nq;
          if (__hip_atomic_load(&Q.head[q * kCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nq) {
            uint32_t share = (nq / wavesPerList + kWave - 1) & ~(uint32_t)(kWave - 1);
            chunk = share < (uint32_t)kWave ? (uint32_t)kWave : (share > kFetchChunk ? kFetchChunk : share);
            if (lane == 0) base = atomicAdd(&Q.head[q * kCursorStride], chunk);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
          }
          if (base < nq) {
            chunkPos = base;
            chunkEnd = (base + chunk < nq) ? base + chunk : nq;
            tried = 0;
          } else {
            q = (q + 1 == Q.numSub) ? 0u : q + 1;
            if (++tried >= Q.numSub) exhausted = true;
          }
        }
        if (exhausted) break;
        const uint32_t avail = chunkEnd - chunkPos;
        const uint32_t take = (want < avail) ? want : avail;
        if (!has && !gotL && rankL >= taken && rankL < taken + take) {
          idL = q * Q.subCap + chunkPos + (rankL - taken);
          gotL = true;
        }
        if (!hasParked && !gotP && rankP >= taken && rankP < taken + take) {
          idP = q * Q.subCap + chunkPos + (rankP - taken);
          gotP = true;
        }
        chunkPos += take;
        taken += take;
        want -= take;
      }
      if (gotL) {
        rid = idL;
        initRay(T, idL);
        has = true;
      }
      if (gotP) {
        TravState R;
        initRay(R, idP);
        storeParked(R, idP);
        hasParked = true;
      }
    }
    if (__ballot(has || hasParked) == 0ull) break;
    // ---- a lane whose live slot is empty, or whose live ray waits at a leaf while the parked one can take node visits, swaps
    {
      const int parkedCur = hasParked ? (int)park[11 * 64] : kDone;
      const bool swapIn = hasParked && (!has || (T.cur < 0 && parkedCur >= 0));
      if (swapIn) {
        TravState R;
        uint32_t id2;
        loadParked(R, id2);
        if (has) storeParked(T, rid);
        hasParked = has;
        T = R;
        rid = id2;
        has = true;
        which ^= 1u;
        bindStack();
      }
    }
    int* stk = s_stack + (int)which * KL * kWave + lane;
    // ---- node visits in bursts; leaves once half of the live rays wait at one (device_trace.hpp trace_shadow_kernel)
    if (has) {
#pragma unroll 1
      for (int k = 0; k < BDPT_NODE_BURST && T.cur >= 0; k++) {
        if (COUNT) nNodes++;
        nodeStep<0, KL>(S, T, stk);
        if (COUNT) maxSp = T.sp > maxSp ? T.sp : maxSp;
      }
    }
    const unsigned long long waitMask = __ballot(has && T.cur < 0), nodeMask = __ballot(has && T.cur >= 0);
    const int waitNeed = (__popcll(waitMask | nodeMask) * BDPT_LEAF_WAIT_FRAC8 + 7) >> 3;
    if ((int)__popcll(waitMask) >= waitNeed || nodeMask == 0ull) {
      if (has && T.cur < 0) {
        bool finished = (T.cur == kDone);
        if (!finished) {
          finished = leafStep<2, COUNT>(S, T, nTris);
          if (!finished) {
            T.cur = travPop<KL>(S, T, stk);
            finished = (T.cur == kDone);
          }
        }
        if (finished) {
          vis[rid] = (T.best.prim < 0) ? (uint8_t)1 : (uint8_t)0;
          has = false;
          T.cur = kDone;
        }
      }
    }
  }
  if (COUNT) {
    waveAddCount(counters, C_NODE_SHADOW, nNodes);
    waveAddCount(counters, C_TRI_SHADOW, nTris);
    if (maxSp > 0) atomicMax(&counters->v[blockIdx.x % kCounterShards][C_STACK_MAX], (unsigned long long)maxSp);
  }
}
#endif  // BDPT_TWO_RAYS

#undef BD
}  // namespace bdpt
