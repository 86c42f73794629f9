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
// First statement of every kernel built on "one workgroup = one wave" (kernels.hip, above launchWave): launched in any
// other shape it does nothing, instead of indexing its one-wave LDS rows with threadIdx.x >= 64.
#define BDPT_ONE_WAVE_PER_GROUP()               \
  do {                                          \
    if (blockDim.x != (unsigned)kWave) return;  \
  } while (0)

struct Hit {
  int prim;
  float t, u, v;
  uint32_t rec;  // record index of the triangle hit (closest-hit queries; only the kernels that ask for it keep it alive)
};

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
  T.best.rec = 0xFFFFFFFFu;
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
  uint32_t prim, flags, aux, rec;
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
  k.rec = 0u;
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
    T.best.rec = k.rec;
  }
  return false;
}
template <int MODE, bool COUNT>
BD bool triStep(const SceneDev& S, TravState& T, const float4 a, const float4 b, const float4 c, uint32_t rec, bool& last, uint32_t& nAlpha) {
  TriCand k = triGeom<MODE>(T, a, b, c);
  k.rec = rec;
  last = k.last;
  if (!k.ok) return false;
  if (COUNT && (k.flags & 1u) && (MODE == 2 || k.t <= T.best.t)) nAlpha++;
  // any-hit shader: IgnoreHit().  A closest-hit candidate beyond the hit already held cannot be committed whatever the
  // test says (DXR does not report such candidates either), so the test and its fetches are skipped for it.
  if ((k.flags & 1u) && (MODE == 2 || k.t <= T.best.t) && alphaTestFails(S, k.aux, k.u, k.v)) return false;
  return triCommit<MODE>(T, k);
}

// All triangles of the leaf in T.cur; returns true when an any-hit query is finished.
// (COUNT: nTris / nAlpha tally triangle tests and any-hit alpha tests: the "any-hit rate" of BASELINE.md section 3 row 5)
template <int MODE, bool COUNT>
BD bool leafStep(const SceneDev& S, TravState& T, uint32_t& nTris, uint32_t& nAlpha) {
  // The first TWO records are fetched together (six loads in flight): half of the leaves hold two triangles and their
  // second test would otherwise wait for a second, dependent fetch; a one-triangle leaf fetches the record behind it
  // for nothing (the array ends with a pad record).  Likewise the alpha tests of the two: both records, then both texel
  // quads, fetched side by side (the test has no side effect, so running the second one although the first triangle may
  // end an any-hit query changes nothing but a little traffic).
  const float4* tp = reinterpret_cast<const float4*>(S.recs) + (size_t)(uint32_t)~T.cur * kRecF4;
  const float4 a0 = tp[0], b0 = tp[1], c0 = tp[2], a1 = tp[kRecF4], b1 = tp[kRecF4 + 1], c1 = tp[kRecF4 + 2];
  if (COUNT) nTris++;
  TriCand k0 = triGeom<MODE>(T, a0, b0, c0), k1;
  k0.rec = (uint32_t)~T.cur;
  k1.ok = false;
  k1.last = true;
  k1.t = k1.u = k1.v = 0.0f;
  k1.prim = k1.flags = k1.aux = k1.rec = 0u;
  if (!k0.last) {
    if (COUNT) nTris++;
    k1 = triGeom<MODE>(T, a1, b1, c1);
    k1.rec = (uint32_t)~T.cur + 1u;
  }
  // (closest hit: a candidate beyond the hit already held cannot be committed whatever its alpha test says: not run)
  const bool n0 = k0.ok && (k0.flags & 1u) && (MODE == 2 || k0.t <= T.best.t);
  const bool n1 = k1.ok && (k1.flags & 1u) && (MODE == 2 || k1.t <= T.best.t);
  if (COUNT) nAlpha += (n0 ? 1u : 0u) + (n1 ? 1u : 0u);
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
  uint32_t rec = (uint32_t)~T.cur + 2u;
  while (!last) {  // leaves of more than two triangles (builder knob BDPT_LEAF_MAX)
    const float4 a = tp[0], b = tp[1], c = tp[2];
    if (COUNT) nTris++;
    if (triStep<MODE, COUNT>(S, T, a, b, c, rec, last, nAlpha)) return true;
    tp += kRecF4;
    rec++;
  }
  return false;
}

// Whole query by one lane (coherent primary rays, test hooks, the few lazy rays of the gather stage).
template <int MODE, bool COUNT>
BD Hit traverse(const SceneDev& S, f3 o, f3 d, float tmin, float tmax, int* stk, uint32_t& nNodes, uint32_t& nTris) {
  uint32_t nAlpha = 0;  // (primary rays and test hooks: not tallied)
  TravState T;
  travInit(T, o, d, tmin, tmax);
  while (T.cur != kDone) {
    while (T.cur >= 0) {
      if (COUNT) nNodes++;
      nodeStep<(MODE != 2) ? 1 : 0>(S, T, stk);
    }
    if (T.cur == kDone) break;
    if (leafStep<MODE, COUNT>(S, T, nTris, nAlpha)) break;
    T.cur = travPop<kStackEntries>(S, T, stk);
  }
  return T.best;
}

// ------------------------------------------------------------------------------------------------
// Occluder hints.  An any-hit answer is an OR over the scene's triangles, so testing ONE triangle of the scene before
// (instead of) the traversal cannot change it: if that triangle occludes the segment the query is answered, if not the
// ray is traced as before.  The test is the leaf's own — triGeom<2> on the packed record, then the any-hit alpha test —
// so a triangle decides here exactly as it decides inside the traversal.  Where hints come from: kernels.hip
// ("Occluder hints": the primary-visibility triangle of a light-tracing ray's target pixel, the light's cube map of
// nearest triangles for a next-event ray).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kNoHint = 0xFFFFFFFFu;
BD bool recOccludes(const SceneDev& S, uint32_t rec, f3 o, f3 d, float tmin, float tmax) {
  if (rec >= S.numRecs) return false;  // kNoHint, or a stale word: never read outside the array
  TravState T;
  travInit(T, o, d, tmin, tmax);
  if (T.cur == kDone) return false;  // (a ray the traversal retires at entry misses: no triangle may say otherwise)
  const float4* tp = reinterpret_cast<const float4*>(S.recs) + (size_t)rec * kRecF4;
  const TriCand k = triGeom<2>(T, tp[0], tp[1], tp[2]);
  if (!k.ok) return false;
  if ((k.flags & 1u) && alphaTestFails(S, k.aux, k.u, k.v)) return false;
  return true;
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
  BDPT_ONE_WAVE_PER_GROUP();
  __shared__ int s_stack[kStackLds * kWave];
  int* stk = s_stack + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63u);
  bool has = false, exhausted = false;
  uint32_t rid = 0;
  TravState T;
  T.cur = kDone;
  uint32_t nNodes = 0, nTris = 0, nAlpha = 0;
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
          travInit(T, mk(r[0], r[c], r[2 * c]),
                   mk(r[3 * c], r[4 * c], r[5 * c]), shadowTmin, r[6 * c]);
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
        nodeStep<0, kStackLds>(S, T, stk);
        if (COUNT) maxSp = T.sp > maxSp ? T.sp : maxSp;
      }
    }
    const unsigned long long waitMask = __ballot(has && T.cur < 0), nodeMask = __ballot(has && T.cur >= 0);
    const int waitNeed = (__popcll(waitMask | nodeMask) * BDPT_LEAF_WAIT_FRAC8 + 7) >> 3;
    if ((int)__popcll(waitMask) >= waitNeed || nodeMask == 0ull) {
      if (has && T.cur < 0) {
        bool finished = (T.cur == kDone);
        if (!finished) {
          finished = leafStep<2, COUNT>(S, T, nTris, nAlpha);
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
        nodeStep<0, kStackLds>(S, T, stk);
      }
      bool finished = (T.cur == kDone);
      if (!finished) {
        finished = leafStep<2, COUNT>(S, T, nTris, nAlpha);
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
    waveAddCount(counters, C_ALPHA_SHADOW, nAlpha);
    if (maxSp > 0) atomicMax(&counters->v[blockIdx.x % kCounterShards][C_STACK_MAX], (unsigned long long)maxSp);
  }
}


#undef BD
}  // namespace bdpt
