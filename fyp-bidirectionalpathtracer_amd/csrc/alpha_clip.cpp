// alpha_clip.cpp — see alpha_clip.h.
#include "alpha_clip.h"

#include <algorithm>
#include <cmath>

namespace bdpt {
namespace {

inline long floorDiv(long a, long n) {
  long q = a / n;
  if ((a % n) != 0 && ((a < 0) != (n < 0))) q--;
  return q;
}
inline int wrapHost(int i, int n) {
  const int m = i % n;
  return (m < 0) ? m + n : m;
}

// Sutherland-Hodgman against A + B bu + C bv <= 0 (closed), in place
int clipInPlace(double (*poly)[2], int n, double A, double B, double C) {
  double out[kBvhPolyMax][2];
  int m = 0;
  for (int k = 0; k < n; k++) {
    const double* p = poly[k];
    const double* q = poly[(k + 1) % n];
    const double fp = A + B * p[0] + C * p[1], fq = A + B * q[0] + C * q[1];
    if (fp <= 0.0 && m < kBvhPolyMax) {
      out[m][0] = p[0];
      out[m][1] = p[1];
      m++;
    }
    if (((fp < 0.0 && fq > 0.0) || (fp > 0.0 && fq < 0.0)) && m < kBvhPolyMax) {
      const double t = fp / (fp - fq);
      out[m][0] = p[0] + t * (q[0] - p[0]);
      out[m][1] = p[1] + t * (q[1] - p[1]);
      m++;
    }
  }
  for (int k = 0; k < m; k++) {
    poly[k][0] = out[k][0];
    poly[k][1] = out[k][1];
  }
  return m;
}

}  // namespace

// Texels the sample of cell (i, j) blends: (i, j), (i+1, j), (i, j+1), (i+1, j+1), wrapped (device_scene.hpp
// alphaTestFails).  The blend is a convex combination evaluated in fp32: it stays within [min, max] of the four up
// to a few ulps, so a cell whose largest texel is below threshold - 1e-5 always fails and one whose smallest is at or
// above threshold + 1e-5 always passes.
AlphaClipper::AlphaClipper(const bdpt_scene_desc* d) : d_(d) {
  mats_.resize(d->numMaterials);
  for (uint32_t mi = 0; mi < d->numMaterials; mi++) {
    const bdpt_material& m = d->materials[mi];
    if (BDPT_FLAG_ALPHA_MODE(m.flags) == BDPT_ALPHA_MODE_OPAQUE) continue;
    const uint32_t type = BDPT_FLAG_DIFFUSE_TYPE(m.flags);
    MatInfo& mi_ = mats_[mi];
    if (type == BDPT_CHANNEL_UNUSED) {
      mi_.verdict = (0.0f < m.alphaThreshold) ? 2 : 1;
    } else if (type == BDPT_CHANNEL_CONST || m.texBaseColor < 0) {
      mi_.verdict = (m.baseColor[3] < m.alphaThreshold) ? 2 : 1;
    } else if ((uint32_t)m.texBaseColor < d->numTextures && d->textures[m.texBaseColor].rgba8 && d->textures[m.texBaseColor].width &&
               d->textures[m.texBaseColor].height && d->textures[m.texBaseColor].width <= 16384 && d->textures[m.texBaseColor].height <= 16384) {
      const bdpt_texture& t = d->textures[m.texBaseColor];
      Mask mk;
      mk.w = (int)t.width;
      mk.h = (int)t.height;
      const size_t W1 = (size_t)mk.w + 1;
      mk.mayPass.assign(W1 * ((size_t)mk.h + 1), 0u);
      mk.mayFail.assign(W1 * ((size_t)mk.h + 1), 0u);
      const float thr = m.alphaThreshold;
      for (int j = 0; j < mk.h; j++) {
        const int j1 = wrapHost(j + 1, mk.h);
        for (int i = 0; i < mk.w; i++) {
          const int i1 = wrapHost(i + 1, mk.w);
          const float a00 = (float)t.rgba8[((size_t)j * mk.w + i) * 4 + 3] / 255.0f, a10 = (float)t.rgba8[((size_t)j * mk.w + i1) * 4 + 3] / 255.0f;
          const float a01 = (float)t.rgba8[((size_t)j1 * mk.w + i) * 4 + 3] / 255.0f, a11 = (float)t.rgba8[((size_t)j1 * mk.w + i1) * 4 + 3] / 255.0f;
          const float hi = std::max(std::max(a00, a10), std::max(a01, a11)), lo = std::min(std::min(a00, a10), std::min(a01, a11));
          const uint32_t pass = !(hi < thr - 1e-5f) ? 1u : 0u, fail = (lo < thr + 1e-5f) ? 1u : 0u;
          const size_t o = ((size_t)j + 1) * W1 + (size_t)i + 1;
          mk.mayPass[o] = pass + mk.mayPass[o - 1] + mk.mayPass[o - W1] - mk.mayPass[o - W1 - 1];
          mk.mayFail[o] = fail + mk.mayFail[o - 1] + mk.mayFail[o - W1] - mk.mayFail[o - W1 - 1];
        }
      }
      mi_.mask = (int)masks_.size();
      masks_.push_back(std::move(mk));
    }
  }
}

// cells [x0, x1] x [y0, y1], unwrapped inclusive coordinates; each span at most one period long
uint32_t AlphaClipper::Mask::count(const std::vector<uint32_t>& sat, long x0, long x1, long y0, long y1) const {
  if (x1 < x0 || y1 < y0) return 0;
  const size_t W1 = (size_t)w + 1;
  auto rect = [&](long xa, long xb, long ya, long yb) -> uint32_t {  // wrapped-in-range inclusive
    return sat[((size_t)yb + 1) * W1 + (size_t)xb + 1] - sat[(size_t)ya * W1 + (size_t)xb + 1] - sat[((size_t)yb + 1) * W1 + (size_t)xa] +
           sat[(size_t)ya * W1 + (size_t)xa];
  };
  long xs[2][2], ys[2][2];
  int nx = 0, ny = 0;
  {
    const long s = floorDiv(x0, w) * w, a = x0 - s, b = x1 - s;
    if (b < w) {
      xs[nx][0] = a, xs[nx][1] = b, nx++;
    } else {
      xs[nx][0] = a, xs[nx][1] = w - 1, nx++;
      xs[nx][0] = 0, xs[nx][1] = std::min<long>(b - w, w - 1), nx++;
    }
  }
  {
    const long s = floorDiv(y0, h) * h, a = y0 - s, b = y1 - s;
    if (b < h) {
      ys[ny][0] = a, ys[ny][1] = b, ny++;
    } else {
      ys[ny][0] = a, ys[ny][1] = h - 1, ny++;
      ys[ny][0] = 0, ys[ny][1] = std::min<long>(b - h, h - 1), ny++;
    }
  }
  uint32_t c = 0;
  for (int i = 0; i < nx; i++)
    for (int j = 0; j < ny; j++) c += rect(xs[i][0], xs[i][1], ys[j][0], ys[j][1]);
  return c;
}

// The cells the samples of a barycentric polygon can fall in.  Returns false when nothing can be said (no deciding
// texture, non-finite or huge coordinates).
bool AlphaClipper::cellRect(uint32_t tri, const double (*poly)[2], int n, const Mask*& m, long& x0, long& x1, long& y0, long& y1,
                            double& margin, double uv[3][2]) const {
  const MatInfo& mi = mats_[d_->triMaterial[tri]];
  if (mi.mask < 0 || !d_->texcoords) return false;
  m = &masks_[(size_t)mi.mask];
  double big = 0.0;
  for (int k = 0; k < 3; k++) {
    const uint32_t vi = d_->indices[(size_t)tri * 3 + (size_t)k];
    uv[k][0] = (double)d_->texcoords[(size_t)vi * 3];
    uv[k][1] = (double)d_->texcoords[(size_t)vi * 3 + 1];
    if (!std::isfinite(uv[k][0]) || !std::isfinite(uv[k][1])) return false;
    big = std::max(big, std::max(std::fabs(uv[k][0]), std::fabs(uv[k][1])));
  }
  if (big > 4096.0) return false;
  // What separates the texel the device's alpha test samples from the one under the exact hit point: the fp32
  // rounding of the coordinate itself (a few ulps of its magnitude) and, for rays that graze the triangle, the error of
  // the Moeller-Trumbore barycentrics (eps * distance / (extent * sin of the incidence angle)).  Half a texel covers
  // the second down to a fraction of a degree for a card a few hundred texels across — the same order of world-space
  // slack as the pad every box of the tree gets (bvh_build.cpp: 2e-5 of the scene diagonal).
  margin = 0.5 + 1e-5 * (big + 1.0) * (double)std::max(m->w, m->h);
  double xa = 1e300, xb = -1e300, ya = 1e300, yb = -1e300;
  for (int k = 0; k < n; k++) {
    const double b0 = 1.0 - poly[k][0] - poly[k][1];
    const double u = uv[0][0] * b0 + uv[1][0] * poly[k][0] + uv[2][0] * poly[k][1];
    const double v = uv[0][1] * b0 + uv[1][1] * poly[k][0] + uv[2][1] * poly[k][1];
    const double x = u * (double)m->w - 0.5, y = v * (double)m->h - 0.5;
    xa = std::min(xa, x);
    xb = std::max(xb, x);
    ya = std::min(ya, y);
    yb = std::max(yb, y);
  }
  x0 = (long)std::floor(xa - margin);
  x1 = (long)std::floor(xb + margin);
  y0 = (long)std::floor(ya - margin);
  y1 = (long)std::floor(yb + margin);
  return true;
}

int AlphaClipper::classify(uint32_t tri) const {
  const MatInfo& mi = mats_[d_->triMaterial[tri]];
  if (mi.mask < 0) return mi.verdict;
  const double whole[3][2] = {{0.0, 0.0}, {1.0, 0.0}, {0.0, 1.0}};
  const Mask* m = nullptr;
  long x0, x1, y0, y1;
  double margin, uv[3][2];
  if (!cellRect(tri, whole, 3, m, x0, x1, y0, y1, margin, uv)) return 0;
  if (x1 - x0 + 1 >= m->w) x0 = 0, x1 = m->w - 1;
  if (y1 - y0 + 1 >= m->h) y0 = 0, y1 = m->h - 1;
  if (m->count(m->mayFail, x0, x1, y0, y1) == 0) return 1;
  if (m->count(m->mayPass, x0, x1, y0, y1) == 0) return 2;
  return 0;
}

bool AlphaClipper::clip(uint32_t tri, double (*poly)[2], int& n) const {
  const MatInfo& mi = mats_[d_->triMaterial[tri]];
  if (mi.mask < 0) return mi.verdict != 2;
  const Mask* m = nullptr;
  long x0, x1, y0, y1;
  double margin, uv[3][2];
  if (!cellRect(tri, poly, n, m, x0, x1, y0, y1, margin, uv)) return true;
  const bool fullX = x1 - x0 + 1 >= m->w, fullY = y1 - y0 + 1 >= m->h;
  if (fullX) x0 = 0, x1 = m->w - 1;
  if (fullY) y0 = 0, y1 = m->h - 1;
  if (m->count(m->mayPass, x0, x1, y0, y1) == 0) return false;
  // smallest rectangle of cells holding every cell of the footprint a sample may pass in
  auto firstCol = [&](long a, long b) {  // smallest x in [a, b] with a passing cell in column range [a, x]
    while (a < b) {
      const long mid = a + (b - a) / 2;
      if (m->count(m->mayPass, x0, mid, y0, y1) > 0)
        b = mid;
      else
        a = mid + 1;
    }
    return a;
  };
  auto lastCol = [&](long a, long b) {
    while (a < b) {
      const long mid = a + (b - a + 1) / 2;
      if (m->count(m->mayPass, mid, x1, y0, y1) > 0)
        a = mid;
      else
        b = mid - 1;
    }
    return a;
  };
  auto firstRow = [&](long a, long b) {
    while (a < b) {
      const long mid = a + (b - a) / 2;
      if (m->count(m->mayPass, x0, x1, y0, mid) > 0)
        b = mid;
      else
        a = mid + 1;
    }
    return a;
  };
  auto lastRow = [&](long a, long b) {
    while (a < b) {
      const long mid = a + (b - a + 1) / 2;
      if (m->count(m->mayPass, x0, x1, mid, y1) > 0)
        a = mid;
      else
        b = mid - 1;
    }
    return a;
  };
  // a sample falls in cell floor(x), x = u w - 0.5 as the device computes it: cells [c0, c1] <=> x in [c0, c1 + 1)
  const double du1 = uv[1][0] - uv[0][0], du2 = uv[2][0] - uv[0][0], dv1 = uv[1][1] - uv[0][1], dv2 = uv[2][1] - uv[0][1];
  if (!fullX) {
    const long c0 = firstCol(x0, x1), c1 = lastCol(x0, x1);
    const double uLo = ((double)c0 + 0.5 - margin) / (double)m->w, uHi = ((double)c1 + 1.5 + margin) / (double)m->w;
    if (c0 > x0) n = clipInPlace(poly, n, uLo - uv[0][0], -du1, -du2);  // u >= uLo
    if (n >= 3 && c1 < x1) n = clipInPlace(poly, n, uv[0][0] - uHi, du1, du2);  // u <= uHi
  }
  if (n >= 3 && !fullY) {
    const long r0 = firstRow(y0, y1), r1 = lastRow(y0, y1);
    const double vLo = ((double)r0 + 0.5 - margin) / (double)m->h, vHi = ((double)r1 + 1.5 + margin) / (double)m->h;
    if (r0 > y0) n = clipInPlace(poly, n, vLo - uv[0][1], -dv1, -dv2);
    if (n >= 3 && r1 < y1) n = clipInPlace(poly, n, uv[0][1] - vHi, dv1, dv2);
  }
  return n >= 3;
}

bool AlphaClipper::tables(BvhClipTables& out) const {
  out.triMaterial = d_->triMaterial;
  out.indices = d_->indices;
  out.texcoords = d_->texcoords;
  out.numTriangles = d_->numTriangles;
  out.numVertices = d_->numVertices;
  out.matMask.clear();
  out.matVerdict.clear();
  for (const MatInfo& m : mats_) {
    out.matMask.push_back(m.mask);
    out.matVerdict.push_back(m.verdict);
  }
  out.masks.clear();
  for (const Mask& m : masks_) out.masks.push_back(BvhClipTables::Mask{m.w, m.h, m.mayPass.data()});
  return true;
}

bool AlphaClipper::testFails(uint32_t tri, float bu, float bv) const {
  const bdpt_material& mm = d_->materials[d_->triMaterial[tri]];
  const uint32_t type = BDPT_FLAG_DIFFUSE_TYPE(mm.flags);
  float alpha = 0.0f;
  if (type == BDPT_CHANNEL_UNUSED) {
    alpha = 0.0f;
  } else if (type == BDPT_CHANNEL_CONST || mm.texBaseColor < 0) {
    alpha = mm.baseColor[3];
  } else {
    const bdpt_texture& t = d_->textures[mm.texBaseColor];
    float uvs[3][2];
    for (int k = 0; k < 3; k++) {
      const uint32_t vi = d_->indices[(size_t)tri * 3 + (size_t)k];
      uvs[k][0] = d_->texcoords ? d_->texcoords[(size_t)vi * 3] : 0.0f;
      uvs[k][1] = d_->texcoords ? d_->texcoords[(size_t)vi * 3 + 1] : 0.0f;
    }
    float u = 0, v = 0;
    const float b0 = 1.0f - bu - bv;
    u += uvs[0][0] * b0;
    v += uvs[0][1] * b0;
    u += uvs[1][0] * bu;
    v += uvs[1][1] * bu;
    u += uvs[2][0] * bv;
    v += uvs[2][1] * bv;
    const int tw = (int)t.width, th = (int)t.height;
    const float x = u * (float)tw - 0.5f, y = v * (float)th - 0.5f;
    const float x0 = std::floor(x), y0 = std::floor(y);
    const float fx = x - x0, fy = y - y0;
    const int ix0 = wrapHost((int)x0, tw), iy0 = wrapHost((int)y0, th);
    const int ix1 = wrapHost(ix0 + 1, tw), iy1 = wrapHost(iy0 + 1, th);
    const uint8_t* px = t.rgba8;
    const float t00 = (float)px[((size_t)iy0 * tw + (size_t)ix0) * 4 + 3] / 255.0f, t10 = (float)px[((size_t)iy0 * tw + (size_t)ix1) * 4 + 3] / 255.0f;
    const float t01 = (float)px[((size_t)iy1 * tw + (size_t)ix0) * 4 + 3] / 255.0f, t11 = (float)px[((size_t)iy1 * tw + (size_t)ix1) * 4 + 3] / 255.0f;
    const float top = t00 + (t10 - t00) * fx, bot = t01 + (t11 - t01) * fx;
    alpha = top + (bot - top) * fy;
  }
  return alpha < mm.alphaThreshold;
}

}  // namespace bdpt
