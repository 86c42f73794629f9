// ImageDecode.cpp — PNG and baseline-JPEG decoders for the scene loader's textures.
//
// The reference reads images through FreeImage (Falcor Utils/Bitmap.cpp:45-140), which is not in this image and
// cannot be fetched, so the two formats its own assets use (.MISSING_LARGE_BLOBS: PNG / JPG textures next to the
// FBX models) are decoded here from their published specifications:
//   * zlib / DEFLATE (RFC 1950, RFC 1951): stored, fixed and dynamic Huffman blocks;
//   * PNG (ISO/IEC 15948): colour types 0, 2, 3, 4, 6 at bit depths 1-8, the five scan-line filters, tRNS for
//     RGB images; Adam7 interlacing and 16-bit samples are refused (FreeImage hands 16-bit images to Falcor as
//     48/64-bit DIBs, which Bitmap.cpp:108-114 mislabels as half-float: nothing to be in parity with);
//   * JPEG (ITU-T T.81) baseline sequential DCT, Huffman coding, 8-bit samples, 1 or 3 components, sampling
//     factors 1 or 2, restart intervals; with the arithmetic of the IJG reference decoder's defaults as shipped up to
//     release 6b (jidctint.c "slow integer" inverse DCT, jdsample.c "fancy" triangle up-sampling, jdcolor.c fixed-point
//     YCbCr -> RGB), which libjpeg-turbo keeps: tests compare bit for bit with Pillow (libjpeg-turbo).  FreeImage
//     3.17 bundles IJG release 9a, whose chroma up-sampling works in the DCT domain: parity with the reference's
//     decoded texels is unpinned.  Progressive and arithmetic-coded files are refused.
// What the loader does with the channel count follows Bitmap.cpp:104-126: 32-bit images keep their alpha (and so
// make a material's alpha mode Mask, Material.cpp:120-126), 24-bit and grey images have none.
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace bdpt {
// Header dimensions are not trusted: a few hundred bytes can announce a 65535 x 65535 image.
constexpr uint32_t kMaxImageSide = 16384;
constexpr uint64_t kMaxImageTexels = 1ull << 28;
namespace {

// ---------------------------------------------------------------------------------------------------------------
// DEFLATE
// ---------------------------------------------------------------------------------------------------------------
struct BitReader {
  const uint8_t* p;
  size_t n, pos = 0;
  uint32_t acc = 0;
  int cnt = 0;
  bool bad = false;
  BitReader(const uint8_t* d, size_t len) : p(d), n(len) {}
  uint32_t bits(int k) {  // k <= 16, LSB first
    while (cnt < k) {
      if (pos >= n) {
        bad = true;
        return 0;
      }
      acc |= (uint32_t)p[pos++] << cnt;
      cnt += 8;
    }
    const uint32_t v = acc & ((1u << k) - 1u);
    acc >>= k;
    cnt -= k;
    return v;
  }
  void alignByte() {
    acc = 0;
    cnt = 0;
  }
};

struct Huffman {  // canonical code, decoded bit by bit (RFC 1951 section 3.2.2)
  uint16_t count[16];
  uint16_t symbol[288];
  bool build(const uint8_t* lengths, int n) {
    std::memset(count, 0, sizeof(count));
    for (int i = 0; i < n; i++) count[lengths[i]]++;
    count[0] = 0;
    int left = 1;
    for (int len = 1; len < 16; len++) {
      left <<= 1;
      left -= count[len];
      if (left < 0) return false;  // over-subscribed
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
    for (int i = 0; i < n; i++)
      if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
    return true;
  }
  int decode(BitReader& br) const {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; len++) {
      code |= (int)br.bits(1);
      if (br.bad) return -1;
      const int c = count[len];
      if (code - c < first) return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};

bool inflateRaw(BitReader& br, std::vector<uint8_t>& out, size_t limit) {
  static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  for (;;) {
    const uint32_t last = br.bits(1), type = br.bits(2);
    if (br.bad) return false;
    if (type == 0) {
      br.alignByte();
      if (br.pos + 4 > br.n) return false;
      const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xffffu) != nlen || br.pos + len > br.n || out.size() + len > limit) return false;
      out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
      br.pos += len;
    } else if (type == 1 || type == 2) {
      Huffman lit, dist;
      uint8_t lengths[320];
      if (type == 1) {
        int i = 0;
        for (; i < 144; i++) lengths[i] = 8;
        for (; i < 256; i++) lengths[i] = 9;
        for (; i < 280; i++) lengths[i] = 7;
        for (; i < 288; i++) lengths[i] = 8;
        lit.build(lengths, 288);
        for (i = 0; i < 30; i++) lengths[i] = 5;
        dist.build(lengths, 30);
      } else {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
        if (br.bad || nlen > 286 || ndist > 30) return false;
        uint8_t cl[19] = {0};
        for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
        Huffman lencode;
        if (!lencode.build(cl, 19)) return false;
        int idx = 0;
        while (idx < nlen + ndist) {
          const int sym = lencode.decode(br);
          if (sym < 0) return false;
          if (sym < 16) {
            lengths[idx++] = (uint8_t)sym;
          } else {
            int rep, val = 0;
            if (sym == 16) {
              if (idx == 0) return false;
              val = lengths[idx - 1];
              rep = 3 + (int)br.bits(2);
            } else if (sym == 17) {
              rep = 3 + (int)br.bits(3);
            } else {
              rep = 11 + (int)br.bits(7);
            }
            if (br.bad || idx + rep > nlen + ndist) return false;
            while (rep--) lengths[idx++] = (uint8_t)val;
          }
        }
        if (lengths[256] == 0) return false;
        if (!lit.build(lengths, nlen)) return false;
        if (!dist.build(lengths + nlen, ndist)) return false;
      }
      for (;;) {
        int sym = lit.decode(br);
        if (sym < 0) return false;
        if (sym < 256) {
          if (out.size() >= limit) return false;
          out.push_back((uint8_t)sym);
        } else if (sym == 256) {
          break;
        } else {
          sym -= 257;
          if (sym >= 29) return false;
          const int len = lbase[sym] + (int)br.bits(lext[sym]);
          const int ds = dist.decode(br);
          if (ds < 0 || ds >= 30) return false;
          const size_t d = dbase[ds] + br.bits(dext[ds]);
          if (br.bad || d > out.size() || out.size() + (size_t)len > limit) return false;
          const size_t from = out.size() - d;
          for (int i = 0; i < len; i++) out.push_back(out[from + (size_t)i]);
        }
      }
    } else {
      return false;
    }
    if (last) return true;
  }
}

bool zlibInflate(const uint8_t* d, size_t n, std::vector<uint8_t>& out, size_t expected) {
  if (n < 6 || (d[0] & 0x0f) != 8 || ((d[0] << 8) | d[1]) % 31 != 0 || (d[1] & 0x20)) return false;
  BitReader br(d + 2, n - 2);
  out.clear();
  out.reserve(std::min<size_t>(expected, n * 1032 + 64));  // (a stored header's size is not trusted: DEFLATE expands at most 1032 : 1)
  if (!inflateRaw(br, out, expected)) return false;
  // Adler-32 of the output closes the stream
  br.alignByte();
  if (br.pos + 4 > br.n) return false;
  uint32_t a = 1, b = 0;
  for (size_t i = 0; i < out.size(); i++) {
    a = (a + out[i]) % 65521u;
    b = (b + a) % 65521u;
  }
  const uint32_t want = ((uint32_t)br.p[br.pos] << 24) | ((uint32_t)br.p[br.pos + 1] << 16) | ((uint32_t)br.p[br.pos + 2] << 8) | br.p[br.pos + 3];
  return want == ((b << 16) | a);
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

// rgba8: width*height*4, row 0 first.  channels: what the file carried after FreeImage's conversions would have
// (4 = alpha present: RGBA, grey + alpha, RGB + tRNS; 3 = RGB or palette; 1 = grey).
bool decodePng(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba8, int& channels, std::string& err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (n < 8 || std::memcmp(d, sig, 8) != 0) {
    err = "not a PNG file";
    return false;
  }
  size_t p = 8;
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  bool end = false;
  while (!end && p + 12 <= n) {
    const uint32_t len = be32(d + p);
    const uint8_t* type = d + p + 4;
    const uint8_t* body = d + p + 8;
    if (p + 12 + (size_t)len > n) {
      err = "truncated PNG chunk";
      return false;
    }
    if (!std::memcmp(type, "IHDR", 4) && len == 13) {
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
      if (body[10] != 0 || body[11] != 0) {
        err = "unknown PNG compression / filter method";
        return false;
      }
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      trns.assign(body, body + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    p += 12 + (size_t)len;
  }
  if (!w || !h || w > kMaxImageSide || h > kMaxImageSide || (uint64_t)w * h > kMaxImageTexels || ctype < 0) {
    err = "bad PNG header (or dimensions beyond the loader's limit)";
    return false;
  }
  if (interlace) {
    err = "interlaced PNG is not supported";
    return false;
  }
  if (depth == 16) {
    err = "16-bit PNG is not supported";
    return false;
  }
  int samples;
  switch (ctype) {
    case 0: samples = 1; break;
    case 2: samples = 3; break;
    case 3: samples = 1; break;
    case 4: samples = 2; break;
    case 6: samples = 4; break;
    default: err = "bad PNG colour type"; return false;
  }
  const bool depthOk = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                       ((ctype == 2 || ctype == 4 || ctype == 6) && depth == 8);
  if (!depthOk) {
    err = "bad PNG bit depth";
    return false;
  }
  if (ctype == 3 && plte.size() < 3) {
    err = "palette PNG without PLTE";
    return false;
  }
  const size_t bpp = (size_t)(samples * depth + 7) / 8;                  // bytes per complete pixel, at least 1 (filter unit)
  const size_t stride = ((size_t)w * samples * depth + 7) / 8;          // bytes per scan line
  std::vector<uint8_t> raw;
  if (!zlibInflate(idat.data(), idat.size(), raw, (stride + 1) * h) || raw.size() != (stride + 1) * h) {
    err = "PNG data stream is corrupt";
    return false;
  }
  // undo the scan-line filters in place (ISO/IEC 15948 section 9)
  std::vector<uint8_t> prev(stride, 0);
  for (uint32_t y = 0; y < h; y++) {
    uint8_t* line = &raw[(size_t)y * (stride + 1)];
    const int ft = line[0];
    uint8_t* cur = line + 1;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = cur[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: err = "bad PNG filter type"; return false;
      }
      cur[i] = (uint8_t)v;
    }
    std::memcpy(prev.data(), cur, stride);
  }
  width = w;
  height = h;
  rgba8.assign((size_t)w * h * 4, 255);
  const bool rgbKey = ctype == 2 && trns.size() >= 6;  // a single transparent colour (16-bit big-endian samples in the chunk)
  channels = (ctype == 6 || ctype == 4 || rgbKey) ? 4 : ((ctype == 2 || ctype == 3) ? 3 : 1);
  const int maxv = (1 << depth) - 1;
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t* cur = &raw[(size_t)y * (stride + 1) + 1];
    uint8_t* o = &rgba8[(size_t)y * w * 4];
    for (uint32_t x = 0; x < w; x++, o += 4) {
      if (ctype == 6) {
        std::memcpy(o, cur + (size_t)x * 4, 4);
      } else if (ctype == 2) {
        std::memcpy(o, cur + (size_t)x * 3, 3);
        if (rgbKey && o[0] == trns[1] && o[1] == trns[3] && o[2] == trns[5]) o[3] = 0;
      } else if (ctype == 4) {
        o[0] = o[1] = o[2] = cur[(size_t)x * 2];
        o[3] = cur[(size_t)x * 2 + 1];
      } else {  // 0 grey or 3 palette: depth-bit samples, most significant bits first
        const size_t bit = (size_t)x * depth;
        const int v = (cur[bit >> 3] >> (8 - depth - (int)(bit & 7))) & maxv;
        if (ctype == 0) {
          o[0] = o[1] = o[2] = (uint8_t)(v * 255 / maxv);
        } else {
          if ((size_t)v * 3 + 2 >= plte.size()) {
            err = "PNG palette index out of range";
            return false;
          }
          o[0] = plte[(size_t)v * 3];
          o[1] = plte[(size_t)v * 3 + 1];
          o[2] = plte[(size_t)v * 3 + 2];
        }
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// JPEG, baseline sequential (ITU-T T.81), with the IJG release-6b decoder's default arithmetic
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct JHuff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  bool present = false;
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    present = true;
  }
};

struct JBits {
  const uint8_t* p;
  size_t n, pos;
  uint32_t acc = 0;
  int cnt = 0;
  bool marker = false;  // ran into a marker: feed zeros (T.81 F.2.2.5)
  int starved = 0;      // zero bytes fed after a marker or the end of the file
  JBits(const uint8_t* d, size_t len, size_t at) : p(d), n(len), pos(at) {}
  void fill() {
    while (cnt <= 24) {
      int b = 0;
      if (marker || pos >= n) starved++;
      if (!marker && pos < n) {
        b = p[pos];
        if (b == 0xff) {
          const int b2 = pos + 1 < n ? p[pos + 1] : 0xd9;
          if (b2 == 0) {
            pos += 2;
          } else {
            marker = true;
            b = 0;
          }
        } else {
          pos++;
        }
      }
      acc |= (uint32_t)b << (24 - cnt);
      cnt += 8;
    }
  }
  int get(int k) {  // k <= 16, MSB first
    if (k == 0) return 0;
    if (cnt < k) fill();
    const int v = (int)(acc >> (32 - k));
    acc <<= k;
    cnt -= k;
    return v;
  }
  void reset() {
    acc = 0;
    cnt = 0;
    marker = false;
    starved = 0;
  }
  // The entropy data ended (marker or end of file) well before the block being decoded: a truncated or corrupt scan.
  // (A few zero bytes are legitimate: the decoder reads ahead of the last MCU.)
  bool ranDry() const { return starved > 8; }
};

inline int jDecode(JBits& br, const JHuff& h) {
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br.get(1);
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}
inline int jExtend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

// jidctint.c (IJG 6b): accurate integer inverse DCT, CONST_BITS = 13, PASS1_BITS = 2; input already dequantised
void idctIslow(const int* in, uint8_t* out, int outStride) {
  constexpr int CB = 13, P1 = 2;
  constexpr long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819,
                 F2562 = 20995, F3072 = 25172;
  auto descale = [](long x, int n) { return (x + (1L << (n - 1))) >> n; };
  long ws[64];
  for (int c = 0; c < 8; c++) {
    const int* ip = in + c;
    long* wp = ws + c;
    if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
      const long dc = (long)ip[0] * (1L << P1);  // (a left shift of a negative value is undefined before C++20)
      for (int r = 0; r < 8; r++) wp[r * 8] = dc;
      continue;
    }
    long z2 = ip[16], z3 = ip[48];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = ip[0];
    z3 = ip[32];
    long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ip[56];
    tmp1 = ip[40];
    tmp2 = ip[24];
    tmp3 = ip[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298;
    tmp1 *= F2053;
    tmp2 *= F3072;
    tmp3 *= F1501;
    z1 *= -F0899;
    z2 *= -F2562;
    z3 *= -F1961;
    z4 *= -F0390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    wp[0] = descale(tmp10 + tmp3, CB - P1);
    wp[56] = descale(tmp10 - tmp3, CB - P1);
    wp[8] = descale(tmp11 + tmp2, CB - P1);
    wp[48] = descale(tmp11 - tmp2, CB - P1);
    wp[16] = descale(tmp12 + tmp1, CB - P1);
    wp[40] = descale(tmp12 - tmp1, CB - P1);
    wp[24] = descale(tmp13 + tmp0, CB - P1);
    wp[32] = descale(tmp13 - tmp0, CB - P1);
  }
  auto clampSample = [](long v) {  // range_limit: level shift by 128, then clamp to 0..255 (after masking as the table does)
    v += 128;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  };
  for (int r = 0; r < 8; r++) {
    const long* wp = ws + r * 8;
    uint8_t* op = out + (size_t)r * outStride;
    long z2 = wp[2], z3 = wp[6];
    long z1 = (z2 + z3) * F0541;
    long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    long tmp0 = (wp[0] + wp[4]) * (1L << CB), tmp1 = (wp[0] - wp[4]) * (1L << CB);
    const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = wp[7];
    tmp1 = wp[5];
    tmp2 = wp[3];
    tmp3 = wp[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    long z4 = tmp1 + tmp3;
    const long z5 = (z3 + z4) * F1175;
    tmp0 *= F0298;
    tmp1 *= F2053;
    tmp2 *= F3072;
    tmp3 *= F1501;
    z1 *= -F0899;
    z2 *= -F2562;
    z3 *= -F1961;
    z4 *= -F0390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    constexpr int SH = CB + P1 + 3;
    op[0] = clampSample(descale(tmp10 + tmp3, SH));
    op[7] = clampSample(descale(tmp10 - tmp3, SH));
    op[1] = clampSample(descale(tmp11 + tmp2, SH));
    op[6] = clampSample(descale(tmp11 - tmp2, SH));
    op[2] = clampSample(descale(tmp12 + tmp1, SH));
    op[5] = clampSample(descale(tmp12 - tmp1, SH));
    op[3] = clampSample(descale(tmp13 + tmp0, SH));
    op[4] = clampSample(descale(tmp13 - tmp0, SH));
  }
}

struct JComp {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int wBlocks = 0, hBlocks = 0;  // allocated size in blocks (whole MCUs)
  std::vector<uint8_t> plane;    // wBlocks*8 x hBlocks*8
  int pred = 0;
};

}  // namespace

bool decodeJpeg(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba8, int& channels, std::string& err) {
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) {
    err = "not a JPEG file";
    return false;
  }
  static const uint8_t zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  uint16_t qt[4][64] = {{0}};
  JHuff dc[4], ac[4];
  JComp comp[3];
  int ncomp = 0, W = 0, H = 0, restart = 0;
  bool adobe = false, sofBaseline = true, wideTables = false;
  int adobeTransform = -1;
  size_t p = 2;
  while (p + 4 <= n) {
    if (d[p] != 0xff) {
      p++;
      continue;
    }
    const int m = d[p + 1];
    if (m == 0xff) {
      p++;
      continue;
    }
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) {
      p += 2;
      continue;
    }
    const size_t len = ((size_t)d[p + 2] << 8) | d[p + 3];
    if (len < 2 || p + 2 + len > n) {
      err = "truncated JPEG segment";
      return false;
    }
    const uint8_t* s = d + p + 4;
    const size_t sl = len - 2;
    if (m == 0xdb) {  // DQT
      size_t q = 0;
      while (q < sl) {
        const int pq = s[q] >> 4, tq = s[q] & 15;
        q++;
        wideTables = wideTables || pq != 0;
        if (tq > 3 || q + (pq ? 128u : 64u) > sl) {
          err = "bad JPEG quantisation table";
          return false;
        }
        for (int i = 0; i < 64; i++) {
          qt[tq][zigzag[i]] = pq ? (uint16_t)((s[q] << 8) | s[q + 1]) : s[q];
          q += pq ? 2 : 1;
        }
      }
    } else if (m == 0xc4) {  // DHT
      size_t q = 0;
      while (q + 17 <= sl) {
        const int tc = s[q] >> 4, th = s[q] & 15;
        if (th > 3 || tc > 1) {
          err = "bad JPEG Huffman table id";
          return false;
        }
        JHuff& hf = tc ? ac[th] : dc[th];
        int total = 0;
        for (int i = 1; i <= 16; i++) {
          hf.bits[i] = s[q + (size_t)i];
          total += hf.bits[i];
        }
        q += 17;
        if (total > 256 || q + (size_t)total > sl) {
          err = "bad JPEG Huffman table";
          return false;
        }
        std::memcpy(hf.vals, s + q, (size_t)total);
        q += (size_t)total;
        hf.build();
      }
    } else if (m == 0xc0 || m == 0xc1) {  // SOF0 / SOF1 (extended sequential, Huffman) with 8-bit samples
      if (sl < 6 || s[0] != 8) {
        err = "only 8-bit JPEG samples are supported";
        return false;
      }
      H = (s[1] << 8) | s[2];
      W = (s[3] << 8) | s[4];
      ncomp = s[5];
      if ((ncomp != 1 && ncomp != 3) || sl < 6 + (size_t)ncomp * 3 || !W || !H) {
        err = "unsupported JPEG component count";
        return false;
      }
      if (W > (int)kMaxImageSide || H > (int)kMaxImageSide || (uint64_t)W * (uint64_t)H > kMaxImageTexels) {
        err = "JPEG dimensions exceed the loader's limit";
        return false;
      }
      sofBaseline = (m == 0xc0);
      for (int i = 0; i < ncomp; i++) {
        comp[i].id = s[6 + i * 3];
        comp[i].h = s[7 + i * 3] >> 4;
        comp[i].v = s[7 + i * 3] & 15;
        comp[i].tq = s[8 + i * 3] & 3;
        if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2) {
          err = "unsupported JPEG sampling factors";
          return false;
        }
      }
    } else if (m == 0xc2 || m == 0xc9 || m == 0xca || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      err = "progressive / arithmetic / lossless JPEG is not supported";
      return false;
    } else if (m == 0xdd && sl >= 2) {
      restart = (s[0] << 8) | s[1];
    } else if (m == 0xee && sl >= 12 && !std::memcmp(s, "Adobe", 5)) {
      adobe = true;
      adobeTransform = s[11];
    } else if (m == 0xda) {  // SOS: the one scan of a baseline file
      if (!ncomp || sl < 1 + (size_t)s[0] * 2 + 3 || s[0] != ncomp) {
        err = "unsupported JPEG scan layout";
        return false;
      }
      if (wideTables && sofBaseline) {  // T.81 B.2.4.1: Pq = 0 for 8-bit sample precision
        err = "16-bit quantisation tables in a baseline JPEG";
        return false;
      }
      for (int i = 0; i < ncomp; i++) {
        int ci = -1;
        for (int k = 0; k < ncomp; k++)
          if (comp[k].id == s[1 + i * 2]) ci = k;
        if (ci != i) {
          err = "unsupported JPEG component order";
          return false;
        }
        comp[i].td = s[2 + i * 2] >> 4;
        comp[i].ta = s[2 + i * 2] & 15;
        if (comp[i].td > 3 || comp[i].ta > 3 || !dc[comp[i].td].present || !ac[comp[i].ta].present) {
          err = "JPEG scan names a missing Huffman table";
          return false;
        }
      }
      int hmax = 1, vmax = 1;
      for (int i = 0; i < ncomp; i++) {
        hmax = comp[i].h > hmax ? comp[i].h : hmax;
        vmax = comp[i].v > vmax ? comp[i].v : vmax;
      }
      if (ncomp == 1) comp[0].h = comp[0].v = hmax = vmax = 1;  // a single-component scan is not interleaved
      if (ncomp == 3 && (comp[0].h != hmax || comp[0].v != vmax || comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1)) {
        err = "unsupported JPEG sampling layout (luma must carry the maximum factors, chroma 1x1)";
        return false;
      }
      const int mcuW = 8 * hmax, mcuH = 8 * vmax, mcusX = (W + mcuW - 1) / mcuW, mcusY = (H + mcuH - 1) / mcuH;
      for (int i = 0; i < ncomp; i++) {
        comp[i].wBlocks = mcusX * comp[i].h;
        comp[i].hBlocks = mcusY * comp[i].v;
        comp[i].plane.assign((size_t)comp[i].wBlocks * 8 * comp[i].hBlocks * 8, 0);
        comp[i].pred = 0;
      }
      JBits br(d, n, p + 2 + len);
      int coef[64];
      int untilRestart = restart;
      for (int my = 0; my < mcusY; my++)
        for (int mx = 0; mx < mcusX; mx++) {
          if (restart && untilRestart == 0) {
            // byte-align, expect RSTn
            br.reset();
            while (br.pos + 1 < n && !(d[br.pos] == 0xff && d[br.pos + 1] >= 0xd0 && d[br.pos + 1] <= 0xd7)) br.pos++;
            br.pos += 2;
            for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
            untilRestart = restart;
          }
          for (int i = 0; i < ncomp; i++) {
            JComp& c = comp[i];
            for (int by = 0; by < c.v; by++)
              for (int bx = 0; bx < c.h; bx++) {
                std::memset(coef, 0, sizeof(coef));
                const int t = jDecode(br, dc[c.td]);
                if (t < 0 || t > 11 || br.ranDry()) {  // T.81 F.1.2.1.1: DC difference categories 0-11 for 8-bit samples
                  err = "corrupt JPEG entropy data";
                  return false;
                }
                const int diff = t ? jExtend(br.get(t), t) : 0;
                c.pred += diff;
                if (c.pred > 32767 || c.pred < -32768) {  // the prediction of valid data stays within 11 + 3 bits
                  err = "corrupt JPEG entropy data";
                  return false;
                }
                coef[0] = c.pred * (int)qt[c.tq][0];
                for (int k = 1; k < 64;) {
                  const int rs = jDecode(br, ac[c.ta]);
                  if (rs < 0) {
                    err = "corrupt JPEG entropy data";
                    return false;
                  }
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz > 10 || br.ranDry()) {  // AC categories 1-10 for 8-bit samples
                    err = "corrupt JPEG entropy data";
                    return false;
                  }
                  if (sz == 0) {
                    if (r == 15) {
                      k += 16;
                      continue;
                    }
                    break;  // end of block
                  }
                  k += r;
                  if (k > 63) {
                    err = "corrupt JPEG entropy data";
                    return false;
                  }
                  coef[zigzag[k]] = jExtend(br.get(sz), sz) * qt[c.tq][zigzag[k]];
                  k++;
                }
                const int stride = c.wBlocks * 8;
                idctIslow(coef, &c.plane[((size_t)(my * c.v + by) * 8) * stride + (size_t)(mx * c.h + bx) * 8], stride);
              }
          }
          if (restart) untilRestart--;
        }
      // ---- output: up-sample chroma (jdsample.c "fancy" triangle filters), convert (jdcolor.c)
      width = (uint32_t)W;
      height = (uint32_t)H;
      rgba8.assign((size_t)W * H * 4, 255);
      channels = ncomp == 1 ? 1 : 3;
      if (ncomp == 1) {
        const int stride = comp[0].wBlocks * 8;
        for (int y = 0; y < H; y++)
          for (int x = 0; x < W; x++) {
            uint8_t* o = &rgba8[((size_t)y * W + x) * 4];
            o[0] = o[1] = o[2] = comp[0].plane[(size_t)y * stride + x];
          }
        return true;
      }
      const bool ycc = !(adobe && adobeTransform == 0);  // Adobe transform 0 = RGB stored directly
      // chroma planes at luma resolution.  libjpeg works on the down-sampled image of ceil(W / h) x ceil(H / v) samples,
      // edge-replicated by one sample for the filter context; rows beyond the image use the last real row.
      const int cw = (W + hmax - 1) / hmax, ch = (H + vmax - 1) / vmax;  // real chroma samples
      std::vector<uint8_t> up[2];
      for (int k = 0; k < 2; k++) {
        const JComp& c = comp[k + 1];
        const int stride = c.wBlocks * 8;
        up[k].assign((size_t)W * H, 0);
        auto at = [&](int x, int y) -> int {
          x = x < 0 ? 0 : (x >= cw ? cw - 1 : x);
          y = y < 0 ? 0 : (y >= ch ? ch - 1 : y);
          return c.plane[(size_t)y * stride + x];
        };
        for (int y = 0; y < H; y++) {
          for (int x = 0; x < W; x++) {
            int v;
            if (hmax == 1 && vmax == 1) {
              v = at(x, y);
            } else if (cw <= 2) {  // jinit_upsampler: the triangle filters need more than two chroma columns, else replication
              v = at(hmax == 2 ? x >> 1 : x, vmax == 2 ? y >> 1 : y);
            } else if (hmax == 2 && vmax == 1) {  // h2v1_fancy_upsample: (3*near + far + 1 or 2) >> 2
              const int cx = x >> 1;
              if (cw == 1)
                v = at(0, y);
              else if (x & 1)
                v = (cx == cw - 1) ? at(cx, y) : ((3 * at(cx, y) + at(cx + 1, y) + 2) >> 2);
              else
                v = (cx == 0) ? at(0, y) : ((3 * at(cx, y) + at(cx - 1, y) + 1) >> 2);
            } else if (hmax == 2 && vmax == 2) {  // h2v2_fancy_upsample: 9/16, 3/16, 3/16, 1/16
              const int cx = x >> 1, cy = y >> 1;
              const int ny = (y & 1) ? cy + 1 : cy - 1;  // the nearer neighbouring chroma row (clamped by `at`)
              auto col = [&](int xx) { return 3 * at(xx, cy) + at(xx, ny); };
              const int thisc = col(cx);
              if (cw == 1)
                v = (thisc * 4 + 8) >> 4;
              else if (x & 1)
                v = (cx == cw - 1) ? ((thisc * 4 + 7) >> 4) : ((thisc * 3 + col(cx + 1) + 7) >> 4);
              else
                v = (cx == 0) ? ((thisc * 4 + 8) >> 4) : ((thisc * 3 + col(cx - 1) + 8) >> 4);
            } else {  // h1v2: replicate rows (jdsample.c has no fancy filter for it in release 6b)
              v = at(x, y >> 1);
            }
            up[k][(size_t)y * W + x] = (uint8_t)v;
          }
        }
      }
      const int ystride = comp[0].wBlocks * 8;
      auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
      for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
          uint8_t* o = &rgba8[((size_t)y * W + x) * 4];
          const int Y = comp[0].plane[(size_t)y * ystride + x], cb = up[0][(size_t)y * W + x], cr = up[1][(size_t)y * W + x];
          if (!ycc) {
            o[0] = (uint8_t)Y;
            o[1] = (uint8_t)cb;
            o[2] = (uint8_t)cr;
            continue;
          }
          // jdcolor.c build_ycc_rgb_table: SCALEBITS = 16, ONE_HALF = 1 << 15
          const long FIX_1_40200 = 91881, FIX_1_77200 = 116130, FIX_0_71414 = 46802, FIX_0_34414 = 22554, HALF = 32768;
          const int crr = (int)((FIX_1_40200 * (cr - 128) + HALF) >> 16);
          const int cbb = (int)((FIX_1_77200 * (cb - 128) + HALF) >> 16);
          const long crg = -FIX_0_71414 * (cr - 128), cbg = -FIX_0_34414 * (cb - 128) + HALF;
          o[0] = clamp8(Y + crr);
          o[1] = clamp8(Y + (int)((cbg + crg) >> 16));
          o[2] = clamp8(Y + cbb);
        }
      return true;
    }
    p += 2 + len;
  }
  err = "JPEG file has no scan";
  return false;
}


// ---------------------------------------------------------------------------------------------------------------
// Radiance .hdr (RGBE), the format of the reference's light probes (Data/ MonValley_*.hdr; loaded through
// createTextureFromFile -> FreeImage, SharedUtils/ResourceManager.cpp:96-110).  Flat and new-style run-length
// scanlines, "-Y h +X w" (top row first) and "+Y h +X w" (bottom row first) orientations.  RGBE -> float as
// FreeImage's PluginHDR does: mantissa * 2^(e - 136), no half-unit offset; rgba32f: row 0 = top, alpha 1.
// ---------------------------------------------------------------------------------------------------------------
bool decodeHdr(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, std::vector<float>& rgba32f, std::string& err) {
  if (n < 11 || (std::memcmp(d, "#?RADIANCE", 10) != 0 && std::memcmp(d, "#?RGBE", 6) != 0)) {
    err = "not a Radiance .hdr file";
    return false;
  }
  size_t p = 0;
  auto line = [&](std::string& out) {
    out.clear();
    while (p < n && d[p] != '\n') {
      if (out.size() < 512) out.push_back((char)d[p]);
      p++;
    }
    if (p >= n) return false;
    p++;
    if (!out.empty() && out.back() == '\r') out.pop_back();
    return true;
  };
  std::string ln;
  bool formatOk = false;
  for (;;) {  // header lines up to the empty one
    if (!line(ln)) {
      err = "truncated .hdr header";
      return false;
    }
    if (ln.empty()) break;
    if (ln.rfind("FORMAT=", 0) == 0) {
      if (ln != "FORMAT=32-bit_rle_rgbe") {
        err = "unsupported .hdr pixel format (only 32-bit_rle_rgbe)";
        return false;
      }
      formatOk = true;
    }
  }
  (void)formatOk;  // (files without a FORMAT line are RGBE by definition)
  if (!line(ln)) {
    err = "truncated .hdr header";
    return false;
  }
  char sy = 0, sx = 0;
  unsigned long hh = 0, ww = 0;
  if (std::sscanf(ln.c_str(), "%cY %lu %cX %lu", &sy, &hh, &sx, &ww) != 4 || (sy != '-' && sy != '+') || sx != '+' || !hh || !ww) {
    err = "unsupported .hdr orientation (expected -Y h +X w or +Y h +X w)";
    return false;
  }
  if (ww > kMaxImageSide || hh > kMaxImageSide || (uint64_t)ww * hh > kMaxImageTexels) {
    err = ".hdr dimensions exceed the loader's limit";
    return false;
  }
  const uint32_t W = (uint32_t)ww, H = (uint32_t)hh;
  std::vector<uint8_t> row((size_t)W * 4);
  rgba32f.assign((size_t)W * H * 4, 1.0f);
  for (uint32_t y = 0; y < H; y++) {
    if (n - p < 4) {
      err = "truncated .hdr pixel data";
      return false;
    }
    if (W >= 8 && W < 32768 && d[p] == 2 && d[p + 1] == 2 && (((uint32_t)d[p + 2] << 8) | d[p + 3]) == W) {
      p += 4;  // new-style RLE: the four channels of the scanline one after another
      for (int c = 0; c < 4; c++) {
        uint32_t x = 0;
        while (x < W) {
          if (p >= n) {
            err = "truncated .hdr pixel data";
            return false;
          }
          uint32_t count = d[p++];
          if (count > 128) {  // a run
            count -= 128;
            if (!count || x + count > W || p >= n) {
              err = "corrupt .hdr run";
              return false;
            }
            const uint8_t v = d[p++];
            for (uint32_t k = 0; k < count; k++) row[(size_t)(x++) * 4 + (size_t)c] = v;
          } else {  // literals
            if (!count || x + count > W || n - p < count) {
              err = "corrupt .hdr run";
              return false;
            }
            for (uint32_t k = 0; k < count; k++) row[(size_t)(x++) * 4 + (size_t)c] = d[p++];
          }
        }
      }
    } else {
      // flat scanline.  A pixel with mantissas (1, 1, 1) is a legal dark RGBE value AND Radiance's old-style repeat
      // marker; FreeImage 3.17's HDR reader (what the reference decodes probes with) knows only the new-style RLE and
      // takes everything else as flat pixels, so such a pixel is a pixel here too.
      if (n - p < (size_t)W * 4) {
        err = "truncated .hdr pixel data";
        return false;
      }
      std::memcpy(row.data(), d + p, (size_t)W * 4);
      p += (size_t)W * 4;
    }
    const uint32_t oy = (sy == '-') ? y : H - 1 - y;
    float* o = rgba32f.data() + (size_t)oy * W * 4;
    for (uint32_t x = 0; x < W; x++) {
      const uint8_t* q = &row[(size_t)x * 4];
      const float f = q[3] ? std::ldexp(1.0f, (int)q[3] - 136) : 0.0f;
      o[(size_t)x * 4] = (float)q[0] * f;
      o[(size_t)x * 4 + 1] = (float)q[1] * f;
      o[(size_t)x * 4 + 2] = (float)q[2] * f;
    }
  }
  width = W;
  height = H;
  return true;
}

}  // namespace bdpt
