// JPEG codec for the edges of the hot path: texture input of the OBJ converter and `.jpg` output of glaze-cli.
// Neither is pinned by a reference test at the pixel level (the reference delegates to the `image` crate); the decoder
// follows ITU-T T.81 with libjpeg's conventions for the parts the standard leaves open (triangle-filter chroma
// upsampling for 2:1 ratios, JFIF YCbCr with 16-bit fixed-point constants), the IDCT is evaluated in double precision.
#include "jpeg.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "jpeg_tables.h"

namespace glz {
namespace {
using namespace jpeg;

// ---------------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------------
struct HuffTable {
  bool present = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  uint16_t lookup[512];   // 9-bit prefix -> (length << 8) | value, 0 = longer code
  bool build() {
    int code = 0, k = 0;
    memset(lookup, 0, sizeof(lookup));
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      // Kraft check BEFORE the table is filled: a crafted DHT (e.g. 200 codes of length 2) must not index past lookup[512]
      if (code + (int)bits[l] > (1 << l)) return false;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (l <= 9) {
          const int shift = 9 - l;
          for (int f = 0; f < (1 << shift); ++f) lookup[(code << shift) | f] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    return true;
  }
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int n = 0;
  bool hit_marker = false;
  void fill() {
    while (n <= 24) {
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            p += 2;
          } else {
            hit_marker = true;   // a marker (RSTn / EOI): feed zeros until the caller resynchronises
            b = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= b << (24 - n);
      n += 8;
    }
  }
  uint32_t peek(int k) { if (n < k) fill(); return acc >> (32 - k); }
  void skip(int k) { acc <<= k; n -= k; }
  int receive(int k) {   // k bits, sign-extended per F.2.2.1
    if (k == 0) return 0;
    if (n < k) fill();
    const int v = (int)(acc >> (32 - k));
    skip(k);
    return v < (1 << (k - 1)) ? v - (1 << k) + 1 : v;
  }
  void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int decode_symbol(BitReader& br, const HuffTable& h) {
  if (br.n < 16) br.fill();
  const uint16_t e = h.lookup[br.acc >> 23];
  if (e) {
    br.skip(e >> 8);
    return e & 0xFF;
  }
  int code = (int)(br.acc >> 22), l = 10;   // first 10 bits
  for (; l <= 16; ++l) {
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) break;
    code = (int)(br.acc >> (32 - l - 1));
  }
  if (l > 16) return -1;
  br.skip(l);
  return h.vals[h.valptr[l] + code - h.mincode[l]];
}

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int blocks_w = 0, blocks_h = 0;      // allocated blocks (MCU-padded)
  std::vector<uint8_t> plane;          // blocks_w*8 x blocks_h*8 samples
  int pred = 0;
};

struct IdctTables {
  double c[8][8];
  IdctTables() {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
  }
};
const IdctTables& idct_tables() {
  static const IdctTables t;
  return t;
}
void idct_block(const int* coef, uint8_t* out, int stride) {
  const IdctTables& t = idct_tables();
  double tmp[64];
  for (int u = 0; u < 8; ++u)            // columns of the coefficient block -> rows
    for (int y = 0; y < 8; ++y) {
      double s = 0.0;
      for (int v = 0; v < 8; ++v) s += t.c[y][v] * coef[v * 8 + u];
      tmp[y * 8 + u] = s;
    }
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      double s = 0.0;
      for (int u = 0; u < 8; ++u) s += t.c[x][u] * tmp[y * 8 + u];
      const long r = std::lround(s) + 128;
      out[y * stride + x] = (uint8_t)std::min(255l, std::max(0l, r));
    }
}

// chroma upsampling to full resolution; 2:1 ratios use libjpeg's "fancy" triangle filters, other ratios replicate
void upsample(const Component& c, int hmax, int vmax, uint32_t width, uint32_t height, std::vector<uint8_t>& out) {
  const int pw = c.blocks_w * 8;
  const int hs = hmax / c.h, vs = vmax / c.v;
  const int cw = (int)((width * (uint32_t)c.h + hmax - 1) / hmax), ch = (int)((height * (uint32_t)c.v + vmax - 1) / vmax);   // real component size
  out.resize((size_t)width * height);
  auto at = [&](int x, int y) { return (int)c.plane[(size_t)std::min(std::max(y, 0), ch - 1) * pw + std::min(std::max(x, 0), cw - 1)]; };
  if (hs == 1 && vs == 1) {
    for (uint32_t y = 0; y < height; ++y) memcpy(&out[(size_t)y * width], &c.plane[(size_t)y * pw], width);
  } else if (hs == 2 && vs == 1) {
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x) {
        const int i = (int)(x >> 1);
        out[(size_t)y * width + x] = (x & 1) ? (uint8_t)((3 * at(i, (int)y) + at(i + 1, (int)y) + 2) >> 2) : (uint8_t)((3 * at(i, (int)y) + at(i - 1, (int)y) + 1) >> 2);
      }
  } else if (hs == 2 && vs == 2) {
    for (uint32_t y = 0; y < height; ++y) {
      const int j = (int)(y >> 1), jn = (y & 1) ? j + 1 : j - 1;
      for (uint32_t x = 0; x < width; ++x) {
        const int i = (int)(x >> 1), in = (x & 1) ? i + 1 : i - 1;
        const int near = 3 * at(i, j) + at(i, jn), far = 3 * at(in, j) + at(in, jn);
        out[(size_t)y * width + x] = (uint8_t)((3 * near + far + ((x & 1) ? 7 : 8)) >> 4);
      }
    }
  } else {
    for (uint32_t y = 0; y < height; ++y)
      for (uint32_t x = 0; x < width; ++x) out[(size_t)y * width + x] = (uint8_t)at((int)(x / hs), (int)(y / vs));
  }
}

inline uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }
}  // namespace

bool jpeg_decode(const uint8_t* data, size_t size, int want_channels, uint32_t& width, uint32_t& height, std::vector<uint8_t>& pixels,
                 std::string& err) {
  auto bad = [&](const char* m) { err = std::string("jpeg: ") + m; return false; };
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return bad("not a JPEG stream");
  if (want_channels != 1 && want_channels != 4) return bad("unsupported output format");
  uint16_t quant[4][64];
  bool quant_present[4] = {false, false, false, false};
  HuffTable dc[4], ac[4];
  std::vector<Component> comps;
  int restart_interval = 0, hmax = 1, vmax = 1;
  bool have_frame = false, decoded = false;
  width = height = 0;
  size_t pos = 2;
  while (pos + 4 <= size && !decoded) {
    if (data[pos] != 0xFF) return bad("marker expected");
    const uint8_t m = data[pos + 1];
    if (m == 0xFF) { ++pos; continue; }   // fill byte
    pos += 2;
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) break;
    if (pos + 2 > size) return bad("truncated segment");
    const size_t len = be16(data + pos);
    if (len < 2 || pos + len > size) return bad("truncated segment");
    const uint8_t* seg = data + pos + 2;
    const size_t n = len - 2;
    if (m == 0xDB) {   // DQT
      size_t i = 0;
      while (i < n) {
        const int pq = seg[i] >> 4, tq = seg[i] & 15;
        if (tq > 3 || pq > 1 || i + 1 + 64 * (size_t)(pq + 1) > n) return bad("bad DQT");
        for (int k = 0; k < 64; ++k) quant[tq][kZigZag[k]] = pq ? be16(seg + i + 1 + 2 * k) : seg[i + 1 + k];
        quant_present[tq] = true;
        i += 1 + 64 * (size_t)(pq + 1);
      }
    } else if (m == 0xC4) {   // DHT
      size_t i = 0;
      while (i < n) {
        if (i + 17 > n) return bad("bad DHT");
        const int tc = seg[i] >> 4, th = seg[i] & 15;
        if (tc > 1 || th > 3) return bad("bad DHT");
        HuffTable& h = tc ? ac[th] : dc[th];
        int total = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = seg[i + l]; total += h.bits[l]; }
        if (total > 256 || i + 17 + (size_t)total > n) return bad("bad DHT");
        memcpy(h.vals, seg + i + 17, (size_t)total);
        if (!h.build()) return bad("bad Huffman code lengths");
        h.present = true;
        i += 17 + (size_t)total;
      }
    } else if (m == 0xC0 || m == 0xC1) {   // SOF0 / SOF1
      if (n < 6 || seg[0] != 8) return bad("only 8-bit precision is supported");
      height = be16(seg + 1);
      width = be16(seg + 3);
      const int nc = seg[5];
      if (!width || !height || (nc != 1 && nc != 3) || n < 6 + 3 * (size_t)nc) return bad("unsupported frame header");
      comps.assign((size_t)nc, Component());
      for (int c = 0; c < nc; ++c) {
        comps[c].id = seg[6 + 3 * c];
        comps[c].h = seg[7 + 3 * c] >> 4;
        comps[c].v = seg[7 + 3 * c] & 15;
        comps[c].tq = seg[8 + 3 * c];
        if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) return bad("bad sampling factors");
        hmax = std::max(hmax, comps[c].h);
        vmax = std::max(vmax, comps[c].v);
      }
      // upsample() assumes integer ratios: a 3-in-4 component would copy `width` bytes out of a narrower plane row
      for (int c = 0; c < nc; ++c)
        if (hmax % comps[c].h || vmax % comps[c].v) return bad("fractional sampling ratios are not supported");
      have_frame = true;
    } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return bad("progressive / lossless / arithmetic-coded JPEG is not supported");
    } else if (m == 0xDD) {   // DRI
      if (n < 2) return bad("bad DRI");
      restart_interval = be16(seg);
    } else if (m == 0xDA) {   // SOS: the one scan of a sequential frame
      if (!have_frame) return bad("scan before frame header");
      const int ns = n ? seg[0] : 0;
      if (ns != (int)comps.size() || n < 1 + 2 * (size_t)ns + 3) return bad("non-interleaved scans are not supported");
      for (int s = 0; s < ns; ++s) {
        bool found = false;
        for (auto& c : comps)
          if (c.id == seg[1 + 2 * s]) {
            c.td = seg[2 + 2 * s] >> 4;
            c.ta = seg[2 + 2 * s] & 15;
            found = c.td <= 3 && c.ta <= 3 && dc[c.td].present && ac[c.ta].present && quant_present[c.tq];
          }
        if (!found) return bad("scan refers to a missing table or component");
      }
      const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
      const int mcus_x = (int)((width + mcu_w - 1) / mcu_w), mcus_y = (int)((height + mcu_h - 1) / mcu_h);
      {
        // every coded block takes at least two bits (a DC code and an EOB): a frame header that promises more blocks than the
        // entropy-coded data can hold is corrupt and must not size the planes (65535 x 65535 over a few bytes = 12 GB of zeros)
        uint64_t blocks_per_mcu = 0;
        for (auto& c : comps) blocks_per_mcu += comps.size() == 1 ? 1u : (uint64_t)c.h * c.v;
        const uint64_t left = (uint64_t)(size - std::min(size, pos + len));
        if ((uint64_t)mcus_x * mcus_y * blocks_per_mcu > left * 4u + 64u) return bad("frame size does not match the coded data");
      }
      for (auto& c : comps) {
        if (comps.size() == 1) { c.h = c.v = 1; }
        c.blocks_w = mcus_x * c.h;
        c.blocks_h = mcus_y * c.v;
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
        c.pred = 0;
      }
      if (comps.size() == 1) { hmax = vmax = 1; }
      const int mx = comps.size() == 1 ? (int)((width + 7) / 8) : mcus_x, my = comps.size() == 1 ? (int)((height + 7) / 8) : mcus_y;
      if (comps.size() == 1) {
        comps[0].blocks_w = mx;
        comps[0].blocks_h = my;
        comps[0].plane.assign((size_t)mx * 8 * my * 8, 0);
      }
      BitReader br{data + pos + len, data + size};
      int until_restart = restart_interval, next_rst = 0;
      for (int j = 0; j < my; ++j)
        for (int i = 0; i < mx; ++i) {
          if (restart_interval && until_restart == 0) {
            // resynchronise on RSTn
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
            if (q + 1 >= br.end) return bad("missing restart marker");
            (void)next_rst;
            br.p = q + 2;
            br.reset();
            for (auto& c : comps) c.pred = 0;
            until_restart = restart_interval;
          }
          for (auto& c : comps)
            for (int by = 0; by < c.v; ++by)
              for (int bx = 0; bx < c.h; ++bx) {
                int coef[64] = {0};
                const int t = decode_symbol(br, dc[c.td]);
                if (t < 0 || t > 11) return bad("bad DC code");
                c.pred += br.receive(t);
                coef[0] = c.pred * quant[c.tq][0];
                for (int k = 1; k < 64;) {
                  const int rs = decode_symbol(br, ac[c.ta]);
                  if (rs < 0) return bad("bad AC code");
                  const int r = rs >> 4, s = rs & 15;
                  if (s == 0) {
                    if (r != 15) break;   // EOB
                    k += 16;
                    continue;
                  }
                  k += r;
                  if (k > 63) return bad("AC run past the block");
                  coef[kZigZag[k]] = br.receive(s) * quant[c.tq][kZigZag[k]];
                  ++k;
                }
                const int px = (i * c.h + bx) * 8, py = (j * c.v + by) * 8;
                idct_block(coef, &c.plane[(size_t)py * c.blocks_w * 8 + px], c.blocks_w * 8);
              }
          if (restart_interval) --until_restart;
        }
      decoded = true;
    }
    pos += len;
  }
  if (!decoded) return bad("no image data");
  const size_t npx = (size_t)width * height;
  pixels.resize(npx * (size_t)want_channels);
  std::vector<uint8_t> y, cb, cr;
  upsample(comps[0], hmax, vmax, width, height, y);
  if (comps.size() == 1 || want_channels == 1) {
    for (size_t i = 0; i < npx; ++i) {
      if (want_channels == 1) {
        pixels[i] = y[i];
      } else {
        pixels[4 * i] = pixels[4 * i + 1] = pixels[4 * i + 2] = y[i];
        pixels[4 * i + 3] = 255;
      }
    }
    return true;
  }
  upsample(comps[1], hmax, vmax, width, height, cb);
  upsample(comps[2], hmax, vmax, width, height, cr);
  auto clamp = [](int v) { return (uint8_t)std::min(255, std::max(0, v)); };
  for (size_t i = 0; i < npx; ++i) {   // JFIF YCbCr -> RGB, libjpeg's 16-bit fixed-point constants
    const int Y = y[i], B = cb[i] - 128, R = cr[i] - 128;
    pixels[4 * i] = clamp(Y + ((91881 * R + 32768) >> 16));
    pixels[4 * i + 1] = clamp(Y + ((-22554 * B - 46802 * R + 32768) >> 16));
    pixels[4 * i + 2] = clamp(Y + ((116130 * B + 32768) >> 16));
    pixels[4 * i + 3] = 255;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------------------------
namespace {
struct HuffCodes {
  uint16_t code[256];
  uint8_t len[256];
  HuffCodes(const uint8_t* bits, const uint8_t* vals) {
    memset(len, 0, sizeof(len));
    int code_v = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code_v) {
        code[vals[k]] = (uint16_t)code_v;
        len[vals[k]] = (uint8_t)l;
      }
      code_v <<= 1;
    }
  }
};
struct BitWriter {
  std::vector<uint8_t>& out;
  uint32_t acc = 0;
  int n = 0;
  void put(uint32_t v, int bits) {
    acc = (acc << bits) | (v & ((1u << bits) - 1));
    n += bits;
    while (n >= 8) {
      const uint8_t b = (uint8_t)(acc >> (n - 8));
      out.push_back(b);
      if (b == 0xFF) out.push_back(0x00);
      n -= 8;
    }
  }
  void flush() { if (n) put(0x7F, 8 - n); }
};
void marker(std::vector<uint8_t>& o, uint8_t m, const std::vector<uint8_t>& body) {
  o.push_back(0xFF);
  o.push_back(m);
  const size_t len = body.size() + 2;
  o.push_back((uint8_t)(len >> 8));
  o.push_back((uint8_t)len);
  o.insert(o.end(), body.begin(), body.end());
}
inline int bit_size(int v) { v = v < 0 ? -v : v; int s = 0; while (v) { ++s; v >>= 1; } return s; }
}  // namespace

bool jpeg_encode(const uint8_t* pixels, uint32_t width, uint32_t height, int channels, int quality, std::vector<uint8_t>& out) {
  out.clear();
  if (!width || !height || width > 65535 || height > 65535 || (channels != 1 && channels != 3 && channels != 4)) return false;
  quality = std::min(100, std::max(1, quality));
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;   // libjpeg jpeg_quality_scaling
  uint8_t q[2][64];
  for (int k = 0; k < 64; ++k) {
    q[0][k] = (uint8_t)std::min(255, std::max(1, (kQuantLuma[k] * scale + 50) / 100));
    q[1][k] = (uint8_t)std::min(255, std::max(1, (kQuantChroma[k] * scale + 50) / 100));
  }
  const int nc = channels == 1 ? 1 : 3;
  out.push_back(0xFF); out.push_back(0xD8);
  marker(out, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
  for (int t = 0; t < (nc == 1 ? 1 : 2); ++t) {
    std::vector<uint8_t> b(65);
    b[0] = (uint8_t)t;
    for (int k = 0; k < 64; ++k) b[1 + k] = q[t][kZigZag[k]];
    marker(out, 0xDB, b);
  }
  {
    std::vector<uint8_t> b = {8, (uint8_t)(height >> 8), (uint8_t)height, (uint8_t)(width >> 8), (uint8_t)width, (uint8_t)nc};
    for (int c = 0; c < nc; ++c) { b.push_back((uint8_t)(c + 1)); b.push_back(0x11); b.push_back(c ? 1 : 0); }
    marker(out, 0xC0, b);
  }
  auto dht = [&](uint8_t tc_th, const uint8_t* bits, const uint8_t* vals, int nvals) {
    std::vector<uint8_t> b;
    b.push_back(tc_th);
    b.insert(b.end(), bits, bits + 16);
    b.insert(b.end(), vals, vals + nvals);
    marker(out, 0xC4, b);
  };
  dht(0x00, kBitsDcLuma, kValsDcLuma, sizeof(kValsDcLuma));
  dht(0x10, kBitsAcLuma, kValsAcLuma, sizeof(kValsAcLuma));
  if (nc == 3) {
    dht(0x01, kBitsDcChroma, kValsDcChroma, sizeof(kValsDcChroma));
    dht(0x11, kBitsAcChroma, kValsAcChroma, sizeof(kValsAcChroma));
  }
  {
    std::vector<uint8_t> b = {(uint8_t)nc};
    for (int c = 0; c < nc; ++c) { b.push_back((uint8_t)(c + 1)); b.push_back(c ? 0x11 : 0x00); }
    b.push_back(0); b.push_back(63); b.push_back(0);
    marker(out, 0xDA, b);
  }
  const HuffCodes dcl(kBitsDcLuma, kValsDcLuma), acl(kBitsAcLuma, kValsAcLuma), dcc(kBitsDcChroma, kValsDcChroma), acc_(kBitsAcChroma, kValsAcChroma);
  const IdctTables& t = idct_tables();   // the same orthonormal basis, used forwards
  BitWriter bw{out};
  int pred[3] = {0, 0, 0};
  const uint32_t bxn = (width + 7) / 8, byn = (height + 7) / 8;
  for (uint32_t by = 0; by < byn; ++by)
    for (uint32_t bx = 0; bx < bxn; ++bx)
      for (int c = 0; c < nc; ++c) {
        double blk[64];
        for (int y = 0; y < 8; ++y)
          for (int x = 0; x < 8; ++x) {
            const uint32_t sx = std::min(width - 1, bx * 8 + (uint32_t)x), sy = std::min(height - 1, by * 8 + (uint32_t)y);   // edge replication
            const uint8_t* p = pixels + ((size_t)sy * width + sx) * (size_t)channels;
            double v;
            if (nc == 1) {
              v = p[0];
            } else {
              const double R = p[0], G = p[1], B = p[2];
              v = c == 0 ? 0.299 * R + 0.587 * G + 0.114 * B : (c == 1 ? -0.168735892 * R - 0.331264108 * G + 0.5 * B + 128.0 : 0.5 * R - 0.418687589 * G - 0.081312411 * B + 128.0);
            }
            blk[y * 8 + x] = v - 128.0;
          }
        double tmp[64];
        int coef[64];
        for (int v = 0; v < 8; ++v)
          for (int x = 0; x < 8; ++x) {
            double s = 0.0;
            for (int y = 0; y < 8; ++y) s += t.c[y][v] * blk[y * 8 + x];
            tmp[v * 8 + x] = s;
          }
        const uint8_t* qt = q[c ? 1 : 0];
        for (int v = 0; v < 8; ++v)
          for (int u = 0; u < 8; ++u) {
            double s = 0.0;
            for (int x = 0; x < 8; ++x) s += t.c[x][u] * tmp[v * 8 + x];
            coef[v * 8 + u] = (int)std::lround(s / qt[v * 8 + u]);
          }
        const HuffCodes& hd = c ? dcc : dcl;
        const HuffCodes& ha = c ? acc_ : acl;
        const int diff = coef[0] - pred[c];
        pred[c] = coef[0];
        int s = bit_size(diff);
        bw.put(hd.code[s], hd.len[s]);
        if (s) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), s);
        int run = 0;
        for (int k = 1; k < 64; ++k) {
          const int v = coef[kZigZag[k]];
          if (v == 0) { ++run; continue; }
          while (run > 15) { bw.put(ha.code[0xF0], ha.len[0xF0]); run -= 16; }
          s = bit_size(v);
          bw.put(ha.code[(run << 4) | s], ha.len[(run << 4) | s]);
          bw.put((uint32_t)(v < 0 ? v - 1 : v), s);
          run = 0;
        }
        if (run) bw.put(ha.code[0x00], ha.len[0x00]);
      }
  bw.flush();
  out.push_back(0xFF); out.push_back(0xD9);
  return true;
}

}  // namespace glz
