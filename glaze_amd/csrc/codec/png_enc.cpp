// PNG writer for the texture chunk of .glaze files (the reference stores every mip level as a PNG,
// lib/src/parser/v1.rs:793-818) and for glaze-cli's image output.
#include "png_enc.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>

namespace glz {
namespace {
void put32be(std::vector<uint8_t>& o, uint32_t v) {
  o.push_back((uint8_t)(v >> 24)); o.push_back((uint8_t)(v >> 16)); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v);
}
void chunk(std::vector<uint8_t>& o, const char* type, const uint8_t* data, size_t n) {
  put32be(o, (uint32_t)n);
  const size_t start = o.size();
  o.insert(o.end(), type, type + 4);
  if (n) o.insert(o.end(), data, data + n);
  put32be(o, (uint32_t)crc32(0L, o.data() + start, (uInt)(n + 4)));
}
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
}  // namespace

bool png_encode(const uint8_t* pixels, uint32_t width, uint32_t height, int channels, std::vector<uint8_t>& out) {
  out.clear();
  if (!width || !height || (channels != 1 && channels != 3 && channels != 4)) return false;
  const size_t bpp = (size_t)channels, stride = (size_t)width * bpp;
  // filter every scanline with the type whose output has the smallest sum of |signed byte|
  std::vector<uint8_t> raw((stride + 1) * height), cand(stride), zero(stride, 0);
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t* cur = pixels + (size_t)y * stride;
    const uint8_t* up = y ? cur - stride : zero.data();
    uint8_t* dst = &raw[(size_t)y * (stride + 1)];
    uint64_t best_cost = ~0ull;
    for (int f = 0; f < 5; ++f) {
      uint64_t cost = 0;
      for (size_t x = 0; x < stride; ++x) {
        const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
        int pred = 0;
        switch (f) {
          case 1: pred = a; break;
          case 2: pred = b; break;
          case 3: pred = (a + b) >> 1; break;
          case 4: pred = paeth(a, b, c); break;
        }
        const uint8_t v = (uint8_t)(cur[x] - pred);
        cand[x] = v;
        cost += v < 128 ? v : 256 - v;
      }
      if (cost < best_cost) {
        best_cost = cost;
        dst[0] = (uint8_t)f;
        memcpy(dst + 1, cand.data(), stride);
      }
    }
  }
  uLongf zlen = compressBound((uLong)raw.size());
  std::vector<uint8_t> z(zlen);
  if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  out.insert(out.end(), sig, sig + 8);
  uint8_t ihdr[13];
  ihdr[0] = (uint8_t)(width >> 24); ihdr[1] = (uint8_t)(width >> 16); ihdr[2] = (uint8_t)(width >> 8); ihdr[3] = (uint8_t)width;
  ihdr[4] = (uint8_t)(height >> 24); ihdr[5] = (uint8_t)(height >> 16); ihdr[6] = (uint8_t)(height >> 8); ihdr[7] = (uint8_t)height;
  ihdr[8] = 8;
  ihdr[9] = channels == 1 ? 0 : (channels == 3 ? 2 : 6);
  ihdr[10] = ihdr[11] = ihdr[12] = 0;
  chunk(out, "IHDR", ihdr, 13);
  chunk(out, "IDAT", z.data(), zlen);
  chunk(out, "IEND", nullptr, 0);
  return true;
}

}  // namespace glz
