// PNG reader for the texture chunk of .glaze V1 (lib/src/parser/v1.rs:840-847 decodes each mip
// with the `image` crate's PngDecoder).  The build image has zlib.h but no png.h, so
// the container, CRC and scanline filters (PNG spec 1.2, sections 5, 6 and 9) are done here and
// only DEFLATE is delegated to zlib.
#include "png_dec.h"

#include <zlib.h>

#include <cstdlib>
#include <cstring>

namespace glz {
namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  if (pa <= pb && pa <= pc) return a;
  return pb <= pc ? b : c;
}
}  // namespace

// The container without the pixels: signature, chunk framing, every chunk's CRC, IHDR -- what catches a damaged file at a fraction
// of the cost of inflating and un-filtering it (the parser checks the mip levels it does not need yet this way).
bool png_check(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::string& err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (size < 8 || memcmp(data, sig, 8) != 0) { err = "png: bad signature"; return false; }
  size_t pos = 8;
  bool have_ihdr = false, have_iend = false, have_idat = false;
  while (pos + 12 <= size && !have_iend) {
    const uint32_t len = be32(data + pos);
    const uint8_t* type = data + pos + 4;
    if (pos + 12 + (size_t)len > size) { err = "png: truncated chunk"; return false; }
    const uint8_t* body = data + pos + 8;
    if ((uint32_t)::crc32(::crc32(0, Z_NULL, 0), type, 4 + len) != be32(body + len)) { err = "png: chunk CRC mismatch"; return false; }
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) { err = "png: bad IHDR"; return false; }
      width = be32(body);
      height = be32(body + 4);
      if (body[10] != 0 || body[11] != 0) { err = "png: unknown compression/filter method"; return false; }
      if (body[12] != 0) { err = "png: interlaced images are not supported"; return false; }
      {   // the colour types and bit depths png_decode takes: a level that could not be decoded later is rejected at parse, like level 0
        const unsigned depth = body[8], ctype = body[9];
        if (!(ctype == 0 || ctype == 2 || ctype == 3 || ctype == 4 || ctype == 6)) { err = "png: bad colour type"; return false; }
        if (!(depth == 8 || depth == 16) || (ctype == 3 && depth != 8)) { err = "png: unsupported bit depth"; return false; }
      }
      have_ihdr = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      have_idat = true;
    } else if (!memcmp(type, "IEND", 4)) {
      have_iend = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || !have_iend || !have_idat) { err = "png: missing IHDR/IDAT/IEND"; return false; }
  if (width == 0 || height == 0 || width > 65535 || height > 65535) { err = "png: bad dimensions"; return false; }
  return true;
}

bool png_decode(const uint8_t* data, size_t size, int want, uint32_t& width, uint32_t& height,
                std::vector<uint8_t>& pixels, std::string& err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (size < 8 || memcmp(data, sig, 8) != 0) { err = "png: bad signature"; return false; }
  if (want != 1 && want != 4) { err = "png: bad channel request"; return false; }
  size_t pos = 8;
  bool have_ihdr = false, have_iend = false;
  unsigned depth = 0, ctype = 0;
  std::vector<uint8_t> idat, plte, trns;
  while (pos + 12 <= size && !have_iend) {
    uint32_t len = be32(data + pos);
    const uint8_t* type = data + pos + 4;
    if (pos + 12 + (size_t)len > size) { err = "png: truncated chunk"; return false; }
    const uint8_t* body = data + pos + 8;
    uint32_t crc = (uint32_t)::crc32(::crc32(0, Z_NULL, 0), type, 4 + len);
    if (crc != be32(body + len)) { err = "png: chunk CRC mismatch"; return false; }
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) { err = "png: bad IHDR"; return false; }
      width = be32(body);
      height = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      if (body[10] != 0 || body[11] != 0) { err = "png: unknown compression/filter method"; return false; }
      if (body[12] != 0) { err = "png: interlaced images are not supported"; return false; }
      have_ihdr = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(body, body + len);
    } else if (!memcmp(type, "tRNS", 4)) {
      trns.assign(body, body + len);
    } else if (!memcmp(type, "IEND", 4)) {
      have_iend = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || !have_iend || idat.empty()) { err = "png: missing IHDR/IDAT/IEND"; return false; }
  if (width == 0 || height == 0 || width > 65535 || height > 65535) { err = "png: bad dimensions"; return false; }
  unsigned channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "png: bad colour type"; return false;
  }
  if (!(depth == 8 || depth == 16) || (ctype == 3 && depth != 8)) { err = "png: unsupported bit depth"; return false; }
  const size_t bpp = channels * depth / 8;          // bytes per complete pixel (filter unit)
  const size_t stride = (size_t)width * bpp;
  // A deflate stream expands at most ~1032x: an IHDR that asks for more than the IDAT can possibly hold is corrupt, and
  // must not be allowed to allocate (a 100-byte file could otherwise request 34 GB)
  const uint64_t raw_bytes = ((uint64_t)stride + 1) * height;
  if (raw_bytes > (uint64_t)idat.size() * 1040u + 1024u) { err = "png: image size does not match the compressed data"; return false; }
  std::vector<uint8_t> raw((size_t)raw_bytes);
  uLongf raw_len = (uLongf)raw.size();
  int zr = ::uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || raw_len != raw.size()) { err = "png: inflate failed"; return false; }
  // un-filter in place, row by row
  std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < height; ++y) {
    uint8_t* row = raw.data() + (stride + 1) * y;
    const uint8_t ft = row[0];
    uint8_t* cur = row + 1;
    const uint8_t* up = y ? raw.data() + (stride + 1) * (y - 1) + 1 : zero.data();
    switch (ft) {
      case 0: break;
      case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
      case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]); break;
      case 3:
        for (size_t i = 0; i < stride; ++i) {
          int a = i >= bpp ? cur[i - bpp] : 0;
          cur[i] = (uint8_t)(cur[i] + ((a + up[i]) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < stride; ++i) {
          int a = i >= bpp ? cur[i - bpp] : 0, c = i >= bpp ? up[i - bpp] : 0;
          cur[i] = (uint8_t)(cur[i] + paeth(a, up[i], c));
        }
        break;
      default: err = "png: bad filter type"; return false;
    }
  }
  // convert to the requested layout
  pixels.assign((size_t)width * height * want, 0);
  const size_t step = depth / 8;  // 16-bit samples: keep the most significant byte
  for (uint32_t y = 0; y < height; ++y) {
    const uint8_t* src = raw.data() + (stride + 1) * y + 1;
    uint8_t* dst = pixels.data() + (size_t)y * width * want;
    for (uint32_t x = 0; x < width; ++x) {
      uint8_t r, g, b, a = 255;
      const uint8_t* s = src + (size_t)x * bpp;
      switch (ctype) {
        case 0: r = g = b = s[0]; break;
        case 2: r = s[0]; g = s[step]; b = s[2 * step]; break;
        case 3: {
          unsigned i = s[0];
          if ((size_t)i * 3 + 2 >= plte.size()) { err = "png: palette index out of range"; return false; }
          r = plte[i * 3]; g = plte[i * 3 + 1]; b = plte[i * 3 + 2];
          if (i < trns.size()) a = trns[i];
          break;
        }
        case 4: r = g = b = s[0]; a = s[step]; break;
        default: r = s[0]; g = s[step]; b = s[2 * step]; a = s[3 * step]; break;
      }
      if (want == 4) { dst[x * 4] = r; dst[x * 4 + 1] = g; dst[x * 4 + 2] = b; dst[x * 4 + 3] = a; }
      else if (ctype == 0 || ctype == 4) dst[x] = r;
      else dst[x] = (uint8_t)((r * 2126u + g * 7152u + b * 722u + 5000u) / 10000u);
    }
  }
  return true;
}

}  // namespace glz
