#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace glz {
// Encodes tightly packed 8-bit pixels (channels = 1 gray, 3 RGB or 4 RGBA) as a non-interlaced PNG.  Deflate comes from
// zlib; scanline filtering (per-row minimum-sum-of-absolute-differences choice), chunk framing and CRCs are done here.
bool png_encode(const uint8_t* pixels, uint32_t width, uint32_t height, int channels, std::vector<uint8_t>& out);
}  // namespace glz
