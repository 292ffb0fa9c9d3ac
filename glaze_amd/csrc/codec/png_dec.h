#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace glz {
// Decodes a non-interlaced PNG into tightly packed 8-bit pixels with `want_channels` (1 = gray,
// 4 = RGBA) per pixel.  Inflate comes from zlib; chunk walking, CRC and un-filtering are done here.
bool png_decode(const uint8_t* data, size_t size, int want_channels, uint32_t& width, uint32_t& height,
                std::vector<uint8_t>& pixels, std::string& err);
// Signature, chunk framing, chunk CRCs and IHDR only (no inflate): true for a file png_decode would not reject on those grounds.
bool png_check(const uint8_t* data, size_t size, uint32_t& width, uint32_t& height, std::string& err);
}  // namespace glz
