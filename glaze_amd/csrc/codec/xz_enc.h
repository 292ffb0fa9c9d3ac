#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace glz {
// Compresses `size` bytes into one complete .xz stream (LZMA2, CRC64 check).  Returns false only for inputs of
// 4 GiB or more, which the match finder's 32-bit positions do not cover.
bool xz_compress(const uint8_t* data, size_t size, std::vector<uint8_t>& out);
}  // namespace glz
