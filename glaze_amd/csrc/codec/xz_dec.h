#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace glz {
// Decompresses a complete .xz stream (one or more concatenated streams, single LZMA2 filter).
// Returns false and sets err on any structural or checksum error.
bool xz_decompress(const uint8_t* data, size_t size, std::vector<uint8_t>& out, std::string& err);
}  // namespace glz
