#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace glz {
// Baseline / extended-sequential 8-bit JPEG (Huffman, 1 or 3 components, any sampling factors, restart intervals) into
// tightly packed pixels with `want_channels` 1 (luma) or 4 (RGBA, alpha 255).  Progressive and arithmetic-coded
// streams are rejected.  Used by the OBJ converter (the reference loads textures with the `image` crate,
// converter/src/main.rs:478-515; resources/checker.jpg is a baseline 4:2:2 JPEG).
bool jpeg_decode(const uint8_t* data, size_t size, int want_channels, uint32_t& width, uint32_t& height, std::vector<uint8_t>& pixels,
                 std::string& err);
// Baseline JPEG, 4:4:4, Annex K tables scaled by `quality` (1..100, libjpeg's scaling).  `channels` 1, 3 or 4 (alpha dropped).
// Used by glaze-cli for `.jpg` outputs (cli/src/main.rs:121 image.save()).
bool jpeg_encode(const uint8_t* pixels, uint32_t width, uint32_t height, int channels, int quality, std::vector<uint8_t>& out);
}  // namespace glz
