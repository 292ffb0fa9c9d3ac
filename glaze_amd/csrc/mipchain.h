// Mip chains of scene textures (host side).
//
// The reference uploads the levels a texture brings along when it has ALL of them (`Texture::has_mipmaps`,
// materials/texture.rs:196-221; written by `glaze-converter --gen-mipmaps` with a Catmull-Rom filter) and otherwise generates
// them on the GPU: level l from level l - 1 with vkCmdBlitImage(LINEAR), one after the other (vulkan/scene.rs:1139-1263).
// Level l is max(1, w >> l) x max(1, h >> l); there are 1 + floor(log2(max(w, h))) levels (texture.rs:200-207).
//
// What a LINEAR blit computes is the driver's business ([ext]); the rule stated here is the Vulkan specification's: the centre of a
// destination texel maps to u = (x + 0.5) * (sw / dw) - 0.5 in the source, the value is the bilinear blend of the four texels
// around u with clamp-to-edge, computed in linear light for sRGB formats (colour channels decoded, blended, encoded; alpha
// and the UNORM formats blend as they are) and rounded to the nearest code.  For the 2:1 steps of power-of-two textures this
// is the average of 2 x 2 texels.  The oracle restates the same rule (build_mip_chain in oracle.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "glaze_abi.h"
#include "host_math.h"

namespace glz {
namespace host {

inline uint32_t mip_level_count(uint32_t w, uint32_t h) {
  uint32_t m = std::max(w, h), n = 1;
  while (m > 1) {
    m >>= 1;
    ++n;
  }
  return n;
}
inline uint32_t mip_dim(uint32_t d, uint32_t level) { return std::max(1u, d >> level); }

struct MipLevel {
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> pixels;   // tightly packed rows, 1 (gray) or 4 bytes per pixel
};

// encodes a linear value to an sRGB code with the thresholds of the 8-bit export (srgb8_thresholds): #{k : c >= T_k}
inline uint8_t srgb_encode(const float thr[256], float c) {
  if (!(c > 0.0f)) return 0;
  uint32_t lo = 0, hi = 255;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) >> 1;
    if (c >= thr[mid]) lo = mid; else hi = mid - 1;
  }
  return (uint8_t)lo;
}

// level `dst` from level `src` by the LINEAR-blit rule above
inline void blit_linear(uint32_t format, const MipLevel& src, MipLevel& dst) {
  const uint32_t bpp = format == GLZ_TEX_GRAY ? 1u : 4u;
  dst.pixels.assign((size_t)dst.width * dst.height * bpp, 0);
  double eotf[256];
  float thr[256];
  for (int i = 0; i < 256; ++i) {
    const double c = i / 255.0;
    eotf[i] = c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
  }
  srgb8_thresholds(thr);
  const double rx = (double)src.width / (double)dst.width, ry = (double)src.height / (double)dst.height;
  for (uint32_t y = 0; y < dst.height; ++y) {
    const double sy = ((double)y + 0.5) * ry - 0.5;
    const double fy0 = std::floor(sy);
    const double fy = sy - fy0;
    const long y0 = std::min<long>(std::max<long>((long)fy0, 0), (long)src.height - 1), y1 = std::min<long>(std::max<long>((long)fy0 + 1, 0), (long)src.height - 1);
    for (uint32_t x = 0; x < dst.width; ++x) {
      const double sx = ((double)x + 0.5) * rx - 0.5;
      const double fx0 = std::floor(sx);
      const double fx = sx - fx0;
      const long x0 = std::min<long>(std::max<long>((long)fx0, 0), (long)src.width - 1), x1 = std::min<long>(std::max<long>((long)fx0 + 1, 0), (long)src.width - 1);
      for (uint32_t ch = 0; ch < bpp; ++ch) {
        const bool srgb = format == GLZ_TEX_RGBA_SRGB && ch < 3;
        auto texel = [&](long tx, long ty) {
          const uint8_t v = src.pixels[((size_t)ty * src.width + (size_t)tx) * bpp + ch];
          return srgb ? eotf[v] : (double)v / 255.0;
        };
        const double a = texel(x0, y0), b = texel(x1, y0), c = texel(x0, y1), d = texel(x1, y1);
        const double v = (a * (1.0 - fx) + b * fx) * (1.0 - fy) + (c * (1.0 - fx) + d * fx) * fy;
        uint8_t q;
        if (srgb) {
          q = srgb_encode(thr, (float)v);
        } else {
          const int r = (int)(v * 255.0 + 0.5);
          q = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
        dst.pixels[((size_t)y * dst.width + x) * bpp + ch] = q;
      }
    }
  }
}

// Full chain: `given` holds level 0 and possibly more.  When it holds every level they are used as they are (the file's
// Catmull-Rom levels), otherwise everything past level 0 is generated.
inline std::vector<MipLevel> build_mip_chain(uint32_t format, std::vector<MipLevel> given) {
  const uint32_t n = mip_level_count(given[0].width, given[0].height);
  bool complete = given.size() == n;
  for (uint32_t l = 0; complete && l < n; ++l)
    complete = given[l].width == mip_dim(given[0].width, l) && given[l].height == mip_dim(given[0].height, l) &&
               given[l].pixels.size() == (size_t)given[l].width * given[l].height * (format == GLZ_TEX_GRAY ? 1u : 4u);
  if (complete) return given;
  given.resize(1);
  for (uint32_t l = 1; l < n; ++l) {
    MipLevel m;
    m.width = mip_dim(given[0].width, l);
    m.height = mip_dim(given[0].height, l);
    blit_linear(format, given[l - 1], m);
    given.push_back(std::move(m));
  }
  return given;
}

}  // namespace host
}  // namespace glz
