// Device-side shading library for the wavefront kernels: 16-bin spectra, RGB<->spectrum, texture
// fetch, shading frames, Fresnel, GGX, the six BSDF families and the four light samplers.
// Replaces the reference's GLSL callables (lib/src/shaders/mat_*.rcall, light_*.rcall) and their
// includes; each function cites the shader lines whose arithmetic (and evaluation order) it keeps.
#pragma once
#include "glaze_abi.h"
#include "glz_tables.h"
#include "math.h"
#include "types.h"

namespace glz {
namespace dev {

// ---------------------------------------------------------------------------------------------
// Spectrum (spectrum.glsl): bin i of the GLSL struct lives in col[i/4][i%4]
// ---------------------------------------------------------------------------------------------
struct Spec {
  float w[16];
};
#define GLZ_BINS _Pragma("unroll") for (int i = 0; i < 16; ++i)

GLZ_D Spec spec_set(float f) { Spec s; GLZ_BINS s.w[i] = f; return s; }
GLZ_D Spec spec_scale(const Spec& a, float f) { Spec r; GLZ_BINS r.w[i] = a.w[i] * f; return r; }
GLZ_D Spec spec_mul(const Spec& a, const Spec& b) { Spec r; GLZ_BINS r.w[i] = a.w[i] * b.w[i]; return r; }
GLZ_D Spec spec_div(const Spec& a, float f) { Spec r; GLZ_BINS r.w[i] = a.w[i] / f; return r; }
GLZ_D Spec spec_load(const Spectrum16& g) { Spec r; GLZ_BINS r.w[i] = g.w[i]; return r; }

__device__ static const float kCieX[16] = GLZ_DEV_CIE_X;
__device__ static const float kCieY[16] = GLZ_DEV_CIE_Y;
__device__ static const float kCieZ[16] = GLZ_DEV_CIE_Z;
__device__ static const float kSurf[7][16] = {GLZ_DEV_SURF_WHITE, GLZ_DEV_SURF_CYAN, GLZ_DEV_SURF_MAGENTA, GLZ_DEV_SURF_YELLOW,
                                              GLZ_DEV_SURF_RED, GLZ_DEV_SURF_GREEN, GLZ_DEV_SURF_BLUE};
__device__ static const float kIllum[7][16] = {GLZ_DEV_ILLUM_WHITE, GLZ_DEV_ILLUM_CYAN, GLZ_DEV_ILLUM_MAGENTA, GLZ_DEV_ILLUM_YELLOW,
                                               GLZ_DEV_ILLUM_RED, GLZ_DEV_ILLUM_GREEN, GLZ_DEV_ILLUM_BLUE};
enum { kWhite = 0, kCyan, kMagenta, kYellow, kRed, kGreen, kBlue };

// (sp.col0*c0 + sp.col1*c1 + sp.col2*c2 + sp.col3*c3) summed .x+.y+.z+.w (spectrum.glsl:45-46, :65-70)
GLZ_D float spec_weighted(const Spec& s, const float* __restrict__ c) {
  float lane[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lane[j] = ((s.w[j] * c[j] + s.w[4 + j] * c[4 + j]) + s.w[8 + j] * c[8 + j]) + s.w[12 + j] * c[12 + j];
  return ((lane[0] + lane[1]) + lane[2]) + lane[3];
}
GLZ_D float spec_luminance(const Spec& s) { return spec_weighted(s, kCieY) * 0.17557178f; }   // spectrum.glsl:39-48
GLZ_D vec3 spec_to_rgb(const Spec& s) {                                                       // spectrum.glsl:50-86
  float X = spec_weighted(s, kCieX) * 0.17557178f;
  float Y = spec_weighted(s, kCieY) * 0.17557178f;
  float Z = spec_weighted(s, kCieZ) * 0.17557178f;
  vec3 r;
  r.x = (3.240479f * X - 1.537150f * Y) - 0.498535f * Z;
  r.y = (-0.969256f * X + 1.875991f * Y) + 0.041556f * Z;
  r.z = (0.055648f * X - 0.204043f * Y) + 1.057311f * Z;
  return r;
}

// GENERATE_COLOR_TO_SPECTRUM (spectrum.glsl:158-200): res = white*lo + A*(mid-lo) + B*(hi-mid), where the
// basis pair (A, B) depends on the ordering of r, g, b.  Unclamped on the device (Q10).
GLZ_D Spec rgb_to_spec(vec3 c, const float (*__restrict__ basis)[16], float scale) {
  int a, b;
  float k0, k1, k2;
  if (c.x <= c.y && c.x <= c.z) {
    k0 = c.x;
    a = kCyan;
    if (c.y <= c.z) { k1 = c.y - c.x; b = kBlue; k2 = c.z - c.y; } else { k1 = c.z - c.x; b = kGreen; k2 = c.y - c.z; }
  } else if (c.y <= c.x && c.y <= c.z) {
    k0 = c.y;
    a = kMagenta;
    if (c.x <= c.z) { k1 = c.x - c.y; b = kBlue; k2 = c.z - c.x; } else { k1 = c.z - c.y; b = kRed; k2 = c.x - c.z; }
  } else {
    k0 = c.z;
    a = kYellow;
    if (c.x <= c.y) { k1 = c.x - c.z; b = kGreen; k2 = c.y - c.x; } else { k1 = c.y - c.z; b = kRed; k2 = c.x - c.y; }
  }
  Spec r;
  GLZ_BINS r.w[i] = ((basis[kWhite][i] * k0 + basis[a][i] * k1) + basis[b][i] * k2) * scale;
  return r;
}
GLZ_D Spec from_surface_color(vec3 c) { return rgb_to_spec(c, kSurf, 0.94f); }        // spectrum.glsl:202-242
GLZ_D Spec from_illuminant_color(vec3 c) { return rgb_to_spec(c, kIllum, 0.86445f); } // spectrum.glsl:244-284

// ---------------------------------------------------------------------------------------------
// Textures: level-0 bilinear, REPEAT, sRGB decode through a 256-entry LUT
// ---------------------------------------------------------------------------------------------
struct vec4 {
  float x, y, z, w;
};
// Texels are stored in 128-byte tiles (one cache line): 8 x 4 RGBA texels or 16 x 8 gray texels, tiles row-major.  The 2 x 2
// footprint of a bilinear fetch then touches 1.4 lines on average instead of the 2.06 of a row-major image (k_shade sits
// at the memory system's random-access rate; with every texel fetch removed it ran 30 % faster).  A 1 x 1 texture (the
// default roughness / metalness / opacity maps of most materials) carries its texel in the descriptor: no texel load at all.
constexpr uint32_t kTexInline = 0x80u;   // TexDesc.format flag: `offset` holds the single RGBA8 / gray texel
GLZ_D vec4 decode_texel(const DeviceScene& S, uint32_t format, uint32_t bits) {
  if (format == GLZ_TEX_GRAY) return vec4{(float)(bits & 0xFFu) / 255.0f, 0.0f, 0.0f, 1.0f};
  const uint32_t r = bits & 0xFFu, g = (bits >> 8) & 0xFFu, b = (bits >> 16) & 0xFFu, a = bits >> 24;
  if (format == GLZ_TEX_RGBA_SRGB) return vec4{S.srgb_lut[r], S.srgb_lut[g], S.srgb_lut[b], (float)a / 255.0f};
  return vec4{(float)r / 255.0f, (float)g / 255.0f, (float)b / 255.0f, (float)a / 255.0f};
}
GLZ_D uint32_t texel_address(const TexDesc& t, uint32_t format, uint32_t ux, uint32_t uy) {
  const uint32_t tiles_x = t.format >> 8;
  if (format == GLZ_TEX_GRAY) return t.offset + ((uy >> 3) * tiles_x + (ux >> 4)) * 128u + (uy & 7u) * 16u + (ux & 15u);
  return t.offset + (((uy >> 2) * tiles_x + (ux >> 3)) * 32u + (uy & 3u) * 8u + (ux & 7u)) * 4u;
}
GLZ_D uint32_t load_texel(const uint8_t* __restrict__ pool, const TexDesc& t, uint32_t format, uint32_t ux, uint32_t uy) {
  const uint32_t addr = texel_address(t, format, ux, uy);
  return format == GLZ_TEX_GRAY ? (uint32_t)pool[addr] : *reinterpret_cast<const uint32_t*>(pool + addr);
}
GLZ_D int wrap_coord(int i, int n) {
  int r = i % n;
  return r < 0 ? r + n : r;
}
GLZ_D float lerp_ab(float a, float b, float t) { return a + (b - a) * t; }
// bilinear, REPEAT fetch from ONE level: `t` describes it, `pool` holds its texels
GLZ_D vec4 bilinear_level(const DeviceScene& S, const TexDesc& t, const uint8_t* __restrict__ pool, float u, float v) {
  const uint32_t format = t.format & 0x7Fu;
  float fu = u * (float)t.width - 0.5f, fv = v * (float)t.height - 0.5f;
  float iu = glz_floorf(fu), iv = glz_floorf(fv);
  float ax = fu - iu, ay = fv - iv;
  uint32_t ta, tb, tc, tdx;
  if (t.format & kTexInline) {
    ta = tb = tc = tdx = t.offset;   // same arithmetic below, so NaN coordinates still give what four equal texels give
  } else {
    if (S.tex_counter) {   // counting pass: one fetch = four texels (the thread's own tallies, flushed once per wave at the kernel's end)
      S.tex_counter[0] += 1ull;
      S.tex_counter[1] += format == GLZ_TEX_GRAY ? 4ull : 16ull;
    }
    const int x0 = wrap_coord((int)iu, (int)t.width), y0 = wrap_coord((int)iv, (int)t.height);
    const int x1 = wrap_coord((int)iu + 1, (int)t.width), y1 = wrap_coord((int)iv + 1, (int)t.height);
    if (format != GLZ_TEX_GRAY && x1 == x0 + 1 && (x0 & 7) != 7) {
      // the two texels of a row are neighbours inside one tile row: one 8-byte load per row
      const uint2 r0 = *reinterpret_cast<const uint2*>(pool + texel_address(t, format, (uint32_t)x0, (uint32_t)y0));
      const uint2 r1 = *reinterpret_cast<const uint2*>(pool + texel_address(t, format, (uint32_t)x0, (uint32_t)y1));
      ta = r0.x; tb = r0.y; tc = r1.x; tdx = r1.y;
    } else {
      ta = load_texel(pool, t, format, (uint32_t)x0, (uint32_t)y0);
      tb = load_texel(pool, t, format, (uint32_t)x1, (uint32_t)y0);
      tc = load_texel(pool, t, format, (uint32_t)x0, (uint32_t)y1);
      tdx = load_texel(pool, t, format, (uint32_t)x1, (uint32_t)y1);
    }
  }
  const vec4 a = decode_texel(S, format, ta), b = decode_texel(S, format, tb), c = decode_texel(S, format, tc), d = decode_texel(S, format, tdx);
  vec4 r;
  r.x = lerp_ab(lerp_ab(a.x, b.x, ax), lerp_ab(c.x, d.x, ax), ay);
  r.y = lerp_ab(lerp_ab(a.y, b.y, ax), lerp_ab(c.y, d.y, ax), ay);
  r.z = lerp_ab(lerp_ab(a.z, b.z, ax), lerp_ab(c.z, d.z, ax), ay);
  r.w = lerp_ab(lerp_ab(a.w, b.w, ax), lerp_ab(c.w, d.w, ax), ay);
  return r;
}
GLZ_D vec4 texture2d(const DeviceScene& S, uint32_t id, float u, float v) {
  // one dwordx4 for the whole descriptor
  const uint4 td = reinterpret_cast<const uint4*>(S.tex_desc)[id];
  return bilinear_level(S, TexDesc{td.x, td.y, td.z, td.w}, S.tex_pool, u, v);
}
// Texture level of detail (build-defined, FrameData::lod_mode; the reference's ray-tracing stages sample level 0).  `lod_base` is
// the texture-independent part of the ray-cone level, 0.5 log2(uv area / world area * cone width^2 / cos^2); the texture adds
// 0.5 log2(width * height).  The level is clamped to the chain and the two nearest levels are blended (LINEAR mipmap mode of the
// reference's sampler, scene.rs:716-749).  kNoLod, no mip chain on the device, or a level <= 0: exactly texture2d().
// taps > 1 (lod mode 2, anisotropic footprint): that many trilinear probes spread evenly over the footprint's long axis (du, dv)
// around (u, v), averaged -- what the anisotropy of the reference's sampler does in its raster viewer.
constexpr float kNoLod = -1e30f;
struct TexFootprint { float lod_base; float du, dv; uint32_t taps; };
GLZ_D vec4 texture2d_lod(const DeviceScene& S, uint32_t id, float u, float v, const TexFootprint& fp) {
  const uint4 td = reinterpret_cast<const uint4*>(S.tex_desc)[id];
  const TexDesc t0{td.x, td.y, td.z, td.w};
  uint32_t first = 0, l0 = 0, levels_used = 1, probes = 1;
  float frac = 0.0f;
  if (fp.lod_base > -1e29f && S.tex_mip_base != nullptr && !(t0.format & kTexInline)) {
    const uint32_t b = S.tex_mip_base[id];
    const uint32_t levels = b >> 24;
    first = b & 0xFFFFFFu;
    float lam = fp.lod_base + 0.5f * glz_log2f((float)t0.width * (float)t0.height);
    lam = lam > 0.0f ? lam : 0.0f;                                   // NaN -> 0
    const float top = (float)(levels - 1u);
    lam = lam < top ? lam : top;
    const float fl = glz_floorf(lam);
    l0 = (uint32_t)fl;
    frac = lam - fl;
    levels_used = frac > 0.0f ? 2u : 1u;
    probes = fp.taps;
  }
  vec4 sum{0.0f, 0.0f, 0.0f, 0.0f}, cur{0.0f, 0.0f, 0.0f, 0.0f};
  const uint32_t fetches = probes * levels_used;
#pragma unroll 1
  for (uint32_t k = 0; k < fetches; ++k) {   // one copy of the fetch code: per probe level l0, then l0 + 1 when the level is fractional
    const uint32_t probe = levels_used == 2u ? k >> 1 : k, upper = levels_used == 2u ? k & 1u : 0u;
    const uint32_t level = l0 + upper;
    TexDesc t = t0;
    const uint8_t* pool = S.tex_pool;
    if (level != 0u) {
      const uint4 md = reinterpret_cast<const uint4*>(S.tex_mip_desc)[first + level - 1u];
      t = TexDesc{md.x, md.y, md.z, md.w};
      pool = S.tex_mip_pool;
    }
    float uu = u, vv = v;
    if (probes > 1u) {
      const float s = ((float)probe + 0.5f) / (float)probes - 0.5f;
      uu = u + s * fp.du;
      vv = v + s * fp.dv;
    }
    const vec4 r = bilinear_level(S, t, pool, uu, vv);
    if (upper == 0u) {
      cur = r;
    } else {
      cur.x = lerp_ab(cur.x, r.x, frac); cur.y = lerp_ab(cur.y, r.y, frac); cur.z = lerp_ab(cur.z, r.z, frac); cur.w = lerp_ab(cur.w, r.w, frac);
    }
    if (upper + 1u == levels_used) {
      if (probe == 0u) {
        sum = cur;
      } else {
        sum.x += cur.x; sum.y += cur.y; sum.z += cur.z; sum.w += cur.w;
      }
    }
  }
  if (probes > 1u) {
    const float n = (float)probes;
    sum.x = sum.x / n; sum.y = sum.y / n; sum.z = sum.z / n; sum.w = sum.w / n;
  }
  return sum;
}
GLZ_D vec3 texture_rgb(const DeviceScene& S, uint32_t id, vec2 uv) {
  vec4 t = texture2d(S, id, uv.x, uv.y);
  return mk3(t.x, t.y, t.z);
}
GLZ_D float texture_r(const DeviceScene& S, uint32_t id, vec2 uv) { return texture2d(S, id, uv.x, uv.y).x; }

// ---------------------------------------------------------------------------------------------
// Shading frame (shading_space.glsl)
// ---------------------------------------------------------------------------------------------
struct Frame {
  vec3 s, t, n;
};
GLZ_D Frame make_frame(vec3 dpdu, vec3 n) {   // :11-16
  Frame f;
  f.s = normalize3(dpdu - n * dot3(n, dpdu));
  f.t = cross3(n, f.s);
  f.n = n;
  return f;
}
GLZ_D vec3 to_world(vec3 v, const Frame& f) {   // :18-24 (normalises)
  return normalize3(mk3((f.s.x * v.x + f.t.x * v.y) + f.n.x * v.z, (f.s.y * v.x + f.t.y * v.y) + f.n.y * v.z,
                        (f.s.z * v.x + f.t.z * v.y) + f.n.z * v.z));
}
GLZ_D vec3 to_local(vec3 w, const Frame& f) {   // :26-30 (normalises)
  return normalize3(mk3(dot3(w, f.s), dot3(w, f.t), dot3(w, f.n)));
}

// ---------------------------------------------------------------------------------------------
// Fresnel (fresnel.glsl)
// ---------------------------------------------------------------------------------------------
GLZ_D Spec fresnel_conductor(float c, const Spectrum16& ior, const Spectrum16& ior2abs2) {   // :7-17
  const float c2 = c * c, twoc = c * 2.0f;
  Spec r;
  GLZ_BINS {
    float e = ior.w[i] * twoc;
    float ep = e + c2, epp = e + 1.0f;
    float a = ior2abs2.w[i];
    float perp = (a - ep) / (a + ep);
    float tmp = a * c2;
    float par = (tmp - epp) / (tmp + epp);
    r.w[i] = (perp + par) / 2.0f;
  }
  return r;
}
GLZ_D float fresnel_dielectric(float costi, float etai, float etat) {   // :19-35
  float sin2ti = gl_max(0.0f, 1.0f - costi * costi);
  float sin2tt = etai * etai / (etat * etat) * sin2ti;
  if (sin2tt >= 1.0f) return 1.0f;
  float costt = sqrtf(gl_max(0.0f, 1.0f - sin2tt));
  float tt = etat * costt, ti = etat * costi, ii = etai * costi, it = etai * costt;
  float rparl = (ti - it) / (ti + it);
  float rperp = (ii - tt) / (ii + tt);
  return (rparl * rparl + rperp * rperp) / 2.0f;
}

// ---------------------------------------------------------------------------------------------
// GGX (microfacets.glsl)
// ---------------------------------------------------------------------------------------------
GLZ_D vec2 ggx_sample_p22(float cost, vec2 r) {   // :23-55
  if (cost > 0.999f) {
    float rr = sqrtf(r.x / (1.0f - r.x));
    float phi = kTwoPi * r.y;
    return vec2{rr * glz_cosf(phi), rr * glz_sinf(phi)};
  }
  float cos2t = cost * cost;
  float sin2t = gl_max(0.0f, 1.0f - cos2t);
  float tan2t = check_inf(sin2t / cos2t);
  float tant = sqrtf(tan2t);
  float a2 = 1.0f / tan2t;
  float G1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / a2));
  float A = 2.0f * r.x / G1 - 1.0f;
  float B = tant;
  float invA2m1 = 1.0f / (A * A - 1.0f);
  float sq = sqrtf(gl_max(0.0f, B * B * invA2m1 * invA2m1 - (A * A - B * B) * invA2m1));
  float sx1 = B * invA2m1 - sq;
  float sx2 = B * invA2m1 + sq;
  float sx = (A < 0.0f || sx2 > 1.0f / tant) ? sx1 : sx2;
  float stepval = gl_step(0.5f, r.y);
  float s = gl_mix(1.0f, -1.0f, stepval);
  float u = gl_mix(2.0f * (r.y - 0.5f), 2.0f * (0.5f - r.y), stepval);
  float z = (u * (u * (u * -0.3657289f + 0.7902350f) - 0.4249658f) + 0.0001529f) /
            (u * (u * (u * (u * 0.1695078f - 0.3972035f) - 0.2325005f) + 1.0f) - 0.5398259f);
  float sy = s * z * sqrtf(1.0f + sx * sx);
  return vec2{sx, sy};
}
GLZ_D float ggx_d(vec3 wh, vec2 a) {   // :57-69
  float cos2t = wh.z * wh.z;
  float cos4t = cos2t * cos2t;
  float sin2t = gl_max(0.0f, 1.0f - cos2t);
  float tan2t = sin2t / cos2t;
  float cos2p = wh.x * wh.x / sin2t;
  float sin2p = wh.y * wh.y / sin2t;
  float e1 = 1.0f + ((cos2p / (a.x * a.x) + sin2p / (a.y * a.y)) * tan2t);
  float d = 1.0f / (kPi * a.x * a.y * cos4t * e1 * e1);
  return isinf(tan2t) ? 0.0f : d;
}
GLZ_D float ggx_lambda(vec3 v, vec2 a) {   // :71-82
  float cos2t = v.z * v.z;
  float sin2t = gl_max(0.0f, 1.0f - cos2t);
  float tan2t = sin2t / cos2t;
  float cos2p = gl_max(0.0f, v.x * v.x / sin2t);
  float sin2p = gl_max(0.0f, v.y * v.y / sin2t);
  float alpha2 = cos2p * a.x * a.x + sin2p * a.y * a.y;
  float lambda = (-1.0f + sqrtf(1.0f + tan2t * alpha2)) * 0.5f;
  return isinf(tan2t) ? 0.0f : lambda;
}
GLZ_D float ggx_g(vec3 wo, vec3 wi, vec2 a) { return 1.0f / (1.0f + ggx_lambda(wo, a) + ggx_lambda(wi, a)); }   // :84-87
GLZ_D float ggx_pdf(float d, vec2 a, vec3 wo, vec3 wh) {   // :94-99, G1 of wh (Q6)
  return d * (1.0f / (1.0f + ggx_lambda(wh, a))) * fabsf(dot3(wo, wh)) / fabsf(wh.z);
}
GLZ_D vec3 ggx_sample_wh(vec3 wo, vec2 r, vec2 a) {   // :102-120
  float flip = gl_sign(wo.z);
  vec3 wi = flip * wo;
  vec3 ws = normalize3(mk3(wi.x * a.x, wi.y * a.y, wi.z));
  float cost = ws.z;
  vec2 slope = ggx_sample_p22(cost, r);
  float sin2t = gl_max(0.0f, 1.0f - cost * cost);
  float cosp = sqrtf(ws.x * ws.x / sin2t);
  float sinp = sqrtf(ws.y * ws.y / sin2t);
  float sx = cosp * slope.x - sinp * slope.y;
  float sy = sinp * slope.x + cosp * slope.y;
  return flip * normalize3(mk3(-a.x * sx, -a.y * sy, 1.0f));
}
GLZ_D vec2 anisotropic_alpha(float a, float anis) { return vec2{a * (1.0f + anis), a * (1.0f - anis)}; }   // :122-125

// ---------------------------------------------------------------------------------------------
// BSDFs.  One switch on RTMaterial::bsdf_index replaces the SBT callable dispatch
// (executeCallableEXT(material.bsdf_index [+1]), path_trace.rgen:103, :218).
// ---------------------------------------------------------------------------------------------
// The scalar part of an RTMaterial, fetched with four dwordx4 loads (bytes 0..15 and 160..207 of the 208-byte record)
// instead of one load per field; the two 64-byte metal spectra stay in memory and are read only by conductor lobes.
struct MatScalars {
  float diffuse_mul[3];
  uint32_t diffuse, roughness, metalness, opacity, normal, bsdf_index;
  float roughness_mul, metalness_mul, anisotropy, ior_dielectric;
  uint32_t is_specular;
  const RTMaterial* spectra;   // metal_ior / metal_fresnel
};
GLZ_D MatScalars load_material(const RTMaterial* m) {
  const float4* q = reinterpret_cast<const float4*>(m);
  const float4 a = q[0], b = q[10], c = q[11], d = q[12];
  MatScalars r;
  r.diffuse_mul[0] = a.x; r.diffuse_mul[1] = a.y; r.diffuse_mul[2] = a.z;
  r.diffuse = __float_as_uint(b.x); r.roughness = __float_as_uint(b.y); r.metalness = __float_as_uint(b.z); r.opacity = __float_as_uint(b.w);
  r.normal = __float_as_uint(c.x); r.bsdf_index = __float_as_uint(c.y); r.roughness_mul = c.z; r.metalness_mul = c.w;
  r.anisotropy = d.x; r.ior_dielectric = d.y; r.is_specular = __float_as_uint(d.z);
  r.spectra = m;
  return r;
}

struct SurfacePoint {
  vec3 woW;
  vec2 uv;
  Frame frame;
  MatScalars mat;
  // texture values of this hit, fetched once (fetch_material_textures) for both the NEE evaluation and the BSDF sample --
  // the reference's value and sample_value callables each call texture() for themselves, with the same uv and result
  vec3 tint;           // texture(diffuse).rgb * diffuse_mul      (Lambert, Uber)
  float rough_tex;     // texture(roughness).r                     (Metal, Frosted, Uber)
  float metal_tex;     // texture(metalness).r                     (Uber)
};

GLZ_D void dielectric_etas(const MatScalars& m, float woz, float& etai, float& etat) {
  float outside = gl_step(0.0f, woz);
  etai = gl_mix(m.ior_dielectric, kDefaultIor, outside);
  etat = gl_mix(kDefaultIor, m.ior_dielectric, outside);
}

// reflective microfacet terms shared by Frosted and Uber (mat_frosted_value.rcall:35-47, mat_uber_value.rcall:39-52)
struct Lobe {
  float d, g, pdf, costi, cwo, cwi;
};
GLZ_D Lobe reflect_lobe(vec3 wo, vec3 wi, vec3 wh, vec2 a) {
  Lobe L;
  float owh = dot3(wo, wh), iwh = dot3(wi, wh);
  L.costi = dot3(wi, gl_sign(dot3(wh, mk3(0.0f, 0.0f, 1.0f))) * wh);
  L.cwo = fabsf(wo.z);
  L.cwi = fabsf(wi.z);
  L.d = gl_step(0.0f, wo.z) * ggx_d(wh, a);
  L.g = gl_step(0.0f, owh) * gl_step(0.0f, iwh) * ggx_g(wo, wi, a);
  L.pdf = ggx_pdf(L.d, a, wo, wh) / (4.0f * owh);
  return L;
}

// Oren-Nayar factor of the Uber diffuse lobe (mat_uber_value.rcall:59-73)
GLZ_D float oren_nayar(float roughness, vec3 wo, vec3 wi) {
  float sigma = roughness * 0.5f;
  float s2 = sigma * sigma;
  float A = 1.0f - s2 / (2.0f * (s2 + 0.33f));
  float B = 0.45f * s2 / (s2 + 0.09f);
  float sinto = sqrtf(gl_max(0.0f, 1.0f - wo.z * wo.z));
  float sinti = sqrtf(gl_max(0.0f, 1.0f - wi.z * wi.z));
  float sinpi = wi.y / sinti, cospi = wi.x / sinti;
  float sinpo = wo.y / sinto, cospo = wo.x / sinto;
  float maxcos = gl_max(0.0f, cospi * cospo + sinpi * sinpo);
  float sel = gl_step(fabsf(wo.z), fabsf(wi.z));
  float sinalpha = gl_mix(sinto, sinti, sel);
  float tanbeta = gl_mix(sinti / fabsf(wi.z), sinto / fabsf(wo.z), sel);
  return kInvPi * (A + B * maxcos * sinalpha * tanbeta);
}

GLZ_D void fetch_material_textures(const DeviceScene& S, SurfacePoint& P, const TexFootprint& fp) {
  const uint32_t kind = P.mat.bsdf_index;
  P.tint = mk3(0.0f, 0.0f, 0.0f);
  P.rough_tex = P.metal_tex = 0.0f;
  if (kind == kBsdfLambert || kind == kBsdfUber) {
    const vec4 tx = texture2d_lod(S, P.mat.diffuse, P.uv.x, P.uv.y, fp);
    P.tint = mk3(tx.x, tx.y, tx.z) * mk3(P.mat.diffuse_mul[0], P.mat.diffuse_mul[1], P.mat.diffuse_mul[2]);
  }
  if (kind == kBsdfMetal || kind == kBsdfFrosted || kind == kBsdfUber) P.rough_tex = texture2d_lod(S, P.mat.roughness, P.uv.x, P.uv.y, fp).x;
  if (kind == kBsdfUber) P.metal_tex = texture2d_lod(S, P.mat.metalness, P.uv.x, P.uv.y, fp).x;
}
GLZ_D vec3 diffuse_tint(const DeviceScene&, const SurfacePoint& P) { return P.tint; }

// BSDF evaluation for next-event estimation; returns the pdf (0 = no contribution).
GLZ_D float bsdf_eval(const DeviceScene& S, const SurfacePoint& P, vec3 wiW, float xi, Spec& value) {
  const MatScalars& m = P.mat;
  const uint32_t kind = m.bsdf_index;
  if (kind == kBsdfMirror || kind == kBsdfGlass) return 0.0f;   // mat_mirror_value.rcall:8-11, mat_glass_value.rcall:8-11
  const vec3 wo = to_local(P.woW, P.frame), wi = to_local(wiW, P.frame);
  if (kind == kBsdfLambert) {   // mat_lambert_value.rcall:23-34
    float same = gl_step(0.0f, wo.z * wi.z);
    value = from_surface_color(diffuse_tint(S, P) * kInvPi);
    return same * fabsf(wi.z) * kInvPi;
  }
  if (kind == kBsdfMetal) {   // mat_metal_value.rcall:19-44
    vec3 wh = normalize3(wo + wi);
    if (!(wo.z * wi.z > 0.0f)) return 0.0f;
    Spec F = fresnel_conductor(dot3(wi, wh), m.spectra->metal_ior, m.spectra->metal_fresnel);
    vec2 a = anisotropic_alpha(P.rough_tex * m.roughness_mul, m.anisotropy);
    float d = ggx_d(wh, a);
    float g = ggx_g(wo, wi, a);
    float term = d * g / (4.0f * fabsf(wo.z) * fabsf(wi.z));
    float pdf = ggx_pdf(d, a, wo, wh) / (4.0f * dot3(wo, wh));
    value = spec_scale(F, term);
    return check_nan(pdf);
  }
  if (kind == kBsdfFrosted) {   // mat_frosted_value.rcall:19-66
    vec2 a = anisotropic_alpha(P.rough_tex * m.roughness_mul, m.anisotropy);
    float etai, etat;
    dielectric_etas(m, wo.z, etai, etat);
    float eta = etai / etat;
    if (wo.z * wi.z > 0.0f) {
      vec3 wh = normalize3(wo + wi);
      Lobe L = reflect_lobe(wo, wi, wh, a);
      float f = fresnel_dielectric(L.costi, etai, etat);
      value = spec_set(L.d * L.g * f / (4.0f * L.cwo * L.cwi));
      return check_nan(L.pdf);
    }
    vec3 wh = normalize3(wo + eta * wi);
    wh = wh * gl_sign(wo.z);
    float owh = dot3(wo, wh), iwh = dot3(wi, wh);
    float f = fresnel_dielectric(owh, etai, etat);
    float cwo = fabsf(wo.z), cwi = fabsf(wi.z);
    float denom = owh + eta * iwh;
    float d = ggx_d(wh, a);
    float g = ggx_g(wo, wi, a);
    float pdf = ggx_pdf(d, a, wo, wh) * fabsf(eta * eta * iwh) / (denom * denom);
    value = spec_set(d * g * (1.0f - f) * fabsf(iwh) * fabsf(owh) / (denom * denom * cwo * cwi));
    return check_nan(pdf);
  }
  // Uber: mat_uber_value.rcall:20-77
  float roughness = P.rough_tex * m.roughness_mul;
  float same = gl_step(0.0f, wo.z * wi.z);
  if (xi < 0.5f) {
    vec2 a = anisotropic_alpha(roughness * m.roughness_mul, m.anisotropy);   // roughness_mul applied twice (Q5)
    vec3 wh = normalize3(wo + wi);
    float metalness = P.metal_tex * m.metalness_mul;
    float etai, etat;
    dielectric_etas(m, wo.z, etai, etat);
    Lobe L = reflect_lobe(wo, wi, wh, a);
    float fd = fresnel_dielectric(L.costi, etai, etat);
    Spec fc = fresnel_conductor(L.costi, m.spectra->metal_ior, m.spectra->metal_fresnel);
    float term = L.d * L.g / (4.0f * L.cwo * L.cwi);
    GLZ_BINS value.w[i] = gl_mix(fd, fc.w[i], metalness) * term;
    return check_nan(same * 0.5f * L.pdf);
  }
  value = from_surface_color(diffuse_tint(S, P) * oren_nayar(roughness, wo, wi));
  return check_nan(same * 0.5f * fabsf(wi.z) * kInvPi);
}

GLZ_D vec3 cosine_hemisphere(float rx, float ry, float woz) {   // mat_lambert_sample_value.rcall:19-29
  float t = kTwoPi * rx;
  float r = sqrtf(ry);
  vec3 wi;
  wi.x = r * glz_cosf(t);
  wi.y = r * glz_sinf(t);
  wi.z = sqrtf(gl_max(0.0f, 1.0f - wi.x * wi.x - wi.y * wi.y));
  wi.z *= gl_sign(woz);
  return wi;
}

// BSDF sampling for the path continuation; returns the pdf (0 = terminate the path).
GLZ_D float bsdf_sample(const DeviceScene& S, const SurfacePoint& P, vec3 xi, Spec& value, vec3& wiW) {
  const MatScalars& m = P.mat;
  const uint32_t kind = m.bsdf_index;
  const vec3 wo = to_local(P.woW, P.frame);
  if (kind == kBsdfLambert) {   // mat_lambert_sample_value.rcall:31-41
    vec3 wi = cosine_hemisphere(xi.x, xi.y, wo.z);
    wiW = normalize3(to_world(wi, P.frame));
    value = from_surface_color(diffuse_tint(S, P) * kInvPi);
    return fabsf(wi.z) * kInvPi;
  }
  if (kind == kBsdfMirror) {   // mat_mirror_sample_value.rcall:16-34
    Spec F = fresnel_conductor(wo.z, m.spectra->metal_ior, m.spectra->metal_fresnel);
    wiW = normalize3(to_world(mk3(-wo.x, -wo.y, wo.z), P.frame));
    value = spec_div(F, fabsf(wo.z));
    return 1.0f;
  }
  if (kind == kBsdfGlass) {   // mat_glass_sample_value.rcall:34-56
    float outside = gl_step(0.0f, wo.z);
    float etai = gl_mix(m.ior_dielectric, kDefaultIor, outside);
    float etat = gl_mix(kDefaultIor, m.ior_dielectric, outside);
    float costi = gl_mix(fabsf(wo.z), wo.z, outside);
    float F = fresnel_dielectric(costi, etai, etat);
    vec3 wi;
    float pdf, e;
    if (xi.z < F) {
      wi = mk3(-wo.x, -wo.y, wo.z);
      e = F / fabsf(wi.z);
      pdf = F;
    } else {
      wi = gl_refract(wo, mk3(0.0f, 0.0f, gl_sign(wo.z)), etai / etat);   // wo is not the incident vector GLSL expects (Q7)
      e = (1.0f - F) * (etai * etai) / (etat * etat * fabsf(wi.z));
      pdf = 1.0f - F;
    }
    value = spec_set(1.0f * e);
    wiW = to_world(wi, P.frame);
    return pdf;
  }
  if (kind == kBsdfMetal) {   // mat_metal_sample_value.rcall:21-49
    vec2 a = anisotropic_alpha(P.rough_tex * m.roughness_mul, m.anisotropy);
    vec3 wh = normalize3(ggx_sample_wh(wo, vec2{xi.x, xi.y}, a));
    vec3 wi = -normalize3(gl_reflect(wo, wh));
    if (!(wo.z * wi.z > 0.0f)) return 0.0f;
    float d = ggx_d(wh, a);
    float g = ggx_g(wo, wi, a);
    float gp = ggx_pdf(d, a, wo, wh);
    Spec F = fresnel_conductor(dot3(wi, wh), m.spectra->metal_ior, m.spectra->metal_fresnel);
    float term = d * g / (4.0f * fabsf(wo.z) * fabsf(wi.z));
    float pdf = gp / (4.0f * dot3(wo, wh));
    value = spec_scale(F, term);
    wiW = normalize3(to_world(wi, P.frame));
    return check_nan(pdf);
  }
  if (kind == kBsdfFrosted) {   // mat_frosted_sample_value.rcall:21-71
    vec2 a = anisotropic_alpha(P.rough_tex * m.roughness_mul, m.anisotropy);
    vec3 wh = normalize3(ggx_sample_wh(wo, vec2{xi.x, xi.y}, a));
    float etai, etat;
    dielectric_etas(m, wo.z, etai, etat);
    float eta = etai / etat;
    vec3 wi;
    float pdf;
    if (xi.z < 0.5f) {
      wi = -normalize3(gl_reflect(wo, wh));
      Lobe L = reflect_lobe(wo, wi, wh, a);
      float f = fresnel_dielectric(L.costi, etai, etat);
      value = spec_set(L.d * L.g * f / (4.0f * L.cwo * L.cwi));
      pdf = check_nan(0.5f * L.pdf);
    } else {
      wi = normalize3(gl_refract(wo, wh, eta));
      float owh = dot3(wo, wh), iwh = dot3(wi, wh);
      float f = fresnel_dielectric(owh, etai, etat);
      float cwo = fabsf(wo.z), cwi = fabsf(wi.z);
      float denom = owh + eta * iwh;
      float d = ggx_d(wh, a);
      float g = ggx_g(wo, wi, a);
      float p = ggx_pdf(d, a, wo, wh) * fabsf(eta * eta * iwh) / (denom * denom);
      value = spec_set(d * g * (1.0f - f) * fabsf(iwh) * fabsf(owh) / (denom * denom * cwo * cwi));
      pdf = owh * iwh < 0.0f ? check_nan(0.5f * p) : 0.0f;
    }
    wiW = normalize3(to_world(wi, P.frame));
    return pdf;
  }
  // Uber: mat_uber_sample_value.rcall:21-86
  float roughness = P.rough_tex * m.roughness_mul;
  vec3 wi;
  float pdf;
  if (xi.z < 0.5f) {
    vec2 a = anisotropic_alpha(roughness * m.roughness_mul, m.anisotropy);
    vec3 wh = normalize3(ggx_sample_wh(wo, vec2{xi.x, xi.y}, a));
    float metalness = P.metal_tex * m.metalness_mul;
    float etai, etat;
    dielectric_etas(m, wo.z, etai, etat);
    wi = -normalize3(gl_reflect(wo, wh));
    Lobe L = reflect_lobe(wo, wi, wh, a);
    float fd = fresnel_dielectric(L.costi, etai, etat);
    Spec fc = fresnel_conductor(L.costi, m.spectra->metal_ior, m.spectra->metal_fresnel);
    float term = L.d * L.g / (4.0f * L.cwo * L.cwi);
    GLZ_BINS value.w[i] = gl_mix(fd, fc.w[i], metalness) * term;
    pdf = check_nan(0.5f * L.pdf);
  } else {
    wi = cosine_hemisphere(xi.x, xi.y, wo.z);
    value = from_surface_color(diffuse_tint(S, P) * oren_nayar(roughness, wo, wi));
    pdf = 0.5f * fabsf(wi.z) * kInvPi;
  }
  wiW = normalize3(to_world(wi, P.frame));
  return pdf;
}

// ---------------------------------------------------------------------------------------------
// Lights (light_*_sample_visible.rcall)
// ---------------------------------------------------------------------------------------------
// The emitted spectrum is not part of the sample: 16 registers that would only wait through the whole BSDF evaluation.  The sample
// keeps what the spectrum is made from (light_emission below makes it, after the evaluation, with the operations the callables use).
struct LightSample {
  vec3 wiW;
  float pdf;
  float distance;
  uint32_t kind;    // RTLight::shader of the light
  const RTLight* light;
  float k;          // omni, area: squared distance / intensity
  vec3 rgb;         // area: the emitter's diffuse_mul; sky: texel * intensity
};

GLZ_D float sq_dist(vec3 a, vec3 b) {
  return ((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)) + (a.z - b.z) * (a.z - b.z);
}

// Piecewise-constant CDF lookup used by both sky searches.  `fetch(i)` returns cdf[i].
// (sample_marginal / sample_conditional, light_sky_sample_visible.rcall:31-98)
template <class Fetch>
GLZ_D float sample_cdf(float xi, int size, Fetch fetch, uint32_t& offset_out) {
  int first = 0, len = size;
  while (len > 0) {
    int half = len >> 1, middle = first + half;
    if (fetch(middle) <= xi) {
      first = middle + 1;
      len -= half + 1;
    } else {
      len = half;
    }
  }
  int off = first - 1;
  off = off < 0 ? 0 : (off > size - 2 ? size - 2 : off);
  float cur = fetch(off), next = fetch(off + 1);
  float du = xi - cur;
  if (next - cur > 0.0f) du /= next - cur;
  offset_out = (uint32_t)off;
  return ((float)off + du) / (float)size;
}

GLZ_D void sample_light(const DeviceScene& S, uint32_t light_index, vec3 p, vec3 xi, float scene_radius, LightSample& out) {
  const RTLight* L = &S.lights[light_index];
  const uint32_t kind = L->shader;
  if (S.tex_counter) {   // counting pass
    if (kind == kLightOmni || kind == kLightSun || kind == kLightArea) S.tex_counter[2] += 1ull; else S.tex_counter[3] += 1ull;
  }
  out.kind = kind;
  out.light = L;
  out.k = 1.0f;
  out.rgb = mk3(0.0f, 0.0f, 0.0f);
  if (kind == kLightOmni) {   // light_omni_sample_visible.rcall:14-25
    vec3 lp = mk3(L->pos[0], L->pos[1], L->pos[2]);
    out.wiW = normalize3(lp - p);
    float d2 = sq_dist(lp, p);
    out.distance = sqrtf(d2);
    out.pdf = 1.0f;
    out.k = d2 / L->intensity;
    return;
  }
  if (kind == kLightSun) {   // light_sun_sample_visible.rcall:22-29; dir is not normalised (Q14)
    out.wiW = mk3(-L->dir[0], -L->dir[1], -L->dir[2]);
    out.pdf = 1.0f;
    out.distance = 2.0f * scene_radius + 1.0f;
    return;
  }
  if (kind == kLightArea) {   // light_area_sample_visible.rcall:29-64
    const RTInstance in = S.instances[L->instance_id];
    const uint32_t ntri = in.index_count / 3u;
    uint32_t tri = (uint32_t)gl_min(xi.x * (float)in.index_count / 3.0f, (float)(ntri - 1u));
    tri += in.index_offset / 3u;
    const uint32_t i0 = S.indices[tri * 3u], i1 = S.indices[tri * 3u + 1u], i2 = S.indices[tri * 3u + 2u];
    const float4 a = S.vertices[2u * i0], b = S.vertices[2u * i1], c = S.vertices[2u * i2];
    const float area = 0.5f * 3.0f;   // `.length()` of a vec3 is its component count (Q1)
    float su = sqrtf(xi.y);
    float ru = 1.0f - su, rv = xi.z * su;
    vec3 rp = (ru * mk3(a.x, a.y, a.z) + rv * mk3(b.x, b.y, b.z)) + (1.0f - ru - rv) * mk3(c.x, c.y, c.z);
    rp = xform_point(S.transforms[in.transform_id].o2w, rp);
    out.wiW = normalize3(p - rp);   // points away from the light (Q2)
    float d2 = sq_dist(rp, p);
    out.distance = sqrtf(d2);
    const RTMaterial* m = &S.materials[in.material_id];
    out.rgb = mk3(m->diffuse_mul[0], m->diffuse_mul[1], m->diffuse_mul[2]);
    out.k = d2 / L->intensity;
    out.pdf = (1.0f / (float)ntri) * (1.0f / area);
    return;
  }
  // Sky: light_sky_sample_visible.rcall:100-135
  uint32_t row, col;
  const float* marg = S.sky_marginal;
  const float* cdf = S.sky_cdf;   // (k_shade: in LDS -- the search is ten dependent reads; the two values read once below come from memory)
  float v = sample_cdf(xi.y, (int)S.sky_header.marginal_cdf_count, [&](int i) { return cdf[i]; }, row);
  float v_pdf = marg[S.sky_header.marginal_cdf_count + row] / S.sky_header.marginal_integral;
  // The conditional lookups hand integer texel coordinates to a normalised REPEAT/NEAREST sampler: every
  // fetch lands on texel (0,0) of the image (Q3).
  const float cdf00 = S.sky_cond_cdf[0];
  float u = sample_cdf(xi.x, (int)S.sky_header.conditional_cdf_count, [&](int) { return cdf00; }, col);
  float u_pdf = S.sky_cond_values[0] / marg[S.sky_header.conditional_integral_offset + row];
  float pdf = u_pdf * v_pdf;
  float theta = v * kPi;
  float sint = glz_sinf(theta);
  if (pdf > 0.0f && sint != 0.0f) {
    float phi = u * kTwoPi;
    float cost = glz_cosf(theta), cosp = glz_cosf(phi), sinp = glz_sinf(phi);
    out.pdf = pdf / (2.0f * kPi * kPi * sint);
    out.wiW = normalize3(xform_dir(S.sky.obj2world, mk3(sint * cosp, sint * sinp, cost)));
    out.distance = 2.0f * scene_radius + 1.0f;
    out.rgb = texture_rgb(S, S.sky.tex_id, vec2{u, v}) * S.sky.intensity;
  } else {
    out.pdf = 0.0f;
  }
}

// the spectrum a light sample carries (the `emission` the *_sample_visible callables return)
GLZ_D Spec light_emission(const LightSample& ls) {
  Spec e;
  if (ls.kind == kLightOmni) {
    GLZ_BINS e.w[i] = ls.light->color.w[i] / ls.k;
  } else if (ls.kind == kLightSun) {
    GLZ_BINS e.w[i] = ls.light->color.w[i] * ls.light->intensity;
  } else if (ls.kind == kLightArea) {
    e = spec_div(from_surface_color(ls.rgb), ls.k);
  } else {
    e = from_illuminant_color(ls.rgb);
  }
  return e;
}

}  // namespace dev
}  // namespace glz
