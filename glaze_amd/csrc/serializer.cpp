#include "serializer.h"

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>

#include "codec/png_enc.h"
#include "codec/xxh64.h"
#include "codec/xz_enc.h"

namespace glz {
namespace {
constexpr uint64_t kHasherSeed = 0x368262AAA1DEB64Dull;   // v1.rs:40
constexpr size_t kHeaderLen = 16;                          // parser/mod.rs:12-13
enum { kVertex = 0, kMesh = 1, kCamera = 2, kTexture = 3, kMaterial = 4, kTransform = 5, kInstance = 6, kLight = 7, kMeta = 250 };

void put16(std::vector<uint8_t>& o, uint16_t v) { o.push_back((uint8_t)v); o.push_back((uint8_t)(v >> 8)); }
void put32(std::vector<uint8_t>& o, uint32_t v) { for (int k = 0; k < 4; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void put64(std::vector<uint8_t>& o, uint64_t v) { for (int k = 0; k < 8; ++k) o.push_back((uint8_t)(v >> (8 * k))); }
void putf(std::vector<uint8_t>& o, float f) { uint32_t v; memcpy(&v, &f, 4); put32(o, v); }
void put_name(std::vector<uint8_t>& o, const char* name, size_t cap) { o.insert(o.end(), name, name + strnlen(name, cap)); }

// prepend_hash (v1.rs:426-433)
ChunkBytes with_hash(const std::vector<uint8_t>& body) {
  ChunkBytes out;
  out.reserve(body.size() + 8);
  put64(out, xxh64(body.data(), body.size(), kHasherSeed));
  out.insert(out.end(), body.begin(), body.end());
  return out;
}
// compress (v1.rs:49-57) + prepend_hash; an empty record list gives an empty chunk (encode_fixed / encode_dynamic)
ChunkBytes xz_chunk(const std::vector<uint8_t>& plain) {
  if (plain.empty()) return {};
  std::vector<uint8_t> packed;
  if (!xz_compress(plain.data(), plain.size(), packed)) return {};
  return with_hash(packed);
}
// encode_dynamic framing (v1.rs:506-527): u16 count, then u32 length + bytes per record
struct Dynamic {
  std::vector<uint8_t> bytes;
  explicit Dynamic(uint64_t n) { put16(bytes, (uint16_t)n); }   // `items.len() as u16`
  void add(const std::vector<uint8_t>& rec) {
    put32(bytes, (uint32_t)rec.size());
    bytes.insert(bytes.end(), rec.begin(), rec.end());
  }
};
}  // namespace

ChunkBytes encode_vertices(const glz_vertex* v, uint64_t n) {   // vertex_to_bytes, v1.rs:613-629: 8 little-endian f32
  if (!n) return {};
  std::vector<uint8_t> plain((size_t)n * 32);
  memcpy(plain.data(), v, plain.size());
  return xz_chunk(plain);
}
ChunkBytes encode_meshes(const glz_mesh* m, uint64_t n, const uint32_t* indices, uint64_t n_indices, Error& err) {
  if (!n) return {};
  Dynamic d(n);
  for (uint64_t i = 0; i < n; ++i) {   // mesh_to_bytes, v1.rs:669-681: id u16, index count u32, material u16, indices
    if ((uint64_t)m[i].index_offset + m[i].index_count > n_indices) {
      err.code = GLZ_E_INVALID_INPUT;
      err.msg = "mesh index range outside the index array";
      return {};
    }
    std::vector<uint8_t> rec;
    put16(rec, m[i].id);
    put32(rec, m[i].index_count);
    put16(rec, m[i].material);
    const size_t at = rec.size();
    rec.resize(at + (size_t)m[i].index_count * 4);
    if (m[i].index_count) memcpy(&rec[at], indices + m[i].index_offset, (size_t)m[i].index_count * 4);
    d.add(rec);
  }
  return xz_chunk(d.bytes);
}
ChunkBytes encode_transforms(const glz_transform* t, uint64_t n) {   // transform_to_bytes, v1.rs:700-716: 16 f32, column-major
  if (!n) return {};
  std::vector<uint8_t> plain((size_t)n * 64);
  memcpy(plain.data(), t, plain.size());
  return xz_chunk(plain);
}
ChunkBytes encode_instances(const glz_mesh_instance* i, uint64_t n) {   // instance_to_bytes, v1.rs:730-737: mesh u16, transform u16
  if (!n) return {};
  std::vector<uint8_t> plain((size_t)n * 4);
  memcpy(plain.data(), i, plain.size());
  return xz_chunk(plain);
}
ChunkBytes encode_cameras(const glz_camera* c, uint64_t n) {   // camera_to_bytes, v1.rs:748-770: 49 bytes
  if (!n) return {};
  std::vector<uint8_t> plain;
  plain.reserve((size_t)n * 49);
  for (uint64_t i = 0; i < n; ++i) {
    plain.push_back(c[i].type);
    for (int k = 0; k < 3; ++k) putf(plain, c[i].position[k]);
    for (int k = 0; k < 3; ++k) putf(plain, c[i].target[k]);
    for (int k = 0; k < 3; ++k) putf(plain, c[i].up[k]);
    putf(plain, c[i].fovx_or_scale);
    putf(plain, c[i].near_plane);
    putf(plain, c[i].far_plane);
  }
  return xz_chunk(plain);
}

// Mip chain the way Texture::gen_mipmaps builds it (lib/src/materials/texture.rs:256-277): level k is level k-1 resized to
// half the size (at least 1) with image::imageops::resize(.., FilterType::CatmullRom).  `image` 0.24 is not vendored in the
// reference tree; this restates its published resampler (imageops/sample.rs): a vertical pass into a float image, then a
// horizontal pass; the kernel is the Catmull-Rom cubic (B = 0, C = 1/2) with support 2 stretched by the down-scaling ratio,
// weights normalised per output sample, result clamped to [0, 255] and rounded to nearest.
namespace {
float catmull_rom(float x) {   // bc_cubic_spline(x, 0, 0.5)
  const float a = std::fabs(x);
  if (a < 1.0f) return ((9.0f * a - 15.0f) * a * a + 6.0f) / 6.0f;       // 1.5 a^3 - 2.5 a^2 + 1
  if (a < 2.0f) return (((-3.0f * a + 15.0f) * a - 24.0f) * a + 12.0f) / 6.0f;   // -0.5 a^3 + 2.5 a^2 - 4 a + 2
  return 0.0f;
}
// one axis of the resampler: `n_in` samples with stride `stride_in` -> `n_out` samples, for `lines` independent lines
void resample_axis(const float* in, size_t line_stride_in, size_t stride_in, uint32_t n_in, float* out, size_t line_stride_out, size_t stride_out,
                   uint32_t n_out, size_t lines, int channels) {
  const float ratio = (float)n_in / (float)n_out;
  const float sratio = ratio < 1.0f ? 1.0f : ratio;
  const float src_support = 2.0f * sratio;
  std::vector<float> ws;
  for (uint32_t o = 0; o < n_out; ++o) {
    float centre = ((float)o + 0.5f) * ratio;
    int64_t left = (int64_t)std::floor(centre - src_support);
    left = std::min<int64_t>(std::max<int64_t>(left, 0), (int64_t)n_in - 1);
    int64_t right = (int64_t)std::ceil(centre + src_support);
    right = std::min<int64_t>(std::max<int64_t>(right, left + 1), (int64_t)n_in);
    centre -= 0.5f;
    ws.clear();
    float sum = 0.0f;
    for (int64_t i = left; i < right; ++i) {
      const float w = catmull_rom(((float)i - centre) / sratio);
      ws.push_back(w);
      sum += w;
    }
    for (float& w : ws) w /= sum;
    for (size_t line = 0; line < lines; ++line) {
      const float* src = in + line * line_stride_in + (size_t)left * stride_in;
      float* dst = out + line * line_stride_out + (size_t)o * stride_out;
      for (int c = 0; c < channels; ++c) {
        float t = 0.0f;
        for (size_t k = 0; k < ws.size(); ++k) t += src[k * stride_in + c] * ws[k];
        dst[c] = t;
      }
    }
  }
}
}  // namespace

std::vector<std::vector<uint8_t>> catmull_rom_mips(const uint8_t* level0, uint32_t w, uint32_t h, int channels, unsigned levels) {
  std::vector<std::vector<uint8_t>> out;
  std::vector<uint8_t> prev(level0, level0 + (size_t)w * h * channels);
  uint32_t pw = w, ph = h;
  for (unsigned lvl = 1; lvl < levels && (pw > 1 || ph > 1); ++lvl) {
    const uint32_t nw = std::max(1u, pw >> 1), nh = std::max(1u, ph >> 1);
    std::vector<float> src((size_t)pw * ph * channels), tmp((size_t)pw * nh * channels), dst((size_t)nw * nh * channels);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)prev[i];
    // vertical: every column is a line of ph samples, stride pw * channels
    resample_axis(src.data(), (size_t)channels, (size_t)pw * channels, ph, tmp.data(), (size_t)channels, (size_t)pw * channels, nh, pw, channels);
    // horizontal: every row is a line of pw samples, stride channels
    resample_axis(tmp.data(), (size_t)pw * channels, (size_t)channels, pw, dst.data(), (size_t)nw * channels, (size_t)channels, nw, nh, channels);
    std::vector<uint8_t> cur(dst.size());
    for (size_t i = 0; i < dst.size(); ++i) cur[i] = (uint8_t)std::lround(std::min(255.0f, std::max(0.0f, dst[i])));
    out.push_back(cur);
    prev.swap(cur);
    pw = nw;
    ph = nh;
  }
  return out;
}

ChunkBytes encode_textures(const glz_texture* t, uint64_t n, Error& err) {
  if (!n) return {};
  Dynamic d(n);
  for (uint64_t i = 0; i < n; ++i) {   // texture_to_bytes, v1.rs:773-806
    const glz_texture& tx = t[i];
    const int channels = tx.format == GLZ_TEX_GRAY ? 1 : 4;
    if (tx.format < 1 || tx.format > 3 || !tx.pixels || !tx.width || !tx.height) {
      err.code = GLZ_E_INVALID_INPUT;
      err.msg = "texture without pixels or with an unknown format";
      return {};
    }
    std::vector<uint8_t> rec;
    const size_t nl = strnlen(tx.name, sizeof(tx.name));   // `assert!(str_len < 256)`: glz_texture names are shorter
    rec.push_back((uint8_t)tx.format);
    rec.push_back((uint8_t)nl);
    rec.insert(rec.end(), tx.name, tx.name + nl);
    const unsigned want = tx.mip_levels ? tx.mip_levels : 1;
    const auto mips = catmull_rom_mips(tx.pixels, tx.width, tx.height, channels, want);
    rec.push_back((uint8_t)(1 + mips.size()));
    uint32_t w = tx.width, h = tx.height;
    for (size_t lvl = 0; lvl <= mips.size(); ++lvl) {
      std::vector<uint8_t> png;
      if (!png_encode(lvl == 0 ? tx.pixels : mips[lvl - 1].data(), w, h, channels, png)) {
        err.code = GLZ_E_INVALID_INPUT;
        err.msg = "Failed to encode texture";
        return {};
      }
      put32(rec, (uint32_t)png.size());
      rec.insert(rec.end(), png.begin(), png.end());
      w = std::max(1u, w >> 1);
      h = std::max(1u, h >> 1);
    }
    d.add(rec);
  }
  return with_hash(d.bytes);   // PNGs are already compressed: no xz around the texture list (v1.rs:571)
}

ChunkBytes encode_materials(const glz_material* m, uint64_t n) {
  if (!n) return {};
  Dynamic d(n);
  for (uint64_t i = 0; i < n; ++i) {   // material_to_bytes, v1.rs:884-910
    std::vector<uint8_t> rec;
    rec.push_back(m[i].mtype);
    rec.push_back(m[i].metal);
    rec.insert(rec.end(), m[i].diffuse_mul, m[i].diffuse_mul + 3);
    const uint8_t none[3] = {0, 0, 0};
    const uint8_t* e = m[i].has_emissive ? m[i].emissive_col : none;   // Option<[u8; 3]>: None is stored as 0,0,0
    rec.insert(rec.end(), e, e + 3);
    putf(rec, m[i].ior);
    putf(rec, m[i].roughness_mul);
    putf(rec, m[i].metalness_mul);
    putf(rec, m[i].anisotropy);
    put16(rec, m[i].diffuse);
    put16(rec, m[i].roughness);
    put16(rec, m[i].metalness);
    put16(rec, m[i].normal);
    put16(rec, m[i].opacity);
    put_name(rec, m[i].name, sizeof(m[i].name));
    d.add(rec);
  }
  return xz_chunk(d.bytes);
}
ChunkBytes encode_lights(const glz_light* l, uint64_t n) {
  if (!n) return {};
  Dynamic d(n);
  for (uint64_t i = 0; i < n; ++i) {   // light_to_bytes, v1.rs:973-1007
    std::vector<uint8_t> rec;
    rec.push_back(l[i].ltype);
    for (int k = 0; k < 3; ++k) putf(rec, l[i].position[k]);
    for (int k = 0; k < 3; ++k) putf(rec, l[i].direction[k]);
    put32(rec, l[i].resource_id);
    putf(rec, l[i].intensity);
    putf(rec, l[i].yaw_deg);
    putf(rec, l[i].pitch_deg);
    putf(rec, l[i].roll_deg);
    for (int k = 0; k < 16; ++k) putf(rec, l[i].color[k]);
    put_name(rec, l[i].name, sizeof(l[i].name));
    d.add(rec);
  }
  return xz_chunk(d.bytes);
}
ChunkBytes encode_meta(const glz_meta& m) {   // meta_to_bytes, v1.rs:1049-1062: 5 f32
  std::vector<uint8_t> plain;
  for (int k = 0; k < 3; ++k) putf(plain, m.scene_centre[k]);
  putf(plain, m.scene_radius);
  putf(plain, m.exposure);
  return xz_chunk(plain);
}

bool write_glaze_file(const std::string& path, const std::vector<std::pair<int, ChunkBytes>>& chunks, Error& err) {
  // OffsetsTable::as_bytes (v1.rs:176-194): hash, chunk count, then (id, absolute offset, length) per non-empty chunk
  std::vector<const std::pair<int, ChunkBytes>*> live;
  for (const auto& c : chunks)
    if (!c.second.empty()) live.push_back(&c);
  std::vector<uint8_t> table;
  table.push_back((uint8_t)live.size());
  uint64_t off = kHeaderLen + 8 + 1 + 17 * (uint64_t)live.size();
  for (const auto* c : live) {
    table.push_back((uint8_t)c->first);
    put64(table, off);
    put64(table, c->second.size());
    off += c->second.size();
  }
  std::vector<uint8_t> head(kHeaderLen, 0);
  memcpy(head.data(), "glaze", 5);   // write_header, parser/mod.rs:236-241
  head[5] = 1;
  put64(head, xxh64(table.data(), table.size(), kHasherSeed));
  head.insert(head.end(), table.begin(), table.end());
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) {
    err.code = GLZ_E_IO;
    err.msg = "cannot create " + path;
    return false;
  }
  bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
  for (const auto* c : live) ok = ok && fwrite(c->second.data(), 1, c->second.size(), f) == c->second.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok) {
    err.code = GLZ_E_IO;
    err.msg = "short write on " + path;
  }
  return ok;
}

bool serialize_scene(const std::string& path, const SerializeInput& in, Error& err) {
  std::vector<std::pair<int, ChunkBytes>> chunks;   // same order as ContentV1::serialize (v1.rs:243-275)
  chunks.emplace_back(kVertex, encode_vertices(in.vertices, in.n_vertices));
  chunks.emplace_back(kMesh, encode_meshes(in.meshes, in.n_meshes, in.indices, in.n_indices, err));
  if (err.code != GLZ_OK) return false;
  chunks.emplace_back(kCamera, encode_cameras(in.cameras, in.n_cameras));
  chunks.emplace_back(kTexture, encode_textures(in.textures, in.n_textures, err));
  if (err.code != GLZ_OK) return false;
  chunks.emplace_back(kMaterial, encode_materials(in.materials, in.n_materials));
  chunks.emplace_back(kTransform, encode_transforms(in.transforms, in.n_transforms));
  chunks.emplace_back(kInstance, encode_instances(in.instances, in.n_instances));
  chunks.emplace_back(kLight, encode_lights(in.lights, in.n_lights));
  if (in.meta) chunks.emplace_back(kMeta, encode_meta(*in.meta));
  return write_glaze_file(path, chunks, err);
}

}  // namespace glz
