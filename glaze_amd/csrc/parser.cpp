// .glaze V1 reader.  See parser.h for the reference items this mirrors.
#include "parser.h"
#include "serializer.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

#include "codec/png_dec.h"
#include "codec/xxh64.h"
#include "codec/xz_dec.h"

namespace glz {

bool TextureData::decode_more_levels(std::string& err) {
  if (more_levels.size() == more_png.size()) return true;
  std::vector<std::vector<uint8_t>> levels(more_png.size());
  for (size_t l = 0; l < more_png.size(); ++l) {
    uint32_t w = 0, h = 0;
    if (!png_decode(more_png[l].data(), more_png[l].size(), info.format == GLZ_TEX_GRAY ? 1 : 4, w, h, levels[l], err)) return false;
    if (w != more_dims[2 * l] || h != more_dims[2 * l + 1]) {
      err = "png: level dimensions changed";
      return false;
    }
  }
  more_levels = std::move(levels);
  // the encoded levels are not needed again (a scene replica copies the decoded ones): a LOD scene does not hold both on the host.
  // (What png_check cannot see at parse -- a damaged deflate stream inside intact chunks -- is reported here, i.e. by
  // glz_renderer_set_texture_lod or the first launch that wants the chain: GLZ_E_INVALID_DATA.)
  for (auto& e : more_png) std::vector<uint8_t>().swap(e);
  return true;
}

namespace {

constexpr uint64_t kHasherSeed = 0x368262AAA1DEB64Dull;  // parser/v1.rs:41
constexpr size_t kHeaderLen = 16;                        // parser/mod.rs:13
constexpr size_t kHashSize = 8;
const uint8_t kMagic[5] = {0x67, 0x6C, 0x61, 0x7A, 0x65};  // "glaze", parser/mod.rs:12

enum ChunkId { kVertex = 0, kMesh = 1, kCamera = 2, kTexture = 3, kMaterial = 4, kTransform = 5, kInstance = 6, kLight = 7, kMeta = 250 };

bool known_chunk(unsigned id) { return id <= 7 || id == 250; }

uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
float rdf(const uint8_t* p) { float v; memcpy(&v, p, 4); return v; }

bool fail(Error& e, int code, const std::string& msg) {
  e.code = code;
  e.msg = msg;
  return false;
}

void copy_name(char* dst, const uint8_t* src, size_t n) {
  if (n > GLZ_NAME_MAX - 1) n = GLZ_NAME_MAX - 1;
  memcpy(dst, src, n);
  dst[n] = 0;
}

// Splits the payload of a "dynamic" chunk (u16 count, then u32 len || bytes per item), v1.rs:527-556.
bool split_dynamic(const std::vector<uint8_t>& p, const char* what, std::vector<std::pair<const uint8_t*, size_t>>& items, Error& err) {
  if (p.size() < 2) return fail(err, GLZ_E_INVALID_DATA, std::string("Corrupted chunk: ") + what);
  size_t idx = 2;
  while (idx < p.size()) {
    if (idx + 4 > p.size()) return fail(err, GLZ_E_INVALID_DATA, std::string("Corrupted chunk: ") + what);
    size_t len = rd32(&p[idx]);
    idx += 4;
    if (idx + len > p.size()) return fail(err, GLZ_E_INVALID_DATA, std::string("Corrupted chunk: ") + what);
    items.emplace_back(&p[idx], len);
    idx += len;
  }
  return true;
}

}  // namespace

glz_camera default_camera() {
  glz_camera c{};
  c.type = GLZ_CAMERA_PERSPECTIVE;
  c.target[2] = 100.0f;
  c.up[1] = 1.0f;
  c.fovx_or_scale = 90.0f * (3.14159265358979323846f / 180.0f);  // f32::to_radians(90.0)
  c.near_plane = 1e-3f;
  c.far_plane = 1e3f;
  return c;
}

glz_meta default_meta() {
  glz_meta m{};
  m.scene_radius = 100.0f;
  m.exposure = 1.0f;
  return m;
}

glz_material default_material() {
  glz_material m{};
  m.mtype = GLZ_MAT_LAMBERT;
  m.metal = 0;
  m.diffuse_mul[0] = m.diffuse_mul[1] = m.diffuse_mul[2] = 255;
  m.ior = 1.46f;
  m.roughness_mul = 1.0f;
  m.metalness_mul = 0.0f;
  strcpy(m.name, "default");
  return m;
}

glz_transform identity_transform() {
  glz_transform t{};
  t.m[0] = t.m[5] = t.m[10] = t.m[15] = 1.0f;
  return t;
}

TextureData default_texture() {
  TextureData t;
  t.info.format = GLZ_TEX_RGBA_SRGB;
  t.info.width = t.info.height = 1;
  t.info.mip_levels = 1;
  strcpy(t.info.name, "default");
  t.level0.assign(4, 255);
  return t;
}

std::unique_ptr<Parsed> Parsed::open(const std::string& path, Error& err) {
  std::unique_ptr<Parsed> p(new Parsed());
  if (!p->load(path, err)) return nullptr;
  return p;
}

bool Parsed::load(const std::string& path, Error& err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(err, GLZ_E_IO, "cannot open " + path);
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint8_t> d(sz > 0 ? (size_t)sz : 0);
  size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), f);
  fclose(f);
  if (got != d.size()) return fail(err, GLZ_E_IO, "short read on " + path);
  // parse(): header (mod.rs:93-116)
  if (d.size() < kHeaderLen || memcmp(d.data(), kMagic, 5) != 0) return fail(err, GLZ_E_INVALID_INPUT, "Wrong or empty input file");
  if (d[5] != 1) return fail(err, GLZ_E_INVALID_INPUT, "Unsupported file version");
  // OffsetsTable::seek_and_parse (v1.rs:135-175)
  if (d.size() < kHeaderLen + kHashSize + 1) return fail(err, GLZ_E_IO, "failed to fill whole buffer");
  const uint64_t expected = rd64(&d[kHeaderLen]);
  const size_t n = d[kHeaderLen + kHashSize];
  size_t table_len = 1 + n * 17;
  const uint8_t* table = &d[kHeaderLen + kHashSize];
  if (kHeaderLen + kHashSize + table_len > d.size()) table_len = d.size() - kHeaderLen - kHashSize;  // take().read_to_end()
  if (xxh64(table, table_len, kHasherSeed) != expected) return fail(err, GLZ_E_INVALID_DATA, "Corrupted file structure");
  Slot slots[256];
  for (size_t i = 0; i < n && 1 + 17 * (i + 1) <= table_len; ++i) {
    const uint8_t* e = table + 1 + 17 * i;
    unsigned id = e[0];
    if (!known_chunk(id)) continue;  // unknown chunks are ignored (v1.rs:160-162)
    slots[id].off = rd64(e + 1);
    slots[id].len = rd64(e + 9);
    slots[id].present = true;
  }
  // commit: file image, offsets, and every cached getter result is dropped
  path_ = path;
  file_.swap(d);
  for (int i = 0; i < 256; ++i) slots_[i] = slots[i];
  vertices_ = {}; meshes_ = {}; indices_.clear(); transforms_ = {}; instances_ = {}; cameras_ = {}; materials_ = {}; lights_ = {};
  textures_ = {};
  return true;
}

bool Parsed::update(const Update& u, Error& err) {
  auto raw = [&](int id, ChunkBytes& out) {   // read_chunk: the stored bytes (hash + body), empty when absent
    const Slot& s = slots_[id];
    out.clear();
    if (!s.present || s.len == 0) return true;
    if (s.off > file_.size() || s.len > file_.size() - s.off) return fail(err, GLZ_E_IO, "failed to fill whole buffer");
    out.assign(file_.begin() + s.off, file_.begin() + s.off + s.len);
    return true;
  };
  ChunkBytes vertices, meshes, transforms, instances, meta, cameras, materials, lights, textures;
  if (!raw(kVertex, vertices) || !raw(kMesh, meshes) || !raw(kTransform, transforms) || !raw(kInstance, instances)) return false;
  if (u.meta) meta = encode_meta(*u.meta); else if (!raw(kMeta, meta)) return false;
  if (u.n_cameras >= 0) cameras = encode_cameras(u.cameras, (uint64_t)u.n_cameras); else if (!raw(kCamera, cameras)) return false;
  if (u.n_materials >= 0) materials = encode_materials(u.materials, (uint64_t)u.n_materials); else if (!raw(kMaterial, materials)) return false;
  if (u.n_lights >= 0) lights = encode_lights(u.lights, (uint64_t)u.n_lights); else if (!raw(kLight, lights)) return false;
  if (u.n_textures >= 0) {
    textures = encode_textures(u.textures, (uint64_t)u.n_textures, err);
    if (err.code != GLZ_OK) return false;
  } else if (!raw(kTexture, textures)) {
    return false;
  }
  std::vector<std::pair<int, ChunkBytes>> chunks;   // order of v1.rs:403-413
  chunks.emplace_back(kVertex, std::move(vertices));
  chunks.emplace_back(kMesh, std::move(meshes));
  chunks.emplace_back(kCamera, std::move(cameras));
  chunks.emplace_back(kTexture, std::move(textures));
  chunks.emplace_back(kMaterial, std::move(materials));
  chunks.emplace_back(kTransform, std::move(transforms));
  chunks.emplace_back(kInstance, std::move(instances));
  chunks.emplace_back(kLight, std::move(lights));
  chunks.emplace_back(kMeta, std::move(meta));
  const std::string path = path_;
  if (!write_glaze_file(path, chunks, err)) return false;
  return load(path, err);   // reopen + OffsetsTable::seek_and_parse (v1.rs:415-419)
}

// read_chunk + verify_hash (+ decompress), v1.rs:298-313, :437-449, :59-67
bool Parsed::chunk_payload(int id, const char* what, bool xz, std::vector<uint8_t>& payload, bool& present, Error& err) {
  const Slot& s = slots_[id];
  present = s.present && s.len > 0;
  if (!present) return true;
  if (s.off > file_.size() || s.len > file_.size() - s.off) return fail(err, GLZ_E_IO, "failed to fill whole buffer");
  if (s.len < kHashSize) return fail(err, GLZ_E_INVALID_DATA, std::string("Corrupted ") + what);
  const uint8_t* c = &file_[s.off];
  if (xxh64(c + kHashSize, s.len - kHashSize, kHasherSeed) != rd64(c)) return fail(err, GLZ_E_INVALID_DATA, std::string("Corrupted ") + what);
  if (!xz) {
    payload.assign(c + kHashSize, c + s.len);
    return true;
  }
  std::string xerr;
  if (!xz_decompress(c + kHashSize, s.len - kHashSize, payload, xerr))
    return fail(err, GLZ_E_INVALID_DATA, std::string("Failed to decompress ") + what + ": " + xerr);
  return true;
}

bool Parsed::vertices(const std::vector<glz_vertex>*& out, Error& err) {
  if (!vertices_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kVertex, "Vertex", true, p, present, err)) return false;
    static_assert(sizeof(glz_vertex) == 32, "Vertex is 32 bytes (v1.rs:631-667)");
    vertices_.v.resize(p.size() / 32);  // chunks_exact(32)
    if (!vertices_.v.empty()) memcpy(vertices_.v.data(), p.data(), vertices_.v.size() * 32);
    vertices_.done = true;
  }
  out = &vertices_.v;
  return true;
}

bool Parsed::meshes(const std::vector<glz_mesh>*& out, const std::vector<uint32_t>*& indices, Error& err) {
  if (!meshes_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kMesh, "Mesh", true, p, present, err)) return false;
    if (present) {
      std::vector<std::pair<const uint8_t*, size_t>> items;
      if (!split_dynamic(p, "Mesh", items, err)) return false;
      for (auto& it : items) {  // bytes_to_mesh, v1.rs:683-698
        if (it.second < 8) return fail(err, GLZ_E_INVALID_DATA, "Corrupted chunk: Mesh");
        glz_mesh m{};
        m.id = rd16(it.first);
        uint32_t count = rd32(it.first + 2);
        m.material = rd16(it.first + 6);
        if (8 + (uint64_t)count * 4 > it.second) return fail(err, GLZ_E_INVALID_DATA, "Corrupted chunk: Mesh");
        m.index_offset = (uint32_t)indices_.size();
        m.index_count = count;
        size_t base = indices_.size();
        indices_.resize(base + count);
        if (count) memcpy(&indices_[base], it.first + 8, (size_t)count * 4);
        meshes_.v.push_back(m);
      }
    }
    meshes_.done = true;
  }
  out = &meshes_.v;
  indices = &indices_;
  return true;
}

bool Parsed::transforms(const std::vector<glz_transform>*& out, Error& err) {
  if (!transforms_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kTransform, "Transform", true, p, present, err)) return false;
    transforms_.v.resize(p.size() / 64);
    if (!transforms_.v.empty()) memcpy(transforms_.v.data(), p.data(), transforms_.v.size() * 64);
    transforms_.done = true;
  }
  out = &transforms_.v;
  return true;
}

bool Parsed::instances(const std::vector<glz_mesh_instance>*& out, Error& err) {
  if (!instances_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kInstance, "Instance", true, p, present, err)) return false;
    static_assert(sizeof(glz_mesh_instance) == 4, "MeshInstance is 4 bytes");
    instances_.v.resize(p.size() / 4);
    if (!instances_.v.empty()) memcpy(instances_.v.data(), p.data(), instances_.v.size() * 4);
    instances_.done = true;
  }
  out = &instances_.v;
  return true;
}

bool Parsed::cameras(const std::vector<glz_camera>*& out, Error& err) {
  if (!cameras_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kCamera, "Camera", true, p, present, err)) return false;
    for (size_t i = 0; i + 49 <= p.size(); i += 49) {  // bytes_to_camera, v1.rs:748-791
      const uint8_t* b = &p[i];
      glz_camera c{};
      c.type = b[0];
      if (c.type > 1) return fail(err, GLZ_E_INVALID_DATA, "Unexpected cam type");  // the reference panics here
      for (int k = 0; k < 3; ++k) {
        c.position[k] = rdf(b + 1 + 4 * k);
        c.target[k] = rdf(b + 13 + 4 * k);
        c.up[k] = rdf(b + 25 + 4 * k);
      }
      c.fovx_or_scale = rdf(b + 37);
      c.near_plane = rdf(b + 41);
      c.far_plane = rdf(b + 45);
      cameras_.v.push_back(c);
    }
    cameras_.done = true;
  }
  out = &cameras_.v;
  return true;
}

bool Parsed::materials(const std::vector<glz_material>*& out, Error& err) {
  if (!materials_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kMaterial, "Material", true, p, present, err)) return false;
    if (present) {
      std::vector<std::pair<const uint8_t*, size_t>> items;
      if (!split_dynamic(p, "Material", items, err)) return false;
      for (auto& it : items) {  // bytes_to_material, v1.rs:912-958
        if (it.second < 34) return fail(err, GLZ_E_INVALID_DATA, "Corrupted chunk: Material");
        const uint8_t* b = it.first;
        glz_material m{};
        m.mtype = b[0] <= 6 ? b[0] : (uint8_t)GLZ_MAT_LAMBERT;  // MaterialType::from(u8), material.rs:290-298
        m.metal = b[1] <= 28 ? b[1] : 0;                        // Metal::from(u8), metal.rs:418-451
        memcpy(m.diffuse_mul, b + 2, 3);
        memcpy(m.emissive_col, b + 5, 3);
        m.has_emissive = (b[5] | b[6] | b[7]) != 0;
        m.ior = rdf(b + 8);
        m.roughness_mul = rdf(b + 12);
        m.metalness_mul = rdf(b + 16);
        m.anisotropy = rdf(b + 20);
        m.diffuse = rd16(b + 24);
        m.roughness = rd16(b + 26);
        m.metalness = rd16(b + 28);
        m.normal = rd16(b + 30);
        m.opacity = rd16(b + 32);
        copy_name(m.name, b + 34, it.second - 34);
        materials_.v.push_back(m);
      }
    }
    materials_.done = true;
  }
  out = &materials_.v;
  return true;
}

bool Parsed::lights(const std::vector<glz_light>*& out, Error& err) {
  if (!lights_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kLight, "Light", true, p, present, err)) return false;
    if (present) {
      std::vector<std::pair<const uint8_t*, size_t>> items;
      if (!split_dynamic(p, "Light", items, err)) return false;
      for (auto& it : items) {  // bytes_to_light, v1.rs:1009-1047
        if (it.second < 109) return fail(err, GLZ_E_INVALID_DATA, "Corrupted chunk: Light");
        const uint8_t* b = it.first;
        glz_light l{};
        l.ltype = b[0];
        if (l.ltype > 3) return fail(err, GLZ_E_INVALID_DATA, "Invalid enum value for LightType");
        for (int k = 0; k < 3; ++k) {
          l.position[k] = rdf(b + 1 + 4 * k);
          l.direction[k] = rdf(b + 13 + 4 * k);
        }
        l.resource_id = rd32(b + 25);
        l.intensity = rdf(b + 29);
        l.yaw_deg = rdf(b + 33);
        l.pitch_deg = rdf(b + 37);
        l.roll_deg = rdf(b + 41);
        memcpy(l.color, b + 45, 64);
        copy_name(l.name, b + 109, it.second - 109);
        lights_.v.push_back(l);
      }
    }
    lights_.done = true;
  }
  out = &lights_.v;
  return true;
}

bool Parsed::textures(const std::vector<TextureData>*& out, Error& err) {
  if (!textures_.done) {
    std::vector<uint8_t> p;
    bool present;
    if (!chunk_payload(kTexture, "textures", false, p, present, err)) return false;
    if (present) {
      std::vector<std::pair<const uint8_t*, size_t>> items;
      if (!split_dynamic(p, "textures", items, err)) return false;
      textures_.v.resize(items.size());
      std::vector<Error> errs(items.size());
      // bytes_to_texture (v1.rs:820-882); the reference decodes textures in parallel with rayon (v1.rs:598-601)
      auto decode_body = [&](size_t i) {
        const uint8_t* b = items[i].first;
        const size_t n = items[i].second;
        TextureData& t = textures_.v[i];
        if (n < 3) { fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted textures"); return; }
        unsigned fmt = b[0];
        if (fmt < 1 || fmt > 3) { fail(errs[i], GLZ_E_INVALID_INPUT, "Unexpected texture format"); return; }
        size_t sl = b[1], idx = 2;
        if (idx + sl + 1 > n) { fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted textures"); return; }
        copy_name(t.info.name, b + idx, sl);
        idx += sl;
        unsigned mips = b[idx++];
        t.info.format = fmt;
        t.info.mip_levels = mips;
        for (unsigned lvl = 0; lvl < mips; ++lvl) {
          if (idx + 4 > n) { fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted textures"); return; }
          size_t ml = rd32(b + idx);
          idx += 4;
          if (idx + ml > n) { fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted textures"); return; }
          if (lvl == 0) {   // level 0 is what the reference's ray-tracing stages sample
            std::string perr;
            uint32_t w, h;
            std::vector<uint8_t> px;
            if (!png_decode(b + idx, ml, fmt == GLZ_TEX_GRAY ? 1 : 4, w, h, px, perr)) {
              fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted image: " + perr);
              return;
            }
            t.level0 = std::move(px);
            t.info.width = w;
            t.info.height = h;
          } else {          // the other levels feed the opt-in texture LOD: checked now, decoded when the chain is wanted
            std::string perr;
            uint32_t w, h;
            if (!png_check(b + idx, ml, w, h, perr)) {
              fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted image: " + perr);
              return;
            }
            t.more_png.emplace_back(b + idx, b + idx + ml);
            t.more_dims.push_back(w);
            t.more_dims.push_back(h);
          }
          idx += ml;
        }
        if (mips == 0) fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted textures: no mip level");
      };
      // the workers below run outside the C ABI's exception guard: nothing may escape a thread (std::terminate)
      auto decode_one = [&](size_t i) {
        try {
          decode_body(i);
        } catch (const std::bad_alloc&) {
          fail(errs[i], GLZ_E_INVALID_DATA, "Corrupted image: out of memory while decoding");
        } catch (const std::exception& ex) {
          fail(errs[i], GLZ_E_INVALID_DATA, std::string("Corrupted image: ") + ex.what());
        }
      };
      unsigned nthreads = std::min<size_t>(items.size(), std::max(1u, std::thread::hardware_concurrency()));
      if (nthreads <= 1) {
        for (size_t i = 0; i < items.size(); ++i) decode_one(i);
      } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; ++t)
          pool.emplace_back([&, t] { for (size_t i = t; i < items.size(); i += nthreads) decode_one(i); });
        for (auto& th : pool) th.join();
      }
      for (auto& e : errs)
        if (e.code != GLZ_OK) {
          textures_.v.clear();
          err = e;
          return false;
        }
    }
    for (auto& t : textures_.v) t.info.pixels = t.level0.data();
    textures_.done = true;
  }
  out = &textures_.v;
  return true;
}

bool Parsed::meta(glz_meta& out, bool& present, Error& err) {
  std::vector<uint8_t> p;
  if (!chunk_payload(kMeta, "Meta", true, p, present, err)) return false;
  if (!present || p.size() < 20) {
    present = false;
    out = default_meta();
    return true;
  }
  // bytes_to_meta (v1.rs:1064-1080); decode_fixed -> pop() takes the LAST record (v1.rs:355-362)
  const uint8_t* b = &p[(p.size() / 20 - 1) * 20];
  for (int k = 0; k < 3; ++k) out.scene_centre[k] = rdf(b + 4 * k);
  out.scene_radius = rdf(b + 12);
  out.exposure = rdf(b + 16);
  return true;
}

bool Parsed::to_scene_data(SceneData& s, Error& err) {
  const std::vector<glz_vertex>* v;
  const std::vector<glz_mesh>* m;
  const std::vector<uint32_t>* idx;
  const std::vector<glz_transform>* tr;
  const std::vector<glz_mesh_instance>* in;
  const std::vector<glz_camera>* cams;
  const std::vector<glz_material>* mats;
  const std::vector<glz_light>* ls;
  const std::vector<TextureData>* tex;
  // RayTraceScene::new uses unwrap_or_default()/unwrap_or_else() on every getter (scene.rs:1427-1434,
  // :1467-1469): a corrupt chunk silently becomes the default.  We are stricter only in that the
  // error text stays available through glz_last_error(); the resulting scene is the same.
  Error ignored;
  if (vertices(v, ignored)) s.vertices = *v;
  if (meshes(m, idx, ignored)) { s.meshes = *m; s.indices = *idx; }
  if (instances(in, ignored)) s.instances = *in;
  if (transforms(tr, ignored)) s.transforms = *tr; else s.transforms = {identity_transform()};
  if (materials(mats, ignored)) s.materials = *mats; else s.materials = {default_material()};
  if (lights(ls, ignored)) s.lights = *ls;
  if (textures(tex, ignored)) s.textures = *tex; else s.textures = {default_texture()};
  for (auto& t : s.textures) t.info.pixels = t.level0.data();
  if (cameras(cams, ignored) && !cams->empty()) { s.camera = cams->back(); s.has_camera = true; }
  else { s.camera = default_camera(); s.has_camera = false; }
  bool present = false;
  if (!meta(s.meta, present, ignored)) s.meta = default_meta();
  s.has_meta = present;
  (void)err;
  return true;
}

}  // namespace glz
