// What the reference's converter does with an OBJ file, restated without assimp (converter/src/main.rs):
//   * assimp's OBJ importer yields one material "DefaultMaterial" (diffuse 0.6) followed by the MTL materials, one mesh
//     per material in use (Triangulate + OptimizeMeshes), flat normals where the file has none (GenerateNormals) and
//     a node graph of identity transforms;
//   * convert_materials (:410-477): glaze's Material::default() and Texture::default() come first, so assimp material i
//     becomes material i + 1 (:386); Kd -> diffuse_mul, a non-black Ke -> emissive colour + an AREA light (:596-606),
//     map_Kd / norm / map_d -> sRGB / linear / gray textures, de-duplicated per (name, format) (:481-529, :608-617);
//   * convert_meshes (:325-392): vertices are (position, normal, uv) with v flipped, de-duplicated on their 32 bytes in
//     first-use order; meshes without uvs get the corner uvs (0,0) (1,0) (1,1);
//   * one identity transform and one instance per mesh (:217-263), Meta from the world bounds (:186-208), and the default
//     camera when the file has none (:394-408; OBJ never has one).
#include "converter.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <unordered_map>

#include "codec/jpeg.h"
#include "codec/png_dec.h"
#include "glz_tables.h"
#include "serializer.h"

namespace glz {
namespace {

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  uint8_t buf[1 << 16];
  size_t n;
  out.clear();
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
  fclose(f);
  return true;
}
std::string dir_of(const std::string& path) {
  const size_t s = path.find_last_of('/');
  return s == std::string::npos ? std::string(".") : path.substr(0, s);
}
std::vector<std::string> tokens(const std::string& line) {
  std::vector<std::string> t;
  size_t i = 0;
  while (i < line.size()) {
    while (i < line.size() && isspace((unsigned char)line[i])) ++i;
    size_t j = i;
    while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
    if (j > i) t.push_back(line.substr(i, j - i));
    i = j;
  }
  return t;
}
std::string rest_after(const std::string& line, const std::string& keyword) {   // argument with spaces (file names)
  size_t i = line.find(keyword);
  i += keyword.size();
  while (i < line.size() && isspace((unsigned char)line[i])) ++i;
  size_t j = line.size();
  while (j > i && isspace((unsigned char)line[j - 1])) --j;
  return line.substr(i, j - i);
}
std::vector<std::string> lines_of(const std::vector<uint8_t>& bytes) {
  std::vector<std::string> out;
  std::string cur;
  for (uint8_t b : bytes) {
    if (b == '\n') {
      if (!cur.empty() && cur.back() == '\\') { cur.pop_back(); continue; }   // line continuation
      out.push_back(cur);
      cur.clear();
    } else if (b != '\r') {
      cur.push_back((char)b);
    }
  }
  if (!cur.empty()) out.push_back(cur);
  return out;
}

struct MtlMaterial {
  std::string name;
  float kd[3] = {0.6f, 0.6f, 0.6f}, ke[3] = {0.0f, 0.0f, 0.0f};
  std::string map_kd, map_norm, map_d;
};
struct Corner { int v, vt, vn; };
struct Face { Corner c[3]; int material; };

uint8_t fcol(float c) {   // fcol_to_ucol: `(col * 255.0) as u8` saturates
  const float v = c * 255.0f;
  return v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v);
}
void set_name(char* dst, size_t cap, const std::string& s) {
  const size_t n = std::min(cap - 1, s.size());
  memcpy(dst, s.data(), n);
  dst[n] = 0;
}

}  // namespace

bool convert_obj(const std::string& input, const std::string& output, bool gen_mipmaps, ConvertReport* report, Error& err) {
  auto fail = [&](int code, const std::string& m) { err.code = code; err.msg = m; return false; };
  std::vector<uint8_t> bytes;
  if (!read_file(input, bytes)) return fail(GLZ_E_IO, "cannot open " + input);
  const std::string base = dir_of(input);

  // ---- OBJ + MTL -------------------------------------------------------------------------------
  std::vector<float> pos, uvs, nrm;
  std::vector<Face> faces;
  std::vector<MtlMaterial> mtl(1);   // assimp's DefaultMaterial
  mtl[0].name = "DefaultMaterial";
  std::map<std::string, int> mtl_index;
  int cur_material = 0;
  auto load_mtl = [&](const std::string& file) {
    std::vector<uint8_t> mb;
    if (!read_file(base + "/" + file, mb)) return;   // a missing library leaves the default material in use, like assimp
    MtlMaterial* cur = nullptr;
    for (const std::string& line : lines_of(mb)) {
      const auto t = tokens(line);
      if (t.empty() || t[0][0] == '#') continue;
      if (t[0] == "newmtl") {
        mtl.emplace_back();
        cur = &mtl.back();
        cur->name = rest_after(line, "newmtl");
        mtl_index[cur->name] = (int)mtl.size() - 1;
      } else if (cur && (t[0] == "Kd" || t[0] == "Ke") && t.size() >= 4) {
        float* d = t[0] == "Kd" ? cur->kd : cur->ke;
        for (int k = 0; k < 3; ++k) d[k] = strtof(t[1 + k].c_str(), nullptr);
      } else if (cur && (t[0] == "map_Kd" || t[0] == "norm" || t[0] == "map_d") && t.size() >= 2) {
        // the file name is the rest of the line (it may contain spaces) unless map options (-bm 1.0, -o u v w ...) precede it
        const std::string arg = rest_after(line, t[0]);
        const std::string file = arg[0] == '-' ? t.back() : arg;
        (t[0] == "map_Kd" ? cur->map_kd : (t[0] == "norm" ? cur->map_norm : cur->map_d)) = file;
      }
    }
  };
  auto resolve = [](int idx, size_t count) -> int {   // OBJ indices are 1-based, negative = relative to the end
    if (idx > 0) return idx - 1;
    if (idx < 0) return (int)count + idx;
    return -1;
  };
  for (const std::string& line : lines_of(bytes)) {
    const auto t = tokens(line);
    if (t.empty() || t[0][0] == '#') continue;
    if (t[0] == "v" && t.size() >= 4) {
      for (int k = 0; k < 3; ++k) pos.push_back(strtof(t[1 + k].c_str(), nullptr));
    } else if (t[0] == "vt" && t.size() >= 3) {
      for (int k = 0; k < 2; ++k) uvs.push_back(strtof(t[1 + k].c_str(), nullptr));
    } else if (t[0] == "vn" && t.size() >= 4) {
      for (int k = 0; k < 3; ++k) nrm.push_back(strtof(t[1 + k].c_str(), nullptr));
    } else if (t[0] == "mtllib") {
      load_mtl(rest_after(line, "mtllib"));
    } else if (t[0] == "usemtl") {
      const auto it = mtl_index.find(rest_after(line, "usemtl"));
      cur_material = it == mtl_index.end() ? 0 : it->second;
    } else if (t[0] == "f" && t.size() >= 4) {
      std::vector<Corner> poly;
      for (size_t k = 1; k < t.size(); ++k) {
        int idx[3] = {0, 0, 0};
        int field = 0;
        const char* s = t[k].c_str();
        while (*s && field < 3) {
          char* e;
          const long v = strtol(s, &e, 10);
          if (e != s) idx[field] = (int)v;
          s = e;
          if (*s == '/') { ++s; ++field; } else break;
        }
        Corner c{resolve(idx[0], pos.size() / 3), resolve(idx[1], uvs.size() / 2), resolve(idx[2], nrm.size() / 3)};
        if (c.v < 0 || (size_t)c.v >= pos.size() / 3) return fail(GLZ_E_INVALID_DATA, "face refers to a missing vertex");
        if (c.vt >= (int)(uvs.size() / 2) || c.vn >= (int)(nrm.size() / 3)) return fail(GLZ_E_INVALID_DATA, "face refers to a missing uv or normal");
        poly.push_back(c);
      }
      for (size_t k = 1; k + 1 < poly.size(); ++k) faces.push_back(Face{{poly[0], poly[k], poly[k + 1]}, cur_material});   // Triangulate (fan)
    }
  }

  // ---- textures + materials (convert_materials) ------------------------------------------------------
  std::vector<TextureData> textures;
  textures.push_back(default_texture());
  std::vector<glz_material> materials;
  materials.push_back(default_material());
  std::vector<glz_light> lights;
  std::unordered_map<std::string, uint16_t> used;
  auto convert_texture = [&](const std::string& name, uint32_t format, uint16_t& id) -> bool {
    static const char* kSuffix[4] = {"", "(R)", "(sRGBA)", "(lRGBA)"};   // used_name, :608-617
    const std::string key = name + kSuffix[format];
    const auto it = used.find(key);
    if (it != used.end()) { id = it->second; return true; }
    std::string rel = name;
    std::replace(rel.begin(), rel.end(), '\\', '/');
    const std::string path = (!rel.empty() && rel[0] == '/') ? rel : base + "/" + rel;
    std::vector<uint8_t> fb;
    if (!read_file(path, fb)) return fail(GLZ_E_IO, "cannot open texture " + path);
    TextureData t;
    uint32_t w = 0, h = 0;
    std::string derr;
    const int channels = format == GLZ_TEX_GRAY ? 1 : 4;
    bool ok;
    if (fb.size() >= 8 && fb[0] == 0x89 && fb[1] == 'P') ok = png_decode(fb.data(), fb.size(), channels, w, h, t.level0, derr);
    else ok = jpeg_decode(fb.data(), fb.size(), channels, w, h, t.level0, derr);
    if (!ok) return fail(GLZ_E_INVALID_DATA, "Could not read texture format (" + derr + ")");
    if (w > 65535 || h > 65535) return fail(GLZ_E_INVALID_DATA, "texture larger than 65535 pixels");   // `as u16`, :491-492
    t.info.format = format;
    t.info.width = w;
    t.info.height = h;
    t.info.mip_levels = 1;
    set_name(t.info.name, sizeof(t.info.name), name);
    id = (uint16_t)textures.size();
    textures.push_back(std::move(t));
    used[key] = id;
    return true;
  };
  for (size_t i = 0; i < mtl.size(); ++i) {
    const MtlMaterial& src = mtl[i];
    glz_material m = default_material();
    set_name(m.name, sizeof(m.name), src.name);
    for (int k = 0; k < 3; ++k) m.diffuse_mul[k] = fcol(src.kd[k]);
    const uint8_t e[3] = {fcol(src.ke[0]), fcol(src.ke[1]), fcol(src.ke[2])};
    if (e[0] || e[1] || e[2]) {
      m.has_emissive = 1;
      memcpy(m.emissive_col, e, 3);
    }
    if (!src.map_kd.empty() && !convert_texture(src.map_kd, GLZ_TEX_RGBA_SRGB, m.diffuse)) return false;
    if (!src.map_norm.empty() && !convert_texture(src.map_norm, GLZ_TEX_RGBA_NORM, m.normal)) return false;
    if (!src.map_d.empty() && !convert_texture(src.map_d, GLZ_TEX_GRAY, m.opacity)) return false;
    if (m.has_emissive) {   // :596-606
      glz_light l{};
      l.ltype = GLZ_LIGHT_AREA;
      set_name(l.name, sizeof(l.name), src.name);
      l.resource_id = (uint32_t)materials.size();
      l.direction[1] = -1.0f;   // ..Default::default() (geometry/light.rs:176-191): white spectrum, direction -Y, intensity 1
      l.intensity = 1.0f;
      memcpy(l.color, GLZ_HOST_SPECTRUM_WHITE, sizeof(l.color));
      lights.push_back(l);
    }
    materials.push_back(m);
  }

  // ---- meshes + vertices (convert_meshes) ----------------------------------------------------------
  std::vector<int> mesh_of_material(mtl.size(), -1);
  std::vector<int> mesh_material;                    // assimp material index per mesh, first-use order
  for (const Face& f : faces)
    if (mesh_of_material[f.material] < 0) {
      mesh_of_material[f.material] = (int)mesh_material.size();
      mesh_material.push_back(f.material);
    }
  std::vector<bool> mesh_has_uv(mesh_material.size(), true);
  for (const Face& f : faces)
    for (const Corner& c : f.c)
      if (c.vt < 0) mesh_has_uv[mesh_of_material[f.material]] = false;
  std::vector<glz_vertex> vertices;
  std::vector<std::vector<uint32_t>> mesh_indices(mesh_material.size());
  struct Key {
    uint8_t b[32];
    bool operator==(const Key& o) const { return memcmp(b, o.b, 32) == 0; }
  };
  struct KeyHash {
    size_t operator()(const Key& k) const {
      uint64_t h = 1469598103934665603ull;
      for (uint8_t x : k.b) h = (h ^ x) * 1099511628211ull;
      return (size_t)h;
    }
  };
  std::unordered_map<Key, uint32_t, KeyHash> seen;
  static const float kDefaultUv[3][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {1.0f, 1.0f}};
  for (size_t mi = 0; mi < mesh_material.size(); ++mi)
    for (const Face& f : faces) {
      if (mesh_of_material[f.material] != (int)mi) continue;
      float flat[3] = {0.0f, 0.0f, 0.0f};
      if (f.c[0].vn < 0 || f.c[1].vn < 0 || f.c[2].vn < 0) {   // GenerateNormals: the face normal
        const float* a = &pos[3 * f.c[0].v];
        const float* b = &pos[3 * f.c[1].v];
        const float* c = &pos[3 * f.c[2].v];
        const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        flat[0] = e1[1] * e2[2] - e1[2] * e2[1];
        flat[1] = e1[2] * e2[0] - e1[0] * e2[2];
        flat[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const float len = sqrtf(flat[0] * flat[0] + flat[1] * flat[1] + flat[2] * flat[2]);
        if (len > 0.0f) { flat[0] /= len; flat[1] /= len; flat[2] /= len; }
      }
      for (int k = 0; k < 3; ++k) {
        const Corner& c = f.c[k];
        glz_vertex v{};
        memcpy(v.vv, &pos[3 * c.v], 12);
        if (c.vn >= 0) memcpy(v.vn, &nrm[3 * c.vn], 12); else memcpy(v.vn, flat, 12);
        const float* uv = mesh_has_uv[mi] ? &uvs[2 * c.vt] : kDefaultUv[k];
        v.vt[0] = uv[0];
        v.vt[1] = 1.0f - uv[1];   // flip y for vulkan (:367)
        Key key;
        memcpy(key.b, &v, 32);
        auto it = seen.find(key);
        uint32_t index;
        if (it == seen.end()) {
          index = (uint32_t)vertices.size();
          seen.emplace(key, index);
          vertices.push_back(v);
        } else {
          index = it->second;
        }
        mesh_indices[mi].push_back(index);
      }
    }
  if (mesh_material.size() > 65535) return fail(GLZ_E_INVALID_DATA, "more than 65535 meshes");
  std::vector<glz_mesh> meshes;
  std::vector<uint32_t> indices;
  std::vector<glz_mesh_instance> instances;
  for (size_t mi = 0; mi < mesh_material.size(); ++mi) {
    glz_mesh m{};
    m.id = (uint16_t)mi;
    m.material = (uint16_t)(mesh_material[mi] + 1);   // +1 because 0 is the default material (:386)
    m.index_offset = (uint32_t)indices.size();
    m.index_count = (uint32_t)mesh_indices[mi].size();
    indices.insert(indices.end(), mesh_indices[mi].begin(), mesh_indices[mi].end());
    meshes.push_back(m);
    instances.push_back(glz_mesh_instance{(uint16_t)mi, 0});
  }
  const glz_transform identity = identity_transform();

  // ---- meta + camera (calc_scene_centre_radius, convert_cameras) -------------------------------------
  float pmin[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, pmax[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
  for (uint32_t i : indices)
    for (int k = 0; k < 3; ++k) {
      pmin[k] = fminf(pmin[k], vertices[i].vv[k]);
      pmax[k] = fmaxf(pmax[k], vertices[i].vv[k]);
    }
  glz_meta meta{};
  float d2 = 0.0f;
  for (int k = 0; k < 3; ++k) {
    meta.scene_centre[k] = pmin[k] + (pmax[k] - pmin[k]) * 0.5f;
    d2 += (pmax[k] - pmin[k]) * (pmax[k] - pmin[k]);
  }
  meta.scene_radius = sqrtf(d2) / 2.0f;
  meta.exposure = 1.0f;
  glz_camera cam{};
  cam.type = GLZ_CAMERA_PERSPECTIVE;
  cam.target[2] = 100.0f;
  cam.up[1] = 1.0f;
  cam.fovx_or_scale = 90.0f * (float)(M_PI / 180.0);   // f32::to_radians(90.0)
  cam.near_plane = fmaxf(1e-3f, meta.scene_radius * 2.0f * 1e-5f);
  cam.far_plane = fmaxf(100.0f, meta.scene_radius * 2.0f);

  // ---- write_output ----------------------------------------------------------------------------------
  std::vector<glz_texture> tex_view;
  for (TextureData& t : textures) {
    glz_texture g = t.info;
    g.pixels = t.level0.data();
    g.mip_levels = gen_mipmaps ? 32 : 1;   // gen_mipmaps: the full chain (catmull_rom_mips stops at 1x1)
    tex_view.push_back(g);
  }
  SerializeInput in;
  in.vertices = vertices.data(); in.n_vertices = vertices.size();
  in.indices = indices.data(); in.n_indices = indices.size();
  in.meshes = meshes.data(); in.n_meshes = meshes.size();
  in.transforms = &identity; in.n_transforms = 1;
  in.instances = instances.data(); in.n_instances = instances.size();
  in.cameras = &cam; in.n_cameras = 1;
  in.textures = tex_view.data(); in.n_textures = tex_view.size();
  in.materials = materials.data(); in.n_materials = materials.size();
  in.lights = lights.data(); in.n_lights = lights.size();
  in.meta = &meta;
  if (!serialize_scene(output, in, err)) return false;
  if (report) {
    report->vertices = vertices.size();
    report->triangles = indices.size() / 3;
    report->meshes = meshes.size();
    report->materials = materials.size();
    report->textures = textures.size();
    report->lights = lights.size();
    report->scene_radius = meta.scene_radius;
  }
  return true;
}

}  // namespace glz
