// .glaze V1 reader -- host-side mirror of `glaze::parse` + `ContentV1`'s read side
// (lib/src/parser/mod.rs:93-116, lib/src/parser/v1.rs:135-175, :298-362, :476-609, :631-1080).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "glaze_abi.h"

namespace glz {

struct Error {
  int code = GLZ_OK;
  std::string msg;
};

struct TextureData {
  glz_texture info{};           // info.pixels is patched to level0.data() by the getters
  std::vector<uint8_t> level0;
  // levels 1.. as stored in the file (`glaze-converter --gen-mipmaps`); used when the chain is complete (materials/texture.rs:196-221),
  // otherwise the chain is generated at upload (mipchain.h)
  std::vector<std::vector<uint8_t>> more_levels;
  std::vector<uint32_t> more_dims;   // width, height per stored level
  // The parser keeps the stored levels 1.. as the PNG files they are (container and CRCs checked at parse, so a damaged file still
  // fails there) and decodes them when the mip chain is first wanted (Scene::ensure_mips -> decode_more_levels): the reference's
  // ray-tracing stages only sample level 0, and a file written with --gen-mipmaps would otherwise cost a third more decode time and
  // host memory for nothing.
  std::vector<std::vector<uint8_t>> more_png;
  bool decode_more_levels(std::string& err);   // more_png -> more_levels (idempotent)
};

// Owned, fully decoded scene ("what Box<dyn ParsedScene> yields when every getter is called").
struct SceneData {
  std::vector<glz_vertex> vertices;
  std::vector<uint32_t> indices;
  std::vector<glz_mesh> meshes;
  std::vector<glz_transform> transforms;
  std::vector<glz_mesh_instance> instances;
  std::vector<glz_material> materials;
  std::vector<glz_light> lights;
  std::vector<TextureData> textures;
  bool has_camera = false;
  glz_camera camera{};
  bool has_meta = false;
  glz_meta meta{};
};

glz_camera default_camera();     // PerspectiveCam::default(), geometry/camera.rs:31-42
glz_meta default_meta();         // Meta::default(), parser/mod.rs:280-288
glz_material default_material(); // Material::default(), materials/material.rs:327-345
glz_transform identity_transform();
TextureData default_texture();   // Texture::default(), materials/texture.rs:236-253 (1x1 white sRGB)

class Parsed {
 public:
  // Opens the file, checks header + offset table (parse() / OffsetsTable::seek_and_parse).
  static std::unique_ptr<Parsed> open(const std::string& path, Error& err);

  // Lazy getters; each verifies the chunk hash and decodes on first use (ParsedScene getters).
  bool vertices(const std::vector<glz_vertex>*& out, Error& err);
  bool meshes(const std::vector<glz_mesh>*& out, const std::vector<uint32_t>*& indices, Error& err);
  bool transforms(const std::vector<glz_transform>*& out, Error& err);
  bool instances(const std::vector<glz_mesh_instance>*& out, Error& err);
  bool cameras(const std::vector<glz_camera>*& out, Error& err);
  bool materials(const std::vector<glz_material>*& out, Error& err);
  bool lights(const std::vector<glz_light>*& out, Error& err);
  bool textures(const std::vector<TextureData>*& out, Error& err);
  bool meta(glz_meta& out, bool& present, Error& err);

  // Everything at once, applying RayTraceScene::new's defaults for missing chunks
  // (vulkan/scene.rs:1427-1434, :1467-1469, :1487-1491, :1532).
  bool to_scene_data(SceneData& out, Error& err);

  // ParsedScene::update (v1.rs:364-422): rewrites the file with the given chunks replaced (a null pointer keeps the chunk
  // as stored, byte for byte; geometry chunks are always kept), then re-reads it.  Unknown chunks are dropped, like the
  // reference does.
  struct Update {
    const glz_camera* cameras = nullptr;      int64_t n_cameras = -1;     // -1 = None
    const glz_material* materials = nullptr;  int64_t n_materials = -1;
    const glz_light* lights = nullptr;        int64_t n_lights = -1;
    const glz_texture* textures = nullptr;    int64_t n_textures = -1;
    const glz_meta* meta = nullptr;
  };
  bool update(const Update& u, Error& err);

 private:
  struct Slot { uint64_t off = 0, len = 0; bool present = false; };
  bool chunk_payload(int id, const char* what, bool xz, std::vector<uint8_t>& payload, bool& present, Error& err);
  bool load(const std::string& path, Error& err);
  std::string path_;
  std::vector<uint8_t> file_;
  Slot slots_[256];
  template <class T> struct Cache { bool done = false; std::vector<T> v; };
  Cache<glz_vertex> vertices_;
  Cache<glz_mesh> meshes_;
  std::vector<uint32_t> indices_;
  Cache<glz_transform> transforms_;
  Cache<glz_mesh_instance> instances_;
  Cache<glz_camera> cameras_;
  Cache<glz_material> materials_;
  Cache<glz_light> lights_;
  Cache<TextureData> textures_;
};

}  // namespace glz
