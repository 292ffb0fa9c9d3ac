// .glaze V1 write side -- host-side mirror of `glaze::Serializer` (lib/src/parser/mod.rs:130-233),
// `ContentV1::serialize` / `write_chunks` (lib/src/parser/v1.rs:230-295), `ParsedScene::update` (:364-422) and the
// record encoders (`*_to_bytes`, :613-1061).
#pragma once
#include <string>
#include <vector>

#include "parser.h"

namespace glz {

// What Serializer::with_*() collects.  Empty arrays write no chunk (OffsetsTable::set_offset skips len 0, v1.rs:188-194).
struct SerializeInput {
  const glz_vertex* vertices = nullptr;          uint64_t n_vertices = 0;
  const uint32_t* indices = nullptr;             uint64_t n_indices = 0;     // all meshes' indices, addressed by glz_mesh
  const glz_mesh* meshes = nullptr;              uint64_t n_meshes = 0;
  const glz_transform* transforms = nullptr;     uint64_t n_transforms = 0;
  const glz_mesh_instance* instances = nullptr;  uint64_t n_instances = 0;
  const glz_camera* cameras = nullptr;           uint64_t n_cameras = 0;
  const glz_texture* textures = nullptr;         uint64_t n_textures = 0;
  const glz_material* materials = nullptr;       uint64_t n_materials = 0;
  const glz_light* lights = nullptr;             uint64_t n_lights = 0;
  const glz_meta* meta = nullptr;                                            // Serializer::with_metadata
};

// One stored chunk: 8-byte XXH64 of the body followed by the body (xz stream, or the raw texture list).
typedef std::vector<uint8_t> ChunkBytes;
ChunkBytes encode_vertices(const glz_vertex* v, uint64_t n);
ChunkBytes encode_meshes(const glz_mesh* m, uint64_t n, const uint32_t* indices, uint64_t n_indices, Error& err);
ChunkBytes encode_transforms(const glz_transform* t, uint64_t n);
ChunkBytes encode_instances(const glz_mesh_instance* i, uint64_t n);
ChunkBytes encode_cameras(const glz_camera* c, uint64_t n);
ChunkBytes encode_textures(const glz_texture* t, uint64_t n, Error& err);
ChunkBytes encode_materials(const glz_material* m, uint64_t n);
ChunkBytes encode_lights(const glz_light* l, uint64_t n);
ChunkBytes encode_meta(const glz_meta& m);

// header + offsets table + chunks, in the order given; empty chunks are skipped (write_chunks, v1.rs:279-295)
bool write_glaze_file(const std::string& path, const std::vector<std::pair<int, ChunkBytes>>& chunks, Error& err);

// Serializer::serialize()
bool serialize_scene(const std::string& path, const SerializeInput& in, Error& err);

// Mip chain of an 8-bit image the way Texture::gen_mipmaps builds it (texture.rs:256-277): every level is the previous one
// resized to half the size with the Catmull-Rom filter of image::imageops::resize.  Level 0 excluded; stops at 1x1 or after
// `levels - 1` reductions.  (The ray-tracing stages only sample level 0.)
std::vector<std::vector<uint8_t>> catmull_rom_mips(const uint8_t* level0, uint32_t w, uint32_t h, int channels, unsigned levels);

}  // namespace glz
