// OBJ/MTL -> .glaze converter: the subset of `glaze-converter` (converter/src/main.rs:116-637) that does not need assimp.
#pragma once
#include <string>

#include "parser.h"

namespace glz {
struct ConvertReport {
  uint64_t vertices = 0, triangles = 0, meshes = 0, materials = 0, textures = 0, lights = 0;
  float scene_radius = 0.0f;
};
// Reads a Wavefront OBJ (+ its mtllib files and the PNG / baseline-JPEG textures they name) and writes a .glaze V1 file
// laid out the way the reference's converter lays it out (see converter.cpp).
bool convert_obj(const std::string& input, const std::string& output, bool gen_mipmaps, ConvertReport* report, Error& err);
}  // namespace glz
