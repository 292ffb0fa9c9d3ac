// glaze-converter for the HIP build: the reference's command line (converter/src/main.rs:45-59:
// `glaze-converter <INPUT> [OUTPUT] [-b|--benchmark] [--gen-mipmaps]`) for Wavefront OBJ input, driving the C ABI.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "glaze_abi.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  std::vector<std::string> positional;
  bool benchmark = false, gen_mipmaps = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-b" || a == "--benchmark") benchmark = true;
    else if (a == "--gen-mipmaps") gen_mipmaps = true;
    else if (a == "-h" || a == "--help") {
      printf("Usage: %s [OPTIONS] <INPUT> [OUTPUT]\n\nArguments:\n  <INPUT>   Input scene (Wavefront .obj, or .glaze with --benchmark)\n"
             "  [OUTPUT]  Converted scene\n\nOptions:\n  -b, --benchmark    Perform a reading benchmark on the input scene\n"
             "      --gen-mipmaps  Calculate and store the mip-maps inside the scene file\n", argv[0]);
      return 0;
    } else if (!a.empty() && a[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
    else positional.push_back(a);
  }
  if (positional.empty() || (!benchmark && positional.size() != 2)) {
    fprintf(stderr, "error: the following required arguments were not provided:\n  %s\n", positional.empty() ? "<INPUT>" : "<OUTPUT>");
    return 2;
  }
  if (!benchmark) {
    printf("[1/3] Preprocessing input...\n[2/3] Converting scene...\n[3/3] Compressing file...\n");
    uint64_t counts[6];
    if (glz_convert_obj(positional[0].c_str(), positional[1].c_str(), gen_mipmaps, counts) != GLZ_OK) {
      fprintf(stderr, "Error: Failed to convert scene. %s\n", glz_last_error());
      return 1;
    }
    printf("Done!\n");
    return 0;
  }
  // benchmark (converter/src/main.rs:640-705): convert if needed, then time the chunk getters
  std::string file = positional[0];
  if (!glz_converted_file(file.c_str())) {
    file = "/tmp/glaze_converter_benchmark.bin";
    const double t0 = now_s();
    if (glz_convert_obj(positional[0].c_str(), file.c_str(), gen_mipmaps, nullptr) != GLZ_OK) {
      fprintf(stderr, "Error: Failed to benchmark scene. %s\n", glz_last_error());
      return 1;
    }
    printf("--- Writing ---\nConversion + Compressing + Writing: %fs\n", now_s() - t0);
  }
  glz_parsed* p = glz_parse(file.c_str());
  if (!p) { fprintf(stderr, "Error: Failed to benchmark scene. %s\n", glz_last_error()); return 1; }
  const double t0 = now_s();
  const long long nv = glz_parsed_vertices(p, nullptr, 0);
  const double t1 = now_s();
  const long long nm = glz_parsed_meshes(p, nullptr, 0);
  const double t2 = now_s();
  const long long nt = glz_parsed_textures(p, nullptr, 0);
  const double t3 = now_s();
  const long long nmat = glz_parsed_materials(p, nullptr, 0);
  const double t4 = now_s();
  printf("Reading and writing results for %s\nTotal vertices: %lld\nTotal meshes: %lld\nTotal textures: %lld\nTotal materials: %lld\n",
         positional[0].c_str(), nv, nm, nt, nmat);
  printf("--- Reading ---\nVertices: %fs\nMeshes: %fs\nTextures: %fs\nMaterials: %fs\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
  glz_parsed_free(p);
  return 0;
}
