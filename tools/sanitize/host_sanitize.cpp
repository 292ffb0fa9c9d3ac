// AddressSanitizer / UBSan run of the host-only parts of libglaze_hip (the GPU parts cannot run under sanitizers on
// this pool): .glaze reader and writer with their own xz / PNG / JPEG codecs, the OBJ converter and the host SAH builder.
// Built and run by tools/sanitize/run.sh; exits non-zero on the first failure (the sanitizers abort on their findings).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "converter.h"
#include "kernels.h"
#include "parser.h"
#include "serializer.h"

using namespace glz;

#define REQUIRE(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #x, err.msg.c_str()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: host_sanitize mattest.glaze cube.obj tmpdir\n"); return 2; }
  const std::string mattest = argv[1], cube_obj = argv[2], tmp = argv[3];
  Error err;
  // 1. read every chunk of the reference's fixture
  auto p = Parsed::open(mattest, err);
  REQUIRE(p);
  SceneData sd;
  REQUIRE(p->to_scene_data(sd, err));
  REQUIRE(sd.vertices.size() == 70995 || !sd.vertices.empty());
  // 2. write it back (xz + PNG encoders) and read the copy
  std::vector<glz_texture> tex;
  for (auto& t : sd.textures) { glz_texture g = t.info; g.pixels = t.level0.data(); tex.push_back(g); }
  SerializeInput in;
  in.vertices = sd.vertices.data(); in.n_vertices = sd.vertices.size();
  in.indices = sd.indices.data(); in.n_indices = sd.indices.size();
  in.meshes = sd.meshes.data(); in.n_meshes = sd.meshes.size();
  in.transforms = sd.transforms.data(); in.n_transforms = sd.transforms.size();
  in.instances = sd.instances.data(); in.n_instances = sd.instances.size();
  in.cameras = &sd.camera; in.n_cameras = 1;
  in.textures = tex.data(); in.n_textures = tex.size();
  in.materials = sd.materials.data(); in.n_materials = sd.materials.size();
  in.lights = sd.lights.data(); in.n_lights = sd.lights.size();
  in.meta = &sd.meta;
  const std::string copy = tmp + "/copy.glaze";
  REQUIRE(serialize_scene(copy, in, err));
  auto q = Parsed::open(copy, err);
  REQUIRE(q);
  SceneData sd2;
  REQUIRE(q->to_scene_data(sd2, err));
  REQUIRE(sd2.vertices.size() == sd.vertices.size() && sd2.indices == sd.indices && sd2.textures.size() == sd.textures.size());
  for (size_t i = 0; i < sd.textures.size(); ++i) REQUIRE(sd2.textures[i].level0 == sd.textures[i].level0);
  // 3. update in place (materials replaced, everything else kept byte for byte)
  Parsed::Update u;
  u.materials = sd.materials.data(); u.n_materials = (int64_t)sd.materials.size();
  REQUIRE(q->update(u, err));
  // 4. a truncated and a bit-flipped copy must fail cleanly
  {
    FILE* f = fopen(copy.c_str(), "rb");
    REQUIRE(f);
    std::vector<unsigned char> bytes;
    unsigned char buf[65536]; size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) bytes.insert(bytes.end(), buf, buf + k);
    fclose(f);
    std::mt19937 rng(5);
    for (int trial = 0; trial < 24; ++trial) {
      std::vector<unsigned char> bad = bytes;
      if (trial % 3 == 0) bad.resize(bad.size() * (size_t)(trial + 1) / 40);
      else for (int j = 0; j < 1 + trial; ++j) bad[rng() % bad.size()] ^= (unsigned char)(1u << (rng() % 8));
      const std::string badpath = tmp + "/bad.glaze";
      FILE* g = fopen(badpath.c_str(), "wb");
      REQUIRE(g);
      fwrite(bad.data(), 1, bad.size(), g);
      fclose(g);
      Error e2;
      auto b = Parsed::open(badpath, e2);
      if (b) { SceneData s3; (void)b->to_scene_data(s3, e2); }   // may succeed (a flip in padding) or fail: it must not crash
    }
  }
  // 5. OBJ -> .glaze (JPEG decoder, mip generation)
  ConvertReport rep;
  REQUIRE(convert_obj(cube_obj, tmp + "/cube.glaze", true, &rep, err));
  REQUIRE(rep.vertices == 24 && rep.materials == 3 && rep.textures == 2);
  // 6. host SAH builder: random, clustered, identical and NaN boxes
  std::mt19937 rng(9);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  for (uint32_t n : {2u, 3u, 17u, 4097u, 70001u}) {
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<float4> lo(n), hi(n);
      for (uint32_t i = 0; i < n; ++i) {
        float c[3] = {U(rng) * 10, U(rng) * 10, U(rng) * 10};
        if (mode == 1) c[0] = c[1] = c[2] = 1.0f;
        if (mode == 2 && i % 5 == 0) c[1] = NAN;
        lo[i] = make_float4(c[0], c[1], c[2], 0); hi[i] = make_float4(c[0] + 0.1f, c[1] + 0.1f, c[2] + 0.1f, 0);
      }
      std::vector<int2> children(n);
      std::vector<int> parent(2 * (size_t)n);
      build_sah_host(n, lo.data(), hi.data(), children.data(), parent.data());
      std::vector<int> seen(2 * (size_t)n - 1, 0);
      for (uint32_t i = 0; i + 1 < n; ++i)
        for (int link : {children[i].x, children[i].y}) seen[link >= 0 ? link : (int)(n - 1) + ~link]++;
      for (size_t s = 1; s < seen.size(); ++s) REQUIRE(seen[s] == 1);
    }
  }
  printf("host sanitize: ok\n");
  return 0;
}
