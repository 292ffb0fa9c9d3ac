// AddressSanitizer / UBSan run of the host-only parts of libglaze_hip (the GPU parts cannot run under sanitizers on
// this pool): .glaze reader and writer with their own xz / PNG / JPEG codecs, the OBJ converter and the host SAH builder.
// Built and run by tools/sanitize/run.sh; exits non-zero on the first failure (the sanitizers abort on their findings).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include <zlib.h>

#include "codec/jpeg.h"
#include "codec/png_dec.h"
#include "codec/png_enc.h"
#include "converter.h"
#include "kernels.h"
#include "parser.h"
#include "serializer.h"

using namespace glz;

#define REQUIRE(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #x, err.msg.c_str()); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: host_sanitize mattest.glaze cube.obj tmpdir checker.jpg\n"); return 2; }
  const std::string mattest = argv[1], cube_obj = argv[2], tmp = argv[3], checker_jpg = argv[4];
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
  // 5b. malformed JPEG: truncations, bit flips, a DHT whose code lengths overflow the code space (200 codes of length 2: the
  //     9-bit lookup table has 512 entries), fractional sampling ratios, absurd dimensions.  Must fail or decode, never crash.
  {
    FILE* f = fopen(checker_jpg.c_str(), "rb");
    REQUIRE(f);
    std::vector<uint8_t> jpg;
    unsigned char buf[65536]; size_t k;
    while ((k = fread(buf, 1, sizeof(buf), f)) > 0) jpg.insert(jpg.end(), buf, buf + k);
    fclose(f);
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> px;
    std::string jerr;
    REQUIRE(jpeg_decode(jpg.data(), jpg.size(), 4, w, h, px, jerr) && w == 512 && h == 512);
    auto find_marker = [&](uint8_t m) -> size_t {
      for (size_t i = 2; i + 3 < jpg.size();) {
        if (jpg[i] != 0xFF) return 0;
        if (jpg[i + 1] == m) return i;
        i += 2 + (((size_t)jpg[i + 2] << 8) | jpg[i + 3]);
      }
      return 0;
    };
    const size_t dht = find_marker(0xC4), sof = find_marker(0xC0);
    REQUIRE(dht && sof);
    std::mt19937 jr(11);
    for (int trial = 0; trial < 200; ++trial) {
      std::vector<uint8_t> bad = jpg;
      if (trial < 40) bad.resize(bad.size() * (size_t)(trial + 1) / 41);
      else if (trial < 160) for (int j = 0; j < 1 + trial % 7; ++j) bad[jr() % (trial < 100 ? std::min<size_t>(bad.size(), 700) : bad.size())] ^= (uint8_t)(1u << (jr() % 8));
      else if (trial < 170) { for (int l = 1; l <= 16; ++l) bad[dht + 4 + l] = 0; bad[dht + 4 + 1 + (trial - 160) % 9] = (uint8_t)(200 + trial % 50); }
      else if (trial < 180) { bad[sof + 11] = (uint8_t)(((1 + trial % 4) << 4) | (1 + (trial / 4) % 4)); bad[sof + 14] = (uint8_t)(((1 + (trial / 2) % 4) << 4) | 3); }
      else { bad[sof + 5] = (uint8_t)jr(); bad[sof + 6] = (uint8_t)jr(); bad[sof + 7] = (uint8_t)jr(); bad[sof + 8] = (uint8_t)jr(); }
      for (int ch : {1, 4}) { px.clear(); (void)jpeg_decode(bad.data(), bad.size(), ch, w, h, px, jerr); }
    }
  }
  // 5c. malformed PNG: truncations, flips, and an IHDR (CRC fixed up) that claims 65535 x 65535 for a tiny IDAT
  {
    std::vector<uint8_t> img(64 * 48 * 4), png;
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)(i * 7 + i / 64);
    REQUIRE(png_encode(img.data(), 64, 48, 4, png));
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> px;
    std::string perr;
    REQUIRE(png_decode(png.data(), png.size(), 4, w, h, px, perr) && w == 64 && h == 48 && px == img);
    auto fix_ihdr_crc = [](std::vector<uint8_t>& b) {
      const uint32_t c = (uint32_t)::crc32(::crc32(0, Z_NULL, 0), b.data() + 12, 17);
      b[29] = (uint8_t)(c >> 24); b[30] = (uint8_t)(c >> 16); b[31] = (uint8_t)(c >> 8); b[32] = (uint8_t)c;
    };
    std::mt19937 pr(13);
    for (int trial = 0; trial < 120; ++trial) {
      std::vector<uint8_t> bad = png;
      if (trial < 30) bad.resize(bad.size() * (size_t)(trial + 1) / 31);
      else if (trial < 90) bad[pr() % bad.size()] ^= (uint8_t)(1u << (pr() % 8));
      else {
        if (trial % 3 == 0) { bad[16] = 0; bad[17] = 0; bad[18] = 0xFF; bad[19] = 0xFF; bad[20] = 0; bad[21] = 0; bad[22] = 0xFF; bad[23] = 0xFF; }
        else for (int j = 16; j < 26; ++j) if (pr() % 3 == 0) bad[j] = (uint8_t)pr();
        fix_ihdr_crc(bad);
      }
      for (int ch : {1, 4}) { px.clear(); (void)png_decode(bad.data(), bad.size(), ch, w, h, px, perr); }
    }
  }
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
