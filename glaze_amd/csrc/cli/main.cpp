// glaze-cli for the HIP render path: same command line as the reference's `glaze-cli`
// (cli/src/main.rs:24-39: `input output -r/--res WxH -s/--spp N -i/--integrator {direct,pt}`),
// same messages and exit codes (cli/src/main.rs:41-135), driving the C ABI of libglaze_hip.so.
// Build-defined extras: --seed, --depth, --device, --devices a,b,c (several GPUs of this process: tiles sharded, RCCL reduce),
// --hdr-out file.pfm, --report (JSON on stdout).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "glaze_abi.h"

namespace {

bool ends_with(const std::string& s, const char* suffix) {
  const size_t n = strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// little-endian PFM (RGB float), bottom-up rows, of the cumulative image normalised by the launch count
bool write_pfm(const std::string& path, const float* rgba, uint32_t w, uint32_t h) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
  std::vector<float> row((size_t)w * 3);
  for (uint32_t y = 0; y < h; ++y) {
    const float* src = rgba + (size_t)(h - 1 - y) * w * 4;
    for (uint32_t x = 0; x < w; ++x) {
      const float n = src[4 * x + 3] > 0 ? 1.0f / src[4 * x + 3] : 0.0f;
      row[3 * x] = src[4 * x] * n; row[3 * x + 1] = src[4 * x + 1] * n; row[3 * x + 2] = src[4 * x + 2] * n;
    }
    fwrite(row.data(), sizeof(float), row.size(), f);
  }
  fclose(f);
  return true;
}

struct Progress {
  size_t done = 0, total = 0;
  std::string msg;
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
};
void on_sample(void* user) {   // pb.inc(1), cli/src/main.rs:111-113
  Progress* p = static_cast<Progress*>(user);
  ++p->done;
  auto now = std::chrono::steady_clock::now();
  if (p->done == p->total || std::chrono::duration<double>(now - p->last).count() > 0.25) {
    p->last = now;
    const int width = 40, fill = (int)(p->total ? (width * p->done) / p->total : width);
    fprintf(stderr, "\r%s %7zu/%-7zu %.*s%.*s", p->msg.c_str(), p->done, p->total, fill, "########################################",
            width - fill, "----------------------------------------");
    fflush(stderr);
  }
}

void usage(const char* argv0) {
  fprintf(stderr,
          "Usage: %s [OPTIONS] <INPUT> <OUTPUT>\n\n"
          "Arguments:\n  <INPUT>   Input scene to be rendered\n  <OUTPUT>  Output image (.jpg or .png only)\n\n"
          "Options:\n  -r, --res <RESOLUTION>        Rendering resolution in form \"WxH\" [default: 1920x1080]\n"
          "  -s, --spp <SPP>               Samples per pixel [default: 256]\n"
          "  -i, --integrator <INTEGRATOR> [default: pt] [possible values: direct, pt]\n"
          "      --seed <N>                seed of the per-launch seed stream [default: 0]\n"
          "      --depth <N>               path depth (the reference's PT_STEPS) [default: 6]\n"
          "      --device <N>              HIP device ordinal [default: first gfx950]\n"
          "      --devices <A,B,..>        render on several GPUs (64x64 tiles sharded over them, RCCL reduce onto the first)\n"
          "      --texture-lod <MODE>      [default: off] [possible values: off, cones, aniso] (off = level 0, what the reference's stages sample)\n"
          "      --hdr-out <FILE.pfm>      also write the float radiance image\n"
          "      --report                  print a JSON timing report on stdout\n",
          argv0);
}

}  // namespace

int main(int argc, char** argv) {
  std::string input, output, resolution = "1920x1080", integrator = "pt", hdr_out, texture_lod = "off";
  size_t spp = 256;
  uint64_t seed = 0;
  uint32_t depth = 6;
  int device = -1;
  std::vector<int> devices;
  bool report = false;
  std::vector<std::string> positional;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto value = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        fprintf(stderr, "error: a value is required for '%s'\n", name);
        exit(2);
      }
      return argv[++i];
    };
    if (a == "-r" || a == "--res") resolution = value("--res");
    else if (a == "-s" || a == "--spp") spp = strtoull(value("--spp"), nullptr, 10);
    else if (a == "-i" || a == "--integrator") integrator = value("--integrator");
    else if (a == "--seed") seed = strtoull(value("--seed"), nullptr, 10);
    else if (a == "--depth") depth = (uint32_t)strtoul(value("--depth"), nullptr, 10);
    else if (a == "--device") device = atoi(value("--device"));
    else if (a == "--devices") {
      const std::string list = value("--devices");
      size_t pos = 0;
      while (pos <= list.size()) {
        const size_t comma = std::min(list.find(',', pos), list.size());
        char* e = nullptr;
        const std::string item = list.substr(pos, comma - pos);
        const long d = strtol(item.c_str(), &e, 10);
        if (item.empty() || *e || d < 0 || d > 1023) { fprintf(stderr, "error: invalid value '%s' for '--devices' (comma-separated HIP device ordinals)\n", list.c_str()); return 2; }
        devices.push_back((int)d);
        pos = comma + 1;
      }
    }
    else if (a == "--texture-lod") texture_lod = value("--texture-lod");
    else if (a == "--hdr-out") hdr_out = value("--hdr-out");
    else if (a == "--report") report = true;
    else if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
    else if (!a.empty() && a[0] == '-') { fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); usage(argv[0]); return 2; }
    else positional.push_back(a);
  }
  if (positional.size() != 2) { usage(argv[0]); return 2; }
  input = positional[0];
  output = positional[1];
  if (integrator != "direct" && integrator != "pt") {
    fprintf(stderr, "error: invalid value '%s' for '--integrator' [possible values: direct, pt]\n", integrator.c_str());
    return 2;
  }
  if (texture_lod != "off" && texture_lod != "cones" && texture_lod != "aniso") {
    fprintf(stderr, "error: invalid value '%s' for '--texture-lod' [possible values: off, cones, aniso]\n", texture_lod.c_str());
    return 2;
  }
  // check output (cli/src/main.rs:46-55)
  if (!(ends_with(output, "jpg") || ends_with(output, "png"))) {
    fprintf(stderr, "[ERROR] The output image must end with .jpg or .png\n");
    return 1;
  }
  {
    FILE* f = fopen(output.c_str(), "wb");
    if (!f) { fprintf(stderr, "[ERROR] The output file can not be written\n"); return 1; }
    fclose(f);
  }
  // check resolution (:56-75)
  const size_t xpos = resolution.find('x');
  if (xpos == std::string::npos) {
    fprintf(stderr, "[ERROR] The resolution must be specified in form WIDTHxHEIGHT, for example 1920x1080\n");
    return 1;
  }
  char* endp = nullptr;
  const unsigned long width = strtoul(resolution.substr(0, xpos).c_str(), &endp, 10);
  if (*endp || width == 0 || width > 65535) { fprintf(stderr, "[ERROR] Failed to parse the requested width\n"); return 1; }
  const unsigned long height = strtoul(resolution.substr(xpos + 1).c_str(), &endp, 10);
  if (*endp || height == 0 || height > 65535) { fprintf(stderr, "[ERROR] Failed to parse the requested height\n"); return 1; }

  if (!devices.empty()) {
    if (device >= 0 && device != devices[0]) { fprintf(stderr, "error: '--device' and the first entry of '--devices' disagree\n"); return 2; }
    device = devices[0];
  }
  glz_instance* instance = glz_instance_create(device);
  if (!instance) {
    fprintf(stderr, "[ERROR] Cannot create the HIP instance. Is there an MI355X (gfx950) in this machine? (%s)\n", glz_last_error());
    return 0;   // the reference logs and falls off the end of main (cli/src/main.rs:132-134)
  }
  glz_parsed* parsed = glz_parse(input.c_str());
  if (!parsed) {
    fprintf(stderr, "[ERROR] %s\n", glz_last_error());
    return 1;
  }
  fprintf(stderr, "Parsing and setting up scene... ");
  const auto t0 = std::chrono::steady_clock::now();
  glz_scene* scene = glz_scene_create(instance, parsed);
  if (!scene) { fprintf(stderr, "\n[ERROR] %s\n", glz_last_error()); return 1; }
  glz_scene_info info{};
  glz_scene_get_info(scene, &info);
  glz_renderer* renderer = glz_renderer_create(instance, scene, (uint32_t)width, (uint32_t)height);
  if (!renderer) { fprintf(stderr, "\n[ERROR] %s\n", glz_last_error()); return 1; }
  glz_renderer_set_integrator(renderer, integrator == "direct" ? GLZ_DIRECT : GLZ_PATH_TRACE);
  if (glz_renderer_set_depth(renderer, depth) != GLZ_OK) { fprintf(stderr, "\n[ERROR] %s\n", glz_last_error()); return 1; }
  if (texture_lod != "off" && glz_renderer_set_texture_lod(renderer, texture_lod == "aniso" ? GLZ_LOD_RAY_CONES_ANISO : GLZ_LOD_RAY_CONES) != GLZ_OK) {
    fprintf(stderr, "\n[ERROR] %s\n", glz_last_error());
    return 1;
  }
  glz_renderer_set_seed(renderer, seed);
  if (devices.size() > 1 && glz_renderer_set_devices(renderer, devices.data(), (int)devices.size()) != GLZ_OK) {
    fprintf(stderr, "\n[ERROR] %s\n", glz_last_error());
    return 1;
  }
  const auto t1 = std::chrono::steady_clock::now();
  const double setup_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  fprintf(stderr, "Done (%.0f ms)\n", setup_ms);

  Progress pb;
  pb.total = spp;
  pb.msg = "Rendering @ " + std::to_string(width) + "x" + std::to_string(height) + "... ";
  std::vector<uint8_t> image((size_t)width * height * 4);
  const auto r0 = std::chrono::steady_clock::now();
  if (glz_renderer_draw(renderer, spp, on_sample, &pb, image.data()) != GLZ_OK) {
    fprintf(stderr, "\n[ERROR] %s\n", glz_last_error());
    return 1;
  }
  const auto r1 = std::chrono::steady_clock::now();
  const double render_ms = std::chrono::duration<double, std::milli>(r1 - r0).count();
  fprintf(stderr, "\r%sDone (%.0f ms)%60s\n", pb.msg.c_str(), render_ms, "");
  int rc = 0;
  if (glz_save_image(output.c_str(), image.data(), (uint32_t)width, (uint32_t)height) != GLZ_OK) {
    fprintf(stderr, "[ERROR] Failed to save image: %s\n", output.c_str());
    rc = 1;
  } else {
    fprintf(stderr, "All done :)\n");
  }
  if (!hdr_out.empty()) {
    std::vector<float> hdr((size_t)width * height * 4);
    if (glz_renderer_read_hdr(renderer, hdr.data()) != GLZ_OK || !write_pfm(hdr_out, hdr.data(), (uint32_t)width, (uint32_t)height)) {
      fprintf(stderr, "[ERROR] Failed to save the HDR image: %s\n", hdr_out.c_str());
      rc = 1;
    }
  }
  if (report) {
    glz_render_stats st{};
    glz_renderer_get_stats(renderer, &st);
    const uint32_t steps = glz_renderer_steps_per_sample(renderer);
    printf("{\"input\": \"%s\", \"width\": %lu, \"height\": %lu, \"spp\": %zu, \"steps_per_sample\": %u, \"launches\": %llu, "
           "\"triangles\": %llu, \"bvh_nodes\": %u, \"bvh_depth\": %u, \"bvh_build_ms\": %.3f, \"setup_ms\": %.1f, \"render_ms\": %.1f, "
           "\"kernel_ms\": {\"trace\": %.2f, \"shade\": %.2f, \"shadow_flush\": %.2f}, \"msamples_per_s\": %.2f, \"devices\": %zu}\n",
           input.c_str(), width, height, spp, steps, (unsigned long long)st.launches, (unsigned long long)info.n_world_triangles, info.bvh_nodes,
           info.bvh_depth, info.build_ms, setup_ms, render_ms, st.trace_closest_ms, st.shade_ms, st.trace_shadow_ms,
           render_ms > 0 ? (double)st.samples / render_ms / 1e3 : 0.0, devices.empty() ? (size_t)1 : devices.size());
  }
  glz_renderer_destroy(renderer);
  glz_scene_destroy(scene);
  glz_instance_destroy(instance);
  return rc;
}
