// RayTraceInstance / RayTraceScene equivalents: device selection, scene upload, LBVH build.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "device/types.h"
#include "glaze_abi.h"
#include "parser.h"

namespace glz {

// RayTraceInstance (lib/src/vulkan/instance.rs:376-427): one HIP device + the stream all work runs on.
struct Instance {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string arch;
  // two triangles share a leaf when area(joint box) <= ratio * (area(a) + area(b)); tuning switch GLAZE_BVH_PAIRS=<ratio>, 0 = never
  float bvh_pair_area_ratio = getenv("GLAZE_BVH_PAIRS") ? (float)atof(getenv("GLAZE_BVH_PAIRS")) : kPairAreaRatio;
  int bvh_builder = 3;   // kBvhBuilder* for scenes created afterwards (glz_instance_set_bvh_builder); 3 = kBvhBuilderAuto
  ~Instance();
  static Instance* create(int hip_device, Error& err);
};

template <class T>
struct DeviceBuffer {
  T* ptr = nullptr;
  size_t count = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  hipError_t alloc(size_t n) {
    release();
    count = n;
    return hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(T) * (n ? n : 1));
  }
  hipError_t upload(const T* host, size_t n, hipStream_t st) {
    hipError_t e = alloc(n);
    if (e != hipSuccess || n == 0) return e;
    return hipMemcpyAsync(ptr, host, sizeof(T) * n, hipMemcpyHostToDevice, st);
  }
};

// RayTraceScene (lib/src/vulkan/scene.rs:1352-1556)
class Scene {
 public:
  static Scene* create(Instance* inst, SceneData&& data, Error& err);
  // update_materials_and_lights (scene.rs:1587-1716): rebuilds RTMaterial / RTLight / sky tables.
  // The BVH is rebuilt only if a material's opacity flag changed (acceleration.rs:136-141).
  // `textures` (may be null = keep) replaces the texture array: the reference passes the raw textures for the sky
  // distributions (scene.rs:1598-1615) and re-binds the shared GPU textures separately (refresh_descriptors).
  bool update_materials_and_lights(const glz_material* mats, uint32_t n_mats, const glz_light* lights, uint32_t n_lights,
                                   const glz_texture* textures, uint32_t n_textures, Error& err);
  bool refresh_textures(const glz_texture* textures, uint32_t n, Error& err);   // refresh_binded_textures, raytracer.rs:328-356

  Instance* instance = nullptr;
  SceneData data;           // host copy (materials/lights are kept for updates)
  glz_scene_info info{};
  DeviceScene dev{};        // what the kernels see
  uint32_t lights_no = 0;   // lights.len() after reorder_lights (scene.rs:1549)
  uint32_t stack_overflow_depth = 1;

  // host mirrors of the uploaded RT arrays (debug read-back / parity with the oracle)
  std::vector<RTInstance> h_instances;
  std::vector<RTMaterial> h_materials;
  std::vector<RTLight> h_lights;
  std::vector<float> h_sky_marginal;
  SkyHeader h_sky_header{};
  RTSky h_sky{};

 private:
  bool upload_geometry(Error& err);
  bool upload_textures(Error& err);
  bool update_textures(const glz_texture* textures, uint32_t n, Error& err);
  bool build_materials(Error& err);
  bool build_lights_and_sky(Error& err);
  bool build_bvh(Error& err);

  DeviceBuffer<float4> d_vertices_, d_derivatives_;
  DeviceBuffer<uint32_t> d_indices_, d_inst_base_;
  DeviceBuffer<RTInstance> d_instances_;
  DeviceBuffer<RTMaterial> d_materials_;
  DeviceBuffer<RTLight> d_lights_;
  DeviceBuffer<TransformPair> d_transforms_;
  DeviceBuffer<TexDesc> d_tex_desc_;
  DeviceBuffer<uint8_t> d_tex_pool_;
  DeviceBuffer<float> d_srgb_lut_, d_sky_marginal_, d_sky_cond_values_, d_sky_cond_cdf_;
  DeviceBuffer<BvhNode4> d_nodes_, d_top_;
  DeviceBuffer<BvhTri> d_tris_;
  DeviceBuffer<float4> d_shade_tris_;
  DeviceBuffer<uint32_t> d_xf_identity_;
  std::vector<uint32_t> inst_base_;
  uint32_t sky_distribution_tex_ = 0xFFFFFFFFu;
};

bool hip_ok(hipError_t e, const char* what, Error& err);

}  // namespace glz
