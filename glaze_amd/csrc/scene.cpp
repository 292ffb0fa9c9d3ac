// RayTraceInstance::new / RayTraceScene::new on HIP (lib/src/vulkan/instance.rs:376-427,
// lib/src/vulkan/scene.rs:1414-1556 and the helpers it calls).
#include "scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include <thread>

#include "host_math.h"
#include "kernels.h"
#include "mipchain.h"

namespace glz {

bool hip_ok(hipError_t e, const char* what, Error& err) {
  if (e == hipSuccess) return true;
  err.code = GLZ_E_DEVICE;
  err.msg = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

Instance::~Instance() {
  if (stream) (void)hipStreamDestroy(stream);
}

Instance* Instance::create(int hip_device, Error& err) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    err.code = GLZ_E_DEVICE;
    err.msg = "no HIP device available";
    return nullptr;
  }
  int chosen = -1;
  std::string arch;
  auto arch_of = [](int d) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess) return std::string();
    return std::string(p.gcnArchName);
  };
  if (hip_device >= 0) {
    if (hip_device >= count) {
      err.code = GLZ_E_ARG;
      err.msg = "HIP device ordinal out of range";
      return nullptr;
    }
    chosen = hip_device;
    arch = arch_of(chosen);
  } else {
    for (int d = 0; d < count; ++d) {
      std::string a = arch_of(d);
      if (a.rfind("gfx950", 0) == 0) {
        chosen = d;
        arch = a;
        break;
      }
    }
  }
  // the code object is built for gfx950 only: any other device cannot run it (no fallback)
  if (chosen < 0 || arch.rfind("gfx950", 0) != 0) {
    err.code = GLZ_E_DEVICE;
    err.msg = "no gfx950 (MI355X) device found" + (arch.empty() ? std::string() : " (device is " + arch + ")");
    return nullptr;
  }
  if (!hip_ok(hipSetDevice(chosen), "hipSetDevice", err)) return nullptr;
  Instance* inst = new Instance();
  inst->device = chosen;
  inst->arch = arch;
  if (!hip_ok(hipStreamCreateWithFlags(&inst->stream, hipStreamNonBlocking), "hipStreamCreate", err)) {
    delete inst;
    return nullptr;
  }
  return inst;
}

static uint32_t sbt_callable_index(uint8_t mtype) {   // materials/material.rs:244-258
  switch (mtype) {
    case GLZ_MAT_FLAT:
    case GLZ_MAT_LAMBERT: return kBsdfLambert;
    case GLZ_MAT_MIRROR: return kBsdfMirror;
    case GLZ_MAT_GLASS: return kBsdfGlass;
    case GLZ_MAT_METAL: return kBsdfMetal;
    case GLZ_MAT_FROSTED: return kBsdfFrosted;
    default: return kBsdfUber;
  }
}

Scene* Scene::create(Instance* inst, SceneData&& data, Error& err) {
  std::unique_ptr<Scene> s(new Scene());
  s->instance = inst;
  s->data = std::move(data);
  SceneData& d = s->data;
  // defaults for absent pieces (scene.rs:1427-1434, :1467-1469; Texture::default is texture 0)
  if (d.transforms.empty()) d.transforms.push_back(identity_transform());
  if (d.materials.empty()) d.materials.push_back(default_material());
  if (d.textures.empty()) d.textures.push_back(default_texture());
  for (auto& t : d.textures) t.info.pixels = t.level0.data();
  if (!hip_ok(hipSetDevice(inst->device), "hipSetDevice", err)) return nullptr;
  // validate references so that no kernel can index out of bounds
  for (const glz_mesh& m : d.meshes) {
    if ((uint64_t)m.index_offset + m.index_count > d.indices.size() || m.index_count % 3 != 0) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "mesh index range outside the index buffer";
      return nullptr;
    }
    if (m.material >= d.materials.size()) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "mesh references a missing material";
      return nullptr;
    }
  }
  for (uint32_t i : d.indices)
    if (i >= d.vertices.size()) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "vertex index out of range";
      return nullptr;
    }
  for (const glz_mesh_instance& in : d.instances)
    if (in.transform_id >= d.transforms.size()) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "instance references a missing transform";
      return nullptr;
    }
  auto tex_ok = [&](uint32_t id) { return id < d.textures.size(); };
  for (const glz_material& m : d.materials)
    if (!tex_ok(m.diffuse) || !tex_ok(m.roughness) || !tex_ok(m.metalness) || !tex_ok(m.normal) || !tex_ok(m.opacity)) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "material references a missing texture";
      return nullptr;
    }
  if (!s->upload_geometry(err)) return nullptr;
  if (!s->build_materials(err)) return nullptr;
  if (!s->build_lights_and_sky(err)) return nullptr;
  if (!s->build_bvh(err)) return nullptr;
  if (!hip_ok(hipStreamSynchronize(inst->stream), "scene upload", err)) return nullptr;
  s->info.n_vertices = d.vertices.size();
  s->info.n_triangles = d.indices.size() / 3;
  s->info.n_instances = (uint32_t)s->h_instances.size();
  s->info.n_materials = (uint32_t)d.materials.size();
  s->info.n_lights = s->lights_no;
  s->info.n_rt_lights = (uint32_t)s->h_lights.size();
  s->info.n_textures = (uint32_t)d.textures.size();
  return s.release();
}

bool Scene::upload_geometry(Error& err) {
  hipStream_t st = instance->stream;
  SceneData& d = data;
  static_assert(sizeof(glz_vertex) == 2 * sizeof(float4), "vertex = 2 x float4");
  if (!hip_ok(d_vertices_.upload(reinterpret_cast<const float4*>(d.vertices.data()), d.vertices.size() * 2, st), "upload vertices", err)) return false;
  if (!hip_ok(d_indices_.upload(d.indices.data(), d.indices.size(), st), "upload indices", err)) return false;
  // load_raytrace_instances_to_gpu (scene.rs:1784-1818): meshes indexed by id (last one wins), dangling instances dropped
  h_instances.clear();
  for (const glz_mesh_instance& in : d.instances) {
    const glz_mesh* found = nullptr;
    for (const glz_mesh& m : d.meshes)
      if (m.id == in.mesh_id) found = &m;
    if (!found) continue;
    h_instances.push_back(RTInstance{found->index_offset, found->index_count, found->material, in.transform_id});
  }
  if (!hip_ok(d_instances_.upload(h_instances.data(), h_instances.size(), st), "upload instances", err)) return false;
  inst_base_.assign(h_instances.size(), 0);
  uint64_t total = 0;
  for (size_t i = 0; i < h_instances.size(); ++i) {
    inst_base_[i] = (uint32_t)total;
    total += h_instances[i].index_count / 3;
  }
  if (total >= 0x7FFFFFFFull) {
    err.code = GLZ_E_UNSUPPORTED;
    err.msg = "more than 2^31 world triangles";
    return false;
  }
  info.n_world_triangles = total;
  if (!hip_ok(d_inst_base_.upload(inst_base_.data(), inst_base_.size(), st), "upload instance bases", err)) return false;
  // transforms + inverses (gl_WorldToObjectEXT is supplied by the driver in the reference)
  std::vector<TransformPair> xf(d.transforms.size());
  for (size_t i = 0; i < xf.size(); ++i) {
    memcpy(xf[i].o2w, d.transforms[i].m, 64);
    host::Mat4d inv;
    if (!host::invert(host::Mat4d::from_f32(d.transforms[i].m), inv)) inv = host::Mat4d::identity();
    inv.to_f32(xf[i].w2o);
  }
  if (!hip_ok(d_transforms_.upload(xf.data(), xf.size(), st), "upload transforms", err)) return false;
  if (!upload_textures(err)) return false;
  float lut[256];
  for (int i = 0; i < 256; ++i) {   // sRGB EOTF of the R8G8B8A8_SRGB format (scene.rs:1028-1032) [ext]
    const double c = i / 255.0;
    lut[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
  }
  if (!hip_ok(d_srgb_lut_.upload(lut, 256, st), "upload sRGB LUT", err)) return false;
  // calculate_geometric_derivatives (scene.rs:2113-2188)
  uint32_t ntri = 0;
  for (const glz_mesh& m : d.meshes) ntri = std::max<uint32_t>(ntri, (m.index_offset + m.index_count) / 3);
  if (!hip_ok(d_derivatives_.alloc((size_t)ntri * 3), "alloc derivatives", err)) return false;
  if (!hip_ok(launch_derivatives(st, d_vertices_.ptr, d_indices_.ptr, ntri, d_derivatives_.ptr), "derivatives kernel", err)) return false;
  // the host staging vectors above must outlive the async copies
  if (!hip_ok(hipStreamSynchronize(st), "geometry upload", err)) return false;
  dev.vertices = d_vertices_.ptr;
  dev.indices = d_indices_.ptr;
  dev.instances = d_instances_.ptr;
  dev.transforms = d_transforms_.ptr;
  dev.derivatives = d_derivatives_.ptr;
  dev.srgb_lut = d_srgb_lut_.ptr;
  return true;
}

// Level 0 of every texture in one byte pool (16-byte aligned starts) + a descriptor per texture; what the reference
// keeps as one VkImage per texture behind a descriptor array (scene.rs:1264-1350).
bool Scene::upload_textures(Error& err) {
  hipStream_t st = instance->stream;
  SceneData& d = data;
  std::vector<TexDesc> desc(d.textures.size());
  std::vector<uint8_t> pool;
  for (size_t i = 0; i < d.textures.size(); ++i) {
    const TextureData& t = d.textures[i];
    const size_t bytes = (size_t)t.info.width * t.info.height * (t.info.format == GLZ_TEX_GRAY ? 1 : 4);
    if (t.level0.size() < bytes || t.info.width == 0 || t.info.height == 0 || t.info.format < 1 || t.info.format > 3) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "texture has inconsistent dimensions";
      return false;
    }
    // 128-byte tiles: 8 x 4 RGBA texels or 16 x 8 gray texels (device/shading.h fetch_texel); ragged edges are padded
    const bool gray = t.info.format == GLZ_TEX_GRAY;
    const uint32_t tw = gray ? 16u : 8u, th = gray ? 8u : 4u, bpp = gray ? 1u : 4u;
    const uint32_t tiles_x = (t.info.width + tw - 1) / tw, tiles_y = (t.info.height + th - 1) / th;
    const size_t tiled_bytes = (size_t)tiles_x * tiles_y * 128u;
    pool.resize((pool.size() + 127) & ~size_t(127));
    if (pool.size() + tiled_bytes > 0xFFFFFFFFull || tiles_x >= (1u << 24)) {
      err.code = GLZ_E_UNSUPPORTED;
      err.msg = "texture pool larger than 4 GiB";
      return false;
    }
    if (t.info.width == 1 && t.info.height == 1) {
      // the texel travels in the descriptor (device/shading.h kTexInline): materials' default maps cost no texel load
      uint32_t bits = 0;
      memcpy(&bits, t.level0.data(), bpp);
      desc[i] = TexDesc{bits, 1u, 1u, t.info.format | 0x80u | (1u << 8)};
      continue;
    }
    desc[i] = TexDesc{(uint32_t)pool.size(), t.info.width, t.info.height, t.info.format | (tiles_x << 8)};
    const size_t base = pool.size();
    pool.resize(base + tiled_bytes, 0);
    for (uint32_t y = 0; y < t.info.height; ++y)
      for (uint32_t x = 0; x < t.info.width; ++x) {
        const size_t dst = base + ((size_t)(y / th) * tiles_x + x / tw) * 128u + ((size_t)(y % th) * tw + x % tw) * bpp;
        memcpy(&pool[dst], &t.level0[((size_t)y * t.info.width + x) * bpp], bpp);
      }
  }
  if (!hip_ok(d_tex_desc_.upload(desc.data(), desc.size(), st), "upload texture descriptors", err)) return false;
  if (!hip_ok(d_tex_pool_.upload(pool.data(), pool.size(), st), "upload texture pool", err)) return false;
  if (!hip_ok(hipStreamSynchronize(st), "texture upload", err)) return false;   // the staging vectors above must outlive the copies
  dev.tex_desc = d_tex_desc_.ptr;
  dev.tex_pool = d_tex_pool_.ptr;
  dev.n_textures = (uint32_t)d.textures.size();
  // the texels changed: whatever mip levels were built belong to the old ones
  mips_ready_ = false;
  dev.tex_mip_desc = nullptr;
  dev.tex_mip_base = nullptr;
  dev.tex_mip_pool = nullptr;
  h_mips_.clear();
  return true;
}

// 128-byte tiles: 8 x 4 RGBA texels or 16 x 8 gray texels (upload_textures above), appended to `pool`
static TexDesc append_tiled(std::vector<uint8_t>& pool, uint32_t format, uint32_t w, uint32_t h, const uint8_t* px) {
  const bool gray = format == GLZ_TEX_GRAY;
  const uint32_t tw = gray ? 16u : 8u, th = gray ? 8u : 4u, bpp = gray ? 1u : 4u;
  const uint32_t tiles_x = (w + tw - 1) / tw, tiles_y = (h + th - 1) / th;
  pool.resize((pool.size() + 127) & ~size_t(127));
  const size_t base = pool.size();
  pool.resize(base + (size_t)tiles_x * tiles_y * 128u, 0);
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      const size_t dst = base + ((size_t)(y / th) * tiles_x + x / tw) * 128u + ((size_t)(y % th) * tw + x % tw) * bpp;
      memcpy(&pool[dst], &px[((size_t)y * w + x) * bpp], bpp);
    }
  return TexDesc{(uint32_t)base, w, h, format | (tiles_x << 8)};
}

bool Scene::ensure_mips(Error& err) {
  if (mips_ready_) return true;
  if (!hip_ok(hipSetDevice(instance->device), "hipSetDevice", err)) return false;
  const size_t nt = data.textures.size();
  h_mips_.assign(nt, {});
  std::vector<std::vector<host::MipLevel>> chains(nt);
  std::vector<std::string> decode_errs(nt);
  auto build_one = [&](size_t i) {
    TextureData& t = data.textures[i];
    if (!t.decode_more_levels(decode_errs[i])) return;   // the file's own levels 1.., still PNG-encoded since parse
    std::vector<host::MipLevel> given(1);
    given[0].width = t.info.width;
    given[0].height = t.info.height;
    given[0].pixels = t.level0;
    for (size_t l = 0; l < t.more_levels.size(); ++l) {
      host::MipLevel m;
      m.width = t.more_dims[2 * l];
      m.height = t.more_dims[2 * l + 1];
      m.pixels = t.more_levels[l];
      given.push_back(std::move(m));
    }
    chains[i] = host::build_mip_chain(t.info.format, std::move(given));
  };
  const unsigned workers = (unsigned)std::min<size_t>(nt, std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
  if (workers <= 1) {
    for (size_t i = 0; i < nt; ++i) build_one(i);
  } else {
    std::vector<std::thread> pool;
    std::vector<std::string> failed(workers);
    for (unsigned w = 0; w < workers; ++w)
      pool.emplace_back([&, w] {
        try {
          for (size_t i = w; i < nt; i += workers) build_one(i);
        } catch (const std::exception& ex) {
          failed[w] = ex.what();
        }
      });
    for (auto& th : pool) th.join();
    for (auto& f : failed)
      if (!f.empty()) {
        err.code = GLZ_E_IO;
        err.msg = "mip chain: " + f;
        return false;
      }
  }
  for (size_t i = 0; i < nt; ++i)
    if (!decode_errs[i].empty()) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "Corrupted image: " + decode_errs[i];
      return false;
    }
  std::vector<uint8_t> pool;
  std::vector<TexDesc> desc;
  std::vector<uint32_t> base(nt, 0);
  for (size_t i = 0; i < nt; ++i) {
    const uint32_t levels = (uint32_t)chains[i].size();
    base[i] = (uint32_t)desc.size() | (levels << 24);
    for (uint32_t l = 1; l < levels; ++l) {
      const host::MipLevel& m = chains[i][l];
      desc.push_back(append_tiled(pool, data.textures[i].info.format, m.width, m.height, m.pixels.data()));
      h_mips_[i].push_back(m.pixels);
    }
    if (pool.size() > 0xFFFFFFF0ull || desc.size() >= (1u << 24)) {
      err.code = GLZ_E_UNSUPPORTED;
      err.msg = "mip pool larger than 4 GiB";
      return false;
    }
  }
  hipStream_t st = instance->stream;
  if (!hip_ok(d_tex_mip_desc_.upload(desc.data(), desc.size(), st), "upload mip descriptors", err)) return false;
  if (!hip_ok(d_tex_mip_pool_.upload(pool.data(), pool.size(), st), "upload mip pool", err)) return false;
  if (!hip_ok(d_tex_mip_base_.upload(base.data(), base.size(), st), "upload mip table", err)) return false;
  if (!hip_ok(hipStreamSynchronize(st), "mip upload", err)) return false;
  dev.tex_mip_desc = d_tex_mip_desc_.ptr;
  dev.tex_mip_pool = d_tex_mip_pool_.ptr;
  dev.tex_mip_base = d_tex_mip_base_.ptr;
  mips_ready_ = true;
  return true;
}

bool Scene::read_mip_level(uint32_t texture, uint32_t level, std::vector<uint8_t>& pixels, uint32_t& w, uint32_t& h, Error& err) {
  pixels.clear();
  w = h = 0;
  if (texture >= data.textures.size()) return true;
  const TextureData& t = data.textures[texture];
  if (level == 0) {
    pixels = t.level0;
    w = t.info.width;
    h = t.info.height;
    return true;
  }
  if (!ensure_mips(err)) return false;
  if (level - 1 >= h_mips_[texture].size()) return true;
  pixels = h_mips_[texture][level - 1];
  w = host::mip_dim(t.info.width, level);
  h = host::mip_dim(t.info.height, level);
  return true;
}

// load_raytrace_materials_to_gpu (scene.rs:1821-1860)
bool Scene::build_materials(Error& err) {
  h_materials.clear();
  for (const glz_material& m : data.materials) {
    RTMaterial r{};
    for (int i = 0; i < 3; ++i) {
      r.diffuse_mul[i] = (float)m.diffuse_mul[i] / 255.0f;   // col_int_to_f32, scene.rs:1930-1937
      r.emissive_col[i] = m.has_emissive ? (float)m.emissive_col[i] / 255.0f : 0.0f;
    }
    r.diffuse_mul[3] = r.emissive_col[3] = 1.0f;
    const unsigned metal = m.metal < GLZ_METAL_COUNT ? m.metal : 0;
    for (int i = 0; i < 16; ++i) {
      const float n = GLZ_METAL_N[metal][i], k = GLZ_METAL_K[metal][i];
      r.metal_ior.w[i] = n;
      r.metal_fresnel.w[i] = (n * n) + (k * k);
    }
    r.diffuse = m.diffuse;
    r.roughness = m.roughness;
    r.metalness = m.metalness;
    r.opacity = m.opacity;
    r.normal = m.normal;
    r.bsdf_index = sbt_callable_index(m.mtype);
    r.roughness_mul = m.roughness_mul;
    r.metalness_mul = m.metalness_mul;
    r.anisotropy = m.anisotropy;
    r.ior_dielectric = m.ior;
    r.is_specular = (m.mtype == GLZ_MAT_MIRROR || m.mtype == GLZ_MAT_GLASS) ? 1u : 0u;   // material.rs:103-114
    r.is_emissive = m.has_emissive ? 1u : 0u;
    h_materials.push_back(r);
  }
  if (!hip_ok(d_materials_.upload(h_materials.data(), h_materials.size(), instance->stream), "upload materials", err)) return false;
  if (!hip_ok(hipStreamSynchronize(instance->stream), "materials upload", err)) return false;
  dev.materials = d_materials_.ptr;
  dev.n_materials = (uint32_t)h_materials.size();
  dev.has_non_opaque = 0u;   // which k_trace runs (launch_trace): the one without alpha code, or the one with the alpha phase
  for (const RTMaterial& m : h_materials) dev.has_non_opaque |= m.opacity != 0u ? 1u : 0u;
  return true;
}

// reorder_lights (scene.rs:628-635), load_raytrace_lights_to_gpu (:1863-1927),
// calculate_skymap_distributions + build_sky_raytrace_buffers (:2191-2313)
bool Scene::build_lights_and_sky(Error& err) {
  hipStream_t st = instance->stream;
  std::vector<glz_light> ordered;
  const glz_light* sky = nullptr;
  for (const glz_light& l : data.lights) {
    if (l.ltype > GLZ_LIGHT_SKY) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "Invalid enum value for LightType";
      return false;
    }
    if (l.ltype == GLZ_LIGHT_SKY) { if (!sky) sky = &l; }
    else ordered.push_back(l);
  }
  const bool has_sky = sky != nullptr;
  if (sky) ordered.push_back(*sky);
  lights_no = (uint32_t)ordered.size();   // scene.rs:1549 -- NOT the number of expanded RTLights
  h_lights.clear();
  for (const glz_light& l : ordered) {
    RTLight r{};
    memcpy(r.color.w, l.color, 64);
    float dx = l.direction[0], dy = l.direction[1], dz = l.direction[2];
    if (dx == 0.0f && dy == 0.0f && dz == 0.0f) dy = -1.0f;
    // the reference calls `dir.normalize()` and drops the result: directions stay un-normalised (Q14)
    r.pos[0] = l.position[0]; r.pos[1] = l.position[1]; r.pos[2] = l.position[2];
    r.dir[0] = dx; r.dir[1] = dy; r.dir[2] = dz;
    r.shader = l.ltype;   // sbt_callable_index, light.rs:111-119
    r.instance_id = 0xFFFFFFFFu;
    r.intensity = l.intensity;
    r.delta = (l.ltype == GLZ_LIGHT_OMNI || l.ltype == GLZ_LIGHT_SUN) ? 1u : 0u;
    if (l.ltype == GLZ_LIGHT_AREA) {
      // one RTLight per instance whose mesh uses the emitting material (map_materials_to_instances, :1764-1781)
      const uint16_t material_id = (uint16_t)l.resource_id;
      std::vector<uint32_t> ids;
      for (size_t i = 0; i < data.instances.size(); ++i) {
        const glz_mesh* found = nullptr;
        for (const glz_mesh& m : data.meshes)
          if (m.id == data.instances[i].mesh_id) found = &m;
        if (found && found->material == material_id) ids.push_back((uint32_t)(uint16_t)i);
      }
      if (ids.empty()) ids.push_back(0);
      for (uint32_t id : ids) {
        if (id >= h_instances.size()) {
          err.code = GLZ_E_INVALID_DATA;
          err.msg = "area light refers to a missing instance";
          return false;
        }
        r.instance_id = id;
        h_lights.push_back(r);
      }
    } else {
      h_lights.push_back(r);
    }
  }
  if (h_lights.empty()) {   // dummy light so the buffer is never empty (:1906-1918)
    RTLight r{};
    r.instance_id = 0xFFFFFFFFu;
    r.intensity = 1.0f;
    r.delta = 1u;
    h_lights.push_back(r);
  }
  if (!hip_ok(d_lights_.upload(h_lights.data(), h_lights.size(), st), "upload lights", err)) return false;
  dev.lights = d_lights_.ptr;
  dev.n_rt_lights = (uint32_t)h_lights.size();

  // sky: Light::default() when the scene has none (scene.rs:2249)
  glz_light dflt{};
  dflt.intensity = 1.0f;
  const glz_light& sl = has_sky ? ordered.back() : dflt;
  if (sl.resource_id >= data.textures.size()) {
    err.code = GLZ_E_INVALID_DATA;
    err.msg = "sky light references a missing texture";
    return false;
  }
  float rot32[16];
  host::sky_rotation(sl.yaw_deg, sl.pitch_deg, sl.roll_deg).to_f32(rot32);   // Matrix4<f32> in the reference
  host::Mat4d inv;
  if (!host::invert(host::Mat4d::from_f32(rot32), inv)) inv = host::Mat4d::identity();
  memcpy(h_sky.obj2world, rot32, 64);
  inv.to_f32(h_sky.world2obj);
  h_sky.tex_id = sl.resource_id;
  h_sky.intensity = sl.intensity;
  if (sky_distribution_tex_ != h_sky.tex_id) {
    // recalculated only when the sky texture changes (scene.rs:2256-2260)
    const TextureData& map = data.textures[h_sky.tex_id];
    const uint32_t W = map.info.width, H = map.info.height;
    const size_t bpp = map.info.format == GLZ_TEX_GRAY ? 1 : 4;
    std::vector<float> values((size_t)W * H);
    const float pi = 3.14159265358979323846f;
    for (uint32_t y = 0; y < H; ++y) {
      const float sint = sinf(pi * ((float)y + 0.5f) / (float)H);
      const uint8_t* row = map.level0.data() + (size_t)y * W * bpp;
      for (uint32_t x = 0; x < W; ++x) {
        const uint8_t* p = row + x * bpp;
        const float r = (float)p[0] / 255.0f, g = (float)p[bpp > 1 ? 1 : 0] / 255.0f, b = (float)p[bpp > 1 ? 2 : 0] / 255.0f;
        values[(size_t)y * W + x] = host::illuminant_luminance(r, g, b) * sint;
      }
    }
    std::vector<float> cond_cdf, integrals(H), marginal_cdf;
    cond_cdf.reserve((size_t)(W + 1) * H);
    for (uint32_t y = 0; y < H; ++y) integrals[y] = host::distribution1d(values.data() + (size_t)y * W, W, cond_cdf);
    h_sky_header.marginal_integral = host::distribution1d(integrals.data(), H, marginal_cdf);
    h_sky_header.marginal_cdf_count = H + 1;
    h_sky_header.conditional_integral_offset = H + (H + 1);
    h_sky_header.conditional_cdf_count = W + 1;
    h_sky_marginal = marginal_cdf;
    h_sky_marginal.insert(h_sky_marginal.end(), integrals.begin(), integrals.end());   // marginal values
    h_sky_marginal.insert(h_sky_marginal.end(), integrals.begin(), integrals.end());   // conditional integrals
    if (!hip_ok(d_sky_marginal_.upload(h_sky_marginal.data(), h_sky_marginal.size(), st), "upload sky marginal", err)) return false;
    if (!hip_ok(d_sky_cond_values_.upload(values.data(), values.size(), st), "upload sky values", err)) return false;
    if (!hip_ok(d_sky_cond_cdf_.upload(cond_cdf.data(), cond_cdf.size(), st), "upload sky cdf", err)) return false;
    if (!hip_ok(hipStreamSynchronize(st), "sky upload", err)) return false;
    dev.sky_w = W;
    dev.sky_h = H;
    sky_distribution_tex_ = h_sky.tex_id;
  }
  if (!hip_ok(hipStreamSynchronize(st), "lights upload", err)) return false;
  dev.sky = h_sky;
  dev.sky_header = h_sky_header;
  dev.sky_marginal = d_sky_marginal_.ptr;
  dev.sky_cdf = d_sky_marginal_.ptr;
  dev.sky_cond_values = d_sky_cond_values_.ptr;
  dev.sky_cond_cdf = d_sky_cond_cdf_.ptr;
  return true;
}

// Two-level structure (types.h TlasInstance; acceleration.rs:319-345: a BLAS per mesh, a TLAS over the instances).  The meshes'
// hierarchies are built by the ordinary builder over ONE pseudo-instance with the identity transform (object space = its "world"),
// the top level by the same builder over the instances' world boxes (LbvhInputs::given_lo).  Memory is O(meshes + instances).
bool Scene::build_two_level(Error& err) {
  hipStream_t st = instance->stream;
  hipEvent_t e0, e1;
  if (!hip_ok(hipEventCreate(&e0), "event", err) || !hip_ok(hipEventCreate(&e1), "event", err)) return false;
  (void)hipEventRecord(e0, st);
  h_mesh_ranges.clear();
  // ---- unique meshes (index ranges) ----
  struct MeshAs {
    uint32_t index_offset, index_count, tri_base, node_base, n_nodes, depth;
    BvhGrid grid;
    float lo[3], hi[3];
    BvhNode4* nodes;
    BvhQuad* quads = nullptr;   // the mesh's leaf records (object space), as build_lbvh handed them out
    uint32_t n_leaves = 0, quad_base = 0;
  };
  std::vector<MeshAs> meshes;
  std::vector<uint32_t> mesh_of(h_instances.size());
  for (size_t i = 0; i < h_instances.size(); ++i) {
    const RTInstance& in = h_instances[i];
    size_t m = 0;
    while (m < meshes.size() && !(meshes[m].index_offset == in.index_offset && meshes[m].index_count == in.index_count)) ++m;
    if (m == meshes.size()) meshes.push_back(MeshAs{in.index_offset, in.index_count, 0, 0, 0, 0, BvhGrid{}, {0, 0, 0}, {0, 0, 0}, nullptr});
    mesh_of[i] = (uint32_t)m;
  }
  uint64_t total_tris = 0;
  for (MeshAs& m : meshes) {
    m.tri_base = (uint32_t)total_tris;
    total_tris += m.index_count / 3u;
  }
  if (total_tris >= 0x3FFFFFFFull) {
    err.code = GLZ_E_UNSUPPORTED;
    err.msg = "more than 2^30 mesh triangles";
    return false;
  }
  auto free_nodes = [&]() {
    for (MeshAs& m : meshes) {
      if (m.nodes) (void)hipFree(m.nodes);
      if (m.quads) (void)hipFree(m.quads);
      m.nodes = nullptr;
      m.quads = nullptr;
    }
  };
  const uint32_t nt = (uint32_t)total_tris;
  d_nodes_.release();
  if (!hip_ok(d_tris_.alloc((size_t)nt + 1), "alloc BVH triangles", err)) return false;
  if (!hip_ok(hipMemsetAsync(d_tris_.ptr + nt, 0, sizeof(BvhTri), st), "clear BVH triangle padding", err)) return false;
  if (!hip_ok(d_shade_tris_.alloc((size_t)nt * 8), "alloc shading records", err)) return false;
  // helpers of the per-mesh builds: an identity transform, an opaque material (the opacity flag is per instance here)
  DeviceBuffer<TransformPair> d_ident;
  DeviceBuffer<RTMaterial> d_opaque;
  DeviceBuffer<RTInstance> d_pseudo;
  DeviceBuffer<uint32_t> d_zero, d_one;
  {
    TransformPair ident{};
    for (int k = 0; k < 4; ++k) ident.o2w[5 * k] = ident.w2o[5 * k] = 1.0f;
    RTMaterial opaque{};
    const uint32_t zero = 0u, one = 1u;
    if (!hip_ok(d_ident.upload(&ident, 1, st), "upload", err) || !hip_ok(d_opaque.upload(&opaque, 1, st), "upload", err) ||
        !hip_ok(d_zero.upload(&zero, 1, st), "upload", err) || !hip_ok(d_one.upload(&one, 1, st), "upload", err) || !hip_ok(d_pseudo.alloc(1), "alloc", err))
      return false;
    if (!hip_ok(hipStreamSynchronize(st), "upload", err)) return false;
  }
  uint32_t total_nodes = 0, max_depth = 0;
  for (MeshAs& m : meshes) {
    const uint32_t n = m.index_count / 3u;
    if (n == 0) continue;
    const RTInstance pseudo{m.index_offset, m.index_count, 0u, 0u};
    if (!hip_ok(hipMemcpyAsync(d_pseudo.ptr, &pseudo, sizeof(pseudo), hipMemcpyHostToDevice, st), "upload", err) || !hip_ok(hipStreamSynchronize(st), "upload", err)) {
      free_nodes();
      return false;
    }
    LbvhInputs in{d_vertices_.ptr, d_indices_.ptr, d_pseudo.ptr, d_zero.ptr, 1u, d_ident.ptr, d_opaque.ptr, n, instance->bvh_builder, instance->bvh_pair_area_ratio};
    in.emit_quads = true;   // 64-byte leaf records (object space here), leaf links ~leaf number relative to the mesh
    LbvhOutputs out{};
    DeviceBuffer<BvhTri> tris;
    if (!hip_ok(tris.alloc((size_t)n + 1), "alloc mesh triangles", err)) { free_nodes(); return false; }
    out.tris = tris.ptr;
    const hipError_t be = build_lbvh(st, in, out);
    m.quads = out.quads;
    m.n_leaves = out.quads ? out.n_leaves : 0u;
    if (!hip_ok(be, "mesh hierarchy", err)) { if (out.nodes) (void)hipFree(out.nodes); free_nodes(); return false; }
    m.nodes = out.nodes;
    m.n_nodes = out.n_nodes;
    m.depth = out.depth;
    m.grid = out.grid;
    for (int k = 0; k < 3; ++k) { m.lo[k] = out.bounds_lo[k]; m.hi[k] = out.bounds_hi[k]; }
    total_nodes += out.n_nodes;
    max_depth = std::max(max_depth, out.depth);
    // the mesh's triangles and per-object-triangle shading records go to their place in the concatenated arrays
    if (!hip_ok(hipMemcpyAsync(d_tris_.ptr + m.tri_base, tris.ptr, sizeof(BvhTri) * n, hipMemcpyDeviceToDevice, st), "copy mesh triangles", err) ||
        !hip_ok(launch_shade_records(st, n, tris.ptr, d_pseudo.ptr, d_indices_.ptr, d_vertices_.ptr, d_derivatives_.ptr, d_one.ptr,
                                     d_shade_tris_.ptr + 8 * (size_t)m.tri_base), "k_shade_records", err) ||
        !hip_ok(hipStreamSynchronize(st), "mesh hierarchy", err)) {
      free_nodes();
      return false;
    }
  }
  // ---- the meshes' leaf records, one array (TlasInstance::quad_base) ----
  {
    uint32_t total_leaves = 0;
    for (MeshAs& m : meshes) {
      m.quad_base = total_leaves;
      total_leaves += m.n_leaves;
    }
    d_quads_.release();
    bool ok = hip_ok(d_quads_.alloc((size_t)total_leaves + 1), "alloc leaf records", err);
    if (ok) ok = hip_ok(hipMemsetAsync(d_quads_.ptr + total_leaves, 0, sizeof(BvhQuad), st), "clear leaf record padding", err);
    for (MeshAs& m : meshes)
      if (ok && m.n_leaves)
        ok = hip_ok(hipMemcpyAsync(d_quads_.ptr + m.quad_base, m.quads, sizeof(BvhQuad) * m.n_leaves, hipMemcpyDeviceToDevice, st), "copy leaf records", err);
    if (!ok) { (void)hipStreamSynchronize(st); free_nodes(); return false; }
  }
  // ---- instance boxes: world AABB of the mesh's VERTICES under the instance's transform, padded.  (The eight corners of the
  // mesh's object box, which is what a top level normally takes, give a box up to sqrt(2) too wide per axis for a rotated instance
  // -- a quarter more instance entries per ray in the column forest.  The vertices cost meshes x instances x vertices on the host;
  // past kExactBoxBudget point transforms the remaining instances fall back to the corners.) ----
  const size_t ni = h_instances.size();
  std::vector<float4> blo(ni), bhi(ni);
  double wlo[3] = {1e300, 1e300, 1e300}, whi[3] = {-1e300, -1e300, -1e300};
  std::vector<std::vector<float>> mesh_points(meshes.size());   // xyz of the vertices a mesh references, once each
  {
    std::vector<uint8_t> seen(data.vertices.size(), 0);
    for (size_t mi = 0; mi < meshes.size(); ++mi) {
      const MeshAs& m = meshes[mi];
      std::vector<float>& pts = mesh_points[mi];
      for (uint32_t k = 0; k < m.index_count; ++k) {
        const uint32_t v = data.indices[m.index_offset + k];
        if (seen[v]) continue;
        seen[v] = 1;
        pts.insert(pts.end(), data.vertices[v].vv, data.vertices[v].vv + 3);
      }
      for (uint32_t k = 0; k < m.index_count; ++k) seen[data.indices[m.index_offset + k]] = 0;
    }
  }
  constexpr uint64_t kExactBoxBudget = 50000000ull;
  uint64_t spent = 0;
  for (size_t i = 0; i < ni; ++i) {
    const MeshAs& m = meshes[mesh_of[i]];
    const float* M = data.transforms[h_instances[i].transform_id].m;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    auto take = [&](double x, double y, double z) {
      for (int k = 0; k < 3; ++k) {
        const double w = (double)M[k] * x + (double)M[4 + k] * y + (double)M[8 + k] * z + (double)M[12 + k];
        lo[k] = std::min(lo[k], w);
        hi[k] = std::max(hi[k], w);
      }
    };
    const std::vector<float>& pts = mesh_points[mesh_of[i]];
    if (!pts.empty() && spent + pts.size() / 3 <= kExactBoxBudget) {
      spent += pts.size() / 3;
      for (size_t k = 0; k + 2 < pts.size(); k += 3) take(pts[k], pts[k + 1], pts[k + 2]);
    } else {
      for (int c = 0; c < 8; ++c) take((c & 1) ? m.hi[0] : m.lo[0], (c & 2) ? m.hi[1] : m.lo[1], (c & 4) ? m.hi[2] : m.lo[2]);
    }
    float l[3], h[3];
    for (int k = 0; k < 3; ++k) {
      if (!(lo[k] <= hi[k])) lo[k] = hi[k] = 0.0;   // a mesh without triangles / non-finite transform: a point nobody hits
      // the tracer's world vertices are single-precision sums of four terms: four roundings of the largest one can get
      const double mag = std::fabs((double)M[k]) * std::max(std::fabs((double)m.lo[0]), std::fabs((double)m.hi[0])) +
                         std::fabs((double)M[4 + k]) * std::max(std::fabs((double)m.lo[1]), std::fabs((double)m.hi[1])) +
                         std::fabs((double)M[8 + k]) * std::max(std::fabs((double)m.lo[2]), std::fabs((double)m.hi[2])) + std::fabs((double)M[12 + k]);
      const double pad = 1e-5 * std::max({std::fabs(lo[k]), std::fabs(hi[k]), 1e-3}) + 1e-6 * (hi[k] - lo[k]) + (std::isfinite(mag) ? 4.8e-7 * mag : 0.0);
      l[k] = std::nextafterf((float)(lo[k] - pad), -INFINITY);
      h[k] = std::nextafterf((float)(hi[k] + pad), INFINITY);
      wlo[k] = std::min(wlo[k], (double)l[k]);
      whi[k] = std::max(whi[k], (double)h[k]);
    }
    blo[i] = make_float4(l[0], l[1], l[2], 0.0f);
    bhi[i] = make_float4(h[0], h[1], h[2], 0.0f);
  }
  DeviceBuffer<float4> d_blo, d_bhi;
  DeviceBuffer<BvhTri> top_tris;
  LbvhOutputs top{};
  if (!hip_ok(d_blo.upload(blo.data(), ni, st), "upload instance boxes", err) || !hip_ok(d_bhi.upload(bhi.data(), ni, st), "upload instance boxes", err) ||
      !hip_ok(top_tris.alloc(ni + 1), "alloc", err) || !hip_ok(hipStreamSynchronize(st), "upload instance boxes", err)) {
    free_nodes();
    return false;
  }
  {
    LbvhInputs in{nullptr, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, (uint32_t)ni, instance->bvh_builder, 0.0f};
    in.given_lo = d_blo.ptr;
    in.given_hi = d_bhi.ptr;
    top.tris = top_tris.ptr;
    if (!hip_ok(build_lbvh(st, in, top), "instance hierarchy", err)) { free_nodes(); return false; }
  }
  // ---- one node array: top level first, then the meshes ----
  for (MeshAs& m : meshes)
    if (m.n_nodes == 0) total_nodes += 1;   // a mesh without triangles gets one node without children: entering it finds nothing
  const uint32_t n_nodes = top.n_nodes + total_nodes;
  if (n_nodes >= (uint32_t)kBvhTopFlag) {
    free_nodes();
    if (top.nodes) (void)hipFree(top.nodes);
    err.code = GLZ_E_UNSUPPORTED;
    err.msg = "more than 2^30 BVH nodes";
    return false;
  }
  bool ok = hip_ok(d_nodes_.alloc(n_nodes), "alloc nodes", err);
  if (ok && top.n_nodes) ok = hip_ok(hipMemcpyAsync(d_nodes_.ptr, top.nodes, sizeof(BvhNode4) * top.n_nodes, hipMemcpyDeviceToDevice, st), "copy nodes", err);
  uint32_t at = top.n_nodes;
  for (MeshAs& m : meshes) {
    m.node_base = at;
    if (ok && m.n_nodes) ok = hip_ok(hipMemcpyAsync(d_nodes_.ptr + at, m.nodes, sizeof(BvhNode4) * m.n_nodes, hipMemcpyDeviceToDevice, st), "copy nodes", err);
    if (ok && m.n_nodes == 0) {
      static BvhNode4 childless;
      for (int k = 0; k < 4; ++k) {
        childless.w[3 * k] = childless.w[3 * k + 1] = childless.w[3 * k + 2] = kBvhGridMax;
        childless.w[12 + k] = (uint32_t)kBvhEmptyChild;
      }
      ok = hip_ok(hipMemcpyAsync(d_nodes_.ptr + at, &childless, sizeof(BvhNode4), hipMemcpyHostToDevice, st), "copy nodes", err);
    }
    h_mesh_ranges.push_back(MeshRange{m.node_base, m.n_nodes, m.quad_base, m.tri_base});
    at += m.n_nodes ? m.n_nodes : 1u;
  }
  // instance records in the top level's leaf order
  std::vector<BvhTri> order(ni);
  if (ok) ok = hip_ok(hipMemcpyAsync(order.data(), top_tris.ptr, sizeof(BvhTri) * ni, hipMemcpyDeviceToHost, st), "read instance order", err);
  if (ok) ok = hip_ok(hipStreamSynchronize(st), "two-level build", err);
  free_nodes();
  if (top.nodes) (void)hipFree(top.nodes);
  if (!ok) return false;
  const double reach = std::sqrt((whi[0] - wlo[0]) * (whi[0] - wlo[0]) + (whi[1] - wlo[1]) * (whi[1] - wlo[1]) + (whi[2] - wlo[2]) * (whi[2] - wlo[2])) +
                       std::max({std::fabs(wlo[0]), std::fabs(wlo[1]), std::fabs(wlo[2]), std::fabs(whi[0]), std::fabs(whi[1]), std::fabs(whi[2])});
  std::vector<TlasInstance> recs(ni);
  std::vector<TransformPair> xf(data.transforms.size());
  for (size_t t = 0; t < xf.size(); ++t) {
    memcpy(xf[t].o2w, data.transforms[t].m, 64);
    host::Mat4d inv;
    if (!host::invert(host::Mat4d::from_f32(data.transforms[t].m), inv)) inv = host::Mat4d::identity();
    inv.to_f32(xf[t].w2o);
  }
  for (size_t s = 0; s < ni; ++s) {
    const uint32_t i = order[s].world_id;   // the box this leaf was built from
    const RTInstance& in = h_instances[i];
    const MeshAs& m = meshes[mesh_of[i]];
    const TransformPair& T = xf[in.transform_id];
    TlasInstance r{};
    for (int row = 0; row < 3; ++row)
      for (int col = 0; col < 4; ++col) r.w2o[4 * row + col] = T.w2o[4 * col + row];   // rows of the column-major inverse
    memcpy(r.o2w, T.o2w, 64);
    r.grid = m.grid;
    // slack of the object-space box tests: rounding of the transformed ray (relative to where in the world it can be) and of
    // the triangle's world position, taken back to object units; generous (x 32)
    double wnorm = 0.0, tr = 0.0, objmax = 0.0;
    for (int row = 0; row < 3; ++row) {
      wnorm = std::max(wnorm, std::fabs((double)r.w2o[4 * row]) + std::fabs((double)r.w2o[4 * row + 1]) + std::fabs((double)r.w2o[4 * row + 2]));
      tr = std::max(tr, std::fabs((double)r.w2o[4 * row + 3]));
      objmax = std::max({objmax, std::fabs((double)m.lo[row]), std::fabs((double)m.hi[row])});
    }
    const double eps = 1.1920929e-7;
    // (in object units: the tracer turns it into cells of each axis -- a thin mesh has tiny cells across its thin side, and the pad
    // that side needs, taken for all three, would make every box of the mesh as wide as the mesh)
    r.slack = (float)(32.0 * eps * (wnorm * reach + tr + objmax));
    r.w2o_norm = (float)wnorm;   // the tracer adds 32 eps wnorm |o|_1 per ray: origins far outside the scene's bounds
    r.node_base = m.node_base;
    r.tri_base = m.tri_base;
    r.quad_base = m.quad_base;
    r.world_base = inst_base_[i];
    r.instance = i;
    r.non_opaque = h_materials[in.material_id].opacity != 0 ? 1u : 0u;
    recs[s] = r;
  }
  if (!hip_ok(d_tlas_instances_.upload(recs.data(), ni, st), "upload instance records", err)) return false;
  std::vector<uint32_t> ident(data.transforms.size(), 0u);
  for (size_t i = 0; i < ident.size(); ++i) {
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    ident[i] = memcmp(data.transforms[i].m, id, 64) == 0 ? 1u : 0u;
  }
  if (!hip_ok(d_xf_identity_.upload(ident.data(), ident.size(), st), "upload transform flags", err)) return false;
  if (!d_top_.ptr && !hip_ok(d_top_.alloc(kBvhTopNodes), "alloc BVH top table", err)) return false;
  if (!hip_ok(launch_top_table(st, d_nodes_.ptr, top.n_nodes, d_top_.ptr), "k_top_table", err)) return false;   // unused by the two-level tracer, kept valid
  (void)hipEventRecord(e1, st);
  if (!hip_ok(hipStreamSynchronize(st), "two-level build", err)) return false;
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  info.bvh_nodes = n_nodes;
  info.bvh_depth = top.depth + max_depth;
  info.bvh_sah_cost = top.sah;
  info.build_ms = ms;
  for (int k = 0; k < 3; ++k) {
    info.bounds_min[k] = top.bounds_lo[k];
    info.bounds_max[k] = top.bounds_hi[k];
    info.bvh_grid_lo[k] = top.grid.lo[k];
    info.bvh_grid_cell[k] = top.grid.cell[k];
  }
  // both levels' entries and one exit marker share the stack
  const uint32_t stack_bound = 3u * (top.depth + max_depth) + 2u;
  stack_overflow_depth = stack_bound > (uint32_t)kTraversalLdsStack ? stack_bound - kTraversalLdsStack + 1 : 1;
  dev.bvh_nodes = d_nodes_.ptr;
  dev.bvh_top = d_top_.ptr;
  dev.bvh_grid = top.grid;
  dev.bvh_tris = d_tris_.ptr;
  dev.bvh_quads = d_quads_.ptr;   // the meshes' leaf records, concatenated
  dev.shade_tris = d_shade_tris_.ptr;
  dev.tlas_nodes = d_nodes_.ptr;
  dev.tlas_instances = d_tlas_instances_.ptr;
  dev.xf_identity = d_xf_identity_.ptr;
  dev.two_level = 1u;
  d_nodes8_.release();   // (the two-level tracer walks 4-wide nodes only)
  dev.bvh_nodes8 = nullptr;
  info.bvh_nodes8 = 0;
  dev.n_world_tris = (uint32_t)info.n_world_triangles;
  info.as_levels = 2;
  info.n_as_triangles = nt;
  info.as_bytes = (uint64_t)n_nodes * sizeof(BvhNode4) + (uint64_t)(nt + 1) * sizeof(BvhTri) + (uint64_t)nt * 128u + (uint64_t)ni * sizeof(TlasInstance) +
                  (uint64_t)d_quads_.count * sizeof(BvhQuad);
  return true;
}

bool Scene::build_bvh(Error& err) {
  // instanced scenes keep one hierarchy per mesh (acceleration.rs:319-345) instead of one over every instanced triangle
  {
    uint64_t unique = 0;
    std::vector<std::pair<uint32_t, uint32_t>> seen;
    for (const RTInstance& in : h_instances) {
      const std::pair<uint32_t, uint32_t> key(in.index_offset, in.index_count);
      if (std::find(seen.begin(), seen.end(), key) == seen.end()) {
        seen.push_back(key);
        unique += in.index_count / 3u;
      }
    }
    const bool wanted = instance->as_levels == 2 || (instance->as_levels == 0 && unique > 0 && info.n_world_triangles > 4 * unique);
    dev.two_level = 0u;
    dev.tlas_nodes = nullptr;
    dev.tlas_instances = nullptr;
    if (wanted && info.n_world_triangles > 0) return build_two_level(err);
  }
  hipStream_t st = instance->stream;
  const uint32_t n = (uint32_t)info.n_world_triangles;
  d_nodes_.release();
  if (!hip_ok(d_tris_.alloc((size_t)n + 1), "alloc BVH triangles", err)) return false;   // + 1: the tracer loads a leaf's partner slot unconditionally
  if (!hip_ok(hipMemsetAsync(d_tris_.ptr + n, 0, sizeof(BvhTri), st), "clear BVH triangle padding", err)) return false;
  LbvhInputs in{d_vertices_.ptr, d_indices_.ptr, d_instances_.ptr, d_inst_base_.ptr, (uint32_t)h_instances.size(), d_transforms_.ptr,
                d_materials_.ptr, n, instance->bvh_builder, instance->bvh_pair_area_ratio};
  in.emit_quads = true;   // the flattened tracer reads one 64-byte record per leaf (types.h BvhQuad); leaf links are ~leaf number
  in.emit_wide8 = true;   // ... and a small tile share the same hierarchy eight wide (types.h BvhNode8, k_trace8)
  LbvhOutputs out{};
  out.tris = d_tris_.ptr;
  d_quads_.release();
  d_nodes8_.release();
  hipEvent_t e0, e1;
  if (!hip_ok(hipEventCreate(&e0), "event", err) || !hip_ok(hipEventCreate(&e1), "event", err)) return false;
  (void)hipEventRecord(e0, st);
  const hipError_t be = build_lbvh(st, in, out);
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.0f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  d_quads_.ptr = out.quads;       // ours whatever the build returned
  d_quads_.count = out.quads ? out.n_leaves : 0;
  d_nodes8_.ptr = out.nodes8;
  d_nodes8_.count = out.nodes8 ? out.n_nodes8 : 0;
  if (!hip_ok(be, "LBVH build", err)) return false;
  d_nodes_.ptr = out.nodes;       // allocated by the build once the number of 4-wide nodes is known
  d_nodes_.count = out.n_nodes;
  info.bvh_nodes = out.n_nodes;
  info.bvh_depth = out.depth;
  info.bvh_sah_cost = out.sah;
  info.build_ms = ms;
  for (int k = 0; k < 3; ++k) {
    info.bounds_min[k] = out.bounds_lo[k];
    info.bounds_max[k] = out.bounds_hi[k];
  }
  // traversal stack: kTraversalLdsStack levels live in LDS, the rest spills to a per-lane HBM area
  // (a 4-wide visit pushes up to three siblings, so the bound is 3 entries per level)
  const uint32_t stack_bound = std::max(3u * out.depth, 7u * out.depth8) + 1u;   // (an 8-wide visit pushes up to seven)
  stack_overflow_depth = stack_bound > (uint32_t)kTraversalLdsStack ? stack_bound - kTraversalLdsStack + 1 : 1;
  dev.bvh_nodes = d_nodes_.ptr;
  if (out.n_nodes >= (uint32_t)kBvhTopFlag) {
    err.code = GLZ_E_UNSUPPORTED;
    err.msg = "more than 2^30 BVH nodes";
    return false;
  }
  // top levels of the tree for the tracers' LDS staging
  if (!d_top_.ptr && !hip_ok(d_top_.alloc(kBvhTopNodes), "alloc BVH top table", err)) return false;
  if (!hip_ok(launch_top_table(st, d_nodes_.ptr, out.n_nodes, d_top_.ptr), "k_top_table", err)) return false;
  dev.bvh_top = d_top_.ptr;
  dev.bvh_nodes8 = d_nodes8_.ptr;
  info.bvh_nodes8 = d_nodes8_.ptr ? out.n_nodes8 : 0;
  dev.bvh_nodes48 = nullptr;
  dev.bvh_top48 = nullptr;
#ifdef GLZ_NODE48
  if (!hip_ok(d_nodes48_.alloc(out.n_nodes), "alloc 48-byte nodes", err) || !hip_ok(d_top48_.alloc(kBvhTopNodes), "alloc 48-byte top table", err) ||
      !hip_ok(launch_compress_nodes(st, d_nodes_.ptr, out.n_nodes, d_nodes48_.ptr), "k_compress_nodes", err) ||
      !hip_ok(launch_compress_nodes(st, d_top_.ptr, kBvhTopNodes, d_top48_.ptr), "k_compress_nodes (top)", err))
    return false;
  dev.bvh_nodes48 = d_nodes48_.ptr;
  dev.bvh_top48 = d_top48_.ptr;
#endif
  dev.bvh_grid = out.grid;
  for (int k = 0; k < 3; ++k) {
    info.bvh_grid_lo[k] = out.grid.lo[k];
    info.bvh_grid_cell[k] = out.grid.cell[k];
  }
  dev.bvh_tris = d_tris_.ptr;
  dev.bvh_quads = d_quads_.ptr;
  // per-leaf shading records (the identity flags let k_shade skip the object->world transform exactly)
  std::vector<uint32_t> ident(data.transforms.size(), 0u);
  for (size_t i = 0; i < ident.size(); ++i) {
    static const float id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    ident[i] = memcmp(data.transforms[i].m, id, 64) == 0 ? 1u : 0u;   // bitwise: -0.0 does not count
  }
  if (!hip_ok(d_xf_identity_.upload(ident.data(), ident.size(), st), "upload transform flags", err)) return false;
  if (!hip_ok(d_shade_tris_.alloc((size_t)n * 8), "alloc shading records", err)) return false;
  if (!hip_ok(launch_shade_records(st, n, d_tris_.ptr, d_instances_.ptr, d_indices_.ptr, d_vertices_.ptr, d_derivatives_.ptr, d_xf_identity_.ptr,
                                   d_shade_tris_.ptr), "k_shade_records", err))
    return false;
  if (!hip_ok(hipStreamSynchronize(st), "shading records", err)) return false;
  dev.shade_tris = d_shade_tris_.ptr;
  dev.xf_identity = d_xf_identity_.ptr;
  dev.n_world_tris = n;
  n_shade_slots_ = n;
  if (!build_alpha_records(err)) return false;
  info.as_levels = 1;
  info.n_as_triangles = n;
  info.as_bytes = (uint64_t)out.n_nodes * sizeof(BvhNode4) + (uint64_t)(n + 1) * sizeof(BvhTri) + (uint64_t)n * 128u + (uint64_t)out.n_leaves * sizeof(BvhQuad) +
                  (uint64_t)info.bvh_nodes8 * sizeof(BvhNode8);
  return true;
}

// What the alpha test of a candidate reads (DeviceScene::alpha_recs): texture coordinates and the opacity map's descriptor per triangle
// slot.  Flattened scenes only (the two-level tracer takes the material from the instance, alpha_test_instance); nothing without an
// opacity map.  Called by the flattened build and again by everything that changes a material's opacity map or moves texels.
bool Scene::build_alpha_records(Error& err) {
  dev.alpha_recs = nullptr;
  if (dev.two_level || !dev.has_non_opaque || n_shade_slots_ == 0 || !d_shade_tris_.ptr) return true;
  hipStream_t st = instance->stream;
  if (!hip_ok(d_alpha_recs_.alloc((size_t)n_shade_slots_ * 3), "alloc alpha records", err)) return false;
  if (!hip_ok(launch_alpha_records(st, n_shade_slots_, d_shade_tris_.ptr, dev.materials, dev.tex_desc, d_alpha_recs_.ptr), "k_alpha_records", err)) return false;
  if (!hip_ok(hipStreamSynchronize(st), "alpha records", err)) return false;
  dev.alpha_recs = d_alpha_recs_.ptr;
  return true;
}

bool Scene::update_textures(const glz_texture* textures, uint32_t n, Error& err) {
  if (!hip_ok(hipSetDevice(instance->device), "hipSetDevice", err)) return false;
  if (n == 0 || !textures) {
    err.code = GLZ_E_ARG;
    err.msg = "the texture list must keep at least the default texture";
    return false;
  }
  std::vector<TextureData> fresh(n);
  for (uint32_t i = 0; i < n; ++i) {
    const glz_texture& t = textures[i];
    const size_t bytes = (size_t)t.width * t.height * (t.format == GLZ_TEX_GRAY ? 1 : 4);
    if (!t.pixels || !bytes || t.format < 1 || t.format > 3) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "texture has inconsistent dimensions";
      return false;
    }
    fresh[i].info = t;
    fresh[i].level0.assign(t.pixels, t.pixels + bytes);
    fresh[i].info.pixels = fresh[i].level0.data();
  }
  data.textures.swap(fresh);
  for (auto& t : data.textures) t.info.pixels = t.level0.data();
  if (!upload_textures(err)) return false;
  info.n_textures = n;
  sky_distribution_tex_ = 0xFFFFFFFFu;   // the sky map's texels may have changed: its distributions are rebuilt
  return true;
}

bool Scene::update_materials_and_lights(const glz_material* mats, uint32_t n_mats, const glz_light* lights, uint32_t n_lights,
                                        const glz_texture* textures, uint32_t n_textures, Error& err) {
  if (!hip_ok(hipSetDevice(instance->device), "hipSetDevice", err)) return false;
  if (n_mats != data.materials.size()) {
    err.code = GLZ_E_ARG;
    err.msg = "update_materials_and_lights: the material count must not change (meshes index materials)";
    return false;
  }
  const size_t nt = textures ? n_textures : data.textures.size();
  for (uint32_t i = 0; i < n_mats; ++i) {
    const glz_material& m = mats[i];
    if (m.diffuse >= nt || m.roughness >= nt || m.metalness >= nt || m.normal >= nt || m.opacity >= nt) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "material references a missing texture";
      return false;
    }
  }
  for (uint32_t i = 0; i < n_lights; ++i)
    if (lights[i].ltype == GLZ_LIGHT_SKY && lights[i].resource_id >= nt) {
      err.code = GLZ_E_INVALID_DATA;
      err.msg = "sky light references a missing texture";
      return false;
    }
  if (textures && !update_textures(textures, n_textures, err)) return false;
  bool opacity_changed = false;
  for (uint32_t i = 0; i < n_mats; ++i) opacity_changed |= (mats[i].opacity != 0) != (data.materials[i].opacity != 0);
  data.materials.assign(mats, mats + n_mats);
  data.lights.assign(lights, lights + n_lights);
  if (!build_materials(err)) return false;
  if (!build_lights_and_sky(err)) return false;
  if (opacity_changed && !build_bvh(err)) return false;   // the non-opaque flag lives in the leaf records
  if (!build_alpha_records(err)) return false;             // which opacity map a material names, where its texels lie
  info.n_lights = lights_no;
  info.n_rt_lights = (uint32_t)h_lights.size();
  return hip_ok(hipStreamSynchronize(instance->stream), "update_materials_and_lights", err);
}

// refresh_binded_textures (raytracer.rs:328-356): the texture array changed under the same materials and lights
bool Scene::refresh_textures(const glz_texture* textures, uint32_t n, Error& err) {
  const size_t nm = data.materials.size();
  std::vector<glz_material> mats = data.materials;
  std::vector<glz_light> lights = data.lights;
  return update_materials_and_lights(mats.data(), (uint32_t)nm, lights.data(), (uint32_t)lights.size(), textures, n, err);
}

}  // namespace glz
