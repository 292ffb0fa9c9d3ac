// Wavefront path-tracing kernels for gfx950 (wave64).  One reference "launch" (one
// vkCmdTraceRaysKHR of path_trace.rgen = one path segment per pixel, raytracer.rs:553-562) is
// two kernels over the SoA path state in HBM:
//
//   k_trace   (persistent) ray generation / resume + closest-hit BVH4 traversal -> hit record; then the any-hit
//             traversal of the shadow rays the previous launch queued + update_count / update_result
//   k_shade   hit attributes, light sample, BSDF eval, Russian roulette, BSDF sample, state update
//             -> next ray + queued shadow ray with its contribution
//
// A wave owns one 8x8 pixel block of a 64x64 tile, so primary rays of a wave are coherent and all
// per-pixel arrays are read and written as one contiguous 1 KiB (float4) line per wave.
#include "device/wavefront.h"

namespace glz {
using namespace dev;

// ---------------------------------------------------------------------------------------------
// k_shade: path_trace.rgen:170-237 minus the two traceRayEXT calls, raytrace_hit.rchit:30-71
// ---------------------------------------------------------------------------------------------
#ifndef GLZ_SKY_LDS_FLOATS
#define GLZ_SKY_LDS_FLOATS 1088
#endif
constexpr uint32_t kSkyLdsFloats = GLZ_SKY_LDS_FLOATS;   // (skies of up to 1 087 rows; a taller one is searched in memory)
constexpr uint32_t kShadeLdsTableBytes = 8192;   // k_shade's LDS copy of the material / light / texture-descriptor tables (38 materials; larger tables are read from memory)   // the sky's marginal cdf (H + 1 floats) is staged in LDS when it fits (the values and row integrals next to it are read once per sample, from memory: staging all 3 H + 1 floats was 12 of the 19 KB a block copies before it starts)

// (GLZ_SHADE_WAVES = 4, device/tuning.h: 128 VGPRs, 4 of them spilled, since the light's spectrum is made after the BSDF evaluation and the
// importance is read where it is used (natural demand 152 -> 132): 0.357 -> 0.348 ms; at three waves the same code takes 0.363 ms.)
// COUNT: the instrumented build of the counting passes (texture fetches, light samples: DeviceScene::tex_counter).  The measured kernel
// sets the counter pointer to a constant null, so the checks in the texture and light code fold away (left as a run-time null they
// cost 2.5 %: a branch per fetch and two more live SGPRs).
// LOD: the build with the texture level of detail (shade_pixel<LOD>).
constexpr uint32_t kShadeBlock = 256, kShadeWavesPerBlock = kShadeBlock / 64;   // the regrouping domain: pixels sorted by code path per block (512 -> 0.354 against 0.352 ms, 1 024 -> 0.368: purer waves do not pay, the kernel waits for memory)
#ifdef GLZ_SECTION_TIMES
static __device__ unsigned long long g_shade_sections[16 * 4096];   // per wave (the first 4 096 of the grid): clocks {prologue, key, sort, [shade_pixel's six], epilogue}, [15] = launches
#endif
template <bool COUNT, bool LOD>
__global__ void __launch_bounds__(kShadeBlock, GLZ_SHADE_WAVES) k_shade(const LaunchArgs A) {
#ifdef GLZ_SECTION_TIMES
  unsigned long long ks[4] = {0, 0, 0, 0}, ks_last = __builtin_amdgcn_s_memtime();
#define GLZ_KS(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ks[k] += now_ - ks_last; ks_last = now_; } while (0)
#else
#define GLZ_KS(k) do { } while (0)
#endif
  // The kernel is bound by the memory system's random-access rate (16 extra scattered loads per pixel cost +27 %, 200 extra
  // VALU instructions nothing; 1 / 2 / 3 / 4 waves per SIMD take 0.76 / 0.45 / 0.36 / 0.35 ms), so the two small tables every texture fetch /
  // sky sample walks are staged in LDS once per block: the sRGB decode LUT (12 lookups per bilinear fetch) and the sky marginal CDF (an 11-step dependent search).
  __shared__ float s_lut[256];
  // One pool: [the importance the block's 256 pixels arrived with, 4 x 4 KB, read in whole lines by the prologue | the sky's marginal cdf |
  // the scene tables] while the block shades; after its last barrier the first 24 KB are the staging area of the state the pixels leave
  // (StagedState): 6 x 16 bytes per pixel, written where the pixel lives instead of where its thread sat.
  __shared__ uint4 s_pool[kShadeBlock * 4 + kSkyLdsFloats / 4 + kShadeLdsTableBytes / 16];
  uint4* s_imp = s_pool;
  float* s_sky = reinterpret_cast<float*>(s_pool + kShadeBlock * 4);
  uint4* s_tables = s_pool + kShadeBlock * 4 + kSkyLdsFloats / 4;
#ifndef GLZ_SHADE_SLOT_TIMING
  static_assert(6u * kShadeBlock <= kShadeBlock * 4 + kSkyLdsFloats / 4 + kShadeLdsTableBytes / 16, "the staging area of the path state fits the pool");
#endif
  // The small scene tables every hit walks through one after the other -- shading record -> RTMaterial -> texture descriptor
  // -> texels, light pick -> RTLight -- are staged in LDS when they fit: each lookup that stays on chip takes a dependent
  // memory round trip (1-2 us under load, the kernel's bound) off the hit's critical path.
  __shared__ uint32_t s_bin[kShadeWavesPerBlock * 64];   // [wave][key] counts, then start offsets
  __shared__ uint16_t s_perm[kShadeBlock];
  __shared__ float4 s_hit[kShadeBlock];   // hit records read by the regrouping prologue, handed to the thread that shades the pixel
  const uint32_t n_sky = A.scene.sky_header.marginal_cdf_count;
  const bool sky_in_lds = A.scene.sky_header.marginal_cdf_count > 1u && n_sky <= kSkyLdsFloats;
  if (threadIdx.x < 256u) s_lut[threadIdx.x] = A.scene.srgb_lut[threadIdx.x];
  if (sky_in_lds)
    for (uint32_t i = threadIdx.x; i < n_sky; i += kShadeBlock) s_sky[i] = A.scene.sky_marginal[i];
  s_bin[threadIdx.x] = 0;
  // [RTMaterial x n_materials | RTLight x n_rt_lights | TexDesc x n_textures] in 16-byte pieces
  const uint32_t qm = A.scene.n_materials * (uint32_t)(sizeof(RTMaterial) / 16), ql = A.scene.n_rt_lights * (uint32_t)(sizeof(RTLight) / 16),
                 qt = A.scene.n_textures * (uint32_t)(sizeof(TexDesc) / 16);
  const bool tables_in_lds = (qm + ql + qt) * 16u <= kShadeLdsTableBytes;   // uniform over the grid
  if (tables_in_lds) {
    const uint4* gm = reinterpret_cast<const uint4*>(A.scene.materials);
    const uint4* gl = reinterpret_cast<const uint4*>(A.scene.lights);
    const uint4* gt = reinterpret_cast<const uint4*>(A.scene.tex_desc);
    for (uint32_t i = threadIdx.x; i < qm + ql + qt; i += kShadeBlock) s_tables[i] = i < qm ? gm[i] : (i < qm + ql ? gl[i - qm] : gt[i - qm - ql]);
  }
  __syncthreads();
  GLZ_KS(0);   // tables staged
  DeviceScene S = A.scene;
  unsigned long long tex_tally[4] = {0ull, 0ull, 0ull, 0ull};
  S.tex_counter = COUNT ? tex_tally : nullptr;
  S.srgb_lut = s_lut;
  if (sky_in_lds) S.sky_cdf = s_sky;
  if (tables_in_lds) {
    S.materials = reinterpret_cast<const RTMaterial*>(s_tables);
    S.lights = reinterpret_cast<const RTLight*>(s_tables + qm);
    S.tex_desc = reinterpret_cast<const TexDesc*>(s_tables + qm + ql);
  }
  const FrameData& F = A.frame;
  // Block-local regrouping: the 256 pixels of the block are bucketed by the code path they are going to take -- miss,
  // or (BSDF kind, light kind of the NEE sample) -- and every thread then shades the pixel at its sorted position, so
  // a wave mostly runs one material and one light routine instead of all of them one after the other (lane utilisation
  // was 32 %, SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU, profiles/r01_pmc.json).  Pixels are independent and
  // all state is addressed by pixel, so the permutation changes no result; only the order of the shadow queue differs.
  // (The hit record requested before the tables are staged and the material id before the barrier they need -- three round trips in the
  // time of one and a half on paper: k_shade 0.326 against 0.325 ms, ten more registers spilt.  Not kept.)
  uint32_t key = 63u;   // pixels outside the image sort last
  {
    const uint32_t lid0 = blockIdx.x * kShadeBlock + threadIdx.x;
    const PixelId px0 = pixel_of(A.map, lid0);
    if (px0.active) {
      const float4 h0 = A.st.hit[lid0];
#pragma unroll
      for (int q = 0; q < 4; ++q) s_imp[q * kShadeBlock + threadIdx.x] = *reinterpret_cast<const uint4*>(&A.st.imp[q][lid0]);   // whole lines, pixel order (a fresh path's is never read)
      s_hit[threadIdx.x] = h0;   // the thread at this pixel's sorted slot reads it back after the barriers below: one dependent global load less
      const uint32_t leaf0 = __float_as_uint(h0.w);
      key = 0u;
      if (leaf0 != 0xFFFFFFFFu) {
        const RTMaterial* m0 = &S.materials[S.two_level ? S.instances[A.st.hit_inst[lid0]].material_id : __float_as_uint(S.shade_tris[8u * (size_t)leaf0 + 6u].w)];
        uint32_t light = 4u;
        if (m0->is_specular == 0 && F.lights_no != 0u) {
          uint32_t rng0 = pcg(__float_as_uint((float)F.seed) ^ pcg(__float_as_uint((float)px0.x) ^ pcg(__float_as_uint((float)px0.y))));
          const uint32_t li0 = (uint32_t)gl_min(rand01(rng0) * (float)F.lights_no, (float)(F.lights_no - 1u));
          light = S.lights[li0].shader;
        }
        key = 1u + (m0->bsdf_index < 6u ? m0->bsdf_index : 5u) * 5u + (light < 4u ? light : 4u);
      }
    }
  }
  GLZ_KS(1);   // hit record -> material -> code-path key
  // Counting sort without same-address atomics (256 atomicAdds on a handful of LDS words serialise: SQ_LDS_BANK_CONFLICT was
  // twice the LDS-active cycles of this kernel): every wave peels off its distinct keys with ballots -- a thread's rank among
  // the wave's threads with the same key is a popcount -- and leaves one count per (wave, key); 64 threads then turn the
  // counts into start offsets, key-major, wave-minor.
  uint32_t rank = 0;
  {
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    unsigned long long todo = __ballot(true);
    while (todo != 0ull) {
      const uint32_t k = (uint32_t)__shfl((int)key, __ffsll((long long)todo) - 1);
      const unsigned long long m = __ballot(key == k);
      if (key == k) {
        rank = (uint32_t)__popcll(m & ((1ull << ln) - 1ull));
        if (rank == 0u) s_bin[wv * 64u + k] = (uint32_t)__popcll(m);
      }
      todo &= ~m;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {   // exclusive prefix sum over the 64 keys of the per-key totals, then the per-wave starts inside a key
    uint32_t c[kShadeWavesPerBlock];
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t w = 0; w < kShadeWavesPerBlock; ++w) {
      c[w] = s_bin[w * 64u + threadIdx.x];
      cnt += c[w];
    }
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      if ((int)threadIdx.x >= off) incl += up;
    }
    uint32_t at = incl - cnt;
#pragma unroll
    for (uint32_t w = 0; w < kShadeWavesPerBlock; ++w) {
      s_bin[w * 64u + threadIdx.x] = at;
      at += c[w];
    }
  }
  __syncthreads();
  {
    uint32_t pos = s_bin[(threadIdx.x >> 6) * 64u + key] + rank;   // the pixel's place in the block's order by code path
#ifdef GLZ_SHADE_DEAL   // EXPERIMENT: the sorted sequence dealt to the four waves in chunks of GLZ_SHADE_DEAL pixels instead of 64 (balance against purity)
    constexpr uint32_t kChunk = GLZ_SHADE_DEAL, kPerWave = 64u / kChunk;
    const uint32_t chunk = pos / kChunk;
    pos = (chunk % kShadeWavesPerBlock) * 64u + (chunk / kShadeWavesPerBlock) * kChunk + pos % kChunk;
    static_assert(kPerWave * kChunk == 64u, "chunk divides a wave");
#endif
    s_perm[pos] = (uint16_t)threadIdx.x;
  }
  __syncthreads();
  GLZ_KS(2);   // regrouped
  const uint32_t lid = blockIdx.x * kShadeBlock + s_perm[threadIdx.x];
  // (Handing the next k_trace the pixels in this regrouped order -- one more word per pixel -- does nothing for the traversal, 0.588 ->
  // 0.591 ms; sorted per block by the octant of the new direction, camera rays last, 0.590 -> 0.573 ms, less than the sort and the
  // indirection cost.)
  const PixelId px = pixel_of(A.map, lid);
  StagedState staged;
  if (px.active) {
    const float4 ro = A.st.ray_o[lid], rd = A.st.ray_d[lid], hr = s_hit[s_perm[threadIdx.x]];
    // (see reread_kernarg: without it the regrouping prologue's view of the arguments stays in SGPRs through the shading code)
    const LaunchArgs& A2 = *(const LaunchArgs*)reread_kernarg();
    SharedQueue queue{A2};
    staged.A = &A2;
    staged.lds_imp = (LdsNodePtr)(reinterpret_cast<u32x4*>(s_imp) + s_perm[threadIdx.x]);
#ifdef GLZ_SECTION_TIMES
    staged.sec_last = __builtin_amdgcn_s_memtime();
#endif
    shade_pixel<LOD>(A2, S, A2.frame, lid, px, ro, rd, hr, queue, staged);
  }
#ifdef GLZ_SECTION_TIMES
  ks_last = __builtin_amdgcn_s_memtime();
#endif
#ifdef GLZ_SHADE_SLOT_TIMING   // TIMING ONLY (images are wrong): every thread stores the state it made at ITS OWN index -- whole lines without the meeting at the barrier, what a slot-indexed path state would do; the pixels of a block exchange paths, the workload stays what it is
  {
    const LaunchArgs& A3 = A;
    const uint32_t at = blockIdx.x * kShadeBlock + threadIdx.x;
    if (staged.mask & 1u) A3.st.ray_o[at] = staged.ro;
    if (staged.mask & 2u) A3.st.ray_d[at] = staged.rd;
    if (staged.mask & 4u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) A3.st.imp[q][at] = staged.im[q];
    }
  }
#else
  {
    // The regrouped threads would store 16-byte pieces scattered over the block's 4 KB of each state array (six arrays); the L2 has
    // to assemble the lines.  Thread i stores pixel i's state instead: the values travel through LDS, which nobody needs any more
    // once the whole block is here (a block's LDS and wave slots are handed on when its last wave ends either way).
    __syncthreads();
    float4* stage = reinterpret_cast<float4*>(s_pool);
    const uint32_t p = s_perm[threadIdx.x];   // the pixel (index in the block) this thread shaded
    s_bin[p] = staged.mask;
    if (staged.mask & 1u) stage[p] = staged.ro;
    if (staged.mask & 2u) stage[kShadeBlock + p] = staged.rd;
    if (staged.mask & 4u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) stage[(2u + q) * kShadeBlock + p] = staged.im[q];
    }
    __syncthreads();
    const LaunchArgs& A3 = *(const LaunchArgs*)reread_kernarg();
    const uint32_t m = s_bin[threadIdx.x], at = blockIdx.x * kShadeBlock + threadIdx.x;
    if (m & 1u) A3.st.ray_o[at] = stage[threadIdx.x];
    if (m & 2u) A3.st.ray_d[at] = stage[kShadeBlock + threadIdx.x];
    if (m & 4u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) A3.st.imp[q][at] = stage[(2u + q) * kShadeBlock + threadIdx.x];
    }
  }
#endif
  if (COUNT) flush_tex_tallies(A.counters->shade_tex, tex_tally);   // every lane of the wave is here (the counting build returns nowhere above)
#ifdef GLZ_SECTION_TIMES
  if (!COUNT) {
    GLZ_KS(3);   // the state leaves through LDS
    unsigned long long v[10] = {ks[0], ks[1], ks[2], staged.sec[0], staged.sec[1], staged.sec[2], staged.sec[3], staged.sec[4], staged.sec[5], ks[3]};
    const uint32_t w = blockIdx.x * kShadeWavesPerBlock + (threadIdx.x >> 6);
#pragma unroll
    for (int k = 0; k < 10; ++k) {   // the wave's value of a section: the largest any of its lanes saw
      unsigned int lo = (unsigned int)v[k], hi = (unsigned int)(v[k] >> 32);
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned int lo2 = __shfl_xor(lo, off), hi2 = __shfl_xor(hi, off);
        const unsigned long long a = ((unsigned long long)hi << 32) | lo, b = ((unsigned long long)hi2 << 32) | lo2;
        if (b > a) { lo = lo2; hi = hi2; }
      }
      if ((threadIdx.x & 63u) == 0u && w < 4096u) g_shade_sections[16u * w + k] += ((unsigned long long)hi << 32) | lo;   // (a wave's own words: no atomics)
    }
    if ((threadIdx.x & 63u) == 0u && w < 4096u) g_shade_sections[16u * w + 15u] += 1ull;
  }
#endif
}

template <bool COUNT, int ALPHA>
__global__ void __launch_bounds__(kBlock, GLZ_TRACE_WAVES) k_trace(const LaunchArgs A) {
  __shared__ int s_stack[kLdsStack * kBlock];
  __shared__ alignas(1024) int s_aux[kAuxPerBlock];
  __shared__ uint4 s_top[kLdsTop ? kBvhTopNodes * 4 : 1];
  static_assert(kBvhTopNodes * 4 <= kBlock, "stage_top copies one 16-byte piece per thread");
  GLZ_WAVE_STAMP(0);
  stage_top(A.scene, s_top);
  int* aux = wave_aux(s_aux, threadIdx.x >> 6);
  int* links = wave_links(s_aux, threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < kQueueShards) A.st.queue_count[A.shade_set * kQueueSetWords + threadIdx.x * kCounterStride] = 0;
  // counting build: the alpha tests' texture fetches are tallied through a copy of the scene that points at this thread's registers
  unsigned long long tex_tally[4] = {0ull, 0ull, 0ull, 0ull};
  DeviceScene counted_scene;
  if (COUNT) {
    counted_scene = A.scene;
    counted_scene.tex_counter = tex_tally;
  }
  const DeviceScene& TS = COUNT ? counted_scene : A.scene;
#if GLZ_TRACE_SPLIT
  // Waves that specialise (device/tuning.h GLZ_TRACE_SPLIT): with more 64-ray groups than waves, a share of the waves -- GLZ_TRACE_SPLIT_NUM /
  // _DEN of the shadow groups' share of all groups, spread evenly over the grid wave by wave -- takes only shadow groups and the others only
  // closest-hit groups: a wave then drains ONCE, at the end of the kernel, instead of once per kind, and a CU holds waves of both kinds at
  // all times.  With fewer groups than waves every group gets a wave of its own either way.  The counting kernels keep the two passes per wave.
  uint32_t split_closest = A.do_closest, split_shadow = A.do_shadow, split_wave_c = wave_index(), split_waves_c = wave_count(), split_wave_s = 0, split_waves_s = 0;
  bool split = false;
  if (!COUNT && A.do_closest && A.do_shadow) {
    const uint32_t* counts0 = A.st.queue_count + (A.shade_set ^ 1u) * kQueueSetWords;
    uint32_t n_sh = 0;
#pragma unroll
    for (uint32_t k = 0; k < kQueueShards; ++k) n_sh += counts0[k * kCounterStride];
    const uint32_t gc = (A.map.n_local_pixels + 63u) / 64u, gs = (n_sh + 63u) / 64u, nw = wave_count();
    if (gc + gs > nw && gs > 0u) {
      split = true;
      uint32_t ws = (uint32_t)(((unsigned long long)nw * gs * GLZ_TRACE_SPLIT_NUM) / ((unsigned long long)gc * GLZ_TRACE_SPLIT_DEN + (unsigned long long)gs * GLZ_TRACE_SPLIT_NUM));
      ws = ws < 1u ? 1u : (ws > nw - 1u ? nw - 1u : ws);
      const uint32_t w = wave_index();
      const uint32_t s_before = (uint32_t)(((unsigned long long)w * ws) / nw), s_after = (uint32_t)(((unsigned long long)(w + 1u) * ws) / nw);
      const bool is_shadow = s_after > s_before;   // the same in all lanes of a wave
      split_closest = is_shadow ? 0u : 1u;
      split_shadow = is_shadow ? 1u : 0u;
      split_wave_c = w - s_before;    // this wave's number among the closest-hit waves ...
      split_waves_c = nw - ws;
      split_wave_s = s_before;        // ... or among the shadow waves
      split_waves_s = ws;
    }
  }
  if (split_closest) {
    TraceTally tally;
    ClosestSource src{A, A.frame, tally, 0u};
    ClosestSink sink{A};
    trace_wave<false, COUNT, false, GLZ_TRACE_PREFETCH != 0, false, ALPHA>(TS, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, A.map.n_local_pixels, split_wave_c,
                                                         split_waves_c, tally);
    if (COUNT) flush_counters(A.counters, false, tally);
  }
  GLZ_WAVE_STAMP(1);
  if (split_shadow) {
    const uint32_t* counts = A.st.queue_count + (A.shade_set ^ 1u) * kQueueSetWords;
    uint32_t start[kQueueShards + 1];
    start[0] = 0;
#pragma unroll
    for (uint32_t k = 0; k < kQueueShards; ++k) start[k + 1] = start[k] + counts[k * kCounterStride];
    TraceTally tally;
    ShadowSource src{A, start, queue_capacity(A.map.n_local_pixels), 0u, make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    ShadowSink sink{A, src};
    uint32_t n_waves = wave_count();
    const uint32_t closest_groups = A.do_closest ? (A.map.n_local_pixels + 63u) / 64u : 0u;
    uint32_t wave = (wave_index() + n_waves - closest_groups % n_waves) % n_waves;   // (not specialised: the shadow groups dealt behind the closest-hit ones)
    if (split) { wave = split_wave_s; n_waves = split_waves_s; }
    trace_wave<true, COUNT, false, GLZ_TRACE_PREFETCH != 0, false, ALPHA>(TS, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, start[kQueueShards], wave, n_waves, tally);
    if (COUNT) flush_counters(A.counters, true, tally);
  }
#else
  if (A.do_closest) {
    TraceTally tally;
    ClosestSource src{A, A.frame, tally, 0u};
    ClosestSink sink{A};
    trace_wave<false, COUNT, false, GLZ_TRACE_PREFETCH != 0, false, ALPHA>(TS, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, A.map.n_local_pixels, wave_index(),
                                                         wave_count(), tally);
    if (COUNT) flush_counters(A.counters, false, tally);
  }
  GLZ_WAVE_STAMP(1);
  if (A.do_shadow) {
    // prefix sums of the eight shard counts (final: the k_shade that filled them has completed)
    const uint32_t* counts = A.st.queue_count + (A.shade_set ^ 1u) * kQueueSetWords;
    uint32_t start[kQueueShards + 1];
    start[0] = 0;
#pragma unroll
    for (uint32_t k = 0; k < kQueueShards; ++k) start[k + 1] = start[k] + counts[k * kCounterStride];
    TraceTally tally;
    ShadowSource src{A, start, queue_capacity(A.map.n_local_pixels), 0u, make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    ShadowSink sink{A, src};
    // The shadow groups are dealt out starting at the wave after the one that received the last closest-hit group: with
    // fewer groups than waves (a small tile share per GPU) every group of either kind gets a wave of its own, with more
    // the per-wave totals stay level.
    const uint32_t n_waves = wave_count();
    const uint32_t closest_groups = A.do_closest ? (A.map.n_local_pixels + 63u) / 64u : 0u;
    const uint32_t wave = (wave_index() + n_waves - closest_groups % n_waves) % n_waves;
    trace_wave<true, COUNT, false, GLZ_TRACE_PREFETCH != 0, false, ALPHA>(TS, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, start[kQueueShards], wave, n_waves, tally);
    if (COUNT) flush_counters(A.counters, true, tally);
  }
#endif
  if (COUNT) flush_tex_tallies(A.counters->trace_tex, tex_tally);
  GLZ_WAVE_STAMP(2);
}

// k_trace for a small tile share: the same two phases over the hierarchy's 8-wide nodes (trace_wave<WIDE8>).  Compiled
// for GLZ_TRACE8_WAVES waves per SIMD: a share of <= 262 144 pixels has a resident wave for every 64-ray group at four, and the node's 32
// words, the eight keys and (PREFETCH) the next node's 32 words want the registers.  No staged top, a deeper LDS stack.
__global__ void __launch_bounds__(kBlock, GLZ_TRACE8_WAVES) k_trace8(const LaunchArgs A) {
  __shared__ int s_stack[kLdsStack8 * kBlock];
  __shared__ alignas(1024) int s_aux[kAuxPerBlock];
  int* aux = wave_aux(s_aux, threadIdx.x >> 6);
  int* links = wave_links(s_aux, threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < kQueueShards) A.st.queue_count[A.shade_set * kQueueSetWords + threadIdx.x * kCounterStride] = 0;
  if (A.do_closest) {
    TraceTally tally;
    ClosestSource src{A, A.frame, tally, 0u};
    ClosestSink sink{A};
    trace_wave<false, false, false, GLZ_TRACE8_PREFETCH != 0, true>(A.scene, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr) nullptr, A.st.overflow, A.st.overflow_depth,
                                                                    A.map.n_local_pixels, wave_index(), wave_count(), tally);
  }
  if (A.do_shadow) {
    const uint32_t* counts = A.st.queue_count + (A.shade_set ^ 1u) * kQueueSetWords;
    uint32_t start[kQueueShards + 1];
    start[0] = 0;
#pragma unroll
    for (uint32_t k = 0; k < kQueueShards; ++k) start[k + 1] = start[k] + counts[k * kCounterStride];
    TraceTally tally;
    ShadowSource src{A, start, queue_capacity(A.map.n_local_pixels), 0u, make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    ShadowSink sink{A, src};
    const uint32_t n_waves = wave_count();
    const uint32_t closest_groups = A.do_closest ? (A.map.n_local_pixels + 63u) / 64u : 0u;
    const uint32_t wave = (wave_index() + n_waves - closest_groups % n_waves) % n_waves;
    trace_wave<true, false, false, GLZ_TRACE8_PREFETCH != 0, true>(A.scene, src, sink, &s_stack[threadIdx.x], aux, links, (LdsNodePtr) nullptr, A.st.overflow, A.st.overflow_depth,
                                                                   start[kQueueShards], wave, n_waves, tally);
  }
}

// the same kernel for two-level scenes (trace_wave_tl): closest hits additionally record the instance
struct ClosestSinkTl {
  const LaunchArgs& A;
  __device__ __forceinline__ void store(uint32_t lid, const HitRecord& h) {
    A.st.hit[lid] = make_float4(h.leaf == 0xFFFFFFFFu ? INFINITY : h.t, h.u, h.v, __uint_as_float(h.leaf));
    A.st.hit_inst[lid] = h.inst;
  }
};
// GLZ_TRACE_TL_WAVES = 4 (device/tuning.h): the instance entry and the on-the-fly world triangle need 153 VGPRs: at 6 waves per SIMD 201 of them live in scratch (forest x2000, tools/gpu_two_level_timing.py: 4.06 ms per launch), at 4 waves 34 (1.44 ms), at 3 none (1.45 ms)
template <bool COUNT>
__global__ void __launch_bounds__(kBlock, GLZ_TRACE_TL_WAVES) k_trace_tl(const LaunchArgs A) {
  __shared__ int s_stack[kLdsStack * kBlock];
  __shared__ alignas(1024) int s_aux[kAuxPerBlock];
  __shared__ float s_top_ray[9 * kBlock];   // per lane: the top level's grid-space ray (trace_wave_tl)
  __shared__ uint4 s_top[kTlLdsTop ? kBvhTopNodes * 4 : 1];
  if (kTlLdsTop) stage_top(A.scene, s_top);
  int* aux = wave_aux(s_aux, threadIdx.x >> 6);
  int* links = wave_links(s_aux, threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < kQueueShards) A.st.queue_count[A.shade_set * kQueueSetWords + threadIdx.x * kCounterStride] = 0;
  if (A.do_closest) {
    TraceTally tally;
    ClosestSource src{A, A.frame, tally, 0u};
    ClosestSinkTl sink{A};
    trace_wave_tl<false, COUNT>(A.scene, src, sink, &s_stack[threadIdx.x], aux, links, &s_top_ray[threadIdx.x], (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, A.map.n_local_pixels, wave_index(), wave_count(), tally);
    if (COUNT) flush_counters(A.counters, false, tally);
  }
  if (A.do_shadow) {
    const uint32_t* counts = A.st.queue_count + (A.shade_set ^ 1u) * kQueueSetWords;
    uint32_t start[kQueueShards + 1];
    start[0] = 0;
#pragma unroll
    for (uint32_t k = 0; k < kQueueShards; ++k) start[k + 1] = start[k] + counts[k * kCounterStride];
    ShadowSource src{A, start, queue_capacity(A.map.n_local_pixels), 0u, make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    ShadowSink sink{A, src};
    const uint32_t n_waves = wave_count();
    const uint32_t closest_groups = A.do_closest ? (A.map.n_local_pixels + 63u) / 64u : 0u;
    const uint32_t wave = (wave_index() + n_waves - closest_groups % n_waves) % n_waves;
    TraceTally tally;
    trace_wave_tl<true, COUNT>(A.scene, src, sink, &s_stack[threadIdx.x], aux, links, &s_top_ray[threadIdx.x], (LdsNodePtr)s_top, A.st.overflow, A.st.overflow_depth, start[kQueueShards], wave, n_waves, tally);
    if (COUNT) flush_counters(A.counters, true, tally);
  }
}

// ---------------------------------------------------------------------------------------------
// image plumbing
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_export(const TileMap map, const float4* __restrict__ tiled, float4* __restrict__ frame) {
  const uint32_t lid = blockIdx.x * kBlock + threadIdx.x;
  const PixelId px = pixel_of(map, lid);
  if (px.active) frame[(size_t)px.y * map.width + px.x] = tiled[lid];
}

// chain s of S holds the rank's local tiles j = jl * S + s at chain-local index jl: lay them out in the rank's own tile order
// (what a rank of a one-process-per-GPU job sends to rank 0, 1 / world of the frame's bytes)
__global__ void __launch_bounds__(kBlock) k_pack_tiles(uint32_t n_pixels, uint32_t S, uint32_t s, const float4* __restrict__ chain, float4* __restrict__ packed) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pixels) return;
  const uint32_t jl = i >> 12, k = i & 4095u;
  packed[(size_t)(jl * S + s) * 4096u + k] = chain[i];
}

// linear -> sRGB OETF, 8 bit, round to nearest (what the R8G8B8A8_SRGB blit does, raytracer.rs:576-584 [ext]).
// Byte work has to be bit-exact, and pow() is not the same function on any two machines, so the quantiser is stated
// without it: q = #{k in 1..255 : c >= T_k} with T_k = (float) EOTF((k - 0.5) / 255), the smallest float whose encoded
// value rounds to k (the OETF is monotonic, so this IS round(255 * OETF(c)) evaluated exactly).  The 255 thresholds are
// computed once on the host in double precision (Renderer::allocate) and searched here from LDS: 8 compares per channel.
__device__ __forceinline__ unsigned char srgb8(const float* __restrict__ thr, float c) {
  if (!(c > 0.0f)) return 0;   // NaN, zero and negatives
  uint32_t lo = 0, hi = 255;   // invariant: c >= T_lo (T_0 = 0), c < T_(hi+1) (T_256 = +inf)
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const uint32_t mid = (lo + hi + 1u) >> 1;
    const bool ge = c >= thr[mid];
    lo = ge ? mid : lo;
    hi = ge ? hi : mid - 1u;
  }
  return (unsigned char)lo;
}
__global__ void __launch_bounds__(kBlock) k_tonemap(uint32_t n, const float4* __restrict__ result, const float* __restrict__ thresholds, uchar4* __restrict__ out) {
  __shared__ float s_thr[256];
  s_thr[threadIdx.x] = thresholds[threadIdx.x];
  __syncthreads();
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float4 r = result[i];
  out[i] = make_uchar4(srgb8(s_thr, r.x), srgb8(s_thr, r.y), srgb8(s_thr, r.z), r.w >= 1.0f ? 255 : 0);
}
// ---------------------------------------------------------------------------------------------
// debug / parity kernels: arbitrary rays through the same traversal code
// ---------------------------------------------------------------------------------------------
struct DebugSource {
  const float* __restrict__ o3;
  const float* __restrict__ d3;
  const float* __restrict__ tmax_arr;   // nullptr = infinity
  float tmin_all;
  __device__ __forceinline__ bool load(uint32_t i, vec3& o, vec3& d, float& tmin, float& tmax) {
    o = mk3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]);
    d = mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
    tmin = tmin_all;
    tmax = tmax_arr ? tmax_arr[i] : INFINITY;
    return true;
  }
};
struct DebugClosestSink {
  const DeviceScene& S;
  float* t; uint32_t* tri; uint32_t* inst; float* u; float* v;
  __device__ __forceinline__ void store(uint32_t i, const HitRecord& h) {
    const bool hit = h.leaf != 0xFFFFFFFFu;
    t[i] = hit ? h.t : INFINITY;
    tri[i] = hit ? (S.two_level ? h.world_id : S.bvh_tris[h.leaf].world_id) : 0xFFFFFFFFu;
    inst[i] = hit ? (S.two_level ? h.inst : S.bvh_tris[h.leaf].instance) : 0xFFFFFFFFu;
    u[i] = hit ? h.u : 0.0f;
    v[i] = hit ? h.v : 0.0f;
  }
};
struct DebugAnySink {
  uint8_t* out;
  __device__ __forceinline__ void store(uint32_t i, const HitRecord& h) { out[i] = h.leaf != 0xFFFFFFFFu; }
};

__global__ void __launch_bounds__(kBlock) k_debug_closest(const DeviceScene S, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                                          float tmin, float* t, uint32_t* tri, uint32_t* inst, float* u, float* v,
                                                          uint32_t* overflow, uint32_t overflow_depth) {
  __shared__ int s_stack[kLdsStack * kBlock];
  __shared__ alignas(1024) int s_aux[kAuxPerBlock];
  __shared__ uint4 s_top[kLdsTop ? kBvhTopNodes * 4 : 1];
  __shared__ float s_top_ray[9 * kBlock];
  stage_top(S, s_top);
  TraceTally tally;
  DebugSource src{o, d, nullptr, tmin};
  DebugClosestSink sink{S, t, tri, inst, u, v};
  if (S.two_level) trace_wave_tl<false, false>(S, src, sink, &s_stack[threadIdx.x], wave_aux(s_aux, threadIdx.x >> 6), wave_links(s_aux, threadIdx.x >> 6), &s_top_ray[threadIdx.x], (LdsNodePtr)s_top, overflow, overflow_depth, n, wave_index(), wave_count(), tally);
  else trace_wave<false, false>(S, src, sink, &s_stack[threadIdx.x], wave_aux(s_aux, threadIdx.x >> 6), wave_links(s_aux, threadIdx.x >> 6), (LdsNodePtr)s_top, overflow, overflow_depth, n, wave_index(), wave_count(), tally);
}
__global__ void __launch_bounds__(kBlock) k_debug_any(const DeviceScene S, const float* __restrict__ o, const float* __restrict__ d,
                                                      const float* __restrict__ tmax, uint32_t n, float tmin, uint8_t* out, uint32_t* overflow,
                                                      uint32_t overflow_depth) {
  __shared__ int s_stack[kLdsStack * kBlock];
  __shared__ alignas(1024) int s_aux[kAuxPerBlock];
  __shared__ uint4 s_top[kLdsTop ? kBvhTopNodes * 4 : 1];
  __shared__ float s_top_ray[9 * kBlock];
  stage_top(S, s_top);
  TraceTally tally;
  DebugSource src{o, d, tmax, tmin};
  DebugAnySink sink{out};
  if (S.two_level) trace_wave_tl<true, false>(S, src, sink, &s_stack[threadIdx.x], wave_aux(s_aux, threadIdx.x >> 6), wave_links(s_aux, threadIdx.x >> 6), &s_top_ray[threadIdx.x], (LdsNodePtr)s_top, overflow, overflow_depth, n, wave_index(), wave_count(), tally);
  else trace_wave<true, false>(S, src, sink, &s_stack[threadIdx.x], wave_aux(s_aux, threadIdx.x >> 6), wave_links(s_aux, threadIdx.x >> 6), (LdsNodePtr)s_top, overflow, overflow_depth, n, wave_index(), wave_count(), tally);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n) { return dim3((n + kBlock - 1) / kBlock); }
// Persistent tracers: the grid is exactly what is resident at once (CUs x blocks per CU from the occupancy query, at
// most 8), and never more waves than there are 64-ray groups.  A block that had to wait for a slot would serialise
// behind a whole persistent block (cdna_hip_programming.md section 1: size persistent grids by residency).
template <class Kernel>
static dim3 persistent_grid(Kernel kernel, uint32_t n_rays) {
  int dev = 0, cus = 256, per_cu = 8;
  if (hipGetDevice(&dev) == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 4;
  per_cu = std::min(per_cu, 8);
  if (const char* cap = getenv("GLAZE_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(cap)));   // tuning: leave room for another chain's k_shade
  // (splitting the resident blocks between concurrent chains measured slower: a chain's blocks fill in as another's retire)
  const uint32_t resident = (uint32_t)cus * (uint32_t)per_cu;
  return dim3(std::max<uint32_t>(1u, std::min<uint32_t>((n_rays + kBlock - 1) / kBlock, resident)));
}

// Blocks of k_trace's persistent grid for a chain of n_local_pixels (the renderer asks once per allocation and passes the
// answer to every launch_trace; needs the device current).  Up to one closest-hit and one shadow ray per pixel: a small
// tile share still gets a wave per 64-ray group of either kind (fewer, longer-lived waves -- 2 to 4 groups per wave --
// measured 25-50 % slower for small shares: spread as wide as possible).
uint32_t trace_grid_blocks(uint32_t n_local_pixels, bool counting, bool two_level, bool wide8) {
  uint32_t rays = 2u * n_local_pixels;
  if (wide8 && !counting && !two_level) return persistent_grid(k_trace8, rays).x;
  if (two_level) return counting ? persistent_grid(k_trace_tl<true>, rays).x : persistent_grid(k_trace_tl<false>, rays).x;   // compiled for fewer waves per SIMD: its own residency
  return counting ? persistent_grid(k_trace<true, kAlphaInline>, rays).x : std::min(persistent_grid(k_trace<false, kAlphaNone>, rays).x, persistent_grid(k_trace<false, kAlphaPhase>, rays).x);
}

hipError_t launch_trace(hipStream_t st, const LaunchArgs& a, uint32_t blocks, bool wide8) {
  if (a.map.n_local_pixels == 0) return hipSuccess;
  if (wide8 && (a.scene.two_level || a.counters || !a.scene.bvh_nodes8)) return hipErrorInvalidValue;   // the 8-wide walk: flattened scenes, no work counters
  if (blocks == 0 || (uint64_t)blocks * kBlock > 2ull * a.map.n_local_pixels + kBlock) return hipErrorInvalidValue;   // the spill area holds one slot per lane of this bound
  if (a.scene.two_level && a.counters) hipLaunchKernelGGL(k_trace_tl<true>, dim3(blocks), dim3(kBlock), 0, st, a);   // node visits of both levels, triangle tests inside the instances
  else if (a.scene.two_level) hipLaunchKernelGGL(k_trace_tl<false>, dim3(blocks), dim3(kBlock), 0, st, a);
  else if (a.counters) hipLaunchKernelGGL((k_trace<true, kAlphaInline>), dim3(blocks), dim3(kBlock), 0, st, a);
  else if (wide8) hipLaunchKernelGGL(k_trace8, dim3(blocks), dim3(kBlock), 0, st, a);
#ifdef GLZ_TRACE_ALPHA_INLINE   // A/B builds (tools/build_variant.sh): the alpha test where the candidate is met, whatever the scene -- rounds 1-4's kernel
  else if (true) hipLaunchKernelGGL((k_trace<false, kAlphaInline>), dim3(blocks), dim3(kBlock), 0, st, a);
#endif
  else if (a.scene.has_non_opaque) hipLaunchKernelGGL((k_trace<false, kAlphaPhase>), dim3(blocks), dim3(kBlock), 0, st, a);   // candidates on non-opaque geometry wait for an alpha phase
  else hipLaunchKernelGGL((k_trace<false, kAlphaNone>), dim3(blocks), dim3(kBlock), 0, st, a);   // no opacity map in the scene: no alpha code in the kernel
  return hipGetLastError();
}
hipError_t launch_shade(hipStream_t st, const LaunchArgs& a) {
  if (a.map.n_local_pixels == 0) return hipSuccess;
  const dim3 grid((a.map.n_local_pixels + kShadeBlock - 1) / kShadeBlock), block(kShadeBlock);
  const bool lod = a.frame.lod_mode != 0u;
  if (a.counters && lod) hipLaunchKernelGGL((k_shade<true, true>), grid, block, 0, st, a);
  else if (a.counters) hipLaunchKernelGGL((k_shade<true, false>), grid, block, 0, st, a);
  else if (lod) hipLaunchKernelGGL((k_shade<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_shade<false, false>), grid, block, 0, st, a);
  return hipGetLastError();
}
hipError_t launch_export(hipStream_t st, const TileMap& map, const float4* tiled, float4* frame, bool zero_first) {
  if (zero_first) {
    hipError_t e = hipMemsetAsync(frame, 0, sizeof(float4) * (size_t)map.width * map.height, st);
    if (e != hipSuccess) return e;
  }
  if (map.n_local_pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(k_export, grid_for(map.n_local_pixels), dim3(kBlock), 0, st, map, tiled, frame);
  return hipGetLastError();
}
hipError_t launch_pack_tiles(hipStream_t st, uint32_t n_chain_pixels, uint32_t n_chains, uint32_t chain, const float4* tiled, float4* packed) {
  if (n_chain_pixels == 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_tiles, grid_for(n_chain_pixels), dim3(kBlock), 0, st, n_chain_pixels, n_chains, chain, tiled, packed);
  return hipGetLastError();
}
hipError_t launch_tonemap(hipStream_t st, uint32_t n, const float4* result_frame, const float* thresholds, uchar4* out) {
  static_assert(kBlock == 256, "k_tonemap stages the 256 thresholds with one load per thread");
  hipLaunchKernelGGL(k_tonemap, grid_for(n), dim3(kBlock), 0, st, n, result_frame, thresholds, out);
  return hipGetLastError();
}
hipError_t launch_debug_closest(hipStream_t st, const DeviceScene& scene, const float* o, const float* d, uint32_t n, float tmin, float* t,
                                uint32_t* tri, uint32_t* inst, float* u, float* v, uint32_t* overflow, uint32_t overflow_depth) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_debug_closest, persistent_grid(k_debug_closest, n), dim3(kBlock), 0, st, scene, o, d, n, tmin, t, tri, inst, u, v, overflow, overflow_depth);
  return hipGetLastError();
}
hipError_t launch_debug_any(hipStream_t st, const DeviceScene& scene, const float* o, const float* d, const float* tmax, uint32_t n, float tmin,
                            uint8_t* hit, uint32_t* overflow, uint32_t overflow_depth) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_debug_any, persistent_grid(k_debug_any, n), dim3(kBlock), 0, st, scene, o, d, tmax, n, tmin, hit, overflow, overflow_depth);
  return hipGetLastError();
}

}  // namespace glz

#ifdef GLZ_SECTION_TIMES
extern "C" int glz_debug_shade_sections(unsigned long long* out, int reset) {
  int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_shade_sections), sizeof(unsigned long long) * 16 * 4096);
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(glz::g_shade_sections)) == hipSuccess) e = (int)hipMemset(p, 0, sizeof(unsigned long long) * 16 * 4096);
  }
  return e;
}
// kernels_render.hip and kernels_path.hip each hold their own copy of g_sections (a __device__ array per translation unit): `which` 0 = k_trace's, see kernels_path.hip for k_path's
extern "C" int glz_debug_sections_trace(unsigned long long* out, int reset) {
  int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_sections), sizeof(unsigned long long) * 16 * 8192);
  if (reset) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(glz::g_sections)) == hipSuccess) e = (int)hipMemset(p, 0, sizeof(unsigned long long) * 16 * 8192);
  }
  return e;
}
#endif
#ifdef GLZ_WAVE_TIMES
extern "C" int glz_debug_wave_times(unsigned long long* out, int n_waves) {
  if (n_waves > 8192) n_waves = 8192;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_wave_times), sizeof(unsigned long long) * 3 * (size_t)n_waves);
}
extern "C" int glz_debug_tl_stats(unsigned long long* out, int reset) {
  int e = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_tl_stats), sizeof(unsigned long long) * 8);
  if (reset && e == 0) {
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = (int)hipMemcpyToSymbol(HIP_SYMBOL(glz::g_tl_stats), zero, sizeof(zero));
  }
  return e;
}
extern "C" int glz_debug_wave_stats(unsigned int* out, int n_waves) {
  if (n_waves > 8192) n_waves = 8192;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(glz::g_wave_stats), sizeof(unsigned int) * 8 * (size_t)n_waves);
}
#endif
