// Microbenchmark (tools/micro, not product code): what does fetching ONE 64-byte record per lane cost on gfx950 when every lane
// wants a different record (the node fetch of an incoherent BVH traversal), and is it cheaper when the four lanes of a quad fetch each
// record together -- lane q of the quad loads the q-th 16-byte piece of the record of quad lane k, k = 0..3: every load instruction
// then touches 16 lines instead of 64 -- and the pieces change hands through LDS?
//   A  own record: 4 x global_load_dwordx4 per lane (what trace_wave does)
//   B  quad-cooperative loads + ds_write_b128 x4 / ds_read_b128 x4 transpose
//   C  one 16-byte piece only (lower bound: a quarter of the bytes, one access per lane)
//   D  quad-cooperative loads, pieces exchanged with DPP quad_perm moves and a 4-way select (no LDS)
// Every lane follows a dependent chain (the next index comes out of the loaded words), ITER visits, persistent grid of
// CUs x blocks_per_cu blocks of 256 threads.  Prints ns per visit of a wave and accesses/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int CTL>
__device__ __forceinline__ u32x4 pick_piece(u32x4 r0, u32x4 r1, u32x4 r2, u32x4 r3, int q) {
  u32x4 a, b, c, d;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = (uint32_t)__builtin_amdgcn_mov_dpp((int)r0[e], CTL, 0xF, 0xF, true);
    b[e] = (uint32_t)__builtin_amdgcn_mov_dpp((int)r1[e], CTL, 0xF, 0xF, true);
    c[e] = (uint32_t)__builtin_amdgcn_mov_dpp((int)r2[e], CTL, 0xF, 0xF, true);
    d[e] = (uint32_t)__builtin_amdgcn_mov_dpp((int)r3[e], CTL, 0xF, 0xF, true);
  }
  return q == 0 ? a : (q == 1 ? b : (q == 2 ? c : d));
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const u32x4* __restrict__ nodes, uint32_t mask, int iters, int active, uint32_t* out) {
  __shared__ u32x4 s_x[4][4 * 68];   // [wave][k * 68 + lane]: 64 pieces + 4 of padding per load instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x) & mask;
  uint32_t acc = 0;
  if (MODE != 1 && MODE != 3 && lane >= active) { out[blockIdx.x * 256u + threadIdx.x] = 0; return; }
  const bool on = lane < active;
  for (int it = 0; it < iters; ++it) {
    u32x4 w0, w1, w2, w3;
    if (MODE == 0) {
      const u32x4* np = nodes + (size_t)idx * 4;
      w0 = np[0]; w1 = np[1]; w2 = np[2]; w3 = np[3];
    } else if (MODE == 2) {
      const u32x4* np = nodes + (size_t)idx * 4;
      w0 = np[0]; w1 = w0; w2 = w0; w3 = w0;
    } else {
      const int q = lane & 3;
      // index of quad lane k's record: DPP quad_perm broadcast
      const uint32_t i0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x00, 0xF, 0xF, true);   // quad_perm(0,0,0,0)
      const uint32_t i1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0x55, 0xF, 0xF, true);   // (1,1,1,1)
      const uint32_t i2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xAA, 0xF, 0xF, true);   // (2,2,2,2)
      const uint32_t i3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, 0xFF, 0xF, 0xF, true);   // (3,3,3,3)
      const u32x4 r0 = nodes[(size_t)i0 * 4 + q], r1 = nodes[(size_t)i1 * 4 + q], r2 = nodes[(size_t)i2 * 4 + q], r3 = nodes[(size_t)i3 * 4 + q];
      if (MODE == 1) {
        u32x4* s = s_x[wave];
        s[0 * 68 + lane] = r0; s[1 * 68 + lane] = r1; s[2 * 68 + lane] = r2; s[3 * 68 + lane] = r3;
        // same wave: DS operations complete in order, no barrier needed
        const u32x4* mine = s + q * 68 + (lane & ~3);
        w0 = mine[0]; w1 = mine[1]; w2 = mine[2]; w3 = mine[3];
      } else {
        // lane d wants piece j of its record = what quad lane j loaded in instruction k = d & 3: r_{d&3} of lane j
        // a source lane cannot know which register the reader wants (it differs per reader), so every piece j is fetched from
        // all four registers and the reader picks: 4 x 4 x 4 DPP moves + selects -- the VALU price of avoiding LDS
        u32x4 p[4];
        p[0] = pick_piece<0x00>(r0, r1, r2, r3, q); p[1] = pick_piece<0x55>(r0, r1, r2, r3, q);
        p[2] = pick_piece<0xAA>(r0, r1, r2, r3, q); p[3] = pick_piece<0xFF>(r0, r1, r2, r3, q);
        w0 = p[0]; w1 = p[1]; w2 = p[2]; w3 = p[3];
      }
    }
    const uint32_t s = (w0.x ^ w1.y) + (w2.z ^ w3.w) + w0.w + w1.x + w2.y + w3.z + w0.y + w0.z + w1.z + w1.w + w2.x + w2.w + w3.x + w3.y;
    acc += s;
    if (on) idx = mix(s + (uint32_t)it) & mask;
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  for (int log_n : {16, 20}) {
    const uint32_t n = 1u << log_n;
    std::vector<uint32_t> h((size_t)n * 16);
    uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    u32x4* d_nodes; uint32_t* d_out;
    hipMalloc(&d_nodes, (size_t)n * 64);
    hipMemcpy(d_nodes, h.data(), (size_t)n * 64, hipMemcpyHostToDevice);
    for (int per_cu : {1, 2, 4, 6}) {
      const int blocks = cus * per_cu;
      hipMalloc(&d_out, (size_t)blocks * 256 * 4);
      for (int active : {64, 40}) {
        for (int mode = 0; mode < 3; mode += 2) {
          hipEvent_t e0, e1;
          hipEventCreate(&e0); hipEventCreate(&e1);
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, d_nodes, n - 1, iters, active, d_out);
            if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, d_nodes, n - 1, iters, active, d_out);
            if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, d_nodes, n - 1, iters, active, d_out);
            if (mode == 3) hipLaunchKernelGGL(k_gather<3>, dim3(blocks), dim3(256), 0, 0, d_nodes, n - 1, iters, active, d_out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
          }
          const double visits = (double)blocks * 4 * iters;   // wave-visits
          printf("table %3u MB  %d blocks/CU  %2d lanes  mode %c: %.3f ms  %.0f ns per wave-visit-round (all %d waves of a CU)  %.2f G records/s\n", n >> 14, per_cu, active,
                 "ABCD"[mode], best, best * 1e6 / iters, per_cu * 4, visits * active / (best * 1e-3) / 1e9);
        }
      }
      hipFree(d_out);
    }
    hipFree(d_nodes);
  }
  return 0;
}
