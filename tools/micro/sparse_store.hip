// Microbenchmark (tools/micro, not product code): what does the memory system write for 16-byte stores that cover only part of a line?
// k_trace's ShadowSink / fresh camera rays store one float4 per pixel for a subset of the pixels; WRITE_SIZE (rocprofv3 --pmc) per launch says how
// much of the neighbourhood goes out with them.  N float4 elements; a thread stores element i when keep(i):
//   MODE 0  every element                      (16 of every 16 bytes)
//   MODE 1  every 2nd element                  (16 of every 32)
//   MODE 2  every 4th element                  (16 of every 64)
//   MODE 3  every 8th element                  (16 of every 128)
//   MODE 4  a hashed 46 % of the elements      (the shadow rays' share of the pixels)
//   MODE 5  a hashed 24 % of the elements      (the fresh camera rays' share)
//   MODE 6  a hashed 46 % of 32-byte records   (two float4 next to each other: accumulator + result interleaved)
// Run under:  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -- ./sparse_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }
template <int MODE>
__global__ void __launch_bounds__(256) k_store(float4* __restrict__ dst, uint32_t n, unsigned long long* stored) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  bool keep = true;
  if (MODE == 1) keep = (i & 1u) == 0u;
  if (MODE == 2) keep = (i & 3u) == 0u;
  if (MODE == 3) keep = (i & 7u) == 0u;
  if (MODE == 4) keep = (mix(i) >> 8) < (uint32_t)(0.459 * 16777216.0);
  if (MODE == 5) keep = (mix(i) >> 8) < (uint32_t)(0.241 * 16777216.0);
  if (MODE == 6) keep = (mix(i >> 1) >> 8) < (uint32_t)(0.459 * 16777216.0);
  if (keep) dst[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
  const unsigned long long m = __ballot(keep);
  if ((threadIdx.x & 63u) == 0u) atomicAdd(stored, (unsigned long long)__popcll(m) * 16ull);
}
template <int MODE>
static void run(float4* d, uint32_t n, unsigned long long* cnt) {
  (void)hipMemset(cnt, 0, 8);
  for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(k_store<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, d, n, cnt);
  (void)hipDeviceSynchronize();
  unsigned long long h = 0;
  (void)hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost);
  printf("mode %d: %.1f MB stored per launch\n", MODE, (double)h / 4.0 / 1e6);
}
int main() {
  const uint32_t n = 16u << 20;   // 256 MB of float4: well past the L2s and the Infinity Cache
  float4* d = nullptr;
  unsigned long long* cnt = nullptr;
  if (hipMalloc(&d, (size_t)n * 16) != hipSuccess || hipMalloc(&cnt, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(d, 0, (size_t)n * 16);
  run<0>(d, n, cnt); run<1>(d, n, cnt); run<2>(d, n, cnt); run<3>(d, n, cnt); run<4>(d, n, cnt); run<5>(d, n, cnt); run<6>(d, n, cnt);
  return 0;
}
