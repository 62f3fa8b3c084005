// HBM bandwidth by read : write mix, on buffers far beyond the 256-MB Infinity Cache: every thread streams 16-byte pieces of RD
// read streams and WR write streams (separate 1-GiB / RD, WR regions), grid-stride, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int RD, int WR>
__global__ __launch_bounds__(256) void k_mix(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, unsigned* sink) {
  unsigned acc = 0; u32x4 v = {1u, 2u, 3u, threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
#pragma unroll
    for (int r = 0; r < RD; ++r) acc += in[(size_t)r * n + i][0];
    v[0] = RD ? acc : v[0];
#pragma unroll
    for (int w = 0; w < WR; ++w) out[(size_t)w * n + i] = v;
  }
  if (acc == 0x12345678u) *sink = acc;
}
template <int RD, int WR> void run(const u32x4* in, u32x4* out, size_t n, unsigned* sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {2048, 8192}) {
    for (int i = 0; i < 2; ++i) k_mix<RD, WR><<<grid, 256>>>(in, out, n, sink);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k_mix<RD, WR><<<grid, 256>>>(in, out, n, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gb = (double)(RD + WR) * n * 16 / 1e9;
    printf("read x%d write x%d (%.2f GB per launch) grid %d: %.1f us  %.2f TB/s\n", RD, WR, gb, grid, ms / 5 * 1000, gb / (ms / 5) );
  }
}
int main() {
  const size_t n = (size_t)256 << 20 >> 4;            // 256 MiB per stream
  u32x4 *in, *out; unsigned* sink;
  (void)hipMalloc(&in, n * 16 * 5); (void)hipMalloc(&out, n * 16 * 5); (void)hipMalloc(&sink, 4);
  (void)hipMemset(in, 1, n * 16 * 5);
  run<4, 0>(in, out, n, sink); run<0, 4>(in, out, n, sink); run<4, 1>(in, out, n, sink); run<2, 2>(in, out, n, sink); run<1, 4>(in, out, n, sink); run<1, 1>(in, out, n, sink);
  return 0;
}
