// Store-pattern probe: how fast can 256 CUs write an [M, N] bf16 matrix (row pitch N*2 bytes) with the store shapes the GEMM
// epilogues use, with and without a concurrent read stream?  hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// mode 0: flat grid-stride 16-B stores.  mode 1: a wave instruction = 8 rows x 128 B (one 64-column slice of 8 rows), a wave
// owns a 64-column slice of a 64-row tile, 4 waves = 256 columns, column groups across workgroups.  mode 2: a wave instruction =
// one row x 1024 B.  mode 3: a wave instruction = 2 rows x 512 B.  rd: bytes of a flat read stream per output row (0 = none).
template <int MODE>
__global__ __launch_bounds__(256) void k_store(char* __restrict__ out, const char* __restrict__ in, int M, int N, int rd, unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t pitch = (size_t)N * 2;
  u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  unsigned acc = 0;
  if (MODE == 0) {
    const size_t total = (size_t)M * pitch / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total; i += (size_t)gridDim.x * 256) {
      if (rd) { if (i * 16 < (size_t)M * rd) acc += reinterpret_cast<const u32x4*>(in)[i][0]; }
      reinterpret_cast<u32x4*>(out)[i] = v;
    }
  } else {
    const int cw = MODE == 1 ? 256 : MODE == 2 ? 512 : 256;      // columns per workgroup pass
    const int ngroups = N / cw;
    const int ntiles = (M / 64) * ngroups;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int grp = t % ngroups, m0 = (t / ngroups) * 64;
      if (rd && grp == 0) {                                       // the tile's input rows: 64 rows x rd bytes, flat
        const char* src = in + (size_t)m0 * rd;
        for (int o = tid * 16; o < 64 * rd; o += 256 * 16) acc += reinterpret_cast<const u32x4*>(src + o)[0][0];
      }
      if (MODE == 1) {
        char* base = out + (size_t)m0 * pitch + (size_t)(grp * 256 + w * 64) * 2;
#pragma unroll
        for (int r = 0; r < 64; r += 8) *reinterpret_cast<u32x4*>(base + (size_t)(r + (lane >> 3)) * pitch + (lane & 7) * 16) = v;
      } else if (MODE == 2) {
        char* base = out + (size_t)m0 * pitch + (size_t)(grp * 512) * 2;
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<u32x4*>(base + (size_t)(w * 16 + r) * pitch + lane * 16) = v;
      } else {
        char* base = out + (size_t)m0 * pitch + (size_t)(grp * 256) * 2;
#pragma unroll
        for (int r = 0; r < 16; r += 2) *reinterpret_cast<u32x4*>(base + (size_t)(w * 16 + r + (lane >> 5)) * pitch + (lane & 31) * 16) = v;
      }
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}
template <int MODE> float run(char* out, const char* in, int M, int N, int rd, int grid, unsigned* sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k_store<MODE><<<grid, 256>>>(out, in, M, N, rd, sink);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) k_store<MODE><<<grid, 256>>>(out, in, M, N, rd, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1000;
}
int main() {
  const int M = 100864;
  char *out, *in; unsigned* sink;
  (void)hipMalloc(&out, (size_t)M * 1536 * 2 + 4096); (void)hipMalloc(&in, (size_t)M * 1536 * 2); (void)hipMalloc(&sink, 4);
  (void)hipMemset(in, 1, (size_t)M * 1536 * 2);
  for (int N : {512, 1536}) for (int rd : {0, 384}) for (int grid : {512, 1024, 2048}) {
    const double mb = ((double)M * N * 2 + (double)M * rd) / 1e6;
    const float t0 = run<0>(out, in, M, N, rd, grid * 4, sink), t1 = run<1>(out, in, M, N, rd, grid, sink);
    const float t2 = run<2>(out, in, M, N, rd, grid, sink), t3 = run<3>(out, in, M, N, rd, grid, sink);
    printf("N %4d rd %3d grid %4d  %.0f MB: flat %.1f us %.2f TB/s | 8x128B %.1f us %.2f | 1x1024B %.1f us %.2f | 2x512B %.1f us %.2f\n",
           N, rd, grid, mb, t0, mb / t0 / 1e6, t1, mb / t1 / 1e6, t2, mb / t2 / 1e6, t3, mb / t3 / 1e6);
  }
  return 0;
}
