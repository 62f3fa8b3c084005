// HBM streaming ceilings on one MI355X for the access shapes the step's kernels use (16 B/lane, grid-stride):
// pure read, pure write, copy, and the 4-read/1-write mix of LayerNorm backward, at footprints inside and far outside
// the 256 MB MALL.   hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o gpurun_out/bw_probe && gpurun_out/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ a, size_t n, float* out) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_write(f32x4* __restrict__ a, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = v;
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_mix41(const f32x4* __restrict__ a, const f32x4* __restrict__ b, const f32x4* __restrict__ c,
                                               const f32x4* __restrict__ d, f32x4* __restrict__ o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i] + b[i] * c[i] + d[i];
}
// same, but a contiguous chunk per block (the row-block shape of the real kernels) instead of a grid-stride
__global__ __launch_bounds__(256) void k_copy_chunk(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n, size_t per_block) {
  const size_t lo = blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) b[i] = a[i];
}

int main() {
  const size_t MAXB = (size_t)1 << 30;
  f32x4 *A, *B, *C, *D, *O; float* out;
  CK(hipMalloc(&A, MAXB)); CK(hipMalloc(&B, MAXB)); CK(hipMalloc(&C, MAXB)); CK(hipMalloc(&D, MAXB)); CK(hipMalloc(&O, MAXB)); CK(hipMalloc(&out, 4));
  CK(hipMemset(A, 0, MAXB)); CK(hipMemset(B, 0, MAXB)); CK(hipMemset(C, 0, MAXB)); CK(hipMemset(D, 0, MAXB)); CK(hipMemset(O, 0, MAXB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes[] = {(size_t)32 << 20, (size_t)77 << 20, (size_t)256 << 20, (size_t)1 << 30};
  const int grids[] = {1024, 2048, 4096, 16384};
  printf("kernel,bytes_per_array_MB,grid,us,GBps_total\n");
  for (size_t sz : sizes) for (int g : grids) {
    const size_t n = sz / 16;
    auto timeit = [&](const char* name, double streams, auto launch) {
      for (int i = 0; i < 3; ++i) launch();
      CK(hipEventRecord(e0, 0));
      const int it = 10;
      for (int i = 0; i < it; ++i) launch();
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s,%zu,%d,%.1f,%.0f\n", name, sz >> 20, g, ms * 1e3 / it, streams * sz / (ms / it * 1e-3) / 1e9);
    };
    timeit("read", 1, [&] { k_read<<<g, 256>>>(A, n, out); });
    timeit("write", 1, [&] { k_write<<<g, 256>>>(O, n); });
    timeit("copy", 2, [&] { k_copy<<<g, 256>>>(A, O, n); });
    timeit("mix4r1w", 5, [&] { k_mix41<<<g, 256>>>(A, B, C, D, O, n); });
    timeit("copy_chunk", 2, [&] { k_copy_chunk<<<g, 256>>>(A, O, n, (n + g - 1) / g); });
  }
  return 0;
}
