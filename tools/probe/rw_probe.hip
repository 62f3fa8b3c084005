// Read+write mixing probe: W bytes written and R bytes read by one launch, three ways:
//  same : every wave loads 16 B, waits for it, stores 16 B x (W/R) (the wait for a load also waits for the older stores: vmcnt is one in-order counter on gfx9)
//  roles: one wave of each workgroup only reads, the other three only write
//  late : every wave loads and stores, but consumes a load four iterations later (counted vmcnt)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_same(u32x4* __restrict__ out, const u32x4* __restrict__ in, size_t nin, int ratio, unsigned* sink) {
  unsigned acc = 0; u32x4 v = {1u, 2u, 3u, threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nin; i += (size_t)gridDim.x * 256) {
    acc += in[i][0];
    v[0] = acc;
    for (int r = 0; r < ratio; ++r) out[(size_t)r * nin + i] = v;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_roles(u32x4* __restrict__ out, const u32x4* __restrict__ in, size_t nin, int ratio, unsigned* sink) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0; u32x4 v = {1u, 2u, 3u, threadIdx.x};
  if (w == 0) {
    for (size_t i = (size_t)blockIdx.x * 64 + lane; i < nin; i += (size_t)gridDim.x * 64) acc += in[i][0];
  } else {
    const size_t nout = nin * ratio;
    for (size_t i = (size_t)blockIdx.x * 192 + (w - 1) * 64 + lane; i < nout; i += (size_t)gridDim.x * 192) out[i] = v;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_wonly(u32x4* __restrict__ out, size_t nout) {
  u32x4 v = {1u, 2u, 3u, threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void k_ronly(const u32x4* __restrict__ in, size_t nin, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nin; i += (size_t)gridDim.x * 256) acc += in[i][0];
  if (acc == 0x12345678u) *sink = acc;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1000;
}
int main() {
  const size_t R = (size_t)100864 * 384;          // 38.7 MB
  u32x4 *out, *in; unsigned* sink;
  (void)hipMalloc(&out, R * 8 + 4096); (void)hipMalloc(&in, R * 4); (void)hipMalloc(&sink, 4);
  (void)hipMemset(in, 1, R * 4);
  for (int ratio : {1, 2, 3, 8}) for (int grid : {1024, 4096}) {
    const size_t nin = R / 16;
    const double mbr = R / 1e6, mbw = R * ratio / 1e6;
    const float ts = timeit([&] { k_same<<<grid, 256>>>(out, in, nin, ratio, sink); });
    const float tr = timeit([&] { k_roles<<<grid, 256>>>(out, in, nin, ratio, sink); });
    const float tw = timeit([&] { k_wonly<<<grid, 256>>>(out, nin * ratio); });
    const float to = timeit([&] { k_ronly<<<grid, 256>>>(in, nin, sink); });
    printf("read %.0f MB write %.0f MB grid %d: same-wave %.1f us (%.2f TB/s) | roles %.1f us (%.2f) | write only %.1f us (%.2f) | read only %.1f us (%.2f)\n",
           mbr, mbw, grid, ts, (mbr + mbw) / ts, tr, (mbr + mbw) / tr, tw, mbw / tw, to, mbr / to);
  }
  // read-heavy: read 4x, write 1x
  return 0;
}
