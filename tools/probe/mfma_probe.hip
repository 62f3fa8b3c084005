// Micro-probe: cycles per v_mfma_f32_16x16x32_bf16 on one SIMD under the instruction mixes of the fused MLP's matrix interval.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ u32x4 ds_rd(unsigned addr) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); return v; }

// MODE 0: 8 independent accumulators, fixed A/B, nothing else
// MODE 1: + one ds_read_b128 per two MFMAs into a register nobody uses soon (ring of 3 groups, forced distinct)
// MODE 2: ds_read overwrites the A operand of the two MFMAs just issued (what hipcc's allocation gives the fused MLP)
// MODE 3: as 2 but the read is issued two MFMAs later (after the NEXT pair)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_probe(float* out, unsigned long long* cyc, int iters, int nwaves_active) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
  __syncthreads();
  if (w >= nwaves_active) return;
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (lane & 15) * 416 + (lane >> 4) * 16;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 fa[8];
  for (int i = 0; i < 8; ++i) fa[i] = ds_rd(base + i * 64);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  u32x4 bq = ds_rd(base + 1024), bq2 = ds_rd(base + 2048);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq), "+v"(bq2));
  const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq), b1 = __builtin_bit_cast(bf16x8, bq2);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {          // 8 units x 2 MFMAs
      if (MODE >= 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[u]));
      const bf16x8 a = __builtin_bit_cast(bf16x8, fa[u]);
      acc[u] = mma(a, b0, acc[u]);
      acc[(u + 4) & 7] = mma(a, b1, acc[(u + 4) & 7]);
      if (MODE == 1) { asm volatile("" :: "v"(fa[u])); }
      if (MODE == 2) fa[u] = ds_rd(base + u * 64 + (it & 1) * 6656);
      if (MODE == 1) fa[(u + 5) & 7] = ds_rd(base + u * 64 + (it & 1) * 6656);
      if (MODE == 3) fa[(u + 7) & 7] = ds_rd(base + u * 64 + (it & 1) * 6656);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  for (int i = 0; i < 8; ++i) s[0] += __builtin_bit_cast(f32x4, fa[i])[0];
  out[threadIdx.x + blockIdx.x * blockDim.x] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int MODE> void run(const char* name, int nw) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  const int iters = 200;
  k_probe<MODE><<<256, 512>>>(out, cyc, iters, nw);
  k_probe<MODE><<<256, 512>>>(out, cyc, iters, nw);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-70s waves/WG %d: %.1f cycles per MFMA (wave 0), %.1f (wave %d)\n", name, nw, (double)h[0] / (iters * 16), (double)h[nw - 1] / (iters * 16), nw - 1);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int nw : {4, 8}) {
    run<0>("0: MFMAs only, 8 accumulators", nw);
    run<1>("1: + ds_read_b128 per 2 MFMAs into an unrelated register", nw);
    run<2>("2: ds_read overwrites the A operand of the 2 MFMAs just issued", nw);
    run<3>("3: ds_read overwrites the A operand of the PREVIOUS pair", nw);
  }
  return 0;
}
