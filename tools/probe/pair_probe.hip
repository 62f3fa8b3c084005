// Micro-probe: one MFMA wave and one VALU (GELU-like) wave on the same SIMD (waves w and w+4 of a 512-thread workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// VMODE 0: partner idle; 1: plain fma chain work; 2: GELU-like (fma + exp2 + rcp); 3: only transcendental
__device__ __forceinline__ u32x4 ds_rd(unsigned addr) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); return v; }
template <int VMODE, int PRIO>
__global__ __launch_bounds__(512, 2) void k_pair(float* out, unsigned long long* cyc, int mfma_iters, int valu_iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 127);
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (lane & 15) * 416 + (lane >> 4) * 16;
  unsigned long long t0 = 0, t1 = 0;
  float res = 0.f;
  __syncthreads();
  if (w < 4) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 q = {0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, q), b = a;
    if (PRIO & 1) __builtin_amdgcn_s_setprio(3);
    t0 = __builtin_readcyclecounter();
    if (PRIO & 2) {
      u32x4 fa[8];
      for (int i = 0; i < 8; ++i) fa[i] = ds_rd(base + i * 64);
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if ((u & 3) == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[u]), "+v"(fa[u + 1]), "+v"(fa[u + 2]), "+v"(fa[u + 3]));
          const bf16x8 af = __builtin_bit_cast(bf16x8, fa[u]);
          acc[u] = mma(af, b, acc[u]);
          acc[(u + 4) & 7] = mma(af, a, acc[(u + 4) & 7]);
          fa[(u + 7) & 7] = ds_rd(base + u * 64 + (it & 1) * 6656);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int i = 0; i < 8; ++i) res += __builtin_bit_cast(f32x4, fa[i])[0];
    } else {
      for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 7] = mma(a, b, acc[u & 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][3];
  } else if (VMODE > 0) {
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = 0.01f * (lane + i) - 0.3f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float v = x[i];
        if (VMODE == 1) { float q = __builtin_fmaf(1.1e-3f, v, -0.107f); q = __builtin_fmaf(q, v, -2.3f); q = __builtin_fmaf(q, v, 0.5f); q = __builtin_fmaf(q, v, 0.25f); q = __builtin_fmaf(q, q, v); q = __builtin_fmaf(q, v, 0.1f); v = __builtin_fmaf(q, 0.001f, v); }
        if (VMODE == 2) { const float xc = fminf(v * v, 47.7f); float q = __builtin_fmaf(1.126e-3f, xc, -0.1075f); q = __builtin_fmaf(q, xc, -2.3f); const float e = __builtin_amdgcn_exp2f(v * q); v = v * __builtin_amdgcn_rcpf(1.0f + e) + 0.01f; }
        if (VMODE == 3) { v = __builtin_amdgcn_exp2f(v); v = __builtin_amdgcn_rcpf(v + 1.5f); }
        x[i] = v;
      }
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 32; ++i) res += x[i];
  }
  out[threadIdx.x + blockIdx.x * blockDim.x] = res;
  if (lane == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}
template <int VMODE, int PRIO> void run(const char* name, int mi, int vi) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  k_pair<VMODE, PRIO><<<256, 512>>>(out, cyc, mi, vi);
  k_pair<VMODE, PRIO><<<256, 512>>>(out, cyc, mi, vi);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-44s prio %d  MFMA wave: %6llu cycles for %d MFMAs (%.1f each) | VALU wave: %6llu cycles for %d GELU-units\n", name, PRIO, h[0], mi * 16, (double)h[0] / (mi * 16), h[4], vi * 32);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0, 0>("partner idle", 6, 0);
  run<1, 0>("partner: 7 fma per unit x 32", 6, 1);
  run<2, 0>("partner: GELU-like x 32", 6, 1);
  run<3, 0>("partner: exp2 + rcp x 32", 6, 1);
  run<2, 1>("partner: GELU-like x 32", 6, 1);
  run<2, 0>("partner: GELU-like x 96", 18, 3);
  run<2, 1>("partner: GELU-like x 96", 18, 3);
  run<1, 0>("partner: fma x 96", 18, 3);
  run<0, 0>("partner idle", 18, 0);
  printf("-- MFMA wave with fragment reads (1 per 2 MFMAs, counted waits)\n");
  run<0, 2>("partner idle", 6, 0);
  run<1, 2>("partner: fma x 32", 6, 1);
  run<2, 2>("partner: GELU-like x 32", 6, 1);
  run<2, 3>("partner: GELU-like x 32", 6, 1);
  run<2, 2>("partner: GELU-like x 96", 18, 3);
  run<3, 2>("partner: exp2 + rcp x 96", 18, 3);
  return 0;
}
