// Shared device/host helpers for the uvc_amd HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define UVC_OK 0
#define UVC_ERR_ARG 1
#define UVC_ERR_LAUNCH 2
#define UVC_ERR_UNSUPPORTED 3

#define UVC_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return uvc_set_error(e__, __FILE__, __LINE__); \
  } while (0)

int uvc_set_error(hipError_t e, const char* file, int line);
int uvc_set_error_msg(int code, const char* msg);

// hipFuncAttributeMaxDynamicSharedMemorySize applies per DEVICE: a kernel that needs more than 64 KB of dynamic LDS must get the
// attribute on every device the process launches it on.  One bit per device ordinal and call site; usage (inside a function that
// returns an int status):  UVC_MAX_LDS(bytes, kernel<template, arguments>);
// (at most 64 devices per process: ordinals beyond that set the attribute on every launch; the mask is atomic, so two host threads
//  launching the same kernel at once at worst both set the attribute)
static inline hipError_t uvc_max_lds_once(std::atomic<uint64_t>& mask, const void* func, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = dev < 64 ? 1ull << dev : 0;
  if (mask.load(std::memory_order_relaxed) & bit) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) mask.fetch_or(bit, std::memory_order_relaxed);
  return e;
}
#define UVC_MAX_LDS(bytes, ...)                                                              \
  do {                                                                                       \
    static std::atomic<uint64_t> lds_mask__{0};                                              \
    const hipError_t lds_e__ = uvc_max_lds_once(lds_mask__, (const void*)(__VA_ARGS__), (int)(bytes)); \
    if (lds_e__ != hipSuccess) return uvc_set_error(lds_e__, __FILE__, __LINE__);            \
  } while (0)

typedef uint16_t bf16_t;  // storage type: raw bfloat16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(T) __attribute__((address_space(3))) T*

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float32 -> bfloat16, round-to-nearest-even: the casts lower to v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// v + (lanes ^16, ^32, ^48): the sum over the four 16-lane rows of a wave, in every lane, by the gfx950 row-swap VALU instructions
// (v_permlane16_swap / v_permlane32_swap) instead of two ds_bpermute round trips through the LDS crossbar.  Same additions in the
// same order as  v += shfl_xor(v, 16); v += shfl_xor(v, 32)  -> same bits.
__device__ __forceinline__ float sum_rows4(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned c = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Sum over aligned groups of W lanes (16, 32 or 64), in every lane: the butterfly  v += shfl_xor(v, 1); v += shfl_xor(v, 2); ... v += shfl_xor(v, W / 2)
// with the SAME additions (same bits) and no LDS crossbar: quad swaps by DPP, then the mirrored half row / row -- after the steps before them every lane of a
// quad / of a half row holds the same partial sum, so the mirrored partner carries exactly what the xor-4 / xor-8 partner would -- then the row swaps.
// (__shfl_xor compiles to ds_bpermute_b32: a dependent chain of 4-6 LDS round trips per reduction; the soft-split kernels spent 40 % of their time there.)
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int W> __device__ __forceinline__ float xor_tree_sum(float v) {
  static_assert(W == 16 || W == 32 || W == 64, "group width");
  v = dpp_add<0xB1>(v);            // quad_perm [1, 0, 3, 2]   = xor 1
  v = dpp_add<0x4E>(v);            // quad_perm [2, 3, 0, 1]   = xor 2
  v = dpp_add<0x141>(v);           // row_half_mirror          = xor 4 (on quad-uniform values)
  v = dpp_add<0x140>(v);           // row_mirror               = xor 8 (on half-row-uniform values)
  if (W >= 32) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if (W == 64) {
    const unsigned c = __float_as_uint(v);
    const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
    v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
  }
  return v;
}

__device__ __forceinline__ float max_rows4(float v) {           // max over lanes l, l^16, l^32, l^48, in every lane
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned c = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  const auto q = __builtin_amdgcn_permlane32_swap(c, c, false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// exact-erf GELU and its derivative (nn.GELU default, model_distilled.py:108,118) -- float32 parity mode
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// bf16 mode: the result is rounded to 8 mantissa bits, so the normal CDF is evaluated as a logistic of an odd quintic,
//     Phi(x) ~= 1 / (1 + exp(-x (c0 + c1 x^2 + c2 x^4))),   c fitted minimax on |GELU error| / max(|GELU|, 2e-3)
// |x Phi~ - GELU(x)| <= 7.4e-4 * max(|GELU(x)|, 2e-3) (below 2^-10 relative wherever |GELU| >= 2e-3, i.e. x >= -3.3; <= 1.5e-6
// absolute in the tail beyond) and <= 6.6e-5 absolute everywhere; GELU'(x) = Phi~ + x phi(x) with the exact density: <= 8.3e-5 absolute.
// 9 VALU instructions (two of them v_exp_f32 / v_rcp_f32) against 15 for the Abramowitz-Stegun erf this replaces (19 -> 13 with
// the derivative): the fc1 epilogue and the fused teacher MLP are VALU-bound on exactly this.  x^2 is clamped at the
// quintic's maximum (47.7, |x| = 6.9, where x p(x) = 23: Phi~ = 1 - 8e-11) so the logistic saturates instead of turning over.
// The float32 parity mode keeps the exact erff (gelu_f above).
__device__ __forceinline__ float gelu_cdf_fast(float x, float x2) {
  const float xc = fminf(x2, 47.7f);
  // coefficients pre-multiplied by -log2(e): exp(-x p) = exp2(x q)
  float q = __builtin_fmaf(1.12616072e-3f, xc, -1.07522990e-1f);
  q = __builtin_fmaf(q, xc, -2.30050278f);
  const float e = __builtin_amdgcn_exp2f(x * q);
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_fast(float x) { return x * gelu_cdf_fast(x, x * x); }
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float x2 = x * x;
  const float c = gelu_cdf_fast(x, x2);
  const float e = __builtin_amdgcn_exp2f(x2 * -0.72134752f);           // exp(-x^2 / 2)
  return __builtin_fmaf(x * 0.39894228040143267794f, e, c);
}
template <typename T> struct Gelu;
template <> struct Gelu<float> {
  static __device__ __forceinline__ float f(float x) { return gelu_f(x); }
  static __device__ __forceinline__ float g(float x) { return gelu_grad_f(x); }
  static __device__ __forceinline__ void fg(float x, float& fo, float& go) { fo = gelu_f(x); go = gelu_grad_f(x); }
};
template <> struct Gelu<bf16_t> {
  static __device__ __forceinline__ float f(float x) { return gelu_fast(x); }
  static __device__ __forceinline__ float g(float x) { return gelu_grad_fast(x); }
  // value and derivative from one evaluation of the cdf / exponential (the forward stores both, so the backward's
  // epilogue is a single multiply)
  static __device__ __forceinline__ void fg(float x, float& fo, float& go) {
    const float x2 = x * x;
    const float c = gelu_cdf_fast(x, x2);
    const float e = __builtin_amdgcn_exp2f(x2 * -0.72134752f);
    fo = x * c;
    go = __builtin_fmaf(x * 0.39894228040143267794f, e, c);
  }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
