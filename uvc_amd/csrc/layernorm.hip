// LayerNorm forward/backward for gfx950 (HBM-bound; one 64-lane wave per token row, the row lives in
// registers, row reductions are wave shuffles).  Reference: nn.LayerNorm(D, eps=1e-6) in
// UVC/models/model_distilled.py:199,204,288 (eps from joint_train.py:138), biased variance.
// Algorithmic bytes/row: fwd 4D read + sizeof(T)*D write; bwd (4 + sizeof(Tdy))*D read (+4D per
// addend) + 4D write.
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "../../include/uvc_kernels.h"

namespace {

// backward: rows per workgroup (three workgroups fit a CU): enough workgroups to fill the chip several times over without
// inflating the partial reduction.
static int ln_rows_per_block(int rows) {
  return rows >= 48 * 1024 ? 64 : 32;     // stand-alone at 100 k rows: 56 / 67 us at 64 against 60 / 77 at 128 and 62 / 72 at 32; T2T-14 (25 k rows) +4 % step rate at 32
}

__device__ __forceinline__ size_t row_off(int r, int rpg, int64_t gs, int D) {
  return (size_t)(r / rpg) * gs + (size_t)(r % rpg) * D;
}

template <typename TX, typename TY, int NV>
__global__ __launch_bounds__(256) void k_ln_fwd(uvc_ln_args a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + w;
  if (r >= a.rows) return;
  const TX* x = reinterpret_cast<const TX*>(a.x) + row_off(r, a.rows_per_group, a.group_stride, a.D);
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < a.D ? ElemIO<TX>::load(x + c) : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)a.D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    const float d = c < a.D ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)a.D + a.eps);
  TY* y = reinterpret_cast<TY*>(a.y) + (size_t)r * a.D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < a.D) ElemIO<TY>::store(y + c, (v[i] - mean) * rstd * a.gamma[c] + a.beta[c]);
  }
  if (lane == 0) { a.mean[r] = mean; a.rstd[r] = rstd; }
}

template <typename TDY, int NV>
__global__ __launch_bounds__(256) void k_ln_bwd(uvc_ln_args a, int rpb) {
  __shared__ float red[4][2 * 64 * NV + 2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float gam[NV], dgam[NV], dbet[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    gam[i] = c < a.D ? a.gamma[c] : 0.f;
    dgam[i] = 0.f; dbet[i] = 0.f;
  }
  const float a1 = a.a1 ? *a.a1 : 1.f, a2 = a.a2 ? *a.a2 : 1.f;
  float dotA = 0.f, dotB = 0.f;
  const int r0 = blockIdx.x * rpb;
  const int r1 = min(a.rows, r0 + rpb);
  const float invD = 1.0f / (float)a.D;
  for (int r = r0 + w; r < r1; r += 4) {
    const size_t off = row_off(r, a.rows_per_group, a.group_stride, a.D);
    const float* x = reinterpret_cast<const float*>(a.x) + off;
    const TDY* dy = reinterpret_cast<const TDY*>(a.dy) + (size_t)r * a.D;
    const float mean = a.mean[r], rstd = a.rstd[r];
    float xh[NV], gy[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < a.D;
      const float d = ok ? ElemIO<TDY>::load(dy + c) : 0.f;
      xh[i] = ok ? (x[c] - mean) * rstd : 0.f;
      gy[i] = d * gam[i];
      dgam[i] += d * xh[i];
      dbet[i] += d;
      c1 += gy[i];
      c2 += gy[i] * xh[i];
    }
    c1 = wave_sum(c1) * invD;
    c2 = wave_sum(c2) * invD;
    float* dx = reinterpret_cast<float*>(a.dx) + off;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < a.D) {
        float o = rstd * (gy[i] - c1 - xh[i] * c2);
        const float xv = x[c];
        if (a.add1) o += a1 * reinterpret_cast<const float*>(a.add1)[off + c];
        if (a.add2) { const float t = reinterpret_cast<const float*>(a.add2)[off + c]; o += a2 * t; dotB += t * xv; }
        dx[c] = o;
        dotA += o * xv;
      }
    }
  }
  // cross-wave reduction in fixed order
#pragma unroll
  for (int i = 0; i < NV; ++i) { red[w][lane + 64 * i] = dgam[i]; red[w][64 * NV + lane + 64 * i] = dbet[i]; }
  dotA = wave_sum(dotA); dotB = wave_sum(dotB);
  if (lane == 0) { red[w][2 * 64 * NV] = dotA; red[w][2 * 64 * NV + 1] = dotB; }
  __syncthreads();
  float* P = a.partial + (size_t)blockIdx.x * (2 * a.D + 2);
  for (int c = threadIdx.x; c < 64 * NV; c += 256) {
    if (c < a.D) {
      P[c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
      P[a.D + c] = ((red[0][64 * NV + c] + red[1][64 * NV + c]) + red[2][64 * NV + c]) + red[3][64 * NV + c];
    }
  }
  if (threadIdx.x < 2) {
    const int c = 2 * 64 * NV + threadIdx.x;
    P[2 * a.D + threadIdx.x] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
  }
}

// sum the per-block partials: 256 threads = 64 columns x 4 slices of blocks
__global__ __launch_bounds__(256) void k_ln_bwd_reduce(uvc_ln_args a, int nblocks) {
  __shared__ float red[4][64];
  const int W = 2 * a.D + 2;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < W)
    for (int b = sl; b < nblocks; b += 4) s += a.partial[(size_t)b * W + c];
  red[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < W) {
    const int t = threadIdx.x;
    const float tot = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
    if (c < a.D) a.dgamma[c] = (a.beta_acc != 0.f ? a.beta_acc * a.dgamma[c] : 0.f) + tot;
    else if (c < 2 * a.D) a.dbeta[c - a.D] = (a.beta_acc != 0.f ? a.beta_acc * a.dbeta[c - a.D] : 0.f) + tot;
    else if (a.dots) a.dots[c - 2 * a.D] = tot;
  }
}


// ---- vectorised path (D % 64 == 0): 16 lanes per row x 16-byte accesses, 4 rows per wave in flight ----
template <typename T> struct Ld4;
template <> struct Ld4<float> {
  static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Ld4<bf16_t> {
  static __device__ __forceinline__ f32x4 ld(const bf16_t* p) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(p);
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
  }
  static __device__ __forceinline__ void st(bf16_t* p, f32x4 v) {
    u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = r;
  }
};
// four elements as they sit in memory (no conversion): what a row set waiting in flight costs in registers
template <typename T> struct Raw4;
template <> struct Raw4<float> {
  typedef f32x4 V;
  static __device__ __forceinline__ V ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ V zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4 cvt(const V& r) { return r; }
};
template <> struct Raw4<bf16_t> {
  typedef u32x2 V;
  static __device__ __forceinline__ V ld(const bf16_t* p) { return *reinterpret_cast<const u32x2*>(p); }
  static __device__ __forceinline__ V zero() { return u32x2{0u, 0u}; }
  static __device__ __forceinline__ f32x4 cvt(const V& r) {
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
  }
};
__device__ __forceinline__ float sum16(float v) { return xor_tree_sum<16>(v); }       // (same bits as the xor-shuffle butterfly 1, 2, 4, 8: common.h)

constexpr int LNV_FWD_ROWS = 64;     // rows per block (16 per wave, 4 at a time)
// TX: element type of x (float32, or bf16 = the bf16 residual stream of the throughput mode; statistics stay float32)
template <typename TX, typename TY, int NV4>
__global__ __launch_bounds__(256) void k_ln_fwd_v(uvc_ln_args a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane & 15, rg = lane >> 4;
  f32x4 gam[NV4], bet[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) { gam[i] = Ld4<float>::ld(a.gamma + (sub + 16 * i) * 4); bet[i] = Ld4<float>::ld(a.beta + (sub + 16 * i) * 4); }
  const float invD = 1.0f / (float)a.D;
  const int rbase = blockIdx.x * LNV_FWD_ROWS + w * 16;
  // the wave's four row sets (4 rows each) are ALL requested up front, in the registers they are loaded into (bf16 rows: 2 registers per 4
  // elements), and converted when their turn comes: beside another stream's kernels this kernel gets a fraction of its three workgroups per
  // CU, and with one or two sets in flight per wave it was an HBM round trip per 4-8 rows (26 us in the DeiT-Small step against 17 alone)
  constexpr int PF = NV4 >= 12 ? 2 : 4;                    // (D = 768: two sets at a time -- four cost 252 registers)
  typename Raw4<TX>::V raw[PF][NV4];
#pragma unroll
  for (int it0 = 0; it0 < 4; it0 += PF) {
#pragma unroll
  for (int it = it0; it < it0 + PF; ++it) {
    const int r = rbase + it * 4 + rg;
    const bool ok = r < a.rows;
    const TX* x = reinterpret_cast<const TX*>(a.x) + (ok ? row_off(r, a.rows_per_group, a.group_stride, a.D) : 0);
#pragma unroll
    for (int i = 0; i < NV4; ++i) raw[it - it0][i] = ok ? Raw4<TX>::ld(x + (sub + 16 * i) * 4) : Raw4<TX>::zero();
  }
#pragma unroll
  for (int it = it0; it < it0 + PF; ++it) {
    const int r = rbase + it * 4 + rg;
    const bool ok = r < a.rows;
    f32x4 v[NV4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      v[i] = Raw4<TX>::cvt(raw[it - it0][i]);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = sum16(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = rsqrtf(sum16(q) * invD + a.eps);
    if (ok) {
      TY* y = reinterpret_cast<TY*>(a.y) + (size_t)r * a.D;
#pragma unroll
      for (int i = 0; i < NV4; ++i) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gam[i][e] + bet[i][e];
        Ld4<TY>::st(y + (sub + 16 * i) * 4, o);
      }
      if (sub == 0) { a.mean[r] = mean; a.rstd[r] = rstd; }
    }
  }
  }
}

// TG: element type of the gradient stream (dx, add1, add2): float32, or bf16 when uvc_ln_args.g_lowp is set.
// LPR lanes share a row (D = 4 * LPR * NV4): 16 for D <= 192, 32 for 384, 64 for 768, so that NV4 stays <= 3 and the two
// row register sets fit (with 16 lanes per row D = 384 took all 256 VGPRs and D = 768 spilled ~750 registers to scratch).
template <int LPR> __device__ __forceinline__ float sum_lpr(float v) { return xor_tree_sum<LPR>(v); }     // (the butterfly 1, 2, ..., LPR / 2)
template <typename TX, typename TDY, typename TG, int NV4, int LPR>
__global__ __launch_bounds__(256) void k_ln_bwd_v(uvc_ln_args a, int rpb) {
  constexpr int DD = 4 * LPR * NV4, RPW = 64 / LPR;        // row length; rows a wave handles at a time
  __shared__ float red[4][2 * DD + 2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, sub = lane & (LPR - 1), rg = lane / LPR;
  f32x4 gam[NV4], dgam[NV4], dbet[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    gam[i] = Ld4<float>::ld(a.gamma + (sub + LPR * i) * 4);
    dgam[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dbet[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float a1 = a.a1 ? *a.a1 : 1.f, a2 = a.a2 ? *a.a2 : 1.f;
  float dotA = 0.f, dotB = 0.f;
  const int r0 = blockIdx.x * rpb;
  const int r1 = min(a.rows, r0 + rpb);
  const float invD = 1.0f / (float)a.D;
  const TG* add1 = reinterpret_cast<const TG*>(a.add1);
  const TG* add2 = reinterpret_cast<const TG*>(a.add2);
  struct Row { f32x4 xv[NV4], dv[NV4], ad1[NV4], ad2[NV4]; float mean, rstd; size_t off; bool ok; };
  // a row set as loaded (bf16 tensors: 2 registers per 4 elements): THREE of them are in flight per wave, converted when their turn comes
  struct RawRow {
    typename Raw4<TX>::V xv[NV4]; typename Raw4<TDY>::V dv[NV4]; typename Raw4<TG>::V ad1[NV4], ad2[NV4];
    float mean, rstd; size_t off; bool ok;
  };
  auto load_raw = [&](RawRow& R, int rb) {
    const int r = rb + rg;
    R.ok = rb < r1 && r < r1;
    R.off = R.ok ? row_off(r, a.rows_per_group, a.group_stride, a.D) : 0;
    const TX* x = reinterpret_cast<const TX*>(a.x) + R.off;
    const TDY* dy = reinterpret_cast<const TDY*>(a.dy) + (size_t)(R.ok ? r : 0) * a.D;
    R.mean = R.ok ? a.mean[r] : 0.f; R.rstd = R.ok ? a.rstd[r] : 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      R.ad1[i] = (R.ok && add1) ? Raw4<TG>::ld(add1 + R.off + (sub + LPR * i) * 4) : Raw4<TG>::zero();
      R.ad2[i] = (R.ok && add2) ? Raw4<TG>::ld(add2 + R.off + (sub + LPR * i) * 4) : Raw4<TG>::zero();
      R.dv[i] = R.ok ? Raw4<TDY>::ld(dy + (sub + LPR * i) * 4) : Raw4<TDY>::zero();
      R.xv[i] = R.ok ? Raw4<TX>::ld(x + (sub + LPR * i) * 4) : Raw4<TX>::zero();
    }
  };
  auto convert = [&](const RawRow& W) {
    Row R;
    R.mean = W.mean; R.rstd = W.rstd; R.off = W.off; R.ok = W.ok;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      R.xv[i] = Raw4<TX>::cvt(W.xv[i]); R.dv[i] = Raw4<TDY>::cvt(W.dv[i]);
      R.ad1[i] = Raw4<TG>::cvt(W.ad1[i]); R.ad2[i] = Raw4<TG>::cvt(W.ad2[i]);
    }
    return R;
  };
  auto process = [&](const Row& R) {
    f32x4 gy[NV4];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (R.xv[i][e] - R.mean) * R.rstd;
        gy[i][e] = R.dv[i][e] * gam[i][e];
        dgam[i][e] += R.dv[i][e] * xh;
        dbet[i][e] += R.dv[i][e];
        c1 += gy[i][e];
        c2 += gy[i][e] * xh;
      }
    c1 = sum_lpr<LPR>(c1) * invD;
    c2 = sum_lpr<LPR>(c2) * invD;
    if (R.ok) {
      TG* dx = reinterpret_cast<TG*>(a.dx) + R.off;
#pragma unroll
      for (int i = 0; i < NV4; ++i) {
        const int c = (sub + LPR * i) * 4;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = R.rstd * (gy[i][e] - c1 - ((R.xv[i][e] - R.mean) * R.rstd) * c2);
        if (a.add1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += a1 * R.ad1[i][e]; }
        if (a.add2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] += a2 * R.ad2[i][e]; dotB += R.ad2[i][e] * R.xv[i][e]; } }
#pragma unroll
        for (int e = 0; e < 4; ++e) dotA += o[e] * R.xv[i][e];
        Ld4<TG>::st(dx + c, o);
      }
    }
  };
  // Three row sets per wave in flight, in the registers they were loaded into; the row in turn is converted and processed (same
  // arithmetic and the same row order per lane as one set at a time: same bits).  Beside the weight-gradient stream this kernel gets
  // one workgroup per CU instead of three, and with one set per wave its loop was an HBM round trip per 2-4 rows: 133 us in the
  // DeiT-Small step against 71 alone (round 2's float32 second set cost 205 VGPRs and lost in the step; the raw sets cost 2-4 each
  // per 4 elements).
  {
    constexpr int S = 4 * RPW;
    RawRow A, B, C;
    int rb = r0 + w * RPW;
    load_raw(A, rb); load_raw(B, rb + S); load_raw(C, rb + 2 * S);
    for (; rb < r1; rb += 3 * S) {
      process(convert(A)); load_raw(A, rb + 3 * S);
      process(convert(B)); load_raw(B, rb + 4 * S);
      process(convert(C)); load_raw(C, rb + 5 * S);
    }
  }
  // reduce the 4 row groups of the wave, then the 4 waves (fixed order)
#pragma unroll
  for (int i = 0; i < NV4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = dgam[i][e], b = dbet[i][e];
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) { g += __shfl_xor(g, o, 64); b += __shfl_xor(b, o, 64); }
      if (rg == 0) { red[w][(sub + LPR * i) * 4 + e] = g; red[w][DD + (sub + LPR * i) * 4 + e] = b; }
    }
  dotA = wave_sum(dotA); dotB = wave_sum(dotB);
  if (lane == 0) { red[w][2 * DD] = dotA; red[w][2 * DD + 1] = dotB; }
  __syncthreads();
  float* P = a.partial + (size_t)blockIdx.x * (2 * a.D + 2);
  for (int c = threadIdx.x; c < 2 * DD + 2; c += 256) P[c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// two-stage reduction of the per-block partials: [nblocks, W] -> [RED_S, W] -> [W]
constexpr int RED_S = 16;
__global__ __launch_bounds__(256) void k_ln_bwd_reduce1(const float* __restrict__ partial, int nblocks, int W, float* __restrict__ p2) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6, s = blockIdx.y;
  const int b0 = (int)((long)nblocks * s / RED_S), b1 = (int)((long)nblocks * (s + 1) / RED_S);
  float acc = 0.f;
  if (c < W)
    for (int b = b0 + sl; b < b1; b += 4) acc += partial[(size_t)b * W + c];
  red[sl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sl == 0 && c < W) { const int t = threadIdx.x; p2[(size_t)s * W + c] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]; }
}
__global__ __launch_bounds__(256) void k_ln_bwd_reduce2(uvc_ln_args a, const float* __restrict__ p2) {
  const int W = 2 * a.D + 2;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= W) return;
  float tot = 0.f;
#pragma unroll
  for (int s = 0; s < RED_S; ++s) tot += p2[(size_t)s * W + c];
  if (c < a.D) a.dgamma[c] = (a.beta_acc != 0.f ? a.beta_acc * a.dgamma[c] : 0.f) + tot;
  else if (c < 2 * a.D) a.dbeta[c - a.D] = (a.beta_acc != 0.f ? a.beta_acc * a.dbeta[c - a.D] : 0.f) + tot;
  else if (a.dots) a.dots[c - 2 * a.D] = tot;
}

// batched finish of deferred calls: block (x, y) = 64 columns of item y; 1024 threads = 64 columns x 16 row slices
struct LnBatch { uvc_ln_reduce_item it[64]; };
__global__ __launch_bounds__(1024) void k_ln_bwd_reduce_batch(LnBatch b, int D, float beta_acc) {
  __shared__ float red[16][64];
  const uvc_ln_reduce_item& it = b.it[blockIdx.y];
  const int W = 2 * D + 2;
  const int tx = threadIdx.x & 63, sl = threadIdx.x >> 6, c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < W)
    for (int r = sl; r < it.nblocks; r += 16) s += it.partial[(size_t)r * W + c];
  red[sl][tx] = s;
  __syncthreads();
  if (sl == 0 && c < W) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][tx];
    if (c < D) it.dgamma[c] = (beta_acc != 0.f ? beta_acc * it.dgamma[c] : 0.f) + tot;
    else if (c < 2 * D) it.dbeta[c - D] = (beta_acc != 0.f ? beta_acc * it.dbeta[c - D] : 0.f) + tot;
    else if (it.dots) it.dots[c - 2 * D] = tot;
  }
}

int check(const uvc_ln_args* p) {
  if (!p || !p->x || !p->gamma || !p->mean || !p->rstd) return uvc_set_error_msg(UVC_ERR_ARG, "layernorm: null pointer");
  if (p->rows <= 0 || p->D <= 0 || p->D > 1024) return uvc_set_error_msg(UVC_ERR_ARG, "layernorm: need 0 < D <= 1024");
  if (p->rows_per_group <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "layernorm: rows_per_group");
  return UVC_OK;
}

template <typename T> int launch_fwd(const uvc_ln_args& a, hipStream_t st) {
  if (a.D % 64 == 0 && (a.group_stride % 4) == 0) {
    const int grid = ceil_div(a.rows, LNV_FWD_ROWS);
#define LNF_CASE(NV4) case NV4: if (a.x_lowp) k_ln_fwd_v<bf16_t, T, NV4><<<grid, 256, 0, st>>>(a); else k_ln_fwd_v<float, T, NV4><<<grid, 256, 0, st>>>(a); break;
    switch (a.D / 64) {
      LNF_CASE(1) LNF_CASE(2) LNF_CASE(3) LNF_CASE(4) LNF_CASE(6) LNF_CASE(8) LNF_CASE(12)
      default: goto generic;
    }
#undef LNF_CASE
    UVC_CHECK_LAUNCH();
    return UVC_OK;
  }
generic:
  // any other width (and ragged group strides): one wave per row, scalar accesses; x may be the bf16 residual stream here too
  // (the engine's bf16 mode defaults to bf16 rows for every embed_dim % 64 == 0 it accepts)
  const int grid = ceil_div(a.rows, 4);
  const int nv = ceil_div(a.D, 64);
#define LNG(NV) do { if (a.x_lowp) k_ln_fwd<bf16_t, T, NV><<<grid, 256, 0, st>>>(a); else k_ln_fwd<float, T, NV><<<grid, 256, 0, st>>>(a); } while (0)
  if (nv <= 2) LNG(2);
  else if (nv <= 3) LNG(3);
  else if (nv <= 6) LNG(6);
  else if (nv <= 12) LNG(12);
  else LNG(16);
#undef LNG
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
template <typename T> int launch_bwd(const uvc_ln_args& a, hipStream_t st) {
  const int rpb = ln_rows_per_block(a.rows);
  const int grid = ceil_div(a.rows, rpb);
  const int nv = ceil_div(a.D, 64);
  bool vec = a.D % 64 == 0 && (a.group_stride % 4) == 0;
  if (vec) {
#define LNB_CASE(DV, NV4, LPR) case DV: \
      if (a.x_lowp) { if (a.g_lowp) k_ln_bwd_v<bf16_t, T, bf16_t, NV4, LPR><<<grid, 256, 0, st>>>(a, rpb); else k_ln_bwd_v<bf16_t, T, float, NV4, LPR><<<grid, 256, 0, st>>>(a, rpb); } \
      else { if (a.g_lowp) k_ln_bwd_v<float, T, bf16_t, NV4, LPR><<<grid, 256, 0, st>>>(a, rpb); else k_ln_bwd_v<float, T, float, NV4, LPR><<<grid, 256, 0, st>>>(a, rpb); } break;
    switch (a.D) {
      LNB_CASE(64, 1, 16) LNB_CASE(128, 2, 16) LNB_CASE(192, 3, 16) LNB_CASE(256, 2, 32) LNB_CASE(384, 3, 32) LNB_CASE(512, 2, 64) LNB_CASE(768, 3, 64)
      default: vec = false;
    }
#undef LNB_CASE
  }
  if (!vec && (a.g_lowp || a.x_lowp)) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "layernorm_bwd: a bf16 gradient stream / bf16 x needs D % 64 == 0");
  if (!vec) {
    if (nv <= 2) k_ln_bwd<T, 2><<<grid, 256, 0, st>>>(a, rpb);
    else if (nv <= 3) k_ln_bwd<T, 3><<<grid, 256, 0, st>>>(a, rpb);
    else if (nv <= 6) k_ln_bwd<T, 6><<<grid, 256, 0, st>>>(a, rpb);
    else if (nv <= 12) k_ln_bwd<T, 12><<<grid, 256, 0, st>>>(a, rpb);
    else k_ln_bwd<T, 16><<<grid, 256, 0, st>>>(a, rpb);
  }
  UVC_CHECK_LAUNCH();
  if (a.defer_reduce) return UVC_OK;
  const int W = 2 * a.D + 2;
  if (grid >= 4 * RED_S) {
    float* p2 = a.partial + (size_t)grid * W;            // scratch tail (uvc_layernorm_bwd_blocks reserves it)
    k_ln_bwd_reduce1<<<dim3(ceil_div(W, 64), RED_S), 256, 0, st>>>(a.partial, grid, W, p2);
    UVC_CHECK_LAUNCH();
    k_ln_bwd_reduce2<<<ceil_div(W, 256), 256, 0, st>>>(a, p2);
  } else {
    k_ln_bwd_reduce<<<ceil_div(W, 64), 256, 0, st>>>(a, grid);
  }
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

}  // namespace

// number of [2D+2]-float rows the backward scratch must hold: per-block partials + the second reduction stage
extern "C" int uvc_layernorm_bwd_blocks(int32_t rows) { return ceil_div(rows, ln_rows_per_block(rows)) + RED_S; }
// number of partial rows a call over `rows` rows writes (the nblocks of its uvc_ln_reduce_item)
extern "C" int uvc_layernorm_bwd_nblocks(int32_t rows) { return ceil_div(rows, ln_rows_per_block(rows)); }

extern "C" int uvc_layernorm_fwd(const uvc_ln_args* p, void* stream) {
  if (int e = check(p)) return e;
  if (!p->y || !p->beta) return uvc_set_error_msg(UVC_ERR_ARG, "layernorm_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (p->y_is_f32 || p->dtype == UVC_F32) return launch_fwd<float>(*p, st);
  return launch_fwd<bf16_t>(*p, st);
}

extern "C" int uvc_layernorm_bwd(const uvc_ln_args* p, void* stream) {
  if (int e = check(p)) return e;
  if (!p->dy || !p->dx || !p->partial || !p->dgamma || !p->dbeta) return uvc_set_error_msg(UVC_ERR_ARG, "layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (p->dy_is_f32 || p->dtype == UVC_F32) return launch_bwd<float>(*p, st);
  return launch_bwd<bf16_t>(*p, st);
}

extern "C" int uvc_layernorm_bwd_reduce_batch(const uvc_ln_reduce_item* items, int32_t n, int32_t D, float beta_acc, void* stream) {
  if (!items || n <= 0 || n > 64 || D <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_layernorm_bwd_reduce_batch: 1..64 items");
  LnBatch b;
  memset(&b, 0, sizeof(b));
  for (int i = 0; i < n; ++i) {
    if (!items[i].partial || !items[i].dgamma || !items[i].dbeta || items[i].nblocks <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_layernorm_bwd_reduce_batch: bad item");
    b.it[i] = items[i];
  }
  k_ln_bwd_reduce_batch<<<dim3(ceil_div(2 * D + 2, 64), n), 1024, 0, (hipStream_t)stream>>>(b, D, beta_acc);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
