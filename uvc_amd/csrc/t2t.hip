// T2T-ViT tokens-to-token front end (include/uvc_t2t.h): soft split fused with the stage's LayerNorm, its adjoint (fold),
// and the Performer's linear attention, forward and backward.  Follows UVC/T2TViT/models/t2t_vit.py:84-105 and
// token_performer.py:31-62 of the reference.  HBM-bound token streams (3136 / 784 / 196 tokens per image, 64-wide):
// one pass over each stream, float32 arithmetic on the VALU, LDS tiles of 64 tokens, deterministic two-level sums.
#include "common.h"
#include "../../include/uvc_kernels.h"
#include "../../include/uvc_t2t.h"

namespace {

// ================================================================================================
//                                  soft split (+ LayerNorm)
// ================================================================================================
struct UG {                       // unfold geometry + pointers, passed by value
  const float* src; int64_t sb, sc, sh, sw;
  int B, C, H, W, k, s, p, Ho, Wo, L, kk, dim, ldo, rows, c_fast;
  const float* gamma; const float* beta; float eps;
  void* out; float* mean; float* rstd;
  const void* dy; float* dxu; float* partial;
};

// One wave gathers one unfolded row into registers in natural feature order e = lane + 64*j (e = c*kk + ki*k + kj).
// CF: token-major sources (C == 64, sc == 1) are read with the channel on the lane -- 256-byte coalesced loads -- and
// transposed to the natural order through a wave-private LDS row (stride kk is odd: conflict-free).
// Integer divisions stay out of the row loop (with runtime k / kk / L they were ~2000 VALU cycles per row, more than the loads):
// workgroups walk (image, output row) pairs and their waves the output columns, the taps of the token-major path advance
// incrementally, and the generic path decomposes its features once per lane (LaneTaps).
template <int NV> struct LaneTaps { int off[NV], ki[NV], kj[NV]; };
template <int NV>
__device__ __forceinline__ LaneTaps<NV> lane_taps(const UG& g, int lane) {
  LaneTaps<NV> t;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = lane + 64 * j;
    const int c = e / g.kk, r = e % g.kk;
    t.ki[j] = r / g.k; t.kj[j] = r % g.k;
    t.off[j] = e < g.dim ? (int)(c * g.sc + t.ki[j] * g.sh + t.kj[j] * g.sw) : -1;
  }
  return t;
}
template <int NV, bool CF>
__device__ __forceinline__ void gather_row(const UG& g, const LaneTaps<NV>& tp, int b, int ho, int wo, bool valid, int lane, float* lrow, float (&v)[NV]) {
  const int h0 = ho * g.s - g.p, w0 = wo * g.s - g.p;
  if (CF) {
    const float* sb = g.src + (int64_t)b * g.sb + lane;
    int ki = 0, kj = 0;
    for (int j = 0; j < g.kk; ++j) {
      const int hi = h0 + ki, wi = w0 + kj;
      float x = 0.f;
      if (valid && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) x = sb[(int64_t)hi * g.sh + (int64_t)wi * g.sw];
      lrow[lane * g.kk + j] = x;
      if (++kj == g.k) { kj = 0; ++ki; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = (lane + 64 * j < g.dim) ? lrow[lane + 64 * j] : 0.f;
    __syncthreads();
  } else {
    const float* sb = g.src + (int64_t)b * g.sb + (int64_t)h0 * g.sh + (int64_t)w0 * g.sw;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int hi = h0 + tp.ki[j], wi = w0 + tp.kj[j];
      float x = 0.f;
      if (valid && tp.off[j] >= 0 && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) x = sb[tp.off[j]];
      v[j] = x;
    }
  }
}
// One output row leaves as 16 bytes per lane: the wave parks its NV values per lane in its LDS row (natural order) and
// reads them back as contiguous chunks -- one or two full-width store instructions instead of NV two-byte ones per lane.
// DS operations of one wave execute in order; the wave barrier only keeps the compiler from reordering them.
template <typename TO, int NV>
__device__ __forceinline__ void store_row(float* lrow, const float (&v)[NV], TO* o, int ldo, int lane, bool valid) {
#pragma unroll
  for (int j = 0; j < NV; ++j) lrow[lane + 64 * j] = v[j];
  __builtin_amdgcn_wave_barrier();
  constexpr int VN = sizeof(TO) == 2 ? 8 : 4;
#pragma unroll
  for (int i = 0; i < (NV * 64 / VN + 63) / 64; ++i) {
    const int c = (lane + 64 * i) * VN;
    if (valid && c < ldo) {
      const f32x4 lo = *reinterpret_cast<const f32x4*>(lrow + c);
      if (sizeof(TO) == 2) {
        const f32x4 hi = *reinterpret_cast<const f32x4*>(lrow + c + 4);
        u32x4 r;
        r[0] = pack_bf16x2(lo[0], lo[1]); r[1] = pack_bf16x2(lo[2], lo[3]); r[2] = pack_bf16x2(hi[0], hi[1]); r[3] = pack_bf16x2(hi[2], hi[3]);
        *reinterpret_cast<u32x4*>(o + c) = r;
      } else {
        *reinterpret_cast<f32x4*>(o + c) = lo;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <typename TO, int NV, bool CF>
__global__ __launch_bounds__(256) void k_unfold_ln(UG g) {
  __shared__ __attribute__((aligned(16))) float lds[4][NV * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  LaneTaps<NV> tp;
  if (!CF) tp = lane_taps<NV>(g, lane);
  float gam[NV], bet[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = lane + 64 * j;
    gam[j] = (g.gamma && e < g.dim) ? g.gamma[e] : 0.f;
    bet[j] = (g.gamma && e < g.dim) ? g.beta[e] : 0.f;
  }
  for (int bh = blockIdx.x; bh < g.B * g.Ho; bh += gridDim.x) {
    const int b = bh / g.Ho, ho = bh - b * g.Ho;
    for (int wo0 = 0; wo0 < g.Wo; wo0 += 4) {
      const int wo = wo0 + wv;
      const bool valid = wo < g.Wo;
      const int row = bh * g.Wo + wo;
      float v[NV];
      gather_row<NV, CF>(g, tp, b, ho, wo, valid, lane, lds[wv], v);
      if (g.gamma) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += v[j];
        const float mean = wave_sum(s) / (float)g.dim;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) { const float d = (lane + 64 * j < g.dim) ? v[j] - mean : 0.f; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)g.dim + g.eps);
        if (valid && lane == 0) { g.mean[row] = mean; g.rstd[row] = rstd; }
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = (lane + 64 * j < g.dim) ? (v[j] - mean) * rstd * gam[j] + bet[j] : 0.f;
      }
      store_row<TO, NV>(lds[wv], v, (TO*)g.out + (int64_t)(valid ? row : 0) * g.ldo, g.ldo, lane, valid);
    }
  }
}

// LayerNorm backward of the fused kernel: recomputes the unfolded row, writes dxu (float32, natural order) and leaves the
// per-workgroup partial sums of dgamma / dbeta in `partial` ([gridDim.x][2*dim]).
template <typename TDY, int NV, bool CF>
__global__ __launch_bounds__(256) void k_unfold_ln_bwd(UG g) {
  __shared__ __attribute__((aligned(16))) float lds[4][NV * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float dgm[NV], dbt[NV], gam[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { dgm[j] = 0.f; dbt[j] = 0.f; gam[j] = (lane + 64 * j < g.dim) ? g.gamma[lane + 64 * j] : 0.f; }
  const float inv = 1.0f / (float)g.dim;
  LaneTaps<NV> tp;
  if (!CF) tp = lane_taps<NV>(g, lane);
  for (int bh = blockIdx.x; bh < g.B * g.Ho; bh += gridDim.x) {
    const int b = bh / g.Ho, ho = bh - b * g.Ho;
    for (int wo0 = 0; wo0 < g.Wo; wo0 += 4) {
      const int wo = wo0 + wv;
      const bool valid = wo < g.Wo;
      const int row = bh * g.Wo + wo;
      float v[NV];
      gather_row<NV, CF>(g, tp, b, ho, wo, valid, lane, lds[wv], v);
      const float mean = valid ? g.mean[row] : 0.f, rstd = valid ? g.rstd[row] : 0.f;
      const TDY* dyr = (const TDY*)g.dy + (int64_t)(valid ? row : 0) * g.ldo;
      float gy[NV], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        const float dy = (valid && e < g.dim) ? ElemIO<TDY>::load(dyr + e) : 0.f;
        v[j] = (e < g.dim) ? (v[j] - mean) * rstd : 0.f;          // xhat
        dgm[j] += dy * v[j];
        dbt[j] += dy;
        gy[j] = dy * gam[j];
        s1 += gy[j];
        s2 += gy[j] * v[j];
      }
      if (g.dxu) {
        s1 = wave_sum(s1) * inv;
        s2 = wave_sum(s2) * inv;
        if (valid) {
          float* o = g.dxu + (int64_t)row * g.dim;
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const int e = lane + 64 * j;
            if (e < g.dim) o[e] = rstd * (gy[j] - s1 - v[j] * s2);
          }
        }
      }
    }
  }
  __syncthreads();
  // waves 1..3 hand their sums to wave 0 through LDS, one quantity at a time (fixed order)
  for (int pass = 0; pass < 2; ++pass) {
    if (wv > 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) lds[wv][lane + 64 * j] = pass == 0 ? dgm[j] : dbt[j];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        const float t = (pass == 0 ? dgm[j] : dbt[j]) + lds[1][e] + lds[2][e] + lds[3][e];
        if (e < g.dim) g.partial[(int64_t)blockIdx.x * 2 * g.dim + pass * g.dim + e] = t;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void k_unfold_bwd_reduce(const float* partial, int nblk, int dim, float* dgamma, float* dbeta, float beta_acc) {
  __shared__ float red[16][64];
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6;        // 64 columns x 16 row groups, combined in a fixed order
  const int c = blockIdx.x * 64 + l;
  float t = 0.f;
  if (c < 2 * dim)
    for (int b = q; b < nblk; b += 16) t += partial[(int64_t)b * 2 * dim + c];
  red[q][l] = t;
  __syncthreads();
  if (q == 0 && c < 2 * dim) {
    t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][l];
    float* o = c < dim ? dgamma + c : dbeta + (c - dim);
    *o = (beta_acc != 0.f ? beta_acc * *o : 0.f) + t;
  }
}

template <typename TS, typename TD>
__global__ void k_fold(const TS* src, int lds, TD* dst, int B, int C, int H, int W, int k, int s, int p, int Ho, int Wo) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * H * W * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int hw = (int)((idx / C) % (H * W));
  const int b = (int)(idx / ((int64_t)C * H * W));
  const int h = hw / W, w = hw % W, kk = k * k, L = Ho * Wo;
  float acc = 0.f;
  for (int ki = 0; ki < k; ++ki) {
    const int hh = h + p - ki;
    if (hh < 0 || hh % s) continue;
    const int ho = hh / s;
    if (ho >= Ho) continue;
    for (int kj = 0; kj < k; ++kj) {
      const int ww = w + p - kj;
      if (ww < 0 || ww % s) continue;
      const int wo = ww / s;
      if (wo >= Wo) continue;
      acc += ElemIO<TS>::load(src + ((int64_t)b * L + ho * Wo + wo) * lds + c * kk + ki * k + kj);
    }
  }
  ElemIO<TD>::store(dst + idx, acc);
}

int fill_geom(const uvc_unfold_args* a, UG& g) {
  if (!a || !a->src) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold: null pointer");
  if (a->B <= 0 || a->C <= 0 || a->k <= 0 || a->s <= 0 || a->p < 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold: geometry");
  g.src = a->src; g.sb = a->sb; g.sc = a->sc; g.sh = a->sh; g.sw = a->sw;
  g.B = a->B; g.C = a->C; g.H = a->H; g.W = a->W; g.k = a->k; g.s = a->s; g.p = a->p;
  g.Ho = (a->H + 2 * a->p - a->k) / a->s + 1; g.Wo = (a->W + 2 * a->p - a->k) / a->s + 1;
  g.L = g.Ho * g.Wo; g.kk = a->k * a->k; g.dim = a->C * g.kk; g.ldo = a->ldo; g.rows = a->B * g.L;
  g.c_fast = (a->sc == 1 && a->C == 64) ? 1 : 0;
  g.gamma = a->gamma; g.beta = a->beta; g.eps = a->eps; g.out = a->out; g.mean = a->mean; g.rstd = a->rstd;
  g.dy = a->dy; g.dxu = a->dxu; g.partial = a->partial;
  if (g.Ho <= 0 || g.Wo <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold: empty output");
  if (g.dim > 576 || g.ldo < g.dim || g.ldo > ((g.dim + 63) / 64) * 64) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_unfold: need C*k*k <= 576 and dim <= ldo <= roundup(dim, 64)");
  return UVC_OK;
}
int bwd_grid(int rows) { const int n = (rows + 3) / 4; return n < 1024 ? n : 1024; }

// ================================================================================================
//                                  Performer linear attention
// ================================================================================================
constexpr int PE = 64, PM = 32, PT = 64, PKV = 65 * 32;
// LDS row strides in floats: 64-wide tiles 68, 32-wide tiles 36 -- rows stay 16-byte aligned, so the inner products read four
// consecutive elements per ds_read_b128 (the scalar-read form was LDS-bound: 9 reads for 8 FMAs), and both strides are 4 x odd:
// a wave's lanes reading 16 bytes each from consecutive rows cover all 64 banks exactly once.
constexpr int S64 = 68, S32 = 36;
#define SQRT_M 5.656854249492381f

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float c) { c += a[0] * b[0]; c += a[1] * b[1]; c += a[2] * b[2]; c += a[3] * b[3]; return c; }

__device__ __forceinline__ void load_w(const float* w, float (*sw)[S64], int tid) {
#pragma unroll
  for (int it = 0; it < 2; ++it) { const int idx = (tid + it * 256) * 4; *reinterpret_cast<f32x4*>(&sw[idx >> 6][idx & 63]) = ld4(w + idx); }
}
// [64 tokens][64] float32 tile of kqv (row stride 192) -> LDS; rows at or beyond T are zero
__device__ __forceinline__ void load_tile_kqv(const float* base, int t0, int T, float (*s)[S64], int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + it * 256, r = idx >> 4, c4 = (idx & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t0 + r < T) v = ld4(base + (int64_t)(t0 + r) * 192 + c4);
    *reinterpret_cast<f32x4*>(&s[r][c4]) = v;
  }
}
// [64 tokens][64] tile of a dense gradient stream [B*T, 64] (float32 or bf16) -> LDS
template <typename TG>
__device__ __forceinline__ void load_tile_g(const TG* base, int t0, int T, float (*s)[S64], int tid) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + it * 256, r = idx >> 4, c4 = (idx & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t0 + r < T) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ElemIO<TG>::load(base + (int64_t)(t0 + r) * 64 + c4 + e);
    }
    *reinterpret_cast<f32x4*>(&s[r][c4]) = v;
  }
}
__device__ __forceinline__ void load_kv(const float* kv, float (*skv)[S32], int tid) {
  for (int idx = tid * 4; idx < PKV; idx += 1024) *reinterpret_cast<f32x4*>(&skv[idx >> 5][idx & 31]) = ld4(kv + idx);
}
// positive random features (token_performer.py:31-43): thread (token t, feature group mg) -> 8 of the 32 features
__device__ __forceinline__ void prm8(const float (*sx)[S64], const float (*sw)[S64], int t, int mg, float (&p)[8]) {
  float d[8], xx = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = 0.f;
#pragma unroll 4
  for (int i = 0; i < PE; i += 4) {
    const f32x4 x = ld4(&sx[t][i]);
    xx = dot4(x, x, xx);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = dot4(x, ld4(&sw[mg * 8 + j][i]), d[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = expf(d[j] - 0.5f * xx) / SQRT_M;
}
// acc[j] += x * row[j], j < 8, row = 8 consecutive floats (two 16-byte broadcast reads)
__device__ __forceinline__ void axpy8(float x, const float* row, float (&acc)[8]) {
  const f32x4 a = ld4(row), b = ld4(row + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { acc[j] += x * a[j]; acc[4 + j] += x * b[j]; }
}
__device__ __forceinline__ void st8(float* row, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(row) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(row + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
// sum_m a[m] * b[m] over the 32 features of two 36-stride rows
__device__ __forceinline__ float dot32(const float* a, const float* b) {
  float c = 0.f;
#pragma unroll
  for (int m = 0; m < PM; m += 4) c = dot4(ld4(a + m), ld4(b + m), c);
  return c;
}

// kptv / ksum partials of one (image, split): sum_t v_t kp_t^T and sum_t kp_t over the split's token tiles
__global__ __launch_bounds__(256) void k_performer_kv(const float* kqv, const float* w, float* part, int T, int S, int tps) {
  __shared__ __attribute__((aligned(16))) float sw[PM][S64], sk[PT][S64], sv[PT][S64], skp[PT][S32];
  const int tid = threadIdx.x, b = blockIdx.x / S, sp = blockIdx.x % S;
  const int ntile = (T + PT - 1) / PT;
  const int t = tid & 63, mg = tid >> 6;
  const float* base = kqv + (int64_t)b * T * 192;
  load_w(w, sw, tid);
  float acc[8], ks = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int tile_end = (sp + 1) * tps < ntile ? (sp + 1) * tps : ntile;
  // the k / v rows of the NEXT tile travel in registers under the arithmetic of the current one: with the loads in front of the barrier every
  // tile of the loop paid an HBM round trip with three workgroups per CU to cover it
  f32x4 rk[4], rv[4];
  auto fetch = [&](int tile) {
    const int t0 = tile * PT;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256, r = idx >> 4, c4 = (idx & 15) * 4;
      const bool ok = tile < tile_end && t0 + r < T;
      rk[it] = ok ? ld4(base + (int64_t)(t0 + r) * 192 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
      rv[it] = ok ? ld4(base + 128 + (int64_t)(t0 + r) * 192 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch(sp * tps);
  for (int tile = sp * tps; tile < tile_end; ++tile) {
    const int t0 = tile * PT;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256, r = idx >> 4, c4 = (idx & 15) * 4;
      *reinterpret_cast<f32x4*>(&sk[r][c4]) = rk[it];
      *reinterpret_cast<f32x4*>(&sv[r][c4]) = rv[it];
    }
    fetch(tile + 1);
    __syncthreads();
    float p[8];
    prm8(sk, sw, t, mg, p);
    if (t0 + t >= T) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = 0.f;
    }
    st8(&skp[t][mg * 8], p);
    __syncthreads();
#pragma unroll 16
    for (int tt = 0; tt < PT; ++tt) axpy8(sv[tt][t], &skp[tt][mg * 8], acc);
    if (tid < PM)
#pragma unroll 8
      for (int tt = 0; tt < PT; ++tt) ks += skp[tt][tid];
  }
  float* o = part + (int64_t)blockIdx.x * PKV;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[t * PM + mg * 8 + j] = acc[j];
  if (tid < PM) o[64 * PM + tid] = ks;
}

__global__ void k_part_reduce(const float* part, float* out, int S) {
  const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (e >= PKV) return;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += part[((int64_t)b * S + s) * PKV + e];
  out[(int64_t)b * PKV + e] = t;
}

template <typename TO>
__global__ __launch_bounds__(256) void k_performer_q(const float* kqv, const float* w, const float* kptv, TO* att, int T, int ntile) {
  __shared__ __attribute__((aligned(16))) float sw[PM][S64], sq[PT][S64], skv[65][S32], sqp[PT][S32], sden[4][PT];
  const int tid = threadIdx.x, b = blockIdx.x / ntile, t0 = (blockIdx.x % ntile) * PT;
  const int t = tid & 63, mg = tid >> 6;
  load_w(w, sw, tid);
  load_tile_kqv(kqv + (int64_t)b * T * 192 + 64, t0, T, sq, tid);
  load_kv(kptv + (int64_t)b * PKV, skv, tid);
  __syncthreads();
  float p[8], pd = 0.f;
  prm8(sq, sw, t, mg, p);
  st8(&sqp[t][mg * 8], p);
#pragma unroll
  for (int j = 0; j < 8; ++j) pd += p[j] * skv[64][mg * 8 + j];
  sden[mg][t] = pd;
  __syncthreads();
  const int n = t;
#pragma unroll 2
  for (int tt = mg * 16; tt < mg * 16 + 16; ++tt) {
    const float num = dot32(sqp[tt], skv[n]);
    const float den = ((sden[0][tt] + sden[1][tt]) + (sden[2][tt] + sden[3][tt])) + 1e-8f;
    if (t0 + tt < T) ElemIO<TO>::store(att + ((int64_t)b * T + t0 + tt) * PE + n, num / den);
  }
}

// q side of the backward: dq, and the (image, split) partials of dkptv / dksum
template <typename TG>
__global__ __launch_bounds__(256) void k_performer_bwd_q(const float* kqv, const float* w, const float* kptv, const TG* datt, TG* dkqv, float* part,
                                                         int T, int S, int tps) {
  __shared__ __attribute__((aligned(16))) float sw[PM][S64], sq[PT][S64], sdy[PT][S64], skv[65][S32], sqp[PT][S32], sden[4][PT], sdden[PT];
  const int tid = threadIdx.x, b = blockIdx.x / S, sp = blockIdx.x % S;
  const int ntile = (T + PT - 1) / PT;
  const int t = tid & 63, mg = tid >> 6;
  load_w(w, sw, tid);
  load_kv(kptv + (int64_t)b * PKV, skv, tid);
  __syncthreads();
  float wc[PM];                                                         // column t of w: dq[.][t] = sum_m g[.][m] * w[m][t]
#pragma unroll
  for (int m = 0; m < PM; ++m) wc[m] = sw[m][t];
  float acc[8], dks = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int tile_end = (sp + 1) * tps < ntile ? (sp + 1) * tps : ntile;
  for (int tile = sp * tps; tile < tile_end; ++tile) {
    const int t0 = tile * PT;
    __syncthreads();
    load_tile_kqv(kqv + (int64_t)b * T * 192 + 64, t0, T, sq, tid);
    load_tile_g<TG>(datt + (int64_t)b * T * 64, t0, T, sdy, tid);
    __syncthreads();
    {                                                                  // P1: qp and the denominator
      float p[8], pd = 0.f;
      prm8(sq, sw, t, mg, p);
#pragma unroll
      for (int j = 0; j < 8; ++j) { if (t0 + t >= T) p[j] = 0.f; pd += p[j] * skv[64][mg * 8 + j]; }
      st8(&sqp[t][mg * 8], p);
      sden[mg][t] = pd;
    }
    __syncthreads();
#pragma unroll 2
  for (int tt = mg * 16; tt < mg * 16 + 16; ++tt) {                  // P2: dnum (in place over dy), dden; thread (n = t, 16 tokens)
      const float num = dot32(sqp[tt], skv[t]);
      const float den = ((sden[0][tt] + sden[1][tt]) + (sden[2][tt] + sden[3][tt])) + 1e-8f;
      const float dy = sdy[tt][t];
      const float dot = wave_sum(dy * num);
      sdy[tt][t] = dy / den;
      if (t == 0) sdden[tt] = -dot / (den * den);
    }
    __syncthreads();
#pragma unroll 8
    for (int tt = 0; tt < PT; ++tt) axpy8(sdy[tt][t], &sqp[tt][mg * 8], acc);     // P4: dkptv[n][m] += dnum[tt][n] qp[tt][m]
    if (tid < PM)
#pragma unroll 8
      for (int tt = 0; tt < PT; ++tt) dks += sdden[tt] * sqp[tt][tid];
    __syncthreads();
    {                                                                  // P3: g = dqp * qp in place over qp; thread (token t, 8 features)
      float dqp[8];
      const float dd = sdden[t];
#pragma unroll
      for (int j = 0; j < 8; ++j) dqp[j] = dd * skv[64][mg * 8 + j];
#pragma unroll 2
      for (int n = 0; n < PE; n += 4) {
        const f32x4 dn = ld4(&sdy[t][n]);
#pragma unroll
        for (int e = 0; e < 4; ++e) axpy8(dn[e], &skv[n + e][mg * 8], dqp);
      }
      const f32x4 q0 = ld4(&sqp[t][mg * 8]), q1 = ld4(&sqp[t][mg * 8 + 4]);
      float gq[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { gq[j] = q0[j] * dqp[j]; gq[4 + j] = q1[j] * dqp[4 + j]; }
      st8(&sqp[t][mg * 8], gq);
    }
    __syncthreads();
#pragma unroll 2
  for (int tt = mg * 16; tt < mg * 16 + 16; ++tt) {                  // P5: dq[tt][i] = sum_m g (w[m][i] - q[tt][i]); thread (i = t, 16 tokens)
      float a = 0.f, gs = 0.f;
#pragma unroll
      for (int m = 0; m < PM; m += 4) {
        const f32x4 gm = ld4(&sqp[tt][m]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a += gm[e] * wc[m + e]; gs += gm[e]; }
      }
      if (t0 + tt < T) ElemIO<TG>::store(dkqv + ((int64_t)b * T + t0 + tt) * 192 + 64 + t, a - sq[tt][t] * gs);
    }
  }
  float* o = part + (int64_t)blockIdx.x * PKV;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[t * PM + mg * 8 + j] = acc[j];
  if (tid < PM) o[64 * PM + tid] = dks;
}

// k / v side of the backward, one 64-token tile per workgroup
template <typename TG>
__global__ __launch_bounds__(256) void k_performer_bwd_k(const float* kqv, const float* w, const float* dkptv, const TG* dskip, TG* dkqv, int T, int ntile) {
  __shared__ __attribute__((aligned(16))) float sw[PM][S64], sk[PT][S64], sv[PT][S64], sdk[65][S32], skp[PT][S32];
  const int tid = threadIdx.x, b = blockIdx.x / ntile, t0 = (blockIdx.x % ntile) * PT;
  const int t = tid & 63, mg = tid >> 6;
  const float* base = kqv + (int64_t)b * T * 192;
  load_w(w, sw, tid);
  load_tile_kqv(base, t0, T, sk, tid);
  load_tile_kqv(base + 128, t0, T, sv, tid);
  load_kv(dkptv + (int64_t)b * PKV, sdk, tid);
  __syncthreads();
  {
    float p[8];
    prm8(sk, sw, t, mg, p);
    st8(&skp[t][mg * 8], p);
  }
  __syncthreads();
#pragma unroll 2
  for (int tt = mg * 16; tt < mg * 16 + 16; ++tt) {                    // dv[tt][n] = sum_m kp[tt][m] dkptv[n][m] (+ skip gradient)
    float a = dot32(skp[tt], sdk[t]);
    if (t0 + tt < T) {
      const int64_t r = (int64_t)b * T + t0 + tt;
      if (dskip) a += ElemIO<TG>::load(dskip + r * 64 + t);
      ElemIO<TG>::store(dkqv + r * 192 + 128 + t, a);
    }
  }
  __syncthreads();
  {                                                                    // g = dkp * kp in place; thread (token t, 8 features)
    float dkp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dkp[j] = sdk[64][mg * 8 + j];
#pragma unroll 2
    for (int n = 0; n < PE; n += 4) {
      const f32x4 vv = ld4(&sv[t][n]);
#pragma unroll
      for (int e = 0; e < 4; ++e) axpy8(vv[e], &sdk[n + e][mg * 8], dkp);
    }
    const f32x4 k0 = ld4(&skp[t][mg * 8]), k1 = ld4(&skp[t][mg * 8 + 4]);
    float gk[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { gk[j] = k0[j] * dkp[j]; gk[4 + j] = k1[j] * dkp[4 + j]; }
    st8(&skp[t][mg * 8], gk);
  }
  __syncthreads();
  float wc[PM];
#pragma unroll
  for (int m = 0; m < PM; ++m) wc[m] = sw[m][t];
#pragma unroll 2
  for (int tt = mg * 16; tt < mg * 16 + 16; ++tt) {                    // dk[tt][i] = sum_m g (w[m][i] - k[tt][i])
    float a = 0.f, gs = 0.f;
#pragma unroll
    for (int m = 0; m < PM; m += 4) {
      const f32x4 gm = ld4(&skp[tt][m]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a += gm[e] * wc[m + e]; gs += gm[e]; }
    }
    if (t0 + tt < T) ElemIO<TG>::store(dkqv + ((int64_t)b * T + t0 + tt) * 192 + t, a - sk[tt][t] * gs);
  }
}

// Token tiles per split are a constant, so the order in which an image's kptv / dkptv sums are formed does not depend on the batch
// size: an image's result is bit-identical whether it is processed alone or in a batch of 512.
constexpr int PTPS = 7;     // 49 tiles (56 x 56 tokens) = 7 even splits
int splits_of(int B, int T) {
  (void)B;
  const int ntile = (T + PT - 1) / PT;
  return (ntile + PTPS - 1) / PTPS;
}
int check_perf(const uvc_performer_args* p) {
  if (!p || !p->kqv || !p->w || !p->part || !p->kptv) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_performer: null pointer");
  if (p->B <= 0 || p->T <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_performer: B, T");
  if (p->dtype != UVC_F32 && p->dtype != UVC_BF16) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_performer: dtype");
  return UVC_OK;
}

}  // namespace

extern "C" int uvc_unfold_bwd_blocks(int32_t rows) { return bwd_grid(rows); }

extern "C" int uvc_unfold_ln_fwd(const uvc_unfold_args* a, void* stream) {
  UG g;
  if (int e = fill_geom(a, g)) return e;
  if (!a->out || (a->gamma && (!a->beta || !a->mean || !a->rstd))) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold_ln_fwd: null pointer");
  if (g.ldo % 8) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold_ln_fwd: ldo must be a multiple of 8 (16-byte row stores)");
  hipStream_t st = (hipStream_t)stream;
  int grid = g.B * g.Ho;
  if (grid > 16384) grid = 16384;
  const bool f32 = a->out_is_f32 || a->dtype == UVC_F32;
  const int nv = (g.dim + 63) / 64;
#define UF_LAUNCH(NVV, CFF) do { if (f32) k_unfold_ln<float, NVV, CFF><<<grid, 256, 0, st>>>(g); else k_unfold_ln<bf16_t, NVV, CFF><<<grid, 256, 0, st>>>(g); } while (0)
  if (nv <= 3) { if (g.c_fast) UF_LAUNCH(3, true); else UF_LAUNCH(3, false); }
  else { if (g.c_fast) UF_LAUNCH(9, true); else UF_LAUNCH(9, false); }
#undef UF_LAUNCH
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_unfold_ln_bwd(const uvc_unfold_args* a, void* stream) {
  UG g;
  if (int e = fill_geom(a, g)) return e;
  if (!a->gamma || !a->mean || !a->rstd || !a->dy || !a->partial || !a->dgamma || !a->dbeta) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold_ln_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int grid = g.B * g.Ho;
  if (grid > bwd_grid(g.rows)) grid = bwd_grid(g.rows);
  const bool f32 = a->dy_is_f32 || a->dtype == UVC_F32;
  const int nv = (g.dim + 63) / 64;
#define UB_LAUNCH(NVV, CFF) do { if (f32) k_unfold_ln_bwd<float, NVV, CFF><<<grid, 256, 0, st>>>(g); else k_unfold_ln_bwd<bf16_t, NVV, CFF><<<grid, 256, 0, st>>>(g); } while (0)
  if (nv <= 3) { if (g.c_fast) UB_LAUNCH(3, true); else UB_LAUNCH(3, false); }
  else { if (g.c_fast) UB_LAUNCH(9, true); else UB_LAUNCH(9, false); }
#undef UB_LAUNCH
  UVC_CHECK_LAUNCH();
  k_unfold_bwd_reduce<<<ceil_div(2 * g.dim, 64), 1024, 0, st>>>(g.partial, grid, g.dim, a->dgamma, a->dbeta, a->beta_acc);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_fold_tokens(const void* src, int32_t src_is_f32, int32_t dtype, int32_t lds, void* dst, int32_t dst_is_f32, int32_t B, int32_t C, int32_t H,
                               int32_t W, int32_t k, int32_t s, int32_t p, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || s <= 0 || p < 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_fold_tokens: arguments");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  if (Ho <= 0 || Wo <= 0 || lds < C * k * k) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_fold_tokens: geometry");
  const int64_t total = (int64_t)B * H * W * C;
  const int grid = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  const bool sf = src_is_f32 || dtype == UVC_F32, df = dst_is_f32 || dtype == UVC_F32;
  if (sf && df) k_fold<float, float><<<grid, 256, 0, st>>>((const float*)src, lds, (float*)dst, B, C, H, W, k, s, p, Ho, Wo);
  else if (sf) k_fold<float, bf16_t><<<grid, 256, 0, st>>>((const float*)src, lds, (bf16_t*)dst, B, C, H, W, k, s, p, Ho, Wo);
  else if (df) k_fold<bf16_t, float><<<grid, 256, 0, st>>>((const bf16_t*)src, lds, (float*)dst, B, C, H, W, k, s, p, Ho, Wo);
  else k_fold<bf16_t, bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, lds, (bf16_t*)dst, B, C, H, W, k, s, p, Ho, Wo);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_performer_splits(int32_t B, int32_t T) { return splits_of(B, T); }

extern "C" int uvc_performer_fwd(const uvc_performer_args* p, void* stream) {
  if (int e = check_perf(p)) return e;
  if (!p->att) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_performer_fwd: null att");
  hipStream_t st = (hipStream_t)stream;
  const int ntile = (p->T + PT - 1) / PT, S = splits_of(p->B, p->T), tps = PTPS;
  k_performer_kv<<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->part, p->T, S, tps);
  UVC_CHECK_LAUNCH();
  k_part_reduce<<<dim3(ceil_div(PKV, 256), p->B), 256, 0, st>>>(p->part, p->kptv, S);
  UVC_CHECK_LAUNCH();
  if (p->att_is_f32 || p->dtype == UVC_F32) k_performer_q<float><<<p->B * ntile, 256, 0, st>>>(p->kqv, p->w, p->kptv, (float*)p->att, p->T, ntile);
  else k_performer_q<bf16_t><<<p->B * ntile, 256, 0, st>>>(p->kqv, p->w, p->kptv, (bf16_t*)p->att, p->T, ntile);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_performer_bwd(const uvc_performer_args* p, void* stream) {
  if (int e = check_perf(p)) return e;
  if (!p->datt || !p->dkqv || !p->dkptv) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_performer_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int ntile = (p->T + PT - 1) / PT, S = splits_of(p->B, p->T), tps = PTPS;
  const bool f32 = p->g_is_f32 || p->dtype == UVC_F32;
  if (f32) k_performer_bwd_q<float><<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->kptv, (const float*)p->datt, (float*)p->dkqv, p->part, p->T, S, tps);
  else k_performer_bwd_q<bf16_t><<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->kptv, (const bf16_t*)p->datt, (bf16_t*)p->dkqv, p->part, p->T, S, tps);
  UVC_CHECK_LAUNCH();
  k_part_reduce<<<dim3(ceil_div(PKV, 256), p->B), 256, 0, st>>>(p->part, p->dkptv, S);
  UVC_CHECK_LAUNCH();
  if (f32) k_performer_bwd_k<float><<<p->B * ntile, 256, 0, st>>>(p->kqv, p->w, p->dkptv, (const float*)p->dskip, (float*)p->dkqv, p->T, ntile);
  else k_performer_bwd_k<bf16_t><<<p->B * ntile, 256, 0, st>>>(p->kqv, p->w, p->dkptv, (const bf16_t*)p->dskip, (bf16_t*)p->dkqv, p->T, ntile);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
