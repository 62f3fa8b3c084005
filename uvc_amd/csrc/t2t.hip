// T2T-ViT tokens-to-token front end (include/uvc_t2t.h): soft split fused with the stage's LayerNorm, its adjoint (fold),
// and the Performer's linear attention, forward and backward.  Follows UVC/T2TViT/models/t2t_vit.py:84-105 and
// token_performer.py:31-62 of the reference.  HBM-bound token streams (3136 / 784 / 196 tokens per image, 64-wide):
// one pass over each stream, float32 arithmetic (the Performer's inner products on v_mfma_f32_16x16x4_f32 since round 4), tiles of 64 tokens,
// deterministic two-level sums.
#include "common.h"
#include "../../include/uvc_kernels.h"
#include "../../include/uvc_t2t.h"
#include <string.h>

namespace {

// ================================================================================================
//                                  soft split (+ LayerNorm)
// ================================================================================================
struct UG {                       // unfold geometry + pointers, passed by value
  const float* src; int64_t sb, sc, sh, sw;
  int B, C, H, W, k, s, p, Ho, Wo, L, kk, dim, ldo, rows, c_fast;
  const float* gamma; const float* beta; float eps;
  void* out; float* mean; float* rstd;
  const void* dy; float* dxu; float* partial; int dxu_tm;
};

// One wave gathers one unfolded row into registers in natural feature order e = lane + 64*j (e = c*kk + ki*k + kj).
// CF: token-major sources (C == 64, sc == 1) are read with the channel on the lane -- 256-byte coalesced loads -- and
// transposed to the natural order through a wave-private LDS row (stride kk is odd: conflict-free).
// Integer divisions stay out of the row loop (with runtime k / kk / L they were ~2000 VALU cycles per row, more than the loads):
// workgroups walk (image, output row) pairs and their waves the output columns, the taps of the token-major path advance
// incrementally, and the generic path decomposes its features once per lane (LaneTaps).
// wave sums of the forward kernels by DPP + row swaps (common.h: xor_tree_sum) instead of wave_sum's six dependent ds_bpermute round trips: with the LayerNorm
// arithmetic removed the image split went 149 -> 87 us; image split forward 139 -> 125 us, stage 2 forward 87 -> 67 us (same box)
__device__ __forceinline__ float wave_sum_dpp(float v) { return xor_tree_sum<64>(v); }
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
// U rows per wave at a time, every load of the U rows issued before the first use (r4): a wave that gathers one row per trip pays a memory round trip per
// row (3 loads in flight on the generic path; on the token-major path the runtime tap loop wrote each tap to LDS as it arrived: nine round trips).  In the
// T2T-ViT-14 step: LayerNorm backward of stage 2 273 -> 199 us, the forwards 139 -> 126 and 65 -> 60 us; the image split (scattered 28-byte pieces per
// lane group) stays 2.5x off its bytes.  Token-major sources have k*k <= 9 taps (C = 64, dim <= 576).
struct TapTable { int ki[9], kj[9]; };
__device__ __forceinline__ TapTable tap_table(const UG& g) {
  TapTable t;
  int ki = 0, kj = 0;
#pragma unroll
  for (int j = 0; j < 9; ++j) { t.ki[j] = ki; t.kj[j] = kj; if (++kj == g.k) { kj = 0; ++ki; } }
  return t;
}
struct RowId { int b, ho, wo, row; bool valid; };
__device__ __forceinline__ RowId row_id(const UG& g, int row) {
  RowId r;
  r.row = row; r.valid = row < g.rows;
  const int rr = r.valid ? row : 0;
  const int bh = rr / g.Wo;
  r.wo = rr - bh * g.Wo; r.b = bh / g.Ho; r.ho = bh - r.b * g.Ho;
  return r;
}
template <int NV, bool CF, int U>
__device__ __forceinline__ void gather_rows(const UG& g, const LaneTaps<NV>& tp, const TapTable& tt, const RowId (&id)[U], int lane, float (*lrow)[NV * 64], float (&v)[U][NV]) {
  if (CF) {
    float x[U][9];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int h0 = id[u].ho * g.s - g.p, w0 = id[u].wo * g.s - g.p;
      const float* sb = g.src + (int64_t)id[u].b * g.sb + lane;
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int hi = h0 + tt.ki[j], wi = w0 + tt.kj[j];
        x[u][j] = (id[u].valid && j < g.kk && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? sb[(int64_t)hi * g.sh + (int64_t)wi * g.sw] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 9; ++j)
        if (j < g.kk) lrow[u][lane * g.kk + j] = x[u][j];
    __builtin_amdgcn_wave_barrier();                 // the rows are the wave's own; DS operations of one wave execute in order
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NV; ++j) v[u][j] = (lane + 64 * j < g.dim) ? lrow[u][lane + 64 * j] : 0.f;
    __builtin_amdgcn_wave_barrier();
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int h0 = id[u].ho * g.s - g.p, w0 = id[u].wo * g.s - g.p;
      const float* sb = g.src + (int64_t)id[u].b * g.sb + (int64_t)h0 * g.sh + (int64_t)w0 * g.sw;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int hi = h0 + tp.ki[j], wi = w0 + tp.kj[j];
        v[u][j] = (id[u].valid && tp.off[j] >= 0 && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W) ? sb[tp.off[j]] : 0.f;
      }
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
  constexpr int U = NV <= 3 ? 4 : 2;                  // rows per wave and trip
  __shared__ __attribute__((aligned(16))) float lds[4][U][NV * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  LaneTaps<NV> tp;
  if (!CF) tp = lane_taps<NV>(g, lane);
  const TapTable tt = tap_table(g);
  float gam[NV], bet[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int e = lane + 64 * j;
    gam[j] = (g.gamma && e < g.dim) ? g.gamma[e] : 0.f;
    bet[j] = (g.gamma && e < g.dim) ? g.beta[e] : 0.f;
  }
  for (int row0 = blockIdx.x * 4 * U; row0 < g.rows; row0 += gridDim.x * 4 * U) {
    RowId id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = row_id(g, row0 + 4 * u + wv);
    float v[U][NV];
    gather_rows<NV, CF, U>(g, tp, tt, id, lane, lds[wv], v);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (g.gamma) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += v[u][j];
        const float mean = wave_sum_dpp(s) / (float)g.dim;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) { const float d = (lane + 64 * j < g.dim) ? v[u][j] - mean : 0.f; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum_dpp(q) / (float)g.dim + g.eps);
        if (id[u].valid && lane == 0) { g.mean[id[u].row] = mean; g.rstd[id[u].row] = rstd; }
#pragma unroll
        for (int j = 0; j < NV; ++j) v[u][j] = (lane + 64 * j < g.dim) ? (v[u][j] - mean) * rstd * gam[j] + bet[j] : 0.f;
      }
      store_row<TO, NV>(lds[wv][u], v[u], (TO*)g.out + (int64_t)(id[u].valid ? id[u].row : 0) * g.ldo, g.ldo, lane, id[u].valid);
    }
  }
}

// LayerNorm backward of the fused kernel: recomputes the unfolded row, writes dxu (float32, natural order) and leaves the
// per-workgroup partial sums of dgamma / dbeta in `partial` ([gridDim.x][2*dim]).
template <typename TDY, int NV, bool CF>
__global__ __launch_bounds__(256) void k_unfold_ln_bwd(UG g) {
  constexpr int U = NV <= 3 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float lds[4][U][NV * 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float dgm[NV], dbt[NV], gam[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { dgm[j] = 0.f; dbt[j] = 0.f; gam[j] = (lane + 64 * j < g.dim) ? g.gamma[lane + 64 * j] : 0.f; }
  const float inv = 1.0f / (float)g.dim;
  LaneTaps<NV> tp;
  if (!CF) tp = lane_taps<NV>(g, lane);
  const TapTable tt = tap_table(g);
  for (int row0 = blockIdx.x * 4 * U; row0 < g.rows; row0 += gridDim.x * 4 * U) {
    RowId id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = row_id(g, row0 + 4 * u + wv);
    float v[U][NV], dyv[U][NV], mean[U], rstd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                     // the gradient rows and the row statistics travel with the gathers
      const TDY* dyr = (const TDY*)g.dy + (int64_t)(id[u].valid ? id[u].row : 0) * g.ldo;
#pragma unroll
      for (int j = 0; j < NV; ++j) dyv[u][j] = (id[u].valid && lane + 64 * j < g.dim) ? ElemIO<TDY>::load(dyr + lane + 64 * j) : 0.f;
      mean[u] = id[u].valid ? g.mean[id[u].row] : 0.f;
      rstd[u] = id[u].valid ? g.rstd[id[u].row] : 0.f;
    }
    gather_rows<NV, CF, U>(g, tp, tt, id, lane, lds[wv], v);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float gy[NV], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        const float dy = dyv[u][j];
        v[u][j] = (e < g.dim) ? (v[u][j] - mean[u]) * rstd[u] : 0.f;          // xhat
        dgm[j] += dy * v[u][j];
        dbt[j] += dy;
        gy[j] = dy * gam[j];
        s1 += gy[j];
        s2 += gy[j] * v[u][j];
      }
      if (g.dxu) {
        s1 = wave_sum(s1) * inv;            // (the DPP form measured slower here: 210 -> 246 us at stage 2)
        s2 = wave_sum(s2) * inv;
        float* o = g.dxu + (int64_t)(id[u].valid ? id[u].row : 0) * g.dim;
        if (CF && g.dxu_tm) {                         // tap-major: [k*k][64 channels], the channel on the lane (the fold reads whole channel rows)
#pragma unroll
          for (int j = 0; j < NV; ++j) lds[wv][u][lane + 64 * j] = rstd[u] * (gy[j] - s1 - v[u][j] * s2);
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int t = 0; t < 9; ++t)
            if (t < g.kk && id[u].valid) o[t * 64 + lane] = lds[wv][u][lane * g.kk + t];
          __builtin_amdgcn_wave_barrier();
        } else if (id[u].valid) {
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const int e = lane + 64 * j;
            if (e < g.dim) o[e] = rstd[u] * (gy[j] - s1 - v[u][j] * s2);
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
      for (int j = 0; j < NV; ++j) lds[wv][0][lane + 64 * j] = pass == 0 ? dgm[j] : dbt[j];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int e = lane + 64 * j;
        const float t = (pass == 0 ? dgm[j] : dbt[j]) + lds[1][0][e] + lds[2][0][e] + lds[3][0][e];
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

// Tap-major source ([rows][k*k][64 channels], uvc_unfold_args.dxu_tap_major): a wave takes PPW consecutive output pixels, the channel on the lane -- every
// tap is one 256-byte row and the pixel arithmetic is wave-uniform (scalar unit).  KK / SS / PP > 0: the taps are compile-time, so the loads of all PPW pixels are
// issued before the first sum (a wave with one pixel's 1-4 loads in flight was latency-bound: 222 us for 282 MB at stage 2 of T2T-ViT-14; the element-per-thread
// kernel above 199 us in either column order).
template <typename TS, typename TD, int KK, int SS, int PP, int PPW>
__global__ __launch_bounds__(256) void k_fold_tm(const TS* src, int lds, TD* dst, int B, int H, int W, int k_, int s_, int p_, int Ho, int Wo) {
  const int k = KK > 0 ? KK : k_, s = KK > 0 ? SS : s_, p = KK > 0 ? PP : p_;
  const int lane = threadIdx.x & 63;
  const int pix0 = ((int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * PPW;
  const int npix = B * H * W;
  float acc[PPW];
#pragma unroll
  for (int u = 0; u < PPW; ++u) {
    acc[u] = 0.f;
    const int pix = pix0 + u;
    if (pix >= npix) continue;
    const int b = pix / (H * W), hw = pix - b * H * W, h = hw / W, w = hw - h * W;
    const TS* sb = src + (int64_t)b * Ho * Wo * lds + lane;
    if (KK > 0) {
#pragma unroll
      for (int ki = 0; ki < (KK > 0 ? KK : 1); ++ki) {
        const int hh = h + p - ki, ho = hh / s;
        if (hh < 0 || hh % s || ho >= Ho) continue;
#pragma unroll
        for (int kj = 0; kj < (KK > 0 ? KK : 1); ++kj) {
          const int ww = w + p - kj, wo = ww / s;
          if (ww < 0 || ww % s || wo >= Wo) continue;
          acc[u] += ElemIO<TS>::load(sb + (int64_t)(ho * Wo + wo) * lds + (ki * k + kj) * 64);
        }
      }
    } else {
      for (int ki = 0; ki < k; ++ki) {
        const int hh = h + p - ki, ho = hh / s;
        if (hh < 0 || hh % s || ho >= Ho) continue;
        for (int kj = 0; kj < k; ++kj) {
          const int ww = w + p - kj, wo = ww / s;
          if (ww < 0 || ww % s || wo >= Wo) continue;
          acc[u] += ElemIO<TS>::load(sb + (int64_t)(ho * Wo + wo) * lds + (ki * k + kj) * 64);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < PPW; ++u)
    if (pix0 + u < npix) ElemIO<TD>::store(dst + (int64_t)(pix0 + u) * 64 + lane, acc[u]);
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
  g.dy = a->dy; g.dxu = a->dxu; g.partial = a->partial; g.dxu_tm = a->dxu_tap_major;
  if (g.dxu_tm && !g.c_fast) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_unfold: dxu_tap_major needs a token-major source (sc == 1, C == 64)");
  if (g.Ho <= 0 || g.Wo <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_unfold: empty output");
  if (g.dim > 576 || g.ldo < g.dim || g.ldo > ((g.dim + 63) / 64) * 64) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_unfold: need C*k*k <= 576 and dim <= ldo <= roundup(dim, 64)");
  return UVC_OK;
}
int bwd_grid(int rows) { const int n = (rows + 3) / 4; return n < 1024 ? n : 1024; }

// ================================================================================================
//                                  Performer linear attention
// ================================================================================================
constexpr int PE = 64, PM = 32, PT = 64, PKV = 65 * 32;
// (token_performer.py:31-69; kernel_ratio 0.5: m = 32 random features of the 64-wide k / q)
constexpr int S64 = 68;              // LDS row stride (floats) of a [token][64] tile written and read as 16-byte pieces (4 x odd: conflict-free for both)
#define SQRT_M 5.656854249492381f

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__global__ void k_part_reduce(const float* part, float* out, int S) {
  const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (e >= PKV) return;
  float t = 0.f;
  for (int s = 0; s < S; ++s) t += part[((int64_t)b * S + s) * PKV + e];
  out[(int64_t)b * PKV + e] = t;
}

// ------------------------------------------------------------------------------------------------
// (r4) The forward on the matrix pipe: v_mfma_f32_16x16x4_f32 (exact float32 products, float32 accumulation; twice the unpacked VALU rate, and the
// operands sit in registers -- the VALU form of rounds 2-3 issued 9 LDS reads per 32 FMAs and was bound by them: 283 -> 135 us forward, 660 -> 281 us
// backward at 128 x 3136 tokens).  Lane (i = lane & 15, g = lane >> 4) of the MFMA
// supplies A[i][k] and B[k][i] for k = 4 step + g and holds D[4 g + r][i], r < 4.  A contraction index may be permuted at will as long as both
// operands agree; a 64-float row is taken as four 16-byte reads at columns 16 q + 4 g (value 4 q + e <-> column 16 q + 4 g + e): the four lane groups
// read adjacent 16-byte pieces, 64-byte segments per row from global memory.
// ------------------------------------------------------------------------------------------------
constexpr int SV = 80, SP = 48;      // LDS row strides (floats) of tiles read by COLUMNS with one float per lane: 16 mod 64, so rows 4 s + g of the four lane groups
                                     // fall on distinct banks (SV: [token][64], SP: [token][32])
template <typename T> struct St4;                   // four consecutive outputs: 16 bytes of float32 / 8 bytes of bf16 (round-to-nearest-even, as ElemIO)
template <> struct St4<float> { static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; } };
template <> struct St4<bf16_t> {
  static __device__ __forceinline__ void st(bf16_t* p, f32x4 v) { u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]); *reinterpret_cast<u32x2*>(p) = r; }
};
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void frag64(const float* row, int g, float (&f)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = ld4(row + 16 * q + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[4 * q + e] = v[e];
  }
}
// positive random features of 16 tokens, transposed: p[ft][r] = feature 16 ft + 4 g + r of token (lane & 15); wf = the lane's W fragments (row 16 ft + i),
// xf = the token's row fragment (token_performer.py:31-43)
__device__ __forceinline__ void prm_t(const float (&wf)[2][16], const float (&xf)[16], float (&p)[2][4]) {
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
  float xx = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    c0 = mfma4(wf[0][s], xf[s], c0);
    c1 = mfma4(wf[1][s], xf[s], c1);
    xx += xf[s] * xf[s];
  }
  xx = sum_rows4(xx);
#pragma unroll
  for (int r = 0; r < 4; ++r) { p[0][r] = expf(c0[r] - 0.5f * xx) / SQRT_M; p[1][r] = expf(c1[r] - 0.5f * xx) / SQRT_M; }
}

// kptv / ksum partials of one (image, split): wave w computes kp of tokens 16 w .. of a 64-token tile (W kp-fragments in registers, the k rows straight from
// global memory, one tile ahead), then the [16 w ..][32] block of sum_t v_t kp_t^T with v^T as the A operand (column reads of the v tile) and kp as B
__global__ __launch_bounds__(256) void k_performer_kv_mfma(const float* kqv, const float* w, float* part, int T, int S, int tps) {
  __shared__ __attribute__((aligned(16))) float sv[PT][SV], skp[PT][SP], sred[4][PM];
  const int tid = threadIdx.x, b = blockIdx.x / S, sp = blockIdx.x % S;
  const int ntile = (T + PT - 1) / PT;
  const int lane = tid & 63, wv = tid >> 6, i = lane & 15, g = lane >> 4;
  const float* base = kqv + (int64_t)b * T * 192;
  float wf[2][16];
  frag64(w + i * 64, g, wf[0]);
  frag64(w + (16 + i) * 64, g, wf[1]);
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float ks[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) ks[0][r] = ks[1][r] = 0.f;
  const int tile_end = (sp + 1) * tps < ntile ? (sp + 1) * tps : ntile;
  float xk[16];
  f32x4 rv[4];
  auto fetch = [&](int tile) {                        // next tile: the wave's k fragments and the workgroup's v rows, in registers under this tile's arithmetic
    const int t0 = tile * PT;
    const bool okk = tile < tile_end && t0 + 16 * wv + i < T;
    if (okk) frag64(base + (int64_t)(t0 + 16 * wv + i) * 192, g, xk);
    else {
#pragma unroll
      for (int e = 0; e < 16; ++e) xk[e] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256, r = idx >> 4, c4 = (idx & 15) * 4;
      rv[it] = (tile < tile_end && t0 + r < T) ? ld4(base + 128 + (int64_t)(t0 + r) * 192 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  fetch(sp * tps);
  for (int tile = sp * tps; tile < tile_end; ++tile) {
    const int t0 = tile * PT;
    __syncthreads();                                  // the column reads of the tile before are done
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256, r = idx >> 4, c4 = (idx & 15) * 4;
      *reinterpret_cast<f32x4*>(&sv[r][c4]) = rv[it];
    }
    float p[2][4];
    prm_t(wf, xk, p);
    if (t0 + 16 * wv + i >= T) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[0][r] = p[1][r] = 0.f;
    }
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      *reinterpret_cast<f32x4*>(&skp[16 * wv + i][16 * ft + 4 * g]) = f32x4{p[ft][0], p[ft][1], p[ft][2], p[ft][3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) ks[ft][r] += p[ft][r];
    }
    fetch(tile + 1);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {                    // tokens 4 s + g
      const float a = sv[4 * s + g][16 * wv + i];
      acc[0] = mfma4(a, skp[4 * s + g][i], acc[0]);
      acc[1] = mfma4(a, skp[4 * s + g][16 + i], acc[1]);
    }
  }
  float* o = part + (int64_t)blockIdx.x * PKV;
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[(16 * wv + 4 * g + r) * PM + 16 * ft + i] = acc[ft][r];
      float v = ks[ft][r];                            // sum over the wave's 16 token columns, then over the four waves in a fixed order
#pragma unroll
      for (int x = 1; x < 16; x <<= 1) v += __shfl_xor(v, x, 64);
      if (i == 0) sred[wv][16 * ft + 4 * g + r] = v;
    }
  __syncthreads();
  if (tid < PM) o[64 * PM + tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
}

// attention rows of one 64-token tile: wave w takes tokens 16 w ..: qp^T by MFMA (q rows straight from global memory), the denominator from the lane's own
// features, num^T[n][token] = sum_m kptv[n][m] qp[token][m] with the qp fragment as it lies in the registers (B operand) and kptv as A
template <typename TO>
__global__ __launch_bounds__(256) void k_performer_q_mfma(const float* kqv, const float* w, const float* kptv, TO* att, int T, int ntile) {
  __shared__ __attribute__((aligned(16))) float so[PT][S64];
  const int tid = threadIdx.x, b = blockIdx.x / ntile, t0 = (blockIdx.x % ntile) * PT;
  const int lane = tid & 63, wv = tid >> 6, i = lane & 15, g = lane >> 4;
  const float* kv = kptv + (int64_t)b * PKV;
  float wf[2][16], xq[16];
  frag64(w + i * 64, g, wf[0]);
  frag64(w + (16 + i) * 64, g, wf[1]);
  const int tok = t0 + 16 * wv + i;
  if (tok < T) frag64(kqv + ((int64_t)b * T + tok) * 192 + 64, g, xq);
  else {
#pragma unroll
    for (int e = 0; e < 16; ++e) xq[e] = 0.f;
  }
  f32x4 akv[4][2];                                     // kptv[16 et + i][16 ft + 4 g ..]: the A operand, contraction index m = 16 ft + 4 g + r <-> step 4 ft + r
#pragma unroll
  for (int et = 0; et < 4; ++et)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) akv[et][ft] = ld4(kv + (16 * et + i) * PM + 16 * ft + 4 * g);
  const f32x4 ksum0 = ld4(kv + 64 * PM + 4 * g), ksum1 = ld4(kv + 64 * PM + 16 + 4 * g);
  float p[2][4];
  prm_t(wf, xq, p);
  float den = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) den += p[0][r] * ksum0[r];
#pragma unroll
  for (int r = 0; r < 4; ++r) den += p[1][r] * ksum1[r];
  den = sum_rows4(den);
  den += 1e-8f;
#pragma unroll
  for (int et = 0; et < 4; ++et) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) c = mfma4(akv[et][ft][r], p[ft][r], c);
    *reinterpret_cast<f32x4*>(&so[16 * wv + i][16 * et + 4 * g]) = f32x4{c[0] / den, c[1] / den, c[2] / den, c[3] / den};
  }
  __builtin_amdgcn_wave_barrier();                     // rows 16 w .. of `so` belong to this wave alone
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int r = 16 * wv + 4 * ps + g;
    const f32x4 v = ld4(&so[r][4 * i]);
    if (t0 + r < T) {
      St4<TO>::st(att + ((int64_t)b * T + t0 + r) * PE + 4 * i, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// (r4) The backward on the matrix pipe.  Every product below keeps the TOKEN as the MFMA column (lane & 15), so a lane's values of one step are the B
// operand of the next as they lie in the registers: qp^T -> num^T -> dnum^T -> dqp^T -> g^T -> dq^T; only the two sums over tokens (dkptv, dksum) cross
// the lanes, through LDS tiles read by columns.  Operand fragments of the small matrices (W, kptv, their transposes) are loaded once per workgroup.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Ld4;
template <> struct Ld4<float> { static __device__ __forceinline__ f32x4 ld(const float* p) { return ld4(p); } };
template <> struct Ld4<bf16_t> {
  static __device__ __forceinline__ f32x4 ld(const bf16_t* p) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(p);
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
  }
};
// A-operand fragments of a [64][32] matrix M (kptv / dkptv): direct (contraction over the 32 columns: a[et][ft] = M[16 et + i][16 ft + 4 g ..]) and
// transposed (contraction over the 64 rows: aT[mt][4 et + r] = M[16 et + 4 g + r][16 mt + i])
__device__ __forceinline__ void kv_frags(const float* M, int i, int g, f32x4 (&a)[4][2], float (&aT)[2][16]) {
#pragma unroll
  for (int et = 0; et < 4; ++et) {
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) a[et][ft] = ld4(M + (16 * et + i) * PM + 16 * ft + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      aT[0][4 * et + r] = M[(16 * et + 4 * g + r) * PM + i];
      aT[1][4 * et + r] = M[(16 * et + 4 * g + r) * PM + 16 + i];
    }
  }
}
// W^T fragments for dx^T[dim][token] = sum_m w[m][dim] g[token][m]: wT[dt][4 mt + r] = w[16 mt + 4 g + r][16 dt + i]
__device__ __forceinline__ void wt_frags(const float* w, int i, int g, float (&wT)[4][8]) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) wT[dt][4 * mt + r] = w[(16 * mt + 4 * g + r) * 64 + 16 * dt + i];
}
// c[et][r] = value of (token 16 wv + i, column 16 et + 4 g + r) -> rows of `dst` (row stride ld elements), through the wave's own 16 rows of `so`
template <typename TO>
__device__ __forceinline__ void rows_out(float (*so)[S64], int wv, int i, int g, const f32x4 (&c)[4], TO* dst, int ld, int t0, int T) {
#pragma unroll
  for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4*>(&so[16 * wv + i][16 * et + 4 * g]) = c[et];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int r = 16 * wv + 4 * ps + g;
    const f32x4 v = ld4(&so[r][4 * i]);
    if (t0 + r < T) St4<TO>::st(dst + (int64_t)(t0 + r) * ld + 4 * i, v);
  }
  __builtin_amdgcn_wave_barrier();
}
// x^T-side gradient of the random features: dx[token][dim] = sum_m g[token][m] w[m][dim] - x[token][dim] sum_m g[token][m]  (g = dp * p; xf = the token's row fragment)
__device__ __forceinline__ void prm_bwd_t(const float (&wT)[4][8], const float (&gq)[2][4], const float (&xf)[16], f32x4 (&dx)[4]) {
  float gs = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) gs += gq[0][r];
#pragma unroll
  for (int r = 0; r < 4; ++r) gs += gq[1][r];
  gs = sum_rows4(gs);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) c = mfma4(wT[dt][4 * mt + r], gq[mt][r], c);
#pragma unroll
    for (int r = 0; r < 4; ++r) dx[dt][r] = c[r] - xf[4 * dt + r] * gs;
  }
}

template <typename TG>
__global__ __launch_bounds__(256) void k_performer_bwd_q_mfma(const float* kqv, const float* w, const float* kptv, const TG* datt, TG* dkqv, float* part,
                                                              int T, int S, int tps) {
  __shared__ __attribute__((aligned(16))) float sdn[PT][SV], sqp[PT][SP], so[PT][S64], sred[4][PM];
  const int tid = threadIdx.x, b = blockIdx.x / S, sp = blockIdx.x % S;
  const int ntile = (T + PT - 1) / PT;
  const int lane = tid & 63, wv = tid >> 6, i = lane & 15, g = lane >> 4;
  const float* kv = kptv + (int64_t)b * PKV;
  float wf[2][16], akT[2][16], wT[4][8];
  f32x4 ak[4][2];
  frag64(w + i * 64, g, wf[0]);
  frag64(w + (16 + i) * 64, g, wf[1]);
  kv_frags(kv, i, g, ak, akT);
  wt_frags(w, i, g, wT);
  const f32x4 ksum0 = ld4(kv + 64 * PM + 4 * g), ksum1 = ld4(kv + 64 * PM + 16 + 4 * g);
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float dks[2][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dks[0][r] = dks[1][r] = 0.f;
  const int tile_end = (sp + 1) * tps < ntile ? (sp + 1) * tps : ntile;
  for (int tile = sp * tps; tile < tile_end; ++tile) {
    const int t0 = tile * PT, tok = t0 + 16 * wv + i;
    const bool live = tok < T;
    float xq[16];
    f32x4 dy[4];
    if (live) {
      frag64(kqv + ((int64_t)b * T + tok) * 192 + 64, g, xq);
#pragma unroll
      for (int et = 0; et < 4; ++et) dy[et] = Ld4<TG>::ld(datt + ((int64_t)b * T + tok) * 64 + 16 * et + 4 * g);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) xq[e] = 0.f;
#pragma unroll
      for (int et = 0; et < 4; ++et) dy[et] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float p[2][4];
    prm_t(wf, xq, p);
    if (!live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) p[0][r] = p[1][r] = 0.f;
    }
    float den = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) den += p[0][r] * ksum0[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) den += p[1][r] * ksum1[r];
    den = sum_rows4(den);
    den += 1e-8f;
    float dot = 0.f;                                   // sum_n dy num
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) c = mfma4(ak[et][ft][r], p[ft][r], c);
#pragma unroll
      for (int r = 0; r < 4; ++r) { dot += dy[et][r] * c[r]; dy[et][r] = dy[et][r] / den; }       // dy becomes dnum
    }
    dot = sum_rows4(dot);
    const float dden = -dot / (den * den);
    __syncthreads();                                   // the column reads of the tile before are done
#pragma unroll
    for (int et = 0; et < 4; ++et) *reinterpret_cast<f32x4*>(&sdn[16 * wv + i][16 * et + 4 * g]) = dy[et];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      *reinterpret_cast<f32x4*>(&sqp[16 * wv + i][16 * ft + 4 * g]) = f32x4{p[ft][0], p[ft][1], p[ft][2], p[ft][3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) dks[ft][r] += dden * p[ft][r];
    }
    float gq[2][4];                                    // g = dqp * qp,  dqp^T[m][token] = sum_n kptv[n][m] dnum[token][n] + dden ksum[m]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int et = 0; et < 4; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) c = mfma4(akT[mt][4 * et + r], dy[et][r], c);
      const f32x4 ksm = mt == 0 ? ksum0 : ksum1;
#pragma unroll
      for (int r = 0; r < 4; ++r) gq[mt][r] = (c[r] + dden * ksm[r]) * p[mt][r];
    }
    f32x4 dq[4];
    prm_bwd_t(wT, gq, xq, dq);
    rows_out<TG>(so, wv, i, g, dq, dkqv + (int64_t)b * T * 192 + 64, 192, t0, T);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {                     // dkptv[n][m] += sum_t dnum[t][n] qp[t][m], tokens 4 s + g; wave w: rows n = 16 w ..
      const float a = sdn[4 * s + g][16 * wv + i];
      acc[0] = mfma4(a, sqp[4 * s + g][i], acc[0]);
      acc[1] = mfma4(a, sqp[4 * s + g][16 + i], acc[1]);
    }
  }
  float* o = part + (int64_t)blockIdx.x * PKV;
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[(16 * wv + 4 * g + r) * PM + 16 * ft + i] = acc[ft][r];
      float v = dks[ft][r];
#pragma unroll
      for (int x = 1; x < 16; x <<= 1) v += __shfl_xor(v, x, 64);
      if (i == 0) sred[wv][16 * ft + 4 * g + r] = v;
    }
  __syncthreads();
  if (tid < PM) o[64 * PM + tid] = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
}

// k / v side: a workgroup walks `tps` consecutive 64-token tiles of one image (the operand fragments are loaded once)
template <typename TG>
__global__ __launch_bounds__(256) void k_performer_bwd_k_mfma(const float* kqv, const float* w, const float* dkptv, const TG* dskip, TG* dkqv, int T, int S, int tps) {
  __shared__ __attribute__((aligned(16))) float so[PT][S64];
  const int tid = threadIdx.x, b = blockIdx.x / S, sp = blockIdx.x % S;
  const int ntile = (T + PT - 1) / PT;
  const int lane = tid & 63, wv = tid >> 6, i = lane & 15, g = lane >> 4;
  const float* dk_ = dkptv + (int64_t)b * PKV;
  float wf[2][16], adT[2][16], wT[4][8];
  f32x4 ad[4][2];
  frag64(w + i * 64, g, wf[0]);
  frag64(w + (16 + i) * 64, g, wf[1]);
  kv_frags(dk_, i, g, ad, adT);
  wt_frags(w, i, g, wT);
  const f32x4 dks0 = ld4(dk_ + 64 * PM + 4 * g), dks1 = ld4(dk_ + 64 * PM + 16 + 4 * g);
  const int tile_end = (sp + 1) * tps < ntile ? (sp + 1) * tps : ntile;
  for (int tile = sp * tps; tile < tile_end; ++tile) {
    const int t0 = tile * PT, tok = t0 + 16 * wv + i;
    const bool live = tok < T;
    const int64_t row = (int64_t)b * T + tok;
    float xk[16], xv[16];
    f32x4 dsk[4];
    if (live) {
      frag64(kqv + row * 192, g, xk);
      frag64(kqv + row * 192 + 128, g, xv);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) xk[e] = xv[e] = 0.f;
    }
#pragma unroll
    for (int et = 0; et < 4; ++et) dsk[et] = (live && dskip) ? Ld4<TG>::ld(dskip + row * 64 + 16 * et + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    float p[2][4];
    prm_t(wf, xk, p);
    f32x4 dv[4];                                       // dv^T[n][token] = sum_m dkptv[n][m] kp[token][m] (+ skip gradient)
#pragma unroll
    for (int et = 0; et < 4; ++et) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) c = mfma4(ad[et][ft][r], p[ft][r], c);
      dv[et] = c + dsk[et];
    }
    rows_out<TG>(so, wv, i, g, dv, dkqv + (int64_t)b * T * 192 + 128, 192, t0, T);
    float gk[2][4];                                    // g = dkp * kp,  dkp^T[m][token] = sum_n dkptv[n][m] v[token][n] + dksum[m]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s) c = mfma4(adT[mt][s], xv[s], c);
      const f32x4 dsm = mt == 0 ? dks0 : dks1;
#pragma unroll
      for (int r = 0; r < 4; ++r) gk[mt][r] = (c[r] + dsm[r]) * p[mt][r];
    }
    f32x4 dk[4];
    prm_bwd_t(wT, gk, xk, dk);
    rows_out<TG>(so, wv, i, g, dk, dkqv + (int64_t)b * T * 192, 192, t0, T);
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
  int grid = (g.rows + 7) / 8;                         // (8 or 16 rows per workgroup and trip)
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
  int grid = bwd_grid(g.rows);
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
                               int32_t W, int32_t k, int32_t s, int32_t p, int32_t tap_major, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || s <= 0 || p < 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_fold_tokens: arguments");
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  if (Ho <= 0 || Wo <= 0 || lds < C * k * k) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_fold_tokens: geometry");
  const int64_t total = (int64_t)B * H * W * C;
  const int grid = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  const bool sf = src_is_f32 || dtype == UVC_F32, df = dst_is_f32 || dtype == UVC_F32;
  if (tap_major) {
    if (C != 64) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_fold_tokens: tap_major needs C == 64");
    constexpr int PPW = 8;
    const int gp = (B * H * W + 4 * PPW - 1) / (4 * PPW);
#define FOLD_TM(TS_, TD_) do { \
      if (k == 3 && s == 2 && p == 1) k_fold_tm<TS_, TD_, 3, 2, 1, PPW><<<gp, 256, 0, st>>>((const TS_*)src, lds, (TD_*)dst, B, H, W, k, s, p, Ho, Wo); \
      else k_fold_tm<TS_, TD_, 0, 0, 0, PPW><<<gp, 256, 0, st>>>((const TS_*)src, lds, (TD_*)dst, B, H, W, k, s, p, Ho, Wo); } while (0)
    if (sf && df) FOLD_TM(float, float);
    else if (sf) FOLD_TM(float, bf16_t);
    else if (df) FOLD_TM(bf16_t, float);
    else FOLD_TM(bf16_t, bf16_t);
#undef FOLD_TM
  } else if (sf && df) k_fold<float, float><<<grid, 256, 0, st>>>((const float*)src, lds, (float*)dst, B, C, H, W, k, s, p, Ho, Wo);
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
  k_performer_kv_mfma<<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->part, p->T, S, tps);
  UVC_CHECK_LAUNCH();
  k_part_reduce<<<dim3(ceil_div(PKV, 256), p->B), 256, 0, st>>>(p->part, p->kptv, S);
  UVC_CHECK_LAUNCH();
  if (p->att_is_f32 || p->dtype == UVC_F32) k_performer_q_mfma<float><<<p->B * ntile, 256, 0, st>>>(p->kqv, p->w, p->kptv, (float*)p->att, p->T, ntile);
  else k_performer_q_mfma<bf16_t><<<p->B * ntile, 256, 0, st>>>(p->kqv, p->w, p->kptv, (bf16_t*)p->att, p->T, ntile);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_performer_bwd(const uvc_performer_args* p, void* stream) {
  if (int e = check_perf(p)) return e;
  if (!p->datt || !p->dkqv || !p->dkptv) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_performer_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int ntile = (p->T + PT - 1) / PT, S = splits_of(p->B, p->T), tps = PTPS;
  const bool f32 = p->g_is_f32 || p->dtype == UVC_F32;
  if (f32) k_performer_bwd_q_mfma<float><<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->kptv, (const float*)p->datt, (float*)p->dkqv, p->part, p->T, S, tps);
  else k_performer_bwd_q_mfma<bf16_t><<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->kptv, (const bf16_t*)p->datt, (bf16_t*)p->dkqv, p->part, p->T, S, tps);
  UVC_CHECK_LAUNCH();
  k_part_reduce<<<dim3(ceil_div(PKV, 256), p->B), 256, 0, st>>>(p->part, p->dkptv, S);
  UVC_CHECK_LAUNCH();
  if (f32) k_performer_bwd_k_mfma<float><<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->dkptv, (const float*)p->dskip, (float*)p->dkqv, p->T, S, tps);
  else k_performer_bwd_k_mfma<bf16_t><<<p->B * S, 256, 0, st>>>(p->kqv, p->w, p->dkptv, (const bf16_t*)p->dskip, (bf16_t*)p->dkqv, p->T, S, tps);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// ------------------------------------------------------------------------------------------------
// One Token_performer stage sequenced in C (include/uvc_t2t.h: uvc_t2t_stage)
// ------------------------------------------------------------------------------------------------
namespace {
#define ST_TRY(x) do { if (int e_ = (x)) return e_; } while (0)
int st_nt(const uvc_t2t_stage* s, const void* A, const void* B, void* C, int c_f32, int N, int K, int epi, const float* bias, const void* R, int r_f32, int ldr,
          const void* aux, void* C2, void* stream) {
  uvc_gemm_nt_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.C = C; a.C2 = C2; a.bias = bias; a.R = R; a.aux = aux;
  a.alpha = 1.0f; a.M = s->B * s->T; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N; a.ldr = ldr ? ldr : N; a.ldaux = N;
  a.dtype = s->dtype; a.a_is_f32 = s->dtype == UVC_F32; a.c_is_f32 = c_f32 || s->dtype == UVC_F32; a.epilogue = epi; a.r_is_f32 = R ? (r_f32 || s->dtype == UVC_F32) : 0;
  return uvc_gemm_nt(&a, stream);
}
int st_tn(const uvc_t2t_stage* s, const void* A, const void* B, float* C, float* colsum, int N1, int N2, void* stream) {
  uvc_gemm_tn_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.B = B; a.C = C; a.workspace = s->tn_ws; a.workspace_bytes = s->tn_ws_bytes; a.colsum_out = colsum; a.alpha = 1.0f; a.beta = s->beta;
  a.M = s->B * s->T; a.N1 = N1; a.N2 = N2; a.lda = N1; a.ldb = N2; a.ldc = N2; a.dtype = s->dtype; a.a_is_f32 = s->dtype == UVC_F32;
  return uvc_gemm_tn(&a, stream);
}
void st_unfold(const uvc_t2t_stage* s, uvc_unfold_args& a) {
  memset(&a, 0, sizeof(a));
  a.src = s->src; a.sb = s->sb; a.sc = s->sc; a.sh = s->sh; a.sw = s->sw;
  a.B = s->B; a.C = s->C; a.H = s->H; a.W = s->W; a.k = s->k; a.s = s->s; a.p = s->p; a.ldo = s->dimp; a.dtype = s->dtype;
  a.gamma = s->norm1_w; a.mean = s->mean1; a.rstd = s->rstd1; a.eps = s->eps;
}
}  // namespace

extern "C" int uvc_t2t_stage_forward(const uvc_t2t_stage* s, void* stream) {
  if (!s || !s->src || !s->xn || !s->kqv || !s->att || !s->x1 || !s->h || !s->u || !s->out) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_t2t_stage_forward: null pointer");
  if (s->training && !s->gp) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_t2t_stage_forward: training needs gp");
  const int f32 = s->dtype == UVC_F32;
  {
    uvc_unfold_args a;
    st_unfold(s, a);
    a.beta = s->norm1_b; a.out = s->xn; a.out_is_f32 = f32;
    ST_TRY(uvc_unfold_ln_fwd(&a, stream));
  }
  ST_TRY(st_nt(s, s->xn, s->kqv_w, s->kqv, 1, 192, s->dimp, UVC_EPI_BIAS, s->kqv_b, nullptr, 0, 0, nullptr, nullptr, stream));
  {
    uvc_performer_args a;
    memset(&a, 0, sizeof(a));
    a.kqv = s->kqv; a.w = s->w; a.part = s->part; a.kptv = s->kptv; a.att = s->att; a.att_is_f32 = f32; a.B = s->B; a.T = s->T; a.dtype = s->dtype;
    ST_TRY(uvc_performer_fwd(&a, stream));
  }
  // y = v + proj(att): v is columns 128 .. 191 of kqv (token_performer.py:52)
  ST_TRY(st_nt(s, s->att, s->proj_w, s->x1, 1, 64, 64, UVC_EPI_BIAS_RESID, s->proj_b, s->kqv + 128, 1, 192, nullptr, nullptr, stream));
  {
    uvc_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = s->x1; a.gamma = s->norm2_w; a.beta = s->norm2_b; a.y = s->h; a.mean = s->mean2; a.rstd = s->rstd2; a.eps = s->eps;
    a.rows = s->B * s->T; a.D = 64; a.rows_per_group = 1; a.group_stride = 64; a.dtype = s->dtype; a.y_is_f32 = f32;
    ST_TRY(uvc_layernorm_fwd(&a, stream));
  }
  if (s->training) ST_TRY(st_nt(s, s->h, s->fc1_w, s->gp, 0, 64, 64, UVC_EPI_BIAS_GELU_GRAD, s->fc1_b, nullptr, 0, 0, nullptr, s->u, stream));   // C = GELU'(a), C2 = GELU(a)
  else ST_TRY(st_nt(s, s->h, s->fc1_w, s->u, 0, 64, 64, UVC_EPI_BIAS_GELU_OUT, s->fc1_b, nullptr, 0, 0, nullptr, nullptr, stream));
  return st_nt(s, s->u, s->fc2_w, s->out, 1, 64, 64, UVC_EPI_BIAS_RESID, s->fc2_b, s->x1, 1, 0, nullptr, nullptr, stream);
}

extern "C" int uvc_t2t_stage_backward(const uvc_t2t_stage* s, void* stream) {
  if (!s || !s->src || !s->dout || !s->da || !s->dh || !s->dx1 || !s->datt || !s->dkqv || !s->dkptv || !s->dxn || !s->tn_ws || !s->ln1_partial || !s->ln2_partial || !s->gp)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_t2t_stage_backward: null pointer");
  if (s->need_dx && !s->dxu) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_t2t_stage_backward: need_dx without dxu");
  const int f32 = s->dtype == UVC_F32;
  ST_TRY(st_tn(s, s->dout, s->u, s->g_fc2_w, s->g_fc2_b, 64, 64, stream));
  ST_TRY(st_nt(s, s->dout, s->fc2_wt, s->da, 0, 64, 64, UVC_EPI_MUL_AUX, nullptr, nullptr, 0, 0, s->gp, nullptr, stream));
  ST_TRY(st_tn(s, s->da, s->h, s->g_fc1_w, s->g_fc1_b, 64, 64, stream));
  ST_TRY(st_nt(s, s->da, s->fc1_wt, s->dh, 0, 64, 64, UVC_EPI_NONE, nullptr, nullptr, 0, 0, nullptr, nullptr, stream));
  {
    uvc_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = s->x1; a.gamma = s->norm2_w; a.mean = s->mean2; a.rstd = s->rstd2; a.dy = s->dh; a.dx = s->dx1; a.add1 = s->dout;
    a.partial = s->ln2_partial; a.dgamma = s->g_norm2_w; a.dbeta = s->g_norm2_b; a.eps = s->eps; a.beta_acc = s->beta;
    a.rows = s->B * s->T; a.D = 64; a.rows_per_group = 1; a.group_stride = 64; a.dtype = s->dtype; a.dy_is_f32 = f32; a.g_lowp = !f32;
    ST_TRY(uvc_layernorm_bwd(&a, stream));
  }
  ST_TRY(st_tn(s, s->dx1, s->att, s->g_proj_w, s->g_proj_b, 64, 64, stream));
  ST_TRY(st_nt(s, s->dx1, s->proj_wt, s->datt, 0, 64, 64, UVC_EPI_NONE, nullptr, nullptr, 0, 0, nullptr, nullptr, stream));
  {
    uvc_performer_args a;
    memset(&a, 0, sizeof(a));
    a.kqv = s->kqv; a.w = s->w; a.part = s->part; a.kptv = s->kptv; a.datt = s->datt; a.dskip = s->dx1; a.dkqv = s->dkqv; a.dkptv = s->dkptv; a.g_is_f32 = f32;
    a.B = s->B; a.T = s->T; a.dtype = s->dtype;
    ST_TRY(uvc_performer_bwd(&a, stream));
  }
  ST_TRY(st_tn(s, s->dkqv, s->xn, s->g_kqv_w, s->g_kqv_b, 192, s->dimp, stream));
  ST_TRY(st_nt(s, s->dkqv, s->kqv_wt, s->dxn, 0, s->dimp, 192, UVC_EPI_NONE, nullptr, nullptr, 0, 0, nullptr, nullptr, stream));
  uvc_unfold_args a;
  st_unfold(s, a);
  a.dy = s->dxn; a.dy_is_f32 = f32; a.partial = s->ln1_partial; a.dgamma = s->g_norm1_w; a.dbeta = s->g_norm1_b; a.beta_acc = s->beta;
  a.dxu = s->need_dx ? s->dxu : nullptr; a.dxu_tap_major = s->need_dx ? 1 : 0;
  return uvc_unfold_ln_bwd(&a, stream);
}
