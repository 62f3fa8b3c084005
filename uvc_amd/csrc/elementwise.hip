// Layout / elementwise / small-reduction kernels of the DeiT step on gfx950 (all HBM-bound).
// Reference lines are cited at each entry point.
#include "common.h"
#include "../../include/uvc_kernels.h"

namespace {

// ---------------------------------------------------------------------------- patchify
template <typename T>
__global__ __launch_bounds__(256) void k_patchify(const float* __restrict__ x, T* __restrict__ out, int B, int C, int S, int P) {
  const int G = S / P, K = C * P * P, K4 = K / 4, P4 = P / 4;
  const int64_t total = (int64_t)B * G * G * K4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int k4 = (int)(idx % K4);
    const int64_t r = idx / K4;
    const int px = (int)(r % G), py = (int)((r / G) % G), b = (int)(r / ((int64_t)G * G));
    const int kx4 = k4 % P4, ky = (k4 / P4) % P, c = k4 / (P4 * P);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * C + c) * S + (py * P + ky)) * S + px * P + kx4 * 4);
    T* o = out + (size_t)r * K + k4 * 4;
    if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(o) = v;
    else { u32x2 q; q[0] = pack_bf16x2(v[0], v[1]); q[1] = pack_bf16x2(v[2], v[3]); *reinterpret_cast<u32x2*>(o) = q; }
  }
}

// ---------------------------------------------------------------------------- token assembly
// TK: element type of the token rows (the residual stream): float32, or bf16 in the throughput mode
template <typename TK>
__global__ __launch_bounds__(256) void k_assemble(const float* __restrict__ pe, const float* __restrict__ cls,
                                                  const float* __restrict__ dist, const float* __restrict__ pos,
                                                  const float* __restrict__ mask, TK* __restrict__ tok, int B, int P, int D,
                                                  int ntok) {
  const int N = P + ntok, D4 = D / 4;
  const int64_t total = (int64_t)B * N * D4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d4 = (int)(idx % D4);
    const int t = (int)((idx / D4) % N), b = (int)(idx / ((int64_t)D4 * N));
    f32x4 v;
    if (t < ntok) v = *reinterpret_cast<const f32x4*>((t == 0 ? cls : dist) + d4 * 4);
    else {
      v = *reinterpret_cast<const f32x4*>(pe + ((size_t)b * P + (t - ntok)) * D + d4 * 4);
      if (mask) { const float m = mask[(size_t)b * P + (t - ntok)]; v[0] *= m; v[1] *= m; v[2] *= m; v[3] *= m; }
    }
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)t * D + d4 * 4);
    v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    TK* o = tok + ((size_t)b * N + t) * D + d4 * 4;
    if (sizeof(TK) == 4) *reinterpret_cast<f32x4*>(o) = v;
    else { u32x2 q; q[0] = pack_bf16x2(v[0], v[1]); q[1] = pack_bf16x2(v[2], v[3]); *reinterpret_cast<u32x2*>(o) = q; }
  }
}

// 4 consecutive elements of the token-gradient stream (float32, or bf16 when the backward keeps it in T)
template <typename TD> __device__ __forceinline__ f32x4 ld_tok4(const TD* p);
template <> __device__ __forceinline__ f32x4 ld_tok4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld_tok4<bf16_t>(const bf16_t* p) {
  const u32x2 r = *reinterpret_cast<const u32x2*>(p);
  return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
}

template <typename T, typename TD>
__global__ __launch_bounds__(256) void k_assemble_bwd_dpe(const TD* __restrict__ dtok, const float* __restrict__ mask,
                                                          T* __restrict__ dpe, int B, int P, int D, int ntok) {
  const int N = P + ntok, D4 = D / 4;
  const int64_t total = (int64_t)B * P * D4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d4 = (int)(idx % D4);
    const int i = (int)((idx / D4) % P), b = (int)(idx / ((int64_t)D4 * P));
    f32x4 v = ld_tok4<TD>(dtok + ((size_t)b * N + ntok + i) * D + d4 * 4);
    if (mask) { const float m = mask[(size_t)b * P + i]; v[0] *= m; v[1] *= m; v[2] *= m; v[3] *= m; }
    T* o = dpe + ((size_t)b * P + i) * D + d4 * 4;
    if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(o) = v;
    else { u32x2 q; q[0] = pack_bf16x2(v[0], v[1]); q[1] = pack_bf16x2(v[2], v[3]); *reinterpret_cast<u32x2*>(o) = q; }
  }
}

// dpos[t,d] = sum_b dtok[b,t,d]; the class / dist token gradients are rows 0 / 1 of the same sum.
// 256 threads = 32 column quads x 8 batch slices, 16-byte loads, fixed-order LDS reduction over the slices.
template <typename TD>
__global__ __launch_bounds__(256) void k_assemble_bwd_dpos(const TD* __restrict__ dtok, float* __restrict__ dpos,
                                                           float* __restrict__ dcls, float* __restrict__ ddist, int B, int N,
                                                           int D, int ntok, float beta) {
  __shared__ f32x4 red[8][32];
  const int tx = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int ND = N * D, col = (blockIdx.x * 32 + tx) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (col < ND) {
#pragma unroll 4
    for (int b = sl; b < B; b += 8) {
      const f32x4 v = ld_tok4<TD>(dtok + (size_t)b * ND + col);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
  }
  red[sl][tx] = s;
  __syncthreads();
  if (sl == 0 && col < ND) {
    f32x4 t = red[0][tx];
#pragma unroll
    for (int q = 1; q < 8; ++q) { const f32x4 v = red[q][tx]; t[0] += v[0]; t[1] += v[1]; t[2] += v[2]; t[3] += v[3]; }
    const int tok = col / D, d = col % D;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dpos[col + e] = (beta != 0.f ? beta * dpos[col + e] : 0.f) + t[e];
      if (tok == 0) dcls[d + e] = (beta != 0.f ? beta * dcls[d + e] : 0.f) + t[e];
      if (tok == 1 && ntok == 2 && ddist) ddist[d + e] = (beta != 0.f ? beta * ddist[d + e] : 0.f) + t[e];
    }
  }
}

// dmask[b,i] = <dtok[b, ntok+i, :], pe[b, i, :]>  (one wave per row)
template <typename TD>
__global__ __launch_bounds__(256) void k_assemble_bwd_dmask(const TD* __restrict__ dtok, const float* __restrict__ pe,
                                                            float* __restrict__ dmask, int B, int P, int D, int ntok) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * P) return;
  const int b = row / P, i = row % P, N = P + ntok;
  const TD* a = dtok + ((size_t)b * N + ntok + i) * D;
  const float* c = pe + (size_t)row * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += ElemIO<TD>::load(a + d) * c[d];
  s = wave_sum(s);
  if (lane == 0) dmask[row] = s;
}

// ---------------------------------------------------------------------------- column sums
constexpr int CS_ROWS = 256;
template <typename T, int VEC>
__global__ __launch_bounds__(256) void k_colsum(const T* __restrict__ X, int M, int N, int ldx, float* __restrict__ partial, const float* __restrict__ rw) {
  __shared__ float red[4][64 * VEC];
  const int c0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * VEC, sl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  if (c0 < N) {
    for (int r = r0 + sl; r < r1; r += 4) {
      const T* p = X + (size_t)r * ldx + c0;
      const float wr = rw ? rw[r] : 1.0f;
      if (VEC == 4) {
        if (sizeof(T) == 4) { const f32x4 v = *reinterpret_cast<const f32x4*>(p); s[0] += wr * v[0]; s[1 % VEC] += wr * v[1]; s[2 % VEC] += wr * v[2]; s[3 % VEC] += wr * v[3]; }
        else {
          const u32x2 q = *reinterpret_cast<const u32x2*>(p);
          s[0] += wr * __uint_as_float(q[0] << 16); s[1 % VEC] += wr * __uint_as_float(q[0] & 0xffff0000u);
          s[2 % VEC] += wr * __uint_as_float(q[1] << 16); s[3 % VEC] += wr * __uint_as_float(q[1] & 0xffff0000u);
        }
      } else s[0] += wr * ElemIO<T>::load(p);
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[sl][(threadIdx.x & 63) * VEC + e] = s[e];
  __syncthreads();
  if (sl == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int k = (threadIdx.x & 63) * VEC + e;
      if (c0 + e < N) partial[(size_t)blockIdx.y * N + c0 + e] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    }
  }
}
__global__ __launch_bounds__(256) void k_colsum_reduce(const float* __restrict__ partial, int nb, int N, float* __restrict__ out,
                                                       float alpha, const float* alpha_ptr, float beta) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < N)
    for (int b = sl; b < nb; b += 4) s += partial[(size_t)b * N + c];
  red[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < N) {
    const int t = threadIdx.x;
    if (alpha_ptr) alpha *= *alpha_ptr;
    out[c] = (beta != 0.f ? beta * out[c] : 0.f) + alpha * (((red[0][t] + red[1][t]) + red[2][t]) + red[3][t]);
  }
}

// ---------------------------------------------------------------------------- weight shadows
// 32x32 tiles through LDS: dst[r,c] = cast(src[r,c]); dstT[c,r] = cast(src[r,c])
template <typename T>
__global__ __launch_bounds__(256) void k_cast_transpose(const float* __restrict__ W, int R, int C, T* __restrict__ w, T* __restrict__ wt) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < R && c < C) ? W[(size_t)r * C + c] : 0.f;
    tile[i][tx] = v;
    if (w && r < R && c < C) ElemIO<T>::store(w + (size_t)r * C + c, v);
  }
  __syncthreads();
  if (wt)
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (r < R && c < C) ElemIO<T>::store(wt + (size_t)c * R + r, tile[tx][i]);
    }
}

// ---------------------------------------------------------------------------- gates
// mode 0: warm-up (.5,.5); 1: soft Gumbel-softmax tau=.5; 2: softL0 g1^2/(g1^2+eps); 3: hard Gumbel (one-hot)
__global__ void k_gate_distrib(const float* g, const float* e, float* d, int L, int mode, float eps) {
  const int l = threadIdx.x;
  if (l >= L) return;
  float d0 = 0.5f, d1 = 0.5f;
  if (mode == 1 || mode == 3) {
    const float u0 = (g[2 * l] + (-__logf(e[2 * l]))) / 0.5f, u1 = (g[2 * l + 1] + (-__logf(e[2 * l + 1]))) / 0.5f;
    const float m = fmaxf(u0, u1);
    const float e0 = __expf(u0 - m), e1 = __expf(u1 - m);
    d0 = e0 / (e0 + e1); d1 = e1 / (e0 + e1);
    if (mode == 3) { const bool one = d1 > d0; d0 = one ? 0.f : 1.f; d1 = one ? 1.f : 0.f; }
  } else if (mode == 2) {
    const float t = g[2 * l + 1] * g[2 * l + 1];
    d1 = t / (t + eps); d0 = 1.0f - d1;
  }
  d[2 * l] = d0; d[2 * l + 1] = d1;
}
// out = d1*x2 + d0*x  =>  dL/dd1 - dL/dd0 = <gA, x2 - x> = (<gA,out> - <gA,x>) / d1 = (A - B) / d1.
__global__ void k_gate_grad(const float* g, const float* d, const float* dots, float* dg, int L, int mode, float eps, float beta) {
  const int l = threadIdx.x;
  if (l >= L) return;
  // raw layout written by the LayerNorm-backward kernels: row l = block l's norm1 backward
  // { <dL/dx_l, x_l>, <gA_l, x_l> }, row L = final norm.  A_l = <gA_l, x_{l+1}> = row l+1, col 0.
  const float A = dots[2 * (l + 1)], Bv = dots[2 * l + 1];
  float g0 = 0.f, g1 = 0.f;
  if (mode == 1) {               // softmax((g+G)/tau): dg1 = d0*d1/tau * <gA, x2-x> = d0/tau * (A - B)
    g1 = d[2 * l] / 0.5f * (A - Bv);
    g0 = -g1;
  } else if (mode == 2) {        // d1 = t/(t+eps), d0 = 1-d1: dg1 = 2 g1 eps/(t+eps)^2 * (A-B)/d1 = 2 eps/(g1 (t+eps)) (A-B)
    const float x = g[2 * l + 1], t = x * x;
    g1 = x != 0.f ? 2.0f * eps / (x * (t + eps)) * (A - Bv) : 0.f;
  }
  dg[2 * l] = (beta != 0.f ? beta * dg[2 * l] : 0.f) + g0;
  dg[2 * l + 1] = (beta != 0.f ? beta * dg[2 * l + 1] : 0.f) + g1;
}


// ---------------------------------------------------------------------------- patch gating
// mode 1 (model_distilled.py:434-444): mask[b,i] = sigmoid(pg[i]) (soft) or [sigmoid >= .5] with token 0 kept (hard)
__global__ __launch_bounds__(256) void k_patch_sigmoid(const float* __restrict__ pg, float* __restrict__ mask, int B, int P, int hard) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * P) return;
  const int t = i % P;
  const float sg = 1.0f / (1.0f + __expf(-pg[t]));
  mask[i] = hard ? ((sg >= 0.5f || t == 0) ? 1.0f : 0.0f) : sg;
}
__global__ __launch_bounds__(256) void k_patch_sigmoid_bwd(const float* __restrict__ pg, const float* __restrict__ dmask,
                                                           float* __restrict__ dpg, int B, int P, float beta) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= P) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dmask[(size_t)b * P + t];
  const float sg = 1.0f / (1.0f + __expf(-pg[t]));
  dpg[t] = (beta != 0.f ? beta * dpg[t] : 0.f) + s * sg * (1.0f - sg);
}
// mode 2 scorer: scores[row] = <pe[row,:], w> + bias   (self.gumbel = Linear(D,1), :450)
__global__ __launch_bounds__(256) void k_patch_scores(const float* __restrict__ pe, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ scores, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = pe + (size_t)row * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += p[d] * w[d];
  s = wave_sum(s);
  if (lane == 0) scores[row] = s + bias[0];
}
__device__ __forceinline__ float block_sum256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float block_max256(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
// custom gumbel_softmax(log_softmax(scores), k, tau, hard=True) + token 0 forced to 1 (:36-63,446-456).
// One block per image, P <= 256.  Outputs the straight-through mask, y_soft and softmax(scores) for backward.
// The reference takes topk of y_soft: every step that decides an index (log_softmax, -log E, the division by tau, the
// softmax of u) uses the correctly-rounded-to-1-ulp expf / logf, in the reference's operation order, NOT the fast
// __expf / __logf (whose x*log2(e) pre-multiply alone is off by |x| * 6e-8); ties in y go to the lower index.
__global__ __launch_bounds__(256) void k_patch_topk(const float* __restrict__ scores, const float* __restrict__ e, float* __restrict__ mask,
                                                    float* __restrict__ ysoft, float* __restrict__ psoft, int P, int k, float tau) {
  __shared__ float sh[4];
  __shared__ float yv[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool ok = t < P;
  const float s = ok ? scores[(size_t)b * P + t] : -INFINITY;
  const float m1 = block_max256(s, sh);
  const float z1 = block_sum256(ok ? expf(s - m1) : 0.f, sh);
  const float logp = (s - m1) - logf(z1);
  const float u = ok ? (logp + (-logf(e[(size_t)b * P + t]))) / tau : -INFINITY;
  const float m2 = block_max256(u, sh);
  const float ex = ok ? expf(u - m2) : 0.f;
  const float z2 = block_sum256(ex, sh);
  const float y = ex / z2;
  yv[t] = ok ? y : -1.0f;
  __syncthreads();
  if (ok) {
    int rank = 0;                                   // descending order, ties -> lower index first
    for (int j = 0; j < P; ++j) { const float o = yv[j]; rank += (o > y) || (o == y && j < t); }
    const float hard = rank < k ? 1.0f : 0.0f;
    mask[(size_t)b * P + t] = t == 0 ? 1.0f : (hard - y) + y;
    ysoft[(size_t)b * P + t] = y;
    psoft[(size_t)b * P + t] = expf(logp);
  }
}
// backward of the above w.r.t. the scores (straight-through: dmask -> dy, token 0 has no gradient)
__global__ __launch_bounds__(256) void k_patch_topk_bwd(const float* __restrict__ dmask, const float* __restrict__ ysoft,
                                                        const float* __restrict__ psoft, float* __restrict__ dscores, int P, float tau) {
  __shared__ float sh[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const bool ok = t < P;
  const float y = ok ? ysoft[(size_t)b * P + t] : 0.f;
  const float dy = (ok && t != 0) ? dmask[(size_t)b * P + t] : 0.f;
  const float ydy = block_sum256(y * dy, sh);
  const float dlogp = ok ? y * (dy - ydy) / tau : 0.f;
  const float sdl = block_sum256(dlogp, sh);
  if (ok) dscores[(size_t)b * P + t] = dlogp - psoft[(size_t)b * P + t] * sdl;
}
// X[row, :] += rw[row] * w[:]   (scorer's contribution to d(patch embedding))
template <typename T>
__global__ __launch_bounds__(256) void k_add_outer(T* __restrict__ X, const float* __restrict__ rw, const float* __restrict__ w, int rows, int D) {
  const int64_t total = (int64_t)rows * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / D), d = (int)(i % D);
    ElemIO<T>::store(X + i, ElemIO<T>::load(X + i) + rw[r] * w[d]);
  }
}

// Stage-2 `m.weight.data *= m.mask` over the whole flat parameter buffer (post_train.py:343-346)
__global__ __launch_bounds__(256) void k_apply_masks(float* __restrict__ p, const float* __restrict__ mask, int64_t n) {
  const int64_t n4 = n >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  const f32x4* m4 = reinterpret_cast<const f32x4*>(mask);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 m = m4[i];
    if (m[0] == 1.0f && m[1] == 1.0f && m[2] == 1.0f && m[3] == 1.0f) continue;      // unmasked: no write
    f32x4 v = p4[i];
    v[0] *= m[0]; v[1] *= m[1]; v[2] *= m[2]; v[3] *= m[3];
    p4[i] = v;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) p[i] *= mask[i];
}

// ---------------------------------------------------------------------------- Mixup / CutMix (timm.data.Mixup, mode "batch")
// image b is mixed with image B-1-b (x.flip(0)); a thread owns the same 4 pixels of both images of a pair, so the in-place
// update needs no temporary.  Products and sum are rounded separately, like x.mul_(lam).add_(x.flip(0).mul_(1 - lam)).
__global__ __launch_bounds__(256) void k_mixup(float* __restrict__ x, int B, int64_t chw4, float lam, float oml) {
  // three separately rounded operations: hipcc contracts a*b + c*d (also through __fmul_rn / __fadd_rn and under a contract(off)
  // pragma) into v_fmac; fma(a, b, 0) is the correctly rounded product and cannot be fused with the following add
  const int64_t total = (int64_t)(B / 2) * chw4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int b = (int)(idx / chw4);
    const int64_t i = idx % chw4;
    f32x4* pa = reinterpret_cast<f32x4*>(x) + (int64_t)b * chw4 + i;
    f32x4* pc = reinterpret_cast<f32x4*>(x) + (int64_t)(B - 1 - b) * chw4 + i;
    const f32x4 a = *pa, c = *pc;
    f32x4 na, nc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      na[e] = __builtin_fmaf(a[e], lam, 0.0f) + __builtin_fmaf(c[e], oml, 0.0f);
      nc[e] = __builtin_fmaf(c[e], lam, 0.0f) + __builtin_fmaf(a[e], oml, 0.0f);
    }
    *pa = na; *pc = nc;
  }
}
// x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]: the box is swapped inside each pair
__global__ __launch_bounds__(256) void k_cutmix(float* __restrict__ x, int B, int C, int H, int W, int yl, int yh, int xl, int xh) {
  const int bw = xh - xl, bh = yh - yl;
  const int64_t total = (int64_t)(B / 2) * C * bh * bw;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int px = (int)(idx % bw), py = (int)((idx / bw) % bh), c = (int)((idx / ((int64_t)bw * bh)) % C), b = (int)(idx / ((int64_t)bw * bh * C));
    const int64_t o = ((int64_t)c * H + (yl + py)) * W + xl + px;
    float* pa = x + (int64_t)b * C * H * W + o;
    float* pc = x + (int64_t)(B - 1 - b) * C * H * W + o;
    const float a = *pa, cc = *pc;
    *pa = cc; *pc = a;
  }
}
// mixup_target: y = onehot_smooth(t) * lam + onehot_smooth(t.flip(0)) * (1 - lam)
__global__ __launch_bounds__(256) void k_mixup_target(const int64_t* __restrict__ t, float* __restrict__ y, int B, int C, float lam, float oml,
                                                      float on, float off) {
  const int64_t total = (int64_t)B * C;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int b = (int)(idx / C), c = (int)(idx % C);
    const float v1 = t[b] == c ? on : off, v2 = t[B - 1 - b] == c ? on : off;
    y[idx] = __builtin_fmaf(v1, lam, 0.0f) + __builtin_fmaf(v2, oml, 0.0f);
  }
}

// ---------------------------------------------------------------------------- Stage-2 MLP compaction
template <typename T>
__global__ __launch_bounds__(256) void k_mlp_gather(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
                                                    const int* __restrict__ idx, int D, int F, int Fe, T* __restrict__ w1c, T* __restrict__ w1t,
                                                    T* __restrict__ w2c, T* __restrict__ w2t, float* __restrict__ b1c) {
  const int64_t total = (int64_t)Fe * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int s = (int)(i / D), d = (int)(i % D), j = idx[s];
    const float a = W1[(size_t)j * D + d], b = W2[(size_t)d * F + j];
    ElemIO<T>::store(w1c + (size_t)s * D + d, a);
    ElemIO<T>::store(w1t + (size_t)d * Fe + s, a);
    ElemIO<T>::store(w2c + (size_t)d * Fe + s, b);
    ElemIO<T>::store(w2t + (size_t)s * D + d, b);
    if (d == 0) b1c[s] = b1[j];
  }
}
template <bool LOWP>
__global__ __launch_bounds__(256) void k_mlp_scatter(const float* __restrict__ dw1c, const float* __restrict__ dw2c, const float* __restrict__ db1c,
                                                     const int* __restrict__ inv, const float* __restrict__ b1, const float* __restrict__ db2, int D,
                                                     int F, int Fe, float* __restrict__ dW1, float* __restrict__ dW2, float* __restrict__ db1,
                                                     float beta) {
  const int64_t total = (int64_t)F * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    {   // dW1 [F, D], row-major walk
      const int j = (int)(i / D), d = (int)(i % D), sl = inv[j];
      const float v = sl >= 0 ? dw1c[(size_t)sl * D + d] : 0.f;
      dW1[i] = (beta != 0.f ? beta * dW1[i] : 0.f) + v;
      if (d == 0) db1[j] = (beta != 0.f ? beta * db1[j] : 0.f) + (sl >= 0 ? db1c[sl] : 0.f);
    }
    {   // dW2 [D, F]
      const int d = (int)(i / F), j = (int)(i % F), sl = inv[j];
      float v;
      if (sl >= 0) v = dw2c[(size_t)d * Fe + sl];
      else {
        float u = LOWP ? gelu_fast(b1[j]) : gelu_f(b1[j]);
        if (LOWP) u = bf16_to_f32(f32_to_bf16(u));        // the forward stores GELU(a) in bf16
        v = u * db2[d];
      }
      dW2[i] = (beta != 0.f ? beta * dW2[i] : 0.f) + v;
    }
  }
}

inline int grid_for(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int uvc_patchify(const float* x, void* out, int32_t B, int32_t C, int32_t S, int32_t P, int32_t dtype, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || S <= 0 || P <= 0 || S % P || P % 4) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_patchify: bad argument");
  const int64_t total = (int64_t)B * (S / P) * (S / P) * C * P * P / 4;
  if (dtype == UVC_F32) k_patchify<float><<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, (float*)out, B, C, S, P);
  else k_patchify<bf16_t><<<grid_for(total), 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)out, B, C, S, P);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_assemble_tokens(const float* pe, const float* cls, const float* dist, const float* pos, const float* row_mask,
                                   void* tok, int32_t B, int32_t P, int32_t D, int32_t ntok, int32_t tok_lowp, void* stream) {
  if (!pe || !cls || !pos || !tok || (ntok == 2 && !dist) || (ntok != 1 && ntok != 2) || D % 4) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_assemble_tokens: bad argument");
  const int grid = grid_for((int64_t)B * (P + ntok) * D / 4);
  if (tok_lowp) k_assemble<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(pe, cls, dist, pos, row_mask, (bf16_t*)tok, B, P, D, ntok);
  else k_assemble<float><<<grid, 256, 0, (hipStream_t)stream>>>(pe, cls, dist, pos, row_mask, (float*)tok, B, P, D, ntok);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_assemble_tokens_bwd(const void* dtok, const float* pe, const float* row_mask, void* dpe, float* dpos, float* dcls,
                                       float* ddist, float* dmask, int32_t B, int32_t P, int32_t D, int32_t ntok, int32_t dtype,
                                       int32_t dpe_is_f32, int32_t dtok_lowp, float beta_acc, void* stream) {
  if (!dtok || !dpe || !dpos || !dcls || D % 4 || (ntok != 1 && ntok != 2)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_assemble_tokens_bwd: bad argument");
  if (dmask && !pe) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_assemble_tokens_bwd: dmask needs pe");
  hipStream_t st = (hipStream_t)stream;
  const bool out32 = dtype == UVC_F32 || dpe_is_f32, in16 = dtok_lowp && dtype == UVC_BF16;
  const int g1 = grid_for((int64_t)B * P * D / 4), g2 = ceil_div((P + ntok) * D, 128), g3 = ceil_div(B * P, 4);
  const int N = P + ntok;
  if (in16) {
    const bf16_t* t = (const bf16_t*)dtok;
    if (out32) k_assemble_bwd_dpe<float, bf16_t><<<g1, 256, 0, st>>>(t, row_mask, (float*)dpe, B, P, D, ntok);
    else k_assemble_bwd_dpe<bf16_t, bf16_t><<<g1, 256, 0, st>>>(t, row_mask, (bf16_t*)dpe, B, P, D, ntok);
    UVC_CHECK_LAUNCH();
    k_assemble_bwd_dpos<bf16_t><<<g2, 256, 0, st>>>(t, dpos, dcls, ddist, B, N, D, ntok, beta_acc);
    UVC_CHECK_LAUNCH();
    if (dmask) { k_assemble_bwd_dmask<bf16_t><<<g3, 256, 0, st>>>(t, pe, dmask, B, P, D, ntok); UVC_CHECK_LAUNCH(); }
  } else {
    const float* t = (const float*)dtok;
    if (out32) k_assemble_bwd_dpe<float, float><<<g1, 256, 0, st>>>(t, row_mask, (float*)dpe, B, P, D, ntok);
    else k_assemble_bwd_dpe<bf16_t, float><<<g1, 256, 0, st>>>(t, row_mask, (bf16_t*)dpe, B, P, D, ntok);
    UVC_CHECK_LAUNCH();
    k_assemble_bwd_dpos<float><<<g2, 256, 0, st>>>(t, dpos, dcls, ddist, B, N, D, ntok, beta_acc);
    UVC_CHECK_LAUNCH();
    if (dmask) { k_assemble_bwd_dmask<float><<<g3, 256, 0, st>>>(t, pe, dmask, B, P, D, ntok); UVC_CHECK_LAUNCH(); }
  }
  return UVC_OK;
}

extern "C" int uvc_colsum_blocks(int32_t M) { return ceil_div(M, CS_ROWS); }

extern "C" int uvc_colsum(const void* X, int32_t M, int32_t N, int32_t ldx, int32_t dtype, int32_t x_is_f32, float* partial, float* out,
                          float alpha, const float* alpha_ptr, float beta, const float* row_weight, void* stream) {
  if (!X || !partial || !out || M <= 0 || N <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_colsum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nb = ceil_div(M, CS_ROWS);
  const bool f32 = (dtype == UVC_F32) || x_is_f32;
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0);
  if (vec) {
    dim3 grid(ceil_div(N, 256), nb);
    if (f32) k_colsum<float, 4><<<grid, 256, 0, st>>>((const float*)X, M, N, ldx, partial, row_weight);
    else k_colsum<bf16_t, 4><<<grid, 256, 0, st>>>((const bf16_t*)X, M, N, ldx, partial, row_weight);
  } else {
    dim3 grid(ceil_div(N, 64), nb);
    if (f32) k_colsum<float, 1><<<grid, 256, 0, st>>>((const float*)X, M, N, ldx, partial, row_weight);
    else k_colsum<bf16_t, 1><<<grid, 256, 0, st>>>((const bf16_t*)X, M, N, ldx, partial, row_weight);
  }
  UVC_CHECK_LAUNCH();
  k_colsum_reduce<<<ceil_div(N, 64), 256, 0, st>>>(partial, nb, N, out, alpha, alpha_ptr, beta);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_patch_gate_sigmoid(const float* pg, float* mask, int32_t B, int32_t P, int32_t hard, void* stream) {
  if (!pg || !mask || B <= 0 || P <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_patch_gate_sigmoid: bad argument");
  k_patch_sigmoid<<<ceil_div(B * P, 256), 256, 0, (hipStream_t)stream>>>(pg, mask, B, P, hard);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
extern "C" int uvc_patch_gate_sigmoid_bwd(const float* pg, const float* dmask, float* dpg, int32_t B, int32_t P, float beta_acc, void* stream) {
  if (!pg || !dmask || !dpg || B <= 0 || P <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_patch_gate_sigmoid_bwd: bad argument");
  k_patch_sigmoid_bwd<<<ceil_div(P, 256), 256, 0, (hipStream_t)stream>>>(pg, dmask, dpg, B, P, beta_acc);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
extern "C" int uvc_patch_scores(const float* pe, const float* w, const float* bias, float* scores, int32_t rows, int32_t D, void* stream) {
  if (!pe || !w || !bias || !scores || rows <= 0 || D <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_patch_scores: bad argument");
  k_patch_scores<<<ceil_div(rows, 4), 256, 0, (hipStream_t)stream>>>(pe, w, bias, scores, rows, D);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
extern "C" int uvc_patch_topk_mask(const float* scores, const float* e, float* mask, float* ysoft, float* psoft, int32_t B, int32_t P, int32_t k,
                                   float tau, void* stream) {
  if (!scores || !e || !mask || !ysoft || !psoft || B <= 0 || P <= 0 || P > 256 || k < 0 || tau <= 0.f)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_patch_topk_mask: bad argument (P <= 256, tau > 0)");
  k_patch_topk<<<B, 256, 0, (hipStream_t)stream>>>(scores, e, mask, ysoft, psoft, P, k, tau);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
extern "C" int uvc_patch_topk_mask_bwd(const float* dmask, const float* ysoft, const float* psoft, float* dscores, int32_t B, int32_t P, float tau,
                                       void* stream) {
  if (!dmask || !ysoft || !psoft || !dscores || B <= 0 || P <= 0 || P > 256 || tau <= 0.f) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_patch_topk_mask_bwd: bad argument");
  k_patch_topk_bwd<<<B, 256, 0, (hipStream_t)stream>>>(dmask, ysoft, psoft, dscores, P, tau);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
extern "C" int uvc_add_outer(void* X, const float* row_weight, const float* w, int32_t rows, int32_t D, int32_t dtype, int32_t x_is_f32, void* stream) {
  if (!X || !row_weight || !w || rows <= 0 || D <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_add_outer: bad argument");
  if (dtype == UVC_F32 || x_is_f32) k_add_outer<float><<<grid_for((int64_t)rows * D), 256, 0, (hipStream_t)stream>>>((float*)X, row_weight, w, rows, D);
  else k_add_outer<bf16_t><<<grid_for((int64_t)rows * D), 256, 0, (hipStream_t)stream>>>((bf16_t*)X, row_weight, w, rows, D);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_mixup_batch(float* x, int32_t B, int32_t C, int32_t H, int32_t W, float lam, float one_minus_lam, int32_t use_cutmix,
                               int32_t yl, int32_t yh, int32_t xl, int32_t xh, void* stream) {
  if (!x || B <= 0 || (B & 1) || C <= 0 || H <= 0 || W <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mixup_batch: needs an even batch (timm asserts the same)");
  hipStream_t st = (hipStream_t)stream;
  if (use_cutmix) {
    if (yl < 0 || xl < 0 || yh > H || xh > W || yl > yh || xl > xh) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mixup_batch: bad box");
    if (yh == yl || xh == xl) return UVC_OK;
    k_cutmix<<<grid_for((int64_t)(B / 2) * C * (yh - yl) * (xh - xl)), 256, 0, st>>>(x, B, C, H, W, yl, yh, xl, xh);
  } else {
    const int64_t chw = (int64_t)C * H * W;
    if (chw % 4 || ((uintptr_t)x & 15)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mixup_batch: image size must be a multiple of 4 floats, 16-byte aligned");
    k_mixup<<<grid_for((int64_t)(B / 2) * (chw / 4)), 256, 0, st>>>(x, B, chw / 4, lam, one_minus_lam);
  }
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_mixup_target(const int64_t* labels, float* y, int32_t B, int32_t C, float lam, float one_minus_lam, float on_value,
                                float off_value, void* stream) {
  if (!labels || !y || B <= 0 || C <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mixup_target: bad argument");
  k_mixup_target<<<grid_for((int64_t)B * C), 256, 0, (hipStream_t)stream>>>(labels, y, B, C, lam, one_minus_lam, on_value, off_value);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_mlp_gather_shadows(const float* W1, const float* b1, const float* W2, const int32_t* idx, int32_t D, int32_t F, int32_t width,
                                      void* w1c, void* w1t, void* w2c, void* w2t, float* b1c, int32_t dtype, void* stream) {
  if (!W1 || !b1 || !W2 || !idx || !w1c || !w1t || !w2c || !w2t || !b1c || D <= 0 || F <= 0 || width <= 0 || width > F)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_gather_shadows: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UVC_F32) k_mlp_gather<float><<<grid_for((int64_t)width * D), 256, 0, st>>>(W1, b1, W2, idx, D, F, width, (float*)w1c, (float*)w1t, (float*)w2c, (float*)w2t, b1c);
  else k_mlp_gather<bf16_t><<<grid_for((int64_t)width * D), 256, 0, st>>>(W1, b1, W2, idx, D, F, width, (bf16_t*)w1c, (bf16_t*)w1t, (bf16_t*)w2c, (bf16_t*)w2t, b1c);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_mlp_scatter_grads(const float* dw1c, const float* dw2c, const float* db1c, const int32_t* inv, const float* b1, const float* db2,
                                     int32_t D, int32_t F, int32_t width, float* dW1, float* dW2, float* db1, float beta_acc, int32_t dtype, void* stream) {
  if (!dw1c || !dw2c || !db1c || !inv || !b1 || !db2 || !dW1 || !dW2 || !db1 || D <= 0 || F <= 0 || width <= 0)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_scatter_grads: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == UVC_F32) k_mlp_scatter<false><<<grid_for((int64_t)F * D), 256, 0, st>>>(dw1c, dw2c, db1c, inv, b1, db2, D, F, width, dW1, dW2, db1, beta_acc);
  else k_mlp_scatter<true><<<grid_for((int64_t)F * D), 256, 0, st>>>(dw1c, dw2c, db1c, inv, b1, db2, D, F, width, dW1, dW2, db1, beta_acc);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_apply_masks(float* params, const float* mask, int64_t n, void* stream) {
  if (!params || !mask || n <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_apply_masks: bad argument");
  if ((((uintptr_t)params | (uintptr_t)mask) & 15) != 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_apply_masks: buffers must be 16-byte aligned");
  k_apply_masks<<<grid_for(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(params, mask, n);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_cast_transpose(const float* W, int32_t R, int32_t C, void* w_cast, void* wt, int32_t dtype, void* stream) {
  if (!W || R <= 0 || C <= 0 || (!w_cast && !wt)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_cast_transpose: bad argument");
  dim3 grid(ceil_div(C, 32), ceil_div(R, 32));
  if (dtype == UVC_F32) k_cast_transpose<float><<<grid, 256, 0, (hipStream_t)stream>>>(W, R, C, (float*)w_cast, (float*)wt);
  else k_cast_transpose<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(W, R, C, (bf16_t*)w_cast, (bf16_t*)wt);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// all weight shadows of a model in ONE launch: table of up to 64 matrices passed by value.  A workgroup owns a 64 x 64 tile (16-byte loads,
// 8-byte bf16 stores in both orientations); with 32 x 32 tiles the launch was 5.4 k (DeiT-Tiny) ... 84 k (DeiT-Base) workgroups of 4 KB
// each and ran at the dispatch rate: 94 us for 33 MB of traffic.
struct CtTable { int n; int tile0[65]; int R[64]; int C[64]; long long src[64]; long long w[64]; long long wt[64]; };
constexpr int CT_T = 64;
template <typename T>
__global__ __launch_bounds__(256) void k_cast_transpose_multi(const float* __restrict__ base, T* __restrict__ sh, CtTable t) {
  __shared__ float tile[CT_T][CT_T + 1];
  // which matrix this tile belongs to: lane l compares against tile0[l + 1] (n <= 64: one wave-wide compare + ballot instead of a chain of up to 64 dependent
  // scalar loads from the 2.4-KB argument table -- r5: the launch took 61 us for 44 MB, almost all of it this loop)
  const int lane = threadIdx.x & 63;
  const unsigned long long ge = __ballot(lane + 1 < t.n && (int)blockIdx.x >= t.tile0[lane + 1 < 65 ? lane + 1 : 64]);
  const int m = __popcll(ge);
  const int R = t.R[m], C = t.C[m];
  const int lt = blockIdx.x - t.tile0[m], tc = (C + CT_T - 1) / CT_T;
  const int r0 = (lt / tc) * CT_T, c0 = (lt % tc) * CT_T;
  const float* W = base + t.src[m];
  T* w = t.w[m] >= 0 ? sh + t.w[m] : nullptr;
  T* wt = t.wt[m] >= 0 ? sh + t.wt[m] : nullptr;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;           // 16 threads x 4 columns, 16 rows at a time
  // vector path: every 4-element group of the tile is whole and 16-byte aligned in the source and in both shadows
  const bool vec = (C % 4 == 0) && (R % 4 == 0) && ((t.src[m] | (t.w[m] >= 0 ? t.w[m] : 0) | (t.wt[m] >= 0 ? t.wt[m] : 0)) % 4 == 0);
  if (vec) {
#pragma unroll
    for (int i = 0; i < CT_T; i += 16) {
      const int r = r0 + i + ty, c = c0 + tx * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < R && c < C) v = *reinterpret_cast<const f32x4*>(W + (size_t)r * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[i + ty][tx * 4 + e] = v[e];
      if (w && r < R && c < C) {
        if constexpr (sizeof(T) == 2) { u32x2 q; q[0] = pack_bf16x2(v[0], v[1]); q[1] = pack_bf16x2(v[2], v[3]); *reinterpret_cast<u32x2*>(w + (size_t)r * C + c) = q; }
        else *reinterpret_cast<f32x4*>(w + (size_t)r * C + c) = v;
      }
    }
    __syncthreads();
    if (wt) {
#pragma unroll
      for (int i = 0; i < CT_T; i += 16) {
        const int c = c0 + i + ty, r = r0 + tx * 4;                   // a row of wt = a column of W
        if (c < C && r < R) {
          const f32x4 v = {tile[tx * 4][i + ty], tile[tx * 4 + 1][i + ty], tile[tx * 4 + 2][i + ty], tile[tx * 4 + 3][i + ty]};
          if constexpr (sizeof(T) == 2) { u32x2 q; q[0] = pack_bf16x2(v[0], v[1]); q[1] = pack_bf16x2(v[2], v[3]); *reinterpret_cast<u32x2*>(wt + (size_t)c * R + r) = q; }
          else *reinterpret_cast<f32x4*>(wt + (size_t)c * R + r) = v;
        }
      }
    }
    return;
  }
  const int sx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  for (int i = sy; i < CT_T; i += 4) {
    const int r = r0 + i, c = c0 + sx;
    const float v = (r < R && c < C) ? W[(size_t)r * C + c] : 0.f;
    tile[i][sx] = v;
    if (w && r < R && c < C) ElemIO<T>::store(w + (size_t)r * C + c, v);
  }
  __syncthreads();
  if (wt)
    for (int i = sy; i < CT_T; i += 4) {
      const int c = c0 + i, r = r0 + sx;
      if (r < R && c < C) ElemIO<T>::store(wt + (size_t)c * R + r, tile[sx][i]);
    }
}
// srcs/ws/wts are element offsets (params / shadow buffer, -1 = skip); n <= 64
extern "C" int uvc_cast_transpose_multi(const float* params, void* shadow, int32_t n, const int64_t* srcs, const int32_t* Rs, const int32_t* Cs,
                                        const int64_t* ws, const int64_t* wts, int32_t dtype, void* stream) {
  if (!params || !shadow || n <= 0 || n > 64 || !srcs || !Rs || !Cs || !ws || !wts) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_cast_transpose_multi: bad argument");
  CtTable t;
  t.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    t.tile0[i] = tiles; t.R[i] = Rs[i]; t.C[i] = Cs[i]; t.src[i] = srcs[i]; t.w[i] = ws[i]; t.wt[i] = wts[i];
    tiles += ceil_div(Rs[i], CT_T) * ceil_div(Cs[i], CT_T);
  }
  t.tile0[n] = tiles;
  if (dtype == UVC_F32) k_cast_transpose_multi<float><<<tiles, 256, 0, (hipStream_t)stream>>>(params, (float*)shadow, t);
  else k_cast_transpose_multi<bf16_t><<<tiles, 256, 0, (hipStream_t)stream>>>(params, (bf16_t*)shadow, t);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// ---------------------------------------------------------------------------- keyed Exp(1) noise
// Counter-based generator for the Gumbel draws of the gates (model_distilled.py:40,485; uvc_utils.py:443-449): element i of the
// draw (seed, step, site) is a pure function of those four numbers (splitmix64 finaliser over a 64-bit key/counter mix), so
// data-parallel replicas produce identical noise without sharing -- or being able to desynchronise -- a stateful global RNG
// (the reference relies on identically seeded torch generators, joint_train.py:191-196, and breaks that itself by sampling the
// FLOPs report on rank 0 only, :509), and a resumed run needs no RNG state.  E = -log(U), U uniform on (0, 1) with 24 bits.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_exp_noise(float* __restrict__ out, int64_t n, uint64_t key) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint64_t h = splitmix64(key ^ splitmix64((uint64_t)i));
    const float u = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
    out[i] = -logf(u);
  }
}
extern "C" int uvc_exp_noise(float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t site, void* stream) {
  if (!out || n <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_exp_noise: bad argument");
  uint64_t key = seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull;
  key ^= (step + 1) * 0x9E3779B97F4A7C15ull;
  key = (key << 17 | key >> 47) ^ ((uint64_t)site + 1) * 0xC2B2AE3D27D4EB4Full;
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  k_exp_noise<<<nb, 256, 0, (hipStream_t)stream>>>(out, n, key);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_gate_distrib(const float* g, const float* e, float* d, int32_t L, int32_t mode, float eps, void* stream) {
  if (!g || !d || L <= 0 || L > 64 || mode < 0 || mode > 3 || ((mode == 1 || mode == 3) && !e)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gate_distrib: bad argument");
  k_gate_distrib<<<1, 64, 0, (hipStream_t)stream>>>(g, e, d, L, mode, eps);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_gate_grad(const float* g, const float* d, const float* dots, float* dg, int32_t L, int32_t mode, float eps,
                             float beta_acc, void* stream) {
  if (!g || !d || !dots || !dg || L <= 0 || L > 64) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gate_grad: bad argument");
  k_gate_grad<<<1, 64, 0, (hipStream_t)stream>>>(g, d, dots, dg, L, mode, eps, beta_acc);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
