// Distillation loss (fwd+bwd fused) and fused global-norm clip + AdamW for gfx950.
// Reference: UVC/utils/losses.py:25-65 over timm SoftTargetCrossEntropy (joint_train.py:940);
// clip_grad_norm_ + torch.optim.AdamW (joint_train.py:271,428-429).
#include "common.h"
#include "../../include/uvc_kernels.h"

namespace {

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

// one block per batch row
__global__ __launch_bounds__(256) void k_loss_rows(uvc_loss_args a) {
  __shared__ float sh[4];
  const int b = blockIdx.x, C = a.C;
  const float* o = a.o + (size_t)b * C;
  const float* y = a.y_soft + (size_t)b * C;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, o[c]);
  mx = block_reduce(mx, sh, true);
  float se = 0.f, sy = 0.f, syo = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) { se += __expf(o[c] - mx); sy += y[c]; syo += y[c] * o[c]; }
  se = block_reduce(se, sh, false);
  sy = block_reduce(sy, sh, false);
  syo = block_reduce(syo, sh, false);
  const float lse = mx + __logf(se);
  const float base = lse * sy - syo;                         // sum_c -y (o - lse)
  const float wb = (1.0f - (a.kind ? a.alpha : 0.f)) / (float)a.B;
  float kd = 0.f;
  const bool same = (a.o_kd == a.o) || (a.d_okd == a.d_o);
  float* d_o = a.d_o + (size_t)b * C;
  if (a.kind == 0) {
    for (int c = threadIdx.x; c < C; c += 256) d_o[c] = wb * (__expf(o[c] - lse) * sy - y[c]);
    if (threadIdx.x == 0) a.row_scratch[b] = wb * base;
    return;
  }
  const float* ok = a.o_kd + (size_t)b * C;
  const float* t = a.teacher + (size_t)b * C;
  float* d_k = a.d_okd + (size_t)b * C;
  if (a.kind == 2) {
    // hard distillation (losses.py:61-62): kd = F.cross_entropy(o_kd, teacher.argmax(dim=1)) = mean_b (lse(o_kd) - o_kd[arg]);
    // tau is not used; the first maximal class wins, like torch.argmax
    float mk = -INFINITY, mt = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) { mk = fmaxf(mk, ok[c]); mt = fmaxf(mt, t[c]); }
    mk = block_reduce(mk, sh, true);
    mt = block_reduce(mt, sh, true);
    float sk = 0.f, arg = 3.0e38f;
    for (int c = threadIdx.x; c < C; c += 256) { sk += __expf(ok[c] - mk); if (t[c] == mt) arg = fminf(arg, (float)c); }
    sk = block_reduce(sk, sh, false);
    const int cls = (int)(-block_reduce(-arg, sh, true));          // min over the block (class indices are exact in float32)
    const float lk = mk + __logf(sk);
    const float wh = a.alpha / (float)a.B;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float gk = wh * (__expf(ok[c] - lk) - (c == cls ? 1.0f : 0.0f));
      const float gb = wb * (__expf(o[c] - lse) * sy - y[c]);
      if (same) d_o[c] = gb + gk;
      else { d_o[c] = gb; d_k[c] = gk; }
    }
    if (threadIdx.x == 0) a.row_scratch[b] = wb * base + wh * (lk - ok[cls]);
    return;
  }
  const float iT = 1.0f / a.tau;
  float mk = -INFINITY, mt = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) { mk = fmaxf(mk, ok[c] * iT); mt = fmaxf(mt, t[c] * iT); }
  mk = block_reduce(mk, sh, true);
  mt = block_reduce(mt, sh, true);
  float sk = 0.f, stt = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) { sk += __expf(ok[c] * iT - mk); stt += __expf(t[c] * iT - mt); }
  sk = block_reduce(sk, sh, false);
  stt = block_reduce(stt, sh, false);
  const float lk = mk + __logf(sk), lt = mt + __logf(stt);
  const float wk = a.alpha * a.tau * a.tau / ((float)a.B * (float)C);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float lpt = t[c] * iT - lt, lpk = ok[c] * iT - lk;
    const float pt = __expf(lpt), pk = __expf(lpk);
    kd += pt * (lpt - lpk);
    const float gk = wk * iT * (pk - pt);
    const float gb = wb * (__expf(o[c] - lse) * sy - y[c]);
    if (same) d_o[c] = gb + gk;
    else { d_o[c] = gb; d_k[c] = gk; }
  }
  kd = block_reduce(kd, sh, false);
  if (threadIdx.x == 0) a.row_scratch[b] = wb * base + wk * kd;
}

__global__ __launch_bounds__(256) void k_loss_final(const float* rows, int B, float* loss) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += rows[i];
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) loss[0] = s;
}

// ---------------------------------------------------------------------------- grad norm + AdamW
constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void k_sqnorm_partial(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_sqnorm_final(const float* partial, int nb, float* sq, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) {
    const float t = (accumulate ? sq[0] : 0.f) + s;
    sq[0] = t;
    sq[1] = sqrtf(t);                              // the total norm so far: what clip_grad_norm_ returns, without a launch of its own
  }
}

// FLAGS: per-element byte, bit 0 = apply weight decay, bit 1 = frozen (no gradient reached it: untouched, like a
// torch parameter whose .grad is None).  Without flags every element decays and updates.
template <bool SHADOW, bool FLAGS>
__global__ __launch_bounds__(256) void k_adamw(uvc_adamw_args a, float bc1, float bc2_sqrt) {
  const float total = sqrtf(a.sq[0]);
  const float coef = fminf(a.max_norm / (total + 1e-6f), 1.0f);
  if (a.gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) a.gnorm_out[0] = total;
  const float decay = 1.0f - a.lr * a.weight_decay;
  const float step_size = a.lr / bc1;
  const float w1 = 1.0f - a.beta1, w2 = 1.0f - a.beta2;
  const int64_t n4 = a.n >> 2;
  f32x4* p4 = reinterpret_cast<f32x4*>(a.p);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(a.g);
  f32x4* m4 = reinterpret_cast<f32x4*>(a.m);
  f32x4* v4 = reinterpret_cast<f32x4*>(a.v);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    uint32_t fl = 0x01010101u;
    if (FLAGS) {
      fl = reinterpret_cast<const uint32_t*>(a.flags)[i];
      if ((fl & 0x02020202u) == 0x02020202u) continue;
    }
    f32x4 p = p4[i], m = m4[i], v = v4[i];
    const f32x4 g0 = g4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (FLAGS && ((fl >> (8 * e)) & 2u)) continue;
      const float g = g0[e] * coef;
      p[e] *= (!FLAGS || ((fl >> (8 * e)) & 1u)) ? decay : 1.0f;
      m[e] = m[e] + w1 * (g - m[e]);
      v[e] = v[e] * a.beta2 + w2 * g * g;
      const float denom = sqrtf(v[e]) / bc2_sqrt + a.eps;
      p[e] = p[e] - step_size * (m[e] / denom);
    }
    p4[i] = p; m4[i] = m; v4[i] = v;
    if (SHADOW) {
      u32x2 r; r[0] = pack_bf16x2(p[0], p[1]); r[1] = pack_bf16x2(p[2], p[3]);
      reinterpret_cast<u32x2*>(a.p_shadow)[i] = r;
    }
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (n4 << 2) + threadIdx.x; i < a.n; i += 256) {
      const uint32_t f1 = FLAGS ? a.flags[i] : 1u;
      if (f1 & 2u) continue;
      const float g = a.g[i] * coef;
      float p = a.p[i] * ((f1 & 1u) ? decay : 1.0f);
      const float m = a.m[i] + w1 * (g - a.m[i]);
      const float v = a.v[i] * a.beta2 + w2 * g * g;
      p = p - step_size * (m / (sqrtf(v) / bc2_sqrt + a.eps));
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
      if (SHADOW) reinterpret_cast<bf16_t*>(a.p_shadow)[i] = f32_to_bf16(p);
    }
  }
}

// in-place clip of a (small) gradient segment, so later readers see clip_grad_norm_'s scaled grads
__global__ void k_scale_by_clip(float* g, int64_t n, const float* sq, float max_norm) {
  const float coef = fminf(max_norm / (sqrtf(sq[0]) + 1e-6f), 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= coef;
}

}  // namespace

extern "C" int uvc_distill_loss(const uvc_loss_args* p, void* stream) {
  if (!p || !p->o || !p->y_soft || !p->loss || !p->d_o || !p->row_scratch) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_distill_loss: null pointer");
  if (p->kind < 0 || p->kind > 2) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_distill_loss: kind is 0 (none), 1 (soft) or 2 (hard)");
  if (p->kind != 0 && (!p->o_kd || !p->teacher || !p->d_okd)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_distill_loss: distillation needs o_kd, teacher, d_okd");
  if (p->B <= 0 || p->C <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_distill_loss: empty");
  hipStream_t st = (hipStream_t)stream;
  k_loss_rows<<<p->B, 256, 0, st>>>(*p);
  UVC_CHECK_LAUNCH();
  k_loss_final<<<1, 256, 0, st>>>(p->row_scratch, p->B, p->loss);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_grad_sqnorm(const float* g, int64_t n, float* partial, float* sq, int32_t accumulate, void* stream) {
  if (!g || !partial || !sq || n <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_grad_sqnorm: bad argument");
  if (((uintptr_t)g & 15) != 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_grad_sqnorm: g must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int nb = (int)((n / 4 + 255) / 256);
  if (nb > NORM_BLOCKS) nb = NORM_BLOCKS;
  if (nb < 1) nb = 1;
  k_sqnorm_partial<<<nb, 256, 0, st>>>(g, n, partial);
  UVC_CHECK_LAUNCH();
  k_sqnorm_final<<<1, 256, 0, st>>>(partial, nb, sq, accumulate);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_adamw_step(const uvc_adamw_args* p, void* stream) {
  if (!p || !p->p || !p->g || !p->m || !p->v || !p->sq || p->n <= 0 || p->step <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_adamw_step: bad argument");
  if ((((uintptr_t)p->p | (uintptr_t)p->g | (uintptr_t)p->m | (uintptr_t)p->v) & 15) != 0)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_adamw_step: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const double bc1 = 1.0 - pow((double)p->beta1, (double)p->step);
  const double bc2 = 1.0 - pow((double)p->beta2, (double)p->step);
  int nb = (int)((p->n / 4 + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  if (p->flags && ((uintptr_t)p->flags & 3) != 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_adamw_step: flags must be 4-byte aligned");
  if (p->p_shadow && p->flags) k_adamw<true, true><<<nb, 256, 0, st>>>(*p, (float)bc1, (float)sqrt(bc2));
  else if (p->p_shadow) k_adamw<true, false><<<nb, 256, 0, st>>>(*p, (float)bc1, (float)sqrt(bc2));
  else if (p->flags) k_adamw<false, true><<<nb, 256, 0, st>>>(*p, (float)bc1, (float)sqrt(bc2));
  else k_adamw<false, false><<<nb, 256, 0, st>>>(*p, (float)bc1, (float)sqrt(bc2));
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_scale_by_clip(float* g, int64_t n, const float* sq, float max_norm, void* stream) {
  if (!g || !sq || n <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_scale_by_clip: bad argument");
  k_scale_by_clip<<<(int)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(g, n, sq, max_norm);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
