// The last transformer block only feeds the class (/ distillation) token rows to the head
// (UVC/models/model_distilled.py:507-526: x = norm(x); x[:, 0] (, x[:, 1])), so everything after its attention scores -- the softmax
// rows of the other 196 queries, attn.proj, LayerNorm2, the MLP and, in the backward, their gradients -- is dead for every row but
// those: no output of the step (loss, logits, any gradient) depends on it.  The engine (vit_engine.hip) runs that block's tail on
// the B * ntok token rows only; these are the pieces that differ from the full-row kernels:
//   uvc_attention_tok_fwd / _bwd   attention of the first `ntok` queries of every (image, head) against all N keys
//                                  (:175-185 restricted to rows 0 .. ntok-1); the backward writes the whole dqkv tensor
//                                  (dq is zero for the other rows, dk / dv are dense)
//   uvc_copy_row_groups            gather / scatter of the token rows between [B, N, D] and [B, ntok, D] buffers
// All arithmetic is float32 (the operands are T = float32 or bf16); a workgroup of 256 threads owns one (image, head) with K and V
// staged in LDS, one thread per key for the score / probability work, (4 key groups x 64 columns) for the two weighted sums.
#include "common.h"
#include "../../include/uvc_kernels.h"

namespace {

constexpr int HD = 64, NTH = 256, MAXN = 256, MAXT = 2;

template <typename T> struct Row;      // LDS image of an [N, 64] operand: row stride in 32-bit words is odd, so that thread j walking
template <> struct Row<bf16_t> {       // row j conflicts with nobody
  static constexpr int WORDS = 33;     // 66 bf16
  static __device__ __forceinline__ void put8(uint32_t* row, int c, const bf16_t* src) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
    for (int e = 0; e < 4; ++e) row[c * 4 + e] = v[e];
  }
  static __device__ __forceinline__ float get(const uint32_t* row, int d) {
    const uint32_t w = row[d >> 1];
    return __uint_as_float((d & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  static __device__ __forceinline__ float dot(const uint32_t* row, const float* q) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const uint32_t w = row[i];
      s = __builtin_fmaf(__uint_as_float(w << 16), q[2 * i], s);
      s = __builtin_fmaf(__uint_as_float(w & 0xffff0000u), q[2 * i + 1], s);
    }
    return s;
  }
  static __device__ __forceinline__ void store8(bf16_t* dst, const float* v) {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4*>(dst) = r;
  }
};
template <> struct Row<float> {
  static constexpr int WORDS = 65;
  static __device__ __forceinline__ void put8(uint32_t* row, int c, const float* src) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(src), b = *reinterpret_cast<const u32x4*>(src + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { row[c * 8 + e] = a[e]; row[c * 8 + 4 + e] = b[e]; }
  }
  static __device__ __forceinline__ float get(const uint32_t* row, int d) { return __uint_as_float(row[d]); }
  static __device__ __forceinline__ float dot(const uint32_t* row, const float* q) {
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) s = __builtin_fmaf(__uint_as_float(row[i]), q[i], s);
    return s;
  }
  static __device__ __forceinline__ void store8(float* dst, const float* v) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};

struct TokArgs {
  const void* qkv; void* o; const void* dout; void* dqkv; const int32_t* head_keep;
  int B, N, H, ntok; float scale;
};

// block-wide max / sum over 256 threads in a fixed order (wave shuffles, then the four wave results through LDS)
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// K and V of one (image, head) -> LDS; q (and dO) of the token rows -> float32
template <typename T>
__device__ __forceinline__ void stage_kv(const TokArgs& a, int b, int h, uint32_t* sK, uint32_t* sV) {
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const size_t rs = (size_t)3 * a.H * HD;
  for (int i = threadIdx.x; i < a.N * 8; i += NTH) {
    const int n = i >> 3, c = i & 7;
    const T* base = qkv + ((size_t)b * a.N + n) * rs + (size_t)h * HD + c * 8;
    Row<T>::put8(sK + n * Row<T>::WORDS, c, base + (size_t)a.H * HD);
    Row<T>::put8(sV + n * Row<T>::WORDS, c, base + (size_t)2 * a.H * HD);
  }
}

template <typename T>
__global__ __launch_bounds__(NTH) void k_attn_tok_fwd(TokArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem_tok[];
  uint32_t* sK = smem_tok;
  uint32_t* sV = sK + MAXN * Row<T>::WORDS;
  float* sQ = reinterpret_cast<float*>(sV + MAXN * Row<T>::WORDS);   // [64]
  float* sP = sQ + HD;                                                // [256]
  float* sPart = sP + MAXN;                                           // [4][64]
  float* red = sPart + 4 * HD;                                        // [4]
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  T* o = reinterpret_cast<T*>(a.o);
  const int Dm = a.H * HD;
  if (a.head_keep && a.head_keep[h] == 0) {                           // pruned head (no-grad forwards only): its slice is zeros
    for (int i = tid; i < a.ntok * HD; i += NTH) ElemIO<T>::store(o + ((size_t)b * a.ntok + i / HD) * Dm + h * HD + (i % HD), 0.f);
    return;
  }
  stage_kv<T>(a, b, h, sK, sV);
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  for (int t = 0; t < a.ntok; ++t) {
    __syncthreads();
    if (tid < HD) sQ[tid] = ElemIO<T>::load(qkv + ((size_t)b * a.N + t) * 3 * Dm + h * HD + tid);
    __syncthreads();
    const bool live = tid < a.N;
    const float s = live ? Row<T>::dot(sK + tid * Row<T>::WORDS, sQ) * a.scale : -INFINITY;
    const float m = block_max(s, red);
    const float e = live ? expf(s - m) : 0.f;
    const float z = block_sum(e, red);
    sP[tid] = e / z;
    __syncthreads();
    {
      const int grp = tid >> 6, d = tid & 63;
      float acc = 0.f;
      for (int j = grp; j < a.N; j += 4) acc = __builtin_fmaf(sP[j], Row<T>::get(sV + j * Row<T>::WORDS, d), acc);
      sPart[grp * HD + d] = acc;
    }
    __syncthreads();
    if (tid < HD) ElemIO<T>::store(o + ((size_t)b * a.ntok + t) * Dm + h * HD + tid, (sPart[tid] + sPart[HD + tid]) + (sPart[2 * HD + tid] + sPart[3 * HD + tid]));
  }
}

template <typename T>
__global__ __launch_bounds__(NTH) void k_attn_tok_bwd(TokArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem_tok[];
  uint32_t* sK = smem_tok;
  uint32_t* sV = sK + MAXN * Row<T>::WORDS;
  float* sQ = reinterpret_cast<float*>(sV + MAXN * Row<T>::WORDS);   // [MAXT][64]
  float* sDO = sQ + MAXT * HD;                                        // [MAXT][64]
  float* sDQ = sDO + MAXT * HD;                                       // [MAXT][64]
  float* sP = sDQ + MAXT * HD;                                        // [MAXT][256]
  float* sDS = sP + MAXT * MAXN;                                      // [MAXT][256]   scale * p * (dP - delta)
  float* sPart = sDS + MAXT * MAXN;                                   // [4][64]
  float* red = sPart + 4 * HD;
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int Dm = a.H * HD;
  const T* qkv = reinterpret_cast<const T*>(a.qkv);
  const T* dout = reinterpret_cast<const T*>(a.dout);
  T* dqkv = reinterpret_cast<T*>(a.dqkv);
  if (a.head_keep && a.head_keep[h] == 0) {                           // pruned head (training, masked attn.proj): its dout slice is exactly zero
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int i = tid; i < a.N * 24; i += NTH) {
      const int n = i / 24, which = (i >> 3) % 3, c = i & 7;
      Row<T>::store8(dqkv + (((size_t)b * a.N + n) * 3 + which) * Dm + h * HD + c * 8, v);
    }
    return;
  }
  stage_kv<T>(a, b, h, sK, sV);
  for (int i = tid; i < a.ntok * HD; i += NTH) {
    const int t = i / HD, d = i % HD;
    sQ[i] = ElemIO<T>::load(qkv + ((size_t)b * a.N + t) * 3 * Dm + h * HD + d);
    sDO[i] = ElemIO<T>::load(dout + ((size_t)b * a.ntok + t) * Dm + h * HD + d);
  }
  __syncthreads();
  const bool live = tid < a.N;
  for (int t = 0; t < a.ntok; ++t) {
    const float s = live ? Row<T>::dot(sK + tid * Row<T>::WORDS, sQ + t * HD) * a.scale : -INFINITY;
    const float m = block_max(s, red);
    const float e = live ? expf(s - m) : 0.f;
    const float z = block_sum(e, red);
    const float p = e / z;
    const float dp = live ? Row<T>::dot(sV + tid * Row<T>::WORDS, sDO + t * HD) : 0.f;
    const float delta = block_sum(p * dp, red);
    sP[t * MAXN + tid] = p;
    sDS[t * MAXN + tid] = a.scale * p * (dp - delta);
    __syncthreads();
    {
      const int grp = tid >> 6, d = tid & 63;
      float acc = 0.f;
      for (int j = grp; j < a.N; j += 4) acc = __builtin_fmaf(sDS[t * MAXN + j], Row<T>::get(sK + j * Row<T>::WORDS, d), acc);
      sPart[grp * HD + d] = acc;
    }
    __syncthreads();
    if (tid < HD) sDQ[t * HD + tid] = (sPart[tid] + sPart[HD + tid]) + (sPart[2 * HD + tid] + sPart[3 * HD + tid]);
  }
  __syncthreads();
  // the whole [N, 3, 64] slice of this (image, head): 16-byte pieces, eight consecutive lanes per 64-element row
  for (int i = tid; i < a.N * 24; i += NTH) {
    const int n = i / 24, which = (i >> 3) % 3, c = i & 7;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (which == 0) {
      if (n < a.ntok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = sDQ[n * HD + c * 8 + e];
      }
    } else {
      const float* coef = which == 1 ? sDS : sP;
      const float* vec = which == 1 ? sQ : sDO;
      for (int t = 0; t < a.ntok; ++t) {
        const float w = coef[t * MAXN + n];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(w, vec[t * HD + c * 8 + e], v[e]);
      }
    }
    Row<T>::store8(dqkv + (((size_t)b * a.N + n) * 3 + which) * Dm + h * HD + c * 8, v);
  }
}

template <typename T> size_t tok_lds(bool bwd) {
  size_t w = (size_t)2 * MAXN * Row<T>::WORDS * 4;
  w += bwd ? (size_t)(3 * MAXT * HD + 2 * MAXT * MAXN + 4 * HD + 4) * 4 : (size_t)(HD + MAXN + 4 * HD + 4) * 4;
  return w;
}

int tok_check(const uvc_attn_tok_args* p, bool bwd) {
  if (!p || !p->qkv || !p->o) return uvc_set_error_msg(UVC_ERR_ARG, "attention_tok: null pointer");
  if (bwd && (!p->dout || !p->dqkv)) return uvc_set_error_msg(UVC_ERR_ARG, "attention_tok backward: null pointer");
  if (p->head_dim != HD) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "attention_tok: head_dim must be 64");
  if (p->B <= 0 || p->H <= 0 || p->N <= 0 || p->N > MAXN) return uvc_set_error_msg(UVC_ERR_ARG, "attention_tok: need 0 < N <= 256");
  if (p->ntok < 1 || p->ntok > MAXT || p->ntok > p->N) return uvc_set_error_msg(UVC_ERR_ARG, "attention_tok: ntok must be 1 or 2");
  if (p->dtype != UVC_F32 && p->dtype != UVC_BF16) return uvc_set_error_msg(UVC_ERR_ARG, "attention_tok: bad dtype");
  return UVC_OK;
}

template <typename T> int tok_launch(const uvc_attn_tok_args* p, bool bwd, hipStream_t st) {
  TokArgs a;
  a.qkv = p->qkv; a.o = p->o; a.dout = p->dout; a.dqkv = p->dqkv; a.head_keep = p->head_keep;
  a.B = p->B; a.N = p->N; a.H = p->H; a.ntok = p->ntok; a.scale = p->scale;
  const size_t sh = tok_lds<T>(bwd);
  if (bwd) UVC_MAX_LDS(sh, k_attn_tok_bwd<T>);
  else UVC_MAX_LDS(sh, k_attn_tok_fwd<T>);
  const dim3 grid(p->H, p->B);
  if (bwd) k_attn_tok_bwd<T><<<grid, NTH, sh, st>>>(a);
  else k_attn_tok_fwd<T><<<grid, NTH, sh, st>>>(a);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// 16-byte pieces of row r of group gi: src + gi * sgs + r * row_bytes -> dst + gi * dgs + r * row_bytes
__global__ __launch_bounds__(256) void k_copy_row_groups(const char* __restrict__ src, char* __restrict__ dst, int64_t pieces, int ppg,
                                                         int64_t sgs, int64_t dgs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= pieces) return;
  const int64_t gi = i / ppg, r = i % ppg;
  *reinterpret_cast<u32x4*>(dst + gi * dgs + r * 16) = *reinterpret_cast<const u32x4*>(src + gi * sgs + r * 16);
}

}  // namespace

extern "C" int uvc_attention_tok_fwd(const uvc_attn_tok_args* p, void* stream) {
  if (int e = tok_check(p, false)) return e;
  return p->dtype == UVC_F32 ? tok_launch<float>(p, false, (hipStream_t)stream) : tok_launch<bf16_t>(p, false, (hipStream_t)stream);
}

extern "C" int uvc_attention_tok_bwd(const uvc_attn_tok_args* p, void* stream) {
  if (int e = tok_check(p, true)) return e;
  return p->dtype == UVC_F32 ? tok_launch<float>(p, true, (hipStream_t)stream) : tok_launch<bf16_t>(p, true, (hipStream_t)stream);
}

extern "C" int uvc_copy_row_groups(const void* src, void* dst, int64_t groups, int64_t group_bytes, int64_t src_group_stride, int64_t dst_group_stride,
                                   void* stream) {
  if (!src || !dst) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_copy_row_groups: null pointer");
  if (groups <= 0 || group_bytes <= 0 || (group_bytes | src_group_stride | dst_group_stride) % 16 || (((uintptr_t)src | (uintptr_t)dst) & 15))
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_copy_row_groups: sizes, strides and pointers must be positive multiples of 16 bytes");
  const int64_t ppg = group_bytes / 16, pieces = groups * ppg;
  k_copy_row_groups<<<(unsigned)((pieces + 255) / 256), 256, 0, (hipStream_t)stream>>>((const char*)src, (char*)dst, pieces, (int)ppg, src_group_stride,
                                                                                      dst_group_stride);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
