// Fused inference MLP half of a DeiT block for gfx950 (embed_dim 192, bf16 MFMA, float32 residual stream):
//     out = x1 + fc2(GELU(fc1(LayerNorm(x1))))          UVC/models/model_distilled.py:153-166,186-189
// used by the no-grad forwards of the step (the distillation teacher, utils/losses.py:47-49, and eval).
//
// Unfused, the three kernels (LayerNorm, fc1+GELU, fc2+residual) move 620 MB per layer at batch 512, 310 MB of it the
// [M, 768] hidden activation written once and read once.  Here a workgroup owns 256 token rows: every wave normalises its
// 32 rows in registers (a row is spread over the four 16-lane groups of the MFMA B-operand layout, so mean / variance
// are two shuffles), keeps them as bf16 operand fragments, and the hidden dimension is streamed in chunks of 64 units:
//     a^T  = W1[chunk] . h^T      (W1 rows from LDS as the A operand; accumulator of lane (row, g) = 4 hidden units)
//     u^T  = GELU(a^T + b1)       in registers; two accumulator tiles packed = the B operand of the next MFMA
//     out^T += W2[:, chunk] . u^T (W2 staged in LDS already permuted to that packing: one ds_read_b128 per fragment)
// so the hidden activation never leaves the register file.  HBM traffic is x1 once in (LayerNorm + residual; the residual
// re-read hits L2/MALL) and out once: 154 MB.  The 590 KB of weights stream L2 -> LDS once per 256 rows, double-buffered,
// one barrier per chunk.  The kernel is VALU-bound, not HBM-bound: the erf GELU of 64 x 32 values per wave per chunk is
// ~770 VALU instructions against 96 MFMAs, and with two 256-VGPR waves per SIMD in lock-step phases the two do not overlap
// (PMC: VALU busy 31 %, MFMA busy 14 %, waves parked 53 %).  168 us per layer at batch 512 against 210 us for the three
// separate kernels, with a quarter of their HBM traffic.  (Measured without GELU: 120 us, with the tanh form: 153 us -- the
// chunk loop itself, LDS read -> MFMA with two waves per SIMD, is the next thing to pipeline.)  In the training step the
// teacher forward runs beside the student forward and the step time does not change (a 256-VGPR, 115 KB-LDS workgroup
// owns its CU); a stand-alone eval forward gets the 1.2x.
#include "common.h"
#include "../../include/uvc_kernels.h"

namespace {

typedef bf16_t T;
constexpr int D = 192, KT = 6, FC = 64, R = 2, NW = 8, NTH = 64 * NW, ROWS = NW * R * 16;
constexpr int W1S = D * 2 + 32;   // 416 B: 104 words = 40 mod 64 -> conflict-free ds_read_b128 fragment reads
constexpr int W2S = FC * 2 + 32;  // 160 B:  40 words (8, 24, 40, 56 mod 64 are the conflict-free strides)
constexpr int BUF = FC * W1S + D * W2S + FC * 4;   // W1 chunk | W2 chunk (permuted) | b1 chunk
constexpr int NP1 = FC * (D / 8), NP2 = D * (FC / 8);   // 16-byte pieces per chunk: 1536 of W1 + 1536 of W2
static_assert((NP1 + NP2) % NTH == 0 && NP1 % 64 == 0, "uniform staging: whole waves on either side of the W1/W2 boundary");
constexpr int NPT = (NP1 + NP2) / NTH;

__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 frag(const char* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) {
  u32x4 r;
  r[0] = pack_bf16x2(lo[0], lo[1]); r[1] = pack_bf16x2(lo[2], lo[3]);
  r[2] = pack_bf16x2(hi[0], hi[1]); r[3] = pack_bf16x2(hi[2], hi[3]);
  return __builtin_bit_cast(bf16x8, r);
}

__global__ __launch_bounds__(NTH, 2) void k_mlp_fused(uvc_mlp_args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const buf0 = smem;
  char* const buf1 = smem + BUF;
  float* const sG = reinterpret_cast<float*>(smem + 2 * BUF);
  float* const sBt = sG + D;
  float* const sB2 = sBt + D;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
  const T* __restrict__ W1 = reinterpret_cast<const T*>(a.w1);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(a.w2);
  const int nch = a.F / FC;
  for (int i = tid; i < D; i += NTH) { sG[i] = a.gamma[i]; sBt[i] = a.beta[i]; sB2[i] = a.b2[i]; }

  // weight chunk staging through registers: NI1 16-byte pieces of W1[c*FC.., :] and NI2 of W2[:, c*FC..] per thread.
  // Hidden units ch*8..ch*8+7 of the chunk belong to accumulator tile t = ch/2, lane groups gq = (ch%2)*2 + {0,1}; the B
  // operand of k-step s = t/2 holds, per lane group, 4 units of tile 2s then 4 units of tile 2s+1 -- W2 is stored that way.
  u32x4 pw[NPT];
  u32x4 pb;
  auto gload = [&](int c) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int id = tid + NTH * i;                      // id < NP1 is uniform over a wave
      if (id < NP1) { const int row = id / (D / 8), ch = id % (D / 8); pw[i] = *reinterpret_cast<const u32x4*>(W1 + (size_t)(c * FC + row) * D + ch * 8); }
      else { const int id2 = id - NP1, row = id2 / (FC / 8), ch = id2 % (FC / 8); pw[i] = *reinterpret_cast<const u32x4*>(W2 + (size_t)row * a.F + c * FC + ch * 8); }
    }
    pb = *reinterpret_cast<const u32x4*>(a.b1 + c * FC + (tid & (FC / 4 - 1)) * 4);
  };
  auto lstore = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int id = tid + NTH * i;
      if (id < NP1) { const int row = id / (D / 8), ch = id % (D / 8); *reinterpret_cast<u32x4*>(buf + row * W1S + ch * 16) = pw[i]; }
      else {
        const int id2 = id - NP1, row = id2 / (FC / 8), ch = id2 % (FC / 8);
        const int t = ch >> 1, sk = t >> 1, gq0 = (ch & 1) * 2;
        char* base = buf + FC * W1S + row * W2S + (t & 1) * 8;
        u32x2 lo, hi;
        lo[0] = pw[i][0]; lo[1] = pw[i][1]; hi[0] = pw[i][2]; hi[1] = pw[i][3];
        *reinterpret_cast<u32x2*>(base + (sk * 4 + gq0) * 16) = lo;
        *reinterpret_cast<u32x2*>(base + (sk * 4 + gq0 + 1) * 16) = hi;
      }
    }
    if (tid < FC / 4) *reinterpret_cast<u32x4*>(buf + FC * W1S + D * W2S + tid * 16) = pb;
  };

  // the weight stream is cyclic: chunk (c+1) % nch is always prefetched, so the last chunk of a pass stages chunk 0 of the
  // next one and the chunk loop has no tail case
  int par = 0;
  gload(0);
  lstore(buf0);
  __syncthreads();                                     // also covers sG / sBt / sB2

  const int npass = (a.M + ROWS - 1) / ROWS;
  for (int pass = blockIdx.x; pass < npass; pass += gridDim.x) {
    const int m0 = pass * ROWS + w * (R * 16);
    // ---- LayerNorm of this wave's 2 x 16 rows, straight into MFMA B-operand fragments
    bf16x8 hf[R][KT];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      __builtin_amdgcn_sched_barrier(0);
      const int row = m0 + r * 16 + li;
      const bool ok = row < a.M;
      const float okf = ok ? 1.0f : 0.0f;
      const float* xr = a.x + (size_t)(ok ? row : 0) * D;
      f32x4 xv[2 * KT];
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        xv[2 * ks] = *reinterpret_cast<const f32x4*>(xr + (ks * 4 + g) * 8);
        xv[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xr + (ks * 4 + g) * 8 + 4);
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i) s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; q += d * d; }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      const float rstd = rsqrtf(q * (1.0f / D) + a.eps);
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        const int c0 = (ks * 4 + g) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sG + c0), g1 = *reinterpret_cast<const f32x4*>(sG + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBt + c0), b1 = *reinterpret_cast<const f32x4*>(sBt + c0 + 4);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y0[e] = ((xv[2 * ks][e] - mean) * rstd * g0[e] + b0[e]) * okf;        // rows past M: zero operands
          y1[e] = ((xv[2 * ks + 1][e] - mean) * rstd * g1[e] + b1[e]) * okf;
        }
        hf[r][ks] = pack8(y0, y1);
      }
    }

    f32x4 out[R][D / 16];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < D / 16; ++j) out[r][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nch; ++c, par ^= 1) {
      const char* buf = par ? buf1 : buf0;
      char* nbuf = par ? buf0 : buf1;
      gload(c + 1 < nch ? c + 1 : 0);
      // ---- a^T = W1c . h^T      (sched barriers keep the operand reads next to their MFMAs: bounded register use; letting
      //      the compiler hoist all 52 fragment reads of a chunk costs 90 spilled VGPRs and 40 % more time)
      f32x4 acc[R][FC / 16];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < FC / 16; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        bf16x8 wf[FC / 16];
#pragma unroll
        for (int t = 0; t < FC / 16; ++t) wf[t] = frag(buf + (t * 16 + li) * W1S + (ks * 4 + g) * 16);
#pragma unroll
        for (int t = 0; t < FC / 16; ++t)
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][t] = mma(wf[t], hf[r][ks], acc[r][t]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- u^T = GELU(a^T + b1), packed as the next B operand
      bf16x8 uf[R][FC / 32];
      {
        const float* sb1 = reinterpret_cast<const float*>(buf + FC * W1S + D * W2S);
#pragma unroll
        for (int t = 0; t < FC / 16; ++t) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + t * 16 + g * 4);
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][t][e] = Gelu<T>::f(acc[r][t][e] + bb[e]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int sk = 0; sk < FC / 32; ++sk) uf[r][sk] = pack8(acc[r][2 * sk], acc[r][2 * sk + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- out^T += W2c . u^T
#pragma unroll
      for (int j = 0; j < D / 16; ++j) {
        bf16x8 wf[FC / 32];
#pragma unroll
        for (int sk = 0; sk < FC / 32; ++sk) wf[sk] = frag(buf + FC * W1S + (j * 16 + li) * W2S + (sk * 4 + g) * 16);
#pragma unroll
        for (int sk = 0; sk < FC / 32; ++sk)
#pragma unroll
          for (int r = 0; r < R; ++r) out[r][j] = mma(wf[sk], uf[r][sk], out[r][j]);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      lstore(nbuf);
      __syncthreads();
    }

    // ---- out = x1 + mlp + b2: lane (row, g) holds columns j*16 + g*4 .. +3 of every 16-column group
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = m0 + r * 16 + li;
      if (row < a.M) {
        const float* xr = a.x + (size_t)row * D;
        float* orow = a.out + (size_t)row * D;
#pragma unroll
        for (int j = 0; j < D / 16; ++j) {
          const int col = j * 16 + g * 4;
          const f32x4 xres = *reinterpret_cast<const f32x4*>(xr + col);
          const f32x4 bb = *reinterpret_cast<const f32x4*>(sB2 + col);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (out[r][j][e] + bb[e]) + xres[e];
          *reinterpret_cast<f32x4*>(orow + col) = o;
        }
      }
    }
  }
}

}  // namespace

extern "C" int uvc_mlp_fused_supported(int32_t D_, int32_t F, int32_t dtype) { return D_ == D && F > 0 && F % FC == 0 && dtype == UVC_BF16; }

extern "C" int uvc_mlp_fused_fwd(const uvc_mlp_args* p, void* stream) {
  if (!p || !p->x || !p->out || !p->gamma || !p->beta || !p->w1 || !p->b1 || !p->w2 || !p->b2) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: null pointer");
  if (p->M <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: empty");
  if (!uvc_mlp_fused_supported(p->D, p->F, UVC_BF16)) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_mlp_fused_fwd: needs D == 192 and F % 64 == 0");
  if ((((uintptr_t)p->x | (uintptr_t)p->out | (uintptr_t)p->w1 | (uintptr_t)p->w2 | (uintptr_t)p->b1) & 15) != 0)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: buffers must be 16-byte aligned");
  const size_t sh = (size_t)2 * BUF + 3 * D * sizeof(float);
  hipError_t e = hipFuncSetAttribute((const void*)k_mlp_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  if (e != hipSuccess) return uvc_set_error(e, __FILE__, __LINE__);
  const int npass = ceil_div(p->M, ROWS);
  k_mlp_fused<<<npass < 256 ? npass : 256, NTH, sh, (hipStream_t)stream>>>(*p);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
