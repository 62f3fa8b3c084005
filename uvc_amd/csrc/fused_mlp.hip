// Fused MLP half of a DeiT block for gfx950 (embed_dim 192, bf16 MFMA, float32 or bf16 residual stream):
//     out = d1 * (x1 + fc2(GELU(fc1(LayerNorm(x1))))) + d0 * x_prev        UVC/models/model_distilled.py:107-124,199-204,241-247,493
// Two instances of one kernel (k_mlp_fused_v3<TRAIN>):
//   inference (teacher, utils/losses.py:47-49, and eval): nothing but `out` is written; the [M, 768] hidden activation never
//     leaves the register file (154 MB of HBM traffic per layer at batch 512 instead of 620 MB for LayerNorm, fc1+GELU, fc2);
//   training (the student's forward, opt-in: uvc_vit_io.fused_train_mlp): the same pass also stores what the backward reads --
//     LayerNorm(x1) (bf16 [M, D], operand of dW1), its mean / rstd, GELU'(a) and GELU(a) (bf16 [M, F]) -- from the accumulator
//     registers.  Measured 218 us against 187 us for LayerNorm + fc1 + fc2: its 64-byte row pieces of GELU / GELU' write badly.
// History (NOTEBOOK.md 5c / 11): round 1-2's kernel ran 8 waves x 32 rows in lockstep with the weights staged through registers
// (133 us); the structure below is the result of the s_memtime traces and probes of round 2.
#include "common.h"
#include "../../include/uvc_kernels.h"
#include <utility>
#include <algorithm>

namespace {

typedef bf16_t T;
constexpr int D = 192, KT = 6, RW = 2;
constexpr int W1S = D * 2 + 32;   // 416 B: 104 words = 40 mod 64 -> conflict-free ds_read_b128 fragment reads of 16 consecutive rows

__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 pack8u(const f32x4& lo, const f32x4& hi) {
  u32x4 r;
  r[0] = pack_bf16x2(lo[0], lo[1]); r[1] = pack_bf16x2(lo[2], lo[3]);
  r[2] = pack_bf16x2(hi[0], hi[1]); r[3] = pack_bf16x2(hi[2], hi[3]);
  return r;
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& lo, const f32x4& hi) { return __builtin_bit_cast(bf16x8, pack8u(lo, hi)); }

// What the probes of round 2 showed (tools/probe/*.hip, s_memtime traces of the kernel itself; NOTEBOOK.md 5c):
//   * one wave issues v_mfma_f32_16x16x32_bf16 every 18.3 ticks; two waves on a SIMD reach 12.3 together;
//   * a dense VALU stream (the GELU) on one wave of a SIMD stalls the other wave's MFMAs almost completely -- matrix and VALU
//     phases of the two waves of a SIMD ADD, whatever s_setprio says; only memory waits overlap with either; inside ONE wave an
//     MFMA leaves ~4 issue slots that independent VALU work fills;
//   * 8 waves in lockstep (one barrier per chunk) expose their row loads / stores (232 MB per layer) before and after a chunk
//     loop that runs at 38 ticks per MFMA.
// So: a workgroup is FOUR waves (one per SIMD) x 32 rows, two workgroups per CU (<= 246 VGPRs, 70 KB of LDS each); 788 of them at
// batch 512.  Hidden chunks of 32 units, both weight chunks double-buffered, brought in by LDS-DMA (global_load_lds_dwordx4: no
// staging registers) and waited for just before the iteration's one barrier.  The matrix phases are spelled out: fragment reads by
// inline ds_read_b128 a few MFMAs ahead, counted lgkmcnt waits tied to the registers they cover, a scheduling fence per unit.
// Measured at batch 512: 105-115 us (inference form).  A lone round of 512 workgroups is [rows in: 12 us][chunks: 37][rows out: 12]
// -- all workgroups start together, so the memory phases of one do not yet hide under the chunks of another: the next step.
constexpr int V3_NW = 4, V3_NTH = 64 * V3_NW, V3_ROWS = V3_NW * RW * 16, V3_FC = 32;
constexpr int V3_W2S = V3_FC * 2 + 32;                       // 96 B rows: 24 words, conflict-free for ds_read_b128 with row = lane & 15
constexpr int V3_W1B = V3_FC * W1S, V3_W2B = D * V3_W2S;     // 13312 + 18432
constexpr int V3_BUF = V3_W1B + V3_W2B;
constexpr int V3_OFF_B1 = 2 * V3_BUF;
constexpr int V3_N1 = V3_FC * 26 / 64, V3_N2 = D * 6 / 64;    // DMA wave-instructions per chunk: 13 + 18
static_assert(V3_FC * 26 % 64 == 0 && D * 6 % 64 == 0, "whole DMA instructions");
constexpr int V3_OFF_DUMMY = V3_OFF_B1 + 4096, V3_OFF_GB = V3_OFF_DUMMY + 1024, V3_LDS = V3_OFF_GB + 3 * D * 4;
// a wave's 16-row x tile in LDS: rows of 48 + 2 slots (float32 residual stream: 800 B, 13 KB per tile) or 24 + 2 (bf16 stream: 416 B, 7 KB);
// both strides are 8 / 40 words mod 64: conflict-free ds_read_b128 with row = lane & 15
template <bool RLOW> struct XT {
  static constexpr int RSZ = RLOW ? 2 : 4, RSL = D * RSZ / 16, XS = RSL + 2, NXI = (16 * XS + 63) / 64, XTILE = NXI * 1024;
};
static_assert(V3_NW * XT<false>::XTILE <= 2 * V3_BUF, "row tiles fit the weight buffers");

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(LDS_PTR(char))(char*)p; }
template <int OFF> __device__ __forceinline__ u32x4 ds_rd(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void wait_tie2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int... Is, class Fn> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, Fn&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class Fn> __device__ __forceinline__ void static_for(Fn&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// RLOW: x, out, x_prev are bf16 rows (the bf16 residual stream): out is rounded once, at its store, and next_h is the LayerNorm of the
// ROUNDED rows -- what every consumer of `out` reads
template <bool TRAIN, bool RLOW>
__global__ __launch_bounds__(V3_NTH, 2) void k_mlp_fused_v3(uvc_mlp_args a) {
  constexpr int RSZ = XT<RLOW>::RSZ, RSL = XT<RLOW>::RSL, V3_XS = XT<RLOW>::XS, V3_NXI = XT<RLOW>::NXI, V3_XTILE = XT<RLOW>::XTILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nch = a.F / V3_FC;
  // b1, b2, gamma, beta -> LDS (behind the weight buffers); they are read by compiler-visible LDS loads in the prologue only
  float* const sB1 = reinterpret_cast<float*>(smem + V3_OFF_B1);
  float* const sG = reinterpret_cast<float*>(smem + V3_OFF_GB);
  float* const sBt = sG + D;
  float* const sB2 = sBt + D;
  for (int i = tid; i < a.F; i += V3_NTH) sB1[i] = a.b1[i];
  if (tid < D) { sG[tid] = a.gamma[tid]; sBt[tid] = a.beta[tid]; sB2[tid] = a.b2[tid]; }
  float d0 = 0.f, d1 = 1.f;
  if (a.gate) { d0 = a.gate[0]; d1 = a.gate[1]; }
  const unsigned s0 = lds_addr(smem);
  const int m0 = blockIdx.x * V3_ROWS + w * (RW * 16);

  // ---- rows.  A wave's 16-row tile of x goes HBM -> LDS by DMA into the (still empty) weight buffers -- 800-byte rows (48 + 2 slots:
  //      conflict-free ds_read_b128 with row = lane & 15), 13 instructions, no staging registers -- and is read from there TWICE: in
  //      the MFMA B-operand layout for the LayerNorm (lane (row, g): columns (ks*4 + g)*8 .. +7) and in the accumulator layout
  //      (columns j*16 + g*4 .. +3), because fc2's accumulator starts as x1 + b2: the epilogue is stores only and x is read from
  //      HBM once.  (Through registers -- 48 per tile for the LayerNorm, loaded again for the residual -- hipcc spilled 90-190
  //      VGPRs around the loads and a row tile took 8-20 k ticks; k_mlp_fused re-reads x from HBM in its epilogue.)
  bf16x8 hf[RW][KT];
  f32x4 out[RW][D / 16];
  {
    char* const xreg = smem + w * V3_XTILE;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      int ln = lane;
      asm volatile("" : "+v"(ln));                                   // per-tile lane addresses: hoisted out of the loop they cost 52 VGPRs for its whole life
      const char* xb = reinterpret_cast<const char*>(a.x);
#pragma unroll
      for (int i = 0; i < V3_NXI; ++i) {
        const int s = i * 64 + ln, row = s / V3_XS, pc = s % V3_XS;
        int grow = m0 + r * 16 + (row < 16 ? row : 15);
        grow = grow < a.M ? grow : a.M - 1;                          // rows past M: any valid row (masked below, never stored)
        const unsigned off = (unsigned)grow * (unsigned)(D * RSZ) + (unsigned)((pc < RSL ? pc : 0) * 16);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(xb + (unsigned long long)off),
                                         (void __attribute__((address_space(3)))*)(xreg + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (r == 0) __syncthreads();                                   // gamma / beta / b2 staged (rides on the first tile's wait)
      const float okf = (m0 + r * 16 + li) < a.M ? 1.0f : 0.0f;
      const char* xrow = xreg + li * (V3_XS * 16);
      // the row tile in registers (48 VGPRs); gamma / beta / b2 by inline ds_read per k-step: as ordinary loads hipcc keeps the first
      // tile's 96 + 48 values alive for the second (common subexpressions) and spills around them
      f32x4 xv[2 * KT];
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        if constexpr (RLOW) {                      // eight consecutive bf16 columns in 16 bytes
          const u32x4 q = *reinterpret_cast<const u32x4*>(xrow + (ks * 4 + g) * 16);
          xv[2 * ks] = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
          xv[2 * ks + 1] = f32x4{__uint_as_float(q[2] << 16), __uint_as_float(q[2] & 0xffff0000u), __uint_as_float(q[3] << 16), __uint_as_float(q[3] & 0xffff0000u)};
        } else {
          xv[2 * ks] = *reinterpret_cast<const f32x4*>(xrow + (ks * 4 + g) * 32);
          xv[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xrow + (ks * 4 + g) * 32 + 16);
        }
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i) s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      s = sum_rows4(s);
      const float mean = s * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; q += d * d; }
      q = sum_rows4(q);
      const float rstd = rsqrtf(q * (1.0f / D) + a.eps);
      const int prow = m0 + r * 16 + li;
      if (TRAIN && prow < a.M && g == 0) { a.mean[prow] = mean; a.rstd[prow] = rstd; }
      const unsigned ga = s0 + (unsigned)(V3_OFF_GB + g * 32), ba = ga + (unsigned)(D * 4);
      __builtin_amdgcn_sched_barrier(0);
      static_for<KT>([&](auto ksv) {
        constexpr int ks = ksv.value;
        u32x4 gq0 = ds_rd<ks * 128>(ga), gq1 = ds_rd<ks * 128 + 16>(ga), bq0 = ds_rd<ks * 128>(ba), bq1 = ds_rd<ks * 128 + 16>(ba);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gq0), "+v"(gq1), "+v"(bq0), "+v"(bq1));
        const f32x4 g0 = __builtin_bit_cast(f32x4, gq0), g1 = __builtin_bit_cast(f32x4, gq1);
        const f32x4 b0 = __builtin_bit_cast(f32x4, bq0), b1 = __builtin_bit_cast(f32x4, bq1);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y0[e] = ((xv[2 * ks][e] - mean) * rstd * g0[e] + b0[e]) * okf;        // rows past M: zero operands
          y1[e] = ((xv[2 * ks + 1][e] - mean) * rstd * g1[e] + b1[e]) * okf;
        }
        hf[r][ks] = pack8(y0, y1);
        if (TRAIN && prow < a.M) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.h) + (size_t)prow * D + (ks * 4 + g) * 8) = __builtin_bit_cast(u32x4, hf[r][ks]);
        asm volatile("" : "+v"(hf[r][ks]));            // materialise here: LLVM otherwise SINKS this arithmetic to the first use inside the
                                                       // chunk loop and keeps (spills) the 24 loaded vectors until then
        __builtin_amdgcn_sched_barrier(0);             // one k-step at a time: unfenced, the scheduler collects all 24 reads first
      });
      const unsigned b2a = s0 + (unsigned)(V3_OFF_GB + 2 * D * 4 + g * 16);
      static_for<D / 16>([&](auto jv) {
        constexpr int j = jv.value;
        f32x4 xr;
        if constexpr (RLOW) {
          const u32x2 q = *reinterpret_cast<const u32x2*>(xrow + (j * 4 + g) * 8);
          xr = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
        } else xr = *reinterpret_cast<const f32x4*>(xrow + (j * 4 + g) * 16);
        u32x4 bq = ds_rd<j * 64>(b2a);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq));
        const f32x4 bv = __builtin_bit_cast(f32x4, bq);
#pragma unroll
        for (int e = 0; e < 4; ++e) out[r][j][e] = xr[e] + bv[e];
        asm volatile("" : "+v"(out[r][j]));
        if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                                  // every wave is done with its rows: the weight buffers may be filled
  const unsigned w1lane = s0 + (unsigned)(li * W1S + g * 16);
  const unsigned w2lane = s0 + (unsigned)(V3_W1B + li * V3_W2S + g * 16);
  const unsigned b1lane = s0 + (unsigned)(V3_OFF_B1 + g * 32);

  // ---- LDS-DMA of a weight chunk: a wave-instruction fills 64 consecutive 16-byte slots.  W1 rows are 24 + 2 slots (416 B), W2 rows
  //      4 + 2 (96 B); pad slots fetch piece 0 again.  Instructions 0..12 = W1 image, 13..30 = W2 image; wave w issues w, w+4, ...
  //      LDS row lr of the W1 image holds hidden unit (lr >> 2 & 3) * 8 + (lr >> 4) * 4 + (lr & 3) of the chunk, so that a lane of the
  //      fc1 result holds 8 consecutive k indices of fc2 (its B operand) -- the permutation costs nothing, it is where the DMA aims.
  // Eight instructions per wave and chunk, ALWAYS (the 32nd aims at a dummy KB): with a branch around an instruction hipcc
  // puts s_waitcnt vmcnt(0) in front of every one of them and the eight L2 round trips serialise (4-6 k cycles per chunk).
  unsigned doff[8];
  const char* gsrc[8];                 // wave-uniform: source matrix, bytes per chunk, LDS offset inside the chunk buffer
  unsigned gstep[8], ldst[8], lbuf[8];
  bool isw1[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = w + 4 * q;
    if (j < V3_N1) {
      const int s = j * 64 + lane, lr = s / 26, pc = s % 26;
      const int h = ((lr >> 2) & 3) * 8 + (lr >> 4) * 4 + (lr & 3);
      doff[q] = (unsigned)(h * D * 2 + (pc < 24 ? pc : 0) * 16);
      gsrc[q] = reinterpret_cast<const char*>(a.w1); gstep[q] = (unsigned)(V3_FC * D * 2);
    } else {
      const int s = ((j < V3_N1 + V3_N2 ? j : V3_N1) - V3_N1) * 64 + lane, r = s / 6, pc = s % 6;
      doff[q] = (unsigned)(r * a.F * 2 + (pc < 4 ? pc : 0) * 16);
      gsrc[q] = reinterpret_cast<const char*>(a.w2); gstep[q] = (unsigned)(V3_FC * 2);
    }
    isw1[q] = j < V3_N1;
    const bool real = j < V3_N1 + V3_N2;
    ldst[q] = real ? (unsigned)(j * 1024) : (unsigned)V3_OFF_DUMMY;           // dummy KB behind b1 (F <= 1024)
    lbuf[q] = real ? (unsigned)V3_BUF : 0u;
  }
  // W1 of chunk c1 and W2 of chunk c2 (the loop is software-pipelined: fc2 runs one chunk behind fc1); a chunk index past the end
  // sends the instruction to the dummy KB
  auto dma = [&](int c1, int c2) {
    // (chunk offset + lane offset) formed per issue and added to the SCALAR base (the other way round hipcc keeps a 64-bit pointer
    // per instruction in spilled VGPRs)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int cc = isw1[q] ? c1 : c2;
      const bool ok = cc < nch && lbuf[q] != 0u;
      const char* src = gsrc[q] + (unsigned long long)((unsigned)(ok ? cc : 0) * gstep[q] + doff[q]);
      char* dst = smem + (ok ? ldst[q] + (unsigned)(cc & 1) * lbuf[q] : (unsigned)V3_OFF_DUMMY);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };

  dma(0, nch);                                                      // W1 of chunk 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                  // chunk 0 landed
  __builtin_amdgcn_sched_barrier(0);

  // ---- chunk pipeline.  Iteration c: fc1 of chunk c (24 MFMAs), then fc2 of chunk c - 1 (24 MFMAs) with the GELU of chunk c issued
  //      BETWEEN its MFMAs: inside one wave an MFMA leaves ~4 issue slots that independent VALU work fills for free, while the same
  //      GELU as a separate phase (or on the other wave of the SIMD) costs its full issue time on top.  W1 of chunk c + 1
  //      and W2 of chunk c arrive during iteration c.
  bf16x8 uf[RW];
  f32x4 acc[RW][2], gpv[RW][2];
  auto fc1 = [&](int c) {
    const unsigned w1a = w1lane + (unsigned)((c & 1) * V3_BUF);
    const unsigned ba = b1lane + (unsigned)(c * (V3_FC * 4));
    // a^T = b1 + W1c . h^T: six k-steps of two fragments (hidden tiles t = 0, 1), each feeding the two row tiles; fragments of
    // k-step ks + 2 are requested under the MFMAs of k-step ks
    u32x4 bi[2], f1[3][2];
    bi[0] = ds_rd<0>(ba); bi[1] = ds_rd<16>(ba);
    f1[0][0] = ds_rd<0>(w1a); f1[0][1] = ds_rd<16 * W1S>(w1a);
    f1[1][0] = ds_rd<64>(w1a); f1[1][1] = ds_rd<16 * W1S + 64>(w1a);
    __builtin_amdgcn_sched_barrier(0);
    static_for<KT>([&](auto ksv) {
      constexpr int ks = ksv.value, cur = ks % 3, nxt = (ks + 2) % 3;
      wait_tie2<(ks + 1 < KT) ? 2 : 0>(f1[cur][0], f1[cur][1]);
      if constexpr (ks == 0) asm volatile("" : "+v"(bi[0]), "+v"(bi[1]));
      static_for<2>([&](auto tv) {
        constexpr int t = tv.value;
        const bf16x8 af = __builtin_bit_cast(bf16x8, f1[cur][t]);
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r][t] = mma(af, hf[r][ks], ks == 0 ? __builtin_bit_cast(f32x4, bi[t]) : acc[r][t]);
        if constexpr (ks + 2 < KT) f1[nxt][t] = ds_rd<t * 16 * W1S + (ks + 2 < KT ? ks + 2 : 0) * 64>(w1a);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };
  // out^T += W2c . u^T (twelve output tiles, one k-step; fragment j + 6 is requested when fragment j has been used), FC2 = false:
  // only the GELU.  GELU = true: units 0 .. 7 each carry the GELU of two values of `acc`; the packed result replaces `uf` at the end.
  auto fc2_gelu = [&](auto fc2v, auto geluv, int cprev) {
    constexpr bool FC2 = decltype(fc2v)::value, GELU = decltype(geluv)::value;
    const unsigned w2a = w2lane + (unsigned)((cprev & 1) * V3_BUF);
    u32x4 f2[6];
    if constexpr (FC2) {
      static_for<6>([&](auto jv) { f2[jv.value] = ds_rd<jv.value * 16 * V3_W2S>(w2a); });
      __builtin_amdgcn_sched_barrier(0);
    }
    static_for<12>([&](auto jv) {
      constexpr int j = jv.value, slot = j % 6;
      if constexpr (FC2) {
        constexpr int younger = (j < 6) ? 5 : 11 - j;               // reads issued after fragment j's that may still be in flight
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f2[slot]) : "n"(younger));
        const bf16x8 af = __builtin_bit_cast(bf16x8, f2[slot]);
#pragma unroll
        for (int r = 0; r < RW; ++r) out[r][j] = mma(af, uf[r], out[r][j]);
        if constexpr (j + 6 < 12) f2[slot] = ds_rd<(j + 6 < 12 ? j + 6 : 0) * 16 * V3_W2S>(w2a);
      }
      if constexpr (GELU && j < 8) {
        constexpr int r = j >> 2, t = (j >> 1) & 1, e0 = (j & 1) * 2;
        if constexpr (TRAIN) {
          float f0, g0, f1, g1;
          Gelu<T>::fg(acc[r][t][e0], f0, g0);
          Gelu<T>::fg(acc[r][t][e0 + 1], f1, g1);
          acc[r][t][e0] = f0; gpv[r][t][e0] = g0; acc[r][t][e0 + 1] = f1; gpv[r][t][e0 + 1] = g1;
        } else {
          acc[r][t][e0] = Gelu<T>::f(acc[r][t][e0]);
          acc[r][t][e0 + 1] = Gelu<T>::f(acc[r][t][e0 + 1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (GELU) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        uf[r] = pack8(acc[r][0], acc[r][1]);
        if constexpr (TRAIN) {                          // what the backward reads: GELU(a) (operand of dW2) and GELU'(a); a lane holds 8 consecutive units
          const int row = m0 + r * 16 + li;
          if (row < a.M) {
            const size_t o = (size_t)row * a.F + (size_t)(cprev + 1) * V3_FC + g * 8;
            *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.u) + o) = __builtin_bit_cast(u32x4, uf[r]);
            *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.gp) + o) = pack8u(gpv[r][0], gpv[r][1]);
          }
        }
      }
    }
  };
  auto end_iter = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's pieces of the next weights have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    dma(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    fc1(0);
    fc2_gelu(std::false_type{}, std::true_type{}, -1);
    end_iter();
  }
  for (int c = 1; c < nch; ++c) {
    dma(c + 1, c);
    __builtin_amdgcn_sched_barrier(0);
    fc1(c);
    fc2_gelu(std::true_type{}, std::true_type{}, c - 1);
    end_iter();
  }
  fc2_gelu(std::true_type{}, std::false_type{}, nch - 1);

  // ---- out = d1 * ((x1 + b2) + mlp) + d0 * x_prev
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int row = m0 + r * 16 + li;
    if (row < a.M) {
      char* orow = reinterpret_cast<char*>(a.out) + (size_t)row * D * RSZ;
#pragma unroll
      for (int j = 0; j < D / 16; ++j) {
        const int col = j * 16 + g * 4;
        f32x4 o = out[r][j];
        if (a.gate) {
          f32x4 xp;
          if constexpr (RLOW) {
            const u32x2 q = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(a.x_prev) + (size_t)row * D + col);
            xp = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
          } else xp = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.x_prev) + (size_t)row * D + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(d1, o[e], __builtin_fmaf(d0, xp[e], 0.0f));
        }
        if constexpr (RLOW) {
          u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]);
          *reinterpret_cast<u32x2*>(orow + col * 2) = q;
          o = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
        } else *reinterpret_cast<f32x4*>(orow + col * 4) = o;
        out[r][j] = o;                                 // (bf16 stream: the rounded row, which next_h below normalises)
      }
    }
  }
  // ---- the NEXT block's LayerNorm1 of the rows just produced (they are whole rows in this wave's registers: lane (row, g) holds
  //      48 of the 192 columns, the other three lanes of the row the rest): next_h = LN(out; next_gamma, next_beta) as bf16, so the
  //      consumer's qkv GEMM starts from it and the stand-alone pass (read 77 MB, write 39 MB per block) disappears.  Two-pass
  //      statistics, same arithmetic as k_ln_fwd_v (mean, then centred squares).
  if (a.next_h) {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int row = m0 + r * 16 + li;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < D / 16; ++j) s += (out[r][j][0] + out[r][j][1]) + (out[r][j][2] + out[r][j][3]);
      s = sum_rows4(s);
      const float mean = s * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < D / 16; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dd = out[r][j][e] - mean; q += dd * dd; }
      q = sum_rows4(q);
      const float rstd = rsqrtf(q * (1.0f / D) + a.eps);
      if (row < a.M) {
        if (a.next_mean && g == 0) { a.next_mean[row] = mean; a.next_rstd[row] = rstd; }
        T* hrow = reinterpret_cast<T*>(a.next_h) + (size_t)row * D;
#pragma unroll
        for (int j = 0; j < D / 16; ++j) {
          const int col = j * 16 + g * 4;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(a.next_gamma + col), bt = *reinterpret_cast<const f32x4*>(a.next_beta + col);
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (out[r][j][e] - mean) * rstd * gm[e] + bt[e];
          u32x2 pk; pk[0] = pack_bf16x2(y[0], y[1]); pk[1] = pack_bf16x2(y[2], y[3]);
          *reinterpret_cast<u32x2*>(hrow + col) = pk;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_mlp_fused_p: the inference form for the bf16 residual stream as ONE persistent 8-wave workgroup per CU (NOTEBOOK.md 5g).
// k_mlp_fused_v3 at batch 512 is two generations of workgroups (788 = 512 + 276), each [rows in: two serial HBM round trips + LayerNorm]
// [24 chunks][rows out], all workgroups of the chip in the same phase.  Here:
//   * a workgroup owns a contiguous range of 16-row tiles (24 or 25 at batch 512) and walks it in passes of up to 13 tiles: tile k of a
//     pass belongs to wave k & 7 (k >= 8: that wave's second tile), i.e. waves 0..3 run two tiles (NT = 2), waves 4..7 one (NT = 1; wave 4
//     two in a 13-tile pass): every SIMD carries three tiles per pass, the passes of a 24-tile range are equal;
//   * the weights stream through LDS ONCE per ~200 rows (half the L2 -> LDS bytes of the 4-wave form, three DMA instructions per wave and
//     chunk, none of them a dummy) through THREE chunk buffers: a chunk is requested two iterations before its use -- with one request
//     per iteration in flight the L2 -> LDS round trip (1-1.5 us) was the iteration time;
//   * W1, W2 and the row tiles are unpadded images whose 16-byte slots are XOR-swizzled in the SOURCE address of the DMA (384-byte rows:
//     slot ^ ((row >> 1) & 7); W2's 64-byte rows: slot ^ ((4 - (row >> 2)) & 3)): conflict-free for the ds_read_b128 / b64 lane groups;
//   * the rows of the NEXT pass are requested by LDS-DMA into the wave's own row slots right behind iteration 0's weight request (vmcnt
//     retires in order: the first wait that covers them is the one at the end of iteration 2) and the stores of a pass drain under the
//     next pass's chunks;
//   * every LDS read is inline assembly (a compiler-visible LDS read behind an LDS-DMA costs s_waitcnt vmcnt(0));
//   * per row the arithmetic is k_mlp_fused_v3's, operation for operation: outputs are bit-identical (tests/test_bf16_residual_gpu.py).
constexpr int P_NW = 8, P_NTH = 64 * P_NW, P_MIN_ROWS = 16384, P_PASS = 13;
constexpr int P_W1B = V3_FC * D * 2, P_W2B = D * V3_FC * 2;   // 12288 + 12288
constexpr int P_BUF = P_W1B + P_W2B, P_NBUF = 3;
constexpr int P_XT = 16 * D * 2;                             // 6144: one 16-row tile of bf16 rows; slot k of the pass at P_OFF_X + k * P_XT
constexpr int P_OFF_X = P_NBUF * P_BUF;
constexpr int P_OFF_B1 = P_OFF_X + P_PASS * P_XT;
constexpr int P_OFF_GB = P_OFF_B1 + 4096, P_LDS = P_OFF_GB + 5 * D * 4;   // gamma, beta, b2, next_gamma, next_beta
constexpr int P_ND = (P_BUF / 1024 - P_NW / 2) / (P_NW / 2);  // weight DMA instructions per loader wave and chunk (5; the other four waves: 1)
constexpr int P_NR = P_XT / 1024;                            // row DMA instructions per loader wave and iteration 0..3 (one slot)
static_assert(P_LDS <= 160 * 1024, "one workgroup per CU");
static_assert(P_XT % 1024 == 0 && P_BUF == (P_ND + 1) * (P_NW / 2) * 1024 && P_PASS <= 13, "whole DMA instructions; four loader waves, four slots each");
static_assert(P_NW * P_XT + 256 < 65536, "second-tile offsets are ds_read immediates");

template <int OFF> __device__ __forceinline__ u32x2 ds_rd64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ f32x4 bf4(const u32x2& q) {
  return f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(P_NTH, 1) void k_mlp_fused_p(uvc_mlp_args a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nch = a.F / V3_FC;
  {
    float* const sB1 = reinterpret_cast<float*>(smem + P_OFF_B1);
    float* const sG = reinterpret_cast<float*>(smem + P_OFF_GB);
    for (int i = tid; i < a.F; i += P_NTH) sB1[i] = a.b1[i];
    if (tid < D) {
      sG[tid] = a.gamma[tid]; sG[D + tid] = a.beta[tid]; sG[2 * D + tid] = a.b2[tid];
      if (a.next_h) { sG[3 * D + tid] = a.next_gamma[tid]; sG[4 * D + tid] = a.next_beta[tid]; }
    }
  }
  float d0 = 0.f, d1 = 1.f;
  if (a.gate) { d0 = a.gate[0]; d1 = a.gate[1]; }
  const unsigned s0 = lds_addr(smem);
  const int t0 = (int)((long long)blockIdx.x * ntiles / gridDim.x), t1 = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  // the passes of this range: equal shares of at most P_PASS tiles
  int npass = (t1 - t0 + P_PASS - 1) / P_PASS;
  // first row of tile k of a pass [pb, pb + cnt) (a.M: no such tile -- its rows are masked everywhere)
  auto tile_rows = [&](int pb, int cnt, int k) { return (k < cnt) ? (pb + k) * 16 : a.M; };

  // ---- rows of a pass: HBM -> LDS by DMA, 6 instructions per tile, issued by the LOADER waves 4..7 (their one-tile passes leave them the
  //      issue slots), one slot per iteration 0..3 of the pass before: wave 4 + i fills slots i, i + 4, i + 8 and 12 (i > 0: slot i + 8 a
  //      second time -- four slots per wave, always).  LDS slot n = row * 24 + s' of a tile image
  //      holds the 16-byte piece s = s' ^ ((row >> 1) & 7) of that row
  auto rowdma = [&](int npb, int ncnt, int q) {
    int ln = lane;
    asm volatile("" : "+v"(ln));                                     // lane addresses recomputed per call (VGPRs for the kernel's life otherwise)
    const char* xb = reinterpret_cast<const char*>(a.x);
    const int slot = (q < 3 || w == P_NW / 2) ? (w & 3) + 4 * q : (w & 3) + 8;
    const int mb = tile_rows(npb, ncnt, slot);
#pragma unroll
    for (int i = 0; i < P_XT / 1024; ++i) {
      const int n = i * 64 + ln, row = n / 24, sp = n - row * 24;
      int grow = mb + row;
      grow = grow < a.M ? grow : a.M - 1;
      const unsigned off = (unsigned)grow * (unsigned)(D * 2) + (unsigned)((sp ^ ((row >> 1) & 7)) * 16);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(xb + (unsigned long long)off),
                                       (void __attribute__((address_space(3)))*)(smem + P_OFF_X + slot * P_XT + i * 1024), 16, 0, 0);
    }
  };

  // ---- weights of a chunk: instructions 0..11 = W1 image (row lr = hidden unit (lr >> 2 & 3) * 8 + (lr >> 4) * 4 + (lr & 3) of the chunk, so
  //      that a lane of the fc1 result holds 8 consecutive k of fc2; slots swizzled like the row tiles'), 12..23 = W2 image (slot
  //      n = row * 4 + s' holds piece s = s' ^ ((4 - (row >> 2)) & 3)).  An LDS-DMA instruction costs its wave 100-200 cycles of issue: the two-tile
  //      waves 0..3 (the longer chain of an iteration) issue ONE each (0..3), the loader waves 4 + i five (4 + i, 8 + i, ..., 20 + i).
  unsigned doff[P_ND];
  const char* gsrc[P_ND];
  unsigned gstep[P_ND], ldst[P_ND];
  bool isw1[P_ND];
#pragma unroll
  for (int q = 0; q < P_ND; ++q) {
    const int j = (w < P_NW / 2) ? w : w + 4 * q;
    if (j < P_W1B / 1024) {
      const int n = j * 64 + lane, lr = n / 24, sp = n - lr * 24;
      const int h = ((lr >> 2) & 3) * 8 + (lr >> 4) * 4 + (lr & 3);
      doff[q] = (unsigned)(h * D * 2 + (sp ^ ((lr >> 1) & 7)) * 16);
      gsrc[q] = reinterpret_cast<const char*>(a.w1); gstep[q] = (unsigned)(V3_FC * D * 2);
      ldst[q] = (unsigned)(j * 1024);
    } else {
      const int jj = j - P_W1B / 1024;
      const int n = jj * 64 + lane, row = n >> 2, sp = n & 3;
      doff[q] = (unsigned)(row * a.F * 2 + (sp ^ ((4 - (row >> 2)) & 3)) * 16);
      gsrc[q] = reinterpret_cast<const char*>(a.w2); gstep[q] = (unsigned)(V3_FC * 2);
      ldst[q] = (unsigned)(P_W1B + jj * 1024);
    }
    isw1[q] = j < P_W1B / 1024;
  }
  // W1 of chunk c1 -> buffer at byte offset o1, W2 of chunk c2 -> buffer at o2
  auto dma = [&](auto nv, int c1, unsigned o1, int c2, unsigned o2) {
#pragma unroll
    for (int q = 0; q < decltype(nv)::value; ++q) {
      const int cc = isw1[q] ? c1 : c2;
      const char* src = gsrc[q] + (unsigned long long)((unsigned)cc * gstep[q] + doff[q]);
      char* dst = smem + ldst[q] + (isw1[q] ? o1 : o2);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };

  // per-lane LDS addresses
  const int fx = (li >> 1) & 7;
  const unsigned sw0 = (unsigned)(li * (D * 2) + (g ^ fx) * 16), sw1 = (unsigned)(li * (D * 2) + ((4 + g) ^ fx) * 16);   // 384-byte rows: k-steps even / odd
  const unsigned xe0 = s0 + (unsigned)(P_OFF_X + w * P_XT) + sw0, xe1 = s0 + (unsigned)(P_OFF_X + w * P_XT) + sw1;
  unsigned xo[4];                                                                                               // residual reads: 8-byte piece j * 4 + g
#pragma unroll
  for (int q = 0; q < 4; ++q) xo[q] = s0 + (unsigned)(P_OFF_X + w * P_XT + li * (D * 2) + (((2 * q + (g >> 1)) ^ fx) * 16) + (g & 1) * 8);
  const unsigned w1e0 = s0 + sw0, w1e1 = s0 + sw1;
  const unsigned w2lane = s0 + (unsigned)(P_W1B + li * (V3_FC * 2) + ((g ^ ((4 - (li >> 2)) & 3)) * 16));
  const unsigned b1lane = s0 + (unsigned)(P_OFF_B1 + g * 32);
  const unsigned ga = s0 + (unsigned)(P_OFF_GB + g * 32), ba = ga + (unsigned)(D * 4);
  const unsigned b2a = s0 + (unsigned)(P_OFF_GB + 2 * D * 4 + g * 16), nga = b2a + (unsigned)(D * 4), nba = nga + (unsigned)(D * 4);

  // chunk buffers: iteration G (counted over all passes) computes fc1 from o_cur, fc2 (of the chunk before) from o_nn, and requests
  // W1 of chunk G + 2 into o_nn and W2 of chunk G + 1 into o_nxt; then (o_cur, o_nxt, o_nn) <- (o_nxt, o_nn, o_cur)
  unsigned o_cur = 0, o_nxt = P_BUF, o_nn = 2 * P_BUF;
  int pb = t0, cnt = (t1 - t0 + npass - 1) / (npass > 0 ? npass : 1);
  int mb0 = tile_rows(pb, cnt, w), mb1 = tile_rows(pb, cnt, P_NW + w);
  const bool loader = w >= P_NW / 2;
  if (loader) {
#pragma unroll
    for (int q = 0; q < 4; ++q) rowdma(pb, cnt, q);
    dma(std::integral_constant<int, P_ND>{}, 0, o_cur, 0, o_cur);
    dma(std::integral_constant<int, P_ND>{}, 1, o_nxt, 0, o_cur);   // (W2 of chunk 0 a second time: every instruction, always)
  } else {
    dma(std::integral_constant<int, 1>{}, 0, o_cur, 0, o_cur);
    dma(std::integral_constant<int, 1>{}, 1, o_nxt, 0, o_cur);
  }
  wait_vm<0>();
  __syncthreads();                                                  // constants staged, chunks 0 / 1 and this wave's rows landed
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 hf[RW][KT];
  f32x4 out[RW][D / 16];
  bf16x8 uf[RW];
  f32x4 acc[RW][2];

  auto pass = [&](auto ntv, auto ldv, int npb, int ncnt) {
    constexpr int NT = decltype(ntv)::value;
    constexpr bool LD = decltype(ldv)::value;
    // ---- rows: LayerNorm2 in the MFMA B-operand layout -> hf; x1 + b2 in the accumulator layout -> out (fc2's initial accumulator)
    static_for<NT>([&](auto rv) {
      constexpr int r = rv.value, XR = r * P_NW * P_XT;
      const int mb = r ? mb1 : mb0;
      u32x4 xq[KT];
      static_for<KT>([&](auto ksv) { constexpr int ks = ksv.value; xq[ks] = ds_rd<XR + (ks >> 1) * 128>((ks & 1) ? xe1 : xe0); });
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]), "+v"(xq[5]));
      f32x4 xv[2 * KT];
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) {
        const u32x4 q = xq[ks];
        xv[2 * ks] = f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
        xv[2 * ks + 1] = f32x4{__uint_as_float(q[2] << 16), __uint_as_float(q[2] & 0xffff0000u), __uint_as_float(q[3] << 16), __uint_as_float(q[3] & 0xffff0000u)};
      }
      const float okf = (mb + li) < a.M ? 1.0f : 0.0f;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i) s += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      s = sum_rows4(s);
      const float mean = s * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; q += d * d; }
      q = sum_rows4(q);
      const float rstd = rsqrtf(q * (1.0f / D) + a.eps);
      __builtin_amdgcn_sched_barrier(0);
      static_for<KT>([&](auto ksv) {
        constexpr int ks = ksv.value;
        u32x4 gq0 = ds_rd<ks * 128>(ga), gq1 = ds_rd<ks * 128 + 16>(ga), bq0 = ds_rd<ks * 128>(ba), bq1 = ds_rd<ks * 128 + 16>(ba);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gq0), "+v"(gq1), "+v"(bq0), "+v"(bq1));
        const f32x4 g0 = __builtin_bit_cast(f32x4, gq0), g1 = __builtin_bit_cast(f32x4, gq1);
        const f32x4 b0 = __builtin_bit_cast(f32x4, bq0), b1 = __builtin_bit_cast(f32x4, bq1);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y0[e] = ((xv[2 * ks][e] - mean) * rstd * g0[e] + b0[e]) * okf;
          y1[e] = ((xv[2 * ks + 1][e] - mean) * rstd * g1[e] + b1[e]) * okf;
        }
        hf[r][ks] = pack8(y0, y1);
        asm volatile("" : "+v"(hf[r][ks]));
        __builtin_amdgcn_sched_barrier(0);
      });
      static_for<D / 64>([&](auto jqv) {                           // four output tiles at a time: 4 x (8-byte residual piece, 16 bytes of b2)
        constexpr int j0 = jqv.value * 4;
        u32x2 xr[4];
        u32x4 bq[4];
        static_for<4>([&](auto qv) {
          constexpr int j = j0 + qv.value;
          xr[qv.value] = ds_rd64<XR + (j >> 2) * 128>(xo[j & 3]);
          bq[qv.value] = ds_rd<j * 64>(b2a);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 x4 = bf4(xr[q4]), bv = __builtin_bit_cast(f32x4, bq[q4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) out[r][j0 + q4][e] = x4[e] + bv[e];
          asm volatile("" : "+v"(out[r][j0 + q4]));
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    __builtin_amdgcn_sched_barrier(0);

    // ---- chunk pipeline (k_mlp_fused_v3's): iteration c = fc1 of chunk c, then fc2 of chunk c - 1 with the GELU of chunk c between its MFMAs
    auto fc1 = [&](int c) {
      const unsigned a0 = w1e0 + o_cur, a1 = w1e1 + o_cur;
      const unsigned bia = b1lane + (unsigned)(c * (V3_FC * 4));
      u32x4 bi[2], f1[3][2];
      bi[0] = ds_rd<0>(bia); bi[1] = ds_rd<16>(bia);
      f1[0][0] = ds_rd<0>(a0); f1[0][1] = ds_rd<16 * D * 2>(a0);
      f1[1][0] = ds_rd<0>(a1); f1[1][1] = ds_rd<16 * D * 2>(a1);
      __builtin_amdgcn_sched_barrier(0);
      static_for<KT>([&](auto ksv) {
        constexpr int ks = ksv.value, cur = ks % 3, nxt = (ks + 2) % 3;
        wait_tie2<(ks + 1 < KT) ? 2 : 0>(f1[cur][0], f1[cur][1]);
        if constexpr (ks == 0) asm volatile("" : "+v"(bi[0]), "+v"(bi[1]));
        static_for<2>([&](auto tv) {
          constexpr int t = tv.value;
          const bf16x8 af = __builtin_bit_cast(bf16x8, f1[cur][t]);
#pragma unroll
          for (int r = 0; r < NT; ++r) acc[r][t] = mma(af, hf[r][ks], ks == 0 ? __builtin_bit_cast(f32x4, bi[t]) : acc[r][t]);
          if constexpr (ks + 2 < KT) f1[nxt][t] = ds_rd<t * 16 * D * 2 + ((ks + 2 < KT ? ks + 2 : 0) >> 1) * 128>((ks & 1) ? a1 : a0);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    };
    auto fc2_gelu = [&](auto fc2v, auto geluv) {
      constexpr bool FC2 = decltype(fc2v)::value, GELU = decltype(geluv)::value;
      const unsigned w2a = w2lane + o_nn;
      u32x4 f2[6];
      if constexpr (FC2) {
        static_for<6>([&](auto jv) { f2[jv.value] = ds_rd<jv.value * 1024>(w2a); });
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<12>([&](auto jv) {
        constexpr int j = jv.value, slot = j % 6;
        if constexpr (FC2) {
          constexpr int younger = (j < 6) ? 5 : 11 - j;
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f2[slot]) : "n"(younger));
          const bf16x8 af = __builtin_bit_cast(bf16x8, f2[slot]);
#pragma unroll
          for (int r = 0; r < NT; ++r) out[r][j] = mma(af, uf[r], out[r][j]);
          if constexpr (j + 6 < 12) f2[slot] = ds_rd<(j + 6 < 12 ? j + 6 : 0) * 1024>(w2a);
        }
        if constexpr (GELU && j < 8) {
          if constexpr (NT == 2) {                                   // units 0..7 carry two values each, as in k_mlp_fused_v3
            constexpr int r = j >> 2, t = (j >> 1) & 1, e0 = (j & 1) * 2;
            acc[r][t][e0] = Gelu<T>::f(acc[r][t][e0]);
            acc[r][t][e0 + 1] = Gelu<T>::f(acc[r][t][e0 + 1]);
          } else {                                                   // one MFMA per unit: one value each
            constexpr int t = j >> 2, e = j & 3;
            acc[0][t][e] = Gelu<T>::f(acc[0][t][e]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (GELU) {
#pragma unroll
        for (int r = 0; r < NT; ++r) uf[r] = pack8(acc[r][0], acc[r][1]);
      }
    };
    auto rotate = [&]() { const unsigned t = o_cur; o_cur = o_nxt; o_nxt = o_nn; o_nn = t; };
    // requests of iteration c: W1 of chunk c + 2, W2 of chunk c + 1 (both wrap into the next pass)
    auto request = [&](int c) {
      const int c1 = c + 2 < nch ? c + 2 : c + 2 - nch, c2 = c + 1 < nch ? c + 1 : 0;
      dma(std::integral_constant<int, LD ? P_ND : 1>{}, c1, o_nn, c2, o_nxt);
    };
    __builtin_amdgcn_s_barrier();                                    // every wave has its rows in registers: the slots may be refilled
    __builtin_amdgcn_sched_barrier(0);
    // vmcnt retires in order.  A loader's queue per iteration c: [W(c): 5][R(c): 6 if c < 4]; the wait at the end of iteration c must
    // cover W(c - 1) and may leave everything younger in flight.  The other waves: [W(c): 1].
    {
      request(0);
      if constexpr (LD) rowdma(npb, ncnt, 0);                        // the next pass's rows, a slot per iteration
      __builtin_amdgcn_sched_barrier(0);
      fc1(0);
      fc2_gelu(std::false_type{}, std::true_type{});
      wait_vm<LD ? P_ND + P_NR : 1>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rotate();
    }
#pragma nounroll
    for (int c = 1; c < 4; ++c) {
      request(c);
      if constexpr (LD) rowdma(npb, ncnt, c);
      __builtin_amdgcn_sched_barrier(0);
      fc1(c);
      fc2_gelu(std::true_type{}, std::true_type{});
      wait_vm<LD ? P_ND + 2 * P_NR : 1>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rotate();
    }
#pragma nounroll
    for (int c = 4; c < nch; ++c) {
      request(c);
      __builtin_amdgcn_sched_barrier(0);
      fc1(c);
      fc2_gelu(std::true_type{}, std::true_type{});
      if (LD && c == 4) wait_vm<LD ? P_ND + P_NR : 1>(); else wait_vm<LD ? P_ND : 1>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rotate();
    }
    fc2_gelu(std::true_type{}, std::false_type{});

    // ---- out = d1 * ((x1 + b2) + mlp) + d0 * x_prev, rounded once; next_h = LayerNorm of the rounded row
    static_for<NT>([&](auto rv) {
      constexpr int r = rv.value;
      const int row = (r ? mb1 : mb0) + li;
      const bool ok = row < a.M;
      // stores: lanes g, g ^ 1 exchange halves (v_permlane16_swap) so that a lane holds 8 consecutive columns -- of tile j (g even) or
      // tile j + 1 (g odd): 16 bytes per lane, 64-byte row pieces, half the store instructions of the accumulator layout
      char* orow = reinterpret_cast<char*>(a.out) + (size_t)(ok ? row : 0) * (D * 2) + ((g & 1) * 16 + (g >> 1) * 8) * 2;
#pragma unroll
      for (int j = 0; j < D / 16; j += 2) {
        u32x2 q[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 o = out[r][j + t];
          if (a.gate) {
            const f32x4 xp = bf4(*reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(a.x_prev) + (size_t)(ok ? row : 0) * D + (j + t) * 16 + g * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(d1, o[e], __builtin_fmaf(d0, xp[e], 0.0f));
          }
          q[t][0] = pack_bf16x2(o[0], o[1]); q[t][1] = pack_bf16x2(o[2], o[3]);
          out[r][j + t] = bf4(q[t]);
        }
        const auto s0_ = __builtin_amdgcn_permlane16_swap(q[0][0], q[1][0], false, false);
        const auto s1_ = __builtin_amdgcn_permlane16_swap(q[0][1], q[1][1], false, false);
        if (ok) *reinterpret_cast<u32x4*>(orow + j * 32) = u32x4{s0_[0], s1_[0], s0_[1], s1_[1]};
      }
      if (a.next_h) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < D / 16; ++j) s += (out[r][j][0] + out[r][j][1]) + (out[r][j][2] + out[r][j][3]);
        s = sum_rows4(s);
        const float mean = s * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < D / 16; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float dd = out[r][j][e] - mean; q += dd * dd; }
        q = sum_rows4(q);
        const float rstd = rsqrtf(q * (1.0f / D) + a.eps);
        if (ok && a.next_mean && g == 0) { a.next_mean[row] = mean; a.next_rstd[row] = rstd; }
        char* hrow = reinterpret_cast<char*>(a.next_h) + (size_t)(ok ? row : 0) * (D * 2) + ((g & 1) * 16 + (g >> 1) * 8) * 2;
        static_for<D / 64>([&](auto jqv) {
          constexpr int j0 = jqv.value * 4;
          u32x4 gq[4], bq[4];
          static_for<4>([&](auto qv) { gq[qv.value] = ds_rd<(j0 + qv.value) * 64>(nga); bq[qv.value] = ds_rd<(j0 + qv.value) * 64>(nba); });
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gq[0]), "+v"(gq[1]), "+v"(gq[2]), "+v"(gq[3]), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
          u32x2 pk[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 gm = __builtin_bit_cast(f32x4, gq[q4]), bt = __builtin_bit_cast(f32x4, bq[q4]);
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (out[r][j0 + q4][e] - mean) * rstd * gm[e] + bt[e];
            pk[q4][0] = pack_bf16x2(y[0], y[1]); pk[q4][1] = pack_bf16x2(y[2], y[3]);
          }
#pragma unroll
          for (int q4 = 0; q4 < 4; q4 += 2) {
            const auto s0_ = __builtin_amdgcn_permlane16_swap(pk[q4][0], pk[q4 + 1][0], false, false);
            const auto s1_ = __builtin_amdgcn_permlane16_swap(pk[q4][1], pk[q4 + 1][1], false, false);
            if (ok) *reinterpret_cast<u32x4*>(hrow + (j0 + q4) * 32) = u32x4{s0_[0], s1_[0], s0_[1], s1_[1]};
          }
        });
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int p = 0; p < npass; ++p) {
    const int npb = pb + cnt, left = npass - 1 - p;
    const int ncnt = left > 0 ? (t1 - npb + left - 1) / left : 0;
    if (loader) {
      if (mb1 < a.M) pass(std::integral_constant<int, 2>{}, std::true_type{}, npb, ncnt);
      else pass(std::integral_constant<int, 1>{}, std::true_type{}, npb, ncnt);
    } else {
      if (mb1 < a.M) pass(std::integral_constant<int, 2>{}, std::false_type{}, npb, ncnt);
      else pass(std::integral_constant<int, 1>{}, std::false_type{}, npb, ncnt);
    }
    pb = npb; cnt = ncnt;
    mb0 = tile_rows(pb, cnt, w); mb1 = tile_rows(pb, cnt, P_NW + w);
  }
  wait_vm<0>();
}

}  // namespace

extern "C" int uvc_mlp_fused_supported(int32_t D_, int32_t F, int32_t dtype) { return D_ == D && F > 0 && F % 64 == 0 && F <= 1024 && dtype == UVC_BF16; }

extern "C" int uvc_mlp_fused_fwd(const uvc_mlp_args* p, void* stream) {
  if (!p || !p->x || !p->out || !p->gamma || !p->beta || !p->w1 || !p->b1 || !p->w2 || !p->b2) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: null pointer");
  if (p->M <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: empty");
  if (!uvc_mlp_fused_supported(p->D, p->F, UVC_BF16)) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_mlp_fused_fwd: needs D == 192, F % 64 == 0 and F <= 1024");
  if ((((uintptr_t)p->x | (uintptr_t)p->out | (uintptr_t)p->w1 | (uintptr_t)p->w2 | (uintptr_t)p->b1 | (uintptr_t)p->x_prev | (uintptr_t)p->h |
        (uintptr_t)p->u | (uintptr_t)p->gp) & 15) != 0)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: buffers must be 16-byte aligned");
  const bool train = p->h || p->u || p->gp || p->mean || p->rstd;
  if (train && (!p->h || !p->u || !p->gp || !p->mean || !p->rstd)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: training needs h, mean, rstd, gp and u");
  if ((p->gate != nullptr) != (p->x_prev != nullptr)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: gate and x_prev come together");
  if (p->next_h && (!p->next_gamma || !p->next_beta || (((uintptr_t)p->next_h | (uintptr_t)p->next_gamma | (uintptr_t)p->next_beta) & 15) != 0))
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: next_h needs 16-byte aligned next_gamma, next_beta");
  if ((p->next_mean != nullptr) != (p->next_rstd != nullptr) || (p->next_mean && !p->next_h))
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_mlp_fused_fwd: next_mean and next_rstd come together, with next_h");
  hipStream_t st = (hipStream_t)stream;
  {
#define MLP_ONE(TR_, RL_) { UVC_MAX_LDS(V3_LDS, k_mlp_fused_v3<TR_, RL_>); k_mlp_fused_v3<TR_, RL_><<<ceil_div(p->M, V3_ROWS), V3_NTH, V3_LDS, st>>>(*p); }
    if (p->rows_lowp && !train && p->M >= P_MIN_ROWS && p->F >= 256) {
      const int ntiles = ceil_div(p->M, 16);
      UVC_MAX_LDS(P_LDS, k_mlp_fused_p);
      // one workgroup per CU (r5, profiles/r5j: on 224 / 192 / 128 workgroups the step does not move -- the teacher's MLP hides under the student's forward)
      k_mlp_fused_p<<<std::min(256, ceil_div(ntiles, P_PASS)), P_NTH, P_LDS, st>>>(*p, ntiles);
    } else if (p->rows_lowp) { if (train) MLP_ONE(true, true) else MLP_ONE(false, true) }
    else { if (train) MLP_ONE(true, false) else MLP_ONE(false, false) }
#undef MLP_ONE
    UVC_CHECK_LAUNCH();
    return UVC_OK;
  }
}
