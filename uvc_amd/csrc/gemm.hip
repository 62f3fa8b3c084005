// MFMA GEMMs for the DeiT step on gfx950.
//   uvc_gemm_nt : C[M,N] = epi( A[M,K] . B[N,K]^T )       forward Linear / dgrad (with W^T copies)
//   uvc_gemm_tn : C[N1,N2] = beta*C + alpha * A[M,N1]^T . B[M,N2]   wgrad, split over M, deterministic
// Compute type T = bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) or float
// (v_mfma_f32_16x16x4_f32, bit-exact fmaf chain) -- the latter is the 1e-3 parity mode.
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves as 2x2; the block tile is staged through
// LDS in 16-byte chunks with register prefetch of the next K tile; rows are padded by 16 B (NT)
// or 32 B (TN) so ds_read_b128 / ds_read_b64_tr_b16 fragment reads are bank-conflict free.
// Operands are swapped in the MFMA (a = B-fragment, b = A-fragment) so every lane ends up with 4
// CONSECUTIVE output columns of one row -> 16-byte (fp32) / 8-byte (bf16) epilogue accesses.
#include <type_traits>
#include "common.h"
#include "../../include/uvc_kernels.h"

// ------------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int CH = 8;     // elements per 16-byte chunk
  static constexpr int KSTEP = 32; // k elements per MFMA step (4 lane groups x 1 chunk)
  typedef bf16x8 Frag;
  static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int CH = 4;
  static constexpr int KSTEP = 16;
  typedef f32x4 Frag;
  static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
  }
};

// 16-byte chunk of compute type T loaded from a global array of type TS (convert on load)
template <typename TS, typename T> struct ChunkLoad;
template <typename T> struct ChunkLoad<T, T> {
  static __device__ __forceinline__ u32x4 ld(const T* p) { return *reinterpret_cast<const u32x4*>(p); }
};
template <> struct ChunkLoad<float, bf16_t> {   // 8 floats -> 8 bf16
  static __device__ __forceinline__ u32x4 ld(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    u32x4 r;
    r[0] = pack_bf16x2(a[0], a[1]); r[1] = pack_bf16x2(a[2], a[3]);
    r[2] = pack_bf16x2(b[0], b[1]); r[3] = pack_bf16x2(b[2], b[3]);
    return r;
  }
};

template <typename T> __device__ __forceinline__ typename Mma<T>::Frag lds_frag(const char* p) {
  return *reinterpret_cast<const typename Mma<T>::Frag*>(p);
}

// ================================================================================================
//                                            NT
// ================================================================================================
constexpr int NT_BM = 128, NT_BN = 128;
// LDS row stride in bytes (8 chunks + 32 B pad).  ds_read_b128 is served in the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ...: a fragment read (row = lane & 15, chunk = lane >> 4) is conflict-free only when the stride
// in words is 8, 24, 40 or 56 mod 64; the obvious +16 B pad (36 mod 64) costs 2 LDS cycles per group.
constexpr int NT_ROWB = 128 + 32;

template <typename T> struct OutIO;
template <> struct OutIO<float> {
  static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
};
template <> struct OutIO<bf16_t> {
  static __device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
    u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = r;
  }
  static __device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(p);
    f32x4 v; v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u);
    v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
    return v;
  }
};

struct NtArgs {
  const void* A; const void* B; void* C; void* C2;
  const float* bias; const void* R; const void* R2; const void* aux; const float* dptr;      // R, R2: the residual rows, element type of C
  int M, N, K, lda, ldb, ldc, ldr, ldaux;
  float alpha;
  const float* alpha_ptr;   // optional device scalar multiplied into alpha
  // optional second output of the N == 192 residual epilogues: LayerNorm of the output rows (uvc_gemm_nt_args.ln_*)
  const float* ln_gamma; const float* ln_beta; void* ln_out; float* ln_mean; float* ln_rstd; float ln_eps;
  int no_dma;               // uvc_gemm_nt_args.force_generic == 2: the register-staged streaming kernels instead of the LDS-DMA rings
};

// vector of VN consecutive outputs in the coalesced epilogue layout
template <typename TC> struct OutVec;
template <> struct OutVec<float> {
  static constexpr int VN = 4;
  static __device__ __forceinline__ void st(float* p, const float* v) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};
template <> struct OutVec<bf16_t> {
  static constexpr int VN = 8;
  static __device__ __forceinline__ void st(bf16_t* p, const float* v) {
    u32x4 r;
    r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]); r[2] = pack_bf16x2(v[4], v[5]); r[3] = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<u32x4*>(p) = r;
  }
  static __device__ __forceinline__ void st_nt(bf16_t* p, const float* v) {      // streaming output: not read again before the caches have turned over
    u32x4 r;
    r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]); r[2] = pack_bf16x2(v[4], v[5]); r[3] = pack_bf16x2(v[6], v[7]);
    __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
  }
};
template <typename T, int VN> __device__ __forceinline__ void load_vec(const T* p, float* v);
template <> __device__ __forceinline__ void load_vec<float, 4>(const float* p, float* v) {
  const f32x4 r = *reinterpret_cast<const f32x4*>(p); v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
}
template <> __device__ __forceinline__ void load_vec<float, 8>(const float* p, float* v) { load_vec<float, 4>(p, v); load_vec<float, 4>(p + 4, v + 4); }
template <> __device__ __forceinline__ void load_vec<bf16_t, 8>(const bf16_t* p, float* v) {
  const u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(r[e] << 16); v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void load_vec<bf16_t, 4>(const bf16_t* p, float* v) {
  const u32x2 r = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(r[0] << 16); v[1] = __uint_as_float(r[0] & 0xffff0000u);
  v[2] = __uint_as_float(r[1] << 16); v[3] = __uint_as_float(r[1] & 0xffff0000u);
}

// Residual operands have the element type of C (float32 stream: 4 floats per 16 bytes; bf16 stream: 8 per 16 bytes, 4 per 8 bytes)
template <typename TC> struct Resid;
template <> struct Resid<float> {
  typedef u32x4 Raw4;                            // four consecutive columns
  static __device__ __forceinline__ Raw4 ld4(const void* base, size_t elem) { return *reinterpret_cast<const u32x4*>(reinterpret_cast<const float*>(base) + elem); }
  static __device__ __forceinline__ f32x4 f4(const Raw4& r) { return __builtin_bit_cast(f32x4, r); }
  // VN = 4 consecutive columns in 16 bytes
  static __device__ __forceinline__ void get(const u32x4& r, float* v) { const f32x4 f = __builtin_bit_cast(f32x4, r); v[0] = f[0]; v[1] = f[1]; v[2] = f[2]; v[3] = f[3]; }
};
template <> struct Resid<bf16_t> {
  typedef u32x2 Raw4;
  static __device__ __forceinline__ Raw4 ld4(const void* base, size_t elem) { return *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(base) + elem); }
  static __device__ __forceinline__ f32x4 f4(const Raw4& r) {
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
  }
  // VN = 8 consecutive columns in 16 bytes
  static __device__ __forceinline__ void get(const u32x4& r, float* v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(r[e] << 16); v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
  }
};

// Epilogue arithmetic with the rounding points spelled out (hipcc contracts a*b + c*d differently from kernel to kernel):
// every NT kernel computes the same bits for the same accumulator, so a row's result does not depend on which kernel the
// problem size selects (tests/test_fullsize_gpu.py compares a 512-image run with an 8-image run bit for bit).
__device__ __forceinline__ float epi_scale_bias(float acc, float alpha, float bias) { return __builtin_fmaf(acc, alpha, bias); }
__device__ __forceinline__ float epi_gate_mix(float v, float r2, float d0, float d1) { return __builtin_fmaf(d1, v, __builtin_fmaf(d0, r2, 0.0f)); }
constexpr int EP_LD = 68;   // floats per staged accumulator row (64 + 4 pad: conflict-free 16-byte LDS writes)

// Epilogue of ROWS staged accumulator rows x 64 columns of one wave (row r = output row mrow0 + r, columns ncol0 .. ncol0 + 63):
// scale / bias / residual / gate mix / activation, then 16-byte row-contiguous stores.  Shared by the NT kernels that leave through an
// LDS transpose, so that a row's bits do not depend on the kernel that produced its accumulator.
// Staging layouts: SWZ = false: [ROWS][EP_LD] floats (padded rows); SWZ = true: [ROWS][64] floats, the 16-byte column group cg of row r
// at group cg ^ (r & 7) (no padding: 4 KB per 16 rows; conflict-free for the accumulator writes and for these reads).
// bias_v: the VN bias values of this lane's columns (0 where the epilogue has no bias or the column is past N), loaded once per tile.
template <bool SWZ> __device__ __forceinline__ float* stg_at(float* stg, int r, int cg) {
  return SWZ ? stg + r * 64 + ((cg ^ (r & 7)) << 2) : stg + r * EP_LD + (cg << 2);
}
template <typename TC, int EPI> __device__ __forceinline__ void nt_load_bias(const NtArgs& g, int lane, int ncol0, float* bias_v) {
  constexpr int VN = OutVec<TC>::VN, LPR = 64 / VN;
  constexpr bool has_bias = EPI == UVC_EPI_BIAS || EPI == UVC_EPI_BIAS_GELU || EPI == UVC_EPI_BIAS_GELU_OUT || EPI == UVC_EPI_BIAS_RESID ||
                            EPI == UVC_EPI_BIAS_RESID_GATE || EPI == UVC_EPI_BIAS_GELU_GRAD;
  const int n = ncol0 + (lane % LPR) * VN;
#pragma unroll
  for (int e = 0; e < VN; ++e) bias_v[e] = (has_bias && n + e < g.N) ? g.bias[n + e] : 0.f;
}
// Operand rows of an epilogue (R, R2, aux) requested AHEAD of the rows they meet: the epilogue of a big tile is a chain of [transpose 16 rows
// through LDS | load their operands | arithmetic | store]; with the loads inside the chain every 16-row step was a memory round trip (dfc2 x GELU'
// of DeiT-Base: 24 us of epilogue per 256 x 256 tile against 16 us of k-loop).  bf16 C only (16 bytes = the lane's 8 columns); a lane whose
// columns are not a whole aligned vector (ragged N, odd leading dimensions) keeps loading in place.
template <int NIT> struct EpiPre { u32x4 r[NIT], r2[NIT], ax[NIT]; };
template <int EPI> struct EpiOperands {
  static constexpr bool R = EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE, R2 = EPI == UVC_EPI_BIAS_RESID_GATE,
                        AUX = EPI == UVC_EPI_DGELU || EPI == UVC_EPI_MUL_AUX;
  static constexpr int COUNT = (R ? 1 : 0) + (R2 ? 1 : 0) + (AUX ? 1 : 0);
};
template <int EPI, int ROWS>
__device__ __forceinline__ void nt_epilogue_prefetch(const NtArgs& g, int lane, int mrow0, int ncol0, EpiPre<ROWS / 8>& P) {
  const int cc = (lane & 7) * 8, n = ncol0 + cc;
  const bool nfull = (n + 8 <= g.N) && ((g.ldc % 8) == 0);
#pragma unroll
  for (int it = 0; it < ROWS / 8; ++it) {
    const int m = mrow0 + it * 8 + (lane >> 3);
    if (m < g.M && nfull) {
      const size_t mo = (size_t)m;
      if (EpiOperands<EPI>::R && (g.ldr % 8) == 0) P.r[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.R) + mo * g.ldr + n);
      if (EpiOperands<EPI>::R2 && (g.ldr % 8) == 0) P.r2[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.R2) + mo * g.ldr + n);
      if (EpiOperands<EPI>::AUX && (g.ldaux % 8) == 0) P.ax[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(g.aux) + mo * g.ldaux + n);
    }
  }
}
template <typename T, typename TC, int EPI, int ROWS = 32, bool SWZ = false, bool PRE = false>
__device__ __forceinline__ void nt_epilogue_rows(const NtArgs& g, float* stg, int lane, int mrow0, int ncol0, float alpha, float d0, float d1,
                                                 const float* bias_v, const EpiPre<ROWS / 8>* pre = nullptr) {
  static_assert(!PRE || (sizeof(TC) == 2 && sizeof(T) == 2), "operand prefetch: bf16 C and operands only");
  constexpr int VN = OutVec<TC>::VN;
  constexpr int LPR = 64 / VN;                  // lanes per 64-column row
  constexpr int RPI = 64 / LPR;                 // rows per wave instruction
  TC* __restrict__ C = reinterpret_cast<TC*>(g.C);
  const int cc = (lane % LPR) * VN;
  const int n = ncol0 + cc;
  const bool nfull = (n + VN <= g.N) && ((g.ldc % VN) == 0);
#pragma unroll
  for (int it = 0; it < ROWS / RPI; ++it) {
    const int r = it * RPI + lane / LPR;
    const int m = mrow0 + r;
    float v[VN];
    load_vec<float, 4>(stg_at<SWZ>(stg, r, cc >> 2), v);
    if (VN == 8) load_vec<float, 4>(stg_at<SWZ>(stg, r, (cc >> 2) + 1), v + 4);
    if (m < g.M && n < g.N) {
      const size_t mo = (size_t)m;
#pragma unroll
      for (int e = 0; e < VN; ++e) v[e] = epi_scale_bias(v[e], alpha, bias_v[e]);
      if (EPI == UVC_EPI_BIAS_GELU_OUT) {
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = Gelu<T>::f(v[e]);
      }
      if (EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE) {
        const TC* rp = reinterpret_cast<const TC*>(g.R) + mo * g.ldr + n;
        float rv[VN];
        if (nfull && (g.ldr % VN) == 0) { if constexpr (PRE) Resid<TC>::get(pre->r[it], rv); else load_vec<TC, VN>(rp, rv); }
        else {
#pragma unroll
          for (int e = 0; e < VN; ++e) rv[e] = (n + e < g.N) ? ElemIO<TC>::load(rp + e) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] += rv[e];
      }
      if (EPI == UVC_EPI_BIAS_RESID_GATE) {
        const TC* rp = reinterpret_cast<const TC*>(g.R2) + mo * g.ldr + n;
        float rv[VN];
        if (nfull && (g.ldr % VN) == 0) { if constexpr (PRE) Resid<TC>::get(pre->r2[it], rv); else load_vec<TC, VN>(rp, rv); }
        else {
#pragma unroll
          for (int e = 0; e < VN; ++e) rv[e] = (n + e < g.N) ? ElemIO<TC>::load(rp + e) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = epi_gate_mix(v[e], rv[e], d0, d1);
      }
      if (EPI == UVC_EPI_DGELU || EPI == UVC_EPI_MUL_AUX) {
        const T* ap = reinterpret_cast<const T*>(g.aux) + mo * g.ldaux + n;
        float av[VN];
        if (nfull && (g.ldaux % VN) == 0) { if constexpr (PRE) Resid<TC>::get(pre->ax[it], av); else load_vec<T, VN>(ap, av); }
        else {
#pragma unroll
          for (int e = 0; e < VN; ++e) av[e] = (n + e < g.N) ? ElemIO<T>::load(ap + e) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] *= (EPI == UVC_EPI_MUL_AUX) ? av[e] : Gelu<T>::g(av[e]);
      }
      float u[VN];
      if (EPI == UVC_EPI_BIAS_GELU_GRAD) {
#pragma unroll
        for (int e = 0; e < VN; ++e) { float fo, go; Gelu<T>::fg(v[e], fo, go); u[e] = fo; v[e] = go; }
      }
      TC* cp = C + mo * g.ldc + n;
      // (GELU' / GELU of the training forward: streaming outputs, non-temporal -- DeiT-Base 23.31 -> 23.06 ms, profiles/r5zz_ab_nt_wide.txt)
      if constexpr (EPI == UVC_EPI_BIAS_GELU_GRAD && sizeof(TC) == 2) { if (nfull) OutVec<TC>::st_nt(cp, v); }
      else
      if (nfull) OutVec<TC>::st(cp, v);
      if (nfull) {}
      else {
#pragma unroll
        for (int e = 0; e < VN; ++e) if (n + e < g.N) ElemIO<TC>::store(cp + e, v[e]);
      }
      if (EPI == UVC_EPI_BIAS_GELU || EPI == UVC_EPI_BIAS_GELU_GRAD) {
        TC* c2 = reinterpret_cast<TC*>(g.C2) + mo * g.ldc + n;
        if (EPI == UVC_EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < VN; ++e) u[e] = Gelu<T>::f(v[e]);
        }
        if constexpr (EPI == UVC_EPI_BIAS_GELU_GRAD && sizeof(TC) == 2) { if (nfull) OutVec<TC>::st_nt(c2, u); }
        else
        if (nfull) OutVec<TC>::st(c2, u);
        if (nfull) {}
        else {
#pragma unroll
          for (int e = 0; e < VN; ++e) if (n + e < g.N) ElemIO<TC>::store(c2 + e, u[e]);
        }
      }
    }
  }
}

template <typename TA, typename T, typename TC, int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(NtArgs g) {
  typedef Mma<T> MM;
  constexpr int CH = MM::CH;
  constexpr int BK = 8 * CH;                    // elements per K tile (128 B of T per row)
  __shared__ __attribute__((aligned(16))) char smem[(NT_BM + NT_BN) * NT_ROWB];
  char* sA = smem;
  char* sB = smem + NT_BM * NT_ROWB;
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(g.B);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  // XCD-aware tile order.  Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8),
  // each with a private L2.  All N tiles of one M tile get block ids that are congruent mod 8 and
  // adjacent in dispatch order, so the A panel is fetched from HBM once and re-read from that XCD's L2.
  const int ntn = (g.N + NT_BN - 1) / NT_BN;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int bn = q % ntn, bm = (q / ntn) * 8 + xcd;
  const int m0 = bm * NT_BM, n0 = bn * NT_BN;
  if (m0 >= g.M) return;
  const int lc = tid & 7, lr = tid >> 3;        // chunk column / first row of this thread's loads

  u32x4 ra[4], rb[4];
  auto gload = [&](int k0) {
    const int kc = k0 + lc * CH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lr + 32 * i;
      const int m = m0 + row, n = n0 + row;
      u32x4 z = {0u, 0u, 0u, 0u};
      ra[i] = (m < g.M && kc < g.K) ? ChunkLoad<TA, T>::ld(A + (size_t)m * g.lda + kc) : z;
      rb[i] = (n < g.N && kc < g.K) ? ChunkLoad<T, T>::ld(B + (size_t)n * g.ldb + kc) : z;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lr + 32 * i;
      *reinterpret_cast<u32x4*>(sA + row * NT_ROWB + lc * 16) = ra[i];
      *reinterpret_cast<u32x4*>(sB + row * NT_ROWB + lc * 16) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BK - 1) / BK;
  gload(0);
  lstore();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typename MM::Frag fa[4], fb[4];
      const int coff = (ks * 4 + (lane >> 4)) * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = lds_frag<T>(sA + (wm * 64 + i * 16 + (lane & 15)) * NT_ROWB + coff);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = lds_frag<T>(sB + (wn * 64 + j * 16 + (lane & 15)) * NT_ROWB + coff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < nk) {
      __syncthreads();
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue.  The MFMA leaves lane (l) with row m = (l&15), columns (l>>4)*4+{0..3} of each 16x16
  // tile.  Each wave transposes its 64x64 sub-tile through LDS, 32 rows at a time, so that global
  // accesses are whole 128-byte (bf16) / 256-byte (fp32) row segments: 16 bytes per lane.
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (EPI == UVC_EPI_BIAS_RESID_GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  float* stg = reinterpret_cast<float*>(smem) + w * (32 * EP_LD);
  float bias_v[OutVec<TC>::VN];
  nt_load_bias<TC, EPI>(g, lane, n0 + wn * 64, bias_v);
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(stg + (ii * 16 + (lane & 15)) * EP_LD + j * 16 + (lane >> 4) * 4) = acc[2 * h + ii][j];
    __syncthreads();
    nt_epilogue_rows<T, TC, EPI>(g, stg, lane, m0 + wm * 64 + h * 32, n0 + wn * 64, alpha, d0, d1, bias_v);
    __syncthreads();
  }
}

// ================================================================================================
//                     NT, weights-stationary / activation-streaming (bf16, K <= 192)
// ================================================================================================
// The DeiT GEMMs are skinny: M = B*197 rows (1e5) against K, N of a few hundred, i.e. HBM-bound
// streaming of A and C with a weight matrix that fits in registers.  Each wave keeps the MFMA B
// fragments of its 64 output columns in VGPRs for its whole life (KT*4 fragments), the workgroup
// (3 waves = 192 columns) walks M persistently in 64-row tiles, A is staged once per tile through a
// double-buffered LDS image (register prefetch of tile i+1 under the MFMAs of tile i) and shared by
// the three waves, and C leaves through a wave-private LDS transpose as whole 128/256-byte rows.
// LDS traffic per MFMA is half of the generic kernel's (no B reads) and A is read exactly once per
// 192-column group; groups of one tile sequence are placed on one XCD so they share its L2.
constexpr int WS_BM = 64;

// NJ = 16-column accumulator tiles per wave: 4 (64 columns, 96 VGPRs of W at K = 192, two waves per SIMD) or 2 (32 columns, 48 VGPRs,
// eight waves per workgroup, four per SIMD: the GELU epilogues of fc1 overlap its stores better)
// KT = 12 (K = 384: DeiT-Small / T2T-ViT widths) runs as 6 waves x 32 columns: the wave's W slice is 96 VGPRs again, the two A images
// (64 rows x 800 B) and the transpose buffers take 116 KB of dynamic LDS, one workgroup per CU.
template <typename TA, typename TC, int EPI, int KT, int WS_NW, int NJ = 4>
__global__ __launch_bounds__(64 * WS_NW, (KT > 6 && WS_NW > 6) ? 1 : (NJ == 4 || KT > 6) ? 2 : 4) void k_gemm_ws(NtArgs g, int ngroups, int nslots) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int ROWB = KT * 64 + 32;             // bytes per staged A row (KT*32 bf16 + pad; words = 8 or 40 mod 64, see NT_ROWB)
  constexpr int CPR = KT * 4;                    // 16-byte chunks per row
  constexpr int NLD = (WS_BM * CPR + 64 * WS_NW - 1) / (64 * WS_NW);
  constexpr int CW = 16 * NJ, EPW = CW + 4;         // columns per wave; floats per staged accumulator row
  constexpr int VN = OutVec<TC>::VN, LPR = CW / VN, RPI = 64 / LPR;
  constexpr bool DYN = KT > 6;
  constexpr int IMGB = WS_BM * ROWB;
  extern __shared__ __attribute__((aligned(16))) char ws_dyn[];
  __shared__ __attribute__((aligned(16))) char sA_st[DYN ? 16 : 2 * IMGB];
  __shared__ __attribute__((aligned(16))) float sStage_st[DYN ? 4 : WS_NW * 16 * EPW];
  char* const sAb = DYN ? ws_dyn : sA_st;
  float* const sStg = DYN ? reinterpret_cast<float*>(ws_dyn + 2 * IMGB) : sStage_st;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, gq = lane >> 4, li = lane & 15;
  const int L = blockIdx.x;
  const int grp = (L >> 3) % ngroups, slot = (L / (8 * ngroups)) * 8 + (L & 7);
  const int n0 = grp * CW * WS_NW + w * CW;
  const bool active = n0 < g.N;                  // N % 64 == 0: a wave is either fully inside or idle
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.B);
  TC* __restrict__ C = reinterpret_cast<TC*>(g.C);
  const int ntiles = (g.M + WS_BM - 1) / WS_BM;

  typename MM::Frag bf[NJ][KT];
  if (active) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int ks = 0; ks < KT; ++ks)
        bf[j][ks] = __builtin_bit_cast(typename MM::Frag, *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + j * 16 + li) * g.ldb + (ks * 4 + gq) * 8));
  }
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (EPI == UVC_EPI_BIAS_RESID_GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  const int cc = (lane % LPR) * VN;
  const int n = n0 + cc;
  float bias_v[VN];
#pragma unroll
  for (int e = 0; e < VN; ++e) bias_v[e] = 0.f;
  if (active && (EPI == UVC_EPI_BIAS || EPI == UVC_EPI_BIAS_GELU || EPI == UVC_EPI_BIAS_GELU_OUT || EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE ||
                 EPI == UVC_EPI_BIAS_GELU_GRAD || EPI == UVC_EPI_BIAS_GELU_GRAD_Q8)) {
#pragma unroll
    for (int e = 0; e < VN; ++e) bias_v[e] = g.bias[n + e];
  }

  u32x4 ra[NLD];
  auto gload = [&](int tile) {
    const int m0 = tile * WS_BM;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + 64 * WS_NW * i, row = id / CPR, c = id % CPR;
      const int m = m0 + row;
      ra[i] = (row < WS_BM && m < g.M) ? ChunkLoad<TA, T>::ld(A + (size_t)m * g.lda + c * 8) : z;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + 64 * WS_NW * i, row = id / CPR, c = id % CPR;
      if (row < WS_BM) *reinterpret_cast<u32x4*>(sAb + buf * IMGB + row * ROWB + c * 16) = ra[i];
    }
  };

  // epilogue operands (residual rows / GELU pre-activation) are prefetched one 16-row sub-tile ahead into
  // registers, so their HBM latency hides under the epilogue math of the previous sub-tile and the MFMAs
  // of the current one.  The accumulator transpose buffer is private to a wave: DS operations of one wave
  // execute in order, so only a compiler-level wave barrier separates its writes from its reads.
  constexpr int IT = 16 / RPI;
  struct Epi { u32x4 r[IT]; u32x4 r2[IT]; u32x4 ax[IT]; };      // r, r2: VN residual values of C's element type in 16 bytes
  auto eload = [&](Epi& E, int tile_, int sub_) {
    if (!active) return;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int m = tile_ * WS_BM + sub_ * 16 + it * RPI + lane / LPR;
      const bool ok = m < g.M && tile_ < ntiles;
      const size_t mo = (size_t)(ok ? m : 0);
      if (EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE) E.r[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const TC*>(g.R) + mo * g.ldr + n);
      if (EPI == UVC_EPI_BIAS_RESID_GATE) E.r2[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const TC*>(g.R2) + mo * g.ldr + n);
      if (EPI == UVC_EPI_DGELU || EPI == UVC_EPI_MUL_AUX) E.ax[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.aux) + mo * g.ldaux + n);
      if (EPI == UVC_EPI_MUL_AUX_Q8) {          // 8 one-byte codes of GELU'(a) (include/uvc_kernels.h)
        const u32x2 q = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(g.aux) + mo * g.ldaux + n);
        E.ax[it][0] = q[0]; E.ax[it][1] = q[1];
      }
    }
  };
  auto subtile = [&](int buf_, int m0_, int sub_, const Epi& E) {
    if (!active) return;
    float* stg = sStg + w * (16 * EPW);
    {
      typename MM::Frag fa[KT];
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) fa[ks] = lds_frag<T>(sAb + buf_ * IMGB + (sub_ * 16 + li) * ROWB + (ks * 4 + gq) * 16);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) c = MM::mma(bf[j][ks], fa[ks], c);
        *reinterpret_cast<f32x4*>(stg + li * EPW + j * 16 + gq * 4) = c;
      }
    }
    __builtin_amdgcn_wave_barrier();
    unsigned qc[IT][2];                                     // UVC_EPI_BIAS_GELU_GRAD_Q8: the lane's 8 one-byte codes of each of its rows
#pragma unroll
    for (int it = 0; it < IT; ++it) { qc[it][0] = 0u; qc[it][1] = 0u; }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int r = it * RPI + lane / LPR;
      const int m = m0_ + sub_ * 16 + r;
      float v[VN];
      load_vec<float, VN>(stg + r * EPW + cc, v);
      if (m < g.M) {
        const size_t mo = (size_t)m;
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = epi_scale_bias(v[e], alpha, bias_v[e]);
        if (EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE) {
          float rv[VN];
          Resid<TC>::get(E.r[it], rv);
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] += rv[e];
        }
        if (EPI == UVC_EPI_BIAS_RESID_GATE) {
          float rv[VN];
          Resid<TC>::get(E.r2[it], rv);
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = epi_gate_mix(v[e], rv[e], d0, d1);
        }
        if (EPI == UVC_EPI_DGELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] *= Gelu<T>::g(__uint_as_float(E.ax[it][e] << 16));
            v[2 * e + 1] *= Gelu<T>::g(__uint_as_float(E.ax[it][e] & 0xffff0000u));
          }
        }
        if (EPI == UVC_EPI_MUL_AUX) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] *= __uint_as_float(E.ax[it][e] << 16);
            v[2 * e + 1] *= __uint_as_float(E.ax[it][e] & 0xffff0000u);
          }
        }
        if constexpr (EPI == UVC_EPI_MUL_AUX_Q8) {
          static_assert(VN == 8, "q8 epilogue: bf16 C");
#pragma unroll
          for (int e = 0; e < 8; ++e)            // (byte -> float is one v_cvt_f32_ubyteN)
            v[e] *= __builtin_fmaf((float)((E.ax[it][e >> 2] >> (8 * (e & 3))) & 0xffu), UVC_Q8_STEP, UVC_Q8_LO);
        }
        if (EPI == UVC_EPI_BIAS_GELU_OUT) {
#pragma unroll
          for (int e = 0; e < VN; ++e) v[e] = Gelu<T>::f(v[e]);
        }
        float u[VN];
        if (EPI == UVC_EPI_BIAS_GELU_GRAD || EPI == UVC_EPI_BIAS_GELU_GRAD_Q8) {
#pragma unroll
          for (int e = 0; e < VN; ++e) { float fo, go; Gelu<T>::fg(v[e], fo, go); u[e] = fo; v[e] = go; }
        }
        if constexpr (EPI == UVC_EPI_BIAS_GELU_GRAD_Q8) {
          // GELU'(a) as one byte per activation (include/uvc_kernels.h): 8 codes = 8 bytes per lane, 64-byte row pieces per wave; GELU(a) as before.  Both stream
          // past the caches (read again in the backward / by fc2)
          static_assert(VN == 8, "q8 epilogue: bf16 C2");
          unsigned qw[2] = {0u, 0u};
#pragma unroll
          for (int e = 0; e < 8; ++e)       // v_cvt_pk_u8_f32: round to nearest, saturate to [0, 255], pack into byte e & 3 -- two VALU instructions per code
            qw[e >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[e], 1.0f / UVC_Q8_STEP, -UVC_Q8_LO / UVC_Q8_STEP), e & 3, qw[e >> 2]);
          qc[it][0] = qw[0]; qc[it][1] = qw[1];            // stored behind the loop: 16 bytes per lane (below)
          OutVec<TC>::st_nt(reinterpret_cast<TC*>(g.C2) + mo * g.ldc + n, u);
        } else {
        // GELU'(a) and GELU(a) of the training forward are STREAMING outputs (310 MB per block, read again in the backward / by fc2 after the caches have
        // turned over): non-temporal stores.  Alone 90.2 -> 83.1 us; in the step 11.19 -> 10.96 ms (GELU' only: 79.8 us alone, 11.09 in the step) -- the rest of
        // the step keeps the Infinity Cache (profiles/r5zz_ab_nt_stores.txt).  K = 192 only: at K = 384 (DeiT-Small, T2T-ViT-14) the same stores are within
        // noise or 0.5 % slower (profiles/r5zz_ab_nt_wide.txt)
        if constexpr (EPI == UVC_EPI_BIAS_GELU_GRAD && sizeof(TC) == 2 && KT == 6) OutVec<TC>::st_nt(C + mo * g.ldc + n, v);
        else
        // (also tried, each within +- 0.4 % of the step: non-temporal dA stores of dfc2 x GELU', non-temporal loads of its GELU' operand, nt on the LDS-DMA loads
        //  of the weight gradients and of the attention backward -- profiles/r5zz_ab_nt_sites.txt)
        OutVec<TC>::st(C + mo * g.ldc + n, v);
        if (EPI == UVC_EPI_BIAS_GELU || EPI == UVC_EPI_BIAS_GELU_GRAD) {
          if (EPI == UVC_EPI_BIAS_GELU) {
#pragma unroll
            for (int e = 0; e < VN; ++e) u[e] = Gelu<T>::f(v[e]);
          }
          if constexpr (EPI == UVC_EPI_BIAS_GELU_GRAD && sizeof(TC) == 2 && KT == 6) OutVec<TC>::st_nt(reinterpret_cast<TC*>(g.C2) + mo * g.ldc + n, u);
          else
          OutVec<TC>::st(reinterpret_cast<TC*>(g.C2) + mo * g.ldc + n, u);
        }
        }
      }
    }
    if constexpr (EPI == UVC_EPI_BIAS_GELU_GRAD_Q8) {
      // The codes leave as 16 bytes per lane, ONE store instruction per 16-row sub-tile instead of two of 8 bytes: fc1's epilogue is bound by its store
      // instructions as much as by its bytes (8-byte stores of the codes: the step 10.93 -> 10.90 ms where the tensor not written at all gives 10.70,
      // profiles/r6d, r6b).  Lanes l, l ^ 1 hold neighbouring 8-column groups of the SAME two rows (it = 0, 1): the even lane takes both groups of row 0,
      // the odd lane both groups of row 1 (one DPP quad permute per dword).
      const bool odd = lane & 1;
      if constexpr (IT == 2) {
        static_assert(LPR == 8 && RPI == 8, "q8 epilogue: four waves x 64 columns");
        const unsigned s0 = odd ? qc[0][0] : qc[1][0], s1 = odd ? qc[0][1] : qc[1][1];      // what the partner stores: my codes of ITS row
        const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]: lane ^ 1
        const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
        const int m = m0_ + sub_ * 16 + (odd ? RPI : 0) + lane / LPR;
        if (m < g.M) {
          const u32x4 o = odd ? u32x4{r0, r1, qc[1][0], qc[1][1]} : u32x4{qc[0][0], qc[0][1], r0, r1};
          __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(g.C) + (size_t)m * g.ldc + (n & ~15)));
        }
      } else {
        // eight waves x 32 columns: one row per lane and sub-tile; the even lane of a pair stores both 8-column groups of the row
        static_assert(IT == 1 && LPR == 4, "q8 epilogue: eight waves x 32 columns");
        const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)qc[0][0], 0xB1, 0xF, 0xF, true);
        const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)qc[0][1], 0xB1, 0xF, 0xF, true);
        const int m = m0_ + sub_ * 16 + lane / LPR;
        if (m < g.M && !odd)
          __builtin_nontemporal_store(u32x4{qc[0][0], qc[0][1], r0, r1}, reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(g.C) + (size_t)m * g.ldc + n));
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  int tile = slot;
  if (tile >= ntiles) return;
  Epi E0, E1;
  gload(tile);
  eload(E0, tile, 0);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += nslots) {
    const int next = tile + nslots;
    if (next < ntiles) gload(next);
    const int m0 = tile * WS_BM;
    eload(E1, tile, 1);
    subtile(buf, m0, 0, E0);
    eload(E0, tile, 2);
    subtile(buf, m0, 1, E1);
    eload(E1, tile, 3);
    subtile(buf, m0, 2, E0);
    eload(E0, next, 0);
    subtile(buf, m0, 3, E1);
    if (next < ntiles) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
}

// eligibility of the streaming kernel: bf16 compute, K in {128, 192}, N a multiple of 64, vector-aligned leading dims
static bool ws_ok(const NtArgs& a, int vn) {
  return (a.K == 192 || a.K == 128) && a.N % 64 == 0 && a.ldb == a.K && a.lda % 8 == 0 && a.ldc % vn == 0 && a.ldr % vn == 0 &&
         a.ldaux % vn == 0 && a.M >= 4096;
}
template <typename TA, typename TC, int KT, int WS_NW>
static int launch_ws_epi(const NtArgs& a, int epi, hipStream_t st) {
  const int ngroups = ceil_div(a.N, 64 * WS_NW);
  const int ntiles = ceil_div(a.M, WS_BM);
  // (r6: more, shorter workgroups -- 4 / 8 per CU in total, so that the dispatcher hands tiles to whichever CU is free beside the weight-gradient stream -- measured:
  //  1024 / 2048 in total: +0.10 / +0.15 ms in the step, profiles/r6o_ab_ws_slots_not_kept.txt; -DUVC_WS_WG_TOTAL builds)
#ifndef UVC_WS_WG_TOTAL
#define UVC_WS_WG_TOTAL 512
#endif
  int nslots = (UVC_WS_WG_TOTAL / ngroups) & ~7;           // ~2 workgroups per CU in total, slots a multiple of the 8 XCDs (r5: these main-stream kernels on 224 / 192 CUs: +1 % step time, profiles/r5u)
  if (nslots < 8) nslots = 8;
  if (nslots > ((ntiles + 7) & ~7)) nslots = (ntiles + 7) & ~7;
  const int grid = nslots * ngroups;
  if (epi == UVC_EPI_BIAS_GELU_GRAD_Q8) {                  // (bf16 operands and outputs, K = 192, four waves x 64 columns: uvc_gemm_nt checked)
    if constexpr (sizeof(TA) == 2 && sizeof(TC) == 2 && KT == 6 && WS_NW == 4) {
      k_gemm_ws<TA, TC, UVC_EPI_BIAS_GELU_GRAD_Q8, KT, WS_NW><<<grid, 64 * WS_NW, 0, st>>>(a, ngroups, nslots);
      UVC_CHECK_LAUNCH();
      return UVC_OK;
    } else return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: UVC_EPI_BIAS_GELU_GRAD_Q8 outside uvc_gemm_nt_q8_supported");
  }
#define WS_CASE(E) case E: k_gemm_ws<TA, TC, E, KT, WS_NW><<<grid, 64 * WS_NW, 0, st>>>(a, ngroups, nslots); break;
  switch (epi) {
    WS_CASE(UVC_EPI_NONE) WS_CASE(UVC_EPI_BIAS) WS_CASE(UVC_EPI_BIAS_GELU) WS_CASE(UVC_EPI_BIAS_RESID)
    WS_CASE(UVC_EPI_BIAS_RESID_GATE) WS_CASE(UVC_EPI_DGELU) WS_CASE(UVC_EPI_BIAS_GELU_OUT) WS_CASE(UVC_EPI_BIAS_GELU_GRAD) WS_CASE(UVC_EPI_MUL_AUX)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: unknown epilogue");
  }
#undef WS_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
// K = 384, N a multiple of 192 (DeiT-Small / T2T-ViT: qkv, proj, fc1 and the dgrads with K = D)
// Measured at M = 50 k rows against the generic tiled kernel: qkv 102 -> 72 us, fc1 (+GELU, GELU') 178 -> 148, proj 47 -> 43, fc2 dgrad x GELU'
// 133 -> 128; the plain and the recomputing-dGELU epilogues were slower (26 -> 28, 149 -> 162) and stay on the generic kernel.
static bool ws384_ok(const NtArgs& a, int epi, int vn, bool a_f32) {
  if (epi == UVC_EPI_DGELU) return false;
  if (epi == UVC_EPI_NONE && !(vn == 8 && a.N % 128 == 0 && a.N >= 512)) return false;      // (plain epilogue: the eight-wave form only -- T2T-ViT's bias-free qkv)
  return !a_f32 && a.K == 384 && a.N % 192 == 0 && a.ldb == a.K && a.lda % 8 == 0 && a.ldc % vn == 0 && a.ldr % vn == 0 && a.ldaux % vn == 0 && a.M >= 4096;
}
// NW = 8 (r4): N a multiple of 128 from 512 up runs eight waves x 32 columns -- two waves on every SIMD (six left two SIMDs with one), six column groups
// instead of eight per A tile at N = 1536; N = 1152 (qkv; T2T-ViT's MLP) is 4.5 groups, the last with four idle waves: still 6-15 % faster than six waves
template <typename TC, int NW = 6>
static int launch_ws384(const NtArgs& a, int epi, hipStream_t st) {
  constexpr int KT = 12, NJ = 2;
  const int ngroups = ceil_div(a.N, 32 * NW);               // (a last group with idle waves: N = 1152 on eight waves is 4.5 groups)
  const int ntiles = ceil_div(a.M, WS_BM);
  int nslots = (256 / ngroups) & ~7;                       // one workgroup per CU
  if (nslots < 8) nslots = 8;
  if (nslots > ((ntiles + 7) & ~7)) nslots = (ntiles + 7) & ~7;
  const int grid = nslots * ngroups;
  const int sh = 2 * WS_BM * (KT * 64 + 32) + NW * 16 * (16 * NJ + 4) * 4;
#define WS_CASE(E) case E: { \
    UVC_MAX_LDS(sh, k_gemm_ws<bf16_t, TC, E, KT, NW, NJ>); \
    k_gemm_ws<bf16_t, TC, E, KT, NW, NJ><<<grid, 64 * NW, sh, st>>>(a, ngroups, nslots); } break;
  switch (epi) {
    WS_CASE(UVC_EPI_NONE) WS_CASE(UVC_EPI_BIAS) WS_CASE(UVC_EPI_BIAS_GELU) WS_CASE(UVC_EPI_BIAS_RESID)
    WS_CASE(UVC_EPI_BIAS_RESID_GATE) WS_CASE(UVC_EPI_BIAS_GELU_OUT) WS_CASE(UVC_EPI_BIAS_GELU_GRAD) WS_CASE(UVC_EPI_MUL_AUX)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: unknown epilogue");
  }
#undef WS_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

template <typename TA, typename TC, int KT>
static int launch_ws_narrow(const NtArgs& a, int epi, hipStream_t st) {      // 8 waves x 32 columns
  const int ngroups = ceil_div(a.N, 256);
  const int ntiles = ceil_div(a.M, WS_BM);
  int nslots = (UVC_WS_WG_TOTAL / ngroups) & ~7;
  if (nslots < 8) nslots = 8;
  if (nslots > ((ntiles + 7) & ~7)) nslots = (ntiles + 7) & ~7;
  const int grid = nslots * ngroups;
#define WS_CASE(E) case E: k_gemm_ws<TA, TC, E, KT, 8, 2><<<grid, 512, 0, st>>>(a, ngroups, nslots); break;
  switch (epi) {
    WS_CASE(UVC_EPI_DGELU) WS_CASE(UVC_EPI_MUL_AUX) WS_CASE(UVC_EPI_MUL_AUX_Q8) WS_CASE(UVC_EPI_BIAS_GELU_GRAD_Q8)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: epilogue");
  }
#undef WS_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
template <typename TA, typename TC>
static int launch_ws(const NtArgs& a, int epi, hipStream_t st) {
  // dL/d(fc1 out) = (g W2) * gelu'(a): measured 95 -> 83 us (stored gelu') and 118 -> 105 us (recomputed) with 32 columns per wave;
  // the forward's GELU epilogues are VALU-bound either way and stay on the 64-column tiling
  if constexpr (sizeof(TC) == 2 && sizeof(TA) == 2) {
    if (a.N % 256 == 0 && a.K == 192 && (epi == UVC_EPI_DGELU || epi == UVC_EPI_MUL_AUX || epi == UVC_EPI_MUL_AUX_Q8)) return launch_ws_narrow<TA, TC, 6>(a, epi, st);
#ifdef UVC_FC1_NARROW        // A/B build: fc1 + GELU, GELU' (one-byte code) on eight waves x 32 columns (109 us against 84, the step 11.07 against 10.64 ms: not kept, profiles/r6j_ab_fc1_narrow_not_kept.txt)
    if (a.N % 256 == 0 && a.K == 192 && epi == UVC_EPI_BIAS_GELU_GRAD_Q8) return launch_ws_narrow<TA, TC, 6>(a, epi, st);
#endif
  }
  if (epi == UVC_EPI_MUL_AUX_Q8) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: UVC_EPI_MUL_AUX_Q8 outside uvc_gemm_nt_q8_supported");
  // 4 waves (256 columns) per workgroup when N divides: more waves per CU to overlap the GELU epilogues
  if (a.N % 256 == 0) return a.K == 192 ? launch_ws_epi<TA, TC, 6, 4>(a, epi, st) : launch_ws_epi<TA, TC, 4, 4>(a, epi, st);
  return a.K == 192 ? launch_ws_epi<TA, TC, 6, 3>(a, epi, st) : launch_ws_epi<TA, TC, 4, 3>(a, epi, st);
}

// ================================================================================================
//        NT, weights-stationary, one 16-column slice of the FULL reduction per wave (bf16, N == 192)
// ================================================================================================
// fc2 (+ residual + gate mix), dgrad of fc1, dgrad of qkv: K = 768 / 576 against N = 192.  W [192, K] does not fit one
// wave's registers as a 64-column slice, and splitting K over waves needs a cross-wave reduction with two barriers per
// sub-tile (tried: 141 us for fc2, 60 us for dgrad fc1; this kernel 136 / 45).  Here the workgroup is 12 waves, wave w owns output columns 16w..16w+15 and keeps that slice of W for the whole K in registers
// (KT fragments = 96 VGPRs at K = 768).  With W as the MFMA A operand the accumulator of lane (row, g) is four
// CONSECUTIVE output columns of one row, so the epilogue needs no LDS transpose and no barrier: residual rows are
// read and results written as 16-byte vectors (64 contiguous bytes per row per wave, the neighbouring waves fill the
// rest of the line).  A streams through a double-buffered 32-row LDS image (one barrier per tile); the two 16-row
// sub-tiles are two independent MFMA chains; next tile's residual operands are prefetched under this tile's math.
constexpr int WN_BM = 32;

template <typename TC, int EPI, int KT>
__global__ __launch_bounds__(768) void k_gemm_wsn(NtArgs g) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int K = KT * 32, NTH = 768;
  constexpr int ROWB = K * 2 + 32, CPR = K / 8;
  constexpr int NLD = (WN_BM * CPR + NTH - 1) / NTH;
  constexpr bool RES = EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA0 = smem;
  char* sA1 = smem + WN_BM * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, gq = lane >> 4, li = lane & 15;
  const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.B);
  TC* __restrict__ C = reinterpret_cast<TC*>(g.C);
  const int ntiles = (g.M + WN_BM - 1) / WN_BM;
  const int n = w * 16 + gq * 4;                       // this lane's four output columns

  typename MM::Frag bf[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks)
    bf[ks] = __builtin_bit_cast(typename MM::Frag, *reinterpret_cast<const u32x4*>(W + (size_t)(w * 16 + li) * g.ldb + (ks * 4 + gq) * 8));
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (EPI == UVC_EPI_BIAS_RESID_GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI == UVC_EPI_BIAS || RES) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);

  u32x4 ra[NLD];
  auto gload = [&](int tile) {
    const int m0 = tile * WN_BM;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + NTH * i, row = id / CPR, c = id % CPR;
      const int m = m0 + row;
      ra[i] = (row < WN_BM && m < g.M) ? *reinterpret_cast<const u32x4*>(A + (size_t)m * g.lda + c * 8) : z;
    }
  };
  auto lstore = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + NTH * i, row = id / CPR, c = id % CPR;
      if (row < WN_BM) *reinterpret_cast<u32x4*>(buf + row * ROWB + c * 16) = ra[i];
    }
  };
  struct Epi { f32x4 r[2]; f32x4 r2[2]; };
  auto eload = [&](Epi& E, int tile_) {
    if (!RES) return;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int m = tile_ * WN_BM + s2 * 16 + li;
      const size_t off = (m < g.M && tile_ < ntiles) ? (size_t)m * g.ldr + n : 0;
      E.r[s2] = Resid<TC>::f4(Resid<TC>::ld4(g.R, off));
      if (EPI == UVC_EPI_BIAS_RESID_GATE) E.r2[s2] = Resid<TC>::f4(Resid<TC>::ld4(g.R2, off));
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  Epi E, En;
  gload(tile);
  eload(E, tile);
  lstore(sA0);
  __syncthreads();
  int par = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < ntiles) gload(next);
    const char* buf = par ? sA1 : sA0;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      const typename MM::Frag f0 = lds_frag<T>(buf + li * ROWB + (ks * 4 + gq) * 16);
      const typename MM::Frag f1 = lds_frag<T>(buf + (16 + li) * ROWB + (ks * 4 + gq) * 16);
      c0 = MM::mma(bf[ks], f0, c0);
      c1 = MM::mma(bf[ks], f1, c1);
    }
    eload(En, next);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int m = tile * WN_BM + s2 * 16 + li;
      f32x4 v = s2 ? c1 : c0;
      if (m < g.M) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = epi_scale_bias(v[e], alpha, bias4[e]);
        if (RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += E.r[s2][e];
        }
        if (EPI == UVC_EPI_BIAS_RESID_GATE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = epi_gate_mix(o[e], E.r2[s2][e], d0, d1);
        }
        if (sizeof(TC) == 4) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (size_t)m * g.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
        else { u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]); *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(C) + (size_t)m * g.ldc + n) = q; }
      }
    }
    E = En;
    if (next < ntiles) lstore(par ? sA0 : sA1);
    __syncthreads();
    par ^= 1;
  }
}

template <typename TC, int EPI, int KT>
__global__ __launch_bounds__(768) void k_gemm_wsn16(NtArgs g) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int K = KT * 32, NTH = 768;
  constexpr int ROWB = K * 2 + 32, CPR = K / 8;
  constexpr int NLD = (16 * CPR + NTH - 1) / NTH;
  constexpr bool RES = EPI == UVC_EPI_BIAS_RESID || EPI == UVC_EPI_BIAS_RESID_GATE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA0 = smem;
  char* sA1 = smem + 16 * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, gq = lane >> 4, li = lane & 15;
  const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.B);
  TC* __restrict__ C = reinterpret_cast<TC*>(g.C);
  const int ntiles = (g.M + 16 - 1) / 16;
  const int n = w * 16 + gq * 4;                       // this lane's four output columns

  typename MM::Frag bf[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks)
    bf[ks] = __builtin_bit_cast(typename MM::Frag, *reinterpret_cast<const u32x4*>(W + (size_t)(w * 16 + li) * g.ldb + (ks * 4 + gq) * 8));
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (EPI == UVC_EPI_BIAS_RESID_GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (EPI == UVC_EPI_BIAS || RES) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);

  u32x4 ra[NLD];
  auto gload = [&](int tile) {
    const int m0 = tile * 16;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + NTH * i, row = id / CPR, c = id % CPR;
      const int m = m0 + row;
      ra[i] = (row < 16 && m < g.M) ? *reinterpret_cast<const u32x4*>(A + (size_t)m * g.lda + c * 8) : z;
    }
  };
  auto lstore = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + NTH * i, row = id / CPR, c = id % CPR;
      if (row < 16) *reinterpret_cast<u32x4*>(buf + row * ROWB + c * 16) = ra[i];
    }
  };
  struct Epi { typename Resid<TC>::Raw4 r[1]; typename Resid<TC>::Raw4 r2[1]; };      // raw: the bf16 stream keeps 2 instead of 4 registers per operand in flight
  auto eload = [&](Epi& E, int tile_) {
    if (!RES) return;
#pragma unroll
    for (int s2 = 0; s2 < 1; ++s2) {
      const int m = tile_ * 16 + s2 * 16 + li;
      const size_t off = (m < g.M && tile_ < ntiles) ? (size_t)m * g.ldr + n : 0;
      E.r[s2] = Resid<TC>::ld4(g.R, off);
      if (EPI == UVC_EPI_BIAS_RESID_GATE) E.r2[s2] = Resid<TC>::ld4(g.R2, off);
    }
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  Epi E, En;
  gload(tile);
  eload(E, tile);
  lstore(sA0);
  __syncthreads();
  int par = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < ntiles) gload(next);
    const char* buf = par ? sA1 : sA0;
    // one accumulation chain in k order: the same summation order as every other NT kernel of the library, so a row's result
    // does not depend on which kernel the batch size selects (tests/test_fullsize_gpu.py); the 12 waves hide the chain latency
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) c0 = MM::mma(bf[ks], lds_frag<T>(buf + li * ROWB + (ks * 4 + gq) * 16), c0);
    eload(En, next);
#pragma unroll
    for (int s2 = 0; s2 < 1; ++s2) {
      const int m = tile * 16 + s2 * 16 + li;
      f32x4 v = c0;
      if (m < g.M) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = epi_scale_bias(v[e], alpha, bias4[e]);
        if (RES) {
          const f32x4 rv = Resid<TC>::f4(E.r[s2]);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
        if (EPI == UVC_EPI_BIAS_RESID_GATE) {
          const f32x4 rv = Resid<TC>::f4(E.r2[s2]);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = epi_gate_mix(o[e], rv[e], d0, d1);
        }
        if (sizeof(TC) == 4) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (size_t)m * g.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
        else { u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]); *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(C) + (size_t)m * g.ldc + n) = q; }
      }
    }
    E = En;
    if (next < ntiles) lstore(par ? sA0 : sA1);
    __syncthreads();
    par ^= 1;
  }
}

static bool wsn_ok(const NtArgs& a, int epi, bool a_f32) {
  return !a_f32 && a.N == 192 && (a.K == 768 || a.K == 576 || a.K == 512 || a.K == 256) && a.ldb == a.K && a.lda % 8 == 0 && a.ldc % 4 == 0 && a.ldr % 4 == 0 && a.M >= 4096 &&
         (epi == UVC_EPI_NONE || epi == UVC_EPI_BIAS || epi == UVC_EPI_BIAS_RESID || epi == UVC_EPI_BIAS_RESID_GATE);
}
template <int EPI, int KT, bool LN, int NST, bool RLOW> __global__ void k_gemm_wsn16_dma(NtArgs g);
template <int EPI, int KT, bool LN, int NST, bool RLOW>
static int launch_wsn16_dma(const NtArgs& a, hipStream_t st) {
  constexpr int NX_ = RLOW ? 7 : 13;                              // KB of a stage per residual operand (16 rows x (24 | 48) + 2 slots)
  constexpr int NA_ = (16 * (KT * 4 + 2) + 63) / 64, NI_ = NA_ + NX_ + (EPI == UVC_EPI_BIAS_RESID_GATE ? NX_ : 0);
  const int sh = NST * NI_ * 1024 + (LN ? 2 * 16 * 12 * 8 + 3 * 192 * 4 : 0);
  UVC_MAX_LDS(sh, k_gemm_wsn16_dma<EPI, KT, LN, NST, RLOW>);
  const int ntiles = ceil_div(a.M, 16);
  k_gemm_wsn16_dma<EPI, KT, LN, NST, RLOW><<<ntiles < 256 ? ntiles : 256, 768, sh, st>>>(a);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
// the shapes whose residual epilogue can also write LayerNorm(output rows): what k_gemm_wsn16_dma takes
static bool wsn16_dma_ok(const NtArgs& a) {
  return a.M >= 16 && a.lda == a.K && a.ldr == 192 && a.ldc % 4 == 0 && (((uintptr_t)a.R | (uintptr_t)a.R2 | (uintptr_t)a.A) & 15) == 0;
}
template <typename TC, int KT>
static int launch_wsn16_kt(const NtArgs& a, int epi, hipStream_t st) {
  if constexpr (KT == 24 || KT == 16 || KT == 8) {
    constexpr bool RLOW = sizeof(TC) == 2;
    // fc2 of DeiT-Tiny on the LDS-DMA ring (uvc_gemm_nt_args.force_generic = 2 keeps the register-staged kernel); with ln_out also the compacted Stage-2
    // widths (K = 512 / 256: more stages of the smaller images fit)
    const bool ok = (!a.no_dma || a.ln_out) && wsn16_dma_ok(a) && (a.ln_out || a.M % 16 == 0);
    constexpr int NST_ = KT == 8 ? 4 : 3;
    if (ok && a.ln_out) {
      if (epi == UVC_EPI_BIAS_RESID) return launch_wsn16_dma<UVC_EPI_BIAS_RESID, KT, true, NST_, RLOW>(a, st);
      if (epi == UVC_EPI_BIAS_RESID_GATE) return launch_wsn16_dma<UVC_EPI_BIAS_RESID_GATE, KT, true, NST_, RLOW>(a, st);
    }
    if constexpr (KT == 24) {
      if (ok && epi == UVC_EPI_BIAS_RESID) return launch_wsn16_dma<UVC_EPI_BIAS_RESID, KT, false, 3, RLOW>(a, st);
      if (ok && epi == UVC_EPI_BIAS_RESID_GATE) return launch_wsn16_dma<UVC_EPI_BIAS_RESID_GATE, KT, false, 3, RLOW>(a, st);
    }
  }
  if (a.ln_out) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: ln_out is not available for this problem (uvc_gemm_nt_ln_supported)");
  const int ntiles = ceil_div(a.M, 16);
  const int grid = ntiles < 256 ? ntiles : 256;
  const size_t sh = (size_t)2 * 16 * (KT * 64 + 32);
#define WN_CASE(E) case E: { \
    UVC_MAX_LDS(sh, k_gemm_wsn16<TC, E, KT>); \
    k_gemm_wsn16<TC, E, KT><<<grid, 768, sh, st>>>(a); } break;
  switch (epi) {
    WN_CASE(UVC_EPI_NONE) WN_CASE(UVC_EPI_BIAS) WN_CASE(UVC_EPI_BIAS_RESID) WN_CASE(UVC_EPI_BIAS_RESID_GATE)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: epilogue");
  }
#undef WN_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
template <typename TC, int KT>
static int launch_wsn_kt(const NtArgs& a, int epi, hipStream_t st) {
  // float32 outputs (fc2 with its residual / gate operands) run on 16-row tiles: with 32 rows the two sub-tiles' residual
  // prefetch spilled 24-45 VGPRs next to the 96 of W and cost a third of the time (129 -> 88 us); bf16 outputs keep 32 rows
  if (sizeof(TC) == 4 || epi == UVC_EPI_BIAS_RESID || epi == UVC_EPI_BIAS_RESID_GATE) {
    return launch_wsn16_kt<TC, KT>(a, epi, st);
  } else {
  const int ntiles = ceil_div(a.M, WN_BM);
  const int grid = ntiles < 256 ? ntiles : 256;                 // one persistent workgroup per CU
  const size_t sh = (size_t)2 * WN_BM * (KT * 64 + 32);
#define WN_CASE(E) case E: { \
    UVC_MAX_LDS(sh, k_gemm_wsn<TC, E, KT>); \
    k_gemm_wsn<TC, E, KT><<<grid, 768, sh, st>>>(a); } break;
  switch (epi) {                       // bf16 outputs only (float32 outputs run k_gemm_wsn16; residual epilogues need float32 C)
    WN_CASE(UVC_EPI_NONE) WN_CASE(UVC_EPI_BIAS)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: epilogue not supported by the column-sliced streaming kernel");
  }
#undef WN_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
  }
}
template <typename TC>
static int launch_wsn(const NtArgs& a, int epi, hipStream_t st) {
  switch (a.K) {                                      // 512 / 256: compacted MLP widths of Stage-2
    case 768: return launch_wsn_kt<TC, 24>(a, epi, st);
    case 576: return launch_wsn_kt<TC, 18>(a, epi, st);
    case 512: return launch_wsn_kt<TC, 16>(a, epi, st);
    default: return launch_wsn_kt<TC, 8>(a, epi, st);
  }
}

template <typename TA, typename T, typename TC>
static int launch_nt_epi(const NtArgs& a, int epi, hipStream_t st) {
  const int grid = ceil_div(ceil_div(a.M, NT_BM), 8) * 8 * ceil_div(a.N, NT_BN);   // M tiles padded to the 8 XCDs
#define NT_CASE(E) case E: k_gemm_nt<TA, T, TC, E><<<grid, 256, 0, st>>>(a); break;
  switch (epi) {
    NT_CASE(UVC_EPI_NONE) NT_CASE(UVC_EPI_BIAS) NT_CASE(UVC_EPI_BIAS_GELU) NT_CASE(UVC_EPI_BIAS_RESID)
    NT_CASE(UVC_EPI_BIAS_RESID_GATE) NT_CASE(UVC_EPI_DGELU) NT_CASE(UVC_EPI_BIAS_GELU_OUT) NT_CASE(UVC_EPI_BIAS_GELU_GRAD) NT_CASE(UVC_EPI_MUL_AUX)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: unknown epilogue");
  }
#undef NT_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// the 256 x 256-tile kernel of the wide models (defined behind the LDS-DMA helpers below)
int launch_nt256_f32(const NtArgs& a, int epi, hipStream_t st);
int launch_nt256_bf16(const NtArgs& a, int epi, hipStream_t st);
bool nt256_takes(const NtArgs& a, bool a_f32, bool any_size);
int launch_nt8p_f32(const NtArgs& a, int epi, int ri, hipStream_t st);
int launch_nt8p_bf16(const NtArgs& a, int epi, int ri, hipStream_t st);
bool nt8p_takes(const NtArgs& a, bool a_f32);
bool row384_fwd_ok(const NtArgs& a, int epi);
int launch_row384_fwd(const NtArgs& a, int epi, hipStream_t st);

extern "C" int uvc_gemm_nt_ln_supported(int32_t M, int32_t N, int32_t K, int32_t dtype, int32_t epilogue) {
  // N = 384 (r4): k_gemm_row384_lnbwd<.., 1>, bf16 C / R only; its rows are bit-identical to the unfused pair, so the row threshold does not show in results
  if (dtype == UVC_BF16 && N == 384) return M >= 4096 && K % 64 == 0 && K >= 384 && (epilogue == UVC_EPI_BIAS_RESID || epilogue == UVC_EPI_BIAS_RESID_GATE);
  if (dtype != UVC_BF16 || N != 192 || M < 16) return 0;      // any row count from 16 up: whether norm is fused must not depend on the batch
  return ((K == 768 || K == 512 || K == 256) && (epilogue == UVC_EPI_BIAS_RESID || epilogue == UVC_EPI_BIAS_RESID_GATE)) || (K == 192 && epilogue == UVC_EPI_BIAS_RESID);
}

extern "C" int uvc_gemm_nt_q8_supported(int32_t M, int32_t hidden, int32_t embed_dim, int32_t dtype) {
  return dtype == UVC_BF16 && embed_dim == 192 && hidden > 0 && hidden % 256 == 0 && M >= 4096;
}

extern "C" int uvc_gemm_nt(const uvc_gemm_nt_args* p, void* stream) {
  if (!p || !p->A || !p->B || !p->C) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: null pointer");
  if (p->M <= 0 || p->N <= 0 || p->K <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: empty problem");
  const int ch = (p->dtype == UVC_F32) ? 4 : 8;
  if (p->K % ch || p->lda % ch || p->ldb % ch) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: K, lda, ldb must be multiples of a 16-byte chunk");
  const int e = p->epilogue;
  if ((e == UVC_EPI_BIAS || e == UVC_EPI_BIAS_GELU || e == UVC_EPI_BIAS_GELU_OUT || e == UVC_EPI_BIAS_RESID || e == UVC_EPI_BIAS_RESID_GATE ||
       e == UVC_EPI_BIAS_GELU_GRAD) && !p->bias)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: epilogue needs bias");
  if ((e == UVC_EPI_BIAS_RESID || e == UVC_EPI_BIAS_RESID_GATE) && !p->R)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: residual epilogue needs R");
  if ((e == UVC_EPI_BIAS_RESID || e == UVC_EPI_BIAS_RESID_GATE) && (p->r_is_f32 != 0) != (p->c_is_f32 != 0 || p->dtype == UVC_F32))
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: R / R2 have C's element type (r_is_f32 must equal c_is_f32)");
  if (e == UVC_EPI_BIAS_RESID_GATE && (!p->R2 || !p->gate)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: gate epilogue needs R2 and gate");
  if ((e == UVC_EPI_BIAS_GELU || e == UVC_EPI_BIAS_GELU_GRAD) && !p->C2) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: GELU epilogue needs C2");
  if ((e == UVC_EPI_DGELU || e == UVC_EPI_MUL_AUX || e == UVC_EPI_MUL_AUX_Q8) && !p->aux) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: dGELU epilogue needs aux");
  const bool q8 = e == UVC_EPI_BIAS_GELU_GRAD_Q8 || e == UVC_EPI_MUL_AUX_Q8;
  if (e == UVC_EPI_BIAS_GELU_GRAD_Q8 && (!p->bias || !p->C2)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: UVC_EPI_BIAS_GELU_GRAD_Q8 needs bias and C2");
  // the one-byte GELU' code exists in the streaming kernel of DeiT-Tiny's width only (uvc_gemm_nt_q8_supported): N here is the hidden width, K the embed_dim
  if (q8 && (p->force_generic == 1 || p->a_is_f32 || p->c_is_f32 || p->ln_out || !uvc_gemm_nt_q8_supported(p->M, p->N, p->K, p->dtype) ||
             (p->ldaux ? p->ldaux : p->ldc) % 8 || p->ldc % 8 || p->ldb != p->K))
    return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: the one-byte GELU' epilogues need bf16 A / C, K = 192, N % 256 == 0, M >= 4096 (uvc_gemm_nt_q8_supported)");
  NtArgs a;
  a.A = p->A; a.B = p->B; a.C = p->C; a.C2 = p->C2; a.bias = p->bias; a.R = p->R; a.R2 = p->R2; a.aux = p->aux;
  a.dptr = p->gate; a.M = p->M; a.N = p->N; a.K = p->K; a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc;
  a.ldr = p->ldr ? p->ldr : p->ldc; a.ldaux = p->ldaux ? p->ldaux : p->ldc; a.alpha = p->alpha; a.alpha_ptr = p->alpha_ptr;
  a.ln_gamma = p->ln_gamma; a.ln_beta = p->ln_beta; a.ln_out = p->ln_out; a.ln_mean = p->ln_mean; a.ln_rstd = p->ln_rstd; a.ln_eps = p->ln_eps;
  a.no_dma = p->force_generic == 2;
  const bool generic = p->force_generic == 1;
  hipStream_t st = (hipStream_t)stream;
  if (p->ln_out) {
    if (!p->ln_gamma || !p->ln_beta || (p->ln_mean != nullptr) != (p->ln_rstd != nullptr))
      return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: ln_out needs ln_gamma, ln_beta (and ln_mean, ln_rstd together)");
    if (p->N == 384) {
      if (!uvc_gemm_nt_ln_supported(p->M, p->N, p->K, p->dtype, e) || generic || p->a_is_f32 || p->c_is_f32 || p->alpha != 1.0f || p->alpha_ptr || !row384_fwd_ok(a, e))
        return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: ln_out is not available for this problem (uvc_gemm_nt_ln_supported; N = 384: bf16 C and R only)");
      return launch_row384_fwd(a, e, st);
    }
    if (!uvc_gemm_nt_ln_supported(p->M, p->N, p->K, p->dtype, e) || generic || p->a_is_f32 || p->alpha != 1.0f || p->alpha_ptr ||
        a.ldb != a.K || !wsn16_dma_ok(a) || ((uintptr_t)p->ln_out & 7) != 0)
      return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: ln_out is not available for this problem (uvc_gemm_nt_ln_supported)");
    if (p->M % 16 != 0 && (p->C == (const void*)p->R || p->C == (const void*)p->R2))
      return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: with ln_out and M % 16 != 0 the last row tile is computed twice: C must not alias R / R2");
    // attn.proj + residual -> norm2 (K = 192): 20-KB stages, seven of them, six in flight
    if (p->K == 192) return p->c_is_f32 ? launch_wsn16_dma<UVC_EPI_BIAS_RESID, 6, true, 7, false>(a, st) : launch_wsn16_dma<UVC_EPI_BIAS_RESID, 6, true, 7, true>(a, st);
    return p->c_is_f32 ? launch_wsn<float>(a, e, st) : launch_wsn<bf16_t>(a, e, st);
  }
  if (p->dtype == UVC_F32) {
    if (!p->a_is_f32 || !p->c_is_f32) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: float32 mode needs float32 A and C");
    return launch_nt_epi<float, float, float>(a, e, st);
  }
  if (p->dtype != UVC_BF16) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: dtype must be UVC_F32 or UVC_BF16");
  {
    // attn.proj + residual (K = N = 192) without ln_out: the same seven-stage ring (same k-ordered chain and epilogue: same bits)
    if (!p->force_generic && !p->a_is_f32 && e == UVC_EPI_BIAS_RESID && p->K == 192 && p->N == 192 && a.ldb == 192 &&
        p->M >= 4096 && p->M % 16 == 0 && p->alpha == 1.0f && !p->alpha_ptr && wsn16_dma_ok(a))
      return p->c_is_f32 ? launch_wsn16_dma<UVC_EPI_BIAS_RESID, 6, false, 7, false>(a, st) : launch_wsn16_dma<UVC_EPI_BIAS_RESID, 6, false, 7, true>(a, st);
  }
  const bool ws = !generic && ws_ok(a, p->c_is_f32 ? 4 : 8);
  if (q8 && !ws) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: the one-byte GELU' epilogues exist in the K = 192 streaming kernel only");
  if (!generic && ws384_ok(a, e, p->c_is_f32 ? 4 : 8, p->a_is_f32 != 0)) {
    if (!p->c_is_f32 && a.N % 128 == 0 && a.N >= 512 && p->force_generic != 5) return launch_ws384<bf16_t, 8>(a, e, st);    // (force_generic == 5: six waves, A/B)
    return p->c_is_f32 ? launch_ws384<float>(a, e, st) : launch_ws384<bf16_t>(a, e, st);
  }
  if (!generic && wsn_ok(a, e, p->a_is_f32 != 0))
    return p->c_is_f32 ? launch_wsn<float>(a, e, st) : launch_wsn<bf16_t>(a, e, st);
  // (force_generic == 2 asks for kernels WITHOUT an LDS-DMA ring: the 256 x 256 kernel is one; == 3 takes it at any size)
  if ((p->force_generic >> 8) == 1) {                 // tuning: the 8-phase kernel with 32 * (force_generic & 255) rows per tile, at any size
    if (!nt8p_takes(a, p->a_is_f32 != 0)) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt: shape not taken by the 8-phase kernel");
    return p->c_is_f32 ? launch_nt8p_f32(a, e, p->force_generic & 255, st) : launch_nt8p_bf16(a, e, p->force_generic & 255, st);
  }
  // Wide GEMMs (K >= 256 against N >= 256).  Many tiles (>= 640 of 256 x 256: qkv, fc1, dfc2 of DeiT-Base): the lock-step 256 x 256 kernel -- at
  // K = 768 the two kernels' k-loops cost the same (both wait for the L2 -> LDS latency of their requests: 64 KB in flight per CU) and its prologue
  // is shorter (88 against 98 us for qkv; at K = 6144 the 8-phase kernel wins 542 : 607).  Fewer tiles with N a multiple of 256 (N = 768: proj,
  // fc2, dfc1, dqkv, dproj -- 297 tiles of 256 x 256 are two rounds at 58 %): the 8-phase kernel at the tile height that fills its rounds
  // (160 rows: 474 tiles): fc2 + residual + gate 153 -> 137 us, dfc1 129 -> 95, dqkv 98 -> 74, dproj 38 -> 32 against the 128 x 128 kernel.
  // (force_generic == 3, "the production batch's kernels at any size": N < 1792 is what stays below 640 tiles at DeiT-Base's 25 216 rows)
  const bool few_tiles_wide = !p->c_is_f32 && a.N % 256 == 0 && nt8p_takes(a, p->a_is_f32 != 0);
  if (!generic && !a.no_dma && !(p->force_generic == 3 && few_tiles_wide && a.N < 1792) &&
      nt256_takes(a, p->a_is_f32 != 0, p->force_generic == 3 || p->force_generic == 4))
    return p->c_is_f32 ? launch_nt256_f32(a, e, st) : launch_nt256_bf16(a, e, st);
  if (!generic && !a.no_dma && few_tiles_wide && (a.M >= 2048 || p->force_generic == 3))
    return launch_nt8p_bf16(a, e, 0, st);
  if (p->a_is_f32) {
    if ((p->lda % 8) != 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: lda");
    if (ws) return p->c_is_f32 ? launch_ws<float, float>(a, e, st) : launch_ws<float, bf16_t>(a, e, st);
    return p->c_is_f32 ? launch_nt_epi<float, bf16_t, float>(a, e, st) : launch_nt_epi<float, bf16_t, bf16_t>(a, e, st);
  }
  if (ws) return p->c_is_f32 ? launch_ws<bf16_t, float>(a, e, st) : launch_ws<bf16_t, bf16_t>(a, e, st);
  return p->c_is_f32 ? launch_nt_epi<bf16_t, bf16_t, float>(a, e, st) : launch_nt_epi<bf16_t, bf16_t, bf16_t>(a, e, st);
}

// ================================================================================================
//      NT dgrad (K = 768 / 576 -> N = 192) with the LayerNorm backward as its epilogue (bf16 mode)
// ================================================================================================
// dL/dx of `y = LayerNorm(x)` fed by a Linear: dy = A . W^T (the dgrad of fc1 / qkv) is consumed where it is produced,
//     dx = rstd * (dy*gamma - mean_c(dy*gamma) - xhat * mean_c(dy*gamma*xhat)) + a1*add1 + a2*add2
// (model_distilled.py:199-204,218-247 through autograd), so the [M, 192] dy never goes to HBM and back (2 x 39 MB of the
// 194-233 MB a stand-alone LayerNorm backward moves, and one launch of the step's most expensive kernel).  Same geometry as
// k_gemm_wsn16: 12 waves, wave w owns output columns 16w..16w+15 for the whole K (KT fragments of W^T in VGPRs), lane
// (row li, group gq) ends the MFMA chain with FOUR CONSECUTIVE columns of one row.  The two row sums run over the 12 waves:
// every wave leaves its (sum dy*gamma, sum dy*gamma*xhat) over its 16 columns in a double-buffered LDS table and picks up the
// 12 partials after the tile's ONE barrier (the same barrier that publishes the next A image), summed in wave order -> the
// result does not depend on timing.  dgamma / dbeta accumulate per lane over the workgroup's rows and leave through the
// partial table of the stand-alone kernel ([grid][2D+2], finished by uvc_layernorm_bwd_reduce_batch); the two gate dot
// products <dx, x>, <add2, x> ride along as there.  The row's x / add1 / add2 / mean / rstd are requested before the MFMA
// chain.  dx may alias add2 (the engine's gA is read and rewritten in place: same lane, same addresses).
// four elements as they sit in memory (no conversion)
template <typename T> struct Raw4L;
template <> struct Raw4L<float> {
  typedef f32x4 V;
  static __device__ __forceinline__ V ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ V zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4 cvt(const V& r) { return r; }
};
template <> struct Raw4L<bf16_t> {
  typedef u32x2 V;
  static __device__ __forceinline__ V ld(const bf16_t* p) { return *reinterpret_cast<const u32x2*>(p); }
  static __device__ __forceinline__ V zero() { return u32x2{0u, 0u}; }
  static __device__ __forceinline__ f32x4 cvt(const V& r) {
    return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
  }
};
struct LnbArgs {
  const void* A; const void* W; const void* x; const float* mean; const float* rstd; const float* gamma;
  const void* add1; const float* a1; const void* add2; const float* a2; void* dx; float* partial;
  int M, K, want_dots;
};

// XLOW: the LayerNorm input x (the residual stream) is stored as bf16
template <int KT, bool XLOW>
__global__ __launch_bounds__(768) void k_gemm_wsn_lnbwd(LnbArgs g) {
  typedef Resid<typename std::conditional<XLOW, bf16_t, float>::type> XS_;
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int K = KT * 32, NTH = 768, D = 192, NWV = 12;
  constexpr int ROWB = K * 2 + 32, CPR = K / 8;
  constexpr int NLD = (16 * CPR + NTH - 1) / NTH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA0 = smem;
  char* sA1 = smem + 16 * ROWB;
  float* sRed = reinterpret_cast<float*>(smem + 2 * 16 * ROWB);            // [2][16 rows][12 waves][2]
  float* sGam = sRed + 2 * 16 * NWV * 2;                                   // gamma [192]: re-read per tile instead of 4 live VGPRs
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, gq = lane >> 4, li = lane & 15;
  const T* __restrict__ A = reinterpret_cast<const T*>(g.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W);
  const T* __restrict__ add1 = reinterpret_cast<const T*>(g.add1);
  const T* __restrict__ add2 = reinterpret_cast<const T*>(g.add2);
  T* __restrict__ dx = reinterpret_cast<T*>(g.dx);
  const int ntiles = (g.M + 15) / 16;
  const int n = w * 16 + gq * 4;                       // this lane's four output columns

  typename MM::Frag bf[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks)
    bf[ks] = __builtin_bit_cast(typename MM::Frag, *reinterpret_cast<const u32x4*>(W + (size_t)(w * 16 + li) * K + (ks * 4 + gq) * 8));
  if (tid < D) sGam[tid] = g.gamma[tid];
  const float a1 = g.a1 ? *g.a1 : 1.f, a2 = g.a2 ? *g.a2 : 1.f;
  f32x4 dgam = {0.f, 0.f, 0.f, 0.f}, dbet = {0.f, 0.f, 0.f, 0.f};
  float dotA = 0.f, dotB = 0.f;
  constexpr float invD = 1.0f / (float)D;

  // 32-bit element offsets everywhere (M * K < 2^31): SGPR base + one VGPR offset per access instead of 64-bit VGPR address pairs
  u32x4 ra[NLD];
  unsigned aoff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) { const int id = tid + NTH * i; aoff[i] = (unsigned)((id / CPR) * K + (id % CPR) * 8); }
  auto gload = [&](int tile) {
    const int m0 = tile * 16;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int row = (tid + NTH * i) / CPR;
      ra[i] = (row < 16 && m0 + row < g.M) ? *reinterpret_cast<const u32x4*>(A + ((unsigned)(m0 * K) + aoff[i])) : z;
    }
  };
  auto lstore = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int id = tid + NTH * i, row = id / CPR, c = id % CPR;
      if (row < 16) *reinterpret_cast<u32x4*>(buf + row * ROWB + c * 16) = ra[i];
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) {
    gload(tile);
    lstore(sA0);
  }
  __syncthreads();
  int par = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < ntiles) gload(next);
    // this row's LayerNorm operands (in flight under the MFMA chain); 32-bit element offsets: one VGPR addresses all four streams
    const int m = tile * 16 + li;
    const bool ok = m < g.M;
    const unsigned ro = (unsigned)(ok ? m : 0) * (unsigned)D + (unsigned)n;
    const f32x4 xv = XS_::f4(XS_::ld4(g.x, ro));
    u32x2 r1 = {0u, 0u}, r2 = {0u, 0u};
    if (add1) r1 = *reinterpret_cast<const u32x2*>(add1 + ro);
    if (add2) r2 = *reinterpret_cast<const u32x2*>(add2 + ro);
    const float mean = g.mean[ok ? m : 0], rstd = ok ? g.rstd[m] : 0.f;
    const char* buf = par ? sA1 : sA0;
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      c0 = MM::mma(bf[ks], lds_frag<T>(buf + li * ROWB + (ks * 4 + gq) * 16), c0);
      // bound the fragment reads the scheduler may hoist: next to 96 VGPRs of W the whole chain's 24 x 4 do not fit
      if (KT > 18 && (ks % 6) == 5) __builtin_amdgcn_sched_barrier(0);
    }
    float* red = sRed + par * (16 * NWV * 2);
    {
      const f32x4 gam = *reinterpret_cast<const f32x4*>(sGam + n);
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mean) * rstd;       // rows past M: rstd = 0, c0 = 0 (zero A rows) -> no contribution
        const float gy = c0[e] * gam[e];
        dgam[e] += c0[e] * xh;
        dbet[e] += c0[e];
        p1 += gy;
        p2 += gy * xh;
      }
      p1 = sum_rows4(p1);
      p2 = sum_rows4(p2);
      if (gq == 0) *reinterpret_cast<f32x2*>(red + (li * NWV + w) * 2) = f32x2{p1, p2};
    }
    if (next < ntiles) lstore(par ? sA0 : sA1);
    __syncthreads();
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int q = 0; q < NWV / 2; ++q) {               // 12 (p1, p2) pairs of this row, in wave order
      const f32x4 t = *reinterpret_cast<const f32x4*>(red + li * NWV * 2 + q * 4);
      c1 += t[0]; c2 += t[1]; c1 += t[2]; c2 += t[3];
    }
    c1 *= invD; c2 *= invD;
    if (ok) {
      const f32x4 gam = *reinterpret_cast<const f32x4*>(sGam + n);
      const f32x4 v1 = {__uint_as_float(r1[0] << 16), __uint_as_float(r1[0] & 0xffff0000u), __uint_as_float(r1[1] << 16), __uint_as_float(r1[1] & 0xffff0000u)};
      const f32x4 v2 = {__uint_as_float(r2[0] << 16), __uint_as_float(r2[0] & 0xffff0000u), __uint_as_float(r2[1] << 16), __uint_as_float(r2[1] & 0xffff0000u)};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        o[e] = rstd * (c0[e] * gam[e] - c1 - xh * c2);
        if (add1) o[e] += a1 * v1[e];
        if (add2) { o[e] += a2 * v2[e]; dotB += v2[e] * xv[e]; }
        dotA += o[e] * xv[e];
      }
      u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]);
      *reinterpret_cast<u32x2*>(dx + ro) = q;
    }
    par ^= 1;
  }
  // ---- dgamma / dbeta: sum over the 16 row lanes; dots over the workgroup (fixed order)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { dgam[e] += __shfl_xor(dgam[e], o, 64); dbet[e] += __shfl_xor(dbet[e], o, 64); }
  }
  float* P = g.partial + (size_t)blockIdx.x * (2 * D + 2);
  if (li == 0) {          // rows of the partial table are 2D+2 floats: 8-byte aligned only
    *reinterpret_cast<f32x2*>(P + n) = f32x2{dgam[0], dgam[1]}; *reinterpret_cast<f32x2*>(P + n + 2) = f32x2{dgam[2], dgam[3]};
    *reinterpret_cast<f32x2*>(P + D + n) = f32x2{dbet[0], dbet[1]}; *reinterpret_cast<f32x2*>(P + D + n + 2) = f32x2{dbet[2], dbet[3]};
  }
  dotA = wave_sum(dotA); dotB = wave_sum(dotB);
  __syncthreads();
  if (lane == 0) { sRed[2 * w] = dotA; sRed[2 * w + 1] = dotB; }
  __syncthreads();
  if (tid < 2) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NWV; ++q) t += sRed[2 * q + tid];
    P[2 * D + tid] = t;
  }
}

// ---- the same kernel with LDS-DMA staging: bytes in flight instead of registers --------------------------------------------
// k_gemm_wsn_lnbwd keeps ONE 16-row tile (49 KB) in flight per CU -- its A rows are requested one iteration ahead through staging
// registers, the row's x / add1 / add2 at the top of its own iteration -- and there is no room for more next to 96 VGPRs of W^T:
// 100 us per launch = 3.1 TB/s, latency-bound (Little: 49 KB / ~3.4 us per CU).  Here every operand of a tile -- A [16, K], x
// [16, 192] float32, add1 / add2 [16, 192] bf16, mean / rstd [16] -- goes HBM -> LDS with global_load_lds_dwordx4 (no staging
// registers) into a ring of THREE stages, two of them in flight while the third is consumed (~100 KB per CU).  A stage is a
// lane-linear sequence of 16-byte slots (the DMA writes wave-base + lane * 16): rows of A padded by two dummy slots (ROWB = 2K + 32,
// the conflict-free fragment stride), rows of x by two (800 B), rows of add1 / add2 by two (416 B); 12 waves x 5 instructions cover
// the 52 KB (invalid slots re-load a valid address); mean / rstd ride in the spare half of the A image's last KB through a sixth,
// exec-masked instruction; every wave issues exactly six per stage so the counted waits are uniform.
// One barrier per tile, as before:  issue DMA(t+2) | fragment reads + MFMA chain (stage t) | operand reads -> registers, row
// partials -> red table | s_waitcnt vmcnt(6): own part of DMA(t+1) landed (loads retire in order, so "at most the 6 youngest
// outstanding" is exact even with the dx store in the stream) | s_barrier: every wave's part of stage t+1 landed, the red table is
// complete, nobody reads stage t any more (so DMA(t+3) may overwrite it next iteration) | row sums, dx, store.
// Every LDS access of the loop is inline assembly: for a compiler-visible LDS read hipcc drains s_waitcnt vmcnt(0) behind an
// LDS-DMA (it may alias), which would serialise the ring; the waits are spelled out and tied to the registers they cover.
// Requires M % 16 == 0 (no partial tile: no masking, no clamped sources); other M run the register-staged kernel above.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(LDS_PTR(char))(char*)p; }
__device__ __forceinline__ u32x4 ds_read128_asm(unsigned addr, int off) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
  return v;
}
template <int OFF> __device__ __forceinline__ u32x4 ds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ u32x2 ds_read64_a(unsigned addr) { u32x2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); return v; }
__device__ __forceinline__ float ds_read32_a(unsigned addr) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr)); return v; }
__device__ __forceinline__ void ds_write64_a(unsigned addr, f32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// "these registers are valid from here": ties a spelled-out wait to the values it covers so nothing consuming them moves above it
#define TIE4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

template <int KT, bool HAS2, bool XLOW>
__global__ __launch_bounds__(768) void k_gemm_wsn_lnbwd_dma(LnbArgs g) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int K = KT * 32, D = 192, NWV = 12;
  constexpr int ROWB = K * 2 + 32, SA = ROWB / 16;            // A row: K/8 data slots + 2 pad slots
  constexpr int NA = (16 * SA + 63) / 64;                      // DMA wave-instructions of the A image
  constexpr int XSZ = XLOW ? 2 : 4, XSL = D * XSZ / 16;        // bytes per element of x (bf16 | float32 residual stream); data slots per row
  constexpr int XS = XSL + 2, XB = XS * 16, NX = (16 * XS + 63) / 64;      // x rows: 48 + 2 slots (800 B) | 24 + 2 (416 B)
  constexpr int PS = 26, PB = PS * 16, NP = (16 * PS + 63) / 64;      // add rows: 24 + 2 slots (416 B)
  constexpr int I_X = NA, I_1 = I_X + NX, I_2 = I_1 + NP, NI = I_2 + NP;
  constexpr int STAGE = NI * 1024;
  // mean / rstd of the tile's 16 rows ride in the spare upper half of the A image's last KB (A fills lanes 0..31 of it for both K):
  // a sixth, exec-masked instruction per wave (lanes 32..35 mean, 36..39 rstd; every wave issues it so the counts stay uniform)
  static_assert((16 * SA) % 64 == 32, "spare half block behind the A image");
  constexpr int MR_OFF = (NA - 1) * 1024 + 512;
  static_assert(NI <= 5 * NWV && 3 * STAGE + 2 * 16 * NWV * 8 + D * 4 <= 160 * 1024, "ring does not fit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sRed = reinterpret_cast<float*>(smem + 3 * STAGE);    // [2][16 rows][12 waves][2]
  const int tid = threadIdx.x, lane = tid & 63, gq = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.W);
  T* __restrict__ dx = reinterpret_cast<T*>(g.dx);
  const int ntiles = g.M / 16;
  const int n = w * 16 + gq * 4;

  typename MM::Frag bf[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks)
    bf[ks] = __builtin_bit_cast(typename MM::Frag, *reinterpret_cast<const u32x4*>(W + (size_t)(w * 16 + li) * K + (ks * 4 + gq) * 8));
  float* sGam = sRed + 2 * 16 * NWV * 2;                       // gamma [192] in the last spare 768 bytes: re-read per tile, 4 VGPRs less
  if (tid < D) sGam[tid] = g.gamma[tid];
  const unsigned gaddr = lds_addr(sGam) + (unsigned)n * 4u;
  const float a1 = g.a1 ? *g.a1 : 1.f, a2 = g.a2 ? *g.a2 : 1.f;
  const bool has1 = g.add1 != nullptr;
  constexpr bool has2 = HAS2;             // add2 (and <add2, x>) compiled out of LayerNorm2's instance: fewer live registers
  f32x4 dgam = {0.f, 0.f, 0.f, 0.f}, dbet = {0.f, 0.f, 0.f, 0.f};
  float dotA = 0.f, dotB = 0.f;
  constexpr float invD = 1.0f / (float)D;

  // ---- this wave's five DMA instructions per stage: wave-uniform region (base pointer, bytes per tile, LDS offset) + lane offset
  const char* rb[5]; unsigned rstride[5], rdst[5], loff[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    int I = q * NWV + w;
    const bool dup = I >= NI || (I >= I_1 && I < I_2 && !has1) || (I >= I_2 && !has2);
    if (dup) I = w;                                   // re-issue this wave's first A instruction: same source, same slots
    const int sl = (I - (I < I_X ? 0 : I < I_1 ? I_X : I < I_2 ? I_1 : I_2)) * 64 + lane;
    rdst[q] = (unsigned)I * 1024u;
    if (I < I_X) {
      const int row = sl / SA, c = sl % SA;
      rb[q] = reinterpret_cast<const char*>(g.A); rstride[q] = 16u * K * 2u;
      loff[q] = row < 16 ? (unsigned)(row * K * 2 + (c < K / 8 ? c : 0) * 16) : 0u;
    } else if (I < I_1) {
      const int row = sl / XS, c = sl % XS;
      rb[q] = reinterpret_cast<const char*>(g.x); rstride[q] = 16u * D * (unsigned)XSZ;
      loff[q] = row < 16 ? (unsigned)(row * D * XSZ + (c < XSL ? c : 0) * 16) : 0u;
    } else {
      const int row = sl / PS, c = sl % PS;
      rb[q] = reinterpret_cast<const char*>(I < I_2 ? g.add1 : g.add2); rstride[q] = 16u * D * 2u;
      loff[q] = row < 16 ? (unsigned)(row * D * 2 + (c < 24 ? c : 0) * 16) : 0u;
    }
  }
  const long long mr_delta = reinterpret_cast<const char*>(g.rstd) - reinterpret_cast<const char*>(g.mean);     // scalar
  const unsigned mr_lo = (unsigned)(lane & 3) * 16u;
  const bool mr_lane = lane >= 32 && lane < 40, mr_r = lane >= 36;
  // DMA instruction q of a stage (q = 5: the tile's mean / rstd)
  auto issue_q = [&](int tile, int st, int q) {
    const int t = tile < ntiles ? tile : ntiles - 1;            // past the end: redundant loads keep the instruction count uniform
    if (q < 5) {
      // (tile * bytes-per-tile + lane offset) is formed per issue and added to the SCALAR region base: written as base + lane offset
      // first, hipcc hoists six loop-invariant 64-bit per-lane pointers (12 VGPRs) and spills W^T fragments to make room
      const char* src = rb[q] + ((unsigned long long)(unsigned)t * (unsigned long long)rstride[q] + (unsigned long long)loff[q]);
      char* dst = smem + st * STAGE + rdst[q];
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    } else if (mr_lane) {
      const char* src = reinterpret_cast<const char*>(g.mean) + ((unsigned long long)(unsigned)t * 64ull + mr_lo) + (mr_r ? mr_delta : 0ll);
      char* dst = smem + st * STAGE + (NA - 1) * 1024;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };
  auto issue = [&](int tile, int st) {
#pragma unroll
    for (int q = 0; q < 6; ++q) issue_q(tile, st, q);
  };
  const unsigned s0 = lds_addr(smem);
  const unsigned fragoff = (unsigned)(li * ROWB + gq * 16);
  const unsigned xoff = (unsigned)(I_X * 1024 + li * XB + n * XSZ), poff = (unsigned)(I_1 * 1024 + li * PB + n * 2), moff = (unsigned)(MR_OFF + li * 4);
  const unsigned redr = lds_addr(sRed) + (unsigned)(li * NWV * 8);

  int tile = blockIdx.x;
  issue(tile, 0);
  issue(tile + gridDim.x, 1);
  wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  int st = 0, par = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    // stage (it + 2) % 3 (nobody reads it since the last barrier) is requested BETWEEN the groups of the MFMA chain: issued as a burst in
    // front of it, the six instructions cost the wave 100-200 cycles each with nothing of its own in the matrix pipe
    const int nt_ = tile + 2 * (int)gridDim.x, ns_ = st == 0 ? 2 : st - 1;
    const unsigned sb = s0 + (unsigned)(st * STAGE);
    // ---- MFMA chain: fragments in pairs, the next pair requested before the current one is waited for (16 VGPRs of fragments
    //      next to the 96 of W^T; groups of four spilled 16 registers at K = 768 and ran 132 us against 104 register-staged)
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    constexpr bool DB = KT <= 18;                     // K = 576: room for a second fragment pair (81 us against 97 single-buffered)
    u32x4 fa[2], fb[2];
    const unsigned fr = sb + fragoff;
#define RD2(dst, ks0) dst[0] = ds_read128<(ks0) * 64>(fr); dst[1] = ds_read128<(ks0) * 64 + 64>(fr);
#define GROUP(gk, cur, nxt) if ((gk) < KT / 2) { \
      if (DB) { if ((gk) + 1 < KT / 2) { RD2(nxt, ((gk) + 1) * 2) wait_lgkm<2>(); } else wait_lgkm<0>(); \
                asm volatile("" : "+v"(cur[0]), "+v"(cur[1])); \
                c0 = MM::mma(bf[(gk) * 2], __builtin_bit_cast(typename MM::Frag, cur[0]), c0); \
                c0 = MM::mma(bf[(gk) * 2 + 1], __builtin_bit_cast(typename MM::Frag, cur[1]), c0); } \
      else if (((gk) & 1) == 0) { RD2(fa, (gk) * 2) RD2(fb, (gk) * 2 + 2) wait_lgkm<0>(); /* K = 768: four reads per wait, single-buffered */ \
             asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1])); \
             c0 = MM::mma(bf[(gk) * 2], __builtin_bit_cast(typename MM::Frag, fa[0]), c0); \
             c0 = MM::mma(bf[(gk) * 2 + 1], __builtin_bit_cast(typename MM::Frag, fa[1]), c0); \
             c0 = MM::mma(bf[(gk) * 2 + 2], __builtin_bit_cast(typename MM::Frag, fb[0]), c0); \
             c0 = MM::mma(bf[(gk) * 2 + 3], __builtin_bit_cast(typename MM::Frag, fb[1]), c0); } }
    static_assert(KT % 2 == 0 && KT <= 24, "pairs of k-steps");
    if (DB) { RD2(fa, 0) }
    GROUP(0, fa, fb) issue_q(nt_, ns_, 0); GROUP(1, fb, fa) GROUP(2, fa, fb) issue_q(nt_, ns_, 1); GROUP(3, fb, fa)
    GROUP(4, fa, fb) issue_q(nt_, ns_, 2); GROUP(5, fb, fa) GROUP(6, fa, fb) issue_q(nt_, ns_, 3); GROUP(7, fb, fa)
    GROUP(8, fa, fb) issue_q(nt_, ns_, 4); issue_q(nt_, ns_, 5); GROUP(9, fb, fa) GROUP(10, fa, fb) GROUP(11, fb, fa)
#undef GROUP
#undef RD2
    // ---- this row's LayerNorm operands from the stage (they stay in registers across the barrier)
    typename Resid<typename std::conditional<XLOW, bf16_t, float>::type>::Raw4 xr;
    if constexpr (XLOW) asm volatile("ds_read_b64 %0, %1" : "=v"(xr) : "v"(sb + xoff));
    else xr = ds_read128<0>(sb + xoff);
    u32x2 r1 = {0u, 0u}, r2 = {0u, 0u};
    if (has1) asm volatile("ds_read_b64 %0, %1" : "=v"(r1) : "v"(sb + poff));
    if (has2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(sb + poff), "n"((I_2 - I_1) * 1024));
    float mean, rstd;
    asm volatile("ds_read_b32 %0, %1" : "=v"(mean) : "v"(sb + moff));
    asm volatile("ds_read_b32 %0, %1 offset:64" : "=v"(rstd) : "v"(sb + moff));
    u32x4 gr = ds_read128<0>(gaddr);
    wait_lgkm<0>();
    asm volatile("" : "+v"(xr), "+v"(r1), "+v"(r2), "+v"(mean), "+v"(rstd), "+v"(gr));
    const f32x4 xv = Resid<typename std::conditional<XLOW, bf16_t, float>::type>::f4(xr);
    f32x4 gam = __builtin_bit_cast(f32x4, gr);
    {
      float p1 = 0.f, p2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        const float gy = c0[e] * gam[e];
        dgam[e] += c0[e] * xh;
        dbet[e] += c0[e];
        p1 += gy;
        p2 += gy * xh;
      }
      p1 = sum_rows4(p1);
      p2 = sum_rows4(p2);
      if (gq == 0) ds_write64_a(redr + (unsigned)(par * 16 * NWV * 8 + w * 8), f32x2{p1, p2});
#pragma unroll
      for (int e = 0; e < 4; ++e) c0[e] *= gam[e];
    }
    wait_vm<6>();                                               // own part of the NEXT stage has landed
    wait_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    float c1 = 0.f, c2 = 0.f;
    {
      const unsigned rr = redr + (unsigned)(par * 16 * NWV * 8);
      // 12 (p1, p2) pairs of this row, in wave order; two batches of three reads (12 VGPRs in flight instead of 24)
      u32x4 t0 = ds_read128<0>(rr), t1 = ds_read128<16>(rr), t2 = ds_read128<32>(rr);
      wait_lgkm<0>();
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));
      {
        const f32x4 a = __builtin_bit_cast(f32x4, t0), b = __builtin_bit_cast(f32x4, t1), c = __builtin_bit_cast(f32x4, t2);
        c1 += a[0]; c2 += a[1]; c1 += a[2]; c2 += a[3]; c1 += b[0]; c2 += b[1]; c1 += b[2]; c2 += b[3]; c1 += c[0]; c2 += c[1]; c1 += c[2]; c2 += c[3];
      }
      t0 = ds_read128<48>(rr); t1 = ds_read128<64>(rr); t2 = ds_read128<80>(rr);
      wait_lgkm<0>();
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2));
      {
        const f32x4 a = __builtin_bit_cast(f32x4, t0), b = __builtin_bit_cast(f32x4, t1), c = __builtin_bit_cast(f32x4, t2);
        c1 += a[0]; c2 += a[1]; c1 += a[2]; c2 += a[3]; c1 += b[0]; c2 += b[1]; c1 += b[2]; c2 += b[3]; c1 += c[0]; c2 += c[1]; c1 += c[2]; c2 += c[3];
      }
    }
    c1 *= invD; c2 *= invD;
    {
      const f32x4 v1 = {__uint_as_float(r1[0] << 16), __uint_as_float(r1[0] & 0xffff0000u), __uint_as_float(r1[1] << 16), __uint_as_float(r1[1] & 0xffff0000u)};
      const f32x4 v2 = {__uint_as_float(r2[0] << 16), __uint_as_float(r2[0] & 0xffff0000u), __uint_as_float(r2[1] << 16), __uint_as_float(r2[1] & 0xffff0000u)};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        o[e] = rstd * (c0[e] - c1 - xh * c2);                   // c0 already holds dy * gamma (scaled before the barrier)
        if (has1) o[e] += a1 * v1[e];
        if (has2) { o[e] += a2 * v2[e]; dotB += v2[e] * xv[e]; }
        dotA += o[e] * xv[e];
      }
      u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]);
      *reinterpret_cast<u32x2*>(dx + ((size_t)(tile * 16 + li) * D + n)) = q;
    }
    st = st == 2 ? 0 : st + 1;
    par ^= 1;
  }
  wait_vm<0>();                                                 // the redundant tail stages
  __syncthreads();
  // ---- dgamma / dbeta: sum over the 16 row lanes; dots over the workgroup (fixed order)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { dgam[e] += __shfl_xor(dgam[e], o, 64); dbet[e] += __shfl_xor(dbet[e], o, 64); }
  }
  float* P = g.partial + (size_t)blockIdx.x * (2 * D + 2);
  if (li == 0) {
    *reinterpret_cast<f32x2*>(P + n) = f32x2{dgam[0], dgam[1]}; *reinterpret_cast<f32x2*>(P + n + 2) = f32x2{dgam[2], dgam[3]};
    *reinterpret_cast<f32x2*>(P + D + n) = f32x2{dbet[0], dbet[1]}; *reinterpret_cast<f32x2*>(P + D + n + 2) = f32x2{dbet[2], dbet[3]};
  }
  dotA = wave_sum(dotA); dotB = wave_sum(dotB);
  if (lane == 0) { sRed[2 * w] = dotA; sRed[2 * w + 1] = dotB; }
  __syncthreads();
  if (tid < 2) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NWV; ++q) t += sRed[2 * q + tid];
    P[2 * D + tid] = t;
  }
}
#undef TIE4

// ---- fc2 (+ bias + residual [+ gate mix]) on the same LDS-DMA ring ------------------------------------------------------------
// k_gemm_wsn16<float, RESID / RESID_GATE> keeps one 16-row tile of A and of the residual rows in flight per CU (register-staged, one
// tile ahead): 87-89 us for 387 MB = 4.4 TB/s.  Same ring as k_gemm_wsn_lnbwd_dma: A [16, K] bf16 and the float32 residual rows
// (R = x1, R2 = x_l) go HBM -> LDS by global_load_lds_dwordx4 into three stages, two in flight; one barrier per tile; the epilogue
// runs from registers after it.  The accumulation is the single k-ordered chain and the epilogue the same fmaf sequence as every other
// NT kernel, so the output bits do not depend on which kernel a problem size selects (tests/test_fullsize_gpu.py).  M % 16 == 0.
// RLOW: the residual stream is bf16 -- R, R2 and C are bf16 rows (24 + 2 slots of 16 bytes in a stage instead of 48 + 2; the result is
// rounded once, at the store, and the LayerNorm that follows is taken of the ROUNDED row: what the consumers of C will read).
template <int EPI, int KT, bool LN, int NST, bool RLOW>
__global__ __launch_bounds__(768) void k_gemm_wsn16_dma(NtArgs g) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int K = KT * 32, D = 192, NWV = 12;
  constexpr int ROWB = K * 2 + 32, SA = ROWB / 16;
  constexpr int NA = (16 * SA + 63) / 64;
  constexpr int RSZ = RLOW ? 2 : 4, RSL = D * RSZ / 16;         // bytes per residual element; data slots per residual row
  constexpr int XS = RSL + 2, XB = XS * 16, NX = (16 * XS + 63) / 64;
  constexpr bool GATE = EPI == UVC_EPI_BIAS_RESID_GATE;
  constexpr int I_R = NA, I_R2 = I_R + NX, NI = I_R2 + (GATE ? NX : 0);
  constexpr int STAGE = NI * 1024;
  constexpr int NPW = (NI + NWV - 1) / NWV;                   // DMA instructions per wave and stage (surplus ones repeat the wave's first)
  // LN: behind the ring, the row-statistics table [2][16 rows][12 waves][2] and gamma / beta of the LayerNorm that follows
  constexpr int LN_BYTES = LN ? 2 * 16 * NWV * 8 + 3 * D * 4 : 0;      // + gamma, beta, bias
  static_assert(NST >= 3 && NPW <= 5 && NST * STAGE + LN_BYTES <= 160 * 1024, "ring does not fit");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, gq = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* __restrict__ W = reinterpret_cast<const T*>(g.B);
  char* __restrict__ C = reinterpret_cast<char*>(g.C);
  const int ntiles = (g.M + 15) / 16;                         // M >= 16; a ragged last tile is the LAST 16 rows (it overlaps its neighbour:
                                                              // those rows are computed twice from the same operands, same bits, same stores)
  const int n = w * 16 + gq * 4;
  float* const sRed = reinterpret_cast<float*>(smem + NST * STAGE);
  float* const sGB = sRed + 2 * 16 * NWV * 2;
  if (LN) {
    if (tid < D) { sGB[tid] = g.ln_gamma[tid]; sGB[D + tid] = g.ln_beta[tid]; sGB[2 * D + tid] = g.bias[tid]; }      // before the first DMA is issued (see k_gemm_wsn_lnbwd_dma)
  }
  const unsigned gaddr = lds_addr(sGB) + (unsigned)n * 4u;
  const unsigned redr = lds_addr(sRed) + (unsigned)(li * NWV * 8);
  typename MM::Frag bf[KT];
#pragma unroll
  for (int ks = 0; ks < KT; ++ks)
    bf[ks] = __builtin_bit_cast(typename MM::Frag, *reinterpret_cast<const u32x4*>(W + (size_t)(w * 16 + li) * K + (ks * 4 + gq) * 8));
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (!LN) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);      // LN: re-read from LDS per tile (four VGPRs less across the MFMA chain)

  const char* rb[NPW]; unsigned rstride[NPW], rdst[NPW], loff[NPW];
#pragma unroll
  for (int q = 0; q < NPW; ++q) {
    int I = q * NWV + w;
    if (I >= NI) I = w;                               // duplicate of this wave's first A instruction: uniform count per wave
    const int sl = (I - (I < I_R ? 0 : I < I_R2 ? I_R : I_R2)) * 64 + lane;
    rdst[q] = (unsigned)I * 1024u;
    if (I < I_R) {
      const int row = sl / SA, c = sl % SA;
      rb[q] = reinterpret_cast<const char*>(g.A); rstride[q] = K * 2u;
      loff[q] = row < 16 ? (unsigned)(row * K * 2 + (c < K / 8 ? c : 0) * 16) : 0u;
    } else {
      const int row = sl / XS, c = sl % XS;
      rb[q] = reinterpret_cast<const char*>(I < I_R2 ? g.R : g.R2); rstride[q] = (unsigned)(D * RSZ);
      loff[q] = row < 16 ? (unsigned)(row * D * RSZ + (c < RSL ? c : 0) * 16) : 0u;
    }
  }
  auto issue = [&](int tile, int st) {
    const int t = tile < ntiles ? tile : ntiles - 1;
    const unsigned r0 = (unsigned)min(t * 16, g.M - 16);          // first row of the tile
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const char* src = rb[q] + ((unsigned long long)r0 * (unsigned long long)rstride[q] + (unsigned long long)loff[q]);
      char* dst = smem + st * STAGE + rdst[q];
      // (nt on these loads -- fc2's GELU(a) operand only, or every operand of the kernel: 0.3-0.6 % slower in the step, profiles/r5zz_ab_nt_wsn.txt)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    }
  };
  const unsigned s0 = lds_addr(smem);
  const unsigned fragoff = (unsigned)(li * ROWB + gq * 16), roff = (unsigned)(I_R * 1024 + li * XB + n * RSZ);

  // NST stages, NST - 1 in flight: stage t + NST - 1 is requested at the top of iteration t into the image iteration t - 1 read
  int tile = blockIdx.x;
#pragma unroll
  for (int q = 0; q < NST - 1; ++q) issue(tile + q * gridDim.x, q);
  wait_vm<(NST - 2) * NPW>();
  __builtin_amdgcn_s_barrier();
  int st = 0, par = 0;
  // LN: the normalisation of tile t is DEFERRED into iteration t + 1.  Its operands (the 12 pairs of the statistics table, gamma, beta)
  // are requested in three small batches between the first k-step groups of tile t + 1 and consumed under its MFMAs (inside one wave the
  // VALU work fills the matrix pipe's issue gaps); as a chain of LDS round trips behind the barrier it cost 20 us of 100.
  bool have_prev = false;
  float po[4] = {0.f, 0.f, 0.f, 0.f};
  int prow0 = 0;
  float lm2 = 0.f, ls1 = 0.f, ls2 = 0.f, lpv = 0.f;
  u32x4 tq0, tq1, lgr, lbr;
  auto ln_issue = [&](auto offv) {
    constexpr int OFF = decltype(offv)::value;
    const unsigned ra = redr + (unsigned)((par ^ 1) * 16 * NWV * 8);
    tq0 = ds_read128<OFF>(ra); tq1 = ds_read128<OFF + 16>(ra);
  };
  auto ln_first = [&]() {
    asm volatile("" : "+v"(tq0), "+v"(tq1));
    const f32x4 a0 = __builtin_bit_cast(f32x4, tq0), a1 = __builtin_bit_cast(f32x4, tq1);
    lpv = a0[0];
    float dd;
    lm2 = a0[1]; dd = a0[2] - lpv; ls1 = dd; ls2 = dd * dd; lm2 += a0[3];
    dd = a1[0] - lpv; ls1 += dd; ls2 += dd * dd; lm2 += a1[1]; dd = a1[2] - lpv; ls1 += dd; ls2 += dd * dd; lm2 += a1[3];
    asm volatile("" : "+v"(lm2), "+v"(ls1), "+v"(ls2), "+v"(lpv));
  };
  auto ln_acc = [&]() {
    asm volatile("" : "+v"(tq0), "+v"(tq1));
    const f32x4 a0 = __builtin_bit_cast(f32x4, tq0), a1 = __builtin_bit_cast(f32x4, tq1);
    float dd;
    dd = a0[0] - lpv; ls1 += dd; ls2 += dd * dd; lm2 += a0[1]; dd = a0[2] - lpv; ls1 += dd; ls2 += dd * dd; lm2 += a0[3];
    dd = a1[0] - lpv; ls1 += dd; ls2 += dd * dd; lm2 += a1[1]; dd = a1[2] - lpv; ls1 += dd; ls2 += dd * dd; lm2 += a1[3];
    asm volatile("" : "+v"(lm2), "+v"(ls1), "+v"(ls2));
  };
  // mean = pivot + s1/12;  sum (m_w - mean)^2 = s2 - 12 (s1/12)^2;  var = (sum M2_w + 16 * that) / D  (Chan et al.; see the epilogue)
  auto ln_finish = [&]() {
    asm volatile("" : "+v"(lgr), "+v"(lbr));
    const float dm = ls1 * (1.0f / 12.0f);
    const float mean = lpv + dm;
    const float dv = ls2 - 12.0f * dm * dm;
    const float rstd = rsqrtf((lm2 + 16.0f * dv) * (1.0f / (float)D) + g.ln_eps);
    const f32x4 gam = __builtin_bit_cast(f32x4, lgr), bet = __builtin_bit_cast(f32x4, lbr);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (po[e] - mean) * rstd * gam[e] + bet[e];
    u32x2 q; q[0] = pack_bf16x2(y[0], y[1]); q[1] = pack_bf16x2(y[2], y[3]);
    const size_t trow = (size_t)prow0;
    *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(reinterpret_cast<T*>(g.ln_out) + trow * D) + (unsigned)((li * D + n) * 2)) = q;
    if (g.ln_mean && w == 0 && gq == 0) {
      *reinterpret_cast<float*>(reinterpret_cast<char*>(g.ln_mean + trow) + (unsigned)(li * 4)) = mean;
      *reinterpret_cast<float*>(reinterpret_cast<char*>(g.ln_rstd + trow) + (unsigned)(li * 4)) = rstd;
    }
  };
  for (; tile < ntiles; tile += gridDim.x) {
    issue(tile + (NST - 1) * gridDim.x, st == 0 ? NST - 1 : st - 1);
    const unsigned sb = s0 + (unsigned)(st * STAGE);
    f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
    u32x4 fa[2], fb[2];
    const unsigned fr = sb + fragoff;
    if (LN && have_prev) ln_issue(std::integral_constant<int, 0>{});
#define RD2(dst, ks0) dst[0] = ds_read128<(ks0) * 64>(fr); dst[1] = ds_read128<(ks0) * 64 + 64>(fr);
#define GROUP(gk, cur, nxt) if ((gk) < KT / 2) { \
      if ((gk) + 1 < KT / 2) { RD2(nxt, ((gk) + 1) * 2) wait_lgkm<2>(); } else wait_lgkm<0>(); \
      asm volatile("" : "+v"(cur[0]), "+v"(cur[1])); \
      c0 = MM::mma(bf[(gk) * 2], __builtin_bit_cast(typename MM::Frag, cur[0]), c0); \
      c0 = MM::mma(bf[(gk) * 2 + 1], __builtin_bit_cast(typename MM::Frag, cur[1]), c0); }
    static_assert(KT % 2 == 0 && KT <= 24, "pairs of k-steps");
    RD2(fa, 0)
    GROUP(0, fa, fb)                       // its wait (two youngest reads may be out) covers the older table reads
    if (LN && have_prev) { ln_first(); ln_issue(std::integral_constant<int, 32>{}); }
    GROUP(1, fb, fa)
    if (LN && have_prev) { ln_acc(); ln_issue(std::integral_constant<int, 64>{}); }
    GROUP(2, fa, fb)
    if (LN && have_prev) {
      ln_acc();
      if constexpr (KT / 2 > 3) { lgr = ds_read128<0>(gaddr); lbr = ds_read128<D * 4>(gaddr); }      // gamma, beta: consumed one group on (registers)
      else { lgr = ds_read128<0>(gaddr); lbr = ds_read128<D * 4>(gaddr); wait_lgkm<0>(); ln_finish(); }
    }
    GROUP(3, fb, fa)
    if (LN && have_prev && KT / 2 > 3) ln_finish();
    GROUP(4, fa, fb) GROUP(5, fb, fa)
    GROUP(6, fa, fb) GROUP(7, fb, fa) GROUP(8, fa, fb) GROUP(9, fb, fa) GROUP(10, fa, fb) GROUP(11, fb, fa)
#undef GROUP
#undef RD2
    // the row's residual operands: four consecutive columns = 16 bytes (float32 stream) or 8 bytes (bf16 stream)
    typename Resid<typename std::conditional<RLOW, bf16_t, float>::type>::Raw4 rr, rr2;
    if constexpr (RLOW) {
      asm volatile("ds_read_b64 %0, %1" : "=v"(rr) : "v"(sb + roff));
      rr2 = rr;
      if (GATE) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rr2) : "v"(sb + roff), "n"(NX * 1024));
    } else {
      rr = ds_read128<0>(sb + roff); rr2 = rr;
      if (GATE) rr2 = ds_read128<NX * 1024>(sb + roff);
    }
    typedef Resid<typename std::conditional<RLOW, bf16_t, float>::type> RS;
    if constexpr (!LN) {
      wait_vm<(NST - 2) * NPW>();                                 // own part of the next stage has landed
      wait_lgkm<0>();
      asm volatile("" : "+v"(rr), "+v"(rr2));
      __builtin_amdgcn_s_barrier();
      const f32x4 r = RS::f4(rr), r2 = RS::f4(rr2);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = epi_scale_bias(c0[e], alpha, bias4[e]);
        o[e] += r[e];
        if (GATE) o[e] = epi_gate_mix(o[e], r2[e], d0, d1);
      }
      char* cp = C + ((size_t)(min(tile * 16, g.M - 16) + li) * g.ldc + n) * RSZ;
      if constexpr (RLOW) { u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]); *reinterpret_cast<u32x2*>(cp) = q; }
      else *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
    } else {
      // The output row is spread over the 12 waves (16 columns each).  Every wave leaves (mean, centred sum of squares) of its 16
      // columns in the table BEFORE the tile's one barrier and picks up the 12 pairs after it; they combine exactly (Chan et al.):
      // mean = sum m_w / 12, M2 = sum M2_w + 16 sum (m_w - mean)^2 -- the two-pass variance of k_ln_fwd_v without a second exchange.
      u32x4 bq = ds_read128<2 * D * 4>(gaddr);
      wait_lgkm<0>();
      asm volatile("" : "+v"(rr), "+v"(rr2), "+v"(bq));
      const f32x4 r = RS::f4(rr), r2 = RS::f4(rr2), bv = __builtin_bit_cast(f32x4, bq);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = epi_scale_bias(c0[e], alpha, bv[e]);
        o[e] += r[e];
        if (GATE) o[e] = epi_gate_mix(o[e], r2[e], d0, d1);
      }
      u32x2 oq = {0u, 0u};
      if constexpr (RLOW) {                       // the stored (rounded) row is what the LayerNorm and every later consumer see
        oq[0] = pack_bf16x2(o[0], o[1]); oq[1] = pack_bf16x2(o[2], o[3]);
        const f32x4 t = Resid<bf16_t>::f4(oq);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = t[e];
      }
      {
        const float sm = sum_rows4((o[0] + o[1]) + (o[2] + o[3]));
        const float mw = sm * (1.0f / 16.0f);
        float qw = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float dd = o[e] - mw; qw += dd * dd; }
        qw = sum_rows4(qw);
        if (gq == 0) ds_write64_a(redr + (unsigned)(par * 16 * NWV * 8 + w * 8), f32x2{mw, qw});
      }
      // the output rows leave now; their normalisation follows inside the next iteration (or behind the loop)
      prow0 = min(tile * 16, g.M - 16);
      if constexpr (RLOW) *reinterpret_cast<u32x2*>(C + (size_t)prow0 * g.ldc * RSZ + (unsigned)((li * g.ldc + n) * RSZ)) = oq;
      else *reinterpret_cast<f32x4*>(C + (size_t)prow0 * g.ldc * RSZ + (unsigned)((li * g.ldc + n) * RSZ)) = f32x4{o[0], o[1], o[2], o[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) po[e] = o[e];
      have_prev = true;
      wait_vm<(NST - 2) * NPW>();                                 // own part of the next stage has landed
      wait_lgkm<0>();
      __builtin_amdgcn_s_barrier();
      par ^= 1;
    }
    st = st == NST - 1 ? 0 : st + 1;
  }
  if (LN && have_prev) {                                         // the last tile of this workgroup
    ln_issue(std::integral_constant<int, 0>{});
    wait_lgkm<0>(); ln_first();
    ln_issue(std::integral_constant<int, 32>{});
    wait_lgkm<0>(); ln_acc();
    ln_issue(std::integral_constant<int, 64>{}); lgr = ds_read128<0>(gaddr); lbr = ds_read128<D * 4>(gaddr);
    wait_lgkm<0>(); ln_acc(); ln_finish();
  }
  wait_vm<0>();
}

int launch_row384_lnbwd(const LnbArgs& a, int x_lowp, hipStream_t st);

extern "C" int uvc_gemm_lnbwd_supported(int32_t M, int32_t D, int32_t K, int32_t dtype) {
  if (dtype != UVC_BF16 || M < 4096) return 0;
  if (D == 192) return K == 768 || K == 576;
  return D == 384 && K % 64 == 0 && K >= 128 && (int64_t)M * K < (1ll << 30);      // the row-tile kernel (DeiT-Small, T2T-ViT-14)
}
extern "C" int uvc_gemm_lnbwd_nblocks(int32_t M) { const int nt = ceil_div(M, 16); return nt < 256 ? nt : 256; }

extern "C" int uvc_gemm_nt_lnbwd(const uvc_gemm_lnbwd_args* p, void* stream) {
  if (!p || !p->A || !p->W || !p->x || !p->mean || !p->rstd || !p->gamma || !p->dx || !p->partial)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt_lnbwd: null pointer");
  if (!uvc_gemm_lnbwd_supported(p->M, p->D, p->K, p->dtype))
    return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_gemm_nt_lnbwd: bf16, M >= 4096, D = 192 with K in {576, 768} or D = 384 with K % 64 == 0");
  if ((((uintptr_t)p->A | (uintptr_t)p->W | (uintptr_t)p->x | (uintptr_t)p->gamma) & 15) != 0 ||
      (((uintptr_t)p->add1 | (uintptr_t)p->add2 | (uintptr_t)p->dx | (uintptr_t)p->partial) & 7) != 0)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt_lnbwd: misaligned buffer");
  LnbArgs a;
  a.A = p->A; a.W = p->W; a.x = p->x; a.mean = p->mean; a.rstd = p->rstd; a.gamma = p->gamma; a.add1 = p->add1; a.a1 = p->a1;
  a.add2 = p->add2; a.a2 = p->a2; a.dx = p->dx; a.partial = p->partial; a.M = p->M; a.K = p->K; a.want_dots = 1;
  const int grid = uvc_gemm_lnbwd_nblocks(p->M);
  hipStream_t st = (hipStream_t)stream;
  if (p->D == 384) return launch_row384_lnbwd(a, p->x_lowp, st);
#define LNB_LAUNCH(KT_) { \
    const size_t sh = (size_t)2 * 16 * (KT_ * 64 + 32) + (2 * 16 * 12 * 2 + 192) * sizeof(float); \
    if (p->x_lowp) { UVC_MAX_LDS(sh, k_gemm_wsn_lnbwd<KT_, true>); k_gemm_wsn_lnbwd<KT_, true><<<grid, 768, sh, st>>>(a); } \
    else { UVC_MAX_LDS(sh, k_gemm_wsn_lnbwd<KT_, false>); k_gemm_wsn_lnbwd<KT_, false><<<grid, 768, sh, st>>>(a); } }
  const bool use_dma = p->variant != 1;
  const bool al16 = (((uintptr_t)p->add1 | (uintptr_t)p->add2 | (uintptr_t)p->mean | (uintptr_t)p->rstd) & 15) == 0;
#define LNB_DMA_ONE(KT_, H2_, XL_) { UVC_MAX_LDS(sh, k_gemm_wsn_lnbwd_dma<KT_, H2_, XL_>); k_gemm_wsn_lnbwd_dma<KT_, H2_, XL_><<<grid, 768, sh, st>>>(a); }
#define LNB_LAUNCH_DMA(KT_) { \
    constexpr int NA_ = (16 * (KT_ * 4 + 2) + 63) / 64; \
    const int NI_ = NA_ + (p->x_lowp ? 7 : 13) + 7 + 7; \
    const size_t sh = (size_t)3 * NI_ * 1024 + (2 * 16 * 12 * 2 + 192) * sizeof(float); \
    if (p->x_lowp) { if (p->add2) LNB_DMA_ONE(KT_, true, true) else LNB_DMA_ONE(KT_, false, true) } \
    else { if (p->add2) LNB_DMA_ONE(KT_, true, false) else LNB_DMA_ONE(KT_, false, false) } }
  if (use_dma && al16 && p->M % 16 == 0) { if (p->K == 768) LNB_LAUNCH_DMA(24) else LNB_LAUNCH_DMA(18) }
  else if (p->K == 768) LNB_LAUNCH(24) else LNB_LAUNCH(18)
#undef LNB_LAUNCH_DMA
#undef LNB_DMA_ONE
#undef LNB_LAUNCH
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// ================================================================================================
//            NT, 256 x 256 output tile, LDS-DMA double buffer (bf16; the wide models' GEMMs)
// ================================================================================================
// DeiT-Small / Base and T2T-ViT have K = 384 ... 3072 against N = 384 ... 3072: compute-bound on the matrix pipe, not streaming
// (2 M N K / (2 (M K + N K + M N)) = 300 ... 600 flop / B), and the 128 x 128 kernel above spends its time around its two barriers
// per 64-deep step (0.5-0.85 PFLOP/s).  Here a workgroup is 8 waves (2 x 4), one per CU by LDS, and owns a 256 x 256 tile: a wave
// accumulates 128 x 64 (32 MFMA tiles, 128 accumulator registers), so a 16-byte fragment read feeds four (A) or eight (B) MFMAs.
//   * Operand tiles (256 rows x 64 k = 32 KB each) go HBM / L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers,
//     rows past M / N read as zeros through the buffer descriptor's bounds check) into TWO 64-KB stages; the DMA writes a wave's 64
//     lanes as 64 consecutive 16-byte slots, so the image is linear [row][8 chunks] and the bank swizzle sits in the SOURCE address:
//     slot (row, c) holds k-chunk c ^ (row & 7), which makes the fragment reads (row = lane & 15, chunk = 4 ks + lane / 16) of every
//     ds_read_b128 lane group hit 16 different 16-byte slots of the 256-byte bank row.
//   * One barrier per 64-deep step.  Fragments of k-half 1 are requested before the MFMAs of k-half 0 and those of the NEXT stage's
//     k-half 0 before the MFMAs of k-half 1 (two fragment sets, 96 registers), and the stage after next is requested right behind the
//     barrier, a whole step before it is needed:   [read F1 | 32 MFMA F0 | wait F1, wait DMA(t+1) | barrier | DMA(t+2) | read F0' | 32 MFMA F1].
//     All LDS reads of the loop are inline assembly: behind an LDS-DMA hipcc drains vmcnt(0) in front of any LDS read it can see.
//   * The 8 DMA instructions of a stage are issued between the rows of the second MFMA block (one per four MFMAs): issued in a burst
//     behind the barrier they kept the matrix pipe idle for ~800 cycles per step (every wave of the lock-stepped workgroup at once).
//   * Epilogue through a wave-private LDS transpose, nt_epilogue_rows: the same arithmetic and stores as k_gemm_nt.
//   * One PERSISTENT workgroup per CU walks its tiles (numbered XCD-major: block b runs on XCD b % 8, so all N tiles of an M tile
//     share an L2); the first two k-steps of the next tile are requested during the last two steps of the current one, so only the
//     first tile of a workgroup pays the operand latency and the epilogue is the only gap between tiles.
// Epilogue of a wave's NH x 16 accumulator rows (acc[h][0..3], rows MROW0 + 16 h .., columns NCOL0 .. + 63) through its swizzled 4-KB transpose
// buffer `stg`, 16 rows at a time; with a bf16 C the operand rows (R / R2 / aux) of step h + PD are requested when step h is done (PD steps in
// flight; PD1 with one operand, PD2 with two: 8 registers per operand and step).  Same arithmetic and stores as every other user of nt_epilogue_rows: same bits.
#define NT_EPILOGUE_TILE(NH, MROW0, NCOL0, PD1, PD2)                                                                                 \
  {                                                                                                                           \
    constexpr bool PRE_ = sizeof(TC) == 2 && EpiOperands<EPI>::COUNT > 0;                                                     \
    constexpr int PD_ = EpiOperands<EPI>::COUNT >= 2 ? (PD2) : (PD1);                                                              \
    EpiPre<2> pre_[PD_];                                                                                                      \
    if constexpr (PRE_) {                                                                                                     \
      _Pragma("unroll") for (int h = 0; h < PD_ && h < (NH); ++h) nt_epilogue_prefetch<EPI, 16>(g, lane, (MROW0) + h * 16, (NCOL0), pre_[h]); \
    }                                                                                                                         \
    _Pragma("unroll") for (int h = 0; h < (NH); ++h) {                                                                        \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                           \
        *reinterpret_cast<f32x4*>(stg_at<true>(stg, lane & 15, j * 4 + (lane >> 4))) = acc[h][j];                            \
      __builtin_amdgcn_wave_barrier();                                                                                        \
      nt_epilogue_rows<T, TC, EPI, 16, true, PRE_>(g, stg, lane, (MROW0) + h * 16, (NCOL0), alpha, d0, d1, bias_v, &pre_[h % PD_]); \
      if constexpr (PRE_) { if (h + PD_ < (NH)) nt_epilogue_prefetch<EPI, 16>(g, lane, (MROW0) + (h + PD_) * 16, (NCOL0), pre_[h % PD_]); } \
      __builtin_amdgcn_wave_barrier();                                                                                        \
    }                                                                                                                         \
  }

constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_OPB = 256 * 128;              // bytes of one operand tile image (256 rows x 128 B)
constexpr int G2_STAGE = 2 * G2_OPB;           // A image + B image
constexpr int G2_EPI = 8 * 16 * 64 * 4;        // epilogue transpose buffers: 8 waves x 16 rows x 64 floats (swizzled, unpadded)
constexpr int G2_LDS = 2 * G2_STAGE + G2_EPI;  // 160 KB: the whole CU

struct G2Frags { u32x4 a[8]; u32x4 b[4]; };

// tile number (XCD-major order: block b runs on XCD b % 8, so the tiles v, v + 8, ... share an L2) -> tile coordinates
__device__ __forceinline__ bool g2_tile(int v, int tiles_m, int tiles_n, int& bm, int& bn) {
  const int xcd = v & 7, q = v >> 3;
  bn = q % tiles_n; bm = (q / tiles_n) * 8 + xcd;
  return bm < tiles_m;
}

template <typename TC, int EPI>
__global__ __launch_bounds__(512, 2) void k_gemm_nt256(NtArgs g, int tiles_m, int tiles_n, int nvirt) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;

  // ---- LDS-DMA: 32 wave-instructions per operand tile (8 rows x 128 B each), wave w issues instructions w, w + 8, w + 16, w + 24
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)(((size_t)(g.M - 1) * g.lda + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)(((size_t)(g.N - 1) * g.ldb + g.K) * 2), 0x00020000);
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;         // slot (row, c) <- k-chunk c ^ (row & 7)
  const unsigned laneA = (unsigned)(((w * 8 + lrow) * g.lda + lchunk * 8) * 2), laneB = (unsigned)(((w * 8 + lrow) * g.ldb + lchunk * 8) * 2);
  const unsigned stepA = (unsigned)(64 * g.lda * 2), stepB = (unsigned)(64 * g.ldb * 2);
  // one of the 8 DMA instructions of a stage (t = 0..3: A rows 64 t .., 4..7: B rows): tile origin (voA, voB), k-step kt, stage st
  // (voA / voB are wave-uniform byte offsets of the tile's first row: kept in SGPRs and added to the lane's offset per instruction --
  //  the bounds check of a raw buffer covers the VGPR offset only, so the whole offset goes there)
  auto issue1 = [&](int t, unsigned voA, unsigned voB, int kt, int st) {
    char* base = smem + st * G2_STAGE + w * 1024;
    const unsigned kb = (unsigned)(kt * G2_BK * 2);
    if (t < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + t * 8192), 16, laneA + (voA + kb + t * stepA), 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(base + G2_OPB + (t - 4) * 8192), 16, laneB + (voB + kb + (t - 4) * stepB), 0, 0, 0);
  };
  // ---- fragment addresses: row = lane & 15 (+ the tile's offset), k-chunk 4 ks + lane / 16, swizzled with row & 7 = lane & 7
  const unsigned s0 = lds_addr(smem);
  const unsigned sw0 = (unsigned)((((lane >> 4)) ^ (lane & 7)) * 16);
  const unsigned fa0 = s0 + (unsigned)((wm * 128 + (lane & 15)) * 128) + sw0;                    // k-half 0; k-half 1 = address ^ 64
  const unsigned fb0 = s0 + (unsigned)(G2_OPB + (wn * 64 + (lane & 15)) * 128) + sw0;
  auto rdfrags = [&](G2Frags& F, int st, int ks) {
    const unsigned a = (fa0 ^ (unsigned)(ks * 64)) + (unsigned)(st * G2_STAGE), b = (fb0 ^ (unsigned)(ks * 64)) + (unsigned)(st * G2_STAGE);
    F.b[0] = ds_read128<0 * 2048>(b); F.b[1] = ds_read128<1 * 2048>(b); F.b[2] = ds_read128<2 * 2048>(b); F.b[3] = ds_read128<3 * 2048>(b);
    F.a[0] = ds_read128<0 * 2048>(a); F.a[1] = ds_read128<1 * 2048>(a); F.a[2] = ds_read128<2 * 2048>(a); F.a[3] = ds_read128<3 * 2048>(a);
    F.a[4] = ds_read128<4 * 2048>(a); F.a[5] = ds_read128<5 * 2048>(a); F.a[6] = ds_read128<6 * 2048>(a); F.a[7] = ds_read128<7 * 2048>(a);
  };
  auto tie = [&](G2Frags& F) {
    asm volatile("" : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.a[2]), "+v"(F.a[3]), "+v"(F.a[4]), "+v"(F.a[5]), "+v"(F.a[6]), "+v"(F.a[7]),
                      "+v"(F.b[0]), "+v"(F.b[1]), "+v"(F.b[2]), "+v"(F.b[3]));
  };
  f32x4 acc[8][4];
  auto mfma_row = [&](const G2Frags& F, int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(__builtin_bit_cast(typename MM::Frag, F.b[j]), __builtin_bit_cast(typename MM::Frag, F.a[i]), acc[i][j]);
  };
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (EPI == UVC_EPI_BIAS_RESID_GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  float* const stg = reinterpret_cast<float*>(smem + 2 * G2_STAGE) + w * (16 * 64);

  // ---- persistent walk over this workgroup's tiles: the k-steps of consecutive tiles form ONE stream through the two stages -- the
  //      first two k-steps of the NEXT tile are requested in the last two iterations of the current one, so a tile's operand latency
  //      and the previous tile's epilogue overlap
  const int nk = g.K / G2_BK;                     // >= 4 (the dispatch requires K >= 256)
  int v = blockIdx.x, bm = 0, bn = 0;
  while (v < nvirt && !g2_tile(v, tiles_m, tiles_n, bm, bn)) v += gridDim.x;
  if (v >= nvirt) return;
  unsigned voA = (unsigned)(bm * G2_BM * g.lda * 2), voB = (unsigned)(bn * G2_BN * g.ldb * 2);
  int par = 0;                                    // stage of the current tile's k-step 0
#pragma unroll
  for (int t = 0; t < 8; ++t) issue1(t, voA, voB, 0, 0);
#pragma unroll
  for (int t = 0; t < 8; ++t) issue1(t, voA, voB, 1, 1);
  wait_vm<8>();                                   // own part of the first stage has landed (loads retire in order; 8 per stage and wave)
  __builtin_amdgcn_s_barrier();
  G2Frags F0, F1;
  rdfrags(F0, 0, 0);
  wait_lgkm<0>();
  tie(F0);
  for (;;) {
    // the tile after this one (its first k-steps are requested while this one finishes)
    int vn = v + gridDim.x, bmn = 0, bnn = 0;
    while (vn < nvirt && !g2_tile(vn, tiles_m, tiles_n, bmn, bnn)) vn += gridDim.x;
    const bool more = vn < nvirt;
    const unsigned voAn = more ? (unsigned)(bmn * G2_BM * g.lda * 2) : voA, voBn = more ? (unsigned)(bnn * G2_BN * g.ldb * 2) : voB;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      const int st = (kt & 1) ^ par;
      // The order of the step is pinned (scheduling fences): left to itself hipcc sinks two thirds of the first MFMA block below the
      // barrier, which exposes the fragment reads' latency in front of it and leaves the matrix pipe idle around it (s_memtime trace:
      // 2950 cycles per step for 1024 cycles of MFMA issue per SIMD).  A wave never waits with nothing queued behind it: the reads of a
      // block are requested a block ahead, the barrier sits between two MFMA blocks, and the first instructions behind it are MFMAs.
      rdfrags(F1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) mfma_row(F0, i);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm<0>();                             // F1 is in registers (requested 32 MFMAs ago): this wave is done with stage st
      tie(F1);
      wait_vm<0>();                               // own part of the next k-step's stage has landed (requested a step ago)
      __builtin_amdgcn_s_barrier();               // everybody's has; nobody reads stage st any more
      __builtin_amdgcn_sched_barrier(0);
      // the stage freed now receives k-step kt + 2 of this tile, or k-step kt + 2 - nk of the next one (all scalar selects;
      // behind the workgroup's last tile the same instructions re-load k-steps 0 / 1 of that tile: no branch around a DMA)
      const bool tail = kt + 2 >= nk;
      const unsigned sA = tail ? voAn : voA, sB = tail ? voBn : voB;
      const int kk = tail ? kt + 2 - nk : kt + 2;
      mfma_row(F1, 0);
      __builtin_amdgcn_sched_barrier(0);
      rdfrags(F0, st ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the 8 DMA instructions of that stage go BETWEEN the MFMA rows: their issue cost (~100 cycles each) hides under the matrix pipe
#pragma unroll
      for (int i = 1; i < 8; ++i) {
        issue1(i - 1, sA, sB, kk, st);
        if (i == 7) issue1(7, sA, sB, kk, st);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(F1, i);
        __builtin_amdgcn_sched_barrier(0);
      }
      wait_lgkm<0>();
      tie(F0);
    }
    // ---- epilogue: 16 rows x 64 columns at a time through this wave's private transpose buffer (behind the stages: the next tile's
    //      operands keep arriving meanwhile).  Measured against an epilogue straight from the accumulator layout (v_permlane16_swap
    //      pairs, 16-byte stores in 64-byte row pieces, operands read in 32-byte pieces): equal with no operands, 10-15 % of the GEMM
    //      slower with a bias / residual / multiplier to read -- whole 128-byte row segments matter more than the LDS round trips
    {
      const int m0 = bm * G2_BM, n0 = bn * G2_BN;
      float bias_v[OutVec<TC>::VN];
      nt_load_bias<TC, EPI>(g, lane, n0 + wn * 64, bias_v);
      NT_EPILOGUE_TILE(8, m0 + wm * 128, n0 + wn * 64, 2, 1)      // (256 registers: two fragment sets are live across the epilogue)
    }
    if (!more) break;
    v = vn; bm = bmn; bn = bnn; voA = voAn; voB = voBn;
    par ^= nk & 1;
  }
  wait_vm<0>();                                   // the two redundant stages behind the last tile
}

// ================================================================================================
// D = 384 (DeiT-Small, T2T-ViT-14): the dgrad of fc1 / qkv with the LayerNorm BACKWARD as its epilogue, on a ROW TILE.
//     dx = LN'(A . W^T; x, mean, rstd, gamma) + a1 * add1 + a2 * add2        (uvc_gemm_nt_lnbwd; model_distilled.py:199-204,218-247)
// The weights (384 x 1152 / 1536) do not fit the register files the D = 192 kernels keep them in, so this is k_gemm_nt256's main loop
// on a 128 x 384 tile -- whole rows of dx per workgroup: 8 waves as 2 x 4, a wave accumulates 64 x 96 (4 A- and 6 B-fragments per
// k-half, 96 accumulator registers), the same two 64-KB stages (16 KB of A, 48 KB of W per k-step: 2 + 6 DMA instructions per wave)
// and the same pinned step.  Behind the last k-step the tile is rounded to bf16 -- exactly what the unfused pair stores between the
// GEMM and the LayerNorm pass -- into the (now free) stages as 128 rows of 776 bytes, and the eight waves run k_ln_bwd_v's row
// arithmetic over it with that kernel's lane mapping (32 lanes x 12 columns, two rows at a time, three row sets of x / add1 / add2 in
// flight): dx is BIT-IDENTICAL to k_gemm_nt + k_ln_bwd_v (tests/test_kernels_gpu.py); dgamma / dbeta / dots leave as one partial row
// per workgroup for the batched LayerNorm reduce (256 rows: workgroups without a tile write zeros).
constexpr int R3_BM = 128, R3_BN = 384, R3_BK = 64;
constexpr int R3_OPA = R3_BM * 128, R3_OPB = R3_BN * 128;   // operand images of a k-step: 16 KB + 48 KB
constexpr int R3_STAGE = R3_OPA + R3_OPB;
constexpr int R3_RS = R3_BN * 2 + 8;                        // row stride of the bf16 result tile: 16 rows = 16 different 8-byte bank slots
constexpr int R3_LDS = 2 * R3_STAGE + 8 * (2 * R3_BN + 2) * 4;   // stages (the result tile lives in them) + the waves' partial rows
static_assert(R3_BM * R3_RS <= 2 * R3_STAGE, "the result tile fits the stages");

struct R3Frags { u32x4 a[4]; u32x4 b[6]; };
// MODE 1 (r4): the FORWARD counterpart on the same main loop -- fc2 / attn.proj (K -> 384) + bias + residual (+ gate mix) with the LayerNorm of the
// output rows as a second output (uvc_gemm_nt with ln_out at N = 384: the next block's norm1 / this block's norm2, two stand-alone passes per block
// before).  Behind the k-loop the float32 tile goes to LDS one 64-row half at a time (rows of 1568 bytes), and the eight waves run the epilogue of
// nt_epilogue_rows and then k_ln_fwd_v<bf16, bf16, 6>'s arithmetic over whole rows with that kernel's lane map (16 lanes x 24 columns, four rows
// at a time): C, the LayerNorm rows and the statistics are BIT-IDENTICAL to k_gemm_nt + k_ln_fwd_v, so whether a batch takes the fused form does
// not show in its results (tests/test_kernels_gpu.py; the engine's batch-independence tests).
struct R3Fwd {
  const float* bias; const void* R; const void* R2; const float* dptr; void* C;
  const float* ln_gamma; const float* ln_beta; void* ln_out; float* ln_mean; float* ln_rstd; float ln_eps; int gate;
};
constexpr int R3_FS = R3_BN * 4 + 32;                        // float32 half-tile row: 392 words = 8 mod 64 (conflict-free 16-byte writes by (row, 4-column group))
static_assert(64 * R3_FS <= 2 * R3_STAGE, "a float32 half tile fits the stages");

template <bool XLOW, int MODE = 0>
__global__ __launch_bounds__(512, 2) void k_gemm_row384_lnbwd(LnbArgs g, int tiles_m, R3Fwd f) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int D = R3_BN, NV4 = 3, LPR = 32;
  typedef typename std::conditional<XLOW, bf16_t, float>::type TX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int sub = lane & (LPR - 1), rg = lane / LPR;

  // ---- main loop: k_gemm_nt8p's schedule (two wave groups half a phase apart, the fragments of a phase requested inside the MFMA block of the
  //      phase before, four LDS slots per k-step read in phases 0, 0, 1, 2) on the 128 x 384 tile: a wave's 64 x 96 block is the quadrants
  //      A0 / A1 = row blocks 0-1 / 2-3, B0 / B1 = column blocks 0-2 / 3-5 (12 MFMAs a phase).  Slots of a k-step: A'0 (64 rows: the A0 rows of
  //      both M groups, 8 KB), B'0 (192 rows of W: the B0 columns of the four N groups, 24 KB), B'1, A'1 = 64 KB, two buffers (the result tile
  //      and the row pass reuse them behind the loop).  A wave issues 1 / 3 / 3 / 1 DMA instructions for them; waits counted accordingly.
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)(((size_t)(g.M - 1) * g.K + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.W), 0, (int)((size_t)R3_BN * g.K * 2), 0x00020000);
  constexpr int SZA = 64 * 128, SZB = 192 * 128, OFF_A0 = 0, OFF_B0 = SZA, OFF_B1 = SZA + SZB, OFF_A1 = SZA + 2 * SZB;
  static_assert(2 * SZA + 2 * SZB == R3_STAGE, "a k-step's four slots fill a stage");
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;         // slot (row, c) <- k-chunk c ^ (row & 7)
  unsigned laneA[2], laneB[2][3];
  {
    const int ra = w * 8 + lrow;                                  // image row of an A slot: M group ra >> 5, row ra & 31 of its 32
#pragma unroll
    for (int h = 0; h < 2; ++h) laneA[h] = (unsigned)((((ra >> 5) * 64 + h * 32 + (ra & 31)) * g.K + lchunk * 8) * 2);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int rb = (w + 8 * q) * 8 + lrow;                      // image row of a W slot: N group rb / 48, column rb % 48 of its 48
#pragma unroll
      for (int h = 0; h < 2; ++h) laneB[h][q] = (unsigned)((((rb / 48) * 96 + h * 48 + rb % 48) * g.K + lchunk * 8) * 2);
    }
  }
  // request item IT (0 = A'0, 1 = B'0, 2 = B'1, 3 = A'1) of the k-step at byte offset kb of the rows, into buffer buf
  auto issue_item = [&](int it, unsigned voA, unsigned kb, int buf) {
    char* base = smem + buf * R3_STAGE + w * 1024;
    if (it == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + OFF_A0), 16, laneA[0] + (voA + kb), 0, 0, 0);
    else if (it == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + OFF_A1), 16, laneA[1] + (voA + kb), 0, 0, 0);
    else {
      const int h = it - 1;
#pragma unroll
      for (int q = 0; q < 3; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(base + (h ? OFF_B1 : OFF_B0) + q * 8192), 16, laneB[h][q] + kb, 0, 0, 0);
    }
  };
  const unsigned s0 = lds_addr(smem);
  const unsigned sw0 = (unsigned)((((lane >> 4)) ^ (lane & 7)) * 16);
  const unsigned fa0 = s0 + (unsigned)((wm * 32 + (lane & 15)) * 128) + sw0;       // inside an A slot; k-half 1 = address ^ 64
  const unsigned fb0 = s0 + (unsigned)((wn * 48 + (lane & 15)) * 128) + sw0;       // inside a W slot
  u32x4 FA[2][2], FB0[3][2], FB1[3][2];
  f32x4 acc[4][6];
#define R8_RD_B(DST, OFF, BO)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                                      \
    DST[j][0] = ds_read128_asm(fb0 + (BO), (OFF) + j * 2048); DST[j][1] = ds_read128_asm((fb0 ^ 64u) + (BO), (OFF) + j * 2048); }
#define R8_RD_A(I, OFF, BO) { FA[I][0] = ds_read128_asm(fa0 + (BO), (OFF) + (I) * 2048); FA[I][1] = ds_read128_asm((fa0 ^ 64u) + (BO), (OFF) + (I) * 2048); }
#define R8_MMA6(I, AI, FB, JO)                                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
  _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                        \
    acc[AI][(JO) + j] = MM::mma(__builtin_bit_cast(typename MM::Frag, FB[j][ks]), __builtin_bit_cast(typename MM::Frag, FA[I][ks]), acc[AI][(JO) + j]);
#define R8_REQ(IT, KB, SBUF, NWAIT)                                                                                   \
  issue_item(IT, voA, KB, SBUF);                                                                                       \
  if ((NWAIT) >= 0) wait_vm<((NWAIT) < 0 ? 0 : (NWAIT))>();                                                            \
  __builtin_amdgcn_s_barrier();                                                                                        \
  wait_lgkm<0>();                                                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_setprio(1);
#define R8_END                                                                                                        \
  __builtin_amdgcn_s_setprio(0);                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                        \
  __builtin_amdgcn_sched_barrier(0);

  // ---- LayerNorm backward state of this lane: columns (sub + 32 i) * 4 .. + 3, i = 0..2 (k_ln_bwd_v<.., 3, 32>'s map)
  f32x4 gam[NV4], dgam[NV4], dbet[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    gam[i] = MODE == 0 ? *reinterpret_cast<const f32x4*>(g.gamma + (sub + LPR * i) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    dgam[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dbet[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float a1 = g.a1 ? *g.a1 : 1.f, a2 = g.a2 ? *g.a2 : 1.f;
  float dotA = 0.f, dotB = 0.f;
  const float invD = 1.0f / (float)D;
  const bf16_t* add1 = reinterpret_cast<const bf16_t*>(g.add1);
  const bf16_t* add2 = reinterpret_cast<const bf16_t*>(g.add2);
  struct RawRow { typename Raw4L<TX>::V xv[NV4]; u32x2 ad1[NV4], ad2[NV4]; float mean, rstd; int r; bool ok; };

  const int nk = g.K / R3_BK;
  for (int bm = blockIdx.x; bm < tiles_m; bm += gridDim.x) {
    const unsigned voA = (unsigned)(bm * R3_BM * g.K * 2);
    __syncthreads();                                              // the last tile's row pass is done with the stages
    // prologue: k-step 0 whole, A'0 and B'0 of k-step 1
#pragma unroll
    for (int it = 0; it < 4; ++it) issue_item(it, voA, 0u, 0);
    issue_item(0, voA, 128u, 1); issue_item(1, voA, 128u, 1);
    wait_vm<5>();                                                 // A'0, B'0, B'1 of k-step 0 have landed (A'1: 1, A'0 + B'0 of k-step 1: 4 in flight)
    __builtin_amdgcn_s_barrier();
    R8_RD_B(FB0, OFF_B0, 0u)
    R8_RD_A(0, OFF_A0, 0u) R8_RD_A(1, OFF_A0, 0u)
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs half a phase behind through the k-loop
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const unsigned bo = (unsigned)(buf * R3_STAGE), bon = bo ^ (unsigned)R3_STAGE;
      // k-steps kt + 1, kt + 2 (behind the last ones: k-steps 0 / 1 of the same tile again -- no branch around a DMA; nobody multiplies them)
      const unsigned kb1 = (unsigned)((kt + 1 < nk ? kt + 1 : kt + 1 - nk) * 128), kb2 = (unsigned)((kt + 2 < nk ? kt + 2 : kt + 2 - nk) * 128);
      // phase 0: (A0, B0); requests B'1 of kt + 1 (3); reads B1.  In flight behind the wait: A'0 + B'0 + B'1 of kt + 1 = 7
      R8_REQ(2, kb1, buf ^ 1, 7)
      R8_RD_B(FB1, OFF_B1, bo)
      __builtin_amdgcn_sched_barrier(0);
      R8_MMA6(0, 0, FB0, 0) R8_MMA6(1, 1, FB0, 0)
      R8_END
      // phase 1: (A0, B1); requests A'1 of kt + 1 (1); reads A1 behind the MFMAs that free A0's registers
      R8_REQ(3, kb1, buf ^ 1, -1)
      R8_MMA6(0, 0, FB1, 3)
      __builtin_amdgcn_sched_barrier(0);
      R8_RD_A(0, OFF_A1, bo)
      __builtin_amdgcn_sched_barrier(0);
      R8_MMA6(1, 1, FB1, 3)
      __builtin_amdgcn_sched_barrier(0);
      R8_RD_A(1, OFF_A1, bo)
      R8_END
      // phase 2: (A1, B1); requests A'0 of kt + 2 (1).  In flight: B'1 + A'1 of kt + 1, A'0 of kt + 2 = 5
      R8_REQ(0, kb2, buf, 5)
      R8_MMA6(0, 2, FB1, 3) R8_MMA6(1, 3, FB1, 3)
      R8_END
      // phase 3: (A1, B0); requests B'0 of kt + 2 (3); reads A0 and B0 of kt + 1.  In flight: A'1 of kt + 1, A'0 + B'0 of kt + 2 = 5
      R8_REQ(1, kb2, buf, 5)
      R8_MMA6(0, 2, FB0, 0)
      __builtin_amdgcn_sched_barrier(0);
      R8_RD_A(0, OFF_A0, bon)
      __builtin_amdgcn_sched_barrier(0);
      R8_MMA6(1, 3, FB0, 0)
      __builtin_amdgcn_sched_barrier(0);
      R8_RD_A(1, OFF_A0, bon)
      R8_RD_B(FB0, OFF_B0, bon)
      R8_END
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                    // pairs with group 1's last barrier: the groups are level again
    wait_lgkm<0>();
    wait_vm<0>();
    __syncthreads();                                              // every wave is done with the stages and nothing is in flight into them
    if constexpr (MODE == 1) {
      const int sub16 = lane & 15, rg4 = lane >> 4;
      const float invD384 = 1.0f / (float)D;
      const float d0 = f.gate ? f.dptr[0] : 0.f, d1 = f.gate ? f.dptr[1] : 1.f;
      const bf16_t* Rp = reinterpret_cast<const bf16_t*>(f.R);
      const bf16_t* R2p = reinterpret_cast<const bf16_t*>(f.R2);
      for (int half = 0; half < 2; ++half) {
        // rows w * 8 .. + 7 of the half: two sets of four rows (16 lanes a row); both sets' residual rows are requested here, in front of the
        // staging writes and the barrier, so that their HBM round trip runs beside them
        u32x2 rr[2][6], rr2[2][6];
        int rows_[2]; bool ok_[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          rows_[it] = bm * R3_BM + half * 64 + w * 8 + it * 4 + rg4;
          ok_[it] = rows_[it] < g.M;
          const size_t off = (size_t)(ok_[it] ? rows_[it] : 0) * D;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            rr[it][i] = ok_[it] ? *reinterpret_cast<const u32x2*>(Rp + off + (sub16 + 16 * i) * 4) : u32x2{0u, 0u};
            rr2[it][i] = (ok_[it] && f.gate) ? *reinterpret_cast<const u32x2*>(R2p + off + (sub16 + 16 * i) * 4) : u32x2{0u, 0u};
          }
        }
        if (wm == half) {                                          // this M group's 64 rows, float32, row-major
          const unsigned base = s0 + (unsigned)((lane & 15) * R3_FS + (wn * 96 + (lane >> 4) * 4) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j)
              asm volatile("ds_write_b128 %0, %1" ::"v"(base + (unsigned)(i * 16 * R3_FS + j * 64)), "v"(acc[i][j]) : "memory");
        }
        wait_lgkm<0>();
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const unsigned la = s0 + (unsigned)((w * 8 + it * 4 + rg4) * R3_FS + sub16 * 16);
          f32x4 v[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) v[i] = __builtin_bit_cast(f32x4, ds_read128_asm(la, i * 256));
          wait_lgkm<0>();
          asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
          float sm = 0.f;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(f.bias + (sub16 + 16 * i) * 4);
            const f32x4 rv = Raw4L<bf16_t>::cvt(rr[it][i]), r2v = Raw4L<bf16_t>::cvt(rr2[it][i]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = epi_scale_bias(v[i][e], 1.0f, bv[e]);
              t += rv[e];
              if (f.gate) t = epi_gate_mix(t, r2v[e], d0, d1);
              v[i][e] = t;
            }
            u32x2 q; q[0] = pack_bf16x2(v[i][0], v[i][1]); q[1] = pack_bf16x2(v[i][2], v[i][3]);
            if (ok_[it]) *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(f.C) + (size_t)rows_[it] * D + (sub16 + 16 * i) * 4) = q;
            v[i] = Raw4L<bf16_t>::cvt(q);                          // the LayerNorm is taken of the stored (rounded) row, as the stand-alone pass does
            sm += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
          }
          sm = xor_tree_sum<16>(sm);                   // (k_ln_fwd_v's sum16)
          const float mean = sm * invD384;
          float qq = 0.f;
#pragma unroll
          for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dd = v[i][e] - mean; qq += dd * dd; }
          qq = xor_tree_sum<16>(qq);
          const float rstd = rsqrtf(qq * invD384 + f.ln_eps);
          if (ok_[it]) {
            bf16_t* y = reinterpret_cast<bf16_t*>(f.ln_out) + (size_t)rows_[it] * D;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              const f32x4 gm = *reinterpret_cast<const f32x4*>(f.ln_gamma + (sub16 + 16 * i) * 4), bt = *reinterpret_cast<const f32x4*>(f.ln_beta + (sub16 + 16 * i) * 4);
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
              u32x2 q; q[0] = pack_bf16x2(o[0], o[1]); q[1] = pack_bf16x2(o[2], o[3]);
              *reinterpret_cast<u32x2*>(y + (sub16 + 16 * i) * 4) = q;
            }
            if (sub16 == 0 && f.ln_mean) { f.ln_mean[rows_[it]] = mean; f.ln_rstd[rows_[it]] = rstd; }
          }
        }
        __syncthreads();                                           // the half is consumed: the other half / the next tile's requests may overwrite it
      }
      continue;
    }
    // ---- the tile, rounded to bf16 (what k_gemm_nt stores), row-major in LDS: lane (li, gq) of acc[i][j] = row i * 16 + li, columns j * 16 + 4 gq ..
    {
      const unsigned base = s0 + (unsigned)((wm * 64 + (lane & 15)) * R3_RS + (wn * 96 + (lane >> 4) * 4) * 2);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          f32x2 q;
          q[0] = __uint_as_float(pack_bf16x2(acc[i][j][0], acc[i][j][1]));
          q[1] = __uint_as_float(pack_bf16x2(acc[i][j][2], acc[i][j][3]));
          ds_write64_a(base + (unsigned)(i * 16 * R3_RS + j * 32), q);
        }
    }
    wait_lgkm<0>();
    __syncthreads();
    // ---- row pass: wave w owns rows w * 16 .. + 15 of the tile, two at a time (k_ln_bwd_v's arithmetic on dy = the LDS row)
    {
      const int rbase = bm * R3_BM + w * 16;
      auto load_raw = [&](RawRow& R, int it) {
        R.r = rbase + it * 2 + rg;
        R.ok = it < 8 && R.r < g.M;
        const size_t off = (size_t)(R.ok ? R.r : 0) * D;
        const TX* x = reinterpret_cast<const TX*>(g.x) + off;
        R.mean = R.ok ? g.mean[R.r] : 0.f; R.rstd = R.ok ? g.rstd[R.r] : 0.f;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
          R.ad1[i] = (R.ok && add1) ? *reinterpret_cast<const u32x2*>(add1 + off + (sub + LPR * i) * 4) : u32x2{0u, 0u};
          R.ad2[i] = (R.ok && add2) ? *reinterpret_cast<const u32x2*>(add2 + off + (sub + LPR * i) * 4) : u32x2{0u, 0u};
          R.xv[i] = R.ok ? Raw4L<TX>::ld(x + (sub + LPR * i) * 4) : Raw4L<TX>::zero();
        }
      };
      auto process = [&](const RawRow& R, int it) {
        f32x4 xv[NV4], dv[NV4], ad1[NV4], ad2[NV4], gy[NV4];
        u32x2 dq[NV4];
        const unsigned la = s0 + (unsigned)((w * 16 + it * 2 + rg) * R3_RS + sub * 8);
#pragma unroll
        for (int i = 0; i < NV4; ++i) dq[i] = ds_read64_a(la + (unsigned)(i * LPR * 8));
        wait_lgkm<0>();
        asm volatile("" : "+v"(dq[0]), "+v"(dq[1]), "+v"(dq[2]));
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
          xv[i] = Raw4L<TX>::cvt(R.xv[i]); dv[i] = Raw4L<bf16_t>::cvt(R.ok ? dq[i] : u32x2{0u, 0u});
          ad1[i] = Raw4L<bf16_t>::cvt(R.ad1[i]); ad2[i] = Raw4L<bf16_t>::cvt(R.ad2[i]);
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (xv[i][e] - R.mean) * R.rstd;
            gy[i][e] = dv[i][e] * gam[i][e];
            dgam[i][e] += dv[i][e] * xh;
            dbet[i][e] += dv[i][e];
            c1 += gy[i][e];
            c2 += gy[i][e] * xh;
          }
        c1 = xor_tree_sum<LPR>(c1);                    // (k_ln_bwd_v's sum_lpr: the same additions)
        c2 = xor_tree_sum<LPR>(c2);
        c1 *= invD; c2 *= invD;
        if (R.ok) {
          bf16_t* dx = reinterpret_cast<bf16_t*>(g.dx) + (size_t)R.r * D;
#pragma unroll
          for (int i = 0; i < NV4; ++i) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = R.rstd * (gy[i][e] - c1 - ((xv[i][e] - R.mean) * R.rstd) * c2);
            if (add1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] += a1 * ad1[i][e]; }
            if (add2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { o[e] += a2 * ad2[i][e]; dotB += ad2[i][e] * xv[i][e]; } }
#pragma unroll
            for (int e = 0; e < 4; ++e) dotA += o[e] * xv[i][e];
            u32x2 r; r[0] = pack_bf16x2(o[0], o[1]); r[1] = pack_bf16x2(o[2], o[3]);
            *reinterpret_cast<u32x2*>(dx + (sub + LPR * i) * 4) = r;
          }
        }
      };
      RawRow Ra, Rb, Rc;
      load_raw(Ra, 0); load_raw(Rb, 1); load_raw(Rc, 2);
      for (int it = 0; it < 8; it += 3) {
        process(Ra, it); load_raw(Ra, it + 3);
        if (it + 1 < 8) { process(Rb, it + 1); load_raw(Rb, it + 4); }
        if (it + 2 < 8) { process(Rc, it + 2); load_raw(Rc, it + 5); }
      }
    }
  }
#undef R8_RD_B
#undef R8_RD_A
#undef R8_MMA6
#undef R8_REQ
#undef R8_END
  if constexpr (MODE == 1) return;
  // ---- this workgroup's partial row [2 D + 2]: the two row groups of a wave, then the eight waves in a fixed order
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem + 2 * R3_STAGE);
  constexpr int PW = 2 * D + 2;
#pragma unroll
  for (int i = 0; i < NV4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gq = dgam[i][e], bq = dbet[i][e];
      gq += __shfl_xor(gq, 32, 64); bq += __shfl_xor(bq, 32, 64);
      if (rg == 0) { red[w * PW + (sub + LPR * i) * 4 + e] = gq; red[w * PW + D + (sub + LPR * i) * 4 + e] = bq; }
    }
  dotA = wave_sum(dotA); dotB = wave_sum(dotB);
  if (lane == 0) { red[w * PW + 2 * D] = dotA; red[w * PW + 2 * D + 1] = dotB; }
  __syncthreads();
  float* P = g.partial + (size_t)blockIdx.x * PW;
  for (int c = tid; c < PW; c += 512) {
    float t = red[c];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += red[q * PW + c];
    P[c] = t;
  }
}

int launch_row384_lnbwd(const LnbArgs& a, int x_lowp, hipStream_t st) {
  const int tiles_m = ceil_div(a.M, R3_BM);
  const int grid = uvc_gemm_lnbwd_nblocks(a.M);           // 256 partial rows (M >= 4096): workgroups past the tiles write zeros
  R3Fwd f = {};
  if (x_lowp) { UVC_MAX_LDS(R3_LDS, k_gemm_row384_lnbwd<true>); k_gemm_row384_lnbwd<true><<<grid, 512, R3_LDS, st>>>(a, tiles_m, f); }
  else { UVC_MAX_LDS(R3_LDS, k_gemm_row384_lnbwd<false>); k_gemm_row384_lnbwd<false><<<grid, 512, R3_LDS, st>>>(a, tiles_m, f); }
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
// uvc_gemm_nt with ln_out at N = 384 (bf16 operands, residual rows and outputs; K a multiple of 64, >= 128)
bool row384_fwd_ok(const NtArgs& a, int epi) {
  return a.N == R3_BN && a.K % 64 == 0 && a.K >= 128 && a.lda == a.K && a.ldb == a.K && a.ldc == R3_BN && a.ldr == R3_BN &&
         (epi == UVC_EPI_BIAS_RESID || epi == UVC_EPI_BIAS_RESID_GATE) && (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C | (uintptr_t)a.R | (uintptr_t)a.ln_out) & 15) == 0 &&
         (size_t)(a.M + 128) * a.K * 2 < (1ull << 31);
}
int launch_row384_fwd(const NtArgs& a, int epi, hipStream_t st) {
  LnbArgs b = {};
  b.A = a.A; b.W = a.B; b.M = a.M; b.K = a.K;
  R3Fwd f;
  f.bias = a.bias; f.R = a.R; f.R2 = a.R2; f.dptr = a.dptr; f.C = a.C; f.ln_gamma = a.ln_gamma; f.ln_beta = a.ln_beta; f.ln_out = a.ln_out;
  f.ln_mean = a.ln_mean; f.ln_rstd = a.ln_rstd; f.ln_eps = a.ln_eps; f.gate = epi == UVC_EPI_BIAS_RESID_GATE ? 1 : 0;
  const int tiles_m = ceil_div(a.M, R3_BM);
  // (r5, measured and not kept: tiles in whole rounds of the grid -- 512 frames of 98-99 rows instead of 394 of 128 at DeiT-Small batch 256, the frame rows past
  //  the tile requested out of the descriptor's range -- DeiT-Small 15.70 -> 16.57 ms, T2T-ViT-14 12.50 -> 12.78: the 30 % more k-loops cost more than the
  //  shorter row passes return, profiles/r5o; fewer workgroups than CUs for this launch: +- 0, profiles/r5k)
  const int grid = tiles_m < 256 ? tiles_m : 256;
  UVC_MAX_LDS(R3_LDS, k_gemm_row384_lnbwd<true, 1>);
  k_gemm_row384_lnbwd<true, 1><<<grid, 512, R3_LDS, st>>>(b, tiles_m, f);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

// the shapes k_gemm_nt256 takes: bf16 operands with 64-deep k tiles and 16-byte aligned rows, enough rows and columns to fill tiles
// ... and enough tiles: one persistent workgroup per CU means ceil(tiles / 256) rounds, and with fewer than ~3 the last, partly
// filled round costs more than the kernel gains (N = 768 at 25 k rows: 297 tiles = 2 rounds at 58 %; the 128 x 128 kernel with three
// workgroups per CU is as fast or faster there, measured).  Operand sizes < 2 GB: 32-bit buffer offsets.
static bool nt256_ok(const NtArgs& a, bool a_f32, bool any_size) {
  const int tiles = ceil_div(a.M, G2_BM) * ceil_div(a.N, G2_BN);
  return !a_f32 && (any_size || (tiles >= 640 && a.M >= 2048)) && a.K % 64 == 0 && a.K >= 256 && a.N >= 256 && a.lda % 8 == 0 && a.ldb % 8 == 0 &&
         (((uintptr_t)a.A | (uintptr_t)a.B) & 15) == 0 && (size_t)a.M * a.lda * 2 < (1ull << 31) && (size_t)a.N * a.ldb * 2 < (1ull << 31);
}
template <typename TC>
static int launch_nt256(const NtArgs& a, int epi, hipStream_t st) {
  const int tm = ceil_div(a.M, G2_BM), tn = ceil_div(a.N, G2_BN);
  const int nvirt = ceil_div(tm, 8) * 8 * tn;               // tile numbers incl. the holes of the XCD-major order (M tiles padded to the 8 XCDs)
  const int grid = nvirt < 256 ? nvirt : 256;               // one persistent workgroup per CU
#define G2_CASE(E) case E: UVC_MAX_LDS(G2_LDS, k_gemm_nt256<TC, E>); k_gemm_nt256<TC, E><<<grid, 512, G2_LDS, st>>>(a, tm, tn, nvirt); break;
  switch (epi) {
    G2_CASE(UVC_EPI_NONE) G2_CASE(UVC_EPI_BIAS) G2_CASE(UVC_EPI_BIAS_GELU) G2_CASE(UVC_EPI_BIAS_RESID)
    G2_CASE(UVC_EPI_BIAS_RESID_GATE) G2_CASE(UVC_EPI_DGELU) G2_CASE(UVC_EPI_BIAS_GELU_OUT) G2_CASE(UVC_EPI_BIAS_GELU_GRAD) G2_CASE(UVC_EPI_MUL_AUX)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: unknown epilogue");
  }
#undef G2_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
int launch_nt256_f32(const NtArgs& a, int epi, hipStream_t st) { return launch_nt256<float>(a, epi, st); }
int launch_nt256_bf16(const NtArgs& a, int epi, hipStream_t st) { return launch_nt256<bf16_t>(a, epi, st); }
bool nt256_takes(const NtArgs& a, bool a_f32, bool any_size) { return nt256_ok(a, a_f32, any_size); }

// ================================================================================================
//        NT on (32 RI) x 256 tiles with two wave groups half a phase apart ("8-phase" schedule)
// ================================================================================================
// k_gemm_nt256 keeps its eight waves in lock step: everybody reads fragments, everybody issues MFMAs, everybody waits at the one
// barrier of a k-step -- the matrix pipe idles ~27 % of every step (NOTEBOOK.md 5f).  Here the workgroup's two M groups (waves 0-3 /
// 4-7: one wave of each on every SIMD) run HALF A PHASE apart: a k-step of 64 is four phases, a phase is
//     [fragment reads + LDS-DMA requests] s_barrier [16 MFMAs] s_barrier
// and group 1 passes one extra barrier up front, so that while one wave of a SIMD issues its MFMA block the other one is in its
// read / request segment (cdna_hip_programming.md 5, "the 256^2 8-phase template"; re-derived here, the example source is not in the
// image).  A wave's 16 RI x 64 output is four quadrants: rows A0 = blocks 0 .. R0-1 / A1 = the rest, columns B0 = 0-31 / B1 = 32-63;
// phases: (A0,B0) (A0,B1) (A1,B1) (A1,B0), so a phase reads at most one new A sub-tile and one new B sub-tile: 12 / 4 / 8 / 0 reads, and
// the fragment registers are 32 + 16 + 16 instead of 96.
//   * LDS: two buffers of four 16-KB slots; a slot is the image of one sub-tile OF ALL EIGHT WAVES: A'0 = rows A0 of both M groups,
//     B'0 = columns B0 of the four N groups, B'1, A'1 likewise -- so that the slots of a k-step are read in DIFFERENT phases (0, 0, 1, 2)
//     and each can be re-filled as soon as its phase is two phases back: per phase ONE slot of a later k-step is requested (two
//     buffer_load ... lds per wave), four of them are in flight at every wait (vmcnt(8), never 0 inside a tile).
//     Order of requests: phase p of k-step X asks for  p=0: B'1(X+1)  p=1: A'1(X+1)  p=2: A'0(X+2)  p=3: B'0(X+2).
//     RAW: a slot is read one phase after the wait that covers it (wait before the phase's first barrier, read in the next phase: the
//     staggered group's waits are half a phase later).  WAR: a slot is re-filled two or more phases after its last read (the other
//     group's reads of phase g are complete only behind the second barrier of phase g).
//   * image of a slot: 128 rows x 128 B, lane-linear for the DMA, 16-byte slot (row, c) <- k-chunk c ^ (row & 7) (swizzle in the source
//     address, the same involution in the fragment address): conflict-free ds_read_b128 (k_gemm_nt256's).  RI < 8: the rows a group does
//     not own are requested out of bounds (zeros, no memory traffic), so every wave issues the same two instructions per slot and the
//     counted waits hold.
//   * persistent, XCD-major tile order, the k-steps of consecutive tiles form one stream; epilogue = nt_epilogue_rows through the
//     wave-private transpose buffer: the same bits as k_gemm_nt / k_gemm_nt256.  The epilogue drains the DMA queue once (vmcnt(0)),
//     and the first k-step behind it skips the waits of phases 0 / 1 (everything they cover was drained; a counted wait there would
//     wait for the epilogue's stores).
constexpr int P8_SLOT = 128 * 128;
constexpr int P8_BUF = 4 * P8_SLOT;
constexpr int P8_LDS = 2 * P8_BUF + G2_EPI;    // 160 KB

template <typename TC, int EPI, int RI>
__global__ __launch_bounds__(512, 2) void k_gemm_nt8p(NtArgs g, int tiles_m, int tiles_n, int nvirt) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int R0 = (RI + 1) / 2, R1 = RI - R0, BM = 32 * RI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)(((size_t)(g.M - 1) * g.lda + g.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)(((size_t)(g.N - 1) * g.ldb + g.K) * 2), 0x00020000);
  // ---- LDS-DMA: a slot is 16 wave-instructions (8 rows x 128 B each); wave w issues instructions w (image rows 8w ..: M group 0 / N
  //      groups 0-1) and w + 8 (image rows 64 + 8w ..: M group 1 / N groups 2-3)
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ lrow;
  constexpr unsigned OOB = 0x80000000u;            // beyond every buffer's range (< 2 GB): reads as zeros
  const int rr = w * 8 + lrow;                     // row inside a group's 64-row share of an A slot
  const unsigned laneA0 = rr < R0 * 16 ? (unsigned)((rr * g.lda + lchunk * 8) * 2) : OOB;
  const unsigned laneA1 = rr < R1 * 16 ? (unsigned)(((R0 * 16 + rr) * g.lda + lchunk * 8) * 2) : OOB;
  const unsigned grpA = (unsigned)(16 * RI * g.lda * 2);                                      // M group 1's rows
  const unsigned laneB = (unsigned)((((w >> 2) * 64 + (w & 3) * 8 + lrow) * g.ldb + lchunk * 8) * 2);
  const unsigned halfB = (unsigned)(32 * g.ldb * 2), grpB = (unsigned)(128 * g.ldb * 2);
  // request q (0..7) of a k-step: slot q >> 1 (A'0, B'0, B'1, A'1), instruction q & 1; src = byte offset of the tile's first row + k
  auto issue = [&](int q, unsigned srcA, unsigned srcB, int buf) {
    char* dst = smem + buf * P8_BUF + (q >> 1) * P8_SLOT + (q & 1) * 8192 + w * 1024;
    const int slot = q >> 1;
    if (slot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)dst, 16, laneA0 + (srcA + (q & 1) * grpA), 0, 0, 0);
    else if (slot == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)dst, 16, laneA1 + (srcA + (q & 1) * grpA), 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)dst, 16, laneB + (srcB + (slot - 1) * halfB + (q & 1) * grpB), 0, 0, 0);
  };
  // ---- fragment addresses (buffer 0, k-half 0; k-half 1 = address ^ 64)
  const unsigned s0 = lds_addr(smem);
  const unsigned sw0 = (unsigned)((((lane >> 4)) ^ (lane & 7)) * 16);
  const unsigned fa0 = s0 + (unsigned)((wm * 64 + (lane & 15)) * 128) + sw0;
  const unsigned fb0 = s0 + (unsigned)((wn * 32 + (lane & 15)) * 128) + sw0;
  u32x4 FA[4][2], FB0[2][2], FB1[2][2];            // [row / column block][k half]
  f32x4 acc[RI][4];
  float alpha = g.alpha;
  if (g.alpha_ptr) alpha *= *g.alpha_ptr;
  float d0 = 0.f, d1 = 1.f;
  if (EPI == UVC_EPI_BIAS_RESID_GATE) { d0 = g.dptr[0]; d1 = g.dptr[1]; }
  float* const stg = reinterpret_cast<float*>(smem + 2 * P8_BUF) + w * (16 * 64);

  const int nk = g.K / 64;                         // >= 4
  int v = blockIdx.x, bm = 0, bn = 0;
  while (v < nvirt && !g2_tile(v, tiles_m, tiles_n, bm, bn)) v += gridDim.x;
  if (v >= nvirt) return;
  unsigned voA = (unsigned)(bm * BM * g.lda * 2), voB = (unsigned)(bn * 256 * g.ldb * 2);
  int par = 0;                                     // buffer of the current tile's k-step 0
  // prologue: k-step 0 whole, A'0 and B'0 of k-step 1 (what the steady state has requested when a k-step begins)
#pragma unroll
  for (int q = 0; q < 8; ++q) issue(q, voA, voB, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue(q, voA + 128, voB + 128, 1);
  wait_vm<6>();                                    // A'0, B'0, B'1 of k-step 0 have landed (this wave's parts)
  __builtin_amdgcn_s_barrier();                    // ... everybody's
  // The fragments of a phase are requested INSIDE the MFMA block of the phase before (behind the MFMAs that last use their registers), so a
  // wave's segment between the two barriers of a phase is [wait for fragments requested a whole request segment ago | 16 MFMAs with the next
  // phase's reads between them], and the segment in front of the first barrier is the two DMA requests and their counted wait alone: that is
  // what runs beside the other group's MFMA block (an LDS-DMA request costs its wave 100-200 cycles of issue; with the 12 reads of phase 0 in
  // the same segment it was longer than the MFMA block and set the phase time: 707 cycles per phase for 272 of MFMA issue, measured by removing
  // one ingredient at a time).  Waits: data consumed by the MFMAs of phase c is read in phase c - 1 and waited for in phase c - 2 (the other
  // group's wait of phase c - 2 is half a phase later): vmcnt(6) in phases 2, 3, 0, three slots in flight.
#define P8_RD_B(DST, SLOT, BO)                                                                                       \
  { const unsigned b0_ = fb0 + (BO), b1_ = (fb0 ^ 64u) + (BO);                                                        \
    DST[0][0] = ds_read128<(SLOT) * P8_SLOT>(b0_); DST[0][1] = ds_read128<(SLOT) * P8_SLOT>(b1_);                      \
    DST[1][0] = ds_read128<(SLOT) * P8_SLOT + 2048>(b0_); DST[1][1] = ds_read128<(SLOT) * P8_SLOT + 2048>(b1_); }
#define P8_RD_A(I, SLOT, BO)                                                                                         \
  { FA[I][0] = ds_read128_asm(fa0 + (BO), (SLOT) * P8_SLOT + (I) * 2048); FA[I][1] = ds_read128_asm((fa0 ^ 64u) + (BO), (SLOT) * P8_SLOT + (I) * 2048); }
#define P8_MMA4(I, AI, FB, JO)                                                                                       \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                    \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                       \
    acc[AI][(JO) + j] = MM::mma(__builtin_bit_cast(typename MM::Frag, FB[j][ks]), __builtin_bit_cast(typename MM::Frag, FA[I][ks]), acc[AI][(JO) + j]);
  // request segment of a phase: two DMA instructions (q0, q0 + 1) of the k-step (sA, sB, sbuf), the counted wait, the first barrier
#define P8_REQ(Q0, SA, SB, SBUF, WAIT)                                                                               \
  issue(Q0, SA, SB, SBUF); issue(Q0 + 1, SA, SB, SBUF);                                                               \
  if (WAIT) wait_vm<6>();                                                                                             \
  __builtin_amdgcn_s_barrier();                                                                                       \
  wait_lgkm<0>();                                                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  __builtin_amdgcn_s_setprio(1);
#define P8_END                                                                                                       \
  __builtin_amdgcn_s_setprio(0);                                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                                  \
  __builtin_amdgcn_s_barrier();                                                                                       \
  __builtin_amdgcn_sched_barrier(0);
  P8_RD_B(FB0, 1, 0u)
#pragma unroll
  for (int i = 0; i < R0; ++i) P8_RD_A(i, 0, 0u)
  __builtin_amdgcn_sched_barrier(0);
  if (wm == 1) __builtin_amdgcn_s_barrier();       // group 1 runs half a phase behind from here on
  bool fresh = true;                               // phase 0's wait is needed (nothing drained the queue since the requests)

  for (;;) {
    int vn = v + gridDim.x, bmn = 0, bnn = 0;
    while (vn < nvirt && !g2_tile(vn, tiles_m, tiles_n, bmn, bnn)) vn += gridDim.x;
    const bool more = vn < nvirt;
    const unsigned voAn = more ? (unsigned)(bmn * BM * g.lda * 2) : voA, voBn = more ? (unsigned)(bnn * 256 * g.ldb * 2) : voB;
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = (kt & 1) ^ par;
      const unsigned bo = (unsigned)(buf * P8_BUF), bon = bo ^ (unsigned)P8_BUF;
      // k-steps kt + 1 and kt + 2 of the stream: of this tile, or k-steps 0 / 1 of the next one (behind the workgroup's last tile the
      // same instructions re-load this tile's: no branch around a DMA)
      const bool t1 = kt + 1 >= nk, t2 = kt + 2 >= nk;
      const unsigned sA1 = (t1 ? voAn : voA) + (unsigned)((t1 ? kt + 1 - nk : kt + 1) * 128), sB1 = (t1 ? voBn : voB) + (unsigned)((t1 ? kt + 1 - nk : kt + 1) * 128);
      const unsigned sA2 = (t2 ? voAn : voA) + (unsigned)((t2 ? kt + 2 - nk : kt + 2) * 128), sB2 = (t2 ? voBn : voB) + (unsigned)((t2 ? kt + 2 - nk : kt + 2) * 128);
      // ---- phase 0: (A0, B0); requests B'1 of k-step kt + 1; reads B1 of this k-step (its registers are free)
      P8_REQ(4, sA1, sB1, buf ^ 1, fresh)
      P8_RD_B(FB1, 2, bo)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < R0; ++i) P8_MMA4(i, i, FB0, 0)
      P8_END
      fresh = true;
      // ---- phase 1: (A0, B1); requests A'1 of kt + 1; reads A1 of this k-step, row block by row block behind the MFMAs that free A0's registers
      P8_REQ(6, sA1, sB1, buf ^ 1, false)
#pragma unroll
      for (int i = 0; i < R0; ++i) {
        P8_MMA4(i, i, FB1, 2)
        __builtin_amdgcn_sched_barrier(0);
        if (i < R1) P8_RD_A(i, 3, bo)
        __builtin_amdgcn_sched_barrier(0);
      }
      P8_END
      // ---- phase 2: (A1, B1); requests A'0 of kt + 2
      P8_REQ(0, sA2, sB2, buf, true)
#pragma unroll
      for (int i = 0; i < R1; ++i) P8_MMA4(i, R0 + i, FB1, 2)
      P8_END
      // ---- phase 3: (A1, B0); requests B'0 of kt + 2; reads A0 and B0 of k-step kt + 1 (the other buffer)
      P8_REQ(2, sA2, sB2, buf, true)
#pragma unroll
      for (int i = R1; i < R0; ++i) P8_RD_A(i, 0, bon)          // (registers phase 3 does not use)
#pragma unroll
      for (int i = 0; i < R1; ++i) {
        P8_MMA4(i, R0 + i, FB0, 0)
        __builtin_amdgcn_sched_barrier(0);
        P8_RD_A(i, 0, bon)
        __builtin_amdgcn_sched_barrier(0);
      }
      P8_RD_B(FB0, 1, bon)
      P8_END
    }
    // ---- epilogue (16 rows x 64 columns at a time through this wave's transpose buffer)
    wait_vm<0>();                                  // the requests of the next k-steps (in flight for 1-4 phases): drained once, here
    fresh = false;
    {
      const int m0 = bm * BM, n0 = bn * 256;
      float bias_v[OutVec<TC>::VN];
      nt_load_bias<TC, EPI>(g, lane, n0 + wn * 64, bias_v);
      NT_EPILOGUE_TILE(RI, m0 + wm * 16 * RI, n0 + wn * 64, 4, 2)
    }
    if (!more) break;
    v = vn; bm = bmn; bn = bnn; voA = voAn; voB = voBn;
    par ^= nk & 1;
  }
#undef P8_RD_B
#undef P8_RD_A
#undef P8_MMA4
#undef P8_REQ
#undef P8_END
  if (wm == 0) __builtin_amdgcn_s_barrier();       // pairs with group 1's last barrier
  wait_vm<0>();
}

static bool nt8p_ok(const NtArgs& a, bool a_f32) {
  return !a_f32 && a.K % 64 == 0 && a.K >= 256 && a.N >= 256 && a.lda % 8 == 0 && a.ldb % 8 == 0 &&
         (((uintptr_t)a.A | (uintptr_t)a.B) & 15) == 0 && (size_t)(a.M + 256) * a.lda * 2 < (1ull << 31) && (size_t)(a.N + 256) * a.ldb * 2 < (1ull << 31);
}
// rows per tile: the height (of 256 / 192 / 160 / 128) with the least work in the longest-running workgroup
static int nt8p_pick_ri(int M, int N) {
  const int tn = ceil_div(N, 256);
  int best = 8; long best_cost = -1;
  const int cand[4] = {8, 6, 5, 4};
  for (int ri : cand) {
    const int tm = ceil_div(M, 32 * ri);
    const long tiles = (long)tm * tn;
    const long rounds = (tiles + 255) / 256;
    const long cost = rounds * (ri + 8);           // tile time ~ a + b * rows with a / b = 8 row blocks (measured at M = 25 216, N = 768: 189 / 150 / 137 / 169 us for
                                                   // ri 8 / 6 / 5 / 4 = 2 / 2 / 2 / 3 rounds; the requests of the B slots and the latency of a k-step do not shrink with the rows)
    if (best_cost < 0 || cost < best_cost) { best = ri; best_cost = cost; }
  }
  return best;
}
template <typename TC, int RI>
static int launch_nt8p_ri(const NtArgs& a, int epi, hipStream_t st) {
  const int tm = ceil_div(a.M, 32 * RI), tn = ceil_div(a.N, 256);
  const int nvirt = ceil_div(tm, 8) * 8 * tn;
  const int grid = nvirt < 256 ? nvirt : 256;
#define P8_CASE(E) case E: UVC_MAX_LDS(P8_LDS, k_gemm_nt8p<TC, E, RI>); k_gemm_nt8p<TC, E, RI><<<grid, 512, P8_LDS, st>>>(a, tm, tn, nvirt); break;
  switch (epi) {
    P8_CASE(UVC_EPI_NONE) P8_CASE(UVC_EPI_BIAS) P8_CASE(UVC_EPI_BIAS_GELU) P8_CASE(UVC_EPI_BIAS_RESID)
    P8_CASE(UVC_EPI_BIAS_RESID_GATE) P8_CASE(UVC_EPI_DGELU) P8_CASE(UVC_EPI_BIAS_GELU_OUT) P8_CASE(UVC_EPI_BIAS_GELU_GRAD) P8_CASE(UVC_EPI_MUL_AUX)
    default: return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_nt: unknown epilogue");
  }
#undef P8_CASE
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
template <typename TC>
static int launch_nt8p(const NtArgs& a, int epi, int ri, hipStream_t st) {
  if (ri <= 0) ri = nt8p_pick_ri(a.M, a.N);
  if constexpr (sizeof(TC) == 4) return launch_nt8p_ri<TC, 8>(a, epi, st);     // float32 C (resid_f32 A/B mode, patch embedding): the 256-row tile only
  else switch (ri) {
    case 8: return launch_nt8p_ri<TC, 8>(a, epi, st);
    case 6: return launch_nt8p_ri<TC, 6>(a, epi, st);
    case 5: return launch_nt8p_ri<TC, 5>(a, epi, st);
    default: return launch_nt8p_ri<TC, 4>(a, epi, st);
  }
}
int launch_nt8p_f32(const NtArgs& a, int epi, int ri, hipStream_t st) { return launch_nt8p<float>(a, epi, ri, st); }
int launch_nt8p_bf16(const NtArgs& a, int epi, int ri, hipStream_t st) { return launch_nt8p<bf16_t>(a, epi, ri, st); }
bool nt8p_takes(const NtArgs& a, bool a_f32) { return nt8p_ok(a, a_f32); }

// ================================================================================================
//                                            TN (wgrad)
// ================================================================================================
// C[n1,n2] = beta*C + alpha * sum_m A[m,n1] * B[m,n2].  Block tile 128(n1) x 64(n2), reduction tile
// 64 rows of m.  The LDS image keeps the global [m][n] orientation; MFMA fragments are gathered with
// ds_read_b64_tr_b16 (bf16) or ds_read_b32 (float32).  Each block reduces one slice of M and writes
// a float32 partial tile; k_tn_reduce sums the slices in a fixed order (deterministic).
constexpr int TN_B1 = 128, TN_B2 = 64, TN_BM = 64;

template <typename T> struct TnGeom;
template <> struct TnGeom<bf16_t> { static constexpr int LD1 = TN_B1 * 2 + 32, LD2 = TN_B2 * 2 + 32; };
template <> struct TnGeom<float>  { static constexpr int LD1 = TN_B1 * 4 + 32, LD2 = TN_B2 * 4 + 32; };

// fragment of 16 columns (c0..c0+15) x one MFMA k-step starting at tile row k0, from an [m][n] LDS tile
template <typename T> struct TrFrag;
template <> struct TrFrag<bf16_t> {
  // k-slot map: group g slots 0-3 <-> rows k0+4g+{0..3}, slots 4-7 <-> rows k0+16+4g+{0..3}
  static __device__ __forceinline__ bf16x8 ld(const char* tile, int ld, int k0, int c0, int lane) {
    const int i = lane & 15, gq = lane >> 4;
    const char* p = tile + (k0 + 4 * gq + (i >> 2)) * ld + (c0 + (i & 3) * 4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(p + 16 * ld));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    return __builtin_bit_cast(bf16x8, v);
  }
};
template <> struct TrFrag<float> {
  // k-slot map: MFMA j of the step uses row k0 + 4g + j
  static __device__ __forceinline__ f32x4 ld(const char* tile, int ld, int k0, int c0, int lane) {
    const int i = lane & 15, gq = lane >> 4;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(tile + (k0 + 4 * gq + j) * ld + (c0 + i) * 4);
    return v;
  }
};

struct TnArgs {
  const void* A; const void* B; float* part; float* bpart;   // bpart: [splits][N1] column sums of A (bias gradient) or null
  int M, N1, N2, lda, ldb, rows_per_split, xcd_remap;
};

template <typename T> struct ChunkSum;    // add the CH elements of a staged chunk into float accumulators
template <> struct ChunkSum<bf16_t> {
  static __device__ __forceinline__ void add(const u32x4& r, float* s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] += __uint_as_float(r[e] << 16); s[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u); }
  }
};
template <> struct ChunkSum<float> {
  static __device__ __forceinline__ void add(const u32x4& r, float* s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] += __uint_as_float(r[e]);
  }
};

template <typename TA, typename T>
__global__ __launch_bounds__(256, 2) void k_gemm_tn(TnArgs g) {
  typedef Mma<T> MM;
  constexpr int CH = MM::CH;
  constexpr int LD1 = TnGeom<T>::LD1, LD2 = TnGeom<T>::LD2;
  constexpr int C1 = TN_B1 / CH, C2 = TN_B2 / CH;           // chunks per tile row
  constexpr int NLA = TN_BM * C1 / 256, NLB = TN_BM * C2 / 256;
  constexpr int KSUB = TN_BM / MM::KSTEP;                   // MFMA k-steps per reduction tile
  __shared__ __attribute__((aligned(16))) char sA[TN_BM * LD1];
  __shared__ __attribute__((aligned(16))) char sB[TN_BM * LD2];
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(g.B);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int w1 = w >> 1, w2 = w & 1;                        // wave tile: 64 (n1) x 32 (n2)
  const int n10 = blockIdx.x * TN_B1, n20 = blockIdx.y * TN_B2;
  const int mbeg = blockIdx.z * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);

  u32x4 ra[NLA], rb[NLB];
  auto gload = [&](int m0) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int id = tid + 256 * i, row = id / C1, c = id % C1;
      const int m = m0 + row, n = n10 + c * CH;
      ra[i] = (m < mend && n < g.N1) ? ChunkLoad<TA, T>::ld(A + (size_t)m * g.lda + n) : z;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int id = tid + 256 * i, row = id / C2, c = id % C2;
      const int m = m0 + row, n = n20 + c * CH;
      rb[i] = (m < mend && n < g.N2) ? ChunkLoad<T, T>::ld(B + (size_t)m * g.ldb + n) : z;
    }
  };
  float csum[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) csum[e] = 0.f;
  const bool do_cs = g.bpart != nullptr && blockIdx.y == 0;
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int id = tid + 256 * i, row = id / C1, c = id % C1;
      *reinterpret_cast<u32x4*>(sA + row * LD1 + c * 16) = ra[i];
      if (do_cs) ChunkSum<T>::add(ra[i], csum);      // this thread always owns chunk column tid % C1
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int id = tid + 256 * i, row = id / C2, c = id % C2;
      *reinterpret_cast<u32x4*>(sB + row * LD2 + c * 16) = rb[i];
    }
  };

  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (mbeg < mend) {
    gload(mbeg);
    lstore();
    __syncthreads();
    for (int m0 = mbeg; m0 < mend; m0 += TN_BM) {
      const bool more = m0 + TN_BM < mend;
      if (more) gload(m0 + TN_BM);
#pragma unroll
      for (int ks = 0; ks < KSUB; ++ks) {
        typename MM::Frag fa[4], fb[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = TrFrag<T>::ld(sA, LD1, ks * MM::KSTEP, w1 * 64 + i * 16, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = TrFrag<T>::ld(sB, LD2, ks * MM::KSTEP, w2 * 32 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = MM::mma(fb[j], fa[i], acc[i][j]);
      }
      if (more) {
        __syncthreads();
        lstore();
        __syncthreads();
      }
    }
  }
  if (do_cs) {     // column sums of the A slice: reduce the 256/C1 threads that share a chunk column (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(sA);                 // [256/C1][TN_B1]
    const int c = tid % C1, grp = tid / C1;
#pragma unroll
    for (int e = 0; e < CH; ++e) red[grp * TN_B1 + c * CH + e] = csum[e];
    __syncthreads();
    if (tid < TN_B1 && n10 + tid < g.N1) {
      float t = 0.f;
      for (int q = 0; q < 256 / C1; ++q) t += red[q * TN_B1 + tid];
      g.bpart[(size_t)blockIdx.z * g.N1 + n10 + tid] = t;
    }
  }
  // partial[z][n1][n2]: lane owns n1 = .. + (lane&15), n2 = .. + (lane>>4)*4 + {0..3}
  float* P = g.part + (size_t)blockIdx.z * g.N1 * g.N2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n1 = n10 + w1 * 64 + i * 16 + (lane & 15);
    if (n1 >= g.N1) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n2 = n20 + w2 * 32 + j * 16 + (lane >> 4) * 4;
      if (n2 + 3 < g.N2 && (g.N2 & 3) == 0) *reinterpret_cast<f32x4*>(P + (size_t)n1 * g.N2 + n2) = acc[i][j];
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n2 + e < g.N2) P[(size_t)n1 * g.N2 + n2 + e] = acc[i][j][e];
      }
    }
  }
}

// ---- big-tile variant (bf16): one workgroup of 8 waves owns a B1 x B2 output tile (192x256, 256x192 or
// 192x192), so for the DeiT shapes A (or B) is read exactly once and the other operand <= 4 times through
// L2, instead of 12x / 2x with the 128x64 tile.  The bias gradient (column sums of A) comes from one extra
// MFMA column against an all-ones fragment in the waves of the first column of tiles.
template <typename TA, int B1, int B2, int W1, int W2>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_big(TnArgs g) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int LD1 = B1 * 2 + 32, LD2 = B2 * 2 + 32;
  constexpr int C1 = B1 / 8, C2 = B2 / 8;
  constexpr int NLA = TN_BM * C1 / 512, NLB = TN_BM * C2 / 512;
  constexpr int TI = B1 / W1 / 16, TJ = B2 / W2 / 16;
  static_assert(W1 * W2 == 8 && (TN_BM * C1) % 512 == 0 && (TN_BM * C2) % 512 == 0, "tile config");
  __shared__ __attribute__((aligned(16))) char sA[TN_BM * LD1];
  __shared__ __attribute__((aligned(16))) char sB[TN_BM * LD2];
  const TA* __restrict__ A = reinterpret_cast<const TA*>(g.A);
  const T* __restrict__ B = reinterpret_cast<const T*>(g.B);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int w1 = w / W2, w2 = w % W2;
  const int n10 = blockIdx.x * B1, n20 = blockIdx.y * B2;
  const int mbeg = blockIdx.z * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);
  const bool do_cs = g.bpart != nullptr && blockIdx.y == 0 && w2 == 0;

  u32x4 ra[NLA], rb[NLB];
  auto gload = [&](int m0) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int id = tid + 512 * i, row = id / C1, c = id % C1;
      const int m = m0 + row;
      ra[i] = (m < mend) ? ChunkLoad<TA, T>::ld(A + (size_t)m * g.lda + n10 + c * 8) : z;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int id = tid + 512 * i, row = id / C2, c = id % C2;
      const int m = m0 + row;
      rb[i] = (m < mend) ? ChunkLoad<T, T>::ld(B + (size_t)m * g.ldb + n20 + c * 8) : z;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NLA; ++i) { const int id = tid + 512 * i; *reinterpret_cast<u32x4*>(sA + (id / C1) * LD1 + (id % C1) * 16) = ra[i]; }
#pragma unroll
    for (int i = 0; i < NLB; ++i) { const int id = tid + 512 * i; *reinterpret_cast<u32x4*>(sB + (id / C2) * LD2 + (id % C2) * 16) = rb[i]; }
  };

  f32x4 acc[TI][TJ], cs[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const typename MM::Frag ones = __builtin_bit_cast(typename MM::Frag, ones_u);

  if (mbeg < mend) {
    gload(mbeg);
    lstore();
    __syncthreads();
    for (int m0 = mbeg; m0 < mend; m0 += TN_BM) {
      const bool more = m0 + TN_BM < mend;
      if (more) gload(m0 + TN_BM);
#pragma unroll
      for (int ks = 0; ks < TN_BM / 32; ++ks) {
        typename MM::Frag fa[TI], fb[TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) fa[i] = TrFrag<T>::ld(sA, LD1, ks * 32, w1 * (B1 / W1) + i * 16, lane);
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[j] = TrFrag<T>::ld(sB, LD2, ks * 32, w2 * (B2 / W2) + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[i][j] = MM::mma(fb[j], fa[i], acc[i][j]);
          if (do_cs) cs[i] = MM::mma(ones, fa[i], cs[i]);
        }
      }
      if (more) {
        __syncthreads();
        lstore();
        __syncthreads();
      }
    }
  }
  if (do_cs && (lane >> 4) == 0) {      // every row of cs[i] holds the column sums of n1 = .. + (lane & 15)
#pragma unroll
    for (int i = 0; i < TI; ++i) g.bpart[(size_t)blockIdx.z * g.N1 + n10 + w1 * (B1 / W1) + i * 16 + (lane & 15)] = cs[i][0];
  }
  float* P = g.part + (size_t)blockIdx.z * g.N1 * g.N2;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int n1 = n10 + w1 * (B1 / W1) + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int n2 = n20 + w2 * (B2 / W2) + j * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4*>(P + (size_t)n1 * g.N2 + n2) = acc[i][j];
    }
  }
}

// ---- big-tile variant with LDS-DMA staging (bf16 A and B): same tiles and split-M scheme as k_gemm_tn_big, but the operand
// rows go HBM -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write phase) through a ring of four 32-row
// images, three of them in flight while the fourth is multiplied: ~90 KB outstanding per CU instead of one 57 KB tile that
// was requested a single MFMA phase before it was needed (the register-staged kernel stalls ~3 us of every 4.6 us step on
// it).  One raw s_barrier per step; the waves wait on their own DMA with a counted s_waitcnt vmcnt, so the two younger
// stages stay in flight across the barrier (hipcc would drain vmcnt(0) at a __syncthreads()).  The LDS destination of a
// DMA instruction is wave-base + lane*16, so an image is the lane-linear sequence of 16-byte slots [row][chunk] with the two
// pad slots of every row (conflict-free transpose reads) filled by lanes that re-load chunk 0.
// (r5: a ring of FIVE stages, 151 KB: dW2 on 128 workgroups 84 us against 78 -- at 19.6 GB/s per CU the kernel is not waiting for a deeper ring; profiles/r5zz)
constexpr int TD_BM = 32, TD_NST = 4;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Transposing fragment read issued as inline assembly: for a compiler-visible LDS load hipcc inserts s_waitcnt vmcnt(0) (any
// outstanding LDS-DMA may alias it), which would drain the whole ring before every step.  Same k-slot map as TrFrag<bf16_t>.
__device__ __forceinline__ s16x4 ds_tr16_asm(const char* p) {
  s16x4 v;
  const unsigned addr = (unsigned)(size_t)(LDS_PTR(char))p;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
__device__ __forceinline__ bf16x8 tr_frag_asm(const char* tile, int ld, int c0, int lane) {
  const int i = lane & 15, gq = lane >> 4;
  const char* p = tile + (4 * gq + (i >> 2)) * ld + (c0 + (i & 3) * 4) * 2;
  const s16x4 lo = ds_tr16_asm(p), hi = ds_tr16_asm(p + 16 * ld);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
  return __builtin_bit_cast(bf16x8, v);
}

// Timing probes (tools/tn_probes.sh builds copies of the library with -DUVC_TN_PROBE=n; 0 in the product): 1 = the operand ring only (no fragment reads, no MFMAs),
// 2 = the compute only (no DMA requests: the stages' LDS images hold whatever they hold), 3 = no partial-tile store.  Wrong results on purpose.
#ifndef UVC_TN_PROBE
#define UVC_TN_PROBE 0
#endif
template <int B1, int B2, int W1, int W2>
__global__ __launch_bounds__(512, 2) void k_gemm_tn_dma(TnArgs g) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  constexpr int LD1 = B1 * 2 + 32, LD2 = B2 * 2 + 32;
  constexpr int S1 = LD1 / 16, S2 = LD2 / 16;
  constexpr int NSLOT = TD_BM * (S1 + S2);
  constexpr int NWI = (NSLOT + 63) / 64;                 // DMA wave-instructions per stage
  constexpr int PER = (NWI + 7) / 8;                     // per wave (waves past NWI aim theirs at a dummy KB)
  constexpr int IMG = NWI * 1024;
  constexpr int TI = B1 / W1 / 16, TJ = B2 / W2 / 16;
  static_assert(W1 * W2 == 8 && PER <= 5, "tile config");
  extern __shared__ __attribute__((aligned(16))) char smem_td[];
  char* const dummy = smem_td + TD_NST * IMG;
  const char* __restrict__ A = reinterpret_cast<const char*>(g.A);
  const char* __restrict__ B = reinterpret_cast<const char*>(g.B);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int w1 = w / W2, w2 = w % W2;
  // Workgroup -> (output tile, M split).  The dispatcher deals workgroups to the 8 XCDs round-robin; with at most one workgroup per
  // CU (grid <= 256) the tiles of one split are renumbered onto ONE XCD, adjacent in its sequence, so the operand rows the tiles
  // share are fetched from HBM once per L2 instead of once per tile (fetch traffic 271 -> 194 MB for dW2).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int tiles = gridDim.x * gridDim.y, total = tiles * gridDim.z;
    if (g.xcd_remap && tiles > 1 && total <= 256) {
      const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
      const int L = xcd * q + (xcd < r ? xcd : r) + slot;       // position in the XCD-major order
      const int t = L % tiles;
      bz = L / tiles; bx = t % gridDim.x; by = t / gridDim.x;
    }
  }
  const int n10 = bx * B1, n20 = by * B2;
  const int mbeg = bz * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);
  const bool do_cs = g.bpart != nullptr && by == 0 && w2 == 0;

  // this lane's source of DMA instruction q of a stage: byte offset at row 0 of the stage, bytes per row, row inside the stage
  int64_t goff[PER];
  int gstr[PER], grow[PER];
  bool isA[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    int slot = (w + 8 * q) * 64 + lane;
    if (slot >= NSLOT) slot = 0;                          // dummy instruction / tail lanes: any valid source
    if (slot < TD_BM * S1) {
      const int row = slot / S1, c = slot % S1;
      isA[q] = true; grow[q] = row; gstr[q] = g.lda * 2;
      goff[q] = (int64_t)row * g.lda * 2 + (n10 + (c < B1 / 8 ? c : 0) * 8) * 2;
    } else {
      const int s2 = slot - TD_BM * S1, row = s2 / S2, c = s2 % S2;
      isA[q] = false; grow[q] = row; gstr[q] = g.ldb * 2;
      goff[q] = (int64_t)row * g.ldb * 2 + (n20 + (c < B2 / 8 ? c : 0) * 8) * 2;
    }
  }
  auto issue_one = [&](int t, int q) {                   // DMA instruction q of stage t (full stages only: every row is inside the split)
    if (UVC_TN_PROBE == 2) return;
    const int m0 = mbeg + t * TD_BM;
    char* img = smem_td + (t % TD_NST) * IMG;
    const char* src = (isA[q] ? A : B) + goff[q] + (int64_t)m0 * gstr[q];
    char* dst = (w + 8 * q < NWI) ? img + (w + 8 * q) * 1024 : dummy;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int q = 0; q < PER; ++q) issue_one(t, q);
  };

  f32x4 acc[TI][TJ], cs[TI];
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const typename MM::Frag ones = __builtin_bit_cast(typename MM::Frag, ones_u);

  // tn >= 0: the DMA instructions of stage tn go BETWEEN the MFMA rows (their issue cost hides under the matrix pipe; issued as a burst
  // behind the barrier they cost every wave of the lock-stepped workgroup 100-200 cycles each with nothing else to issue)
  auto compute = [&](const char* sA, int tn) {
    if (UVC_TN_PROBE == 1) {                                 // the ring alone: the next stage's requests, nothing else
      if (tn >= 0) {
#pragma unroll
        for (int q = 0; q < PER; ++q) issue_one(tn, q);
      }
      return;
    }
    const char* sB = sA + TD_BM * LD1;
    typename MM::Frag fa[TI], fb[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) fa[i] = tr_frag_asm(sA, LD1, w1 * (B1 / W1) + i * 16, lane);
#pragma unroll
    for (int j = 0; j < TJ; ++j) fb[j] = tr_frag_asm(sB, LD2, w2 * (B2 / W2) + j * 16, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);                       // no MFMA may move above the wait of the hand-issued reads
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[i][j] = MM::mma(fb[j], fa[i], acc[i][j]);
      if (do_cs) cs[i] = MM::mma(ones, fa[i], cs[i]);
      if (tn >= 0) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (q * TI / PER == i) issue_one(tn, q);
      }
    }
  };
  const int rows = mbeg < mend ? mend - mbeg : 0;
  const int nfull = rows / TD_BM;
  for (int t = 0; t < TD_NST - 1 && t < nfull; ++t) issue(t);
  for (int t = 0; t < nfull; ++t) {
    const int younger = min(TD_NST - 2, nfull - 1 - t);      // stages issued after t that may stay in flight
    if (younger >= 2) wait_vmcnt<2 * PER>();
    else if (younger == 1) wait_vmcnt<PER>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                            // stage t complete for every wave; image (t-1) % 4 is free again
    compute(smem_td + (t % TD_NST) * IMG, t + TD_NST - 1 < nfull ? t + TD_NST - 1 : -1);
  }
  if (rows % TD_BM) {                                        // partial last stage of the split: through registers, rows past mend = 0
    __builtin_amdgcn_s_barrier();
    const int m0 = mbeg + nfull * TD_BM;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (m0 + grow[q] < mend) v = *reinterpret_cast<const u32x4*>((isA[q] ? A : B) + goff[q] + (int64_t)m0 * gstr[q]);
      if (w + 8 * q < NWI) *reinterpret_cast<u32x4*>(smem_td + (w + 8 * q) * 1024 + lane * 16) = v;
    }
    __syncthreads();
    compute(smem_td, -1);
  }
  if (do_cs && (lane >> 4) == 0) {
#pragma unroll
    for (int i = 0; i < TI; ++i) g.bpart[(size_t)bz * g.N1 + n10 + w1 * (B1 / W1) + i * 16 + (lane & 15)] = cs[i][0];
  }
  float* P = g.part + (size_t)bz * g.N1 * g.N2;
  if (UVC_TN_PROBE == 3 && g.M > 0) return;
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int n1 = n10 + w1 * (B1 / W1) + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int n2 = n20 + w2 * (B2 / W2) + j * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4*>(P + (size_t)n1 * g.N2 + n2) = acc[i][j];
    }
  }
}

// ---- 256 x 256 weight-gradient tiles with two wave groups half a phase apart (k_gemm_nt8p's schedule, transposing reads).
// k_gemm_tn_dma keeps its eight waves in lock step: [wait | barrier | 24 transposing fragment reads | wait | 32 MFMAs] per 32 rows, the read
// latency in front of every MFMA block (0.82-0.86 PFLOP/s on DeiT-Base's shapes).  Here a k-step is 64 rows of the split, a wave's 128 x 64
// block of the tile is four quadrants as in k_gemm_nt8p -- (A0, B0) (A0, B1) (A1, B1) (A1, B0), 16 MFMAs each -- the M groups w1 = 0 / 1 run
// half a phase apart, the fragments of a phase are requested inside the MFMA block of the phase before, and the operand rows go HBM / L2 -> LDS
// by buffer_load ... lds into two buffers of four 16-KB slots (A'0, B'0, B'1, A'1: read in phases 0, 0, 1, 2; one slot of a later k-step
// requested per phase; vmcnt(6) in phases 2, 3, 0).  A slot is 64 rows (of M) x 128 columns: [row][16 chunks of 16 B], chunk c of row r holds
// the columns of global chunk c ^ 2 (r & 7) -- the eight rows a 32-lane group of ds_read_b64_tr_b16 touches land in eight different 32-byte
// bank groups (the padded rows of k_gemm_tn_dma, without the pad slots).  Rows past the split's end are requested past the buffer
// descriptor's range and arrive as zeros, so a split need not be a whole number of k-steps.  The bias gradient (column sums of A: an
// all-ones operand against the A fragments) is spread over the four waves of a row group, two column blocks each.  Same k order per
// accumulator as k_gemm_tn_dma: the partial tiles are bit-identical.
constexpr int T8_SLOT = 64 * 256, T8_BUF = 4 * T8_SLOT, T8_LDS = 2 * T8_BUF;       // 128 KB

template <int OFF> __device__ __forceinline__ u32x2 ds_tr16_off(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
struct T8Frag { u32x2 lo, hi; };
__device__ __forceinline__ bf16x8 t8_frag(const T8Frag& f) {
  return __builtin_bit_cast(bf16x8, u32x4{f.lo[0], f.lo[1], f.hi[0], f.hi[1]});
}

// RI / RJ: 16-column blocks of N1 / N2 per wave (2 x 4 waves): tiles 256 x 256 (8, 4), 192 x 256 (6, 4), 256 x 192 (8, 3), 192 x 192 (6, 3).  The slot
// images keep their 64 x 128 shape; the columns a narrower tile does not have are requested out of range (zeros, no traffic).
template <int RI, int RJ>
__global__ __launch_bounds__(512, 2) void k_gemm_tn8p(TnArgs g) {
  typedef Mma<bf16_t> MM;
  constexpr int R0 = (RI + 1) / 2, R1 = RI - R0, J0 = (RJ + 1) / 2, J1 = RJ - J0, B1 = 32 * RI, B2 = 64 * RJ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w1 = w >> 2, w2 = w & 3;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {                                                       // (tile, split) -> XCD-major order, as k_gemm_tn_dma
    const int tiles = gridDim.x * gridDim.y, total = tiles * gridDim.z;
    if (g.xcd_remap && tiles > 1 && total <= 256) {
      const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
      const int L = xcd * q + (xcd < r ? xcd : r) + slot;
      const int t = L % tiles;
      bz = L / tiles; bx = t % gridDim.x; by = t / gridDim.x;
    }
  }
  const int n10 = bx * B1, n20 = by * B2;
  const int mbeg = bz * g.rows_per_split;
  const int mend = min(g.M, mbeg + g.rows_per_split);
  const int nk = (mend - mbeg + 63) / 64;
  const bool do_cs = g.bpart != nullptr && by == 0;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)((size_t)mend * g.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)((size_t)mend * g.ldb * 2), 0x00020000);
  // ---- LDS-DMA: a slot is 16 wave-instructions of 4 rows x 256 B; wave w issues instructions w (rows 4w ..) and w + 8 (rows 32 + 4w ..)
  unsigned laneA0, laneA1, laneB0, laneB1;
  {
    const int r = 4 * w + (lane >> 4), cg = (lane & 15) ^ (2 * (r & 7));        // global chunk this lane fetches
    const int ca = cg & 7, cb = cg & 3;                                          // chunk inside a row group's / column group's share of the slot
    const int colA = (cg >> 3) * (16 * RI) + ca * 8, colB = (cg >> 2) * (16 * RJ) + cb * 8;
    constexpr unsigned OOB = 0x80000000u;
    laneA0 = ca < 2 * R0 ? (unsigned)((r * g.lda + n10 + colA) * 2) : OOB;
    laneA1 = ca < 2 * R1 ? (unsigned)((r * g.lda + n10 + colA + 16 * R0) * 2) : OOB;
    laneB0 = cb < 2 * J0 ? (unsigned)((r * g.ldb + n20 + colB) * 2) : OOB;
    laneB1 = cb < 2 * J1 ? (unsigned)((r * g.ldb + n20 + colB + 16 * J0) * 2) : OOB;
  }
  const unsigned hiA = (unsigned)(32 * g.lda * 2), hiB = (unsigned)(32 * g.ldb * 2);
  // request q (0..7) of the k-step whose first row is m: slot q >> 1 (A'0, B'0, B'1, A'1), instruction q & 1
  auto issue = [&](int q, int m, int buf) {
    char* dst = smem + buf * T8_BUF + (q >> 1) * T8_SLOT + (q & 1) * 8192 + w * 1024;
    const int slot = q >> 1;
    const unsigned ra = (unsigned)(m * g.lda * 2), rb = (unsigned)(m * g.ldb * 2);
    if (slot == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)dst, 16, laneA0 + (ra + (q & 1) * hiA), 0, 0, 0);
    else if (slot == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)dst, 16, laneA1 + (ra + (q & 1) * hiA), 0, 0, 0);
    else if (slot == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)dst, 16, laneB0 + (rb + (q & 1) * hiB), 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)dst, 16, laneB1 + (rb + (q & 1) * hiB), 0, 0, 0);
  };
  // ---- fragment addresses (buffer 0, rows 0-31 of the slot, first of the two reads; + 4096: rows + 16; + 8192: rows 32-63)
  const unsigned s0 = lds_addr(smem);
  unsigned fa[4], fb[2];
  {
    const int il = lane & 15, gq = lane >> 4;
    const int R = 4 * gq + (il >> 2), sw = R & 7, b = (il >> 1) & 1, half = il & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = s0 + (unsigned)(R * 256 + ((((w1 * 4 + i) ^ sw) * 2 + b) * 16) + half * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = s0 + (unsigned)(R * 256 + ((((w2 * 2 + j) ^ sw) * 2 + b) * 16) + half * 8);
  }
  T8Frag FA[4][2], FB0[2][2], FB1[2][2];
  f32x4 acc[RI][RJ], cs[2];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < RJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  cs[0] = f32x4{0.f, 0.f, 0.f, 0.f}; cs[1] = cs[0];
  const u32x4 ones_u = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const typename MM::Frag ones = __builtin_bit_cast(typename MM::Frag, ones_u);

#define T8_RD(F, ADDR, SLOT, KS) { F.lo = ds_tr16_off<(SLOT) * T8_SLOT + (KS) * 8192>(ADDR); F.hi = ds_tr16_off<(SLOT) * T8_SLOT + (KS) * 8192 + 4096>(ADDR); }
#define T8_RD_B(DST, SLOT, BO, NJ)                                                                                    \
  _Pragma("unroll") for (int j = 0; j < (NJ); ++j) { T8_RD(DST[j][0], fb[j] + (BO), SLOT, 0) T8_RD(DST[j][1], fb[j] + (BO), SLOT, 1) }
#define T8_RD_A(I, SLOT, BO) { T8_RD(FA[I][0], fa[I] + (BO), SLOT, 0) T8_RD(FA[I][1], fa[I] + (BO), SLOT, 1) }
#define T8_MMA4(I, AI, FB, JO, NJ)                                                                                    \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                     \
  _Pragma("unroll") for (int j = 0; j < (NJ); ++j)                                                                     \
    acc[AI][(JO) + j] = MM::mma(t8_frag(FB[j][ks]), t8_frag(FA[I][ks]), acc[AI][(JO) + j]);
#define T8_CS(U, I) { cs[U] = MM::mma(ones, t8_frag(FA[I][0]), cs[U]); cs[U] = MM::mma(ones, t8_frag(FA[I][1]), cs[U]); }
  // column sums: wave w2 of a row group takes the column blocks 2 w2, 2 w2 + 1 of its RI; HALF = 0: blocks in A0 (phase 0), 1: in A1 (phase 2)
#define T8_CS_PHASE(HALF)                                                                                             \
  if (do_cs) {                                                                                                         \
    _Pragma("unroll") for (int W = 0; W < 4; ++W)                                                                      \
      if (w2 == W) {                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                \
          const int gi = 2 * W + u;                                                                                    \
          if (gi < RI && (gi >= R0) == ((HALF) == 1)) { if (u == 0) T8_CS(0, gi - (HALF) * R0) else T8_CS(1, gi - (HALF) * R0) }  \
        }                                                                                                              \
      }                                                                                                                \
  }
#define T8_REQ(Q0, M, SBUF, WAIT)                                                                                     \
  issue(Q0, M, SBUF); issue(Q0 + 1, M, SBUF);                                                                          \
  if (WAIT) wait_vm<6>();                                                                                              \
  __builtin_amdgcn_s_barrier();                                                                                        \
  wait_lgkm<0>();                                                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_setprio(1);
#define T8_END                                                                                                        \
  __builtin_amdgcn_s_setprio(0);                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                        \
  __builtin_amdgcn_sched_barrier(0);

  // prologue: k-step 0 whole, A'0 and B'0 of k-step 1
#pragma unroll
  for (int q = 0; q < 8; ++q) issue(q, mbeg, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue(q, mbeg + 64, 1);
  wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  T8_RD_B(FB0, 1, 0u, J0)
#pragma unroll
  for (int i = 0; i < R0; ++i) T8_RD_A(i, 0, 0u)
  __builtin_amdgcn_sched_barrier(0);
  if (w1 == 1) __builtin_amdgcn_s_barrier();      // group 1 runs half a phase behind from here on
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const unsigned bo = (unsigned)(buf * T8_BUF), bon = bo ^ (unsigned)T8_BUF;
    const int m1 = mbeg + (kt + 1) * 64, m2 = m1 + 64;   // (past the split's end: out of the descriptors' range, zeros nobody multiplies)
    // ---- phase 0: (A0, B0); requests B'1 of k-step kt + 1; reads B1 of this k-step
    T8_REQ(4, m1, buf ^ 1, true)
    T8_RD_B(FB1, 2, bo, J1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < R0; ++i) T8_MMA4(i, i, FB0, 0, J0)
    T8_CS_PHASE(0)
    T8_END
    // ---- phase 1: (A0, B1); requests A'1 of kt + 1; reads A1 behind the MFMAs that free A0's registers
    T8_REQ(6, m1, buf ^ 1, false)
#pragma unroll
    for (int i = 0; i < R0; ++i) {
      T8_MMA4(i, i, FB1, J0, J1)
      __builtin_amdgcn_sched_barrier(0);
      if (i < R1) T8_RD_A(i, 3, bo)
      __builtin_amdgcn_sched_barrier(0);
    }
    T8_END
    // ---- phase 2: (A1, B1); requests A'0 of kt + 2
    T8_REQ(0, m2, buf, true)
#pragma unroll
    for (int i = 0; i < R1; ++i) T8_MMA4(i, R0 + i, FB1, J0, J1)
    T8_CS_PHASE(1)
    T8_END
    // ---- phase 3: (A1, B0); requests B'0 of kt + 2; reads A0 and B0 of k-step kt + 1 (the other buffer)
    T8_REQ(2, m2, buf, true)
#pragma unroll
    for (int i = R1; i < R0; ++i) T8_RD_A(i, 0, bon)            // (registers phase 3 does not use)
#pragma unroll
    for (int i = 0; i < R1; ++i) {
      T8_MMA4(i, R0 + i, FB0, 0, J0)
      __builtin_amdgcn_sched_barrier(0);
      T8_RD_A(i, 0, bon)
      __builtin_amdgcn_sched_barrier(0);
    }
    T8_RD_B(FB0, 1, bon, J0)
    T8_END
  }
#undef T8_RD
#undef T8_RD_B
#undef T8_RD_A
#undef T8_MMA4
#undef T8_CS
#undef T8_CS_PHASE
#undef T8_REQ
#undef T8_END
  if (w1 == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier
  wait_vm<0>();
  wait_lgkm<0>();
  if (do_cs && (lane >> 4) == 0) {                // every row of cs[u] holds the column sums of n1 = .. + (lane & 15)
    float* bp = g.bpart + (size_t)bz * g.N1 + n10 + w1 * (16 * RI) + w2 * 32 + (lane & 15);       // column blocks 2 w2, 2 w2 + 1
    if (2 * w2 < RI) bp[0] = cs[0][0];
    if (2 * w2 + 1 < RI) bp[16] = cs[1][0];
  }
  float* P = g.part + (size_t)bz * g.N1 * g.N2;
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int n1 = n10 + w1 * (16 * RI) + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
      const int n2 = n20 + w2 * (16 * RJ) + j * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4*>(P + (size_t)n1 * g.N2 + n2) = acc[i][j];
    }
  }
}

// sum the split-M partial tiles (and partial column sums) in a fixed order.  VEC = 4: 16-byte accesses, 4 slices in flight
template <int VEC>
__global__ __launch_bounds__(256) void k_tn_reduce(const float* __restrict__ part, float* __restrict__ C, int n, int ldc,
                                                   int N2, int splits, float alpha, const float* alpha_ptr, float beta,
                                                   const float* __restrict__ bpart, float* __restrict__ bias_out, int N1) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i >= n + (bpart ? N1 : 0)) return;
  if (alpha_ptr) alpha *= *alpha_ptr;
  const bool main_part = i < n;
  const float* src = main_part ? part + i : bpart + (i - n);
  const size_t stride = main_part ? (size_t)n : (size_t)N1;
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  int z = 0;
  for (; z + 4 <= splits; z += 4) {
    float v[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (VEC == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(src + (size_t)(z + u) * stride); v[u][0] = t[0]; v[u][1 % VEC] = t[1]; v[u][2 % VEC] = t[2]; v[u][3 % VEC] = t[3]; }
      else v[u][0] = src[(size_t)(z + u) * stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e) s[e] += v[u][e];
  }
  for (; z < splits; ++z)
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] += src[(size_t)z * stride + e];
  float* dst = main_part ? C + (size_t)(i / N2) * ldc + (i % N2) : bias_out + (i - n);
#pragma unroll
  for (int e = 0; e < VEC; ++e) dst[e] = (beta != 0.f ? beta * dst[e] : 0.f) + alpha * s[e];
}

// 0 = generic 128x64 tiles; 1 = 192x256, 2 = 256x192, 3 = 192x192, 4 = 96x192 (bf16 big tiles)
static int tn_config(int N1, int N2) {
  // 5 = 256 x 256 (LDS-DMA kernel, bf16 operands): DeiT-Base's shapes.  131 flop per operand byte through LDS against 108 for 192 x 256; the
  // split-M kernels are bound by the L2 -> LDS rate of their 32-row stages (pipe 31 % busy at 0.75 PFLOP/s).  Wide matrices only: at
  // DeiT-Tiny / Small widths the 192-wide tiles divide the shapes and leave more splits.
  if (N1 % 256 == 0 && N2 % 256 == 0 && (N1 / 256) * (N2 / 256) >= 9) return 5;     // (r4: k_gemm_tn8p; 9 tiles = DeiT-Base's dW_proj)
  if (N1 % 192 == 0 && N2 % 256 == 0) return 1;
  if (N1 % 256 == 0 && N2 % 192 == 0) return 2;
  if (N1 % 192 == 0 && N2 % 192 == 0 && (N1 / 192) * (N2 / 192) >= 2) return 3;   // a single 192x192 tile would need 256 splits
  if (N1 == 192 && N2 == 192) return 4;       // dW_proj of DeiT-Tiny: two 96x192 tiles x 128 splits (the generic kernel ran it at 1.75 TB/s)
  return 0;
}
static void tn_tile(int cfg, int& b1, int& b2) { b1 = (cfg == 2 || cfg == 5) ? 256 : cfg == 4 ? 96 : cfg == 6 ? 128 : 192; b2 = (cfg == 1 || cfg == 5 || cfg == 6) ? 256 : 192; }
static int tn_splits(int M, int N1, int N2, int cfg) {
  int tiles, target;
  if (cfg == 0) { tiles = ceil_div(N1, TN_B1) * ceil_div(N2, TN_B2); target = 768; }
  else {
    int b1, b2; tn_tile(cfg, b1, b2); tiles = (N1 / b1) * (N2 / b2);
    // one 8-wave group per CU -- on HALF the CUs where the output is a handful of tiles (DeiT-Tiny's 192-wide weights: 2-3 tiles): a split writes a float32 partial
    // tile and the reduce reads it back, so 128 workgroups of twice the rows move half the partial bytes (50 MB of dW2's 245), and the dgrad chain on the other
    // stream gets the CUs this HBM-bound GEMM does not need.  Measured in the step, same box (profiles/r5t): 256 / 192 / 128 / 96 / 64 workgroups
    // 11.58 / 11.44 / 11.38 / 11.64 / 12.31 ms (the stand-alone sum RISES, 11.2 -> 11.9 ms: these GEMMs are slower alone on half the chip).
    // The wider models keep the whole chip: DeiT-Small 15.6 -> 16.7 ms and DeiT-Base 23.4 -> 24.5 at 128 (192: +- 0 / -0.7 %), T2T-ViT-14 +0.7 %.
    // Measured again once the side stream had become the backward's critical path (one event per block, vit_engine.hip): 128 / 160 / 192 / 256 workgroups
    // 11.14 / 11.18 / 11.23 / 11.44 ms (profiles/r5z_ab_wgrad_targets.txt).
    // (r6, with the two-group schedule for these tiles: 96 / 128 / 192 / 256 workgroups measured again, profiles/r6i_ab_wgrad_targets.txt)
#ifndef UVC_TN_TARGET_FEW
#define UVC_TN_TARGET_FEW 128
#endif
    target = tiles <= 4 ? UVC_TN_TARGET_FEW : 256;
  }
  int splits = cfg == 0 ? ceil_div(target, tiles) : target / tiles;       // big tiles: one workgroup per CU (LDS), so at most 256 of them -- one more is a second round
  const int max_splits = ceil_div(M, TN_BM);
  if (splits > max_splits) splits = max_splits;
  // (r4) not more splits than rows justify: a split writes and the reduce re-reads a whole float32 tile, and below ~1 k rows per split that traffic is what the
  // kernel moves (dW_proj of DeiT-Tiny: 128 splits of 788 rows; T2T-ViT-14 at 25 k rows: partial bytes = operand bytes).  Measured in the step, same box:
  // >= 1024 rows: DeiT-Tiny 12.19 -> 12.06 ms (1536: 12.5), DeiT-Small / Base unchanged; the 192 x 192 tiles at M < 32 k (T2T-ViT-14) >= 2048 rows: 13.22 -> 12.88 ms
  const int rmin = (cfg == 3 && M < 32768) ? 2048 : 1024;
  if (cfg != 0 && M >= 2 * rmin && splits > M / rmin) splits = M / rmin;
  return splits < 1 ? 1 : splits;
}

extern "C" int uvc_gemm_tn_workspace_bytes(int M, int N1, int N2, int64_t* bytes, int* splits_out) {
  if (!bytes) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn_workspace_bytes: null");
  const int s0 = tn_splits(M, N1, N2, 0), s1 = tn_splits(M, N1, N2, tn_config(N1, N2));
  const int splits = s0 > s1 ? s0 : s1;                      // large enough for either kernel
  *bytes = (int64_t)splits * ((int64_t)N1 * N2 + N1) * 4;   // partial tiles + partial column sums
  if (splits_out) *splits_out = splits;
  return UVC_OK;
}

extern "C" int uvc_gemm_tn(const uvc_gemm_tn_args* p, void* stream) {
  if (!p || !p->A || !p->B || !p->C || !p->workspace) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn: null pointer");
  if (p->M <= 0 || p->N1 <= 0 || p->N2 <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn: empty problem");
  const int ch = (p->dtype == UVC_F32) ? 4 : 8;
  if (p->N1 % ch || p->N2 % ch || p->lda % ch || p->ldb % ch)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn: N1, N2, lda, ldb must be multiples of a 16-byte chunk");
  int64_t need; int splits;
  uvc_gemm_tn_workspace_bytes(p->M, p->N1, p->N2, &need, &splits);
  if (p->workspace_bytes < need) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn: workspace too small");
  int cfg = (p->dtype == UVC_BF16 && p->lda % 8 == 0 && p->ldb % 8 == 0) ? tn_config(p->N1, p->N2) : 0;
  if (cfg == 4 && p->a_is_f32) cfg = 0;                      // the 96x192 tile exists as an LDS-DMA (bf16 operands) kernel only
  if (cfg == 5 && p->a_is_f32) cfg = (p->N1 % 192 == 0 && p->N2 % 256 == 0) ? 1 : (p->N1 % 256 == 0 && p->N2 % 192 == 0) ? 2 : 0;
  // the 256 x 256 kernel addresses its operands with 32-bit buffer offsets: beyond 2 GB the problem goes to the generic kernel (64-bit
  // addresses; the workspace is sized for either, see uvc_gemm_tn_workspace_bytes) -- checked on the configuration that will RUN
  if (cfg == 5 && (size_t)(p->M + 256) * (p->lda > p->ldb ? p->lda : p->ldb) * 2 >= (1ull << 31)) cfg = 0;
  // variant 3 (A/B, r6): 128 x 256 tiles for the shapes that take 256 x 256 -- twice the tiles, half the splits: half the float32 partial bytes (DeiT-Base dW1: 66 -> 28 MB
  // written and read back per launch) for 25 % fewer MFMAs per fragment read.  Measured (profiles/r6n_tn_variants_base.txt, GEMM + reduce, M = 25 344): dW2 141.9 -> 168.1 us,
  // dW1 131.3 -> 154.2, dWqkv 98.4 -> 110.8, dWproj 49.4 -> 44.6: not the default
  if (cfg == 5 && p->variant == 3) cfg = 6;
  splits = tn_splits(p->M, p->N1, p->N2, cfg);
  TnArgs a;
  a.A = p->A; a.B = p->B; a.part = (float*)p->workspace; a.M = p->M; a.N1 = p->N1; a.N2 = p->N2; a.lda = p->lda; a.ldb = p->ldb;
  int rps = ceil_div(p->M, splits);
  rps = ceil_div(rps, TN_BM) * TN_BM;
  splits = ceil_div(p->M, rps);
  a.rows_per_split = rps;
  a.xcd_remap = 1;
  a.bpart = p->colsum_out ? a.part + (size_t)splits * p->N1 * p->N2 : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == UVC_F32) {
    if (!p->a_is_f32) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn: float32 mode needs float32 A");
    dim3 grid(ceil_div(p->N1, TN_B1), ceil_div(p->N2, TN_B2), splits);
    k_gemm_tn<float, float><<<grid, 256, 0, st>>>(a);
  } else if (p->dtype == UVC_BF16) {
    if (cfg == 0) {
      dim3 grid(ceil_div(p->N1, TN_B1), ceil_div(p->N2, TN_B2), splits);
      if (p->a_is_f32) k_gemm_tn<float, bf16_t><<<grid, 256, 0, st>>>(a);
      else k_gemm_tn<bf16_t, bf16_t><<<grid, 256, 0, st>>>(a);
    } else {
      int b1, b2;
      tn_tile(cfg, b1, b2);
      dim3 grid(p->N1 / b1, p->N2 / b2, splits);
#define TN_BIG(TA_) \
      if (cfg == 1) k_gemm_tn_big<TA_, 192, 256, 2, 4><<<grid, 512, 0, st>>>(a); \
      else if (cfg == 2) k_gemm_tn_big<TA_, 256, 192, 4, 2><<<grid, 512, 0, st>>>(a); \
      else k_gemm_tn_big<TA_, 192, 192, 2, 4><<<grid, 512, 0, st>>>(a);
#define TN_DMA_ONE(B1_, B2_, W1_, W2_) { \
        const int sh_ = TD_NST * (((TD_BM * (((B1_) * 2 + 32) / 16 + ((B2_) * 2 + 32) / 16) + 63) / 64) * 1024) + 1024; \
        UVC_MAX_LDS(sh_, k_gemm_tn_dma<B1_, B2_, W1_, W2_>); \
        k_gemm_tn_dma<B1_, B2_, W1_, W2_><<<grid, 512, sh_, st>>>(a); }
      if (p->a_is_f32) { TN_BIG(float) }           // float32 A (converted on load): register-staged kernel
      // (r6: these two stood BEHIND the unconditional cfg == 1 / cfg == 2 cases until now, i.e. variant 1 never reached them and the r4 / r6c "equal" were
      //  the ring kernel measured against itself)
      // r6: the two-group schedule for the 192 x 256 / 256 x 192 tiles too.  tools/tn_probes.sh on 126 workgroups (profiles/r6h_tn_probes.txt): the ring kernel's
      // operand ring ALONE 32 us, its compute ALONE (no requests) 60 us of the 68 -- [24 transposing reads | wait | 32 MFMAs] in lock step add up to ~1 900 clocks per
      // 32-row stage where either half needs ~800; k_gemm_tn8p requests a phase's fragments inside the MFMA block of the phase before: dW2 79.4 -> 64.6 us, dW1
      // 77.4 -> 60.9 (bit-identical partial tiles).  variant 2: the ring kernel.
      else if (p->variant != 2 && cfg == 1) { UVC_MAX_LDS(T8_LDS, k_gemm_tn8p<6, 4>); k_gemm_tn8p<6, 4><<<grid, 512, T8_LDS, st>>>(a); }
      else if (p->variant != 2 && cfg == 2) { UVC_MAX_LDS(T8_LDS, k_gemm_tn8p<8, 3>); k_gemm_tn8p<8, 3><<<grid, 512, T8_LDS, st>>>(a); }
      else if (cfg == 1) TN_DMA_ONE(192, 256, 2, 4)
      else if (cfg == 2) TN_DMA_ONE(256, 192, 4, 2)
      else if (cfg == 4) TN_DMA_ONE(96, 192, 2, 4)
      else if (cfg == 5) { UVC_MAX_LDS(T8_LDS, k_gemm_tn8p<8, 4>); k_gemm_tn8p<8, 4><<<grid, 512, T8_LDS, st>>>(a); }
      else if (cfg == 6) { UVC_MAX_LDS(T8_LDS, k_gemm_tn8p<4, 4>); k_gemm_tn8p<4, 4><<<grid, 512, T8_LDS, st>>>(a); }
      // 192 x 192 tiles (dW_qkv of DeiT-Tiny / Small, every block weight of T2T-ViT-14): the two-group schedule by default -- 45.9 -> 40.9 us at
      // 100 864 x 576 x 192, 67.6 -> 55.6 at 50 432 x 1152 x 384, 44 -> 37 on T2T's three shapes; bit-identical partial tiles (variant 2: the ring kernel).
      // The 192 x 256 / 256 x 192 tiles measured equal on both kernels and stay on the ring kernel (variant 1 moves them).
      else if (p->variant != 2 && cfg == 3) { UVC_MAX_LDS(T8_LDS, k_gemm_tn8p<6, 3>); k_gemm_tn8p<6, 3><<<grid, 512, T8_LDS, st>>>(a); }
      else TN_DMA_ONE(192, 192, 2, 4)
#undef TN_DMA_ONE
#undef TN_BIG
    }
  } else return uvc_set_error_msg(UVC_ERR_ARG, "uvc_gemm_tn: dtype must be UVC_F32 or UVC_BF16");
  UVC_CHECK_LAUNCH();
  const int n = p->N1 * p->N2;
  const int tot = n + (a.bpart ? p->N1 : 0);
  if (n % 4 == 0 && p->N2 % 4 == 0 && p->ldc % 4 == 0 && p->N1 % 4 == 0)
    k_tn_reduce<4><<<ceil_div(tot / 4, 256), 256, 0, st>>>(a.part, p->C, n, p->ldc, p->N2, splits, p->alpha, p->alpha_ptr, p->beta, a.bpart, p->colsum_out, p->N1);
  else
    k_tn_reduce<1><<<ceil_div(tot, 256), 256, 0, st>>>(a.part, p->C, n, p->ldc, p->N2, splits, p->alpha, p->alpha_ptr, p->beta, a.bpart, p->colsum_out, p->N1);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
