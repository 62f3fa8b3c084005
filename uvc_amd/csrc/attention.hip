// Fused multi-head attention forward/backward for DeiT sequences (N <= 256, head_dim = 64) on gfx950.
// Reference arithmetic: UVC/models/model_distilled.py:175-185  (softmax(q k^T * hd^-0.5) v, dropout p=0).
//
// One workgroup (4 waves) per (batch, head).  N = 197/198 means a whole head's K and V fit in LDS
// (bf16: 2 x 224 x 144 B = 63 KB), so there is no online-softmax rescaling: each wave owns 16-row
// query tiles, computes S^T = K Q^T with the keys on the MFMA rows and the query on the lane
// (lane & 15), so the softmax row lives in one lane's registers + 2 wave shuffles, and P^T is
// already the B operand of O^T = V^T P^T.  V^T / K^T / Q^T / dO^T operands are gathered straight
// from the row-major LDS image with ds_read_b64_tr_b16 (bf16) or ds_read_b32 (float32 mode).
//
// Backward is two kernels with the same structure:
//   dQ  kernel: per query tile   S^T, dP^T = V dO^T, dS^T -> dQ^T = K^T dS^T     (K, V in LDS)
//   dKV kernel: per key tile     S = Q K^T, dP = dO V^T, dS -> dV^T = dO^T P, dK^T = Q^T dS  (Q, dO in LDS)
#include "common.h"
#include "../../include/uvc_kernels.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int CH = 8, KSTEP = 32;
  typedef bf16x8 Frag;
  static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // two accumulator tiles (8 floats: keys/queries 4g+{0..3} of tile A then tile B) -> operand fragment
  static __device__ __forceinline__ Frag pack(const f32x4& lo, const f32x4& hi) {
    u32x4 r;
    r[0] = pack_bf16x2(lo[0], lo[1]); r[1] = pack_bf16x2(lo[2], lo[3]);
    r[2] = pack_bf16x2(hi[0], hi[1]); r[3] = pack_bf16x2(hi[2], hi[3]);
    return __builtin_bit_cast(Frag, r);
  }
};
template <> struct Mma<float> {
  static constexpr int CH = 4, KSTEP = 16;
  typedef f32x4 Frag;
  static __device__ __forceinline__ f32x4 mma(const Frag& a, const Frag& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ Frag pack(const f32x4& lo, const f32x4&) { return lo; }
};

template <typename T> struct TrFrag;
template <> struct TrFrag<bf16_t> {
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
  static __device__ __forceinline__ f32x4 ld(const char* tile, int ld, int k0, int c0, int lane) {
    const int i = lane & 15, gq = lane >> 4;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float*>(tile + (k0 + 4 * gq + j) * ld + (c0 + i) * 4);
    return v;
  }
};

constexpr int HD = 64;
template <typename T> struct Geom {
  // LDS row stride (bytes).  bf16: 128 + 32 -- with ds_read_b128's lane groups ({0-3,12-15,20-27}, ...) and the
  // 2 x 32 groups of ds_read_b64_tr_b16, a 144-byte stride is 2-way conflicted on every operand read (PMC:
  // SQ_LDS_BANK_CONFLICT = 45 % of SQ_LDS_IDX_ACTIVE); 160 is conflict-free for both.  float32 keeps 256 + 16.
  static constexpr int ROWB = HD * (int)sizeof(T) + (sizeof(T) == 2 ? 32 : 16);
  static constexpr int KS = HD / Mma<T>::KSTEP;              // MFMA k-steps across head_dim
  static constexpr int TPS = Mma<T>::KSTEP / 16;             // 16-row tiles consumed per "pair" step (2 bf16, 1 f32)
};

// stage `rows` rows of 64 T (row stride ld_g elements) into LDS rows of ROWB bytes; rows >= nvalid zero.
// All global loads of a batch are issued before the first LDS write so their latencies overlap
// (a load->store loop exposes one HBM round trip per iteration: 7 trips per matrix at N = 197).
template <typename T>
__device__ __forceinline__ void stage_rows(char* lds, const T* g, size_t ld_g, int nvalid, int nrows_pad) {
  constexpr int CPR = HD / Mma<T>::CH;                        // chunks per row
  constexpr int BATCH = 8;
  const u32x4 z = {0u, 0u, 0u, 0u};
  const int total = nrows_pad * CPR;
  for (int base = 0; base < total; base += BATCH * (int)blockDim.x) {
    u32x4 v[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int id = base + i * (int)blockDim.x + (int)threadIdx.x;
      const int row = id / CPR, c = id % CPR;
      v[i] = (id < total && row < nvalid) ? *reinterpret_cast<const u32x4*>(g + (size_t)row * ld_g + c * Mma<T>::CH) : z;
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int id = base + i * (int)blockDim.x + (int)threadIdx.x;
      if (id < total) *reinterpret_cast<u32x4*>(lds + (id / CPR) * Geom<T>::ROWB + (id % CPR) * 16) = v[i];
    }
  }
}
// two matrices at once (K and V / Q and dO): twice the bytes in flight per thread
template <typename T>
__device__ __forceinline__ void stage_rows2(char* lds0, const T* g0, size_t ld0, char* lds1, const T* g1, size_t ld1, int nvalid, int nrows_pad) {
  constexpr int CPR = HD / Mma<T>::CH;
  constexpr int BATCH = 8;
  const u32x4 z = {0u, 0u, 0u, 0u};
  const int total = nrows_pad * CPR;
  for (int base = 0; base < total; base += BATCH * (int)blockDim.x) {
    u32x4 v0[BATCH], v1[BATCH];
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int id = base + i * (int)blockDim.x + (int)threadIdx.x;
      const int row = id / CPR, c = id % CPR;
      const bool ok = id < total && row < nvalid;
      v0[i] = ok ? *reinterpret_cast<const u32x4*>(g0 + (size_t)row * ld0 + c * Mma<T>::CH) : z;
      v1[i] = ok ? *reinterpret_cast<const u32x4*>(g1 + (size_t)row * ld1 + c * Mma<T>::CH) : z;
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const int id = base + i * (int)blockDim.x + (int)threadIdx.x;
      if (id < total) {
        *reinterpret_cast<u32x4*>(lds0 + (id / CPR) * Geom<T>::ROWB + (id % CPR) * 16) = v0[i];
        *reinterpret_cast<u32x4*>(lds1 + (id / CPR) * Geom<T>::ROWB + (id % CPR) * 16) = v1[i];
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag row_frag_lds(const char* lds, int row, int chunk) {
  return *reinterpret_cast<const typename Mma<T>::Frag*>(lds + row * Geom<T>::ROWB + chunk * 16);
}
template <typename T>
__device__ __forceinline__ typename Mma<T>::Frag row_frag_global(const T* g, size_t ld_g, int row, int nvalid, int chunk) {
  if (row < nvalid) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(g + (size_t)row * ld_g + chunk * Mma<T>::CH);
    return __builtin_bit_cast(typename Mma<T>::Frag, v);
  }
  const u32x4 z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(typename Mma<T>::Frag, z);
}
template <typename T> __device__ __forceinline__ float frag_dot(const typename Mma<T>::Frag& a, const typename Mma<T>::Frag& b);
template <> __device__ __forceinline__ float frag_dot<bf16_t>(const bf16x8& a, const bf16x8& b) {
  const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s += __uint_as_float(ua[e] << 16) * __uint_as_float(ub[e] << 16);
    s += __uint_as_float(ua[e] & 0xffff0000u) * __uint_as_float(ub[e] & 0xffff0000u);
  }
  return s;
}
template <> __device__ __forceinline__ float frag_dot<float>(const f32x4& a, const f32x4& b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

template <typename T> struct Store4;
template <> struct Store4<float> {
  static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Store4<bf16_t> {
  static __device__ __forceinline__ void st(bf16_t* p, f32x4 v) {
    u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(p) = r;
  }
};

// A 16 x 64 result tile in the accumulator layout (lane (row li, g): columns dt * 16 + 4 g .. + 3 of dt = 0 .. 3) leaves as 16 bytes per lane: lanes
// g and g ^ 1 exchange halves (v_permlane16_swap: the odd 16-lane rows of one register with the even rows of the other), after which an even g holds
// columns 8 (g >> 1) .. + 7 of block dtA and an odd g those of block dtB -- two store instructions of 64-byte row pieces where the 8-byte form
// issues four of 32-byte pieces (the store ISSUE, not HBM, was what the one-pass backward's first form spent half its time on; float32: as before).
template <typename T> __device__ __forceinline__ void store_tile16(T* rowp, int g, const f32x4 (&t)[4], float mul) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x2 x, y;
      x[0] = pack_bf16x2(t[2 * h][0] * mul, t[2 * h][1] * mul); x[1] = pack_bf16x2(t[2 * h][2] * mul, t[2 * h][3] * mul);
      y[0] = pack_bf16x2(t[2 * h + 1][0] * mul, t[2 * h + 1][1] * mul); y[1] = pack_bf16x2(t[2 * h + 1][2] * mul, t[2 * h + 1][3] * mul);
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(x[e], y[e], false, false);
        o[e] = r[0]; o[2 + e] = r[1];
      }
      *reinterpret_cast<u32x4*>(rowp + (2 * h + (g & 1)) * 16 + (g >> 1) * 8) = o;
    }
  } else {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 v = t[dt];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= mul;
      Store4<T>::st(rowp + dt * 16 + g * 4, v);
    }
  }
}

// the same for a 16 x 32 tile (two column blocks): one 16-byte store per lane
__device__ __forceinline__ void store_tile16_half(bf16_t* rowp, int g, const f32x4 (&t)[2]) {
  u32x2 x, y;
  x[0] = pack_bf16x2(t[0][0], t[0][1]); x[1] = pack_bf16x2(t[0][2], t[0][3]);
  y[0] = pack_bf16x2(t[1][0], t[1][1]); y[1] = pack_bf16x2(t[1][2], t[1][3]);
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const auto r = __builtin_amdgcn_permlane16_swap(x[e], y[e], false, false);
    o[e] = r[0]; o[2 + e] = r[1];
  }
  *reinterpret_cast<u32x4*>(rowp + (g & 1) * 16 + (g >> 1) * 8) = o;
}

struct AttnArgs {
  const void* qkv;   // [B, N, 3, H, 64]
  void* o;           // [B, N, H, 64]
  float* lse;        // [B, H, N]
  const void* dout;  // [B, N, H, 64]
  void* dqkv;        // [B, N, 3, H, 64]
  float* delta;      // [B, H, N]
  int B, N, H;
  float scale;
  const int* head_keep;
};

// ------------------------------------------------------------------------------------------------
// forward.  NT16 = number of 16-key tiles (multiple of TPS).
// The keys are taken in blocks of at most 8 tiles with a running maximum / sum (two blocks at N = 197): the scores of one
// block (32 VGPRs) instead of all 56 stay live, the kernel fits 128 VGPRs and runs 8 waves per workgroup, two workgroups per CU
// (the K / V image in LDS is the limit) = 4 waves per SIMD to hide its MFMA -> max -> exp -> MFMA chain.
// NFULL >= 0: the caller guarantees N / 16 == NFULL, so which tiles hold padded keys is known at compile time -- tiles below NFULL
// carry no mask code at all, tile NFULL masks by element, tiles above it are all padding (no QK^T MFMAs either).  With N only known
// at run time (NFULL = -1) hipcc if-converts the wave-uniform "last tile?" test and EVERY tile pays 8 v_cndmask plus the v_readlane
// reloads of their spilled SGPR masks, and consumes its scores straight behind the two MFMAs that produce them (s_nop 6-7):
// 180 of the ~600 VALU instructions of a query tile.
template <typename T, int T0, int NTB, int NFULL>
__device__ __forceinline__ void attn_key_block(const char* sK, const char* sV, const typename Mma<T>::Frag (&qf)[Geom<T>::KS], int N, float c2, int lane, int g, int li,
                                               float& m_run, float& l_run, f32x4 (&ot)[4]) {
  typedef Mma<T> MM;
  typedef Geom<T> G;
  __builtin_amdgcn_sched_barrier(0);              // keep the next block's fragment reads from being hoisted over this one (VGPRs)
  f32x4 st[NTB];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NTB; ++t) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (NFULL >= 0 && T0 + t > NFULL) {             // a tile of padding only
      st[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) c = MM::mma(row_frag_lds<T>(sK, (T0 + t) * 16 + li, ks * 4 + g), qf[ks], c);
    if (NFULL >= 0 ? T0 + t == NFULL : (T0 + t) * 16 + 16 > N) {      // only the last tile(s) hold padded keys
#pragma unroll
      for (int e = 0; e < 4; ++e) if ((T0 + t) * 16 + g * 4 + e >= N) c[e] = -INFINITY;
    }
    mx = fmaxf(mx, fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3])));
    st[t] = c;
  }
  mx = max_rows4(mx);
  const float m_new = fmaxf(m_run, mx);
  // p = exp(scale*(s - max)) = exp2(s*c2 - max*c2): one FMA + one v_exp_f32 per score (scale > 0)
  const float mb = m_new * c2;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float p = __builtin_amdgcn_exp2f(st[t][e] * c2 - mb); st[t][e] = p; sum += p; }
  sum = sum_rows4(sum);
  if (T0 > 0) {                                     // rescale what the earlier blocks accumulated
    const float alpha = __builtin_amdgcn_exp2f(m_run * c2 - mb);
    l_run = l_run * alpha + sum;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 4; ++e) ot[dt][e] *= alpha;
  } else {
    l_run = sum;
  }
  m_run = m_new;
#pragma unroll
  for (int s = 0; s < NTB / G::TPS; ++s) {
    const typename MM::Frag pf = MM::pack(st[s * G::TPS], st[s * G::TPS + G::TPS - 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = MM::mma(TrFrag<T>::ld(sV, G::ROWB, (T0 / G::TPS + s) * MM::KSTEP, dt * 16, lane), pf, ot[dt]);
  }
}

template <typename T, int T0, int NT16, int ATT_BT, int NFULL>
__device__ __forceinline__ void attn_key_blocks(const char* sK, const char* sV, const typename Mma<T>::Frag (&qf)[Geom<T>::KS], int N, float c2, int lane, int g, int li,
                                                float& m_run, float& l_run, f32x4 (&ot)[4]) {
  if constexpr (T0 < NT16) {
    constexpr int NTB = (NT16 - T0) < ATT_BT ? (NT16 - T0) : ATT_BT;
    attn_key_block<T, T0, NTB, NFULL>(sK, sV, qf, N, c2, lane, g, li, m_run, l_run, ot);
    attn_key_blocks<T, T0 + NTB, NT16, ATT_BT, NFULL>(sK, sV, qf, N, c2, lane, g, li, m_run, l_run, ot);
  }
}

// bf16: persistent workgroups (two per CU) walk (image, head) pairs; the K / V rows of the NEXT pair are fetched into registers
// (4 + 4 chunks per thread) before the tiles of the current one are computed and written to LDS after them, so the HBM latency
// of the staging -- a third of a head's time when it ran ahead of the tile loop -- hides under the MFMA / exp work.
template <typename T, int NT16, int ATT_BT = 8, int NFULL = -1>
__global__ __launch_bounds__(sizeof(T) == 2 ? 512 : 256, sizeof(T) == 2 ? 4 : 1) void k_attn_fwd(AttnArgs a) {
  typedef Mma<T> MM;
  typedef Geom<T> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT16 * 16;
  constexpr bool PF = sizeof(T) == 2;                    // register prefetch of the next head (512 threads)
  constexpr int CPR = HD / MM::CH;
  constexpr int NCH = PF ? (NP * CPR + 511) / 512 : 1;
  char* sK = smem;
  char* sV = smem + NP * G::ROWB;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, g = lane >> 4, li = lane & 15;
  const size_t ldq = (size_t)3 * a.H * HD;
  const int nbh = a.B * a.H;
  const int nqt = (a.N + 15) / 16;
  const float c2 = a.scale * 1.44269504088896340736f;
  u32x4 pk[NCH], pv[NCH];
  auto pf_load = [&](int bh2) {
    const T* kb2 = reinterpret_cast<const T*>(a.qkv) + (size_t)(bh2 / a.H) * a.N * ldq + (bh2 % a.H) * HD + a.H * HD;
    const T* vb2 = kb2 + a.H * HD;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = i * 512 + (int)threadIdx.x, row = id / CPR, c = id % CPR;
      const bool ok = id < NP * CPR && row < a.N;
      pk[i] = ok ? *reinterpret_cast<const u32x4*>(kb2 + (size_t)row * ldq + c * MM::CH) : z;
      pv[i] = ok ? *reinterpret_cast<const u32x4*>(vb2 + (size_t)row * ldq + c * MM::CH) : z;
    }
  };
  auto pf_store = [&]() {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = i * 512 + (int)threadIdx.x;
      if (id < NP * CPR) {
        *reinterpret_cast<u32x4*>(sK + (id / CPR) * G::ROWB + (id % CPR) * 16) = pk[i];
        *reinterpret_cast<u32x4*>(sV + (id / CPR) * G::ROWB + (id % CPR) * 16) = pv[i];
      }
    }
  };
  int bh = blockIdx.x;
  if (bh >= nbh) return;
  if (PF) {
    pf_load(bh);
    pf_store();
  } else {
    const T* kb = reinterpret_cast<const T*>(a.qkv) + (size_t)(bh / a.H) * a.N * ldq + (bh % a.H) * HD + a.H * HD;
    stage_rows2<T>(sK, kb, ldq, sV, kb + a.H * HD, ldq, a.N, NP);
  }
  __syncthreads();
  for (; bh < nbh; bh += gridDim.x) {
    const int b = bh / a.H, h = bh % a.H;
    const int nxt = bh + (int)gridDim.x;
    if (PF && nxt < nbh) pf_load(nxt);
    const T* qb = reinterpret_cast<const T*>(a.qkv) + (size_t)b * a.N * ldq + h * HD;
    T* ob = reinterpret_cast<T*>(a.o) + (size_t)b * a.N * a.H * HD + h * HD;
    if (a.head_keep && a.head_keep[h] == 0) {               // pruned head (inference): its output slice is zeros
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      for (int i = threadIdx.x; i < a.N * 16; i += blockDim.x) Store4<T>::st(ob + (size_t)(i >> 4) * a.H * HD + (i & 15) * 4, z);
    } else {
      for (int qt = w; qt < nqt; qt += nw) {
        typename MM::Frag qf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[ks] = row_frag_global<T>(qb, ldq, qt * 16 + li, a.N, ks * 4 + g);
        f32x4 ot[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        attn_key_blocks<T, 0, NT16, ATT_BT, NFULL>(sK, sV, qf, a.N, c2, lane, g, li, m_run, l_run, ot);
        const int q = qt * 16 + li;
        if (q < a.N) {
          const float inv = 1.0f / l_run;
          store_tile16<T>(ob + (size_t)q * a.H * HD, g, ot, inv);
          if (g == 0 && a.lse) a.lse[((size_t)b * a.H + h) * a.N + q] = m_run * a.scale + __logf(l_run);
        }
      }
    }
    if (PF && nxt < nbh) {
      __syncthreads();
      pf_store();
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dQ (and delta = rowsum(dO * O)).  K, V in LDS.
template <typename T, int NT16>
__global__ __launch_bounds__(512) void k_attn_bwd_dq(AttnArgs a) {
  typedef Mma<T> MM;
  typedef Geom<T> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT16 * 16;
  char* sK = smem;
  char* sV = smem + NP * G::ROWB;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, g = lane >> 4, li = lane & 15;
  const size_t ldq = (size_t)3 * a.H * HD, ldo = (size_t)a.H * HD;
  const T* qb = reinterpret_cast<const T*>(a.qkv) + (size_t)b * a.N * ldq + h * HD;
  const T* kb = qb + a.H * HD;
  const T* vb = qb + 2 * a.H * HD;
  const T* ob = reinterpret_cast<const T*>(a.o) + (size_t)b * a.N * ldo + h * HD;
  const T* dob = reinterpret_cast<const T*>(a.dout) + (size_t)b * a.N * ldo + h * HD;
  T* dqb = reinterpret_cast<T*>(a.dqkv) + (size_t)b * a.N * ldq + h * HD;
  if (a.head_keep && a.head_keep[h] == 0) {                 // pruned head (see uvc_attn_args.head_keep): dO is exactly zero, so is dq
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < a.N * 16; i += blockDim.x) Store4<T>::st(dqb + (size_t)(i >> 4) * ldq + (i & 15) * 4, z);
    return;
  }
  stage_rows2<T>(sK, kb, ldq, sV, vb, ldq, a.N, NP);
  __syncthreads();
  const int nqt = (a.N + 15) / 16;
  for (int qt = w; qt < nqt; qt += nw) {
    const int q = qt * 16 + li;
    typename MM::Frag qf[G::KS], dof[G::KS];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      qf[ks] = row_frag_global<T>(qb, ldq, q, a.N, ks * 4 + g);
      dof[ks] = row_frag_global<T>(dob, ldo, q, a.N, ks * 4 + g);
      dl += frag_dot<T>(dof[ks], row_frag_global<T>(ob, ldo, q, a.N, ks * 4 + g));
    }
    dl = sum_rows4(dl);
    const float c2 = a.scale * 1.44269504088896340736f;
    const float lse2 = (q < a.N ? a.lse[((size_t)b * a.H + h) * a.N + q] : 0.f) * 1.44269504088896340736f;
    if (q < a.N && g == 0) a.delta[((size_t)b * a.H + h) * a.N + q] = dl;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int s = 0; s < NT16 / G::TPS; ++s) {
      f32x4 ds[G::TPS];
#pragma unroll
      for (int u = 0; u < G::TPS; ++u) {
        const int t = s * G::TPS + u;
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          c = MM::mma(row_frag_lds<T>(sK, t * 16 + li, ks * 4 + g), qf[ks], c);
          dp = MM::mma(row_frag_lds<T>(sV, t * 16 + li, ks * 4 + g), dof[ks], dp);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // padded keys need no mask here: their K rows are zero in LDS, so their (finite) ds only ever multiplies zeros in
          // dq += ds . K (masking cost two compares / selects per score on every tile)
          const float p = __builtin_amdgcn_exp2f(c[e] * c2 - lse2);
          ds[u][e] = p * ((dp[e] - dl) * a.scale);
        }
      }
      const typename MM::Frag dsf = MM::pack(ds[0], ds[G::TPS - 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = MM::mma(TrFrag<T>::ld(sK, G::ROWB, s * MM::KSTEP, dt * 16, lane), dsf, dq[dt]);
    }
    if (q < a.N) store_tile16<T>(dqb + (size_t)q * ldq, g, dq, 1.0f);
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dK and dV.  Q, dO (+ lse, delta) in LDS; each wave owns 16-key tiles.
template <typename T, int NT16>
__global__ __launch_bounds__(512) void k_attn_bwd_dkv(AttnArgs a) {
  typedef Mma<T> MM;
  typedef Geom<T> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NT16 * 16;
  char* sQ = smem;
  char* sDO = smem + NP * G::ROWB;
  float* sLse = reinterpret_cast<float*>(smem + 2 * NP * G::ROWB);
  float* sDel = sLse + NP;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6, g = lane >> 4, li = lane & 15;
  const size_t ldq = (size_t)3 * a.H * HD, ldo = (size_t)a.H * HD;
  const T* qb = reinterpret_cast<const T*>(a.qkv) + (size_t)b * a.N * ldq + h * HD;
  const T* kb = qb + a.H * HD;
  const T* vb = qb + 2 * a.H * HD;
  const T* dob = reinterpret_cast<const T*>(a.dout) + (size_t)b * a.N * ldo + h * HD;
  T* dkb = reinterpret_cast<T*>(a.dqkv) + (size_t)b * a.N * ldq + (a.H + h) * HD;
  T* dvb = dkb + a.H * HD;
  if (a.head_keep && a.head_keep[h] == 0) {                 // pruned head: dk = dv = 0 exactly
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < a.N * 16; i += blockDim.x) {
      Store4<T>::st(dkb + (size_t)(i >> 4) * ldq + (i & 15) * 4, z);
      Store4<T>::st(dvb + (size_t)(i >> 4) * ldq + (i & 15) * 4, z);
    }
    return;
  }
  stage_rows2<T>(sQ, qb, ldq, sDO, dob, ldo, a.N, NP);
  for (int i = threadIdx.x; i < NP; i += blockDim.x) {      // lse pre-multiplied by log2(e); +inf on padded queries -> p = 0
    sLse[i] = i < a.N ? a.lse[((size_t)b * a.H + h) * a.N + i] * 1.44269504088896340736f : INFINITY;
    sDel[i] = i < a.N ? a.delta[((size_t)b * a.H + h) * a.N + i] : 0.f;
  }
  const float c2 = a.scale * 1.44269504088896340736f;
  __syncthreads();
  const int nkt = (a.N + 15) / 16;
  for (int kt = w; kt < nkt; kt += nw) {
    const int key = kt * 16 + li;
    typename MM::Frag kf[G::KS], vf[G::KS];
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      kf[ks] = row_frag_global<T>(kb, ldq, key, a.N, ks * 4 + g);
      vf[ks] = row_frag_global<T>(vb, ldq, key, a.N, ks * 4 + g);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int s = 0; s < NT16 / G::TPS; ++s) {
      f32x4 pp[G::TPS], ds[G::TPS];
#pragma unroll
      for (int u = 0; u < G::TPS; ++u) {
        const int t = s * G::TPS + u;
        f32x4 c = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          c = MM::mma(row_frag_lds<T>(sQ, t * 16 + li, ks * 4 + g), kf[ks], c);
          dp = MM::mma(row_frag_lds<T>(sDO, t * 16 + li, ks * 4 + g), vf[ks], dp);
        }
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + t * 16 + g * 4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + t * 16 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __builtin_amdgcn_exp2f(c[e] * c2 - l4[e]);
          pp[u][e] = p;
          ds[u][e] = p * ((dp[e] - d4[e]) * a.scale);
        }
      }
      const typename MM::Frag pf = MM::pack(pp[0], pp[G::TPS - 1]);
      const typename MM::Frag dsf = MM::pack(ds[0], ds[G::TPS - 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = MM::mma(TrFrag<T>::ld(sDO, G::ROWB, s * MM::KSTEP, dt * 16, lane), pf, dv[dt]);
        dk[dt] = MM::mma(TrFrag<T>::ld(sQ, G::ROWB, s * MM::KSTEP, dt * 16, lane), dsf, dk[dt]);
      }
    }
    if (key < a.N) {
      store_tile16<T>(dkb + (size_t)key * ldq, g, dk, 1.0f);
      store_tile16<T>(dvb + (size_t)key * ldq, g, dv, 1.0f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward as ONE persistent kernel (bf16, round 5): every operand of a head crosses the memory system once (q, k, v, dO, O in, dq, dk, dv
// out = the 8 u of the roofline accounting), S and dP are computed once per (query tile, key tile) and feed all three products
// (5 matmuls where the dq + dk/dv pair above runs 7, one exp2 per score instead of two).
//
// One workgroup per CU walks (image, head) pairs.  NT compute waves + 3 helper waves:
//   * compute wave w owns KEY tile w (K fragments from the LDS image, V fragments straight from global memory, dK^T / dV^T accumulators) and
//     QUERY tile w (dQ^T accumulators).  Step s = 0 .. NT-1 pairs its keys with query tile t = (w + s) % NT:
//         S = Q_t K_w^T, dP = dO_t V_w^T (lane (key, g) holds queries 4g .. 4g+3), P = exp2(S c2 - lse), dS = P (dP - delta) scale,
//         dV_w^T += dO_t^T P, dK_w^T += Q_t^T dS (the 16 queries on the k dimension), dS -> this wave's 512-byte exchange tile [key][query];
//     then it reads the tile wave (w - s) % NT left for query tile w back TRANSPOSED (ds_read_b64_tr_b16) and adds K_p^T dS^T to dQ_w^T.
//     Every query tile receives its NT contributions in step order: deterministic, no atomics.  The hand-over is point to point -- a
//     produced / consumed counter pair per tile in LDS (DS operations of a wave execute in order, so the counter write follows the tile
//     write) -- NOT a workgroup barrier: the r4 form of this kernel spent 30 of its 146 us in 14 barriers per head, all 16 waves in the
//     same phase at the same time.  One barrier per HEAD is left.
//   * the helper waves bring the NEXT head's Q, dO, K images into the other half of the LDS by LDS-DMA (global_load_lds_dwordx4, no
//     registers, no ds_write) while the compute waves run the steps of the current one, and take delta = rowsum(dO * O) and lse * log2(e)
//     for it: the r4 form ran [stage 100 KB | steps | store 75 KB] back to back, 78 + 68 us.  The compute waves' code contains no DMA, so
//     hipcc does not drain vmcnt in front of their LDS reads.
// LDS (NT = 13, N <= 200): 2 x 3 images of 200 rows x 128 bytes (unpadded: chunk c of row r sits at c ^ swz(r), conflict-free for the
// b128 row fragments and the b64 transposing reads alike; the DMA writes lane-linearly, so the swizzle is on its SOURCE address) = 153 600,
// exchange tiles 6 656, lse / delta 2 x 1 664, counters 128: 163 712 of 163 840 bytes.  Tile NT-1 reads 8 rows past its image: whatever
// lies there is finite bf16 (the LDS is zeroed at the start, then only images and dS tiles are written) and meets P = dS = 0: padded
// queries have lse = +inf, padded keys are masked in the last key tile's wave.
namespace one {
// timing probes (tools/attn_probes.sh builds the variants; wrong numbers on purpose; the library is built with 0):
//   1 = no steps (images in, results out)   2 = steps without the counter waits   3 = no images (the helpers fetch nothing)   4 = no result stores
#ifndef UVC_ATTN_PROBE
#define UVC_ATTN_PROBE 0
#endif
constexpr int ROW = 128;
__device__ __forceinline__ int swz(int r) { return ((r >> 1) & 3) << 1; }
typedef short s16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mma16(const s16x4v& a, const s16x4v& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ s16x4v trd(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4v))(p)); }
__device__ __forceinline__ s16x4v pack4(const f32x4& v) {
  u32x2 r; r[0] = pack_bf16x2(v[0], v[1]); r[1] = pack_bf16x2(v[2], v[3]);
  return __builtin_bit_cast(s16x4v, r);
}
// Produced / consumed counters in LDS, accessed by DS instructions written out by hand: through a generic pointer hipcc emits FLAT loads /
// stores (vmcnt, and no ordering against the wave's DS operations); DS operations of one wave execute in order, which is what makes
// "tile write, then counter write" / "counter read, then tile read" a hand-over.  (An asm DS operation the compiler does not count only
// makes its own s_waitcnt lgkmcnt(n) wait for more, never less: the counter retires in order.)
__device__ __forceinline__ int lds_poll(unsigned addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
// (bounded: 2^18 polls are ~20 ms, a hand-over takes < 1 us.  A poll that runs out is a protocol error: the kernel TRAPS -- the launch fails and every later
//  HIP call of the process reports it -- instead of carrying on with a tile that was never handed over and writing plausible wrong dq / dk / dv.  Probe builds,
//  which break the protocol on purpose to time its parts, carry on.)
__device__ __forceinline__ void spin_timeout() {
  if (UVC_ATTN_PROBE == 0) __builtin_trap();
}
__device__ __forceinline__ void spin_ge(unsigned addr, int v) {
  if (UVC_ATTN_PROBE == 2) return;
  int n = 0;
  for (; lds_poll(addr) < v && n < (1 << 18); ++n) __builtin_amdgcn_s_sleep(1);
  if (n == (1 << 18)) spin_timeout();
}
__device__ __forceinline__ void post(unsigned addr, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
constexpr int SPIN_MAX = 1 << 18;
template <int NT, int R> struct Lay {
  static constexpr int IMG = R * ROW, BUF = 3 * IMG;
  static constexpr int OFF_X = 2 * BUF;                         // exchange tiles [NT][512]
  static constexpr int OFF_LD = OFF_X + NT * 512;               // [2 buffers][lse2, delta][NT * 16] float
  static constexpr int OFF_FL = OFF_LD + 2 * 2 * NT * 16 * 4;   // produced[16], consumed[16]
  static constexpr int TOTAL = OFF_FL + 128;
  static_assert(TOTAL <= 163840 && TOTAL % 16 == 0 && R % 8 == 0 && R <= NT * 16 && R + 8 >= NT * 16, "layout");
};
constexpr int NH = 3;                                           // helper waves
// Heads are handed out DYNAMICALLY: workgroup b starts on head b and draws every further head from a device counter, one head ahead of its
// steps (the images of the next head travel under the steps of the current one).  With a static stride a workgroup that reaches its CU late
// -- the weight-gradient stream's one-workgroup-per-CU GEMMs run beside this kernel -- still owed all its heads at the end: the Tiny step
// was 5 % SLOWER with this kernel than with the dq + dk/dv pair although the kernel alone is 20 us faster (profiles/r5e).  A launch owns one
// of 64 counter pairs (the host rotates them; the last workgroup to leave resets the pair), so launches on different streams do not share one.
struct Ticket { unsigned next, done; };
__device__ Ticket g_ticket[64];

template <int NT, int R>
__global__ __launch_bounds__((NT + NH) * 64) void k_attn_bwd_one(AttnArgs a, int slot) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  typedef Lay<NT, R> L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform, and the compiler knows it: tile indices and counters in SGPRs
  const int nbh = a.B * a.H;
  const size_t ldq = (size_t)3 * a.H * HD, ldo = (size_t)a.H * HD;
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid * 16; i < L::TOTAL; i += (NT + NH) * 64 * 16) *reinterpret_cast<u32x4*>(smem + i) = z;
  }
  __syncthreads();
  const unsigned fP = (unsigned)(size_t)(LDS_PTR(char))smem + L::OFF_FL, fC = fP + 64;      // LDS byte addresses of produced[16], consumed[16]
  const unsigned fN = fC + 14 * 4;                                // (two unused counters) the head after this one, + 1: slots [it & 1]
  // the next head of this workgroup, published by the first helper wave at the START of a head, read by everybody else much later (polled: no
  // barrier lies between); heads only grow, so a value above the current head is the new one
  auto next_head = [&](int it, int cur) {
    int v = 0, n = 0;
    for (; n < SPIN_MAX; ++n) {
      v = lds_poll(fN + ((it + 1) & 1) * 4) - 1;
      if (v > cur) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (n == SPIN_MAX) spin_timeout();
    return v;
  };
  float* sLD = reinterpret_cast<float*>(smem + L::OFF_LD);
  if ((int)blockIdx.x >= nbh) return;

  if (w >= NT) {
    // ------------------------------------------------------------------------------ helper waves: the next head's images, lse, delta
    constexpr int NP = R / 8;                                   // 1-KB pieces (8 rows) per image
    constexpr int NPH = (NP + NH - 1) / NH;                     // ... per helper (the last one may be issued twice: no branch around a DMA)
    constexpr int NLS = (NT * 16 + NH * 64 - 1) / (NH * 64);
    const int j = w - NT;
    const int rr = lane >> 3, gc = (lane & 7) ^ swz(rr);
    auto fill = [&](int bh, int buf) {
      const int b = bh / a.H, h = bh % a.H;
      const char* qb = reinterpret_cast<const char*>(a.qkv) + ((size_t)b * a.N * ldq + h * HD) * 2;
      const char* kb = qb + a.H * HD * 2;
      const char* dob = reinterpret_cast<const char*>(a.dout) + ((size_t)b * a.N * ldo + h * HD) * 2;
      const char* ob = reinterpret_cast<const char*>(a.o) + ((size_t)b * a.N * ldo + h * HD) * 2;
      char* img = smem + buf * L::BUF;
      u32x4 ov[NPH];
      float ls[NLS];
      // piece i of this helper is p = j + NH i: rows 8 p + rr.  Only the LAST one can run past the image (p > NP - 1: this helper's
      // FIRST piece is fetched again instead -- no branch around a DMA, and no helper ever writes a piece another one reads results
      // from) or past the sequence (row > N - 1: that row's lanes fetch row N - 1 -- finite, never used);
      // the others are one 32-bit lane offset against a wave-uniform base that advances by NH * 8 rows (SGPR arithmetic).
      const unsigned lq = (unsigned)((8 * j + rr) * ldq * 2 + gc * 16), lo = (unsigned)((8 * j + rr) * ldo * 2 + gc * 16);
      const int pl = j + NH * (NPH - 1) < NP ? j + NH * (NPH - 1) : j, rl = min(8 * pl + rr, a.N - 1);
      const unsigned lql = (unsigned)(rl * ldq * 2 + gc * 16), lol = (unsigned)(rl * ldo * 2 + gc * 16);
#pragma unroll
      for (int i = 0; i < NPH; ++i) {
        const bool last = i == NPH - 1;
        ov[i] = *reinterpret_cast<const u32x4*>(last ? ob + lol : ob + (size_t)i * NH * 8 * ldo * 2 + lo);
      }
#pragma unroll
      for (int i = 0; i < NLS; ++i) {
        const int r = (i * NH + j) * 64 + lane;
        ls[i] = r < a.N ? a.lse[((size_t)b * a.H + h) * a.N + r] * 1.44269504088896340736f : INFINITY;
      }
#pragma unroll
      for (int i = 0; i < NPH; ++i) {
        const bool last = i == NPH - 1;
        const int p = last ? pl : j + NH * i;
        const size_t sq = (size_t)i * NH * 8 * ldq * 2, so = (size_t)i * NH * 8 * ldo * 2;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(last ? qb + lql : qb + sq + lq),
                                         (void __attribute__((address_space(3)))*)(img + p * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(last ? dob + lol : dob + so + lo),
                                         (void __attribute__((address_space(3)))*)(img + L::IMG + p * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(last ? kb + lql : kb + sq + lq),
                                         (void __attribute__((address_space(3)))*)(img + 2 * L::IMG + p * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces have landed (its own dO rows are read next)
      float* sLse = sLD + buf * 2 * NT * 16;
      float* sDel = sLse + NT * 16;
#pragma unroll
      for (int i = 0; i < NPH; ++i) {
        const int p = (i == NPH - 1) ? pl : j + NH * i, row = 8 * p + rr;
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(img + L::IMG + p * 1024 + lane * 16);
        float d = frag_dot<T>(x, __builtin_bit_cast(bf16x8, ov[i]));
        d = dpp_add<0xB1>(d); d = dpp_add<0x4E>(d); d = dpp_add<0x141>(d);      // the 8 lanes of a row
        if ((lane & 7) == 0) {
          sDel[row] = row < a.N ? d * a.scale : 0.f;              // (scaled: ds = p (dp scale - delta scale))
          if (row < a.N && a.delta) a.delta[((size_t)b * a.H + h) * a.N + row] = d;
        }
      }
#pragma unroll
      for (int i = 0; i < NLS; ++i) {
        const int r = (i * NH + j) * 64 + lane;
        if (r < NT * 16) sLse[r] = ls[i];
      }
    };
    auto dead = [&](int bh) { return a.head_keep && a.head_keep[bh % a.H] == 0; };
    // dq, dk, dv of a finished head lie in ITS image buffer (slots of Q, dO, K; same swizzle), left there by the compute waves between the
    // head's two barriers: this helper's pieces leave as 1-KB instructions of 16 bytes per lane, 8 whole rows each -- the 8-byte-per-lane
    // stores from the accumulator layout (16 rows x 32 bytes per instruction) cost the first form of this kernel 57 of its 117 us.
    auto store_out = [&](int bh, int buf) {
      const int b = bh / a.H, h = bh % a.H;
      char* dqb = reinterpret_cast<char*>(a.dqkv) + ((size_t)b * a.N * ldq + h * HD) * 2;
      const char* stg = smem + buf * L::BUF;
      const bool zero = dead(bh);
      // the staged rows have their own swizzle (chunk c of row r at c ^ (r & 7), its 8-byte halves exchanged in rows 8 .. 15 of a tile): the 16
      // rows x 8 bytes of one ds_write_b64 of the accumulator layout then cover all the banks once (under the images' swizzle four times:
      // the staging alone was 3 - 4.7 k of a head's 36 k cycles, profiles/r5c)
      const unsigned lq = (unsigned)((8 * j + rr) * ldq * 2 + ((lane & 7) ^ rr) * 16);
#pragma unroll
      for (int i = 0; i < NPH; ++i) {
        const int p = j + NH * i;
        if (p < NP && 8 * p + rr < a.N) {
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            u32x4 x = {0u, 0u, 0u, 0u};
            if (!zero) x = *reinterpret_cast<const u32x4*>(stg + m * L::IMG + p * 1024 + lane * 16);
            if (p & 1) x = u32x4{x[2], x[3], x[0], x[1]};
            // (non-temporal: 10.96 against 10.94 ms -- dqkv is read again at once, profiles/r5zz_ab_nt_dqkv.txt)
            *reinterpret_cast<u32x4*>(dqb + (size_t)m * a.H * HD * 2 + (size_t)i * NH * 8 * ldq * 2 + lq) = x;
          }
        }
      }
    };
    int bh = blockIdx.x, it = 0, prev = -1;
    if (!dead(bh) && UVC_ATTN_PROBE != 3) fill(bh, 0);
    __syncthreads();
    for (; bh < nbh; ++it) {
      int nxt;
      if (j == 0) {                                             // draw the next head and publish it
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&g_ticket[slot].next, 1u);
        // a launch makes exactly B * H draws (one per head it runs), so a ticket >= B * H means the pair was not zero when the launch started -- a launch before
        // it on this slot died before its last workgroup reset the pair: heads would be skipped and their dq / dk / dv rows left unwritten.  Fail loudly instead.
        if ((int)t >= nbh && UVC_ATTN_PROBE == 0) __builtin_trap();
        nxt = (int)gridDim.x + __builtin_amdgcn_readfirstlane((int)t);
        post(fN + ((it + 1) & 1) * 4, nxt + 1);
      } else {
        nxt = next_head(it, bh);
      }
      if (prev >= 0 && UVC_ATTN_PROBE != 4) store_out(prev, (it + 1) & 1);      // head it - 1 ran on buffer (it - 1) & 1, which the next fill reuses
      if (nxt < nbh && !dead(nxt) && UVC_ATTN_PROBE != 3) fill(nxt, (it + 1) & 1);
      __syncthreads();                                          // A: the steps of head it are done
      __syncthreads();                                          // B: its results are staged
      prev = bh;
      bh = nxt;
    }
    if (UVC_ATTN_PROBE != 4) store_out(prev, (it + 1) & 1);
    if (j == 0 && lane == 0) {                                  // the last workgroup to leave (every draw of the launch has been made) resets the pair
      if (atomicAdd(&g_ticket[slot].done, 1u) == gridDim.x - 1) { g_ticket[slot].next = 0; g_ticket[slot].done = 0; }
    }
    return;
  }

  // -------------------------------------------------------------------------------- compute waves
  const int key = w * 16 + li;
  const float c2 = a.scale * 1.44269504088896340736f;
  const bool last_keys = w * 16 + 16 > a.N;                     // (wave-uniform) this wave's key tile holds padded keys
  const int offA = li * ROW + ((g ^ swz(li)) << 4);             // row fragment: row li of a tile, chunk g (k-step 1: ^ 64)
  const int rT = 4 * g + (li >> 2);
  const int offT = rT * ROW + ((swz(rT) | ((li & 3) >> 1)) << 4) + (li & 1) * 8;   // transposing read, columns 0 .. 15 (dt: ^ (dt << 5))
  const int offXw = li * 32 + g * 8, offXr = rT * 32 + (li & 3) * 8;
  char* sX = smem + L::OFF_X;
  const unsigned voff = (unsigned)(key * ldq * 2 + g * 16);     // this lane's V fragment rows: one 32-bit offset against a wave-uniform base
  auto load_v = [&](int bh, u32x4 (&v)[2]) {
    const char* vb = reinterpret_cast<const char*>(a.qkv) + ((size_t)(bh / a.H) * a.N * ldq + (bh % a.H) * HD + 2 * a.H * HD) * 2;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) v[ks] = key < a.N ? *reinterpret_cast<const u32x4*>(vb + voff + ks * 64) : z;
  };
  u32x4 vnext[2];
  int gstep = 0, it = 0, bh = blockIdx.x;
  load_v(bh, vnext);
  __syncthreads();
  for (; bh < nbh; ++it) {
    const int b = bh / a.H, h = bh % a.H;
    if (a.head_keep && a.head_keep[h] == 0) {                   // pruned head (uvc_attn_args.head_keep): the helpers write its zeros
      const int nxt = next_head(it, bh);
      if (nxt < nbh) load_v(nxt, vnext);
      __syncthreads();
      __syncthreads();
      bh = nxt;
      continue;
    }
    const char* sQ = smem + (it & 1) * L::BUF;
    const char* sDO = sQ + L::IMG;
    const char* sK = sDO + L::IMG;
    const float* sLse = sLD + (it & 1) * 2 * NT * 16;
    const float* sDel = sLse + NT * 16;
    typename MM::Frag kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vf[ks] = __builtin_bit_cast(typename MM::Frag, vnext[ks]);
      kf[ks] = *reinterpret_cast<const typename MM::Frag*>(sK + w * 16 * ROW + (offA ^ (ks * 64)));
    }
    f32x4 dk[4], dv[4], dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; dq[dt] = dk[dt]; }
    auto step = [&](int s) {
      int t = w + s; if (t >= NT) t -= NT;                      // the query tile this wave's keys meet now
      const char* qt = sQ + t * 16 * ROW;
      const char* dt_ = sDO + t * 16 * ROW;
      f32x4 c = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        c = MM::mma(*reinterpret_cast<const typename MM::Frag*>(qt + (offA ^ (ks * 64))), kf[ks], c);
        dp = MM::mma(*reinterpret_cast<const typename MM::Frag*>(dt_ + (offA ^ (ks * 64))), vf[ks], dp);
      }
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + t * 16 + g * 4);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDel + t * 16 + g * 4);
      f32x4 pp, ds;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pr = __builtin_amdgcn_exp2f(c[e] * c2 - l4[e]);
        pp[e] = pr;
        ds[e] = pr * __builtin_fmaf(dp[e], a.scale, -d4[e]);      // d4 = delta * scale
      }
      if (last_keys) {                                           // padded keys: their K rows in LDS are not zeros (a branch: one wave of NT takes it)
        asm volatile("" ::: "memory");
        if (key >= a.N) { pp = f32x4{0.f, 0.f, 0.f, 0.f}; ds = pp; }
      }
      const s16x4v pf = pack4(pp), dsf = pack4(ds);
      spin_ge(fC + w * 4, gstep);                                   // the tile of the step before has been read
      *reinterpret_cast<s16x4v*>(sX + w * 512 + offXw) = dsf;   // [key][query]
      post(fP + w * 4, gstep + 1);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dv[d] = mma16(trd(dt_ + (offT ^ (d << 5))), pf, dv[d]);
        dk[d] = mma16(trd(qt + (offT ^ (d << 5))), dsf, dk[d]);
      }
      int pw = w - s; if (pw < 0) pw += NT;                     // the wave whose keys met query tile w in this step
      spin_ge(fP + pw * 4, gstep + 1);
      const s16x4v dst = trd(sX + pw * 512 + offXr);
      const char* kp = sK + pw * 16 * ROW;
#pragma unroll
      for (int d = 0; d < 4; ++d) dq[d] = mma16(trd(kp + (offT ^ (d << 5))), dst, dq[d]);
      post(fC + pw * 4, gstep + 1);
      ++gstep;
    };
    constexpr int SPLIT = NT >= 3 ? NT - 2 : 0;                 // the next head's V rows are requested under the last two steps
#pragma unroll 1
    for (int s = 0; s < (UVC_ATTN_PROBE == 1 ? 0 : SPLIT); ++s) step(s);
    const int nxt = next_head(it, bh);                          // (published at the start of this head: no wait)
    if (nxt < nbh) load_v(nxt, vnext);
#pragma unroll 1
    for (int s = SPLIT; s < (UVC_ATTN_PROBE == 1 ? 0 : NT); ++s) step(s);
    __syncthreads();                                            // A: every wave is done with this head's images ...
    if (key < a.N) {                                            // ... which now take the results (slot of Q: dq, of dO: dk, of K: dv), for the helpers
      char* stg = smem + (it & 1) * L::BUF + key * ROW + (((g & 1) ^ (li >> 3)) << 3);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int co = ((d * 2 + (g >> 1)) ^ (li & 7)) << 4;
        *reinterpret_cast<s16x4v*>(stg + co) = pack4(dq[d]);    // (query tile w: the same row index)
        *reinterpret_cast<s16x4v*>(stg + L::IMG + co) = pack4(dk[d]);
        *reinterpret_cast<s16x4v*>(stg + 2 * L::IMG + co) = pack4(dv[d]);
      }
    }
    __syncthreads();                                            // B: staged; the next head's images have landed
    bh = nxt;
  }
}
}  // namespace one

// ------------------------------------------------------------------------------------------------
// qkv projection + attention forward as ONE persistent kernel (bf16, D = 192; round 5).  reference: model_distilled.py:175-185 (qkv Linear, softmax(q k^T) v).
// The qkv GEMM wrote [M, 3D] and the attention kernel read it back: 6 u of the 8 u the two kernels move per block (u = M D 2 bytes).  Here a workgroup owns an
// IMAGE: compute wave w keeps the 16 LayerNorm rows of token tile w as MFMA B fragments (24 registers) and, per head, multiplies them with the head's 64 rows of
// Wq, Wk, Wv -- streamed through two 24-KB LDS buffers by the helper waves (LDS-DMA; the weight slices come from L2: the same 9 chunks for every image) -- into
// Q, K, V tiles that go straight into the LDS images the attention phase reads (and, when the backward needs them, to global memory as 16 bytes per lane):
// h in, o (+ lse, + qkv for training) out.  The arithmetic is the unfused pair's, operation for operation (one k-ordered accumulation chain + bias, rounded to
// bf16 once; then attn_key_blocks on the same images): o, lse and qkv are BIT-IDENTICAL to uvc_gemm_nt + uvc_attention_fwd (tests/test_kernels_gpu.py).
namespace qa {
constexpr int NT = 13;                                         // token tiles (N <= 208) = compute waves
constexpr int NHW = 3;                                         // helper waves
constexpr int HG = 3;                                          // heads per work item: a workgroup owns (image, group of 3 heads) -- DeiT-Tiny: the image
constexpr int WCH = 24576;                                     // a weight chunk in LDS: 24 KB = 64 rows at D = 192, 32 rows at D = 384
constexpr int IMG = 224 * Geom<bf16_t>::ROWB;                  // Q / K / V image of a head: 14 tiles of 160-byte rows (rows >= N stay zero)
constexpr int OFF_W = 3 * IMG;
constexpr int TOTAL = OFF_W + 2 * WCH;                         // 156 672
static_assert(TOTAL <= 163840, "LDS");
// conflict-free ds_read_b128 of row fragments (lane li = row, g = 16-byte slot) from unpadded weight rows: 384-byte rows alternate bank halves, so
// slot ^ ((row >> 1) & 7) (k_mlp_fused_p's W1 image); 768-byte rows all start on bank 0, so slot ^ (row & 15)
template <int KS> __device__ __forceinline__ int swzw(int row) { return KS == 6 ? (row >> 1) & 7 : row & 15; }

struct Args {
  const bf16_t* h; const bf16_t* w; const float* bias; bf16_t* qkv; bf16_t* o; float* lse;
  int B, N, H; float scale;
};

// KS = D / 32 (6: DeiT-Tiny; 12: DeiT-Small, T2T-ViT-14)
template <int KS, bool STORE, int NFULL>
__global__ __launch_bounds__((NT + NHW) * 64) void k_qkv_attn_fwd(Args a) {
  typedef bf16_t T;
  typedef Mma<T> MM;
  typedef Geom<T> G;
  constexpr int WROW = KS * 64;                                 // bytes of a weight row
  constexpr int CR = WCH / WROW;                                // rows of a chunk (64 / 32)
  constexpr int NCM = 64 / CR;                                  // chunks per matrix and head (1 / 2)
  constexpr int NJB = CR / 16;                                  // 16-row blocks of a chunk (4 / 2)
  constexpr int NCH = HG * 3 * NCM;                             // chunks per work item
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = a.H * HD, ngrp = a.H / HG;
  const int nitems = a.B * ngrp;
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid * 16; i < OFF_W; i += (NT + NHW) * 64 * 16) *reinterpret_cast<u32x4*>(smem + i) = z;
  }
  __syncthreads();
  const int nmine = (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // work items of this workgroup

  if (w >= NT) {
    // ------------------------------------------------------------------ helper waves: the weight chunks, one ahead of their use
    const int j = w - NT;
    auto issue = [&](int c) {
      const int item = (int)blockIdx.x + (c / NCH) * (int)gridDim.x, cc = c % NCH;
      const int hh = (item % ngrp) * HG + cc / (3 * NCM), m = (cc / NCM) % 3, part = cc % NCM;
      const char* src = reinterpret_cast<const char*>(a.w) + (size_t)(m * D + hh * HD + part * CR) * WROW;
      char* dst = smem + OFF_W + (c & 1) * WCH;
#pragma unroll
      for (int i = 0; i < 24 / NHW; ++i) {
        const int p = j + NHW * i;                               // 1-KB piece: 16-byte slots 64 p .. 64 p + 63 of the chunk
        const int L = p * 64 + lane, row = L / (KS * 4), slot = L % (KS * 4);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + row * WROW + ((slot ^ swzw<KS>(row)) << 4)),
                                         (void __attribute__((address_space(3)))*)(dst + p * 1024), 16, 0, 0);
      }
    };
    const int nchunk = nmine * NCH;
    if (nchunk > 0) issue(0);
    for (int c = 0; c < nchunk; ++c) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // chunk c has landed
      __syncthreads();                                          // X_c: ... and the chunk before it has been used
      if (c + 1 < nchunk) issue(c + 1);
      if (c % (3 * NCM) == 3 * NCM - 1) __syncthreads();        // A: the head's K / V images are complete (the compute waves' barrier)
    }
    return;
  }

  // -------------------------------------------------------------------- compute waves
  const int tok = w * 16 + li;
  const float c2 = a.scale * 1.44269504088896340736f;
  char* sQ = smem; char* sK = smem + IMG; char* sV = smem + 2 * IMG;
  int c = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int img = item / ngrp, h0 = (item % ngrp) * HG;
    const bf16_t* hp = a.h + ((size_t)img * a.N + tok) * D;
    for (int hh = h0; hh < h0 + HG; ++hh) {
      // this wave's 16 LayerNorm rows as B fragments (rows past the sequence: zeros); asked for again per head (L1 / L2): not live under the attention phase
      typename MM::Frag hf[KS];
      {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) hf[ks] = __builtin_bit_cast(typename MM::Frag, tok < a.N ? *reinterpret_cast<const u32x4*>(hp + (ks * 4 + g) * 8) : z);
      }
#pragma unroll 1
      for (int mp = 0; mp < 3 * NCM; ++mp, ++c) {
        const int m = mp / NCM, part = mp % NCM;
        __syncthreads();                                        // X_c
        const char* wb = smem + OFF_W + (c & 1) * WCH;
        f32x4 acc[NJB];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          acc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int row = jb * 16 + li;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
            acc[jb] = MM::mma(*reinterpret_cast<const typename MM::Frag*>(wb + row * WROW + (((ks * 4 + g) ^ swzw<KS>(row)) << 4)), hf[ks], acc[jb]);
        }
        // lane (token li, g) holds outputs d = part * CR + 16 jb + 4 g + e of this matrix and head: + bias, rounded once
        if (a.bias) {
#pragma unroll
          for (int jb = 0; jb < NJB; ++jb) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + m * D + hh * HD + part * CR + jb * 16 + g * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[jb][e] += b4[e];
          }
        }
        if (tok < a.N) {
          char* ip = smem + m * IMG + tok * G::ROWB + part * CR * 2 + g * 8;
#pragma unroll
          for (int jb = 0; jb < NJB; ++jb) {
            u32x2 r; r[0] = pack_bf16x2(acc[jb][0], acc[jb][1]); r[1] = pack_bf16x2(acc[jb][2], acc[jb][3]);
            *reinterpret_cast<u32x2*>(ip + jb * 32) = r;
          }
        }
        if (STORE && part == NCM - 1) {
          // the tile's rows leave from the LDS image (this wave's own writes: DS operations of a wave execute in order) as WHOLE 128-byte rows, 8 lanes a row:
          // from the accumulator layout a store instruction covered 64 bytes of a row, and the PMC counted 1.38 x the kernel's algorithmic bytes
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int r = w * 16 + half * 8 + (lane >> 3);
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + m * IMG + r * G::ROWB + (lane & 7) * 16);
            // (qkv is written for the BACKWARD only -- the forward has used it from LDS: a non-temporal store, 10.82 against 10.90 ms in the step, profiles/r5zz_ab_nt_*)
            if (r < a.N) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.qkv + ((size_t)img * a.N + r) * 3 * D + m * D + hh * HD) + (lane & 7) * 16));
          }
        }
      }
      __syncthreads();                                          // A: every wave's K / V rows are in the images
      // ---- attention of query tile w against the head's keys (k_attn_fwd's tile body)
      {
        typename MM::Frag qf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[ks] = row_frag_lds<T>(sQ, tok, ks * 4 + g);
        f32x4 ot[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        attn_key_blocks<T, 0, 14, 8, NFULL>(sK, sV, qf, a.N, c2, lane, g, li, m_run, l_run, ot);
        {
          // o leaves the same way, through this wave's rows of the Q image (dead since qf was read; only this wave ever reads them)
          const float inv = 1.0f / l_run;
          char* ip = sQ + tok * G::ROWB + g * 8;
          if (tok < a.N) {                                      // (rows past the sequence stay zero)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              u32x2 r; r[0] = pack_bf16x2(ot[dt][0] * inv, ot[dt][1] * inv); r[1] = pack_bf16x2(ot[dt][2] * inv, ot[dt][3] * inv);
              *reinterpret_cast<u32x2*>(ip + dt * 32) = r;
            }
          }
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int r = w * 16 + half * 8 + (lane >> 3);
            const u32x4 v = *reinterpret_cast<const u32x4*>(sQ + r * G::ROWB + (lane & 7) * 16);
            if (r < a.N) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(a.o + ((size_t)img * a.N + r) * D + hh * HD) + (lane & 7) * 16) = v;
          }
          if (tok < a.N && g == 0 && a.lse) a.lse[((size_t)img * a.H + hh) * a.N + tok] = m_run * a.scale + __logf(l_run);
        }
      }
    }
  }
}
}  // namespace qa

template <typename T, int NT16> int launch(const AttnArgs& a, int which, hipStream_t st) {
  const int NP = NT16 * 16;
  size_t sh = (size_t)2 * NP * Geom<T>::ROWB;
  const int grid = a.B * a.H;
  // forward: 4 waves per (image, head); backward: 8 (two workgroups per CU either way -- the LDS image is the limit --
  // so backward runs 4 waves per SIMD, which hides its longer dependent MFMA -> exp -> MFMA chains: 153 -> 135 us)
  const int threads = which == 0 ? (sizeof(T) == 2 ? 512 : 256) : 512;
  if (which == 0) {
    const int fgrid = (sizeof(T) == 2 && grid > 512) ? 512 : grid;      // bf16: two persistent workgroups per CU
    // DeiT's N = 197 / 198: twelve full key tiles, a partial one and a tile of padding -- known at compile time (see attn_key_block)
    constexpr int NF = NT16 == 14 ? 12 : -1;
    if (NF >= 0 && a.N / 16 == NF) {
      UVC_MAX_LDS(sh, k_attn_fwd<T, NT16, 8, NF>);      // sh is a function of the instantiation
      k_attn_fwd<T, NT16, 8, NF><<<fgrid, sizeof(T) == 2 ? 512 : 256, sh, st>>>(a);
    } else {
      UVC_MAX_LDS(sh, k_attn_fwd<T, NT16>);
      k_attn_fwd<T, NT16><<<fgrid, sizeof(T) == 2 ? 512 : 256, sh, st>>>(a);
    }
  } else if (which == 1) {
    UVC_MAX_LDS(sh, k_attn_bwd_dq<T, NT16>);
    k_attn_bwd_dq<T, NT16><<<grid, threads, sh, st>>>(a);
  } else {
    sh += (size_t)2 * NP * sizeof(float);
    UVC_MAX_LDS(sh, k_attn_bwd_dkv<T, NT16>);
    k_attn_bwd_dkv<T, NT16><<<grid, threads, sh, st>>>(a);
  }
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

template <typename T> int dispatch(const AttnArgs& a, int which, hipStream_t st) {
  const int nt = (a.N + 15) / 16;
  if (nt <= 2) return launch<T, 2>(a, which, st);
  if (nt <= 4) return launch<T, 4>(a, which, st);
  if (nt <= 8) return launch<T, 8>(a, which, st);
  if (nt <= 14) return launch<T, 14>(a, which, st);
  if (nt <= 16) return launch<T, 16>(a, which, st);
  return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "attention: sequence length > 256 not supported");
}

int check(const uvc_attn_args* p, bool bwd) {
  if (!p || !p->qkv || !p->o || !p->lse) return uvc_set_error_msg(UVC_ERR_ARG, "attention: null pointer");
  if (bwd && (!p->dout || !p->dqkv || !p->delta)) return uvc_set_error_msg(UVC_ERR_ARG, "attention backward: null pointer");
  if (p->head_dim != HD) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "attention: head_dim must be 64");
  if (p->B <= 0 || p->N <= 0 || p->H <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "attention: empty problem");
  if (p->dtype != UVC_F32 && p->dtype != UVC_BF16) return uvc_set_error_msg(UVC_ERR_ARG, "attention: bad dtype");
  return UVC_OK;
}

// the one-pass backward: bf16, 13 key / query tiles whose images fit 200 rows (DeiT's N = 197 / 198, T2T-ViT's 197)
bool one_pass_supported(const uvc_attn_args* p) { return p->dtype == UVC_BF16 && p->N > 192 && p->N <= 200; }
int launch_one_pass(const AttnArgs& a, int grid_arg, hipStream_t st) {
  typedef one::Lay<13, 200> L;
  static std::atomic<int> ncu_cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return uvc_set_error_msg(UVC_ERR_LAUNCH, "attention backward: hipGetDevice");
  int ncu = dev < 64 ? ncu_cache[dev].load(std::memory_order_relaxed) : 0;
  if (ncu == 0) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    if (dev < 64) ncu_cache[dev].store(ncu, std::memory_order_relaxed);
  }
  const int nbh = a.B * a.H;
  // A workgroup of this kernel takes a whole CU (160 KB of LDS, all 512 registers of every SIMD), and lives as long as the launch: at one
  // workgroup per CU nothing of the weight-gradient stream runs beside it -- and this latency-bound kernel is the one place of the
  // backward where an HBM-bound GEMM beside it costs nothing.  Measured in the step (profiles/r5f_gridexp*.txt, same box): DeiT-Tiny
  // 12.70 ms with 256 workgroups, 12.62 with 232, 11.98 with 224 (the dq + dk/dv pair: 12.20), 12.02 with 200: one free CU in EVERY
  // shader engine (256 CUs = 32 engines x 8) is what the other stream's dispatch needs -- with one engine full it stalls as if all were.
  // DeiT-Small (H = 6) is 2 % faster with the whole chip, DeiT-Base indifferent: the narrow model alone leaves the CUs.
  // (measured again with one event per block and the weight-gradient stream as the backward's critical path: 256 / 240 / 224 / 208 / 192 / 176 / 160 workgroups
  //  +0.09 / +0.11 / 0 / +0.06 / +0.18 / +0.19 / +0.23 ms, profiles/r5end_ab_attn_bwd_cus.txt)
  const int ncu_use = a.H <= 3 && ncu >= 64 ? ncu - ncu / 8 : ncu;
  const int grid = grid_arg > 0 ? (grid_arg < nbh ? grid_arg : nbh) : (nbh < ncu_use ? nbh : ncu_use);
  static std::atomic<unsigned> seq{0};
  UVC_MAX_LDS(L::TOTAL, one::k_attn_bwd_one<13, 200>);
  one::k_attn_bwd_one<13, 200><<<grid, (13 + one::NH) * 64, L::TOTAL, st>>>(a, (int)(seq.fetch_add(1, std::memory_order_relaxed) & 63));
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

AttnArgs conv(const uvc_attn_args* p) {
  AttnArgs a;
  a.qkv = p->qkv; a.o = p->o; a.lse = p->lse; a.dout = p->dout; a.dqkv = p->dqkv; a.delta = p->delta;
  a.B = p->B; a.N = p->N; a.H = p->H; a.scale = p->scale; a.head_keep = p->head_keep;
  return a;
}

}  // namespace

extern "C" int uvc_qkv_attention_supported(int32_t B, int32_t N, int32_t H, int32_t D, int32_t dtype) {
  return dtype == UVC_BF16 && (D == 192 || D == 384) && H * 64 == D && N > 192 && N <= 208 && B >= 1;
}

extern "C" int uvc_qkv_attention_fwd(const uvc_qkv_attn_args* p, void* stream) {
  if (!p || !p->h || !p->w || !p->o) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_qkv_attention_fwd: null pointer");
  if (!uvc_qkv_attention_supported(p->B, p->N, p->H, p->D, p->dtype)) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_qkv_attention_fwd: bf16, D = 64 H = 192 or 384, 193 <= N <= 208");
  if ((((uintptr_t)p->h | (uintptr_t)p->w | (uintptr_t)p->o | (uintptr_t)p->qkv | (uintptr_t)p->bias) & 15) != 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_qkv_attention_fwd: 16-byte alignment");
  qa::Args a;
  a.h = (const bf16_t*)p->h; a.w = (const bf16_t*)p->w; a.bias = p->bias; a.qkv = (bf16_t*)p->qkv; a.o = (bf16_t*)p->o; a.lse = p->lse;
  a.B = p->B; a.N = p->N; a.H = p->H; a.scale = p->scale;
  int ncu = 256;
  { int dev = 0; if (hipGetDevice(&dev) == hipSuccess) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n; } }
  const int nitems = p->B * (p->H / qa::HG);
  const int grid = p->grid > 0 ? (p->grid < nitems ? p->grid : nitems) : (nitems < ncu ? nitems : ncu);
  hipStream_t st = (hipStream_t)stream;
  const bool nf12 = p->N / 16 == 12;
#define QA_LAUNCH(KS_, ST_, NF_) { UVC_MAX_LDS(qa::TOTAL, qa::k_qkv_attn_fwd<KS_, ST_, NF_>); qa::k_qkv_attn_fwd<KS_, ST_, NF_><<<grid, (qa::NT + qa::NHW) * 64, qa::TOTAL, st>>>(a); }
#define QA_KS(KS_) { if (p->qkv) { if (nf12) QA_LAUNCH(KS_, true, 12) else QA_LAUNCH(KS_, true, -1) } else { if (nf12) QA_LAUNCH(KS_, false, 12) else QA_LAUNCH(KS_, false, -1) } }
  if (p->D == 192) QA_KS(6) else QA_KS(12)
#undef QA_KS
#undef QA_LAUNCH
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_attention_fwd(const uvc_attn_args* p, void* stream) {
  if (int e = check(p, false)) return e;
  const AttnArgs a = conv(p);
  return p->dtype == UVC_F32 ? dispatch<float>(a, 0, (hipStream_t)stream) : dispatch<bf16_t>(a, 0, (hipStream_t)stream);
}

extern "C" int uvc_attention_bwd(const uvc_attn_args* p, void* stream) {
  if (int e = check(p, true)) return e;
  const AttnArgs a = conv(p);
  hipStream_t st = (hipStream_t)stream;
  if (p->dtype == UVC_F32) {
    if (p->variant == 2) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "attention backward: the one-pass kernel is bf16 only");
    if (int e = dispatch<float>(a, 1, st)) return e;
    return dispatch<float>(a, 2, st);
  }
  if (p->variant < 0 || p->variant > 2) return uvc_set_error_msg(UVC_ERR_ARG, "attention backward: bad variant");
  if (p->variant == 2 && !one_pass_supported(p)) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "attention backward: the one-pass kernel takes 193 <= N <= 200");
#ifdef UVC_ATTN_BWD_PAIR_DEFAULT                       // A/B builds (tools/exp_ab.sh): variant 0 = the pair
  if (p->variant == 2) return launch_one_pass(a, p->grid, st);
#else
  // variant 0: the one-pass kernel where a persistent workgroup gets at least ~4 heads (its prologue, and the CUs it keeps from the other stream, have to
  // pay: T2T-ViT-14 at batch 128 -- 768 heads on 256 CUs -- is 2.8 % faster in the step with the pair, profiles/r5h_ab_attn_t2t.txt); 1024 heads stand
  // for 4 x 256 CUs: the choice must not depend on the device the call happens to run on
  if (p->variant == 2 || (p->variant == 0 && one_pass_supported(p) && (int64_t)p->B * p->H >= 1024)) return launch_one_pass(a, p->grid, st);
#endif
  if (int e = dispatch<bf16_t>(a, 1, st)) return e;
  return dispatch<bf16_t>(a, 2, st);
}
