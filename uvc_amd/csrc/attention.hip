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
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            f32x4 v = ot[dt];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= inv;
            Store4<T>::st(ob + (size_t)q * a.H * HD + dt * 16 + g * 4, v);
          }
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
    if (q < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) Store4<T>::st(dqb + (size_t)q * ldq + dt * 16 + g * 4, dq[dt]);
    }
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
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        Store4<T>::st(dkb + (size_t)key * ldq + dt * 16 + g * 4, dk[dt]);
        Store4<T>::st(dvb + (size_t)key * ldq + dt * 16 + g * 4, dv[dt]);
      }
    }
  }
}

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

AttnArgs conv(const uvc_attn_args* p) {
  AttnArgs a;
  a.qkv = p->qkv; a.o = p->o; a.lse = p->lse; a.dout = p->dout; a.dqkv = p->dqkv; a.delta = p->delta;
  a.B = p->B; a.N = p->N; a.H = p->H; a.scale = p->scale; a.head_keep = p->head_keep;
  return a;
}

}  // namespace

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
    if (int e = dispatch<float>(a, 1, st)) return e;
    return dispatch<float>(a, 2, st);
  }
  if (int e = dispatch<bf16_t>(a, 1, st)) return e;
  return dispatch<bf16_t>(a, 2, st);
}
