// UVC primal-dual engine kernels for gfx950 (HBM-bound streaming + tiny scalar state).
// Reference: UVC/uvc_utils.py, UVC/uvc_optimizer.py (cited per kernel).  Compiled with
// -ffp-contract=off: the scalar update must round like the reference's separate float32 ops.
#include "common.h"
#include "../../include/uvc_engine.h"
#include <stdio.h>
#include <string.h>

// ---------------------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
int uvc_set_error(hipError_t e, const char* file, int line) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
  return UVC_ERR_LAUNCH;
}
int uvc_set_error_msg(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
extern "C" const char* uvc_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------- scores
// One block = 64 consecutive columns x all rows of one weight matrix; 256 threads = 64 columns x
// 4 row slices.  Lanes of a wave read 64 consecutive floats of a row (256 B, coalesced).
// Algorithmic bytes: 4*L*(D*D + D*F) read once.
// MODE 0: plain scores (uvc_utils.py:54-73).
// MODE 1: proximal shrink of the selected columns then scores of the result (uvc_utils.py:315-345).
template <int MODE>
__global__ __launch_bounds__(256) void k_scores(float* const* __restrict__ W1, float* const* __restrict__ W3,
                                                uvc_dims d, double* __restrict__ ws64,
                                                float* __restrict__ sc1, float* __restrict__ sc3,
                                                const int32_t* __restrict__ rank1, const int32_t* __restrict__ rankh,
                                                const int32_t* __restrict__ rank3, const float* __restrict__ s,
                                                const float* __restrict__ r, const float* __restrict__ y,
                                                const float* __restrict__ p, double lr) {
  const int l = blockIdx.y;
  const int nchunk1 = d.D / 64;
  const bool isW1 = (int)blockIdx.x < nchunk1;
  const int ncols = isW1 ? d.D : d.F;
  const int c = (isW1 ? blockIdx.x : blockIdx.x - nchunk1) * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  float* W = isW1 ? W1[l] : W3[l];
  const int rows = d.D;
  float div1 = 1.0f, div2 = 1.0f;
  bool sel1 = false, sel2 = false;
  if (MODE == 1) {
    if (isW1) {
      const int h = c / d.hd;
      const int kr = (int)ceilf(r[l * d.H + h]);
      const int ks = (int)ceilf(s[l * 2 + 0]);
      sel1 = rank1[l * d.D + c] < kr;                       // first-level projection (:325-330)
      sel2 = rankh[l * d.H + h] < ks;                       // second-level, pre-prox scores2 (:333-337)
      div1 = (float)(1.0 + 2.0 * lr * (double)p[l * d.H + h]);
      div2 = (float)(1.0 + 2.0 * lr * (double)y[l * 2 + 0]);
    } else {
      const int ks = (int)ceilf(s[l * 2 + 1]);
      sel1 = rank3[l * d.F + c] < ks;                       // :340-345
      div1 = (float)(1.0 + 2.0 * lr * (double)y[l * 2 + 1]);
    }
  }
  double acc = 0.0;
  for (int row = slice; row < rows; row += 4) {
    float w = W[(size_t)row * ncols + c];
    if (MODE == 1) {
      if (sel1) w = w / div1;
      if (sel2) w = w / div2;
      if (sel1 || sel2) W[(size_t)row * ncols + c] = w;
    }
    acc += (double)w * (double)w;
  }
  __shared__ double part[4][64];
  part[slice][threadIdx.x & 63] = acc;
  __syncthreads();
  if (slice == 0) {
    const int t = threadIdx.x;
    const double tot = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
    if (isW1) {
      ws64[(size_t)l * (d.D + d.F) + c] = tot;
      sc1[l * d.D + c] = (float)tot;
    } else {
      sc3[l * d.F + c] = (float)tot;
    }
  }
}

// scores2[l,h] = float32( sum of the head's float64 column sums )
__global__ void k_head_scores(const double* __restrict__ ws64, uvc_dims d, float* __restrict__ sc2) {
  const int l = blockIdx.x, h = threadIdx.x;
  if (h >= d.H) return;
  double a = 0.0;
  for (int j = 0; j < d.hd; ++j) a += ws64[(size_t)l * (d.D + d.F) + h * d.hd + j];
  sc2[l * d.H + h] = (float)a;
}

// ---------------------------------------------------------------------------- ranks
// rank[i] = #{ j : v[j] < v[i]  or (v[j] == v[i] and j < i) }.  Block (c, l): 256 of layer l's F fc2-column scores against all F
// (staged in LDS, F <= 8192; every lane of a wave reads the same sh[j]: a broadcast); block (0, l) also ranks the D proj columns inside
// their heads and the H heads.  (One block per layer, as in rounds 1-3, was 12 blocks on 256 CUs: 372 us per launch at DeiT-Base's F = 3072.)
__global__ __launch_bounds__(256) void k_rank(const float* __restrict__ sc1, const float* __restrict__ sc2,
                                              const float* __restrict__ sc3, uvc_dims d,
                                              int32_t* __restrict__ rank1, int32_t* __restrict__ rankh,
                                              int32_t* __restrict__ rank3) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int l = blockIdx.y;
  // W3 columns
  for (int i = threadIdx.x; i < d.F; i += blockDim.x) sh[i] = sc3[l * d.F + i];
  __syncthreads();
  {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.F) {
      const float v = sh[i];
      int rk = 0;
      int j = 0;
      for (; j + 4 <= d.F; j += 4) {                  // four scores per (broadcast) LDS read
        const float4 u = *reinterpret_cast<const float4*>(sh + j);
        rk += (u.x < v) || (u.x == v && j < i);
        rk += (u.y < v) || (u.y == v && j + 1 < i);
        rk += (u.z < v) || (u.z == v && j + 2 < i);
        rk += (u.w < v) || (u.w == v && j + 3 < i);
      }
      for (; j < d.F; ++j) {
        const float u = sh[j];
        rk += (u < v) || (u == v && j < i);
      }
      rank3[l * d.F + i] = rk;
    }
  }
  if (blockIdx.x != 0) return;
  __syncthreads();
  // W1 columns inside each head
  for (int i = threadIdx.x; i < d.D; i += blockDim.x) sh[i] = sc1[l * d.D + i];
  __syncthreads();
  for (int i = threadIdx.x; i < d.D; i += blockDim.x) {
    const int h0 = (i / d.hd) * d.hd;
    const float v = sh[i];
    int rk = 0;
    for (int j = h0; j < h0 + d.hd; ++j) {
      const float u = sh[j];
      rk += (u < v) || (u == v && j < i);
    }
    rank1[l * d.D + i] = rk;
  }
  if ((int)threadIdx.x < d.H) {
    const int i = threadIdx.x;
    const float v = sc2[l * d.H + i];
    int rk = 0;
    for (int j = 0; j < d.H; ++j) {
      const float u = sc2[l * d.H + j];
      rk += (u < v) || (u == v && j < i);
    }
    rankh[l * d.H + i] = rk;
  }
}

// ---------------------------------------------------------------------------- masks
__global__ __launch_bounds__(256) void k_masks(float* const* __restrict__ mproj, float* const* __restrict__ mfc2,
                                               float* const* __restrict__ mfc1, uvc_dims d,
                                               const int32_t* __restrict__ rank1, const int32_t* __restrict__ rankh,
                                               const int32_t* __restrict__ rank3, const float* __restrict__ s,
                                               const float* __restrict__ r) {
  const int l = blockIdx.y;
  const int nchunk1 = d.D / 64;
  const bool isW1 = (int)blockIdx.x < nchunk1;
  const int c = (isW1 ? blockIdx.x : blockIdx.x - nchunk1) * 64 + (threadIdx.x & 63);
  const int slice = threadIdx.x >> 6;
  if (isW1) {
    const int h = c / d.hd;
    const bool pruned = (rank1[l * d.D + c] < (int)ceilf(r[l * d.H + h])) ||
                        (rankh[l * d.H + h] < (int)ceilf(s[l * 2 + 0]));
    const float v = pruned ? 0.0f : 1.0f;
    float* M = mproj[l];
    for (int row = slice; row < d.D; row += 4) M[(size_t)row * d.D + c] = v;
  } else {
    const bool pruned = rank3[l * d.F + c] < (int)ceilf(s[l * 2 + 1]);
    const float v = pruned ? 0.0f : 1.0f;
    float* M2 = mfc2[l];
    for (int row = slice; row < d.D; row += 4) M2[(size_t)row * d.F + c] = v;
    // fc1 rows follow fc2 columns (uvc_utils.py:401) -- but the reference resets only the W1 / W3 masks to 1 (:382,393) and
    // only ever writes ZEROS into the fc1 mask: it is the sticky union of every pruned set seen so far
    // (tests/golden/mask_sticky_micro.npz).  A released column therefore leaves its fc1 row masked.
    if (pruned) {
      float* M1 = mfc1[l] + (size_t)c * d.D;
      for (int k = slice; k < d.D; k += 4) M1[k] = 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------- scalar update
#define UVC_MAX_L 32
#define UVC_MAX_H 32

struct GroupStat {   // least-k statistics of one score group
  float next;        // (k+1)-th smallest (or max if k+1 > n)  -- LeastSsum backward factor
  float ksum;        // sum of the k smallest (float64 accumulate) -- get_least_*_norm
};

// one WAVE computes the least-k statistics of one score group (no block barriers): result valid in every lane
__device__ GroupStat group_stat(const float* __restrict__ sc, const int32_t* __restrict__ rk, int n, int k) {
  const int lane = threadIdx.x & 63;
  double part = 0.0;
  float nxt = -1.0f;                               // scores are sums of squares (>= 0): -1 is a safe "not mine"
  const int want = (k + 1 <= n) ? k : n - 1;     // rank of the backward factor
  for (int i = lane; i < n; i += 64) {
    const int rr = rk[i];
    if (rr < k) part += (double)sc[i];
    if (rr == want) nxt = sc[i];
  }
  GroupStat g;
  g.ksum = (float)wave_sum_f64(part);              // butterfly: same order on every run
  g.next = wave_max(nxt);
  return g;
}

struct ResourceOut { float R; };

// calc_flops with full_model_flops set (uvc_utils.py:409-462); executed by thread 0 only.
__device__ float resource_eval(const uvc_state& st, uvc_dims d, const uvc_hyper& hp, const float* e, bool hard,
                               float* gs2, float* gr2, float* gg2, bool want_grad) {
  const int L = d.L, H = d.H;
  const float Dsum = (float)H * (float)d.hd;       // r_ub.sum(1)
  float t[6] = {0, 0, 0, 0, 0, 0};
  const float c = 2.0f / st.resource_ub;
  for (int l = 0; l < L; ++l) {
    const float cs0 = ceilf(st.s[l * 2 + 0]), cs1 = ceilf(st.s[l * 2 + 1]);
    const float ub0 = (float)H, ub1 = (float)d.F;
    const float sraw0 = (ub0 - cs0) / ub0, sraw1 = (ub1 - cs1) / ub1;
    const float sr0 = fminf(fmaxf(sraw0, 0.0f), 1.0f), sr1 = fminf(fmaxf(sraw1, 0.0f), 1.0f);
    float ap = Dsum;
    ap -= cs0 * (float)d.hd;
    const int ks = (int)cs0;
    for (int h = 0; h < H; ++h)
      if (!(st.rankh[l * H + h] < ks)) ap -= ceilf(st.r[l * H + h]);
    const float rraw = ap / Dsum;
    const float rr = fminf(fmaxf(rraw, 0.0f), 1.0f);
    float d1 = 1.0f, dd0 = 0.0f, dd1 = 0.0f;
    if (st.gate != nullptr && hp.enable_block_gating) {
      const float g0 = st.gate[l * 2 + 0], g1 = st.gate[l * 2 + 1];
      if (hp.use_gumbel) {
        const float u0 = (g0 + (-logf(e[l * 2 + 0]))) / 0.5f, u1 = (g1 + (-logf(e[l * 2 + 1]))) / 0.5f;
        const float m = fmaxf(u0, u1);
        const float e0 = expf(u0 - m), e1 = expf(u1 - m);
        const float y0 = e0 / (e0 + e1), y1 = e1 / (e0 + e1);
        d1 = hard ? ((y1 > y0) ? 1.0f : 0.0f) : y1;
        dd0 = -y0 * y1 / 0.5f;
        dd1 = y0 * y1 / 0.5f;
      } else {
        const float tmp = g1 * g1;
        d1 = tmp / (tmp + hp.eps);
        dd1 = 2.0f * g1 * hp.eps / ((tmp + hp.eps) * (tmp + hp.eps));
      }
    }
    const float* tm = st.total_macs + l * 6;
    const float m0 = tm[0] * d1, m1 = tm[1] * d1, m2 = tm[2] * d1, m3 = tm[3] * d1, m4 = tm[4] * d1, m5 = tm[5] * d1;
    t[0] += m0 * sr0; t[1] += m1 * sr0; t[2] += m2 * rr; t[3] += m3 * rr; t[4] += m4 * sr1; t[5] += m5 * sr1;
    if (want_grad) {
      const float in0 = (sraw0 >= 0.0f && sraw0 <= 1.0f) ? 1.0f : 0.0f;
      const float in1 = (sraw1 >= 0.0f && sraw1 <= 1.0f) ? 1.0f : 0.0f;
      const float inr = (rraw >= 0.0f && rraw <= 1.0f) ? 1.0f : 0.0f;
      const float dsr0 = c * (m0 + m1), dsr1 = c * (m4 + m5), drr = c * (m2 + m3);
      gs2[l * 2 + 0] = dsr0 * in0 * (-1.0f / ub0) + drr * inr * (-(float)d.hd / Dsum);
      gs2[l * 2 + 1] = dsr1 * in1 * (-1.0f / ub1);
      for (int h = 0; h < H; ++h)
        gr2[l * H + h] = (st.rankh[l * H + h] < ks) ? 0.0f : drr * inr * (-1.0f / Dsum);
      const float A = tm[0] * sr0 + tm[1] * sr0 + tm[2] * rr + tm[3] * rr + tm[4] * sr1 + tm[5] * sr1;
      gg2[l * 2 + 0] = (c * A) * dd0;
      gg2[l * 2 + 1] = (c * A) * dd1;
    }
  }
  float macs = st.embed_macs + t[0];
  macs = macs + t[1]; macs = macs + t[2]; macs = macs + t[3]; macs = macs + t[4]; macs = macs + t[5];
  return macs * 2.0f / st.resource_ub;
}

__global__ __launch_bounds__(256) void k_resource(uvc_state st, uvc_dims d, uvc_hyper hp, const float* e, int hard,
                                                  float* out) {
  if (threadIdx.x == 0) out[0] = resource_eval(st, d, hp, e, hard != 0, nullptr, nullptr, nullptr, false);
}

__global__ __launch_bounds__(1024) void k_dual_step(uvc_state gst, uvc_dims d, uvc_hyper hp, const float* ge1,
                                                   const float* ge2, int enable_warmup, int global_step) {
  // The scalar update is a few hundred dependent float ops run by one lane: every global access would be a
  // full memory round trip, so the whole state is staged in LDS first and written back at the end.
  __shared__ float l_s[UVC_MAX_L * 2], l_r[UVC_MAX_L * UVC_MAX_H], l_y[UVC_MAX_L * 2], l_p[UVC_MAX_L * UVC_MAX_H], l_z[1];
  __shared__ float l_gate[UVC_MAX_L * 2], l_ggrad[UVC_MAX_L * 2], l_gmom[UVC_MAX_L * 2], l_gsum[UVC_MAX_L * 2];
  __shared__ float l_macs[UVC_MAX_L * 6], l_e1[UVC_MAX_L * 2], l_e2[UVC_MAX_L * 2], l_out[4];
  __shared__ int32_t l_rankh[UVC_MAX_L * UVC_MAX_H], l_cnt[2];
  const bool gating = gst.gate != nullptr && hp.enable_block_gating;
  {
    const int L = d.L, H = d.H, t = threadIdx.x;
    for (int i = t; i < L * 2; i += blockDim.x) {
      l_s[i] = gst.s[i]; l_y[i] = gst.y[i];
      if (gating) { l_gate[i] = gst.gate[i]; l_gmom[i] = gst.gate_momentum ? gst.gate_momentum[i] : 0.f; l_gsum[i] = gst.gate_gsum ? gst.gate_gsum[i] : 0.f;
                    l_ggrad[i] = gst.gate_grad ? gst.gate_grad[i] : 0.f; }
      l_e1[i] = ge1 ? ge1[i] : 1.f; l_e2[i] = ge2 ? ge2[i] : 1.f;
    }
    for (int i = t; i < L * H; i += blockDim.x) { l_r[i] = gst.r[i]; l_p[i] = gst.p[i]; l_rankh[i] = gst.rankh[i]; }
    for (int i = t; i < L * 6; i += blockDim.x) l_macs[i] = gst.total_macs[i];
    if (t == 0) { l_z[0] = gst.z[0]; l_cnt[0] = gst.gate_counters ? gst.gate_counters[0] : 0; l_cnt[1] = gst.gate_counters ? gst.gate_counters[1] : 0; }
    if (t < 4) l_out[t] = gst.out[t];
  }
  __syncthreads();
  uvc_state st = gst;
  st.s = l_s; st.r = l_r; st.y = l_y; st.p = l_p; st.z = l_z; st.total_macs = l_macs; st.rankh = l_rankh; st.out = l_out;
  st.gate = gating ? l_gate : nullptr; st.gate_grad = l_ggrad; st.gate_momentum = l_gmom; st.gate_gsum = l_gsum; st.gate_counters = l_cnt;
  const float* e1 = l_e1;
  const float* e2 = l_e2;
  auto write_back = [&]() {
    __syncthreads();
    const int L = d.L, H = d.H, t = threadIdx.x;
    for (int i = t; i < L * 2; i += blockDim.x) {
      gst.s[i] = l_s[i]; gst.y[i] = l_y[i];
      if (gating && !enable_warmup) { gst.gate[i] = l_gate[i]; gst.gate_momentum[i] = l_gmom[i]; gst.gate_gsum[i] = l_gsum[i]; }
    }
    for (int i = t; i < L * H; i += blockDim.x) { gst.r[i] = l_r[i]; gst.p[i] = l_p[i]; }
    if (t == 0) { gst.z[0] = l_z[0]; if (gating && !enable_warmup) { gst.gate_counters[0] = l_cnt[0]; gst.gate_counters[1] = l_cnt[1]; } }
    if (t < 4) gst.out[t] = l_out[t];
  };
  __shared__ float nx_s[UVC_MAX_L * 2], nx_r[UVC_MAX_L * UVC_MAX_H];
  __shared__ float ks_s[UVC_MAX_L * 2], ks_r[UVC_MAX_L * UVC_MAX_H];
  __shared__ float gs2[UVC_MAX_L * 2], gr2[UVC_MAX_L * UVC_MAX_H], gg2[UVC_MAX_L * 2];
  __shared__ int go_on;
  const int L = d.L, H = d.H;
  // ---- phase 1: backward factors of sloss1 / rloss1 at k = ceil(s), ceil(r) (uvc_utils.py:177-217); one wave per layer
  const int wv = threadIdx.x >> 6, nwv = blockDim.x >> 6, ln0 = threadIdx.x & 63;
  for (int l = wv; l < L; l += nwv) {
    GroupStat a = group_stat(st.scores2 + l * H, st.rankh + l * H, H, (int)ceilf(st.s[l * 2 + 0]));
    GroupStat b = group_stat(st.scores3 + l * d.F, st.rank3 + l * d.F, d.F, (int)ceilf(st.s[l * 2 + 1]));
    if (ln0 == 0) { nx_s[l * 2 + 0] = a.next; nx_s[l * 2 + 1] = b.next; }
    for (int h = 0; h < H; ++h) {
      GroupStat c = group_stat(st.scores1 + l * d.D + h * d.hd, st.rank1 + l * d.D + h * d.hd, d.hd, (int)ceilf(st.r[l * H + h]));
      if (ln0 == 0) nx_r[l * H + h] = c.next;
    }
  }
  __syncthreads();
  // ---- phase 2 (thread 0): resource sample #1, primal + gating updates (uvc_optimizer.py:48-123)
  if (threadIdx.x == 0) {
    const float R = resource_eval(st, d, hp, e1, false, gs2, gr2, gg2, true);
    const float diff = R - hp.budget;
    st.out[0] = diff + hp.budget;                                             // cur_resource (:49)
    const float inside = (diff >= -hp.z_grad_clip && diff <= hp.z_grad_clip) ? 1.0f : 0.0f;   // clamp (:50)
    go_on = enable_warmup ? 0 : 1;
    if (!enable_warmup) {
      const float z = st.z[0];
      // gating (:89-98)
      if (st.gate != nullptr && hp.enable_block_gating) {
        const float w = (float)(global_step % hp.gating_interval);
        for (int i = 0; i < L * 2; ++i) {
          const float gg = st.gate_grad[i] + z * hp.gating_weight * (gg2[i] * inside);
          st.gate_gsum[i] = st.gate_gsum[i] + gg * w;
        }
        st.gate_counters[0] += 1;
        if ((global_step + 1) % hp.gating_interval == 0) {
          const float n = (float)st.gate_counters[0];
          for (int i = 0; i < L * 2; ++i) {
            const float grad = st.gate_gsum[i] / n;
            const float dp = grad + 1e-4f * st.gate[i];
            float buf = st.gate_counters[1] ? (st.gate_momentum[i] * 0.9f + dp) : dp;
            st.gate_momentum[i] = buf;
            st.gate[i] = st.gate[i] + (-hp.glr) * buf;
            st.gate_gsum[i] = 0.0f;
          }
          st.gate_counters[0] = 0;
          st.gate_counters[1] = 1;
        }
      }
      // s (:63-66,83-84,100-110)
      float mx = 0.0f;
      float grad_s[UVC_MAX_L * 2];
      bool over_s[UVC_MAX_L * 2];
      for (int l = 0; l < L; ++l)
        for (int j = 0; j < 2; ++j) {
          const int i = l * 2 + j;
          const float ub = j == 0 ? (float)H : (float)d.F;
          const float smax = fmaxf(ub - 1.0f - 1e-8f, 0.0f);
          float g1 = st.y[i] * nx_s[i] + hp.sl2wd * (st.s[i] / ub);
          float g = g1 + z * (gs2[i] * inside);
          const bool over = st.s[i] >= smax, under = st.s[i] <= 0.0f;
          if (over) g = fmaxf(g, 0.0f);
          if (under) g = fminf(g, 0.0f);
          over_s[i] = over;
          grad_s[i] = g;
          mx = fmaxf(mx, fabsf(g));
        }
      float coef = fminf(1.0f / (mx + 1e-6f), 1.0f);
      st.out[2] = mx;
      for (int l = 0; l < L; ++l)
        for (int j = 0; j < 2; ++j) {
          const int i = l * 2 + j;
          const float ub = j == 0 ? (float)H : (float)d.F;
          const float smax = fmaxf(ub - 1.0f - 1e-8f, 0.0f);
          float v = st.s[i] + (-hp.slr) * (grad_s[i] * coef);
          v = fmaxf(v, 0.0f);
          if (over_s[i]) v = smax;
          st.s[i] = v;
        }
      // r (:66,86-87,113-123)
      mx = 0.0f;
      const float rmax = fmaxf((float)d.hd - 1.0f - 1e-8f, 0.0f);
      for (int i = 0; i < L * H; ++i) {
        float g1 = st.p[i] * nx_r[i] + hp.sl2wd * (st.r[i] / (float)d.hd);
        float g = g1 + z * (gr2[i] * inside);
        const bool over = st.r[i] >= rmax, under = st.r[i] <= 0.0f;
        if (over) g = fmaxf(g, 0.0f);
        if (under) g = fminf(g, 0.0f);
        gr2[i] = g;                       // reuse as the projected gradient
        ks_r[i] = over ? 1.0f : 0.0f;     // reuse as overflow flag
        mx = fmaxf(mx, fabsf(g));
      }
      coef = fminf(1.0f / (mx + 1e-6f), 1.0f);
      st.out[3] = mx;
      for (int i = 0; i < L * H; ++i) {
        float v = st.r[i] + (-hp.rlr) * (gr2[i] * coef);
        v = fmaxf(v, 0.0f);
        if (ks_r[i] != 0.0f) v = rmax;
        st.r[i] = v;
      }
    }
    __threadfence_block();
  }
  __syncthreads();
  if (!go_on) { write_back(); return; }
  // ---- phase 3: least-k sums at the UPDATED s, r (uvc_utils.py:231-254)
  for (int l = wv; l < L; l += nwv) {
    GroupStat a = group_stat(st.scores2 + l * H, st.rankh + l * H, H, (int)ceilf(st.s[l * 2 + 0]));
    GroupStat b = group_stat(st.scores3 + l * d.F, st.rank3 + l * d.F, d.F, (int)ceilf(st.s[l * 2 + 1]));
    if (ln0 == 0) { ks_s[l * 2 + 0] = a.ksum; ks_s[l * 2 + 1] = b.ksum; }
    for (int h = 0; h < H; ++h) {
      GroupStat c = group_stat(st.scores1 + l * d.D + h * d.hd, st.rank1 + l * d.D + h * d.hd, d.hd, (int)ceilf(st.r[l * H + h]));
      if (ln0 == 0) ks_r[l * H + h] = c.ksum;
    }
  }
  __syncthreads();
  // ---- phase 4 (thread 0): dual ascent + projection (uvc_optimizer.py:126-135, uvc_utils.py:256-269,403-406)
  if (threadIdx.x == 0) {
    const float R2 = resource_eval(st, d, hp, e2, false, nullptr, nullptr, nullptr, false);
    st.out[1] = R2;
    for (int i = 0; i < L * 2; ++i) st.y[i] = fmaxf(st.y[i] + hp.ylr * ks_s[i], 0.0f);
    for (int i = 0; i < L * H; ++i) st.p[i] = fmaxf(st.p[i] + hp.plr * ks_r[i], 0.0f);
    st.z[0] = fmaxf(st.z[0] + hp.zlr * (R2 - hp.budget), 0.0f);
  }
  write_back();
}

// ---------------------------------------------------------------------------- C-ABI
static int check_dims(const uvc_dims& d) {
  if (d.L <= 0 || d.L > UVC_MAX_L || d.H <= 0 || d.H > UVC_MAX_H || d.hd <= 0 || d.D != d.H * d.hd ||
      d.D % 64 != 0 || d.F % 64 != 0 || d.F <= 0 || d.F > 8192 || d.D > 8192)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_dims: need 0<L<=32, 0<H<=32, D=H*hd, D%64==0, F%64==0, D,F<=8192");
  return UVC_OK;
}

extern "C" int uvc_scores(const float* const* W1, const float* const* W3, uvc_dims d, double* ws64, float* scores1,
                          float* scores2, float* scores3, void* stream) {
  if (int e = check_dims(d)) return e;
  if (!W1 || !W3 || !ws64 || !scores1 || !scores2 || !scores3) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_scores: null pointer");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((d.D + d.F) / 64, d.L);
  k_scores<0><<<grid, 256, 0, st>>>((float* const*)W1, (float* const*)W3, d, ws64, scores1, scores3, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, 0.0);
  UVC_CHECK_LAUNCH();
  k_head_scores<<<d.L, 64, 0, st>>>(ws64, d, scores2);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_rank(const float* scores1, const float* scores2, const float* scores3, uvc_dims d, int32_t* rank1,
                        int32_t* rankh, int32_t* rank3, void* stream) {
  if (int e = check_dims(d)) return e;
  if (!scores1 || !scores2 || !scores3 || !rank1 || !rankh || !rank3) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_rank: null pointer");
  const size_t sh = sizeof(float) * (size_t)(d.F > d.D ? d.F : d.D);
  k_rank<<<dim3((d.F + 255) / 256, d.L), 256, sh, (hipStream_t)stream>>>(scores1, scores2, scores3, d, rank1, rankh, rank3);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_prox(float* const* W1, float* const* W3, uvc_dims d, const int32_t* rank1, const int32_t* rankh,
                        const int32_t* rank3, const float* s, const float* r, const float* y, const float* p, double lr,
                        double* ws64, float* scores1_post, float* scores2_post, float* scores3_post, void* stream) {
  if (int e = check_dims(d)) return e;
  if (!W1 || !W3 || !rank1 || !rankh || !rank3 || !s || !r || !y || !p || !ws64 || !scores1_post || !scores2_post ||
      !scores3_post)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_prox: null pointer");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((d.D + d.F) / 64, d.L);
  k_scores<1><<<grid, 256, 0, st>>>(W1, W3, d, ws64, scores1_post, scores3_post, rank1, rankh, rank3, s, r, y, p, lr);
  UVC_CHECK_LAUNCH();
  k_head_scores<<<d.L, 64, 0, st>>>(ws64, d, scores2_post);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

static int check_state(const uvc_state* st, const uvc_hyper& hp, bool need_grad) {
  if (!st) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_state: null");
  if (!st->s || !st->r || !st->y || !st->p || !st->z || !st->total_macs || !st->scores1 || !st->scores2 ||
      !st->scores3 || !st->rank1 || !st->rankh || !st->rank3 || !st->out)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_state: null member");
  if (hp.enable_block_gating && st->gate && need_grad &&
      (!st->gate_grad || !st->gate_momentum || !st->gate_gsum || !st->gate_counters))
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_state: gating enabled but gate_grad/momentum/gsum/counters missing");
  if (hp.gating_interval <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_hyper: gating_interval must be > 0");
  return UVC_OK;
}

extern "C" int uvc_dual_step(const uvc_state* st, uvc_dims d, uvc_hyper hp, const float* e1, const float* e2,
                             int32_t enable_warmup, int32_t global_step, void* stream) {
  if (int e = check_dims(d)) return e;
  if (int e = check_state(st, hp, !enable_warmup)) return e;
  const bool gum = hp.enable_block_gating && st->gate && hp.use_gumbel;
  if (gum && (!e1 || (!enable_warmup && !e2))) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_dual_step: Exp(1) draws missing");
  const int nwaves = d.L < 16 ? d.L : 16;                  // one wave per layer for the least-k scans
  k_dual_step<<<1, 64 * nwaves, 0, (hipStream_t)stream>>>(*st, d, hp, e1, e2, enable_warmup, global_step);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_resource(const uvc_state* st, uvc_dims d, uvc_hyper hp, const float* e, int32_t hard, float* out,
                            void* stream) {
  if (int e_ = check_dims(d)) return e_;
  if (int e_ = check_state(st, hp, false)) return e_;
  if (!out) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_resource: out is null");
  if (hp.enable_block_gating && st->gate && hp.use_gumbel && !e) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_resource: Exp(1) draws missing");
  k_resource<<<1, 64, 0, (hipStream_t)stream>>>(*st, d, hp, e, hard, out);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}

extern "C" int uvc_write_masks(float* const* mask_proj, float* const* mask_fc2, float* const* mask_fc1, uvc_dims d,
                               const int32_t* rank1, const int32_t* rankh, const int32_t* rank3, const float* s,
                               const float* r, void* stream) {
  if (int e = check_dims(d)) return e;
  if (!mask_proj || !mask_fc2 || !mask_fc1 || !rank1 || !rankh || !rank3 || !s || !r)
    return uvc_set_error_msg(UVC_ERR_ARG, "uvc_write_masks: null pointer");
  dim3 grid((d.D + d.F) / 64, d.L);
  k_masks<<<grid, 256, 0, (hipStream_t)stream>>>(mask_proj, mask_fc2, mask_fc1, d, rank1, rankh, rank3, s, r);
  UVC_CHECK_LAUNCH();
  return UVC_OK;
}
