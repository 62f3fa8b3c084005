// DeiT forward / backward sequencer: enqueues the fixed kernel sequence of one
// DistilledVisionTransformer pass (UVC/models/model_distilled.py:429-531) on a HIP stream.
// Host code only; every arithmetic step is one of the kernels behind uvc_kernels.h.
#include "common.h"
#include "../../include/uvc_kernels.h"
#include "../../include/uvc_vit.h"
#include <string.h>

namespace {

struct Dims {
  int B, S, P, C, D, L, H, F, NC, ntok, np, N, M, K0, dtype, qkv_bias;
  int rlow;              // the residual stream (x_l, x1 of every block, the final rows) is bf16: the throughput mode unless uvc_vit_cfg.resid_f32
  float eps;
  size_t tsz, rsz;       // bytes per operand element (T); per residual-stream element
};
Dims dims_of(const uvc_vit_cfg& c, int B) {
  Dims d;
  d.B = B; d.S = c.img_size; d.P = c.patch_size; d.C = c.in_chans; d.D = c.embed_dim; d.L = c.depth; d.H = c.num_heads;
  d.F = c.hidden; d.NC = c.num_classes; d.ntok = c.ntok; d.np = (c.img_size / c.patch_size) * (c.img_size / c.patch_size);
  d.N = d.np + d.ntok; d.M = B * d.N; d.K0 = c.in_chans * c.patch_size * c.patch_size; d.dtype = c.dtype;
  d.tsz = c.dtype == UVC_F32 ? 4 : 2;
  d.rlow = (c.dtype == UVC_BF16 && !c.resid_f32) ? 1 : 0;
  d.rsz = d.rlow ? 2 : 4;
  d.eps = c.ln_eps > 0.f ? c.ln_eps : 1e-6f;
  d.qkv_bias = c.no_qkv_bias ? 0 : 1;
  return d;
}

int check_cfg(const uvc_vit_cfg* c) {
  if (!c) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: null cfg");
  if (c->depth <= 0 || c->depth > UVC_VIT_MAX_DEPTH) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: depth out of range");
  if (c->embed_dim != c->num_heads * 64) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_vit: head_dim must be 64");
  if (c->embed_dim % 64 || c->hidden % 64 || c->embed_dim > 1024) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_vit: embed_dim/hidden must be multiples of 64, embed_dim <= 1024");
  if (c->img_size % c->patch_size || c->patch_size % 4) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: img_size/patch_size");
  if (c->ntok != 1 && c->ntok != 2) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: ntok must be 1 or 2");
  if (c->dtype != UVC_F32 && c->dtype != UVC_BF16) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: dtype");
  if (c->num_classes <= 0 || c->num_classes % 8) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_vit: num_classes must be a positive multiple of 8");
  const int np = (c->img_size / c->patch_size) * (c->img_size / c->patch_size);
  if (np + c->ntok > 256) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_vit: sequence length > 256");
  return UVC_OK;
}

inline int64_t al4(int64_t n) { return (n + 3) & ~(int64_t)3; }

// ---- workspace carving ---------------------------------------------------------------------------
struct Carver {
  char* base; int64_t off;
  void* take(int64_t bytes) { void* p = base ? base + off : nullptr; off += (bytes + 255) & ~(int64_t)255; return p; }
};

struct BlockBufs {      // x, x1: residual-stream rows (float32, or bf16 when Dims::rlow)
  void* x; void* h1; void* qkv; void* o; float* lse; float* mean1; float* rstd1;
  void* x1; void* h2; float* mean2; float* rstd2; void* a; void* u;
};
// Compact [B * ntok, .] buffers of the last block's tail (token_tail.hip): only the class / distillation token rows of the last
// block reach the head, so its attention output, proj, LayerNorm2, MLP and their backward run on those rows alone
struct TailBufs {
  void* xc; void* oc; void* x1c; void* h2c; float* mean2c; float* rstd2c; void* ac; void* uc; void* xoutc;      // xc, x1c, xoutc: residual-stream rows
  void* gAc; void* dAc; void* dHc; void* gBc; void* dOc;      // backward
};
struct Work {
  TailBufs tail;
  void* patches; float* pe; BlockBufs blk[UVC_VIT_MAX_DEPTH]; void* xL;
  void* hc; float* meanf; float* rstdf;
  // backward scratch
  void* gA; void* gB;      // dL/dx streams of the backward: T (bf16 in the throughput mode: every consumer is a bf16 GEMM
                           // operand or a LayerNorm backward that accumulates in float32), float32 in the exact mode
  void* dA; void* dH; void* dqkv; float* delta; float* ln_partial; int64_t ln_region; float* cs_partial;
  // every block has its OWN copies of the streams its weight gradients read (gAs[l]: dL/dx_l, written by block l; gAs[L]: the top), so the main stream never
  // overwrites what the side stream may still be reading and needs no event wait inside a backward call; gA / gB / dA / dqkv above point at the current block's
  void* gAs[UVC_VIT_MAX_DEPTH + 1]; void* gBs[UVC_VIT_MAX_DEPTH]; void* dAs[UVC_VIT_MAX_DEPTH]; void* dqkvs[UVC_VIT_MAX_DEPTH];
  void* gA_pp[2];          // training == 2 (one shared set of backward streams): the two dL/dx buffers gAs[] alternates between
  void* tn_ws; int64_t tn_ws_bytes; float* dotsraw; void* dhc; void* dpe;
  void* tn_ws_main;        // split-M partials of the weight gradients that run on the MAIN stream beside the side stream's (tn_ws_bytes too)
};

int64_t max_tn_ws(const Dims& d) {
  int64_t best = 0, b; int s;
  const int shapes[6][3] = {{d.M, d.D, d.F}, {d.M, d.F, d.D}, {d.M, d.D, d.D}, {d.M, 3 * d.D, d.D}, {d.B, d.NC, d.D}, {d.B * d.np, d.D, d.K0}};
  for (auto& sh : shapes) { uvc_gemm_tn_workspace_bytes(sh[0], sh[1], sh[2], &b, &s); if (b > best) best = b; }
  // compacted MLP widths (Stage-2: multiples of 256 below F) tile differently: fewer tiles, more M splits
  for (int fe = 256; fe < d.F; fe += 256) {
    uvc_gemm_tn_workspace_bytes(d.M, d.D, fe, &b, &s); if (b > best) best = b;
    uvc_gemm_tn_workspace_bytes(d.M, fe, d.D, &b, &s); if (b > best) best = b;
  }
  return best;
}

int64_t carve(const Dims& d, int training, char* base, Work& w) {
  Carver c{base, 0};
  const int64_t MD = (int64_t)d.M * d.D, MF = (int64_t)d.M * d.F;
  w.patches = c.take((int64_t)d.B * d.np * d.K0 * d.tsz);
  w.pe = (float*)c.take((int64_t)d.B * d.np * d.D * 4);
  const int nb = training ? d.L : 1;
  for (int l = 0; l < nb; ++l) {
    BlockBufs& b = w.blk[l];
    b.x = c.take(MD * d.rsz);
    b.h1 = c.take(MD * d.tsz); b.qkv = c.take(3 * MD * d.tsz); b.o = c.take(MD * d.tsz);
    b.lse = (float*)c.take((int64_t)d.B * d.H * d.N * 4);
    b.mean1 = (float*)c.take((int64_t)d.M * 4); b.rstd1 = (float*)c.take((int64_t)d.M * 4);
    b.x1 = c.take(MD * d.rsz);
    b.h2 = c.take(MD * d.tsz);
    b.mean2 = (float*)c.take((int64_t)d.M * 4); b.rstd2 = (float*)c.take((int64_t)d.M * 4);
    b.a = c.take(MF * d.tsz); b.u = c.take(MF * d.tsz);
  }
  w.xL = c.take(MD * d.rsz);
  if (!training) {                       // inference: every block reuses block 0's buffers, x ping-pongs with xL
    for (int l = 1; l < d.L; ++l) { w.blk[l] = w.blk[0]; }
  }
  w.hc = c.take((int64_t)d.B * d.ntok * d.D * d.tsz);
  w.meanf = (float*)c.take((int64_t)d.B * d.ntok * 4); w.rstdf = (float*)c.take((int64_t)d.B * d.ntok * 4);
  if (training == 2) {
    // ONE set of backward streams for all blocks (uvc_vit_workspace_bytes(.., training = 2), uvc_vit_io.shared_bwd_streams): for a backward WITHOUT a side
    // stream, where every weight gradient runs in order behind its operands' producers and the per-block copies buy nothing -- L x (5 MD + MF) elements less
    // (17 GB for DeiT-Base at batch 512).  dL/dx ping-pongs between two buffers by the count of blocks that ran (assign_shared_streams).
    w.gA_pp[0] = c.take(MD * d.tsz); w.gA_pp[1] = c.take(MD * d.tsz);
    void* gB = c.take(MD * d.tsz); void* dA = c.take(MF * d.tsz); void* dqkv = c.take(3 * MD * d.tsz);
    for (int l = 0; l <= d.L; ++l) w.gAs[l] = w.gA_pp[(d.L - l) & 1];
    for (int l = 0; l < d.L; ++l) { w.gBs[l] = gB; w.dAs[l] = dA; w.dqkvs[l] = dqkv; }
  } else if (training) {
    for (int l = 0; l <= d.L; ++l) w.gAs[l] = c.take(MD * d.tsz);
    for (int l = 0; l < d.L; ++l) { w.gBs[l] = c.take(MD * d.tsz); w.dAs[l] = c.take(MF * d.tsz); w.dqkvs[l] = c.take(3 * MD * d.tsz); }
  }
  if (training) {
    w.gA = w.gAs[d.L]; w.gB = w.gBs[0]; w.dA = w.dAs[0]; w.dqkv = w.dqkvs[0];
    w.dH = c.take(MD * d.tsz);
    w.delta = (float*)c.take((int64_t)d.B * d.H * d.N * 4);
    // one private region of per-block dgamma/dbeta/dots partials per LayerNorm-backward call of a pass (2 per block + final norm):
    // the calls leave their partials there and ONE batched launch per backward call finishes them (49 tiny launches less per step)
    // (a region must hold the rows of WHICHEVER kernel uses it: the stand-alone backward writes uvc_layernorm_bwd_blocks(M) rows, the
    // dgrad GEMM with the LayerNorm-backward epilogue uvc_gemm_lnbwd_nblocks(M) = min(ceil(M / 16), 256), which is the larger count for
    // 4096 <= M < ~7.7 k -- sized for one only, the fused call of such a batch wrote past its region into the next call's)
    {
      const int64_t r1 = uvc_layernorm_bwd_blocks(d.M), r2 = uvc_gemm_lnbwd_nblocks(d.M);
      w.ln_region = (r1 > r2 ? r1 : r2) * (2 * d.D + 2);
    }
    w.ln_partial = (float*)c.take(w.ln_region * 4 * (2 * d.L + 1));
    const int maxN = d.F > 3 * d.D ? d.F : 3 * d.D;
    w.cs_partial = (float*)c.take((int64_t)uvc_colsum_blocks(d.M) * (maxN > d.NC ? maxN : d.NC) * 4);
    w.tn_ws_bytes = max_tn_ws(d);
    w.tn_ws = c.take(w.tn_ws_bytes);
    w.tn_ws_main = c.take(w.tn_ws_bytes);
    w.dotsraw = (float*)c.take((int64_t)(d.L + 1) * 2 * 4);
    w.dhc = c.take((int64_t)d.B * d.ntok * d.D * d.tsz);
    w.dpe = c.take((int64_t)d.B * d.np * d.D * d.tsz);
  }
  {
    const int64_t Rt = (int64_t)d.B * d.ntok;
    TailBufs& t = w.tail;
    t.xc = c.take(Rt * d.D * d.rsz); t.oc = c.take(Rt * d.D * d.tsz); t.x1c = c.take(Rt * d.D * d.rsz);
    t.h2c = c.take(Rt * d.D * d.tsz); t.mean2c = (float*)c.take(Rt * 4); t.rstd2c = (float*)c.take(Rt * 4);
    t.ac = c.take(Rt * d.F * d.tsz); t.uc = c.take(Rt * d.F * d.tsz); t.xoutc = c.take(Rt * d.D * d.rsz);
    if (training) {
      t.gAc = c.take(Rt * d.D * d.tsz); t.dAc = c.take(Rt * d.F * d.tsz); t.dHc = c.take(Rt * d.D * d.tsz);
      t.gBc = c.take(Rt * d.D * d.tsz); t.dOc = c.take(Rt * d.D * d.tsz);
    }
  }
  return c.off;
}

// ---- small helpers around the kernel entry points -----------------------------------------------
struct Ctx {
  const uvc_vit_cfg* cfg; Dims d; uvc_vit_offsets off; uvc_vit_shadow_offsets soff; const uvc_vit_io* io; void* st; Work w;
  void* side;            // optional second stream for the weight-gradient GEMMs (backward)
  struct PendingTn { const void* A; int a_f32; const void* B; float* C; float* bias_grad; int M, N1, N2; const float* alpha_ptr; int lda, ldb; bool scratch; };
  PendingTn pend[8]; int n_pend = 0;      // weight gradients of the block in progress: launched together on the side stream behind ONE event (flush_tn)
  bool tn_inline = false;                 // the next weight gradients run on the main stream, in order (the call's last block, see uvc_vit_backward)
  uvc_ln_reduce_item ln_items[64]; int n_ln = 0;   // LayerNorm-backward calls of this pass whose reductions are still pending
};

// Events for the two-stream backward.  Created once per process (host objects, no device memory).
// What an event costs the MAIN stream (tools/probe/event_cost_probe.py, profiles/r5z): a record between two of its kernels 10.6 us, a wait for another stream's
// event 18.5 us more -- packets the command processor handles between the kernels.  Rounds 2-5 recorded one event per weight gradient and waited before every
// buffer a weight gradient might still be reading (4 + 4 per block: 50 us of gaps per block in the step's timeline, profiles/r5w_timeline.txt).  Now: the block's
// four weight gradients are launched together behind ONE record at the block's end, and nothing is waited for (per-block buffers, Work::gAs) until the call's end.
hipEvent_t g_ev_raw = nullptr, g_ev_join = nullptr;
int ensure_events() {
  if (g_ev_raw) return UVC_OK;
  hipError_t e = hipEventCreateWithFlags(&g_ev_raw, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g_ev_join, hipEventDisableTiming);
  if (e != hipSuccess) { g_ev_raw = nullptr; return uvc_set_error(e, __FILE__, __LINE__); }
  return UVC_OK;
}
int launch_tn(Ctx& c, const Ctx::PendingTn& t, void* stream);
// the weight gradients collected since the last flush go to the side stream, behind everything the main stream has enqueued so far
int flush_tn(Ctx& c) {
  if (!c.side || c.n_pend == 0) return UVC_OK;
  hipError_t e = hipEventRecord(g_ev_raw, (hipStream_t)c.st);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)c.side, g_ev_raw, 0);
  if (e != hipSuccess) return uvc_set_error(e, __FILE__, __LINE__);
  // (the four split-M reductions of a block as ONE launch behind the four GEMMs -- built, bit-identical, 11.31 against 11.30 ms in the step: not kept,
  //  tools/probe/tn_batched_reduce.diff.txt, profiles/r5z)
  for (int i = 0; i < c.n_pend; ++i) { if (int r = launch_tn(c, c.pend[i], c.side)) return r; }
  c.n_pend = 0;
  return UVC_OK;
}
int join_side(Ctx& c) {
  if (!c.side) return UVC_OK;
  if (int r = flush_tn(c)) return r;
  hipError_t e = hipEventRecord(g_ev_join, (hipStream_t)c.side);
  if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)c.st, g_ev_join, 0);
  if (e != hipSuccess) return uvc_set_error(e, __FILE__, __LINE__);
  return UVC_OK;
}

const void* sh(const Ctx& c, int64_t off) { return (const char*)c.io->shadow + off * c.d.tsz; }
// GEMM B operand [out,in]: float32 mode reads the master weights directly
const void* wmat(const Ctx& c, int64_t poff, int64_t soff) { return c.d.dtype == UVC_F32 ? (const void*)(c.io->params + poff) : sh(c, soff); }

// norm1 of the next block that runs, written by the kernel that produces its input rows (uvc_vit_io.fuse_next_ln)
struct NextLn { const float* gamma; const float* beta; void* h; float* mean; float* rstd; };
int nt(const Ctx& c, const void* A, int a_f32, const void* B, void* C, int c_f32, int M, int N, int K, int epi, const float* bias = nullptr,
       const void* R = nullptr, const void* R2 = nullptr, const void* aux = nullptr, const float* gate = nullptr, void* C2 = nullptr,
       const float* alpha_ptr = nullptr, int lda = 0, int ldc = 0, const NextLn* ln = nullptr) {
  uvc_gemm_nt_args a;
  memset(&a, 0, sizeof(a));
  if (ln) { a.ln_gamma = ln->gamma; a.ln_beta = ln->beta; a.ln_out = ln->h; a.ln_mean = ln->mean; a.ln_rstd = ln->rstd; a.ln_eps = c.d.eps; }
  a.A = A; a.B = B; a.C = C; a.C2 = C2; a.bias = bias; a.R = R; a.R2 = R2; a.aux = aux; a.gate = gate; a.alpha_ptr = alpha_ptr;
  a.alpha = 1.0f; a.M = M; a.N = N; a.K = K; a.lda = lda ? lda : K; a.ldb = K; a.ldc = ldc ? ldc : N; a.ldr = a.ldc; a.ldaux = a.ldc;
  a.dtype = c.d.dtype; a.a_is_f32 = a_f32 || c.d.dtype == UVC_F32; a.c_is_f32 = c_f32 || c.d.dtype == UVC_F32; a.epilogue = epi;
  a.r_is_f32 = a.c_is_f32;
  a.force_generic = ln ? 0 : c.io->force_generic;      // (a producer that also writes the next LayerNorm exists in one form only)
  return uvc_gemm_nt(&a, c.st);
}
// weight gradient + (same pass over A) bias gradient.  With a side stream the launch is deferred to the next flush_tn (the block's end): its operands live in
// the block's own buffers until the call's end.
int launch_tn(Ctx& c, const Ctx::PendingTn& t, void* stream) {
  uvc_gemm_tn_args a;
  memset(&a, 0, sizeof(a));
  a.colsum_out = t.bias_grad;
  a.A = t.A; a.B = t.B; a.C = t.C; a.workspace = (c.side && stream == c.st) ? c.w.tn_ws_main : c.w.tn_ws; a.workspace_bytes = c.w.tn_ws_bytes; a.alpha_ptr = t.alpha_ptr; a.alpha = 1.0f;
  a.beta = t.scratch ? 0.f : c.io->accumulate; a.M = t.M; a.N1 = t.N1; a.N2 = t.N2; a.lda = t.lda ? t.lda : t.N1; a.ldb = t.ldb ? t.ldb : t.N2; a.ldc = t.N2;
  a.dtype = c.d.dtype; a.a_is_f32 = t.a_f32 || c.d.dtype == UVC_F32;
  return uvc_gemm_tn(&a, stream);
}
int tn(Ctx& c, const void* A, int a_f32, const void* B, float* C, float* bias_grad, int M, int N1, int N2, const float* alpha_ptr = nullptr,
       int lda = 0, int ldb = 0, bool scratch = false) {
  const Ctx::PendingTn t = {A, a_f32, B, C, bias_grad, M, N1, N2, alpha_ptr, lda, ldb, scratch};
  if (!c.side || c.tn_inline) return launch_tn(c, t, c.st);
  if (c.n_pend == 8) { if (int r = flush_tn(c)) return r; }
  c.pend[c.n_pend++] = t;
  return UVC_OK;
}
int csum(const Ctx& c, const void* X, int x_f32, float* out, int M, int N, const float* alpha_ptr = nullptr, int ldx = 0) {
  return uvc_colsum(X, M, N, ldx ? ldx : N, c.d.dtype, x_f32 || c.d.dtype == UVC_F32, c.w.cs_partial, out, 1.0f, alpha_ptr, c.io->accumulate, nullptr, c.st);
}
int ln_fwd(const Ctx& c, const void* x, int64_t pw, int64_t pb, void* y, float* mean, float* rstd, int rows, int rpg, int64_t gs) {
  uvc_ln_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_lowp = c.d.rlow; a.gamma = c.io->params + pw; a.beta = c.io->params + pb; a.y = y; a.mean = mean; a.rstd = rstd; a.eps = c.d.eps;
  a.rows = rows; a.D = c.d.D; a.rows_per_group = rpg; a.group_stride = gs; a.dtype = c.d.dtype;
  return uvc_layernorm_fwd(&a, c.st);
}
int flush_ln(Ctx& c) {
  if (c.n_ln == 0) return UVC_OK;
  const int e = uvc_layernorm_bwd_reduce_batch(c.ln_items, c.n_ln, c.d.D, c.io->accumulate, c.st);
  c.n_ln = 0;
  return e;
}
int ln_bwd(Ctx& c, const void* dy, const void* x, int64_t pw, int64_t pb, const float* mean, const float* rstd, void* dx,
           const void* add1, const float* a1, const void* add2, const float* a2, float* dots, int rows, int rpg, int64_t gs) {
  uvc_ln_args a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.x_lowp = c.d.rlow; a.gamma = c.io->params + pw; a.mean = (float*)mean; a.rstd = (float*)rstd; a.dy = dy; a.dx = dx; a.add1 = add1; a.a1 = a1;
  if (c.n_ln >= 2 * c.d.L + 1) { if (int e = flush_ln(c)) return e; }
  a.add2 = add2; a.a2 = a2; a.partial = c.w.ln_partial + c.w.ln_region * c.n_ln; a.dgamma = c.io->grads + pw; a.dbeta = c.io->grads + pb; a.dots = dots;
  a.defer_reduce = 1;
  uvc_ln_reduce_item& it = c.ln_items[c.n_ln++];
  it.partial = a.partial; it.dgamma = a.dgamma; it.dbeta = a.dbeta; it.dots = dots; it.nblocks = uvc_layernorm_bwd_nblocks(rows); it.reserved = 0;
  a.eps = c.d.eps; a.beta_acc = c.io->accumulate; a.rows = rows; a.D = c.d.D; a.rows_per_group = rpg; a.group_stride = gs; a.dtype = c.d.dtype;
  a.g_lowp = c.d.dtype == UVC_BF16;
  return uvc_layernorm_bwd(&a, c.st);
}
// dgrad GEMM + LayerNorm backward in one kernel (uvc_gemm_nt_lnbwd) where the shape allows; registers the call's partial
// region with the batched finish exactly like ln_bwd
bool lnb_fused_ok(const Ctx& c, int K) { return c.io->force_generic != 1 && uvc_gemm_lnbwd_supported(c.d.M, c.d.D, K, c.d.dtype) != 0; }
int dgrad_ln_bwd(Ctx& c, const void* A, const void* Wt, int K, const void* x, int64_t pw, int64_t pb, const float* mean, const float* rstd, void* dx,
                 const void* add1, const float* a1, const void* add2, const float* a2, float* dots) {
  if (c.n_ln >= 2 * c.d.L + 1) { if (int e = flush_ln(c)) return e; }
  uvc_gemm_lnbwd_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.W = Wt; a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = c.io->params + pw; a.add1 = add1; a.a1 = a1; a.add2 = add2; a.a2 = a2;
  a.dx = dx; a.partial = c.w.ln_partial + c.w.ln_region * c.n_ln; a.M = c.d.M; a.D = c.d.D; a.K = K; a.dtype = c.d.dtype;
  a.variant = c.io->force_generic == 2 ? 1 : 0; a.x_lowp = c.d.rlow;
  uvc_ln_reduce_item& it = c.ln_items[c.n_ln++];
  it.partial = a.partial; it.dgamma = c.io->grads + pw; it.dbeta = c.io->grads + pb; it.dots = dots; it.nblocks = uvc_gemm_lnbwd_nblocks(c.d.M); it.reserved = 0;
  return uvc_gemm_nt_lnbwd(&a, c.st);
}
int attn(const Ctx& c, const BlockBufs& b, bool bwd, int layer = -1) {
  uvc_attn_args a;
  memset(&a, 0, sizeof(a));
  // no-grad forwards skip pruned heads; a BACKWARD skips them when the caller says their attn.proj input columns are masked in the
  // weights (uvc_vit_io.head_keep_bwd: Stage-2): dO of such a head is exactly zero, so dq / dk / dv are written as zeros
  if (layer >= 0 && c.io->head_keep && (bwd ? c.io->head_keep_bwd != 0 : !c.io->training)) a.head_keep = c.io->head_keep + (size_t)layer * c.d.H;
  a.qkv = b.qkv; a.o = b.o; a.lse = b.lse; a.dout = c.w.dH; a.dqkv = c.w.dqkv; a.delta = c.w.delta;
  a.B = c.d.B; a.N = c.d.N; a.H = c.d.H; a.head_dim = 64; a.dtype = c.d.dtype; a.scale = 0.125f;
  return bwd ? uvc_attention_bwd(&a, c.st) : uvc_attention_fwd(&a, c.st);
}

int attn_tok(const Ctx& c, const BlockBufs& b, bool bwd, int layer = -1) {
  uvc_attn_tok_args a;
  memset(&a, 0, sizeof(a));
  if (layer >= 0 && c.io->head_keep && (bwd ? c.io->head_keep_bwd != 0 : !c.io->training)) a.head_keep = c.io->head_keep + (size_t)layer * c.d.H;
  a.qkv = b.qkv; a.o = c.w.tail.oc; a.dout = c.w.tail.dOc; a.dqkv = c.w.dqkv;
  a.B = c.d.B; a.N = c.d.N; a.H = c.d.H; a.head_dim = 64; a.ntok = c.d.ntok; a.dtype = c.d.dtype; a.scale = 0.125f;
  return bwd ? uvc_attention_tok_bwd(&a, c.st) : uvc_attention_tok_fwd(&a, c.st);
}
// GELU'(a) of this MLP call as one byte per activation (uvc_vit_io.gelu_grad_bf16 = 0; UVC_EPI_BIAS_GELU_GRAD_Q8 / UVC_EPI_MUL_AUX_Q8)?  The same answer in the
// forward and in the backward: it depends on the call's shape and the io switches only.
bool gp_q8(const Ctx& c, int rows, int Fe) {
  if (c.io->gelu_grad_bf16 || c.io->force_generic == 1) return false;
  if (c.io->fused_train_mlp && uvc_mlp_fused_supported(c.d.D, Fe, c.d.dtype)) return false;      // (that kernel writes bf16 GELU')
  return uvc_gemm_nt_q8_supported(rows, Fe, c.d.D, c.d.dtype) != 0;
}
// token rows of a [B, N, D] tensor (element size esz) -> compact [B, ntok, D], or back
int gather_tok(const Ctx& c, const void* full, void* compact, size_t esz) {
  return uvc_copy_row_groups(full, compact, c.d.B, (int64_t)c.d.ntok * c.d.D * esz, (int64_t)c.d.N * c.d.D * esz, (int64_t)c.d.ntok * c.d.D * esz, c.st);
}
int scatter_tok(const Ctx& c, const void* compact, void* full, size_t esz) {
  return uvc_copy_row_groups(compact, full, c.d.B, (int64_t)c.d.ntok * c.d.D * esz, (int64_t)c.d.ntok * c.d.D * esz, (int64_t)c.d.N * c.d.D * esz, c.st);
}
// the last block that runs: its tail is computed on the token rows only (unless the caller asks for the reference's full rows)
int tail_block(const Ctx& c) {
  if (c.io->full_tail || c.d.N > 256 || c.d.ntok > 2) return -1;      // beyond what uvc_attention_tok_* takes: all rows, like any other block
  int last = -1;
  for (int l = 0; l < c.d.L; ++l)
    if (c.io->gate_d || !c.io->run_block || c.io->run_block[l] != 0) last = l;
  return last;
}

#define TRY(x) do { if (int e_ = (x)) return e_; } while (0)

int setup(Ctx& c, const uvc_vit_cfg* cfg, const uvc_vit_io* io, void* stream, bool bwd) {
  TRY(check_cfg(cfg));
  if (!io || !io->params || !io->workspace || io->batch <= 0) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: null io member");
  if (cfg->dtype == UVC_BF16 && !io->shadow) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: bf16 mode needs the shadow buffer");
  if (!io->shadow) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: shadow buffer (W^T copies) missing");
  c.cfg = cfg; c.io = io; c.st = stream; c.d = dims_of(*cfg, io->batch);
  c.side = bwd ? io->side_stream : nullptr;
  c.n_pend = 0; c.tn_inline = false;
  if (c.side) TRY(ensure_events());
  TRY(uvc_vit_layout(cfg, &c.off, &c.soff));
  const int mode = (bwd || io->training) ? (io->shared_bwd_streams ? 2 : 1) : 0;
  if (mode == 2 && c.side) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: shared_bwd_streams (workspace mode 2) cannot be combined with a side stream");
  const int64_t need = carve(c.d, mode, (char*)io->workspace, c.w);
  if (io->workspace_bytes < need) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit: workspace too small");
  if (io->patches_in) c.w.patches = const_cast<void*>(io->patches_in);      // rearranged once for student and teacher
  return UVC_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int uvc_vit_layout(const uvc_vit_cfg* cfg, uvc_vit_offsets* off, uvc_vit_shadow_offsets* soff) {
  TRY(check_cfg(cfg));
  const Dims d = dims_of(*cfg, 1);
  if (off) {
    memset(off, 0xff, sizeof(*off));
    int64_t o = 0;
    auto put = [&](int64_t& slot, int64_t n) { slot = o; o += al4(n); };
    put(off->cls_token, d.D);
    if (d.ntok == 2) put(off->dist_token, d.D);
    put(off->pos_embed, (int64_t)d.N * d.D);
    put(off->patch_w, (int64_t)d.D * d.K0); put(off->patch_b, d.D);
    for (int l = 0; l < d.L; ++l) {
      int64_t* b = off->blk[l];
      put(b[0], d.D); put(b[1], d.D); put(b[2], (int64_t)3 * d.D * d.D); put(b[3], 3 * d.D); put(b[4], (int64_t)d.D * d.D); put(b[5], d.D);
      put(b[6], d.D); put(b[7], d.D); put(b[8], (int64_t)d.F * d.D); put(b[9], d.F); put(b[10], (int64_t)d.D * d.F); put(b[11], d.D);
    }
    put(off->norm_w, d.D); put(off->norm_b, d.D);
    put(off->head_w, (int64_t)d.NC * d.D); put(off->head_b, d.NC);
    if (d.ntok == 2) { put(off->headd_w, (int64_t)d.NC * d.D); put(off->headd_b, d.NC); }
    off->n_main = o;
    put(off->gate, 2 * d.L);
    put(off->gumbel_w, d.D); put(off->gumbel_b, 1);
    put(off->patch_gating, d.np);
    for (int l = 0; l < d.L; ++l) { put(off->skip[l][0], 2); put(off->skip[l][1], 2); }
    off->n_total = o;
  }
  if (soff) {
    memset(soff, 0xff, sizeof(*soff));
    int64_t o = 0;
    auto put = [&](int64_t& slot, int64_t n) { slot = o; o += (n + 7) & ~(int64_t)7; };
    put(soff->patch_w, (int64_t)d.D * d.K0);
    const int64_t sz[4] = {(int64_t)3 * d.D * d.D, (int64_t)d.D * d.D, (int64_t)d.F * d.D, (int64_t)d.D * d.F};
    for (int l = 0; l < d.L; ++l)
      for (int j = 0; j < 4; ++j) { put(soff->blk_w[l][j], sz[j]); put(soff->blk_wt[l][j], sz[j]); }
    put(soff->head_w, (int64_t)d.NC * d.D); put(soff->head_wt, (int64_t)d.NC * d.D);
    if (d.ntok == 2) { put(soff->headd_w, (int64_t)d.NC * d.D); put(soff->headd_wt, (int64_t)d.NC * d.D); }
    soff->n_total = o;
  }
  return UVC_OK;
}

extern "C" int64_t uvc_vit_workspace_bytes(const uvc_vit_cfg* cfg, int32_t batch, int32_t training) {
  if (check_cfg(cfg) || batch <= 0) return -1;
  Work w;
  return carve(dims_of(*cfg, batch), training, nullptr, w);
}

// byte offsets inside the training workspace of pe [B*P, D] float32 and dpe [B*P, D] T (patch-gating hooks)
extern "C" int uvc_vit_ws_offsets(const uvc_vit_cfg* cfg, int32_t batch, int32_t training, int64_t* pe_off, int64_t* dpe_off) {
  TRY(check_cfg(cfg));
  if (batch <= 0 || !pe_off || !dpe_off) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit_ws_offsets: bad argument");
  Work w;
  memset(&w, 0, sizeof(w));
  char* base = (char*)256;            // any non-null base: only differences are used
  carve(dims_of(*cfg, batch), training, base, w);
  *pe_off = (char*)w.pe - base;
  *dpe_off = training ? (char*)w.dpe - base : -1;
  return UVC_OK;
}

extern "C" int uvc_stream_create(int32_t priority_class, void** out) {
  if (!out) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_stream_create: null out");
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) return uvc_set_error(e, __FILE__, __LINE__);
  const int prio = priority_class < 0 ? least : (priority_class > 0 ? greatest : 0);
  hipStream_t s = nullptr;
  e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio);
  if (e != hipSuccess) return uvc_set_error(e, __FILE__, __LINE__);
  *out = (void*)s;
  return UVC_OK;
}

extern "C" int uvc_vit_update_shadows(const uvc_vit_cfg* cfg, const float* params, void* shadow, void* stream) {
  TRY(check_cfg(cfg));
  if (!params || !shadow) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit_update_shadows: null pointer");
  const Dims d = dims_of(*cfg, 1);
  uvc_vit_offsets off; uvc_vit_shadow_offsets so;
  TRY(uvc_vit_layout(cfg, &off, &so));
  const bool f32 = d.dtype == UVC_F32;           // float32 mode: only the transposed copies are needed
  int64_t srcs[64], ws[64], wts[64];
  int32_t Rs[64], Cs[64];
  int n = 0;
  auto flush = [&]() -> int {
    if (n == 0) return UVC_OK;
    const int e = uvc_cast_transpose_multi(params, shadow, n, srcs, Rs, Cs, ws, wts, d.dtype, stream);
    n = 0;
    return e;
  };
  auto one = [&](int64_t p, int R, int C, int64_t sw, int64_t swt) -> int {
    srcs[n] = p; Rs[n] = R; Cs[n] = C; ws[n] = f32 ? -1 : sw; wts[n] = swt; ++n;
    return n == 64 ? flush() : UVC_OK;
  };
  if (!f32) TRY(one(off.patch_w, d.D, d.K0, so.patch_w, -1));
  const int RC[4][2] = {{3 * d.D, d.D}, {d.D, d.D}, {d.F, d.D}, {d.D, d.F}};
  const int pi[4] = {2, 4, 8, 10};
  for (int l = 0; l < d.L; ++l)
    for (int j = 0; j < 4; ++j) TRY(one(off.blk[l][pi[j]], RC[j][0], RC[j][1], so.blk_w[l][j], so.blk_wt[l][j]));
  TRY(one(off.head_w, d.NC, d.D, so.head_w, so.head_wt));
  if (d.ntok == 2) TRY(one(off.headd_w, d.NC, d.D, so.headd_w, so.headd_wt));
  return flush();
}

// ------------------------------------------------------------------------------------------------
extern "C" int uvc_vit_forward(const uvc_vit_cfg* cfg, const uvc_vit_io* io, void* stream) {
  Ctx c;
  memset(&c.w, 0, sizeof(c.w));
  TRY(setup(c, cfg, io, stream, false));
  if (!io->x || !io->logits || (cfg->ntok == 2 && !io->logits_dist)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit_forward: null io member");
  const Dims& d = c.d;
  const float* P = io->params;
  const uvc_vit_offsets& o = c.off;
  Work& w = c.w;
  const int rows_p = d.B * d.np;
  // forward stages: 0 = patch embedding (PatchEmbed.forward :145-153), 1 = token assembly (:434-471) + blocks + heads.
  // The cut lets the host compute the patch-gating mask from the patch embedding in between.
  const int fsb = io->stage_begin, fse = (io->stage_begin == 0 && io->stage_end == 0) ? 2 : io->stage_end;
  if (fsb < 0 || fse > 2 || fsb >= fse) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit_forward: bad stage range");
  if (fsb == 0) {
    if (!io->patches_in) TRY(uvc_patchify(io->x, w.patches, d.B, d.C, d.S, d.P, d.dtype, stream));
    TRY(nt(c, w.patches, 0, wmat(c, o.patch_w, c.soff.patch_w), w.pe, 1, rows_p, d.D, d.K0, UVC_EPI_BIAS, P + o.patch_b));
  }
  if (fse < 2) return UVC_OK;
  // hard block skip (:496-500).  In training the input of a block that runs must sit in its own w.blk[l].x (backward
  // reads it there), so the buffer chain hops over skipped blocks: producer -> x of the next block that runs.
  auto runs = [&](int l) { return io->gate_d || !io->run_block || io->run_block[l] != 0; };
  auto next_in = [&](int l) -> void* {
    for (int j = l; j < d.L; ++j)
      if (runs(j)) return w.blk[j].x;
    return w.xL;
  };
  void* x0 = io->training ? next_in(0) : w.blk[0].x;
  TRY(uvc_assemble_tokens(w.pe, P + o.cls_token, d.ntok == 2 ? P + o.dist_token : nullptr, P + o.pos_embed, io->patch_mask, x0, d.B, d.np,
                          d.D, d.ntok, d.rlow, stream));
  void* xin = x0;
  const int tl = tail_block(c);
  const TailBufs& t = w.tail;
  const int Rt = d.B * d.ntok;
  const int rf = d.rlow ? 0 : 1;             // "C is float32" flag of the GEMMs that write residual-stream rows (their R / R2 have C's type)
  bool h1_ready = false;                     // norm1 of the coming block already in its h1 (epilogue of the producer of its input)
  auto next_running = [&](int l) { for (int j = l + 1; j < d.L; ++j) if (runs(j)) return j; return -1; };
  for (int l = 0; l < d.L; ++l) {
    if (!runs(l)) continue;
    const bool tail = l == tl;
    void* xout = tail ? t.xoutc : io->training ? next_in(l + 1) : (xin == w.blk[0].x ? w.xL : w.blk[0].x);
    const BlockBufs& b = w.blk[l];
    const int64_t* q = o.blk[l];
    if (!h1_ready) TRY(ln_fwd(c, xin, q[0], q[1], b.h1, b.mean1, b.rstd1, d.M, 1, d.D));      // else: written by the previous block's MLP kernel
    h1_ready = false;
    // r5: the qkv Linear and the attention forward as ONE kernel where it exists (DeiT-Tiny's shape, bf16): the same bits as the two kernels, h1 in and o out
    // -- qkv is written only for a backward to read.  Not in the last block (its attention runs on the token rows, its K / V on all of them), not with pruned
    // heads skipped (inference on a pruned model), not under force_generic = 1.
    const bool head_skip = io->head_keep && !io->training;
#ifdef UVC_NO_QKV_ATTN_FUSION                            // A/B builds only (tools/exp_ab.sh)
    const bool qkv_attn_fused = false;
#else
    // (D = 192 only: at D = 384 the kernel exists and is bit-identical too, but its weight-fragment reads -- every wave reads the whole chunk from LDS -- make
    //  it no faster than the pair: DeiT-Small 15.70 against 15.67 ms, T2T-ViT-14 12.52 against 12.55, profiles/r5m)
    const bool qkv_attn_fused = !tail && !head_skip && io->force_generic != 1 && d.D == 192 && uvc_qkv_attention_supported(d.B, d.N, d.H, d.D, d.dtype);
#endif
    if (qkv_attn_fused) {
      uvc_qkv_attn_args qa;
      memset(&qa, 0, sizeof(qa));
      qa.h = b.h1; qa.w = wmat(c, q[2], c.soff.blk_w[l][0]); qa.bias = d.qkv_bias ? P + q[3] : nullptr;
      qa.qkv = io->training ? b.qkv : nullptr; qa.o = b.o; qa.lse = b.lse;
      qa.B = d.B; qa.N = d.N; qa.H = d.H; qa.D = d.D; qa.dtype = d.dtype; qa.scale = 0.125f;
      // (persistent workgroups: one per CU.  Measured in the step, profiles/r5l, r5q: 224 / 171 / 128 workgroups for both passes 11.66 / 11.74 / 11.73 ms against 11.61;
      //  the teacher's pass alone on 192 / 128 / 64: within the noise)
      TRY(uvc_qkv_attention_fwd(&qa, c.st));
    } else {
      TRY(nt(c, b.h1, 0, wmat(c, q[2], c.soff.blk_w[l][0]), b.qkv, 0, d.M, 3 * d.D, d.D, d.qkv_bias ? UVC_EPI_BIAS : UVC_EPI_NONE, d.qkv_bias ? P + q[3] : nullptr));
    }
    // From here on the last block works on its token rows only (rows = B * ntok, compact buffers): nothing else of it reaches the head
    const int rows = tail ? Rt : d.M;
    const void* xres = xin;                  // the block's input rows: residual of proj, R2 of the gate mix
    void* x1 = tail ? t.x1c : b.x1;
    void* h2 = tail ? t.h2c : b.h2;
    float* mean2 = tail ? t.mean2c : b.mean2; float* rstd2 = tail ? t.rstd2c : b.rstd2;
    void* ga = tail ? t.ac : b.a; void* gu = tail ? t.uc : b.u;
    if (tail) {
      TRY(attn_tok(c, b, false, l));
      TRY(gather_tok(c, xin, t.xc, d.rsz));
      xres = t.xc;
    } else if (!qkv_attn_fused) {
      TRY(attn(c, b, false, l));
    }
    // Stage-2 compaction: pruned hidden units are skipped (compact weights gathered by the host, uvc_mlp_compact)
    const uvc_mlp_compact* mc = (io->mlp_compact && io->mlp_compact[l].width > 0 && io->mlp_compact[l].width < d.F) ? &io->mlp_compact[l] : nullptr;
    const int Fe = mc ? mc->width : d.F;
    const void* w1 = mc ? mc->w1 : wmat(c, q[8], c.soff.blk_w[l][2]);
    const void* w2 = mc ? mc->w2 : wmat(c, q[10], c.soff.blk_w[l][3]);
    const float* b1 = mc ? mc->b1 : P + q[9];
    const bool mlp_one_kernel = io->force_generic != 1 && uvc_mlp_fused_supported(d.D, Fe, d.dtype) && (io->training ? io->fused_train_mlp != 0 : !io->gate_d);
    // norm2 as a second output of attn.proj + residual (the MLP kernel, where it runs, normalises its rows itself)
    // (not on the compact token rows of a `tail` block: B * ntok may be below the kernel's 16 rows, and whether a norm is fused must
    // not depend on the batch -- the two forms round differently in the last bit)
    const bool ln2_in_proj = !tail && !mlp_one_kernel && io->fuse_next_ln != 0 && uvc_gemm_nt_ln_supported(rows, d.D, d.D, d.dtype, UVC_EPI_BIAS_RESID) && (d.D != 384 || d.rlow);   // (D = 384: bf16 residual rows only)
    {
      const NextLn n2 = {P + q[6], P + q[7], h2, io->training ? mean2 : nullptr, io->training ? rstd2 : nullptr};
      TRY(nt(c, tail ? t.oc : b.o, 0, wmat(c, q[4], c.soff.blk_w[l][1]), x1, rf, rows, d.D, d.D, UVC_EPI_BIAS_RESID, P + q[5], xres, nullptr, nullptr, nullptr, nullptr,
             nullptr, 0, 0, ln2_in_proj ? &n2 : nullptr));
    }
    if (mlp_one_kernel) {
      // LayerNorm + fc1 + GELU + fc2 + residual (+ gate mix) in one kernel, hidden activation in registers.  No-grad forwards
      // (teacher / eval) write nothing but the output.  The training form also stores what the backward reads -- LayerNorm2(x1), its
      // mean / rstd, GELU'(a), GELU(a) -- from the same registers; at 218 us against 187 us for the three kernels it replaces
      // (its 64-byte row pieces of GELU / GELU' write badly) it is opt-in (uvc_vit_io.fused_train_mlp), tested, not the default
      uvc_mlp_args m;
      memset(&m, 0, sizeof(m));
      m.x = x1; m.gamma = P + q[6]; m.beta = P + q[7]; m.w1 = w1; m.b1 = b1;
      m.w2 = w2; m.b2 = P + q[11]; m.out = xout; m.M = rows; m.D = d.D; m.F = Fe; m.eps = d.eps; m.rows_lowp = d.rlow;
      if (io->training) {
        m.h = h2; m.mean = mean2; m.rstd = rstd2; m.gp = ga; m.u = gu;
        if (io->gate_d) { m.x_prev = xres; m.gate = io->gate_d + 2 * l; }
      }
      const int ln = tail ? -1 : next_running(l);
      if (ln >= 0 && io->fuse_next_ln != 0) {      // norm1 of block ln on the rows this kernel holds (all M rows: `tail` blocks stop at their token rows)
        m.next_gamma = P + o.blk[ln][0]; m.next_beta = P + o.blk[ln][1]; m.next_h = w.blk[ln].h1;
        if (io->training) { m.next_mean = w.blk[ln].mean1; m.next_rstd = w.blk[ln].rstd1; }
        h1_ready = true;
      }
      TRY(uvc_mlp_fused_fwd(&m, c.st));
      xin = xout;
      continue;
    }
    if (!ln2_in_proj) TRY(ln_fwd(c, x1, q[6], q[7], h2, mean2, rstd2, rows, 1, d.D));
    if (io->training)
      // `ga` receives GELU'(pre-activation): that is all the backward needs of it (one multiply in the dgrad epilogue)
      TRY(nt(c, h2, 0, w1, ga, 0, rows, Fe, d.D, gp_q8(c, rows, Fe) ? UVC_EPI_BIAS_GELU_GRAD_Q8 : UVC_EPI_BIAS_GELU_GRAD, b1, nullptr, nullptr, nullptr, nullptr, gu));
    else   // inference (teacher / eval): the pre-activation is not needed, write GELU(a) only
      TRY(nt(c, h2, 0, w1, gu, 0, rows, Fe, d.D, UVC_EPI_BIAS_GELU_OUT, b1));
    NextLn nl;
    const NextLn* pnl = nullptr;
    {
      const int ln = tail ? -1 : next_running(l);
      const int epi = io->gate_d ? UVC_EPI_BIAS_RESID_GATE : UVC_EPI_BIAS_RESID;
      if (ln >= 0 && io->fuse_next_ln != 0 && uvc_gemm_nt_ln_supported(rows, d.D, Fe, d.dtype, epi) && (d.D != 384 || d.rlow)) {
        nl.gamma = P + o.blk[ln][0]; nl.beta = P + o.blk[ln][1]; nl.h = w.blk[ln].h1;
        nl.mean = io->training ? w.blk[ln].mean1 : nullptr; nl.rstd = io->training ? w.blk[ln].rstd1 : nullptr;
        pnl = &nl;
        h1_ready = true;
      }
    }
    if (io->gate_d)
      TRY(nt(c, gu, 0, w2, xout, rf, rows, d.D, Fe, UVC_EPI_BIAS_RESID_GATE, P + q[11], x1, xres, nullptr, io->gate_d + 2 * l, nullptr, nullptr, 0, 0, pnl));
    else
      TRY(nt(c, gu, 0, w2, xout, rf, rows, d.D, Fe, UVC_EPI_BIAS_RESID, P + q[11], x1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, pnl));
    xin = xout;
  }
  if (io->training && tl < 0 && xin != w.xL) return uvc_set_error_msg(UVC_ERR_LAUNCH, "uvc_vit_forward: internal buffer chain broken");
  // final norm on the class(/dist) token rows only (:507-508), then the head(s) (:522-526)
  if (tl >= 0) TRY(ln_fwd(c, t.xoutc, o.norm_w, o.norm_b, w.hc, w.meanf, w.rstdf, Rt, 1, d.D));
  else
  TRY(ln_fwd(c, xin, o.norm_w, o.norm_b, w.hc, w.meanf, w.rstdf, d.B * d.ntok, d.ntok, (int64_t)d.N * d.D));
  TRY(nt(c, w.hc, 0, wmat(c, o.head_w, c.soff.head_w), io->logits, 1, d.B, d.NC, d.D, UVC_EPI_BIAS, P + o.head_b, nullptr, nullptr, nullptr, nullptr,
         nullptr, nullptr, d.ntok * d.D));
  if (d.ntok == 2)
    TRY(nt(c, (const char*)w.hc + (size_t)d.D * d.tsz, 0, wmat(c, o.headd_w, c.soff.headd_w), io->logits_dist, 1, d.B, d.NC, d.D, UVC_EPI_BIAS,
           P + o.headd_b, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d.ntok * d.D));
  return UVC_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int uvc_vit_backward(const uvc_vit_cfg* cfg, const uvc_vit_io* io, void* stream) {
  Ctx c;
  memset(&c.w, 0, sizeof(c.w));
  TRY(setup(c, cfg, io, stream, true));
  if (!io->grads || !io->d_logits || (cfg->ntok == 2 && !io->d_logits_dist)) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit_backward: null io member");
  const Dims& d = c.d;
  const float* P = io->params;
  float* G = io->grads;
  const uvc_vit_offsets& o = c.off;
  const uvc_vit_shadow_offsets& so = c.soff;
  Work& w = c.w;
  hipStream_t hs = (hipStream_t)stream;
  const int rh = d.B * d.ntok;
  const int gf = d.dtype == UVC_F32 ? 1 : 0;     // "operand is float32" flag of the gA / gB streams
  const int sb = io->stage_begin, se = (io->stage_begin == 0 && io->stage_end == 0) ? d.L + 3 : io->stage_end;
  if (sb < 0 || se > d.L + 3 || sb >= se) return uvc_set_error_msg(UVC_ERR_ARG, "uvc_vit_backward: bad stage range");
  const int tl = tail_block(c);
  const TailBufs& t = w.tail;
  // dL/dx_k lives in gAs[k] (written by block k; gAs[L]: the final norm's): block l reads the one of the nearest block above it that ran
  auto runs = [&](int l) { return io->gate_d || !io->run_block || io->run_block[l]; };
  auto g_in = [&](int l) { int k = l + 1; while (k < d.L && !runs(k)) ++k; return w.gAs[k]; };
  if (io->shared_bwd_streams) {
    // dL/dx alternates between the two shared buffers by the number of blocks that RAN above (a hard-skipped block neither reads nor writes one), whichever
    // stage range this call covers: block l must never write the buffer it reads
    int cnt = 0;
    w.gAs[d.L] = w.gA_pp[0];
    for (int l = d.L - 1; l >= 0; --l) { if (runs(l)) ++cnt; w.gAs[l] = w.gA_pp[cnt & 1]; }
  }
  w.gA = w.gAs[d.L];
  // The side stream runs a block behind (a block's weight gradients start when the block is through), so the LAST block's would run after the main stream has
  // finished, alone (370 us of the step's timeline, profiles/r5z_timeline.txt): the two MLP weight gradients of that block run on the main stream instead, in
  // order behind their operands' producers (their own split-M workspace), beside the side stream's work on the block before.
  int last_l = -1;
  for (int l = 0; l < d.L; ++l) { const int stage = d.L - l; if (stage >= sb && stage < se && runs(l)) { last_l = l; break; } }
  if (sb == 0) {
  // heads: dhc = dlogits . W ; dW = dlogits^T . hc ; db = colsum(dlogits)
  TRY(nt(c, io->d_logits, 1, sh(c, so.head_wt), w.dhc, 0, d.B, d.D, d.NC, UVC_EPI_NONE, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
         d.ntok * d.D));
  TRY(tn(c, io->d_logits, 1, w.hc, G + o.head_w, G + o.head_b, d.B, d.NC, d.D, nullptr, 0, d.ntok * d.D));
  if (d.ntok == 2) {
    TRY(nt(c, io->d_logits_dist, 1, sh(c, so.headd_wt), (char*)w.dhc + (size_t)d.D * d.tsz, 0, d.B, d.D, d.NC, UVC_EPI_NONE, nullptr, nullptr, nullptr,
           nullptr, nullptr, nullptr, nullptr, 0, d.ntok * d.D));
    TRY(tn(c, io->d_logits_dist, 1, (const char*)w.hc + (size_t)d.D * d.tsz, G + o.headd_w, G + o.headd_b, d.B, d.NC, d.D, nullptr, 0, d.ntok * d.D));
  }
  // final norm backward -> gA = dL/dx_L (zero except the token rows)
  hipError_t he = hipMemsetAsync(w.gA, 0, (size_t)d.M * d.D * d.tsz, hs);
  if (he != hipSuccess) return uvc_set_error(he, __FILE__, __LINE__);
  if (tl >= 0) {      // the last block's output exists on the token rows only: compact gradient, copied into the (zero) full stream
    TRY(ln_bwd(c, w.dhc, t.xoutc, o.norm_w, o.norm_b, w.meanf, w.rstdf, t.gAc, nullptr, nullptr, nullptr, nullptr, w.dotsraw + 2 * d.L, rh, 1, d.D));
    TRY(scatter_tok(c, t.gAc, w.gA, d.tsz));
  } else
  TRY(ln_bwd(c, w.dhc, w.xL, o.norm_w, o.norm_b, w.meanf, w.rstdf, w.gA, nullptr, nullptr, nullptr, nullptr, w.dotsraw + 2 * d.L, rh, d.ntok,
             (int64_t)d.N * d.D));
  }
  for (int l = d.L - 1; l >= 0; --l) {
    const int stage = d.L - l;
    if (stage < sb || stage >= se) continue;
    if (!io->gate_d && io->run_block && !io->run_block[l]) continue;   // skipped in forward: no gradient reaches its parameters
    const BlockBufs& b = w.blk[l];
    const int64_t* q = o.blk[l];
    const float* g0 = io->gate_d ? io->gate_d + 2 * l : nullptr;       // d0
    const float* g1 = io->gate_d ? io->gate_d + 2 * l + 1 : nullptr;   // d1
    // MLP: out = d1*(x1 + fc2(u)) + d0*x.  The last block that ran did so on its token rows only (forward): its gradient
    // streams up to the attention are the compact ones, with the same kernels at rows = B * ntok.
    const bool tail = l == tl;
    const int rows = tail ? d.B * d.ntok : d.M;
    void* const gA_in = g_in(l);                                       // dL/d(this block's output); w.gA: where this block leaves dL/dx_l
    w.gA = w.gAs[l]; w.gB = w.gBs[l]; w.dA = w.dAs[l]; w.dqkv = w.dqkvs[l];
    void* gA = tail ? t.gAc : gA_in; void* gB = tail ? t.gBc : w.gB; void* dA = tail ? t.dAc : w.dA; void* dH = tail ? t.dHc : w.dH;
    const void* fa = tail ? t.ac : b.a; const void* fu = tail ? t.uc : b.u; const void* fh2 = tail ? t.h2c : b.h2;
    const void* fx1 = tail ? t.x1c : b.x1; const float* fm2 = tail ? t.mean2c : b.mean2; const float* fr2 = tail ? t.rstd2c : b.rstd2;
    const uvc_mlp_compact* mc = (io->mlp_compact && io->mlp_compact[l].width > 0 && io->mlp_compact[l].width < d.F) ? &io->mlp_compact[l] : nullptr;
    const bool fuse2 = !mc && !tail && lnb_fused_ok(c, d.F);
    c.tn_inline = c.side && l == last_l && !mc;
    if (!mc) {
      TRY(nt(c, gA, gf, sh(c, so.blk_wt[l][3]), dA, 0, rows, d.F, d.D, gp_q8(c, rows, d.F) ? UVC_EPI_MUL_AUX_Q8 : UVC_EPI_MUL_AUX, nullptr, nullptr, nullptr, fa, nullptr,
             nullptr, g1));
      TRY(tn(c, gA, gf, fu, G + q[10], G + q[11], rows, d.D, d.F, g1));
      if (!fuse2) TRY(nt(c, dA, 0, sh(c, so.blk_wt[l][2]), dH, 0, rows, d.D, d.F, UVC_EPI_NONE));
      TRY(tn(c, dA, 0, fh2, G + q[8], G + q[9], rows, d.F, d.D));
    } else {
      const int Fe = mc->width;
      if (io->accumulate != 0.f) return uvc_set_error_msg(UVC_ERR_UNSUPPORTED, "uvc_vit_backward: gradient accumulation with MLP compaction");
      TRY(nt(c, gA, gf, mc->w2t, dA, 0, rows, Fe, d.D, gp_q8(c, rows, Fe) ? UVC_EPI_MUL_AUX_Q8 : UVC_EPI_MUL_AUX, nullptr, nullptr, nullptr, fa, nullptr, nullptr, g1));
      TRY(tn(c, gA, gf, fu, mc->dw2, G + q[11], rows, d.D, Fe, g1, 0, 0, true));     // db2 goes straight to its place
      TRY(nt(c, dA, 0, mc->w1t, dH, 0, rows, d.D, Fe, UVC_EPI_NONE));
      TRY(tn(c, dA, 0, fh2, mc->dw1, mc->db1, rows, Fe, d.D, nullptr, 0, 0, true));
      TRY(flush_tn(c));                       // (the two compact gradients must be on the side stream before their expansion)
      // expand into the full gradient tensors on the stream the wgrads ran on (rank-1 columns for the pruned units)
      TRY(uvc_mlp_scatter_grads(mc->dw1, mc->dw2, mc->db1, mc->inv, P + q[9], G + q[11], d.D, d.F, Fe, G + q[8], G + q[10], G + q[9], 0.f, d.dtype,
                                c.side ? c.side : c.st));
    }
    c.tn_inline = false;
    if (fuse2)      // gB = dL/dx1 = LN2'(dA . W1) + d1*gA, the dgrad of fc1 consumed in its epilogue
      TRY(dgrad_ln_bwd(c, w.dA, sh(c, so.blk_wt[l][2]), d.F, b.x1, q[6], q[7], b.mean2, b.rstd2, w.gB, gA_in, g1, nullptr, nullptr, nullptr));
    else
    TRY(ln_bwd(c, dH, fx1, q[6], q[7], fm2, fr2, gB, gA, g1, nullptr, nullptr, nullptr, rows, 1, d.D));   // gB = dL/dx1
    // attention
    TRY(nt(c, gB, gf, sh(c, so.blk_wt[l][1]), tail ? t.dOc : w.dH, 0, rows, d.D, d.D, UVC_EPI_NONE));                          // dO
    // (dW_proj on the main stream in every / every second / every third block, to even the two streams out: 11.47 / 11.48 / 11.42 against 11.38 ms, profiles/r5z)
    TRY(tn(c, gB, gf, tail ? t.oc : b.o, G + q[4], G + q[5], rows, d.D, d.D));
    // (the last block's dW_proj flushed to the idle side stream here, under the attention backward: 11.10 against 11.08 ms -- nothing, profiles/r5zz_ab_early_last_flush.txt)
    if (tail) {
      TRY(attn_tok(c, b, true, l));         // writes all of dqkv: dq is zero off the token rows, dk / dv are dense
      // the full-row stream of dL/dx1 that the LayerNorm1 backward adds: zero but for the token rows
      const hipError_t he = hipMemsetAsync(w.gB, 0, (size_t)d.M * d.D * d.tsz, hs);
      if (he != hipSuccess) return uvc_set_error(he, __FILE__, __LINE__);
      TRY(scatter_tok(c, t.gBc, w.gB, d.tsz));
    } else
    TRY(attn(c, b, true, l));
    const bool fuse1 = lnb_fused_ok(c, 3 * d.D);
    if (!fuse1) TRY(nt(c, w.dqkv, 0, sh(c, so.blk_wt[l][0]), w.dH, 0, d.M, d.D, 3 * d.D, UVC_EPI_NONE));
    TRY(tn(c, w.dqkv, 0, b.h1, G + q[2], d.qkv_bias ? G + q[3] : nullptr, d.M, 3 * d.D, d.D));
    // gA (this block's) <- dL/dx_l = LN1'(dH) + gB + d0 * (dL/d output) ; dots: <new gA, x_l>, <old gA, x_l>
    // (a `tail` block's full-row output gradient is zero off its token rows, which is what its compact stream gAc holds: the d0 term reads the full-row
    //  stream the final norm's backward left in gAs[L])
    if (fuse1)
      TRY(dgrad_ln_bwd(c, w.dqkv, sh(c, so.blk_wt[l][0]), 3 * d.D, b.x, q[0], q[1], b.mean1, b.rstd1, w.gA, w.gB, nullptr, io->gate_d ? gA_in : nullptr, g0,
                       w.dotsraw + 2 * l));
    else
    TRY(ln_bwd(c, w.dH, b.x, q[0], q[1], b.mean1, b.rstd1, w.gA, w.gB, nullptr, io->gate_d ? gA_in : nullptr, g0, w.dotsraw + 2 * l, d.M, 1, d.D));
    TRY(flush_tn(c));                        // this block's four weight gradients (and, behind the first block, the heads'): one event
  }
  w.gA = g_in(-1);                           // dL/dx_0: the lowest block that ran left it
  if (sb <= d.L + 1 && se > d.L + 1) {
    // gate logits (block_skip_gating) gradient
    TRY(flush_ln(c));                          // the gate gradient reads the dot products of the blocks' LayerNorm backwards
    if (io->gate_d && io->gate_mode != 0)
      TRY(uvc_gate_grad(P + o.gate, io->gate_d, w.dotsraw, G + o.gate, d.L, io->gate_mode, io->gate_eps, io->accumulate, stream));
    // token assembly
    TRY(uvc_assemble_tokens_bwd(w.gA, w.pe, io->patch_mask, w.dpe, G + o.pos_embed, G + o.cls_token, d.ntok == 2 ? G + o.dist_token : nullptr,
                                io->d_patch_mask, d.B, d.np, d.D, d.ntok, d.dtype, 0, d.dtype == UVC_BF16, io->accumulate, stream));
  }
  TRY(flush_ln(c));
  if (se < d.L + 3) return join_side(c);
  // patch embedding (weight gradient only: the image needs no gradient)
  TRY(tn(c, w.dpe, 0, w.patches, G + o.patch_w, G + o.patch_b, d.B * d.np, d.D, d.K0));
  return join_side(c);
}
