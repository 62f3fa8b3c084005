/* C-ABI of the T2T-ViT tokens-to-token front end on MI355X.
 *
 * Replaces, for BASELINE config 5, what the reference runs through ATen + autograd over
 *   UVC/T2TViT/models/t2t_vit.py:84-105     T2T_module.forward (soft split = nn.Unfold, re-structurisation, project)
 *   UVC/T2TViT/models/token_performer.py:31-69  Token_performer (prm_exp, single_attn, forward)
 * The Linear layers, LayerNorm(64) and the GELU MLP of a Performer stage run on the GEMM / LayerNorm kernels of
 * uvc_kernels.h; this header adds what those do not cover: the soft split fused with the stage's first LayerNorm
 * (forward and backward), the fold that is the soft split's adjoint, and the Performer's linear attention.
 * All tensors are device pointers; every call only enqueues work on `stream`; no allocation inside.
 */
#ifndef UVC_T2T_H
#define UVC_T2T_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Soft split (+ LayerNorm): rows of nn.Unfold(k, stride s, padding p)(x).transpose(1, 2) (t2t_vit.py:86,93,100), feature
 * index c*k*k + ki*k + kj, optionally followed by LayerNorm over the `dim` = C*k*k features (token_performer.py:65,
 * norm1).  The source is addressed by element strides, so both the NCHW image (sc = H*W, sh = W, sw = 1) and the
 * token-major output of the previous Performer stage [B, H*W, C] (sc = 1, sh = W*C, sw = C) are read in place -- the
 * reference's transpose + reshape (t2t_vit.py:91,98) never materialises.  Output rows are `ldo` elements apart
 * (ldo >= dim, multiple of 8); columns [dim, ldo) are written as zeros (GEMM K padding). */
typedef struct uvc_unfold_args {
  const float* src; int64_t sb, sc, sh, sw;
  int32_t B, C, H, W, k, s, p;
  int32_t ldo, out_is_f32, dtype;       /* out: T of `dtype`, or float32 when out_is_f32 */
  const float* gamma; const float* beta; float eps;   /* gamma NULL: plain soft split */
  void* out; float* mean; float* rstd;  /* mean / rstd [B*L] (LayerNorm only) */
  /* backward (uvc_unfold_ln_bwd): dY [B*L, ldo] is the gradient wrt `out` */
  const void* dy; int32_t dy_is_f32;
  float* dxu;                            /* [B*L, dim] float32 gradient wrt the unfolded row, or NULL when the source needs none.  Column order: natural
                                          * (c*k*k + ki*k + kj), or tap-major ((ki*k + kj)*C + c) when dxu_tap_major (last field) is set */
  float* partial;                        /* scratch [uvc_unfold_bwd_blocks() * 2 * dim] */
  float* dgamma; float* dbeta; float beta_acc;   /* written as beta_acc*old + sum */
  int32_t dxu_tap_major;                 /* token-major sources (sc == 1, C == 64) only: dxu leaves as [B*L][k*k][C], which uvc_fold_tokens(tap_major = 1) reads as
                                          * whole 256-byte channel rows (one wave per pixel, the channel on the lane) */
} uvc_unfold_args;
int uvc_unfold_ln_fwd(const uvc_unfold_args* args, void* stream);
int uvc_unfold_ln_bwd(const uvc_unfold_args* args, void* stream);
int uvc_unfold_bwd_blocks(int32_t rows);

/* Fold = adjoint of the soft split onto a token-major map: dst[b, h*W + w, c] = sum over the windows (ho, wo) and taps
 * (ki, kj) that cover (h, w) of src[b*L + ho*Wo + wo, c*k*k + ki*k + kj].  A gather (deterministic, no atomics).
 * src is float32 or T (`src_is_f32`), rows `lds` apart; dst [B, H*W, C] is float32 or T (`dst_is_f32`; sums in float32).
 * tap_major: the columns of src are ordered (ki*k + kj)*C + c (uvc_unfold_args.dxu_tap_major) instead of c*k*k + ki*k + kj. */
int uvc_fold_tokens(const void* src, int32_t src_is_f32, int32_t dtype, int32_t lds, void* dst, int32_t dst_is_f32, int32_t B, int32_t C, int32_t H,
                    int32_t W, int32_t k, int32_t s, int32_t p, int32_t tap_major, void* stream);

/* Performer linear attention (token_performer.py:31-62) for emb = 64, m = 32 random features.
 * kqv [B*T, 192] float32 = Linear(norm1(x)) split as k | q | v (:46); w [32, 64] float32 (the fixed random features).
 * forward:  kptv[b] = [ sum_t v_t kp_t^T (64 x 32) ; sum_t kp_t (32) ]  with kp = exp(w k - |k|^2/2) / sqrt(m)   (:47,49)
 *           att[t]  = (qp_t kptv^T) / (qp_t . ksum + 1e-8)                                                        (:48,50)
 * backward: given datt, writes dkqv = d(loss)/d(k | q | v); `dskip` [B*T, 64] (optional) is added to dv -- v is also the
 *           skip connection of the stage (:52).
 * part: scratch [B * uvc_performer_splits(B, T) * 65 * 32] float32; kptv / dkptv: [B, 65, 32] float32. */
typedef struct uvc_performer_args {
  const float* kqv; const float* w;
  float* part; float* kptv;
  void* att; int32_t att_is_f32;          /* [B*T, 64] T or float32 */
  /* backward */
  const void* datt; const void* dskip;    /* gradient streams: T, or float32 when g_is_f32 */
  void* dkqv;                              /* [B*T, 192] same element type as datt */
  float* dkptv;
  int32_t g_is_f32;
  int32_t B, T, dtype;
} uvc_performer_args;
int uvc_performer_splits(int32_t B, int32_t T);
int uvc_performer_fwd(const uvc_performer_args* args, void* stream);
int uvc_performer_bwd(const uvc_performer_args* args, void* stream);

/* One Token_performer stage of the tokens-to-token module (T2TViT/models/t2t_vit.py:84-105, token_performer.py:45-69), forward and
 * backward, sequenced in C: soft split + norm1 -> kqv Linear -> linear attention -> v + proj -> norm2 -> MLP (GELU) + residual, and the
 * reverse with every weight / bias / LayerNorm gradient.  The same launches, in the same order and with the same arguments, as the calls
 * to uvc_unfold_ln_*, uvc_gemm_nt / _tn, uvc_performer_*, uvc_layernorm_* they replace (results bit-identical): a stage is one call
 * across the boundary instead of 7 (forward) / 11 (backward).  T = element type of `dtype`; "f32" buffers are float32 in both modes. */
typedef struct uvc_t2t_stage {
  /* geometry of the soft split that feeds the stage */
  const float* src; int64_t sb, sc, sh, sw;
  int32_t B, C, H, W, k, s, p;
  int32_t T, dim, dimp, dtype, training;      /* T = tokens per image after the split, dim = C*k*k, dimp = dim padded to the GEMM's K granularity */
  float eps, beta;                            /* LayerNorm eps; beta = 1: gradients accumulate into the g_* buffers, 0: overwrite */
  int32_t need_dx, reserved;                  /* backward: leave d(unfolded row) in dxu (tap-major, see uvc_unfold_args.dxu_tap_major) */
  /* parameters (float32) and their gradients */
  const float *norm1_w, *norm1_b, *kqv_b, *w, *proj_b, *norm2_w, *norm2_b, *fc1_b, *fc2_b;
  float *g_norm1_w, *g_norm1_b, *g_kqv_w, *g_kqv_b, *g_proj_w, *g_proj_b, *g_norm2_w, *g_norm2_b, *g_fc1_w, *g_fc1_b, *g_fc2_w, *g_fc2_b;
  /* T-typed weight copies W [out, in] and W^T [in, out] */
  const void *kqv_w, *kqv_wt, *proj_w, *proj_wt, *fc1_w, *fc1_wt, *fc2_w, *fc2_wt;
  /* forward buffers: xn [M, dimp] T, kqv [M, 192] f32, att [M, 64] T, x1 [M, 64] f32, h, u, gp [M, 64] T (gp: training only), out [M, 64] f32 */
  void* xn; float* mean1; float* rstd1; float* kqv; float* part; float* kptv; void* att; float* x1; void* h; float* mean2; float* rstd2;
  void* u; void* gp; float* out;
  /* backward: dout [M, 64] T (input), da, dh, dx1, datt [M, 64] T, dkqv [M, 192] T, dkptv [B, 65, 32] f32, dxn [M, dimp] T, scratch */
  const void* dout; void* da; void* dh; void* dx1; void* datt; void* dkqv; float* dkptv; void* dxn;
  float* ln2_partial; float* ln1_partial; float* dxu; void* tn_ws; int64_t tn_ws_bytes;
} uvc_t2t_stage;
int uvc_t2t_stage_forward(const uvc_t2t_stage* st, void* stream);
int uvc_t2t_stage_backward(const uvc_t2t_stage* st, void* stream);

#ifdef __cplusplus
}
#endif
#endif
