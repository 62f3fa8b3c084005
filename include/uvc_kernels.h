/* C-ABI of the DeiT Stage-1 compute kernels on MI355X (libuvc_hip.so): MFMA GEMMs with fused
 * epilogues, LayerNorm, fused attention, distillation loss, fused clip+AdamW, token/gate helpers.
 *
 * The reference reaches these through ATen/cuBLAS/cuDNN from UVC/models/model_distilled.py,
 * UVC/utils/losses.py and torch.optim; it has no operator/FFI layer of
 * its own, so each entry point cites the reference lines whose arithmetic it performs.
 *
 * Conventions as in uvc_engine.h: device pointers owned by the caller, no allocation, no host
 * sync, `stream` is a hipStream_t, int status return (0 = ok; uvc_last_error()).
 * dtype selects the arithmetic: UVC_F32 = float32 storage + v_mfma_f32_16x16x4_f32 (exact fmaf
 * chain; the 1e-3 parity mode), UVC_BF16 = bfloat16 operands + v_mfma_f32_16x16x32_bf16 with
 * float32 accumulation (the throughput mode).  "T" below means float or bfloat16 accordingly.
 */
#ifndef UVC_KERNELS_H
#define UVC_KERNELS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { UVC_F32 = 0, UVC_BF16 = 1 };

enum {
  UVC_EPI_NONE = 0,            /* C = alpha*acc                                                        */
  UVC_EPI_BIAS = 1,            /* C = alpha*acc + bias[n]                       (nn.Linear)            */
  UVC_EPI_BIAS_GELU = 2,       /* C = a = acc + bias ; C2 = GELU_erf(a)         (Mlp fc1+act, :116-118) */
  UVC_EPI_BIAS_RESID = 3,      /* C = acc + bias + R[m,n]                       (x + proj(...), :240)   */
  UVC_EPI_BIAS_RESID_GATE = 4, /* C = g1*(acc + bias + R[m,n]) + g0*R2[m,n]     (:244 then :493)        */
  UVC_EPI_DGELU = 5,           /* C = alpha*acc * GELU'(aux[m,n])               (backward of :118)      */
  UVC_EPI_BIAS_GELU_OUT = 6,   /* C = GELU_erf(acc + bias)   (inference: pre-activation not kept)        */
  UVC_EPI_BIAS_GELU_GRAD = 7,  /* a = acc + bias ; C = GELU'(a) ; C2 = GELU_erf(a): training forward of fc1 -- the backward
                                  needs the pre-activation only through GELU'(a), so that is what is kept            */
  UVC_EPI_MUL_AUX = 8,         /* C = alpha*acc * aux[m,n]   (backward of the activation with aux = stored GELU'(a)) */
  /* r6: GELU'(a) in ONE byte.  The backward needs the fc1 pre-activation only through GELU'(a), a BOUNDED quantity ([-0.1290, 1.1290]), and fc1 + GELU, GELU'
   * is write-bound (profiles/r6b_bound_probes.txt: the step is 3.2 % faster with that tensor not written at all).  Uniform code over [UVC_Q8_LO, UVC_Q8_LO +
   * 255 UVC_Q8_STEP] = [-0.13, 1.13]:  q = min(255, trunc(max(0, (GELU' - LO) / STEP + 0.5))),  GELU' ~ LO + q STEP,  |error| <= STEP / 2 = 2.47e-3 everywhere --
   * finer than bf16 where GELU' >= 0.5 (bf16 spacing 3.9e-3 on [0.5, 1), 7.8e-3 on [1, 2)), coarser only where GELU' is small and weighs least; relative L2
   * error of dA / dW1 / dh against float64 2.0e-3 - 2.6e-3 against 1.6e-3 - 2.0e-3 for bf16 GELU' (tools/gelu_grad_codes.py, profiles/r6_gelu_grad_codes_cpu.txt).
   * bf16 compute, K = 192, N % 256 == 0, M >= 4096 (uvc_gemm_nt_q8_supported): the streaming kernel of DeiT-Tiny's width; UVC_ERR_UNSUPPORTED elsewhere. */
  UVC_EPI_BIAS_GELU_GRAD_Q8 = 9,  /* a = acc + bias ; C [M, N] BYTES (ldc in bytes = elements) = code(GELU'(a)) ; C2 = GELU_erf(a) (T)          */
  UVC_EPI_MUL_AUX_Q8 = 10         /* C = alpha*acc * (LO + STEP * aux[m,n]),  aux [M, N] bytes (ldaux in elements) written by _GELU_GRAD_Q8      */
};
#define UVC_Q8_LO (-0.13f)
#define UVC_Q8_STEP (1.26f / 255.0f)

/* C[M,N] = epilogue( A[M,K] . B[N,K]^T ); A, B row-major with K contiguous. */
typedef struct uvc_gemm_nt_args {
  const void* A;       /* [M,K]  T, or float32 when a_is_f32 (converted to T while staging) */
  const void* B;       /* [N,K]  T */
  void* C;             /* [M,N]  T, or float32 when c_is_f32 */
  void* C2;            /* second output of UVC_EPI_BIAS_GELU / _GELU_GRAD, same type/ld as C */
  const float* bias;   /* [N] */
  const void* R;       /* [M,N] residual rows, element type of C (float32 stream: c_is_f32; bf16 stream: C, R, R2 all bf16) */
  const void* R2;      /* [M,N] gate epilogue, same type */
  const void* aux;     /* [M,N] T: pre-activation for UVC_EPI_DGELU, multiplier for UVC_EPI_MUL_AUX */
  const float* gate;   /* device float[2] = (g0, g1) block-gate distribution */
  const float* alpha_ptr; /* optional device scalar multiplied into alpha */
  float alpha;
  int32_t M, N, K, lda, ldb, ldc, ldr, ldaux;
  int32_t dtype, a_is_f32, c_is_f32, epilogue;
  int32_t force_generic;  /* tests/tuning: 0 = pick the kernel by shape; 1 = the generic LDS-tiled kernel; 2 = the streaming kernels in their
                             register-staged forms (no LDS-DMA ring; the 256 x 256-tile kernel IS an LDS-DMA kernel and is skipped too);
                             3 = the wide-tile kernels (256 x 256) wherever the shape admits them, whatever the row / tile count -- what
                             the production batch of DeiT-Small / Base selects, at a size the host oracle finishes in seconds.  Every
                             choice computes the same bits (one k-ordered accumulation chain, the same epilogue arithmetic) */
  int32_t r_is_f32;       /* element type of R / R2, stated by the caller and checked: they have C's type (c_is_f32, or float32 mode), so
                             this must equal it whenever R is given.  Rounds 1-2 took float32 R beside a bf16 C; a caller still built to
                             that contract gets UVC_ERR_ARG instead of its rows read as bf16 */
  /* optional second output of the residual epilogues where uvc_gemm_nt_ln_supported(): ln_out[M,N] (T) = LayerNorm(C rows; ln_gamma,
   * ln_beta, ln_eps) -- norm1 of the next block on the rows fc2 + residual (+ gate mix) just produced (model_distilled.py:241-244) --
   * and its float32 statistics ln_mean / ln_rstd [M] (both or neither).  Needs alpha == 1, contiguous A / R (lda == K, ldr == N).
   * With a bf16 C the LayerNorm is taken of the ROUNDED rows (what C holds). */
  float ln_eps;
  const float* ln_gamma; const float* ln_beta; void* ln_out; float* ln_mean; float* ln_rstd;
} uvc_gemm_nt_args;
int uvc_gemm_nt(const uvc_gemm_nt_args* args, void* stream);
int uvc_gemm_nt_ln_supported(int32_t M, int32_t N, int32_t K, int32_t dtype, int32_t epilogue);
/* 1 where UVC_EPI_BIAS_GELU_GRAD_Q8 (N = hidden width, K = embed_dim) AND its partner UVC_EPI_MUL_AUX_Q8 (N and K swapped roles: [M, K'] x [N', K'] with
 * K' = embed_dim) exist: bf16, embed_dim 192, hidden % 256 == 0, M >= 4096.  Callers fall back to UVC_EPI_BIAS_GELU_GRAD / UVC_EPI_MUL_AUX elsewhere. */
int uvc_gemm_nt_q8_supported(int32_t M, int32_t hidden, int32_t embed_dim, int32_t dtype);

/* C[N1,N2] = beta*C + alpha * A[M,N1]^T . B[M,N2]  (weight gradients; float32 C).
 * Deterministic: M is cut into slices reduced in a fixed order through `workspace`. */
typedef struct uvc_gemm_tn_args {
  const void* A;       /* [M,N1] T, or float32 when a_is_f32 */
  const void* B;       /* [M,N2] T */
  float* C;            /* [N1,N2] float32 */
  void* workspace;     /* >= uvc_gemm_tn_workspace_bytes() */
  int64_t workspace_bytes;
  const float* alpha_ptr;
  float* colsum_out;   /* optional [N1]: beta*old + alpha * column sums of A (the bias gradient, same pass) */
  float alpha, beta;
  int32_t M, N1, N2, lda, ldb, ldc;
  int32_t dtype, a_is_f32;
  int32_t variant;     /* tuning / A-B: 0 (and 1) = kernel picked by shape -- since r6 the two-group schedule (k_gemm_tn8p) for every 192- / 256-wide tile but
                          the 96 x 192 one; 2 = the LDS-DMA ring kernel (k_gemm_tn_dma) for the 192 x 192, 192 x 256 and 256 x 192 tiles.  All of them write
                          the same bits.  3 = 128 x 256 tiles where the shape takes 256 x 256 (half the splits: another summation order) */
} uvc_gemm_tn_args;
int uvc_gemm_tn(const uvc_gemm_tn_args* args, void* stream);
int uvc_gemm_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2, int64_t* bytes, int32_t* splits);

/* Fused softmax(q k^T * scale) v for one DeiT sequence per (batch, head)
 * (UVC/models/model_distilled.py:175-185).  qkv is the packed Linear(dim, 3*dim) output
 * [B, N, 3, H, 64]; o is [B, N, H*64]; lse/delta are float32 [B, H, N].  N <= 256, head_dim = 64. */
typedef struct uvc_attn_args {
  const void* qkv;   /* T */
  void* o;           /* T  (forward output; backward input) */
  float* lse;        /* log-sum-exp of the scaled scores (forward output; backward input) */
  const void* dout;  /* T  [B,N,H*64]      backward only */
  void* dqkv;        /* T  [B,N,3,H,64]    backward output */
  float* delta;      /* scratch [B,H,N]    backward only */
  int32_t B, N, H, head_dim, dtype;
  float scale;
  const int32_t* head_keep;  /* optional device [H].  Forward: heads with 0 are skipped and their slice of o is written as zeros.
                                For inference on a pruned model whose attn.proj input columns of that head are all zero (masked),
                                which makes the skip exact.  Not for training: the reference's clip norm includes the gradients of
                                masked proj columns, which need the head's output.
                                Backward: dq / dk / dv of heads with 0 are written as zeros without being computed -- exact when dout is
                                zero on the head's 64 columns, i.e. when attn.proj's input columns of that head are zero in the weights
                                the dgrad used (Stage-2 masked fine-tuning) */
  int32_t variant;   /* backward, tuning / A-B: 0 = kernel picked by shape (bf16, 193 <= N <= 200, B * H >= 1024: the one-pass persistent kernel
                        k_attn_bwd_one, otherwise the dq + dk/dv pair); 1 = the pair; 2 = the one-pass kernel or UVC_ERR_UNSUPPORTED */
  int32_t grid;      /* backward, one-pass kernel: 0 = one persistent workgroup per CU; > 0 = that many (tests: several heads per workgroup
                        at small B * H) */
} uvc_attn_args;
int uvc_attention_fwd(const uvc_attn_args* args, void* stream);
int uvc_attention_bwd(const uvc_attn_args* args, void* stream);

/* The qkv Linear and the attention forward of a block as ONE kernel (UVC/models/model_distilled.py:175-185: `self.qkv(x)` ... `attn @ v`):
 * h [B*N, D] (the LayerNorm-1 rows) and the qkv weight [3D, D] in, o [B,N,H*64] and lse out; qkv [B,N,3,H,64] is written only when the
 * pointer is given (the backward needs it; a no-grad forward does not).  Same bits as uvc_gemm_nt (UVC_EPI_BIAS) + uvc_attention_fwd.
 * Where uvc_qkv_attention_supported() (bf16, D = 64 H = 192 or 384, 193 <= N <= 208). */
typedef struct uvc_qkv_attn_args {
  const void* h;       /* T [B*N, D] */
  const void* w;       /* T [3D, D]   qkv weight, out-major (the bf16 shadow) */
  const float* bias;   /* [3D] or NULL */
  void* qkv;           /* T [B,N,3,H,64] or NULL */
  void* o;             /* T [B,N,H*64] */
  float* lse;          /* [B,H,N] or NULL */
  int32_t B, N, H, D, dtype;
  float scale;
  int32_t grid;        /* 0 = one persistent workgroup per CU (a workgroup takes (image, group of 3 heads) items); > 0 = that many (tests) */
} uvc_qkv_attn_args;
int uvc_qkv_attention_supported(int32_t B, int32_t N, int32_t H, int32_t D, int32_t dtype);
int uvc_qkv_attention_fwd(const uvc_qkv_attn_args* args, void* stream);

/* Attention of the first `ntok` query rows (class / distillation token) of every (image, head) against all N keys: what the LAST
 * block needs, whose other rows never reach the head (UVC/models/model_distilled.py:175-185 for rows 0 .. ntok-1, :507-526).
 * o / dout are the compact [B, ntok, H*64] rows; the backward recomputes the probabilities and writes the WHOLE dqkv
 * [B, N, 3, H, 64] (dq = 0 for the other rows, dk / dv dense).  ntok is 1 or 2, N <= 256, head_dim = 64. */
typedef struct uvc_attn_tok_args {
  const void* qkv;   /* T [B, N, 3, H, 64] */
  void* o;           /* T [B, ntok, H*64]  forward output */
  const void* dout;  /* T [B, ntok, H*64]  backward only */
  void* dqkv;        /* T [B, N, 3, H, 64] backward output */
  int32_t B, N, H, head_dim, ntok, dtype;
  float scale;
  const int32_t* head_keep;  /* optional device [H], forward and backward as in uvc_attn_args */
} uvc_attn_tok_args;
int uvc_attention_tok_fwd(const uvc_attn_tok_args* args, void* stream);
int uvc_attention_tok_bwd(const uvc_attn_tok_args* args, void* stream);

/* Copies `groups` blocks of `group_bytes` bytes: block g from src + g * src_group_stride to dst + g * dst_group_stride (all
 * multiples of 16 bytes).  Gathers the token rows of a [B, N, D] tensor into [B, ntok, D] (and scatters them back). */
int uvc_copy_row_groups(const void* src, void* dst, int64_t groups, int64_t group_bytes, int64_t src_group_stride, int64_t dst_group_stride,
                        void* stream);

/* nn.LayerNorm(D, eps) forward over `rows` rows (model_distilled.py:199,204,288; eps=1e-6 from
 * joint_train.py:138).  Row r of x starts at x + (r / rows_per_group) * group_stride + (r % rows_per_group) * D
 * (dense: rows_per_group = 1, group_stride = D; class/dist-token rows only: group_stride = N*D).
 * y is dense [rows, D] of type T (or float32 when y_is_f32); mean/rstd float32 [rows]. */
typedef struct uvc_ln_args {
  const void* x;        /* float32, or bf16 with x_lowp (the bf16 residual stream of the throughput mode) */
  const float* gamma; const float* beta;
  void* y; float* mean; float* rstd;
  /* backward */
  const void* dy;       /* dense [rows, D]; T, or float32 when dy_is_f32 */
  void* dx;             /* same row addressing as x; dx = LN'(dy) + a1*add1 + a2*add2; float32, or T when g_lowp */
  const void* add1; const float* a1;    /* optional dense-as-x addends (same element type as dx); a1/a2 device scalars (NULL = 1) */
  const void* add2; const float* a2;
  float* partial;       /* scratch [ln_bwd_blocks, 2*D + 2] */
  float* dgamma; float* dbeta;          /* [D], written as beta_acc*old + sum */
  float* dots;          /* optional [2]: { <dx, x>, <add2, x> } over all rows (gate-logit gradients) */
  float eps, beta_acc;
  int32_t rows, D, rows_per_group, dtype, y_is_f32, dy_is_f32;
  int64_t group_stride;
  int32_t g_lowp;       /* backward: the gradient stream (dx, add1, add2) is stored as T (bf16) instead of float32;
                           all arithmetic and the dots stay float32 */
  int32_t defer_reduce; /* backward: leave the per-block partials of dgamma / dbeta / dots in `partial` (one private region per
                           call) and skip the two small reduction launches; uvc_layernorm_bwd_reduce_batch finishes many calls at once */
  int32_t x_lowp;       /* x is stored as bf16 (statistics, normalisation and every sum stay float32); D % 64 == 0 only */
  int32_t reserved;
} uvc_ln_args;
/* one deferred uvc_layernorm_bwd call: its partial region and outputs (dots may be NULL) */
typedef struct uvc_ln_reduce_item { const float* partial; float* dgamma; float* dbeta; float* dots; int32_t nblocks; int32_t reserved; } uvc_ln_reduce_item;
/* finish up to 64 deferred LayerNorm backward calls (same D) in ONE launch; sums in a fixed order (deterministic);
 * outputs written as beta_acc*old + sum.  `items` is a HOST array. */
int uvc_layernorm_bwd_reduce_batch(const uvc_ln_reduce_item* items, int32_t n, int32_t D, float beta_acc, void* stream);
int uvc_layernorm_fwd(const uvc_ln_args* args, void* stream);
int uvc_layernorm_bwd(const uvc_ln_args* args, void* stream);
int uvc_layernorm_bwd_blocks(int32_t rows);
int uvc_layernorm_bwd_nblocks(int32_t rows);

/* dgrad GEMM with the LayerNorm backward as its epilogue (bf16 mode, D == 192, K in {576, 768}, M >= 4096): the backward of
 * `Linear(LayerNorm(x))` w.r.t. x, i.e. the pair uvc_gemm_nt(A, W -> dy) + uvc_layernorm_bwd(dy, x, ...) of
 * model_distilled.py:199-204,218-247's autograd without the [M, D] dy round trip through HBM:
 *     dy = A[M,K] . W[D,K]^T ;  dx = LN'(dy; x, mean, rstd, gamma) + a1*add1 + a2*add2   (dx, add1, add2: bf16 [M, D]; dx may alias add2)
 * dgamma / dbeta / { <dx,x>, <add2,x> } partials go to partial[uvc_gemm_lnbwd_nblocks(M)][2*D+2] (float32) and are finished by
 * uvc_layernorm_bwd_reduce_batch like a deferred uvc_layernorm_bwd call.  Deterministic (fixed summation orders). */
typedef struct uvc_gemm_lnbwd_args {
  const void* A; const void* W;                 /* bf16 [M,K], bf16 [D,K] (the W^T shadow of the Linear) */
  const void* x; const float* mean; const float* rstd; const float* gamma;    /* LayerNorm input [M,D] (float32, or bf16 with x_lowp) and saved statistics */
  const void* add1; const float* a1; const void* add2; const float* a2;       /* optional bf16 addends, device scalars (NULL = 1) */
  void* dx; float* partial;
  int32_t M, D, K, dtype;
  int32_t variant;      /* tests/tuning: 0 = LDS-DMA ring where the shape allows, 1 = the register-staged kernel */
  int32_t x_lowp;       /* 1: x is bf16 [M,D] (bf16 residual stream) instead of float32 */
} uvc_gemm_lnbwd_args;
int uvc_gemm_lnbwd_supported(int32_t M, int32_t D, int32_t K, int32_t dtype);
int uvc_gemm_lnbwd_nblocks(int32_t M);
int uvc_gemm_nt_lnbwd(const uvc_gemm_lnbwd_args* args, void* stream);

/* Fused MLP half of a block (model_distilled.py:107-124,199-204,241-247,493), bf16 mode, D == 192, F % 64 == 0:
 *     out = d1 * (x + fc2(GELU(fc1(LayerNorm(x))))) + d0 * x_prev           (gate = {d0, d1} device pair; NULL: out = x + mlp)
 * x, out, x_prev float32 [M, D]; w1 [F, D], w2 [D, F] are the bf16 weight shadows in their natural layouts; gamma/beta/b1/b2 float32.
 * Inference (teacher: utils/losses.py:47-49; eval): h = mean = rstd = gp = u = NULL, the [M, F] hidden activation never reaches memory.
 * Training: all five given -- the same pass stores what the backward reads: h = LayerNorm(x) (bf16 [M, D]), mean / rstd [M],
 * gp = GELU'(a) and u = GELU(a) (bf16 [M, F]); it replaces uvc_layernorm_fwd + two uvc_gemm_nt launches of the step. */
typedef struct uvc_mlp_args {
  const void* x; const float* gamma; const float* beta;
  const void* w1; const float* b1; const void* w2; const float* b2;
  void* out;
  int32_t M, D, F;
  float eps;
  const void* x_prev; const float* gate;
  void* h; float* mean; float* rstd; void* gp; void* u;
  /* optional: LayerNorm of the OUTPUT rows with the next block's norm1 parameters (model_distilled.py:241 of block l+1),
   * written as compute-dtype rows next_h [M,D] (+ float32 next_mean / next_rstd [M] for a backward) */
  const float* next_gamma; const float* next_beta; void* next_h; float* next_mean; float* next_rstd;
  int32_t rows_lowp;    /* 1: x, out and x_prev are bf16 rows (bf16 residual stream; out is rounded once, next_h is the LayerNorm of the rounded rows) */
  int32_t reserved;
} uvc_mlp_args;
int uvc_mlp_fused_supported(int32_t D, int32_t F, int32_t dtype);
int uvc_mlp_fused_fwd(const uvc_mlp_args* args, void* stream);

/* DistillationLoss over SoftTargetCrossEntropy (UVC/utils/losses.py:25-65, joint_train.py:940):
 * loss = (1-alpha) * mean_b sum_c -y log_softmax(o) + alpha * KL(softmax(t/T) || softmax(o_kd/T)) * T^2 / (B*C)   (kind 1, 'soft')
 *      = (1-alpha) * base + alpha * mean_b CE(o_kd, argmax_c teacher)                                            (kind 2, 'hard', :61-62)
 * All float32 [B,C].  Writes loss[0] and the gradients d_o, d_okd (d_okd may alias d_o when
 * o_kd == o, i.e. enable_deit = 0: the two contributions are summed). */
typedef struct uvc_loss_args {
  const float* o; const float* o_kd; const float* y_soft; const float* teacher;
  float* loss; float* d_o; float* d_okd; float* row_scratch; /* [B] */
  float alpha, tau;
  int32_t B, C, kind; /* kind: 0 none, 1 soft, 2 hard */
} uvc_loss_args;
int uvc_distill_loss(const uvc_loss_args* args, void* stream);

/* clip_grad_norm_(max_norm) + AdamW over flat float32 buffers (joint_train.py:428-429;
 * torch.optim.AdamW(lr, betas, eps, weight_decay) semantics, decoupled decay).
 * uvc_grad_sqnorm accumulates sum(g^2) of a segment into sq[0] (accumulate = 0 for the first segment) and leaves sqrt(sq[0]) -- the
 * total norm so far, clip_grad_norm_'s return value -- in sq[1]: sq is float[2];
 * uvc_adamw_step applies clip coefficient min(1, max_norm/(sqrt(sq[0])+1e-6)) read on the device. */
int uvc_grad_sqnorm(const float* g, int64_t n, float* partial /*[1024]*/, float* sq /*[2]*/, int32_t accumulate, void* stream);
typedef struct uvc_adamw_args {
  float* p; const float* g; float* m; float* v;
  void* p_shadow;        /* optional bf16 copy of p (GEMM operands), same layout */
  const float* sq;       /* device: total squared grad norm */
  float* gnorm_out;      /* optional device float: total norm (pre-clip) */
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay, max_norm;
  int32_t step;          /* this segment's 1-based step count (bias correction) */
  const uint8_t* flags;  /* optional device [n]: bit 0 = weight decay applies to this element (timm add_weight_decay groups,
                            post_train.py:299), bit 1 = frozen: leave p/m/v untouched (parameter whose .grad is None) */
} uvc_adamw_args;
int uvc_adamw_step(const uvc_adamw_args* args, void* stream);
/* g *= clip coefficient (so later readers see what clip_grad_norm_ left in .grad; uvc_optimizer.py:90 reads it). */
int uvc_scale_by_clip(float* g, int64_t n, const float* sq, float max_norm, void* stream);

/* misc elementwise / layout kernels */
/* im2col of 16x16/stride-16 patches (PatchEmbed conv as a GEMM, model_distilled.py:142-151):
 * x [B,C,S,S] float32 -> out [B*(S/P)^2, C*P*P] T, k = c*P*P + ky*P + kx. */
int uvc_patchify(const float* x, void* out, int32_t B, int32_t C, int32_t S, int32_t P, int32_t dtype, void* stream);
/* tokens = cat(cls[, dist], patches * mask) + pos  (model_distilled.py:434-471).
 * pe [B,P,D] float32; row_mask optional [B,P] (patch gating), tok [B,N,D] float32. */
int uvc_assemble_tokens(const float* pe, const float* cls, const float* dist, const float* pos, const float* row_mask,
                        void* tok, int32_t B, int32_t P, int32_t D, int32_t ntok, int32_t tok_lowp /* 1: tok is bf16 */, void* stream);
/* backward of uvc_assemble_tokens: dpe [B,P,D] (T or f32) = dtok rows * mask; dpos/dcls/ddist = sums over batch
 * (written as beta_acc*old + sum); optional dmask[B,P] = <dtok row, pe row>.  dtok [B,N,D] is float32, or T when
 * dtok_lowp (the bf16 gradient stream of the backward). */
int uvc_assemble_tokens_bwd(const void* dtok, const float* pe, const float* row_mask, void* dpe, float* dpos, float* dcls,
                            float* ddist, float* dmask, int32_t B, int32_t P, int32_t D, int32_t ntok, int32_t dtype,
                            int32_t dpe_is_f32, int32_t dtok_lowp, float beta_acc, void* stream);
/* column sums: out[n] = beta*out[n] + alpha * sum_m X[m,n];  X is T or float32.  partial: [uvc_colsum_blocks(M), N]. */
int uvc_colsum(const void* X, int32_t M, int32_t N, int32_t ldx, int32_t dtype, int32_t x_is_f32, float* partial, float* out,
               float alpha, const float* alpha_ptr, float beta, const float* row_weight /* optional [M] */, void* stream);
/* patch gating (model_distilled.py:434-456, :36-63).
 * mode 1: mask[b,i] = sigmoid(patch_gating[i]) (or the hard >= .5 threshold with token 0 kept); backward sums over the batch.
 * mode 2: scores = Linear(D->1)(patch embedding); mask = straight-through Gumbel top-k of log_softmax(scores)
 *         with k = int(ratio*P), tau, Exp(1) draws e[B,P]; token 0 forced to 1.  ysoft/psoft [B,P] are kept for backward. */
int uvc_patch_gate_sigmoid(const float* pg, float* mask, int32_t B, int32_t P, int32_t hard, void* stream);
int uvc_patch_gate_sigmoid_bwd(const float* pg, const float* dmask, float* dpg, int32_t B, int32_t P, float beta_acc, void* stream);
int uvc_patch_scores(const float* pe, const float* w, const float* bias, float* scores, int32_t rows, int32_t D, void* stream);
int uvc_patch_topk_mask(const float* scores, const float* e, float* mask, float* ysoft, float* psoft, int32_t B, int32_t P, int32_t k,
                        float tau, void* stream);
int uvc_patch_topk_mask_bwd(const float* dmask, const float* ysoft, const float* psoft, float* dscores, int32_t B, int32_t P, float tau,
                            void* stream);
/* X[row,:] += row_weight[row] * w[:]   (X of type T, or float32 when x_is_f32) */
int uvc_add_outer(void* X, const float* row_weight, const float* w, int32_t rows, int32_t D, int32_t dtype, int32_t x_is_f32, void* stream);
int uvc_colsum_blocks(int32_t M);
/* timm.data.Mixup in "batch" mode (joint_train.py:409,924-933; post_train.py:362,618-621), in place on the device:
 * uvc_mixup_batch: x [B,C,H,W] float32, B even; mixup: x = x*lam + x.flip(0)*(1-lam) (both factors passed already rounded
 * to float32); cutmix: x[:,:,yl:yh,xl:xh] = x.flip(0)[:,:,yl:yh,xl:xh].
 * uvc_mixup_target: y [B,C] = onehot(t)*lam + onehot(t.flip(0))*(1-lam) with on/off values of label smoothing. */
int uvc_mixup_batch(float* x, int32_t B, int32_t C, int32_t H, int32_t W, float lam, float one_minus_lam, int32_t use_cutmix,
                    int32_t yl, int32_t yh, int32_t xl, int32_t xh, void* stream);
int uvc_mixup_target(const int64_t* labels, float* y, int32_t B, int32_t C, float lam, float one_minus_lam, float on_value,
                     float off_value, void* stream);
/* MLP compaction helpers (uvc_vit.h: uvc_mlp_compact).  idx [width]: hidden unit of each compact slot; inv [F]: slot or -1.
 * gather: w1c[s,:] = W1[idx[s],:], w1t = w1c^T, w2c[:,s] = W2[:,idx[s]], w2t = w2c^T (cast to T), b1c[s] = b1[idx[s]].
 * scatter: dW1[j,:] = dw1c[inv[j],:] or 0; db1[j] = db1c[inv[j]] or 0; dW2[:,j] = dw2c[:,inv[j]] or GELU(b1[j]) * db2[:]
 * (GELU as the forward of that precision mode evaluates and stores it); written as beta_acc*old + value. */
int uvc_mlp_gather_shadows(const float* W1, const float* b1, const float* W2, const int32_t* idx, int32_t D, int32_t F, int32_t width,
                           void* w1c, void* w1t, void* w2c, void* w2t, float* b1c, int32_t dtype, void* stream);
int uvc_mlp_scatter_grads(const float* dw1c, const float* dw2c, const float* db1c, const int32_t* inv, const float* b1, const float* db2,
                          int32_t D, int32_t F, int32_t width, float* dW1, float* dW2, float* db1, float beta_acc, int32_t dtype, void* stream);
/* params[i] *= mask[i] over a flat parameter buffer: the Stage-2 `m.weight.data *= m.mask` for every module with a
 * mask buffer (post_train.py:343-346); mask is 1 where no module mask covers the element (biases, tokens). */
int uvc_apply_masks(float* params, const float* mask, int64_t n, void* stream);
/* float32 -> bf16 copy and transposed copy of a [R,C] matrix (weight shadows for the GEMMs). */
int uvc_cast_transpose(const float* W, int32_t R, int32_t C, void* w_bf16, void* wt, int32_t dtype, void* stream);
/* the same for up to 64 matrices in one launch: srcs[i] = element offset into params, ws[i]/wts[i] = element
 * offsets (units of T) into shadow for the cast / transposed copy, -1 = skip.  Host arrays. */
int uvc_cast_transpose_multi(const float* params, void* shadow, int32_t n, const int64_t* srcs, const int32_t* Rs, const int32_t* Cs,
                             const int64_t* ws, const int64_t* wts, int32_t dtype, void* stream);
/* Keyed Exp(1) noise for the Gumbel draws (model_distilled.py:40,485; uvc_utils.py:443-449): out[i] is a pure function of
 * (seed, step, site, i) -- counter-based, no generator state, identical on every data-parallel replica and after a resume. */
int uvc_exp_noise(float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t site, void* stream);

/* block-gate distributions (model_distilled.py:480-488): d[L,2] from g[L,2] and Exp(1) draws. */
int uvc_gate_distrib(const float* g, const float* e, float* d, int32_t L, int32_t mode, float eps, void* stream);
/* gradient of the gate logits from the dot products the LayerNorm-backward kernels leave behind
 * (DESIGN.md): dots [L+1,2], row l = { <dL/dx_l, x_l>, <dL/dout_l, x_l> } (row L: final norm), so
 * <dL/dout_l, out_l> = dots[l+1][0]; dg written as beta_acc*old + grad. */
int uvc_gate_grad(const float* g, const float* d, const float* dots, float* dg, int32_t L, int32_t mode, float eps,
                  float beta_acc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
