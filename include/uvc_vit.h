/* C-ABI of the DeiT (DistilledVisionTransformer) forward / backward sequencer on MI355X.
 *
 * Replaces, for the Stage-1 hot path, what the reference runs through autograd over
 * UVC/models/model_distilled.py:429-531 (forward_features + heads) -- one call enqueues the
 * whole forward (or backward) as a fixed sequence of the kernels of uvc_kernels.h on `stream`.
 * Parameters live in ONE flat float32 buffer (layout from uvc_vit_layout) so the optimiser,
 * the gradient all-reduce and the UVC engine address them without per-tensor launches; the
 * Python module (uvc_amd/model_distilled.py) exposes them under the reference's state_dict
 * names as views.
 *
 * No allocation inside: the caller provides `workspace` (uvc_vit_workspace_bytes) which holds
 * the activations saved for backward and all scratch.
 */
#ifndef UVC_VIT_H
#define UVC_VIT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UVC_VIT_MAX_DEPTH 32

typedef struct uvc_vit_cfg {
  int32_t img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads, hidden;
  int32_t ntok;    /* 1, or 2 with the distillation token (enable_dist) */
  int32_t dtype;   /* UVC_F32 (exact float32 MFMA) or UVC_BF16 */
  float ln_eps;    /* LayerNorm epsilon; <= 0 selects DeiT's 1e-6 (model_distilled.py:402).  T2T-ViT uses nn.LayerNorm's 1e-5 (t2t_vit.py:111) */
  int32_t no_qkv_bias;  /* 1: attn.qkv has no bias (T2T blocks, transformer_block.py:49); its slot in the flat buffers is never read or written */
  int32_t resid_f32;    /* UVC_BF16 only.  0 (default): the residual stream -- the rows x_l every block reads and writes (model_distilled.py:240,244,
                           493) -- is stored as bf16 like every other activation (LayerNorm statistics, the residual additions and the gate mix
                           are float32 arithmetic on the loaded values; one rounding per stored row): the reference's own half-precision mode keeps
                           half activations (joint_train.py:283-287).  1: float32 rows as in rounds 1-2 (A/B runs).  UVC_F32 always has float32 rows */
  int32_t reserved;
} uvc_vit_cfg;

/* element offsets into the flat parameter / gradient buffers (every tensor 16-byte aligned) */
typedef struct uvc_vit_offsets {
  int64_t cls_token, dist_token, pos_embed, patch_w, patch_b;
  /* per block: norm1.w norm1.b qkv.w qkv.b proj.w proj.b norm2.w norm2.b fc1.w fc1.b fc2.w fc2.b */
  int64_t blk[UVC_VIT_MAX_DEPTH][12];
  int64_t norm_w, norm_b, head_w, head_b, headd_w, headd_b;
  int64_t n_main;        /* [0, n_main) always receives gradients in Stage-1 */
  int64_t gate;          /* block_skip_gating [L,2] (no gradient during warm-up) */
  int64_t gumbel_w, gumbel_b;   /* patch-gating scorer (gradient only in patch-gating mode 2) */
  int64_t patch_gating;  /* [P] (mode 1) */
  int64_t skip[UVC_VIT_MAX_DEPTH][2];  /* attn_skip_gating, mlp_skip_gating: never get gradients */
  int64_t n_total;
} uvc_vit_offsets;

/* element offsets (in units of T) into the shadow buffer: cast copies W and transposed copies W^T */
typedef struct uvc_vit_shadow_offsets {
  int64_t patch_w;
  int64_t blk_w[UVC_VIT_MAX_DEPTH][4];    /* qkv proj fc1 fc2   [out,in] */
  int64_t blk_wt[UVC_VIT_MAX_DEPTH][4];   /* transposed          [in,out] */
  int64_t head_w, head_wt, headd_w, headd_wt;
  int64_t n_total;
} uvc_vit_shadow_offsets;

int uvc_vit_layout(const uvc_vit_cfg* cfg, uvc_vit_offsets* off, uvc_vit_shadow_offsets* soff);
/* training: 0 = no-grad forward only; 1 = training, per-block backward streams (two-stream backward); 2 = training, one shared set of backward
 * streams (uvc_vit_io.shared_bwd_streams = 1; no side stream) */
int64_t uvc_vit_workspace_bytes(const uvc_vit_cfg* cfg, int32_t batch, int32_t training);
/* A HIP stream for uvc_vit_io.side_stream with a scheduling class: -1 = lowest priority the device offers (weight
 * gradients / teacher forward that should only fill idle CUs), 0 = default, +1 = highest.  Never destroyed by the
 * library; wrap it with the host framework's external-stream handle. */
int uvc_stream_create(int32_t priority_class, void** out);
int uvc_vit_ws_offsets(const uvc_vit_cfg* cfg, int32_t batch, int32_t training, int64_t* pe_off, int64_t* dpe_off);

/* refresh the T-typed shadows (W and W^T) from the float32 master weights */
int uvc_vit_update_shadows(const uvc_vit_cfg* cfg, const float* params, void* shadow, void* stream);

/* Stage-2 structured sparsity of one block's MLP: the hidden units whose fc1 row and fc2 column are masked
 * out are skipped instead of multiplied as zeros.  `width` (a multiple of 64; padding slots are pruned units, whose weights
 * are zero) compact units; w1 [width, D], w1t [D, width], w2 [D, width], w2t [width, D] of the model's operand type T and
 * b1 [width] float32 are gathered copies (uvc_mlp_gather_shadows); dw1 [width, D], dw2 [D, width], db1 [width] float32 are
 * scratch for the compact weight gradients, which uvc_mlp_scatter_grads expands into the full gradient tensors -- rows of
 * pruned units are exact zeros, and the fc2 gradient column of a pruned unit j is the rank-1 GELU(b1[j]) * db2 that the
 * dense computation produces (its activation is the constant GELU(b1[j])), so global-norm clipping sees the same norm. */
typedef struct uvc_mlp_compact {
  int32_t width; int32_t reserved;
  const void* w1; const void* w1t; const void* w2; const void* w2t; const float* b1;
  float* dw1; float* dw2; float* db1;
  const int32_t* inv;       /* device [F]: compact slot of hidden unit j, or -1 */
} uvc_mlp_compact;

typedef struct uvc_vit_io {
  const float* params;      /* flat float32 parameters */
  void* shadow;             /* flat T shadows */
  float* grads;             /* flat float32 gradients (backward) */
  void* workspace; int64_t workspace_bytes;
  const float* x;           /* [B, C, S, S] float32 */
  float* logits;            /* [B, num_classes] */
  float* logits_dist;       /* [B, num_classes], only with ntok == 2 */
  const float* d_logits;    /* backward inputs */
  const float* d_logits_dist;
  const float* gate_d;      /* device [L,2] block-gate distributions (model_distilled.py:480-488) or NULL */
  const int32_t* run_block; /* HOST [L] 0/1: hard block skip when gate_d is NULL (:496-500); NULL = run all */
  const float* patch_mask;  /* device [B,P] token mask (patch gating) or NULL */
  float* d_patch_mask;      /* backward: optional [B,P] gradient wrt patch_mask */
  int32_t batch;
  int32_t training;         /* forward: keep activations for backward */
  int32_t gate_mode;        /* 0 warm-up / none, 1 soft Gumbel, 2 softL0: selects d(gate logits) formula */
  float gate_eps;           /* softL0 eps */
  float accumulate;         /* backward: 0 = overwrite gradients, 1 = add (gradient accumulation) */
  /* stages [stage_begin, stage_end); 0,0 = everything.
   * backward: 0 = heads + final norm, 1..L = blocks L-1..0, L+1 = gate logits + token assembly,
   *           L+2 = patch-embedding weight gradient.  Lets the host cut the backward at gradient-bucket
   *           boundaries and start the RCCL all-reduce of a finished bucket on a second stream while the
   *           rest of the backward runs (and add the patch-scorer term to dpe before stage L+2).
   * forward:  0 = patch embedding, 1 = everything after it (the patch-gating mask is computed in between). */
  int32_t stage_begin, stage_end;
  /* backward, optional: a second hipStream_t.  The weight-gradient GEMMs (which hang off the dgrad chain) are
   * enqueued there and overlap the chain on `stream`; events order operand reads against buffer reuse, and
   * `stream` waits for the side stream before the call's last stage returns. */
  void* side_stream;
  const uvc_mlp_compact* mlp_compact;  /* HOST [L] or NULL; entries with width 0 or width == hidden run dense */
  const int32_t* head_keep;            /* device [L, H] or NULL: no-grad forwards skip the attention of heads marked 0 (their
                                          attn.proj input columns are masked to zero, so the result is unchanged); training forwards ignore it, backward
                                          uses it when head_keep_bwd is set */
  int32_t full_tail;                   /* 0 (default): the last block that runs computes everything behind its qkv projection on the class /
                                          distillation token rows only -- the only rows of it that reach the head (:507-526), so no output of the
                                          step changes (uvc_attention_tok_*); 1: all rows, as the reference executes it */
  int32_t fused_train_mlp;             /* 1: the training forward runs LayerNorm2 + fc1 (+GELU, GELU') + fc2 (+residual, gate mix) as ONE kernel
                                          (uvc_mlp_fused_fwd's training form, DeiT-Tiny width) instead of three; measured slower (218 us
                                          against 187), so 0 is the default */
  const void* patches_in;              /* optional T [B * np, C * P * P]: the patch rows of `x` already laid out by uvc_patchify (same image size and
                                          patch size).  The forward uses them instead of running uvc_patchify, the backward reads them for the
                                          patch-embedding weight gradient.  Lets student and teacher share the one rearrangement of a batch. */
  int32_t fuse_next_ln;                /* 1: a kernel that produces a block's output rows (uvc_mlp_fused_fwd; fc2 + residual + gate mix) also writes
                                          norm1 of the NEXT block that runs (model_distilled.py:241) from the rows it holds, and that block skips its
                                          stand-alone LayerNorm pass.  0: every LayerNorm is its own pass. */
  int32_t force_generic;               /* tests / A-B runs: passed to every uvc_gemm_nt of the pass (uvc_gemm_nt_args.force_generic: 1 = the generic
                                          LDS-tiled kernel everywhere, 2 = register-staged streaming kernels instead of the LDS-DMA rings);
                                          1 also runs every dgrad + LayerNorm backward as the unfused pair, 2 runs uvc_gemm_nt_lnbwd's
                                          register-staged variant; 3 = the production batch's wide-tile kernels at any row count.  0 = kernels
                                          picked by shape */
  int32_t head_keep_bwd;               /* 1: uvc_vit_backward skips dq / dk / dv of the heads head_keep marks 0 (written as zeros).  Exact when the
                                          64 attn.proj input columns of such a head are zero IN THE WEIGHTS the forward and the dgrad used (Stage-2:
                                          post_train.py:343-346 multiplies weight by mask before every step): dL/d(attention output) of the head is then
                                          exactly zero and so are its dq, dk, dv.  The forward still computes the head (dW_proj of the masked columns
                                          needs its output: the reference's clip norm sees it).  0: every head's backward runs */
  int32_t shared_bwd_streams;          /* 0 (default): the workspace (uvc_vit_workspace_bytes(.., training = 1)) holds per-block copies of the backward's
                                          streams dL/dx_l, dL/dx1, dA, dqkv, so that weight gradients on `side_stream` can read a block's streams while the
                                          main stream is blocks ahead (L x (5 M D + M F) elements more).  1: the workspace was sized with training = 2 --
                                          ONE shared set (dL/dx ping-pongs between two buffers); forward and backward of a step must agree, and the
                                          backward must run without a side stream (UVC_ERR_ARG otherwise).  Same results bit for bit. */
  int32_t gelu_grad_bf16;              /* 0 (default): where uvc_gemm_nt_q8_supported (bf16, embed_dim 192, hidden % 256 == 0, >= 4096 rows) the training forward
                                          leaves GELU'(a) of fc1 as ONE byte per activation (UVC_EPI_BIAS_GELU_GRAD_Q8: a uniform code over [-0.13, 1.13], |error|
                                          <= 2.47e-3 -- as accurate as bf16 on this bounded quantity, include/uvc_kernels.h) and the dgrad of fc2 decodes it
                                          (UVC_EPI_MUL_AUX_Q8): fc1 + GELU, GELU' is write-bound, 77 MB per block less at DeiT-Tiny batch 512.  1: bf16 GELU'(a)
                                          everywhere, as before r6 (A/B runs, tests).  Forward and backward of a step must agree. */
} uvc_vit_io;

int uvc_vit_forward(const uvc_vit_cfg* cfg, const uvc_vit_io* io, void* stream);
int uvc_vit_backward(const uvc_vit_cfg* cfg, const uvc_vit_io* io, void* stream);

#ifdef __cplusplus
}
#endif
#endif
