/* C-ABI of the UVC primal-dual engine on MI355X (libuvc_hip.so).
 *
 * Drop-in boundary for the reference's per-step primal-dual update.  The reference has no FFI
 * layer -- the boundary is the Python call surface of UVC/uvc_optimizer.py and UVC/uvc_utils.py
 * (SURVEY.md §8b); each entry point below cites the reference function it replaces.  The
 * Python mirror that binds these through ctypes is uvc_amd/uvc_utils.py + uvc_amd/uvc_optimizer.py;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller unless stated; no
 * allocation, no ownership transfer, no host synchronisation inside; kernels are enqueued on
 * `stream` (a hipStream_t passed as void*); return 0 on success, non-zero on error
 * (uvc_last_error() gives the message); nothing throws across the ABI.
 */
#ifndef UVC_ENGINE_H
#define UVC_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct uvc_dims {
  int32_t L;   /* number of transformer blocks (len(uvc_layers["W1"])) */
  int32_t H;   /* heads */
  int32_t hd;  /* head size (args.head_size) */
  int32_t D;   /* embed dim = H*hd = attn.proj in_features */
  int32_t F;   /* mlp hidden = mlp.fc2 in_features */
} uvc_dims;

/* Hyper-parameters read by uvc_optimizer (UVC/uvc_optimizer.py:37-144; argparse defaults
 * UVC/joint_train.py:747-853). */
typedef struct uvc_hyper {
  float budget, slr, rlr, glr, ylr, plr, zlr, sl2wd, z_grad_clip, gating_weight, eps;
  int32_t gating_interval, use_gumbel, enable_block_gating;
} uvc_hyper;

/* Device-resident state of UVC_CP_MiniMax (UVC/uvc_utils.py:129-169) + the SGD state of the
 * gating optimiser (uvc_optimizer.py:251-255) + scratch produced by the score/rank kernels. */
typedef struct uvc_state {
  float* s;          /* [L,2]  */
  float* r;          /* [L,H]  */
  float* y;          /* [L,2]  */
  float* p;          /* [L,H]  */
  float* z;          /* [1]    */
  float* gate;       /* [L,2] model.block_skip_gating.data, or NULL */
  const float* gate_grad; /* [L,2] its task-loss gradient, NULL in warm-up */
  float* gate_momentum;   /* [L,2] SGD momentum buffer */
  float* gate_gsum;       /* [L,2] running weighted sum of gating grads (gating_grad_list) */
  int32_t* gate_counters; /* [2]: {len(gating_grad_list), momentum-initialised flag} */
  const float* total_macs;/* [L,6] float32 MAC table (uvc_utils.py:413) */
  float embed_macs;
  float resource_ub;      /* full-model FLOPs (uvc_optimizer.py:178-187) */
  /* scores of the CURRENT (post-prox) weights and their ranks */
  const float* scores1;   /* [L,D] per-column  */
  const float* scores2;   /* [L,H] per-head    */
  const float* scores3;   /* [L,F] per-column  */
  const int32_t* rank1;   /* [L,D] rank of a column inside its head (0 = smallest) */
  const int32_t* rankh;   /* [L,H] */
  const int32_t* rank3;   /* [L,F] */
  float* out;             /* [4]: {cur_resource, R2 (zloss sample), |grad_s|_inf, |grad_r|_inf} */
} uvc_state;

const char* uvc_last_error(void);

/* weight_list_to_scores for every layer in one pass (UVC/uvc_utils.py:54-73).
 * W1/W3: device arrays of L device pointers to attn.proj.weight[D,D] / mlp.fc2.weight[D,F].
 * ws64: float64 scratch [L*(D+F)].  float64 accumulation, one rounding to float32. */
int uvc_scores(const float* const* W1, const float* const* W3, uvc_dims d, double* ws64,
               float* scores1, float* scores2, float* scores3, void* stream);

/* Ascending ranks (ties -> lower index first) == the order torch.topk(largest=False) selects in
 * (uvc_utils.py:81,238,328,334,343,387,390,398,422). */
int uvc_rank(const float* scores1, const float* scores2, const float* scores3, uvc_dims d,
             int32_t* rank1, int32_t* rankh, int32_t* rank3, void* stream);

/* prox_w (uvc_utils.py:315-345) in place + scores of the shrunk weights in the same pass.
 * lr = optimizer.param_groups[0]['lr'] after scheduler.step() (float64, as Python computes
 * 1.0 + 2.0*lr*dual). */
int uvc_prox(float* const* W1, float* const* W3, uvc_dims d, const int32_t* rank1, const int32_t* rankh,
             const int32_t* rank3, const float* s, const float* r, const float* y, const float* p, double lr,
             double* ws64, float* scores1_post, float* scores2_post, float* scores3_post, void* stream);

/* The scalar part of uvc_optimizer (uvc_optimizer.py:46-135): sloss1/rloss1 gradients,
 * calc_flops + its gradient (uvc_utils.py:409-462), gating accumulation/SGD, box-projected SGD
 * on s and r, dual ascent on y,p,z, proj_dual.  e1/e2: Exp(1) draws [L,2] of the two resource
 * samples.  enable_warmup != 0 -> early return after cur_resource (uvc_optimizer.py:52-58). */
int uvc_dual_step(const uvc_state* st, uvc_dims d, uvc_hyper hp, const float* e1, const float* e2,
                  int32_t enable_warmup, int32_t global_step, void* stream);

/* run_resource_fn(gumbel_hard) (uvc_utils.py:220-224): out[0] = FLOPs ratio. */
int uvc_resource(const uvc_state* st, uvc_dims d, uvc_hyper hp, const float* e, int32_t hard, float* out,
                 void* stream);

/* prune_w_mask (uvc_utils.py:376-401): 0/1 masks attn.proj.mask[D,D], mlp.fc2.mask[D,F],
 * mlp.fc1.mask[F,D] from the ranks of the current scores and ceil(s), ceil(r). */
int uvc_write_masks(float* const* mask_proj, float* const* mask_fc2, float* const* mask_fc1, uvc_dims d,
                    const int32_t* rank1, const int32_t* rankh, const int32_t* rank3, const float* s,
                    const float* r, void* stream);

#ifdef __cplusplus
}
#endif
#endif
