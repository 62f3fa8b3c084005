"""Thin tensor-level wrappers over the C-ABI kernels (include/uvc_kernels.h).  They check
devices/dtypes/contiguity, fill the argument structs and enqueue on torch's current HIP
stream.  No arithmetic happens in Python and there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GELU_GRAD, EPI_BIAS_GELU_GRAD_Q8, EPI_BIAS_GELU_OUT, EPI_BIAS_RESID, EPI_BIAS_RESID_GATE, EPI_DGELU,
                   EPI_MUL_AUX, EPI_MUL_AUX_Q8, EPI_NONE, Q8_LO, Q8_STEP, UVC_BF16, UVC_F32)

__all__ = ["EPI_NONE", "EPI_BIAS", "EPI_BIAS_GELU", "EPI_BIAS_RESID", "EPI_BIAS_RESID_GATE", "EPI_DGELU", "UVC_F32",
           "UVC_BF16"]


def tdtype(dtype: int) -> torch.dtype:
    return torch.float32 if dtype == UVC_F32 else torch.bfloat16


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        L.require_cuda(t)
        if not t.is_contiguous():
            raise L.UvcHipError("kernel operands must be contiguous")


def _is_f32(t) -> int:
    return int(t.dtype == torch.float32)


def gemm_nt(A, B, C_, *, dtype, epilogue=EPI_NONE, bias=None, R=None, R2=None, aux=None, gate=None, C2=None, alpha=1.0,
            alpha_ptr=None, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, ldr=None, ldaux=None, force_generic=False,
            ln_gamma=None, ln_beta=None, ln_out=None, ln_mean=None, ln_rstd=None, ln_eps=1e-6):
    """C[M,N] = epi(A[M,K] . B[N,K]^T).  ln_out (where uvc_gemm_nt_ln_supported): also LayerNorm(C rows; ln_gamma, ln_beta) in the
    compute dtype, with ln_mean / ln_rstd."""
    _chk(A, B, C_, bias, R, R2, aux, gate, C2, alpha_ptr, ln_gamma, ln_beta, ln_out, ln_mean, ln_rstd)
    a = L.uvc_gemm_nt_args()
    a.ln_gamma, a.ln_beta, a.ln_out, a.ln_mean, a.ln_rstd, a.ln_eps = L.ptr(ln_gamma), L.ptr(ln_beta), L.ptr(ln_out), L.ptr(ln_mean), L.ptr(ln_rstd), ln_eps
    a.A, a.B, a.C, a.C2 = L.ptr(A), L.ptr(B), L.ptr(C_), L.ptr(C2)
    a.bias, a.R, a.R2, a.aux, a.gate, a.alpha_ptr = L.ptr(bias), L.ptr(R), L.ptr(R2), L.ptr(aux), L.ptr(gate), L.ptr(alpha_ptr)
    a.alpha = alpha
    a.M = M if M is not None else A.shape[0]
    a.K = K if K is not None else A.shape[-1]
    a.N = N if N is not None else B.shape[0]
    a.lda = lda if lda is not None else a.K
    a.ldb = ldb if ldb is not None else a.K
    a.ldc = ldc if ldc is not None else a.N
    a.ldr = ldr if ldr is not None else a.ldc
    a.ldaux = ldaux if ldaux is not None else a.ldc
    a.dtype, a.a_is_f32, a.c_is_f32, a.epilogue = dtype, _is_f32(A), _is_f32(C_), epilogue
    for t in (R, R2):
        if t is not None and t.dtype != C_.dtype:
            raise L.UvcHipError("gemm_nt: the residual operands R / R2 have C's element type")
    a.force_generic = int(force_generic)
    a.r_is_f32 = _is_f32(R) if R is not None else 0
    L.check(L.lib().uvc_gemm_nt(C.byref(a), L.cur_stream()), "uvc_gemm_nt")


def gemm_tn_workspace_bytes(M, N1, N2) -> int:
    b = C.c_int64()
    s = C.c_int32()
    L.check(L.lib().uvc_gemm_tn_workspace_bytes(M, N1, N2, C.byref(b), C.byref(s)), "uvc_gemm_tn_workspace_bytes")
    return int(b.value)


def gemm_tn(A, B, C_, workspace, *, dtype, alpha=1.0, alpha_ptr=None, beta=0.0, M=None, N1=None, N2=None, lda=None,
            ldb=None, colsum_out=None, variant=0):
    """C[N1,N2] = beta*C + alpha * A[M,N1]^T . B[M,N2]  (+ optional colsum_out[N1] = beta*old + alpha*colsum(A))."""
    _chk(A, B, C_, workspace, alpha_ptr, colsum_out)
    a = L.uvc_gemm_tn_args()
    a.A, a.B, a.C, a.workspace = L.ptr(A), L.ptr(B), L.ptr(C_), L.ptr(workspace)
    a.workspace_bytes = workspace.numel() * workspace.element_size()
    a.alpha_ptr, a.alpha, a.beta, a.colsum_out = L.ptr(alpha_ptr), alpha, beta, L.ptr(colsum_out)
    a.M = M if M is not None else A.shape[0]
    a.N1 = N1 if N1 is not None else A.shape[1]
    a.N2 = N2 if N2 is not None else B.shape[1]
    a.lda = lda if lda is not None else a.N1
    a.ldb = ldb if ldb is not None else a.N2
    a.ldc = a.N2
    a.dtype, a.a_is_f32 = dtype, _is_f32(A)
    a.variant = int(variant)
    L.check(L.lib().uvc_gemm_tn(C.byref(a), L.cur_stream()), "uvc_gemm_tn")


def _attn_args(qkv, o, lse, B, N, H, dtype, dout=None, dqkv=None, delta=None):
    _chk(qkv, o, lse, dout, dqkv, delta)
    a = L.uvc_attn_args()
    a.qkv, a.o, a.lse, a.dout, a.dqkv, a.delta = (L.ptr(t) for t in (qkv, o, lse, dout, dqkv, delta))
    a.B, a.N, a.H, a.head_dim, a.dtype = B, N, H, 64, dtype
    a.scale = 64 ** -0.5
    return a


def attention_fwd(qkv, o, lse, B, N, H, dtype, head_keep=None):
    a = _attn_args(qkv, o, lse, B, N, H, dtype)
    if head_keep is not None:
        _chk(head_keep)
        a.head_keep = L.ptr(head_keep)
    L.check(L.lib().uvc_attention_fwd(C.byref(a), L.cur_stream()), "uvc_attention_fwd")


def attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, dtype, head_keep=None, variant=0, grid=0):
    a = _attn_args(qkv, o, lse, B, N, H, dtype, dout, dqkv, delta)
    a.variant, a.grid = int(variant), int(grid)
    if head_keep is not None:
        _chk(head_keep)
        a.head_keep = L.ptr(head_keep)
    L.check(L.lib().uvc_attention_bwd(C.byref(a), L.cur_stream()), "uvc_attention_bwd")


def qkv_attention_supported(B, N, H, D, dtype):
    return bool(L.lib().uvc_qkv_attention_supported(B, N, H, D, dtype))


def qkv_attention_fwd(h, w, bias, o, lse, B, N, H, dtype, qkv=None, grid=0):
    """The qkv Linear + attention forward as one kernel (include/uvc_kernels.h: uvc_qkv_attention_fwd); qkv is written when given."""
    _chk(h, w, o)
    a = L.uvc_qkv_attn_args()
    a.h, a.w, a.bias, a.qkv, a.o, a.lse = (L.ptr(t) for t in (h, w, bias, qkv, o, lse))
    a.B, a.N, a.H, a.D, a.dtype, a.scale, a.grid = B, N, H, H * 64, dtype, 64 ** -0.5, int(grid)
    L.check(L.lib().uvc_qkv_attention_fwd(C.byref(a), L.cur_stream()), "uvc_qkv_attention_fwd")


def _attn_tok_args(qkv, o, B, N, H, ntok, dtype, dout=None, dqkv=None):
    a = L.uvc_attn_tok_args()
    for t in (qkv, o, dout, dqkv):
        if t is not None:
            _chk(t)
    a.qkv, a.o = L.ptr(qkv), L.ptr(o)
    a.dout, a.dqkv = L.ptr(dout), L.ptr(dqkv)
    a.B, a.N, a.H, a.head_dim, a.ntok, a.dtype, a.scale = B, N, H, 64, ntok, dtype, 0.125
    return a


def attention_tok_fwd(qkv, o, B, N, H, ntok, dtype, head_keep=None):
    """Attention of the first ntok queries of every (image, head): o is [B, ntok, H*64]."""
    a = _attn_tok_args(qkv, o, B, N, H, ntok, dtype)
    if head_keep is not None:
        _chk(head_keep)
        a.head_keep = L.ptr(head_keep)
    L.check(L.lib().uvc_attention_tok_fwd(C.byref(a), L.cur_stream()), "uvc_attention_tok_fwd")


def attention_tok_bwd(qkv, o, dout, dqkv, B, N, H, ntok, dtype):
    a = _attn_tok_args(qkv, o, B, N, H, ntok, dtype, dout, dqkv)
    L.check(L.lib().uvc_attention_tok_bwd(C.byref(a), L.cur_stream()), "uvc_attention_tok_bwd")


def copy_row_groups(src, dst, groups, group_bytes, src_group_stride, dst_group_stride):
    _chk(src)
    _chk(dst)
    L.check(L.lib().uvc_copy_row_groups(L.ptr(src), L.ptr(dst), groups, group_bytes, src_group_stride, dst_group_stride, L.cur_stream()),
            "uvc_copy_row_groups")


def _ln_args(x, gamma, beta, rows, D, dtype, rows_per_group=1, group_stride=None, eps=1e-6):
    a = L.uvc_ln_args()
    a.x, a.gamma, a.beta = L.ptr(x), L.ptr(gamma), L.ptr(beta)
    a.rows, a.D, a.rows_per_group, a.dtype = rows, D, rows_per_group, dtype
    a.group_stride = group_stride if group_stride is not None else D * rows_per_group
    a.eps = eps
    a.x_lowp = int(not _is_f32(x))                   # bf16 residual stream
    return a


def layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, dtype, rows_per_group=1, group_stride=None, eps=1e-6):
    _chk(x, gamma, beta, y, mean, rstd)
    a = _ln_args(x, gamma, beta, rows, D, dtype, rows_per_group, group_stride, eps)
    a.y, a.mean, a.rstd, a.y_is_f32 = L.ptr(y), L.ptr(mean), L.ptr(rstd), _is_f32(y)
    L.check(L.lib().uvc_layernorm_fwd(C.byref(a), L.cur_stream()), "uvc_layernorm_fwd")


def layernorm_bwd_blocks(rows) -> int:
    return int(L.lib().uvc_layernorm_bwd_blocks(rows))


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, partial, dgamma, dbeta, rows, D, dtype, *, add1=None, a1=None, add2=None,
                  a2=None, dots=None, beta_acc=0.0, rows_per_group=1, group_stride=None, eps=1e-6):
    _chk(dy, x, gamma, mean, rstd, dx, partial, dgamma, dbeta, add1, a1, add2, a2, dots)
    a = _ln_args(x, gamma, None, rows, D, dtype, rows_per_group, group_stride, eps)
    a.mean, a.rstd, a.dy, a.dx = L.ptr(mean), L.ptr(rstd), L.ptr(dy), L.ptr(dx)
    a.add1, a.a1, a.add2, a.a2 = L.ptr(add1), L.ptr(a1), L.ptr(add2), L.ptr(a2)
    a.partial, a.dgamma, a.dbeta, a.dots = L.ptr(partial), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dots)
    a.beta_acc, a.dy_is_f32 = beta_acc, _is_f32(dy)
    a.g_lowp = int(not _is_f32(dx))          # gradient stream (dx, add1, add2) kept in bf16
    for t in (add1, add2):
        if t is not None and t.dtype != dx.dtype:
            raise L.UvcHipError("layernorm_bwd: add1/add2 must have dx's element type")
    L.check(L.lib().uvc_layernorm_bwd(C.byref(a), L.cur_stream()), "uvc_layernorm_bwd")


def gemm_nt_q8_supported(M, hidden, embed_dim, dtype) -> bool:
    """Where the one-byte GELU'(a) epilogues exist (EPI_BIAS_GELU_GRAD_Q8 / EPI_MUL_AUX_Q8)."""
    return bool(L.lib().uvc_gemm_nt_q8_supported(M, hidden, embed_dim, dtype))


def gemm_lnbwd_supported(M, D, K, dtype) -> bool:
    return bool(L.lib().uvc_gemm_lnbwd_supported(M, D, K, dtype))


def gemm_nt_lnbwd(A, Wt, x, mean, rstd, gamma, dx, partial, dgamma, dbeta, *, add1=None, a1=None, add2=None, a2=None, dots=None, beta_acc=0.0,
                  variant=0):
    """dx = LN'(A . Wt^T; x, mean, rstd, gamma) + a1*add1 + a2*add2 in one kernel (include/uvc_kernels.h: uvc_gemm_nt_lnbwd), then the
    batched finish of dgamma / dbeta / dots."""
    _chk(A, Wt, x, mean, rstd, gamma, dx, partial, dgamma, dbeta, add1, a1, add2, a2, dots)
    a = L.uvc_gemm_lnbwd_args()
    a.A, a.W, a.x, a.mean, a.rstd, a.gamma = (L.ptr(t) for t in (A, Wt, x, mean, rstd, gamma))
    a.add1, a.a1, a.add2, a.a2, a.dx, a.partial = (L.ptr(t) for t in (add1, a1, add2, a2, dx, partial))
    a.M, a.D, a.K, a.dtype, a.variant, a.x_lowp = A.shape[0], x.shape[1], A.shape[1], UVC_BF16, int(variant), int(not _is_f32(x))
    L.check(L.lib().uvc_gemm_nt_lnbwd(C.byref(a), L.cur_stream()), "uvc_gemm_nt_lnbwd")
    item = (L.uvc_ln_reduce_item * 1)()
    item[0].partial, item[0].dgamma, item[0].dbeta, item[0].dots = L.ptr(partial), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dots)
    item[0].nblocks = int(L.lib().uvc_gemm_lnbwd_nblocks(a.M))
    L.check(L.lib().uvc_layernorm_bwd_reduce_batch(item, 1, a.D, beta_acc, L.cur_stream()), "uvc_layernorm_bwd_reduce_batch")


def mlp_fused_fwd(x, gamma, beta, w1, b1, w2, b2, out, eps=1e-6, *, x_prev=None, gate=None, h=None, mean=None, rstd=None, gp=None, u=None,
                  next_gamma=None, next_beta=None, next_h=None, next_mean=None, next_rstd=None):
    """out = d1 * (x + fc2(GELU(fc1(LayerNorm(x))))) + d0 * x_prev for [M, 192] float32 rows; w1 [F,192] / w2 [192,F] bf16.
    Training form: h / mean / rstd / gp / u receive LayerNorm(x), its statistics, GELU'(a) and GELU(a).
    next_h (bf16 [M,192], optional): LayerNorm(out; next_gamma, next_beta) -- the next block's norm1 -- with its statistics in
    next_mean / next_rstd when given."""
    _chk(x, gamma, beta, w1, b1, w2, b2, out, x_prev, gate, h, mean, rstd, gp, u, next_gamma, next_beta, next_h, next_mean, next_rstd)
    a = L.uvc_mlp_args()
    a.next_gamma, a.next_beta, a.next_h, a.next_mean, a.next_rstd = (L.ptr(t) for t in (next_gamma, next_beta, next_h, next_mean, next_rstd))
    a.x, a.gamma, a.beta, a.w1, a.b1, a.w2, a.b2, a.out = (L.ptr(t) for t in (x, gamma, beta, w1, b1, w2, b2, out))
    a.x_prev, a.gate, a.h, a.mean, a.rstd, a.gp, a.u = (L.ptr(t) for t in (x_prev, gate, h, mean, rstd, gp, u))
    a.M, a.D, a.F, a.eps = x.shape[0], x.shape[1], w1.shape[0], eps
    a.rows_lowp = int(not _is_f32(x))                # bf16 residual stream: x, out, x_prev are bf16 rows
    for t in (out, x_prev):
        if t is not None and t.dtype != x.dtype:
            raise L.UvcHipError("mlp_fused_fwd: x, out and x_prev must share one element type")
    L.check(L.lib().uvc_mlp_fused_fwd(C.byref(a), L.cur_stream()), "uvc_mlp_fused_fwd")


def distill_loss(o, o_kd, y_soft, teacher, loss, d_o, d_okd, row_scratch, alpha, tau, kind=1):
    _chk(o, o_kd, y_soft, teacher, loss, d_o, d_okd, row_scratch)
    a = L.uvc_loss_args()
    a.o, a.o_kd, a.y_soft, a.teacher = L.ptr(o), L.ptr(o_kd), L.ptr(y_soft), L.ptr(teacher)
    a.loss, a.d_o, a.d_okd, a.row_scratch = L.ptr(loss), L.ptr(d_o), L.ptr(d_okd), L.ptr(row_scratch)
    a.alpha, a.tau, a.B, a.C, a.kind = alpha, tau, o.shape[0], o.shape[1], kind
    L.check(L.lib().uvc_distill_loss(C.byref(a), L.cur_stream()), "uvc_distill_loss")


def grad_sqnorm(g, partial, sq, accumulate=False, n=None):
    """sq: float32[2] -- sq[0] = sum of squares (accumulated over calls with accumulate=True), sq[1] = its square root."""
    _chk(g, partial, sq)
    if sq.numel() < 2:
        raise L.UvcHipError("grad_sqnorm: sq must hold 2 floats (sum of squares, norm)")
    L.check(L.lib().uvc_grad_sqnorm(L.ptr(g), n if n is not None else g.numel(), L.ptr(partial), L.ptr(sq),
                                    int(accumulate), L.cur_stream()), "uvc_grad_sqnorm")


def adamw_step(p, g, m, v, sq, *, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.05, max_norm=1.0,
               p_shadow=None, gnorm_out=None, n=None, flags=None):
    _chk(p, g, m, v, sq, p_shadow, gnorm_out, flags)
    a = L.uvc_adamw_args()
    a.p, a.g, a.m, a.v, a.p_shadow, a.sq, a.gnorm_out = (L.ptr(t) for t in (p, g, m, v, p_shadow, sq, gnorm_out))
    a.n = n if n is not None else p.numel()
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.max_norm, a.step = lr, beta1, beta2, eps, weight_decay, max_norm, step
    a.flags = L.ptr(flags)
    L.check(L.lib().uvc_adamw_step(C.byref(a), L.cur_stream()), "uvc_adamw_step")


def scale_by_clip(g, sq, max_norm):
    _chk(g, sq)
    L.check(L.lib().uvc_scale_by_clip(L.ptr(g), g.numel(), L.ptr(sq), max_norm, L.cur_stream()), "uvc_scale_by_clip")


def patchify(x, out, P, dtype):
    _chk(x, out)
    B, Cc, S, _ = x.shape
    L.check(L.lib().uvc_patchify(L.ptr(x), L.ptr(out), B, Cc, S, P, dtype, L.cur_stream()), "uvc_patchify")


def assemble_tokens(pe, cls, dist, pos, row_mask, tok, B, P, D, ntok):
    _chk(pe, cls, dist, pos, row_mask, tok)
    L.check(L.lib().uvc_assemble_tokens(L.ptr(pe), L.ptr(cls), L.ptr(dist), L.ptr(pos), L.ptr(row_mask), L.ptr(tok), B, P,
                                        D, ntok, int(not _is_f32(tok)), L.cur_stream()), "uvc_assemble_tokens")


def assemble_tokens_bwd(dtok, pe, row_mask, dpe, dpos, dcls, ddist, dmask, B, P, D, ntok, dtype, beta_acc=0.0):
    _chk(dtok, pe, row_mask, dpe, dpos, dcls, ddist, dmask)
    L.check(L.lib().uvc_assemble_tokens_bwd(L.ptr(dtok), L.ptr(pe), L.ptr(row_mask), L.ptr(dpe), L.ptr(dpos), L.ptr(dcls),
                                            L.ptr(ddist), L.ptr(dmask), B, P, D, ntok, dtype, _is_f32(dpe), int(not _is_f32(dtok)), beta_acc,
                                            L.cur_stream()), "uvc_assemble_tokens_bwd")


def colsum_blocks(M) -> int:
    return int(L.lib().uvc_colsum_blocks(M))


def colsum(X, partial, out, dtype, *, M=None, N=None, ldx=None, alpha=1.0, alpha_ptr=None, beta=0.0, row_weight=None):
    _chk(X, partial, out, alpha_ptr, row_weight)
    M = M if M is not None else X.shape[0]
    N = N if N is not None else X.shape[1]
    L.check(L.lib().uvc_colsum(L.ptr(X), M, N, ldx if ldx is not None else N, dtype, _is_f32(X), L.ptr(partial), L.ptr(out),
                               alpha, L.ptr(alpha_ptr), beta, L.ptr(row_weight), L.cur_stream()), "uvc_colsum")


def patch_gate_sigmoid(pg, mask, B, P, hard):
    _chk(pg, mask)
    L.check(L.lib().uvc_patch_gate_sigmoid(L.ptr(pg), L.ptr(mask), B, P, int(bool(hard)), L.cur_stream()), "uvc_patch_gate_sigmoid")


def patch_gate_sigmoid_bwd(pg, dmask, dpg, B, P, beta_acc=0.0):
    _chk(pg, dmask, dpg)
    L.check(L.lib().uvc_patch_gate_sigmoid_bwd(L.ptr(pg), L.ptr(dmask), L.ptr(dpg), B, P, beta_acc, L.cur_stream()),
            "uvc_patch_gate_sigmoid_bwd")


def patch_scores(pe, w, bias, scores, rows, D):
    _chk(pe, w, bias, scores)
    L.check(L.lib().uvc_patch_scores(L.ptr(pe), L.ptr(w), L.ptr(bias), L.ptr(scores), rows, D, L.cur_stream()), "uvc_patch_scores")


def patch_topk_mask(scores, e, mask, ysoft, psoft, B, P, k, tau):
    _chk(scores, e, mask, ysoft, psoft)
    L.check(L.lib().uvc_patch_topk_mask(L.ptr(scores), L.ptr(e), L.ptr(mask), L.ptr(ysoft), L.ptr(psoft), B, P, k, tau,
                                        L.cur_stream()), "uvc_patch_topk_mask")


def patch_topk_mask_bwd(dmask, ysoft, psoft, dscores, B, P, tau):
    _chk(dmask, ysoft, psoft, dscores)
    L.check(L.lib().uvc_patch_topk_mask_bwd(L.ptr(dmask), L.ptr(ysoft), L.ptr(psoft), L.ptr(dscores), B, P, tau, L.cur_stream()),
            "uvc_patch_topk_mask_bwd")


def apply_masks(params, mask):
    _chk(params, mask)
    L.check(L.lib().uvc_apply_masks(L.ptr(params), L.ptr(mask), params.numel(), L.cur_stream()), "uvc_apply_masks")


def add_outer(X, row_weight, w, rows, D, dtype):
    _chk(X, row_weight, w)
    L.check(L.lib().uvc_add_outer(L.ptr(X), L.ptr(row_weight), L.ptr(w), rows, D, dtype, _is_f32(X), L.cur_stream()), "uvc_add_outer")


def cast_transpose(W, R, Cc, w_cast, wt, dtype):
    _chk(W, w_cast, wt)
    L.check(L.lib().uvc_cast_transpose(L.ptr(W), R, Cc, L.ptr(w_cast), L.ptr(wt), dtype, L.cur_stream()), "uvc_cast_transpose")


def gate_distrib(g, e, d, Lb, mode, eps):
    _chk(g, e, d)
    L.check(L.lib().uvc_gate_distrib(L.ptr(g), L.ptr(e), L.ptr(d), Lb, mode, eps, L.cur_stream()), "uvc_gate_distrib")


def gate_grad(g, d, dots, dg, Lb, mode, eps, beta_acc=0.0):
    _chk(g, d, dots, dg)
    L.check(L.lib().uvc_gate_grad(L.ptr(g), L.ptr(d), L.ptr(dots), L.ptr(dg), Lb, mode, eps, beta_acc, L.cur_stream()),
            "uvc_gate_grad")


# ---- T2T tokens-to-token front end (include/uvc_t2t.h) --------------------------------------------
def _unfold_args(src, strides, B, Cc, H, W, k, s, p, ldo, dtype):
    a = L.uvc_unfold_args()
    a.src = L.ptr(src)
    a.sb, a.sc, a.sh, a.sw = strides
    a.B, a.C, a.H, a.W, a.k, a.s, a.p, a.ldo, a.dtype = B, Cc, H, W, k, s, p, ldo, dtype
    return a


def unfold_out_hw(H, W, k, s, p):
    return (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1


def unfold_ln_fwd(src, strides, B, Cc, H, W, k, s, p, out, dtype, *, gamma=None, beta=None, mean=None, rstd=None, eps=1e-5):
    """out[B*L, ldo] = (LayerNorm of) the soft split rows of `src` (element strides (sb, sc, sh, sw)); ldo = out.shape[1]."""
    _chk(src, out, gamma, beta, mean, rstd)
    a = _unfold_args(src, strides, B, Cc, H, W, k, s, p, out.shape[1], dtype)
    a.out, a.out_is_f32 = L.ptr(out), _is_f32(out)
    a.gamma, a.beta, a.mean, a.rstd, a.eps = L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), eps
    L.check(L.lib().uvc_unfold_ln_fwd(C.byref(a), L.cur_stream()), "uvc_unfold_ln_fwd")


def unfold_bwd_blocks(rows) -> int:
    return int(L.lib().uvc_unfold_bwd_blocks(rows))


def unfold_ln_bwd(src, strides, B, Cc, H, W, k, s, p, dy, dtype, *, gamma, mean, rstd, partial, dgamma, dbeta, dxu=None, beta_acc=0.0, eps=1e-5,
                  dxu_tap_major=False):
    """dxu_tap_major (token-major sources with 64 channels): dxu leaves as [rows][k*k][C] for fold_tokens(..., tap_major=True)."""
    _chk(src, dy, gamma, mean, rstd, partial, dgamma, dbeta, dxu)
    a = _unfold_args(src, strides, B, Cc, H, W, k, s, p, dy.shape[1], dtype)
    a.gamma, a.mean, a.rstd, a.eps = L.ptr(gamma), L.ptr(mean), L.ptr(rstd), eps
    a.dy, a.dy_is_f32, a.dxu, a.partial = L.ptr(dy), _is_f32(dy), L.ptr(dxu), L.ptr(partial)
    a.dgamma, a.dbeta, a.beta_acc = L.ptr(dgamma), L.ptr(dbeta), beta_acc
    a.dxu_tap_major = int(bool(dxu_tap_major))
    L.check(L.lib().uvc_unfold_ln_bwd(C.byref(a), L.cur_stream()), "uvc_unfold_ln_bwd")


def fold_tokens(src, dst, B, Cc, H, W, k, s, p, dtype, lds=None, tap_major=False):
    """dst[B, H*W, C] (float32 or T) = adjoint of the soft split applied to src [B*L, lds] (columns c*k*k + tap, or tap*C + c with tap_major)."""
    _chk(src, dst)
    L.check(L.lib().uvc_fold_tokens(L.ptr(src), _is_f32(src), dtype, lds if lds is not None else src.shape[1], L.ptr(dst), _is_f32(dst), B, Cc, H, W,
                                    k, s, p, int(bool(tap_major)), L.cur_stream()), "uvc_fold_tokens")


def performer_splits(B, T) -> int:
    return int(L.lib().uvc_performer_splits(B, T))


def _perf_args(kqv, w, part, kptv, B, T, dtype):
    a = L.uvc_performer_args()
    a.kqv, a.w, a.part, a.kptv, a.B, a.T, a.dtype = L.ptr(kqv), L.ptr(w), L.ptr(part), L.ptr(kptv), B, T, dtype
    return a


def performer_fwd(kqv, w, part, kptv, att, B, T, dtype):
    _chk(kqv, w, part, kptv, att)
    a = _perf_args(kqv, w, part, kptv, B, T, dtype)
    a.att, a.att_is_f32 = L.ptr(att), _is_f32(att)
    L.check(L.lib().uvc_performer_fwd(C.byref(a), L.cur_stream()), "uvc_performer_fwd")


def performer_bwd(kqv, w, part, kptv, datt, dkqv, dkptv, B, T, dtype, dskip=None):
    _chk(kqv, w, part, kptv, datt, dkqv, dkptv, dskip)
    if dkqv.dtype != datt.dtype or (dskip is not None and dskip.dtype != datt.dtype):
        raise L.UvcHipError("performer_bwd: datt, dskip and dkqv must share one element type")
    a = _perf_args(kqv, w, part, kptv, B, T, dtype)
    a.datt, a.dskip, a.dkqv, a.dkptv, a.g_is_f32 = L.ptr(datt), L.ptr(dskip), L.ptr(dkqv), L.ptr(dkptv), _is_f32(datt)
    L.check(L.lib().uvc_performer_bwd(C.byref(a), L.cur_stream()), "uvc_performer_bwd")


class KeyedExpSource:
    """Drop-in for the ``exp_source`` hooks of the model and of UVC_CP_MiniMax: Exp(1) draws from the counter-based HIP generator
    (uvc_exp_noise), keyed by (seed, step, call number within the step).  Every rank that runs the same call sequence gets the same
    numbers; nothing is shared with torch's global generators, so an extra torch draw on one rank (the reference samples its
    FLOPs report on rank 0 only, joint_train.py:509) cannot desynchronise the replicas' s, r, y, p."""

    def __init__(self, seed, device):
        self.seed, self.device = int(seed) & 0xFFFFFFFFFFFFFFFF, device
        self.step, self.site = -1, 0

    def begin_step(self, step, resume_window=False):
        """First call of an accumulation window.  The default REPLAYS: the key restarts at (seed, step, 0) also when ``step`` repeats the
        current number (a retried step, an A/B replay, a determinism test get the same numbers again).  ``resume_window=True`` is for the
        one case that must continue instead: a window left unfinished at an epoch boundary is followed by a new window with the SAME
        optimiser step number (the reference restarts its window count per epoch, joint_train.py:423) -- the trainer knows when that is
        (the previous window ended without an optimiser step) and says so; the draws then continue the site numbering."""
        if resume_window and int(step) == self.step:
            return
        self.step, self.site = int(step), 0

    def __call__(self, shape):
        import torch
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        L.check(L.lib().uvc_exp_noise(L.ptr(out), out.numel(), self.seed, self.step, self.site, L.cur_stream()), "uvc_exp_noise")
        self.site += 1
        return out
