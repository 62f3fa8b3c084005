"""ctypes binding of libuvc_hip.so (the C-ABI declared in include/*.h).

The product path has no CPU fallback: if the library is missing or a call fails this raises.
`import torch` happens first on purpose -- torch bundles its own libamdhip64.so.7 and the
library must bind to the SAME HIP runtime instance so that torch's device pointers and
streams are valid inside our kernels (the dynamic loader reuses the already-loaded soname).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuvc_hip.so")


class UvcHipError(RuntimeError):
    pass


class uvc_dims(C.Structure):
    _fields_ = [("L", C.c_int32), ("H", C.c_int32), ("hd", C.c_int32), ("D", C.c_int32), ("F", C.c_int32)]


class uvc_hyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("budget", "slr", "rlr", "glr", "ylr", "plr", "zlr", "sl2wd", "z_grad_clip",
                                         "gating_weight", "eps")] + \
               [(n, C.c_int32) for n in ("gating_interval", "use_gumbel", "enable_block_gating")]


class uvc_state(C.Structure):
    _fields_ = [("s", C.c_void_p), ("r", C.c_void_p), ("y", C.c_void_p), ("p", C.c_void_p), ("z", C.c_void_p),
                ("gate", C.c_void_p), ("gate_grad", C.c_void_p), ("gate_momentum", C.c_void_p),
                ("gate_gsum", C.c_void_p), ("gate_counters", C.c_void_p), ("total_macs", C.c_void_p),
                ("embed_macs", C.c_float), ("resource_ub", C.c_float),
                ("scores1", C.c_void_p), ("scores2", C.c_void_p), ("scores3", C.c_void_p),
                ("rank1", C.c_void_p), ("rankh", C.c_void_p), ("rank3", C.c_void_p), ("out", C.c_void_p)]


class uvc_gemm_nt_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("A", "B", "C", "C2", "bias", "R", "R2", "aux", "gate", "alpha_ptr")] + \
               [("alpha", C.c_float)] + \
               [(n, C.c_int32) for n in ("M", "N", "K", "lda", "ldb", "ldc", "ldr", "ldaux", "dtype", "a_is_f32",
                                         "c_is_f32", "epilogue", "force_generic", "r_is_f32")] + \
               [("ln_eps", C.c_float)] + [(n, C.c_void_p) for n in ("ln_gamma", "ln_beta", "ln_out", "ln_mean", "ln_rstd")]


class uvc_gemm_tn_args(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("alpha_ptr", C.c_void_p), ("colsum_out", C.c_void_p), ("alpha", C.c_float),
                ("beta", C.c_float)] + \
               [(n, C.c_int32) for n in ("M", "N1", "N2", "lda", "ldb", "ldc", "dtype", "a_is_f32", "variant")]


class uvc_attn_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv", "o", "lse", "dout", "dqkv", "delta")] + \
               [(n, C.c_int32) for n in ("B", "N", "H", "head_dim", "dtype")] + [("scale", C.c_float), ("head_keep", C.c_void_p),
                                                                                ("variant", C.c_int32), ("grid", C.c_int32)]


class uvc_qkv_attn_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("h", "w", "bias", "qkv", "o", "lse")] + \
               [(n, C.c_int32) for n in ("B", "N", "H", "D", "dtype")] + [("scale", C.c_float), ("grid", C.c_int32)]


class uvc_attn_tok_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv", "o", "dout", "dqkv")] + \
               [(n, C.c_int32) for n in ("B", "N", "H", "head_dim", "ntok", "dtype")] + [("scale", C.c_float), ("head_keep", C.c_void_p)]


class uvc_ln_reduce_item(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("partial", "dgamma", "dbeta", "dots")] + [("nblocks", C.c_int32), ("reserved", C.c_int32)]


class uvc_ln_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "gamma", "beta", "y", "mean", "rstd", "dy", "dx", "add1", "a1", "add2",
                                          "a2", "partial", "dgamma", "dbeta", "dots")] + \
               [("eps", C.c_float), ("beta_acc", C.c_float)] + \
               [(n, C.c_int32) for n in ("rows", "D", "rows_per_group", "dtype", "y_is_f32", "dy_is_f32")] + \
               [("group_stride", C.c_int64), ("g_lowp", C.c_int32), ("defer_reduce", C.c_int32), ("x_lowp", C.c_int32), ("reserved", C.c_int32)]


class uvc_gemm_lnbwd_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("A", "W", "x", "mean", "rstd", "gamma", "add1", "a1", "add2", "a2", "dx", "partial")] + \
               [(n, C.c_int32) for n in ("M", "D", "K", "dtype", "variant", "x_lowp")]


class uvc_mlp_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "gamma", "beta", "w1", "b1", "w2", "b2", "out")] + \
               [(n, C.c_int32) for n in ("M", "D", "F")] + [("eps", C.c_float)] + \
               [(n, C.c_void_p) for n in ("x_prev", "gate", "h", "mean", "rstd", "gp", "u",
                                          "next_gamma", "next_beta", "next_h", "next_mean", "next_rstd")] + \
               [("rows_lowp", C.c_int32), ("reserved", C.c_int32)]


class uvc_loss_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("o", "o_kd", "y_soft", "teacher", "loss", "d_o", "d_okd", "row_scratch")] + \
               [("alpha", C.c_float), ("tau", C.c_float), ("B", C.c_int32), ("C", C.c_int32), ("kind", C.c_int32)]


class uvc_adamw_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("p", "g", "m", "v", "p_shadow", "sq", "gnorm_out")] + [("n", C.c_int64)] + \
               [(n, C.c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "max_norm")] + \
               [("step", C.c_int32), ("flags", C.c_void_p)]


class uvc_unfold_args(C.Structure):          # include/uvc_t2t.h
    _fields_ = [("src", C.c_void_p)] + [(n, C.c_int64) for n in ("sb", "sc", "sh", "sw")] + \
               [(n, C.c_int32) for n in ("B", "C", "H", "W", "k", "s", "p", "ldo", "out_is_f32", "dtype")] + \
               [("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float), ("out", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("dy", C.c_void_p), ("dy_is_f32", C.c_int32), ("dxu", C.c_void_p), ("partial", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("beta_acc", C.c_float), ("dxu_tap_major", C.c_int32)]


class uvc_t2t_stage(C.Structure):            # include/uvc_t2t.h
    _fields_ = [("src", C.c_void_p)] + [(n, C.c_int64) for n in ("sb", "sc", "sh", "sw")] + \
               [(n, C.c_int32) for n in ("B", "C", "H", "W", "k", "s", "p", "T", "dim", "dimp", "dtype", "training")] + \
               [("eps", C.c_float), ("beta", C.c_float), ("need_dx", C.c_int32), ("reserved", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("norm1_w", "norm1_b", "kqv_b", "w", "proj_b", "norm2_w", "norm2_b", "fc1_b", "fc2_b",
                                          "g_norm1_w", "g_norm1_b", "g_kqv_w", "g_kqv_b", "g_proj_w", "g_proj_b", "g_norm2_w", "g_norm2_b", "g_fc1_w", "g_fc1_b",
                                          "g_fc2_w", "g_fc2_b",
                                          "kqv_w", "kqv_wt", "proj_w", "proj_wt", "fc1_w", "fc1_wt", "fc2_w", "fc2_wt",
                                          "xn", "mean1", "rstd1", "kqv", "part", "kptv", "att", "x1", "h", "mean2", "rstd2", "u", "gp", "out",
                                          "dout", "da", "dh", "dx1", "datt", "dkqv", "dkptv", "dxn", "ln2_partial", "ln1_partial", "dxu", "tn_ws")] + \
               [("tn_ws_bytes", C.c_int64)]


class uvc_performer_args(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("kqv", "w", "part", "kptv", "att")] + [("att_is_f32", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("datt", "dskip", "dkqv", "dkptv")] + \
               [(n, C.c_int32) for n in ("g_is_f32", "B", "T", "dtype")]


UVC_F32, UVC_BF16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_BIAS_RESID_GATE, EPI_DGELU, EPI_BIAS_GELU_OUT, EPI_BIAS_GELU_GRAD, EPI_MUL_AUX = range(9)
EPI_BIAS_GELU_GRAD_Q8, EPI_MUL_AUX_Q8 = 9, 10      # GELU'(a) as one byte per activation (include/uvc_kernels.h)
Q8_LO, Q8_STEP = -0.13, 1.26 / 255.0               # UVC_Q8_LO, UVC_Q8_STEP

_lib = None
VP = C.c_void_p
I32, I64, F32 = C.c_int32, C.c_int64, C.c_float

_SIGNATURES = {
    # include/uvc_engine.h
    "uvc_scores": [VP, VP, uvc_dims, VP, VP, VP, VP, VP],
    "uvc_rank": [VP, VP, VP, uvc_dims, VP, VP, VP, VP],
    "uvc_prox": [VP, VP, uvc_dims, VP, VP, VP, VP, VP, VP, VP, C.c_double, VP, VP, VP, VP, VP],
    "uvc_dual_step": [C.POINTER(uvc_state), uvc_dims, uvc_hyper, VP, VP, C.c_int32, C.c_int32, VP],
    "uvc_resource": [C.POINTER(uvc_state), uvc_dims, uvc_hyper, VP, C.c_int32, VP, VP],
    "uvc_write_masks": [VP, VP, VP, uvc_dims, VP, VP, VP, VP, VP, VP],
    # include/uvc_kernels.h
    "uvc_gemm_nt": [C.POINTER(uvc_gemm_nt_args), VP],
    "uvc_gemm_nt_ln_supported": [I32, I32, I32, I32, I32],
    "uvc_gemm_nt_q8_supported": [I32, I32, I32, I32],
    "uvc_gemm_tn": [C.POINTER(uvc_gemm_tn_args), VP],
    "uvc_gemm_tn_workspace_bytes": [I32, I32, I32, C.POINTER(I64), C.POINTER(I32)],
    "uvc_attention_fwd": [C.POINTER(uvc_attn_args), VP],
    "uvc_attention_bwd": [C.POINTER(uvc_attn_args), VP],
    "uvc_qkv_attention_supported": [I32, I32, I32, I32, I32],
    "uvc_qkv_attention_fwd": [C.POINTER(uvc_qkv_attn_args), VP],
    "uvc_attention_tok_fwd": [C.POINTER(uvc_attn_tok_args), VP],
    "uvc_attention_tok_bwd": [C.POINTER(uvc_attn_tok_args), VP],
    "uvc_copy_row_groups": [VP, VP, I64, I64, I64, I64, VP],
    "uvc_layernorm_fwd": [C.POINTER(uvc_ln_args), VP],
    "uvc_layernorm_bwd": [C.POINTER(uvc_ln_args), VP],
    "uvc_layernorm_bwd_blocks": [I32],
    "uvc_layernorm_bwd_nblocks": [I32],
    "uvc_layernorm_bwd_reduce_batch": [C.POINTER(uvc_ln_reduce_item), I32, I32, F32, VP],
    "uvc_gemm_lnbwd_supported": [I32, I32, I32, I32],
    "uvc_gemm_lnbwd_nblocks": [I32],
    "uvc_gemm_nt_lnbwd": [C.POINTER(uvc_gemm_lnbwd_args), VP],
    "uvc_distill_loss": [C.POINTER(uvc_loss_args), VP],
    "uvc_grad_sqnorm": [VP, I64, VP, VP, I32, VP],
    "uvc_adamw_step": [C.POINTER(uvc_adamw_args), VP],
    "uvc_scale_by_clip": [VP, I64, VP, F32, VP],
    "uvc_patchify": [VP, VP, I32, I32, I32, I32, I32, VP],
    "uvc_assemble_tokens": [VP, VP, VP, VP, VP, VP, I32, I32, I32, I32, I32, VP],
    "uvc_assemble_tokens_bwd": [VP, VP, VP, VP, VP, VP, VP, VP, I32, I32, I32, I32, I32, I32, I32, F32, VP],
    "uvc_colsum": [VP, I32, I32, I32, I32, I32, VP, VP, F32, VP, F32, VP, VP],
    "uvc_patch_gate_sigmoid": [VP, VP, I32, I32, I32, VP],
    "uvc_patch_gate_sigmoid_bwd": [VP, VP, VP, I32, I32, F32, VP],
    "uvc_patch_scores": [VP, VP, VP, VP, I32, I32, VP],
    "uvc_patch_topk_mask": [VP, VP, VP, VP, VP, I32, I32, I32, F32, VP],
    "uvc_patch_topk_mask_bwd": [VP, VP, VP, VP, I32, I32, F32, VP],
    "uvc_add_outer": [VP, VP, VP, I32, I32, I32, I32, VP],
    "uvc_colsum_blocks": [I32],
    "uvc_apply_masks": [VP, VP, I64, VP],
    "uvc_mlp_gather_shadows": [VP, VP, VP, VP, I32, I32, I32, VP, VP, VP, VP, VP, I32, VP],
    "uvc_mlp_scatter_grads": [VP, VP, VP, VP, VP, VP, I32, I32, I32, VP, VP, VP, F32, I32, VP],
    "uvc_mixup_batch": [VP, I32, I32, I32, I32, F32, F32, I32, I32, I32, I32, I32, VP],
    "uvc_mixup_target": [VP, VP, I32, I32, F32, F32, F32, F32, VP],
    "uvc_stream_create": [I32, C.POINTER(VP)],
    "uvc_mlp_fused_supported": [I32, I32, I32],
    "uvc_mlp_fused_fwd": [C.POINTER(uvc_mlp_args), VP],
    "uvc_cast_transpose": [VP, I32, I32, VP, VP, I32, VP],
    "uvc_cast_transpose_multi": [VP, VP, I32, VP, VP, VP, VP, VP, I32, VP],
    "uvc_exp_noise": [VP, I64, C.c_uint64, C.c_uint64, C.c_uint32, VP],
    "uvc_gate_distrib": [VP, VP, VP, I32, I32, F32, VP],
    "uvc_gate_grad": [VP, VP, VP, VP, I32, I32, F32, F32, VP],
    # include/uvc_t2t.h
    "uvc_unfold_ln_fwd": [C.POINTER(uvc_unfold_args), VP],
    "uvc_unfold_ln_bwd": [C.POINTER(uvc_unfold_args), VP],
    "uvc_unfold_bwd_blocks": [I32],
    "uvc_fold_tokens": [VP, I32, I32, I32, VP, I32, I32, I32, I32, I32, I32, I32, I32, I32, VP],
    "uvc_performer_splits": [I32, I32],
    "uvc_performer_fwd": [C.POINTER(uvc_performer_args), VP],
    "uvc_performer_bwd": [C.POINTER(uvc_performer_args), VP],
    "uvc_t2t_stage_forward": [C.POINTER(uvc_t2t_stage), VP],
    "uvc_t2t_stage_backward": [C.POINTER(uvc_t2t_stage), VP],
}


# include/uvc_vit.h (bound in uvc_amd/model_distilled.py next to its ctypes structures)
VIT_SYMBOLS = ["uvc_vit_layout", "uvc_vit_workspace_bytes", "uvc_vit_ws_offsets", "uvc_vit_update_shadows", "uvc_vit_forward",
               "uvc_vit_backward"]


def side_stream(device, priority_class=None):
    """A torch handle on a HIP stream created by the library with a scheduling class (see uvc_stream_create):
    side work (wgrads, teacher forward) runs at the lowest priority so it only fills CUs the critical path leaves
    idle.  UVC_SIDE_PRIORITY=-1|0|1 overrides the class."""
    import torch
    if priority_class is None:
        priority_class = int(os.environ.get("UVC_SIDE_PRIORITY", "0"))
    out = VP()
    with torch.cuda.device(device):
        check(lib().uvc_stream_create(priority_class, C.byref(out)), "uvc_stream_create")
    return torch.cuda.ExternalStream(out.value, device=device)


def exported_symbols():
    return list(_SIGNATURES) + ["uvc_last_error"] + VIT_SYMBOLS


def lib():
    """Load (once) and return the CDLL.  Raises UvcHipError when the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UvcHipError(f"{LIB_PATH} not found: build it with `python -m uvc_amd.build` "
                              "(there is no CPU fallback for the product path)")
        _lib = C.CDLL(LIB_PATH)
        _lib.uvc_last_error.restype = C.c_char_p
        for name, args in _SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise UvcHipError(f"{what} failed (rc={rc}): {lib().uvc_last_error().decode()}")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def cur_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise UvcHipError("uvc_amd runs on MI355X only: got a CPU tensor (no CPU fallback)")
