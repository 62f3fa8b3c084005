"""MI355X mirror of ``UVC/models/model_distilled.py``: ``DistilledVisionTransformer`` with the
reference's constructor, attributes, ``forward(x, tau=-1, number=0.9)`` contract and
state_dict keys, executing as the HIP kernel sequence of ``uvc_vit_forward`` /
``uvc_vit_backward`` (include/uvc_vit.h).

Parameters are views into ONE flat float32 buffer (and ``.grad`` into one flat gradient
buffer), which is what the fused clip+AdamW, the RCCL gradient all-reduce and the UVC engine
operate on.  ``loss.backward()`` works through a single autograd node for the whole model.
There is no CPU path: constructing the model needs an MI355X.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import partial
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops

__all__ = ["DistilledVisionTransformer", "PatchEmbed", "Attention", "Mlp", "Block"]


class uvc_vit_cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("img_size", "patch_size", "in_chans", "num_classes", "embed_dim", "depth",
                                         "num_heads", "hidden", "ntok", "dtype")] + [("ln_eps", C.c_float), ("no_qkv_bias", C.c_int32),
                                                                                       ("resid_f32", C.c_int32), ("reserved", C.c_int32)]


MAXD = 32


class uvc_vit_offsets(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("cls_token", "dist_token", "pos_embed", "patch_w", "patch_b")] + \
               [("blk", (C.c_int64 * 12) * MAXD)] + \
               [(n, C.c_int64) for n in ("norm_w", "norm_b", "head_w", "head_b", "headd_w", "headd_b", "n_main", "gate",
                                         "gumbel_w", "gumbel_b", "patch_gating")] + \
               [("skip", (C.c_int64 * 2) * MAXD), ("n_total", C.c_int64)]


class uvc_vit_shadow_offsets(C.Structure):
    _fields_ = [("patch_w", C.c_int64), ("blk_w", (C.c_int64 * 4) * MAXD), ("blk_wt", (C.c_int64 * 4) * MAXD)] + \
               [(n, C.c_int64) for n in ("head_w", "head_wt", "headd_w", "headd_wt", "n_total")]


class uvc_vit_io(C.Structure):
    _fields_ = [("params", C.c_void_p), ("shadow", C.c_void_p), ("grads", C.c_void_p), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_int64), ("x", C.c_void_p), ("logits", C.c_void_p), ("logits_dist", C.c_void_p),
                ("d_logits", C.c_void_p), ("d_logits_dist", C.c_void_p), ("gate_d", C.c_void_p),
                ("run_block", C.POINTER(C.c_int32)), ("patch_mask", C.c_void_p), ("d_patch_mask", C.c_void_p),
                ("batch", C.c_int32), ("training", C.c_int32), ("gate_mode", C.c_int32), ("gate_eps", C.c_float),
                ("accumulate", C.c_float), ("stage_begin", C.c_int32), ("stage_end", C.c_int32), ("side_stream", C.c_void_p),
                ("mlp_compact", C.c_void_p), ("head_keep", C.c_void_p), ("full_tail", C.c_int32), ("fused_train_mlp", C.c_int32), ("patches_in", C.c_void_p),
                ("fuse_next_ln", C.c_int32), ("force_generic", C.c_int32), ("head_keep_bwd", C.c_int32), ("shared_bwd_streams", C.c_int32),
                ("gelu_grad_bf16", C.c_int32)]


class uvc_mlp_compact(C.Structure):
    _fields_ = [("width", C.c_int32), ("reserved", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("w1", "w1t", "w2", "w2t", "b1", "dw1", "dw2", "db1", "inv")]


# The last block's rows other than the class / distillation token never reach the head (model_distilled.py:507-526): the engine
# computes that block's tail on the token rows only (uvc_vit_io.full_tail = 0; identical outputs and gradients).  UVC_FULL_TAIL=1, or
# model.full_tail = True, runs every row as the reference does (A/B measurements, tests).
_FULL_TAIL_DEFAULT = os.environ.get("UVC_FULL_TAIL", "0") not in ("", "0")
_GELU_GRAD_BF16_DEFAULT = os.environ.get("UVC_GELU_GRAD_BF16", "0") not in ("", "0")   # A/B: GELU'(a) of fc1 in bf16 instead of the one-byte code
_FUSE_NEXT_LN_DEFAULT = os.environ.get("UVC_FUSE_NEXT_LN", "1") not in ("", "0")       # norm1 of block l+1 written by the kernel that produces its input rows
# bf16 mode: the residual stream (the rows every block reads and writes) is bf16 like every other activation; UVC_RESID_F32=1 or
# DistilledVisionTransformer(..., resid_f32=True) keeps the float32 rows of rounds 1-2 (A/B runs).  uvc_vit_cfg.resid_f32.
_RESID_F32_DEFAULT = os.environ.get("UVC_RESID_F32", "0") not in ("", "0")
_FUSED_TRAIN_MLP_DEFAULT = os.environ.get("UVC_FUSED_TRAIN_MLP", "0") not in ("", "0")     # training forward: one MLP kernel instead of three (slower: opt-in)


# Student and teacher see the same batch and the same patch geometry: the [B*196, 768] rearrangement of the images (uvc_patchify,
# 462 MB of traffic at batch 512) is done once per batch by whichever model runs first and handed to the other one through
# uvc_vit_io.patches_in (ordered by an event when they run on different streams).  An entry is consumed once, so a loop that feeds
# the same tensor again (bench.py) still rearranges it once per step.  UVC_SHARE_PATCHES=0 turns the sharing off.
_SHARE_PATCHES = os.environ.get("UVC_SHARE_PATCHES", "1") not in ("", "0")
_PATCH_SHARE = dict(x=None, key=None, buf=None, ev=None, owner=None)


def _shared_patches(model, x, x_key=None):
    """Patch rows of the batch `x` (uvc_patchify), shared between the two models that see the same batch.  A batch is identified by
    the caller's tensor OBJECT (`x_key`: the tensor as the caller passed it, before any dtype / layout conversion) and its version
    counter; the entry keeps a strong reference to it, so its storage cannot be handed to another batch while the entry lives -- a
    (data_ptr, version, shape) key alone matched whatever later batch the caching allocator placed on the freed address."""
    cfg = model._cfg
    x_key = x if x_key is None else x_key
    key = (x_key._version, tuple(x.shape), cfg.patch_size, model.precision)
    c = _PATCH_SHARE
    cur = torch.cuda.current_stream()
    if _SHARE_PATCHES and c["x"] is x_key and c["key"] == key and c["owner"] != id(model):
        buf, ev = c["buf"], c["ev"]
        c.update(x=None, key=None, buf=None, ev=None, owner=None)
        cur.wait_event(ev)
        buf.record_stream(cur)
        return buf
    B = x.shape[0]
    rows = B * (cfg.img_size // cfg.patch_size) ** 2
    buf = torch.empty(rows, cfg.in_chans * cfg.patch_size * cfg.patch_size, device=x.device,
                      dtype=torch.float32 if model.precision == "fp32" else torch.bfloat16)
    ops.patchify(x, buf, cfg.patch_size, ops.UVC_F32 if model.precision == "fp32" else ops.UVC_BF16)
    if _SHARE_PATCHES:
        ev = torch.cuda.Event()
        ev.record(cur)
        c.update(x=x_key, key=key, buf=buf, ev=ev, owner=id(model))
    return buf


def drop_shared_patches():
    """Forget a pending entry (a pass by one model alone -- validation, a teacher-only forward -- leaves one behind; it holds the
    batch and its patch rows alive until the next forward replaces it)."""
    _PATCH_SHARE.update(x=None, key=None, buf=None, ev=None, owner=None)


def _bind():
    lib = L.lib()
    if getattr(lib, "_vit_bound", False):
        return lib
    lib.uvc_vit_layout.argtypes = [C.POINTER(uvc_vit_cfg), C.POINTER(uvc_vit_offsets), C.POINTER(uvc_vit_shadow_offsets)]
    lib.uvc_vit_layout.restype = C.c_int
    lib.uvc_vit_workspace_bytes.argtypes = [C.POINTER(uvc_vit_cfg), C.c_int32, C.c_int32]
    lib.uvc_vit_workspace_bytes.restype = C.c_int64
    lib.uvc_vit_update_shadows.argtypes = [C.POINTER(uvc_vit_cfg), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.uvc_vit_update_shadows.restype = C.c_int
    for n in ("uvc_vit_forward", "uvc_vit_backward"):
        f = getattr(lib, n)
        f.argtypes = [C.POINTER(uvc_vit_cfg), C.POINTER(uvc_vit_io), C.c_void_p]
        f.restype = C.c_int
    lib._vit_bound = True
    return lib


VIT_SYMBOLS = ["uvc_vit_layout", "uvc_vit_workspace_bytes", "uvc_vit_update_shadows", "uvc_vit_forward", "uvc_vit_backward"]


# --------------------------------------------------------------------------------------------------
# module tree with the reference's names (so get_uvc_layers / state_dict / masks work unchanged)
class PatchEmbed(nn.Module):
    """model_distilled.py:129-153 (parameter holder; the conv runs as patchify + MFMA GEMM)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.attn_skip_gating = nn.Parameter(torch.Tensor([-1, 1]))      # :213-214 (unused unless part gating)
        self.mlp_skip_gating = nn.Parameter(torch.Tensor([-1, 1]))


class _VitFunction(torch.autograd.Function):
    """One autograd node for the whole model.  Gradients of ALL parameters are written straight into the
    flat gradient buffer (the ``.grad`` views) by the HIP backward; nothing flows back through autograd."""

    @staticmethod
    def forward(ctx, model, x, anchor, tau, ratio):
        logits, logits_dist = model._run_forward(x, tau, ratio, training=True)
        ctx.model = model
        ctx.two = logits_dist is not None
        return (logits, logits_dist) if ctx.two else logits

    @staticmethod
    def backward(ctx, *grads):
        ctx.model._run_backward(grads[0], grads[1] if ctx.two else None)
        return None, None, None, None, None


class DistilledVisionTransformer(nn.Module):
    """Same call surface as the reference class (model_distilled.py:390-531)."""

    N_EXTRA = 4      # spare floats behind the flat gradient buffer (all-reduce piggy-back slots)
    _ddp = None

    def __init__(self, enable_dist, enable_jumping=0, enable_block_gating=0, enable_part_gating=0,
                 enable_patch_gating=0, gumbel_hard=True, use_gumbel=False, eps=0.1, enable_warmup=False,
                 patch_hard=False, *, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768,
                 depth=12, num_heads=12, mlp_ratio=4., qkv_bias=True, representation_size=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=None, act_layer=None, weight_init='',
                 precision="bf16", device=None, resid_f32=None):
        super().__init__()
        if drop_rate or attn_drop_rate or drop_path_rate or representation_size:
            raise NotImplementedError("dropout / drop-path / representation layer are 0/None on the UVC path "
                                      "(joint_train.py:137-138) and are not implemented")
        if not qkv_bias:
            raise NotImplementedError("qkv_bias=False (T2T blocks) is not on the DeiT hot path")
        if precision == "bf16_f32resid":      # the bf16 mode of rounds 1-2: bf16 operands, float32 residual-stream rows (A/B runs)
            precision, resid_f32 = "bf16", True
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' (throughput), 'bf16_f32resid' (bf16 operands, float32 residual rows) or 'fp32' "
                             "(exact float32 MFMA, parity mode)")
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise L.UvcHipError("uvc_amd models run on MI355X only (no CPU fallback)")
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 2 if enable_dist else 1
        self.precision = precision
        self.gumbel_hard = gumbel_hard
        self.patch_hard = patch_hard
        self.enable_block_gating = enable_block_gating
        self.enable_part_gating = enable_part_gating
        self.enable_jumping = enable_jumping
        self.enable_patch_gating = enable_patch_gating
        self.use_gumbel = use_gumbel
        self.eps = eps
        self.enable_warmup = enable_warmup
        self.frozen_weights = False          # set on the teacher: shadows are refreshed once
        self.two_stream_backward = True      # weight-gradient GEMMs on a side stream, overlapping the dgrad chain
        self._wgrad_stream = None
        self.grad_accumulate = False         # True: backward adds into .grad (gradient_accumulation_steps > 1)
        # --- registration order follows the reference so state_dict keys line up
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, (img_size // patch_size) ** 2 + self.num_tokens, embed_dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if enable_dist else None
        self.block_skip_gating = nn.Parameter(torch.Tensor([-1, 1]).expand(depth, 2).contiguous())
        self.patch_gating = nn.Parameter(torch.zeros(1, (img_size // patch_size) ** 2, 1)) if enable_patch_gating == 1 else None
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pre_logits = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes)
        self.head_dist = nn.Linear(embed_dim, num_classes) if enable_dist else None
        self.gumbel = nn.Linear(embed_dim, 1)
        if self.enable_block_gating:
            print("=====> Block gating enabled <=====")
        self._init_weights()
        # --- engine state
        self._cfg = uvc_vit_cfg(img_size, patch_size, in_chans, num_classes, embed_dim, depth, num_heads,
                                int(embed_dim * mlp_ratio), self.num_tokens, ops.UVC_F32 if precision == "fp32" else ops.UVC_BF16)
        self.resid_f32 = bool(_RESID_F32_DEFAULT if resid_f32 is None else resid_f32) or precision == "fp32"
        self._cfg.resid_f32 = int(self.resid_f32)
        self._off = uvc_vit_offsets()
        self._soff = uvc_vit_shadow_offsets()
        L.check(_bind().uvc_vit_layout(C.byref(self._cfg), C.byref(self._off), C.byref(self._soff)), "uvc_vit_layout")
        self._flat = None
        self._flat_grad = None
        self._ws = {}
        self._shadow = None
        self._last = None
        self._run_block_host = None
        self._run_block_ver = -1
        self._flat_mask = None
        self._mlp_compact = None
        self._mlp_bufs = None
        self._head_keep = None
        self._skip_grads_clean = False
        self._front_state = None
        self.exp_source = lambda shape: torch.empty(shape, device=self._flat.device, dtype=torch.float32).exponential_()
        self.to(dev)

    # -- init ----------------------------------------------------------------------------------------
    def _init_weights(self):
        """model_distilled.py:65-97,311-319,423-427: trunc_normal(.02) weights/tokens, zero biases, LN 1/0."""
        for t in (self.pos_embed, self.cls_token, self.dist_token):
            if t is not None:
                nn.init.trunc_normal_(t, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)

    # -- flat storage ---------------------------------------------------------------------------------
    @property
    def n_flat(self):
        """Elements of the flat parameter buffer (the engine's layout; T2T-ViT appends its tokens-to-token parameters)."""
        return self._off.n_total

    def _extra_live_segments(self):
        """(offset, count) ranges behind the engine's layout that receive gradients every step (none for DeiT)."""
        return []

    def _frozen_ranges(self):
        """(offset, count) ranges the optimiser must never touch (requires_grad False tensors inside the live segments)."""
        return []

    def _front_end_forward(self, x, B, training):
        """Hook for a token embedding other than the patch convolution: write pe into the engine workspace and return True."""
        return False

    def _front_end_backward(self, st):
        pass

    def _slots(self):
        """(parameter, offset) pairs of the canonical flat layout (uvc_vit_layout)."""
        o = self._off
        out = [(self.cls_token, o.cls_token), (self.pos_embed, o.pos_embed), (self.patch_embed.proj.weight, o.patch_w),
               (self.patch_embed.proj.bias, o.patch_b)]
        if self.dist_token is not None:
            out.append((self.dist_token, o.dist_token))
        for l, blk in enumerate(self.blocks):
            q = o.blk[l]
            out += [(blk.norm1.weight, q[0]), (blk.norm1.bias, q[1]), (blk.attn.qkv.weight, q[2]), (blk.attn.qkv.bias, q[3]),
                    (blk.attn.proj.weight, q[4]), (blk.attn.proj.bias, q[5]), (blk.norm2.weight, q[6]), (blk.norm2.bias, q[7]),
                    (blk.mlp.fc1.weight, q[8]), (blk.mlp.fc1.bias, q[9]), (blk.mlp.fc2.weight, q[10]), (blk.mlp.fc2.bias, q[11]),
                    (blk.attn_skip_gating, o.skip[l][0]), (blk.mlp_skip_gating, o.skip[l][1])]
        out += [(self.norm.weight, o.norm_w), (self.norm.bias, o.norm_b), (self.head.weight, o.head_w), (self.head.bias, o.head_b),
                (self.block_skip_gating, o.gate), (self.gumbel.weight, o.gumbel_w), (self.gumbel.bias, o.gumbel_b)]
        if self.head_dist is not None:
            out += [(self.head_dist.weight, o.headd_w), (self.head_dist.bias, o.headd_b)]
        if self.patch_gating is not None:
            out.append((self.patch_gating, o.patch_gating))
        return out

    def _flatten(self, device):
        """(Re)build the flat parameter/gradient buffers and point every Parameter at its slice."""
        n = self.n_flat
        flat = torch.zeros(n, device=device, dtype=torch.float32)
        grad = torch.zeros(n + self.N_EXTRA, device=device, dtype=torch.float32)   # + comm scratch (dual scalar)
        for p, off in self._slots():
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[off:off + k].view(p.shape)
            p.grad = None
        self._flat, self._flat_grad = flat, grad
        tsz = 4 if self.precision == "fp32" else 2
        self._shadow = torch.zeros(self._soff.n_total * tsz, device=device, dtype=torch.uint8)
        self._ws = {}
        self._shadow_fresh = False
        self._run_block_host = None

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        dev = self.cls_token.device
        if dev.type == "cuda" and hasattr(self, "_off"):
            self._flatten(dev)
        return self

    def _check_flat(self):
        if self._flat is None or self.cls_token.data_ptr() != self._flat.data_ptr() + 4 * self._off.cls_token or \
                self.head.weight.data_ptr() != self._flat.data_ptr() + 4 * self._off.head_w:
            self._flatten(self.cls_token.device)
        # the UVC minimax object swaps in its own patch_gating parameter (uvc_utils.py:286-288)
        if self.patch_gating is not None and self.patch_gating.data_ptr() != self._flat.data_ptr() + 4 * self._off.patch_gating:
            off, k = self._off.patch_gating, self.patch_gating.numel()
            self._flat[off:off + k].copy_(self.patch_gating.data.reshape(-1))
            self.patch_gating.data = self._flat[off:off + k].view(self.patch_gating.shape)

    def grad_views(self, patch_mode2=False):
        """Point .grad of every parameter that receives a gradient at its slice of the flat gradient
        buffer (idempotent).  Like autograd in the reference, tensors the loss does not reach keep
        .grad = None: attn/mlp_skip_gating always, gumbel.* unless patch-gating mode 2 ran,
        block_skip_gating in warm-up (requires_grad False / constant .5,.5 gates)."""
        o = self._off
        dead = {o.skip[l][j] for l in range(self._cfg.depth) for j in (0, 1)}
        if not patch_mode2:
            dead |= {o.gumbel_w, o.gumbel_b}
        if not self.block_skip_gating.requires_grad or self._gate_mode() == 0:
            dead.add(o.gate)
        skipped = self.skipped_block_ranges()
        for p, off in self._slots():
            if off in dead:
                continue
            if any(a <= off < a + n for a, n in skipped):
                p.grad = None
                continue
            if p.grad is None or p.grad.data_ptr() != self._flat_grad.data_ptr() + 4 * off:
                p.grad = self._flat_grad[off:off + p.numel()].view(p.shape)

    def mark_weights_changed(self):
        self._shadow_fresh = False

    def no_weight_decay(self):
        """model_distilled.py:330-331 (read by timm's create_optimizer in Stage-2)."""
        return {"pos_embed", "cls_token", "dist_token"}

    # -- Stage-2 masks ---------------------------------------------------------------------------------
    def _mask_flat(self):
        """One float32 buffer congruent with the flat parameter buffer: every module's ``mask`` buffer
        (joint_train.py:169-171 / post_train.py:155-157) is a view at its weight's offset, 1 elsewhere.  Mask buffers
        that were registered or moved after the fact are adopted (copied in and re-pointed) here."""
        if self._flat_mask is None or self._flat_mask.device != self._flat.device:
            self._flat_mask = torch.ones(self.n_flat, device=self._flat.device, dtype=torch.float32)
        base = self._flat_mask.data_ptr()
        off_of = {id(p): off for p, off in self._slots()}
        for _, m in self.named_modules():
            mask = m._buffers.get("mask") if hasattr(m, "_buffers") else None
            w = getattr(m, "weight", None)
            if mask is None or w is None or id(w) not in off_of:
                continue
            off, k = off_of[id(w)], w.numel()
            if mask.data_ptr() != base + 4 * off:
                if mask.numel() != k:
                    raise L.UvcHipError("mask buffer does not match its weight")
                self._flat_mask[off:off + k].copy_(mask.reshape(-1).to(device=self._flat.device, dtype=torch.float32))
                m._buffers["mask"] = self._flat_mask[off:off + k].view(w.shape)
        return self._flat_mask

    def set_mlp_compaction(self, keep=None, multiple=256):
        """Skip pruned MLP hidden units (Stage-2 structured sparsity, uvc_vit.h: uvc_mlp_compact).  ``keep``: per layer a
        bool tensor [hidden] of the units that are NOT pruned, or None to derive it from the mask buffers (a unit is pruned
        when its fc1 mask row and its fc2 mask column are both all-zero).  The compact width is the kept count rounded up to
        ``multiple`` (the streaming GEMMs take 256 / 512 / 576 / 768) and padded with pruned units, which is exact.
        ``keep=False`` switches compaction off.  Returns the list of widths."""
        cfg = self._cfg
        Lz, D, F = cfg.depth, cfg.embed_dim, cfg.hidden
        # (how the table was made: one derived from the mask buffers is derived again when load_state_dict replaces them)
        self._mlp_compact_derived = int(multiple) if keep is None else None
        if keep is False:
            self._mlp_compact, self._mlp_bufs = None, None
            return [F] * Lz
        self._check_flat()
        dev = self._flat.device
        tdt = torch.float32 if self.precision == "fp32" else torch.bfloat16
        arr = (uvc_mlp_compact * Lz)()
        bufs, widths = [], []
        for l, blk in enumerate(self.blocks):
            if keep is None:
                m1, m2 = getattr(blk.mlp.fc1, "mask", None), getattr(blk.mlp.fc2, "mask", None)
                k = torch.ones(F, dtype=torch.bool) if m1 is None or m2 is None else ((m1 != 0).any(dim=1) | (m2 != 0).any(dim=0)).cpu()
            else:
                k = keep[l].cpu().bool()
            kept = torch.nonzero(k).flatten()
            width = min(F, max(multiple, -(-int(kept.numel()) // multiple) * multiple))
            if width >= F:
                widths.append(F); bufs.append(None)
                continue
            pad = torch.nonzero(~k).flatten()[: width - kept.numel()]
            idx = torch.cat([kept, pad]).to(torch.int32)
            inv = torch.full((F,), -1, dtype=torch.int32)
            inv[idx.long()] = torch.arange(width, dtype=torch.int32)
            b = dict(idx=idx.to(dev), inv=inv.to(dev), width=width,
                     w1=torch.empty(width, D, device=dev, dtype=tdt), w1t=torch.empty(D, width, device=dev, dtype=tdt),
                     w2=torch.empty(D, width, device=dev, dtype=tdt), w2t=torch.empty(width, D, device=dev, dtype=tdt),
                     b1=torch.empty(width, device=dev), dw1=torch.empty(width, D, device=dev), dw2=torch.empty(D, width, device=dev),
                     db1=torch.empty(width, device=dev))
            e = arr[l]
            e.width = width
            for n in ("w1", "w1t", "w2", "w2t", "b1", "dw1", "dw2", "db1", "inv"):
                setattr(e, n, L.ptr(b[n]))
            bufs.append(b); widths.append(width)
        if all(b is None for b in bufs):
            self._mlp_compact, self._mlp_bufs = None, None
        else:
            self._mlp_compact, self._mlp_bufs = arr, bufs
        self._shadow_fresh = False
        return widths

    def set_head_skipping(self, keep=None):
        """No-grad forwards (eval) skip the attention of pruned heads.  ``keep``: bool [depth, heads], or None to derive it from
        the masks (a head is pruned when every attn.proj mask column of its 64 inputs is zero: its output then meets only
        zero weights, so skipping it is exact once ``apply_masks()`` has run); ``False`` switches it off.  Training forwards
        never skip heads: the reference's global-norm clip includes the gradients of the masked proj columns."""
        self._head_keep_derived = keep is None
        if keep is False:
            self._head_keep = None
            return None
        cfg = self._cfg
        if keep is None:
            rows = []
            for blk in self.blocks:
                m = getattr(blk.attn.proj, "mask", None)
                rows.append(torch.ones(cfg.num_heads, dtype=torch.bool) if m is None else
                            (m != 0).any(dim=0).reshape(cfg.num_heads, -1).any(dim=1).cpu())
            keep = torch.stack(rows)
        keep = keep.to(torch.int32).contiguous()
        self._head_keep = None if bool(keep.all()) else keep.to(self._flat.device)
        return keep

    def _gather_compact(self, stream):
        """Refresh the gathered operand copies of the compacted MLPs from the (masked) master weights."""
        dt = ops.UVC_F32 if self.precision == "fp32" else ops.UVC_BF16
        cfg = self._cfg
        for blk, b in zip(self.blocks, self._mlp_bufs):
            if b is None:
                continue
            L.check(L.lib().uvc_mlp_gather_shadows(L.ptr(blk.mlp.fc1.weight.data), L.ptr(blk.mlp.fc1.bias.data), L.ptr(blk.mlp.fc2.weight.data),
                                                   L.ptr(b["idx"]), cfg.embed_dim, cfg.hidden, b["width"], L.ptr(b["w1"]), L.ptr(b["w1t"]),
                                                   L.ptr(b["w2"]), L.ptr(b["w2t"]), L.ptr(b["b1"]), dt, stream), "uvc_mlp_gather_shadows")

    def apply_masks(self):
        """``for m in modules: m.weight.data *= m.mask`` (post_train.py:343-346) as one launch over the flat buffer."""
        self._check_flat()
        ops.apply_masks(self._flat, self._mask_flat())
        self.mark_weights_changed()

    def load_state_dict(self, *a, **k):
        """nn.Module.load_state_dict, then everything that was DERIVED from the mask buffers is derived again from the loaded ones: the
        pruned-head table (since r4 the training backward writes dq / dk / dv of a head marked pruned as zeros -- a table from other
        masks would silently drop the gradients of live heads) and the MLP compaction.  Tables the caller passed explicitly
        (``keep=`` a tensor) are the caller's to refresh."""
        r = super().load_state_dict(*a, **k)
        self.mark_weights_changed()
        self._run_block_host = None
        if getattr(self, "_head_keep_derived", False):
            self.set_head_skipping()
        if getattr(self, "_mlp_compact_derived", None) is not None:
            self.set_mlp_compaction(multiple=self._mlp_compact_derived)
        return r

    # -- engine calls ---------------------------------------------------------------------------------
    def _ws_mode(self, training):
        """uvc_vit_workspace_bytes' `training` argument: 0 = no-grad forward; 1 = training with per-block copies of the backward's streams (what the weight
        gradients on the side stream read while the main stream is blocks ahead); 2 = training WITHOUT the two-stream backward: one shared set, L x (5 M D + M F)
        elements less (17 GB for DeiT-Base at batch 512).  Same results bit for bit (tests/test_streaming_batch_gpu.py)."""
        if not training:
            return 0
        return 1 if self.two_stream_backward else 2

    def _workspace(self, B, training):
        mode = self._ws_mode(training)
        key = (B, bool(training), mode)
        if key not in self._ws:
            nbytes = _bind().uvc_vit_workspace_bytes(C.byref(self._cfg), B, mode)
            if nbytes < 0:
                raise L.UvcHipError(f"uvc_vit_workspace_bytes: {L.lib().uvc_last_error().decode()}")
            self._ws = {k: v for k, v in self._ws.items() if k[1] != bool(training)}   # keep one per mode
            self._ws[key] = torch.empty(nbytes, device=self._flat.device, dtype=torch.uint8)
        return self._ws[key]

    def _gate_mode(self):
        if not self.enable_block_gating or self.enable_warmup:
            return 0
        return 1 if self.use_gumbel == 1 else 2

    def _io(self, B, training):
        io = uvc_vit_io()
        ws = self._workspace(B, training)
        io.params, io.shadow, io.grads = L.ptr(self._flat), L.ptr(self._shadow), L.ptr(self._flat_grad)
        io.workspace, io.workspace_bytes = L.ptr(ws), ws.numel()
        io.batch, io.training = B, int(training)
        io.shared_bwd_streams = int(self._ws_mode(training) == 2)
        io.gate_mode, io.gate_eps = self._gate_mode(), float(self.eps)
        io.accumulate = 1.0 if self.grad_accumulate else 0.0
        io.mlp_compact = C.addressof(self._mlp_compact) if self._mlp_compact is not None else None
        # eval forwards skip pruned heads; a training pass hands the table over for the BACKWARD only (uvc_vit_io.head_keep_bwd), and only when
        # the trainer vouches that the masks are applied to the weights every step (Stage2Trainer sets skip_pruned_head_grads)
        skip_bwd = bool(training and getattr(self, "skip_pruned_head_grads", False))
        io.head_keep = L.ptr(self._head_keep) if (self._head_keep is not None and (not training or skip_bwd)) else None
        io.head_keep_bwd = int(skip_bwd and self._head_keep is not None)
        io.full_tail = int(getattr(self, "full_tail", _FULL_TAIL_DEFAULT))
        io.fused_train_mlp = int(getattr(self, "fused_train_mlp", _FUSED_TRAIN_MLP_DEFAULT))
        io.fuse_next_ln = int(getattr(self, "fuse_next_ln", _FUSE_NEXT_LN_DEFAULT))
        io.force_generic = int(getattr(self, "force_generic", 0))      # tests / A-B runs (uvc_vit_io.force_generic)
        io.gelu_grad_bf16 = int(getattr(self, "gelu_grad_bf16", _GELU_GRAD_BF16_DEFAULT))      # 1: bf16 GELU'(a) everywhere (uvc_vit_io.gelu_grad_bf16)
        return io

    def _ws_view(self, B, training, which):
        """float32 view of pe [B*P, D] / T view of dpe inside the workspace (patch-gating hooks)."""
        lib = _bind()
        if not hasattr(lib, "_wsoff_bound"):
            lib.uvc_vit_ws_offsets.argtypes = [C.POINTER(uvc_vit_cfg), C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
            lib.uvc_vit_ws_offsets.restype = C.c_int
            lib._wsoff_bound = True
        pe_off, dpe_off = C.c_int64(), C.c_int64()
        L.check(lib.uvc_vit_ws_offsets(C.byref(self._cfg), B, self._ws_mode(training), C.byref(pe_off), C.byref(dpe_off)), "uvc_vit_ws_offsets")
        ws = self._workspace(B, training)
        cfg = self._cfg
        rows = B * (cfg.img_size // cfg.patch_size) ** 2
        if which == "pe":
            return ws[pe_off.value:pe_off.value + rows * cfg.embed_dim * 4].view(torch.float32).view(rows, cfg.embed_dim)
        tdt = torch.float32 if self.precision == "fp32" else torch.bfloat16
        nb = rows * cfg.embed_dim * (4 if self.precision == "fp32" else 2)
        return ws[dpe_off.value:dpe_off.value + nb].view(tdt).view(rows, cfg.embed_dim)

    def _run_forward(self, x, tau, ratio, training):
        L.require_cuda(x)
        x_key = x                               # the caller's tensor: identifies the batch for the patch-row sharing
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.contiguous().float()
        B = x.shape[0]
        cfg = self._cfg
        if tuple(x.shape[1:]) != (cfg.in_chans, cfg.img_size, cfg.img_size):
            raise AssertionError(f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model ({cfg.img_size}*{cfg.img_size}).")
        self._check_flat()
        lib = _bind()
        stream = L.cur_stream()
        if not (self.frozen_weights and self._shadow_fresh):
            L.check(lib.uvc_vit_update_shadows(C.byref(cfg), L.ptr(self._flat), L.ptr(self._shadow), stream), "uvc_vit_update_shadows")
            if self._mlp_compact is not None:
                self._gather_compact(stream)
            self._shadow_fresh = True
        io = self._io(B, training)
        dev = self._flat.device
        logits = torch.empty(B, cfg.num_classes, device=dev)
        logits_dist = torch.empty(B, cfg.num_classes, device=dev) if self.num_tokens == 2 else None
        gate_d = None
        run_block = None
        if self.enable_block_gating:                                   # model_distilled.py:479-494
            gate_d = torch.empty(cfg.depth, 2, device=dev)
            if self.enable_warmup:
                mode = 0
            elif self.use_gumbel == 1:
                mode = 3 if self.gumbel_hard else 1
            else:
                mode = 2
        else:                                                          # :496-500 hard skip by logit order
            run_block = self._hard_run_blocks()
        io.x, io.logits, io.logits_dist = L.ptr(x), L.ptr(logits), L.ptr(logits_dist)
        io.run_block = run_block if run_block is not None else None
        # ---- patch gating (model_distilled.py:434-456): the mask needs the patch embedding, so the forward is cut there
        P = (cfg.img_size // cfg.patch_size) ** 2
        mode1 = self.enable_patch_gating == 1
        mode2 = tau > 0
        if mode1 and mode2:
            raise NotImplementedError("patch gating modes 1 and 2 together (never produced by the reference driver)")
        patch = None
        # a custom token embedding (T2T-ViT's tokens-to-token module) writes pe [B*P, D] into the workspace itself
        front = self._front_end_forward(x, B, training)
        patches = None
        if front:
            io.stage_begin, io.stage_end = 1, 2
        else:
            patches = _shared_patches(self, x, x_key)   # the batch's patch rows, rearranged once for student and teacher
            io.patches_in = L.ptr(patches)
        if mode1 or mode2:
            if not front:
                io.stage_begin, io.stage_end = 0, 1
                L.check(lib.uvc_vit_forward(C.byref(cfg), C.byref(io), stream), "uvc_vit_forward")
            mask = torch.empty(B, P, device=dev)
            if mode1:
                ops.patch_gate_sigmoid(self.patch_gating.data, mask, B, P, self.patch_hard)
                patch = dict(mode=1, mask=mask)
            else:
                pe = self._ws_view(B, training, "pe")
                scores, ys, ps = (torch.empty(B, P, device=dev) for _ in range(3))
                ops.patch_scores(pe, self.gumbel.weight.data, self.gumbel.bias.data, scores, B * P, cfg.embed_dim)
                ops.patch_topk_mask(scores, self.exp_source((B, P)), mask, ys, ps, B, P, int(ratio * P), float(tau))
                patch = dict(mode=2, mask=mask, ysoft=ys, psoft=ps, tau=float(tau))
            io.patch_mask = L.ptr(mask)
            io.stage_begin, io.stage_end = 1, 2
        if gate_d is not None:     # drawn after the patch-gating noise, in the reference's RNG order (:446-485)
            e = self.exp_source((cfg.depth, 2)) if mode in (1, 3) else None
            ops.gate_distrib(self.block_skip_gating.data, e, gate_d, cfg.depth, mode, float(self.eps))
        io.gate_d = L.ptr(gate_d)
        L.check(lib.uvc_vit_forward(C.byref(cfg), C.byref(io), stream), "uvc_vit_forward")
        self._last = dict(x=x, gate_d=gate_d, patch=patch, B=B, run_block=run_block, front=self._front_state, patches=patches,
                          ws_mode=self._ws_mode(True)) if training else None
        self.last_distrib = gate_d
        self.last_patch_mask = patch["mask"] if patch else None
        return logits, logits_dist

    def _hard_run_blocks(self):
        """Host 0/1 list of the blocks the hard skip keeps (`block_skip_gating[i][1] > [i][0]`, :496-500).  The logits
        live on the device; the host copy is refreshed only when they may have changed (in-place edits bump the
        tensor version, FusedAdamW / load_state_dict / re-flattening drop the cache), so a Stage-2 step never syncs."""
        ver = self.block_skip_gating._version
        if self._run_block_host is None or self._run_block_ver != ver:
            g = self.block_skip_gating.detach().cpu()
            depth = self._cfg.depth
            self._run_block_host = (C.c_int32 * depth)(*[int(g[i, 1] > g[i, 0]) for i in range(depth)])
            self._run_block_ver = ver
            self._skip_grads_clean = False
        return self._run_block_host

    def skipped_block_ranges(self):
        """(offset, count) in the flat buffers of the parameters of hard-skipped blocks (empty with block gating on)."""
        if self.enable_block_gating:
            return []
        run = self._hard_run_blocks()
        o, depth = self._off, self._cfg.depth
        out = []
        for l in range(depth):
            if not run[l]:
                end = o.blk[l + 1][0] if l + 1 < depth else o.norm_w
                out.append((o.blk[l][0], end - o.blk[l][0]))
        return out

    def _patch_backward(self, st, dmask):
        """Gradients of the patch-gating parameters from d(mask) (model_distilled.py:434-456)."""
        cfg = self._cfg
        B, pt = st["B"], st["patch"]
        P = (cfg.img_size // cfg.patch_size) ** 2
        beta = 1.0 if self.grad_accumulate else 0.0
        if pt["mode"] == 1:
            ops.patch_gate_sigmoid_bwd(self.patch_gating.data, dmask, self.patch_gating.grad, B, P, beta)
            return
        dev = self._flat.device
        ds = torch.empty(B, P, device=dev)
        ops.patch_topk_mask_bwd(dmask, pt["ysoft"], pt["psoft"], ds, B, P, pt["tau"])
        pe = self._ws_view(B, True, "pe")
        dt = ops.UVC_F32 if self.precision == "fp32" else ops.UVC_BF16
        part = torch.empty(ops.colsum_blocks(B * P) * cfg.embed_dim, device=dev)
        ops.colsum(pe, part, self.gumbel.weight.grad.view(-1), dt, row_weight=ds.view(-1), beta=beta)       # dW = sum ds * pe
        ops.colsum(ds.view(-1, 1), part, self.gumbel.bias.grad.view(-1), dt, beta=beta)                      # db = sum ds
        ops.add_outer(self._ws_view(B, True, "dpe"), ds.view(-1), self.gumbel.weight.data.view(-1), B * P, cfg.embed_dim, dt)

    def _run_backward(self, d_logits, d_logits_dist):
        st = self._last
        if st is None:
            raise RuntimeError("backward without a training forward")
        patch = st["patch"]
        if st.get("ws_mode", self._ws_mode(True)) != self._ws_mode(True):
            raise RuntimeError("two_stream_backward was changed between a training forward and its backward (the two lay the workspace out differently)")
        self.grad_views(patch_mode2=bool(patch and patch["mode"] == 2))
        io = self._io(st["B"], True)
        io.patches_in = L.ptr(st.get("patches"))
        d_logits = d_logits.contiguous()
        io.d_logits = L.ptr(d_logits)
        if self.num_tokens == 2:
            d_logits_dist = d_logits_dist.contiguous()
            io.d_logits_dist = L.ptr(d_logits_dist)
        io.gate_d = L.ptr(st["gate_d"])
        io.run_block = st["run_block"] if st["run_block"] is not None else None
        if st["run_block"] is not None and not self._skip_grads_clean:
            # hard-skipped blocks get no gradient (.grad stays None); their slice of the flat gradient buffer is zeroed
            # once so the global-norm pass over the whole buffer sees nothing there
            for off, n in self.skipped_block_ranges():
                self._flat_grad[off:off + n].zero_()
            self._skip_grads_clean = True
        if self.two_stream_backward:
            if self._wgrad_stream is None:
                self._wgrad_stream = L.side_stream(self._flat.device)
            io.side_stream = self._wgrad_stream.cuda_stream
        dmask = None
        if patch:
            io.patch_mask = L.ptr(patch["mask"])
            dmask = torch.empty_like(patch["mask"])
            io.d_patch_mask = L.ptr(dmask)
        ddp = self._ddp if (self._ddp is not None and self._ddp.world > 1) else None
        depth = self._cfg.depth
        cuts = set(ddp.stage_ends) if ddp else {depth + 3}
        if patch:
            cuts.add(depth + 2)            # after token assembly: d(mask) is known, dpe not yet consumed
        front = st.get("front")            # custom front end: it consumes dpe instead of the patch-embedding wgrad (stage depth+2)
        begin = 0
        for end in sorted(cuts):
            io.stage_begin, io.stage_end = begin, (min(end, depth + 2) if front else end)
            if io.stage_end > io.stage_begin:
                L.check(_bind().uvc_vit_backward(C.byref(self._cfg), C.byref(io), L.cur_stream()), "uvc_vit_backward")
            if front and end == depth + 3:
                self._front_end_backward(st)
            if patch and end == depth + 2:
                self._patch_backward(st, dmask)
            if ddp and end in ddp.stage_ends:
                i = ddp.stage_ends.index(end)
                if i == len(ddp.stage_ends) - 1:
                    ddp.pack_dual()
                ddp.reducer.launch(i)      # RCCL on the side stream, overlaps the remaining stages
            begin = end
        if ddp:
            ddp.reducer.finish()
            ddp.unpack_dual()

    # -- reference API ----------------------------------------------------------------------------------
    def macs(self, B):
        """MAC bookkeeping of the reference forward (model_distilled.py:115,121,177,182,185,189,460)."""
        c = self._cfg
        P = (c.img_size // c.patch_size) ** 2
        N, D, F, H = P + self.num_tokens, c.embed_dim, c.hidden, c.num_heads
        embed = B * P * D * c.patch_size * c.patch_size * c.in_chans
        blk = [B * 3 * D * N * D, N * B * H * N * 64, N * B * H * N * 64, B * N * D * D, F * B * N * D, D * B * N * F]
        return embed, [list(blk) for _ in range(c.depth)]

    def forward(self, x, tau=-1, number=0.9):
        if self.enable_jumping:
            raise NotImplementedError("enable_jumping is off on the UVC hot path")
        macs = self.macs(x.shape[0])
        if not self.enable_block_gating:                      # skipped blocks report an empty MAC list (:496-500)
            run = self._hard_run_blocks()
            macs = (macs[0], [m if run[l] else [] for l, m in enumerate(macs[1])])
        if self.training and torch.is_grad_enabled():
            out = _VitFunction.apply(self, x, self.cls_token, tau, number)
            return (out if self.num_tokens == 2 else (out, out)), macs       # x_dist = x without the dist token (:523-524)
        o, od = self._run_forward(x, tau, number, training=False)
        if od is None:
            od = o
        if self.training:
            return (o, od), macs
        if od is o:                                           # one token: (o + o) / 2 is o, bit for bit -- no launches (the teacher's forward, every step)
            return o, macs
        return (o + od) / 2, macs
