"""MI355X mirror of ``UVC/T2TViT/models/t2t_vit.py``: ``T2T_ViT`` / ``t2t_vit_14`` with the reference's constructor,
state_dict keys and ``forward(x) -> (logits | (logits, logits), (macs_embed, macs_list))`` contract.

The transformer blocks, final norm, head and block gating run on the same engine as DeiT (uvc_vit_forward /
uvc_vit_backward with LayerNorm eps 1e-5 and no qkv bias); the tokens-to-token module (soft split -> Performer ->
soft split -> Performer -> soft split -> project, t2t_vit.py:84-105, token_performer.py:31-69) is sequenced here from
the kernels of include/uvc_t2t.h and the GEMM / LayerNorm kernels of include/uvc_kernels.h and writes the token
embedding straight into the engine's workspace.  Its parameters live behind the engine's layout in the same flat
float32 buffer, so the fused clip + AdamW and the RCCL bucket all-reduce cover them without extra launches.

What the reference defines and what this defines (the reference's gated T2T forward raises as shipped):
  * eval / hard block skip forward: as the reference (pinned by tests/golden/t2t_*.npz);
  * block gating: the lines at t2t_vit.py:181-189 as written (== model_distilled.py:480-494);
  * ``block_skip_gating`` is a real [depth, 2] parameter (the reference's rows alias one storage through .expand(), :139);
  * the Performer's Dropout(0.1) layers (token_performer.py:13,24) are not applied in training mode: the UVC path runs
    DeiT with drop_rate 0 (joint_train.py:137) and the engine has no RNG-dependent layers besides the gates;
  * ``enable_patch_gating=2`` (BASELINE config 5, joint_train.py:404-410 calls ``model(x, tau, ratio)``): the reference's
    T2T forward_features has no patch-gating step at all, so it is DEFINED here by transplanting DeiT's
    (model_distilled.py:446-456) onto the tokens the tokens-to-token module produces: scorer ``gumbel = Linear(D -> 1)``
    (an extra module, only registered in this mode, so the reference's state_dict key set is unchanged otherwise),
    log_softmax -> Gumbel top-k with k = int(ratio * 196) -> straight-through mask, mask[:, 0] = 1, tokens multiplied
    before cls / pos are added (oracle/t2t.py:forward, UNPINNED).  Mode 1 (a learnt per-position sigmoid) is not defined
    for T2T and raises.
There is no CPU path: constructing the model needs an MI355X.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .model_distilled import (Block, DistilledVisionTransformer, _bind, uvc_vit_cfg, uvc_vit_offsets, uvc_vit_shadow_offsets)

__all__ = ["T2T_ViT", "T2T_module", "Token_performer", "t2t_vit_14", "get_sinusoid_encoding"]

LN_EPS = 1e-5


def get_sinusoid_encoding(n_position, d_hid):
    """transformer_block.py:115-125."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    table = pos / np.power(10000.0, 2 * (np.arange(d_hid) // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.from_numpy(table.astype(np.float32)).unsqueeze(0)


class Token_performer(nn.Module):
    """Parameter holder with the reference's names (token_performer.py:8-29)."""

    def __init__(self, dim, in_dim, head_cnt=1, kernel_ratio=0.5, dp1=0.1, dp2=0.1):
        super().__init__()
        self.emb = in_dim * head_cnt
        self.kqv = nn.Linear(dim, 3 * self.emb)
        self.dp = nn.Dropout(dp1)
        self.proj = nn.Linear(self.emb, self.emb)
        self.head_cnt = head_cnt
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(self.emb)
        self.epsilon = 1e-8
        self.mlp = nn.Sequential(nn.Linear(self.emb, self.emb), nn.GELU(), nn.Linear(self.emb, self.emb), nn.Dropout(dp2))
        self.m = int(self.emb * kernel_ratio)
        self.w = nn.Parameter(nn.init.orthogonal_(torch.randn(self.m, self.emb)) * math.sqrt(self.m), requires_grad=False)


class T2T_module(nn.Module):
    """t2t_vit.py:46-82 (performer tokens type)."""

    def __init__(self, img_size=224, tokens_type="performer", in_chans=3, embed_dim=768, token_dim=64):
        super().__init__()
        if tokens_type != "performer":
            raise NotImplementedError("tokens_type 'performer' is the one t2t_vit_14 uses (t2t_vit.py:248)")
        self.attention1 = Token_performer(dim=in_chans * 7 * 7, in_dim=token_dim, kernel_ratio=0.5)
        self.attention2 = Token_performer(dim=token_dim * 3 * 3, in_dim=token_dim, kernel_ratio=0.5)
        self.project = nn.Linear(token_dim * 3 * 3, embed_dim)
        self.num_patches = (img_size // (4 * 2 * 2)) * (img_size // (4 * 2 * 2))


def _al4(n):
    return (n + 3) & ~3


class T2T_ViT(DistilledVisionTransformer):
    """Same call surface as the reference class (t2t_vit.py:107-208)."""

    def __init__(self, img_size=224, tokens_type="performer", in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 token_dim=64, enable_block_gating=False, enable_jumping=False, enable_patch_gating=0, gumbel_hard=True, use_gumbel=False,
                 *, eps=0.1, enable_warmup=False, precision="bf16", device=None):
        nn.Module.__init__(self)
        if drop_rate or attn_drop_rate or drop_path_rate:
            raise NotImplementedError("dropout / drop-path are 0 on the UVC path")
        if qkv_bias or qk_scale is not None:
            raise NotImplementedError("t2t_vit_14 as built by joint_train.py:145: qkv_bias=False, default qk_scale")
        if token_dim != 64 or in_chans != 3 or img_size % 16:
            raise NotImplementedError("tokens-to-token kernels: token_dim 64, 3 input channels, img_size a multiple of 16")
        resid_f32 = None
        if precision == "bf16_f32resid":
            precision, resid_f32 = "bf16", True
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16', 'bf16_f32resid' or 'fp32'")
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise L.UvcHipError("uvc_amd models run on MI355X only (no CPU fallback)")
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1
        self.precision = precision
        self.gumbel_hard = gumbel_hard
        self.patch_hard = False
        self.enable_block_gating = enable_block_gating
        self.enable_part_gating = 0
        self.enable_jumping = enable_jumping
        if enable_patch_gating not in (0, 2):
            raise NotImplementedError("T2T-ViT: enable_patch_gating 0 or 2 (Gumbel top-k, defined in this module's docstring)")
        self.t2t_enable_patch_gating = enable_patch_gating        # 2: forward(x, tau > 0, ratio) gates the tokens
        self.enable_patch_gating = 0                              # mode 1 (model_distilled.py:434-444) does not exist here
        self.use_gumbel = use_gumbel
        self.eps = eps
        self.enable_warmup = enable_warmup
        self.frozen_weights = False
        self.front_in_c = True                # a Token_performer stage is one C call (uvc_t2t_stage_forward / _backward); False: the same launches from Python (A/B, tests)
        self.two_stream_backward = True
        self._wgrad_stream = None
        self.grad_accumulate = False
        self.dist_token = None
        self.patch_gating = None
        self.head_dist = None
        self.gumbel = None
        # --- registration order of the reference (t2t_vit.py:115-139)
        self.tokens_to_token = T2T_module(img_size=img_size, tokens_type=tokens_type, in_chans=in_chans, embed_dim=embed_dim, token_dim=token_dim)
        num_patches = self.tokens_to_token.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(data=get_sinusoid_encoding(n_position=num_patches + 1, d_hid=embed_dim), requires_grad=False)
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, False, nn.LayerNorm) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes)
        self.block_skip_gating = nn.Parameter(torch.Tensor([-1, 1]).expand(depth, 2).contiguous())
        if enable_patch_gating == 2:
            self.gumbel = nn.Linear(embed_dim, 1)                 # the token scorer of model_distilled.py:419
        nn.init.trunc_normal_(self.cls_token, std=.02)
        for m in self.modules():                                   # _init_weights (:144-151)
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.zeros_(m.bias)
                nn.init.ones_(m.weight)
        # --- engine state: the DeiT engine with T2T's LayerNorm eps and bias-free qkv; "patch_size" 16 only sizes the sequence
        self._cfg = uvc_vit_cfg(img_size, 16, in_chans, num_classes, embed_dim, depth, num_heads, int(embed_dim * mlp_ratio), 1,
                                ops.UVC_F32 if precision == "fp32" else ops.UVC_BF16, LN_EPS, 1)
        from .model_distilled import _RESID_F32_DEFAULT
        self.resid_f32 = bool(_RESID_F32_DEFAULT if resid_f32 is None else resid_f32) or precision == "fp32"
        self._cfg.resid_f32 = int(self.resid_f32)
        self._off = uvc_vit_offsets()
        self._soff = uvc_vit_shadow_offsets()
        L.check(_bind().uvc_vit_layout(C.byref(self._cfg), C.byref(self._off), C.byref(self._soff)), "uvc_vit_layout")
        # tokens-to-token parameters behind the engine's layout.  attention1.kqv.weight is stored with its 147 input columns
        # padded to 160 (the GEMM's K granularity); the pad columns are zero, receive zero gradients and stay zero.
        o = self._off.n_total
        self._front = {}
        for name, att, dim in (("attention1", self.tokens_to_token.attention1, in_chans * 49), ("attention2", self.tokens_to_token.attention2, token_dim * 9)):
            dimp = -(-dim // 32) * 32
            f = dict(dim=dim, dimp=dimp)
            for key, n in (("w", 32 * 64), ("kqv_w", 192 * dimp), ("kqv_b", 192), ("proj_w", 4096), ("proj_b", 64), ("norm1_w", dim), ("norm1_b", dim),
                           ("norm2_w", 64), ("norm2_b", 64), ("fc1_w", 4096), ("fc1_b", 64), ("fc2_w", 4096), ("fc2_b", 64)):
                f[key] = o
                o += _al4(n)
            self._front[name] = f
        self._front["project_w"] = o
        o += _al4(embed_dim * 576)
        self._front["project_b"] = o
        o += _al4(embed_dim)
        self._front_begin, self._n_flat = self._off.n_total, o
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
        self._front_bufs = {}
        self._front_shadow = None
        self.exp_source = lambda shape: torch.empty(shape, device=self._flat.device, dtype=torch.float32).exponential_()
        self.to(dev)

    # -- flat storage ---------------------------------------------------------------------------------
    @property
    def n_flat(self):
        return self._n_flat

    def _extra_live_segments(self):
        return [(self._front_begin, self._n_flat - self._front_begin)]

    def _frozen_ranges(self):
        out = [(self._off.pos_embed, self.pos_embed.numel())]
        for name in ("attention1", "attention2"):
            out.append((self._front[name]["w"], 32 * 64))
        return out

    def _front_slots(self):
        out = []
        for name in ("attention1", "attention2"):
            a, f = getattr(self.tokens_to_token, name), self._front[name]
            out += [(a.w, f["w"]), (a.kqv.bias, f["kqv_b"]), (a.proj.weight, f["proj_w"]), (a.proj.bias, f["proj_b"]),
                    (a.norm1.weight, f["norm1_w"]), (a.norm1.bias, f["norm1_b"]), (a.norm2.weight, f["norm2_w"]), (a.norm2.bias, f["norm2_b"]),
                    (a.mlp[0].weight, f["fc1_w"]), (a.mlp[0].bias, f["fc1_b"]), (a.mlp[2].weight, f["fc2_w"]), (a.mlp[2].bias, f["fc2_b"])]
            if f["dim"] == f["dimp"]:
                out.append((a.kqv.weight, f["kqv_w"]))
        out += [(self.tokens_to_token.project.weight, self._front["project_w"]), (self.tokens_to_token.project.bias, self._front["project_b"])]
        return out

    def _padded(self):
        """(parameter, offset, rows, dim, dimp) of the weights stored with padded rows."""
        out = []
        for name in ("attention1", "attention2"):
            a, f = getattr(self.tokens_to_token, name), self._front[name]
            if f["dim"] != f["dimp"]:
                out.append((a.kqv.weight, f["kqv_w"], 192, f["dim"], f["dimp"]))
        return out

    def _slots(self):
        o = self._off
        out = [(self.cls_token, o.cls_token), (self.pos_embed, o.pos_embed)]
        for l, blk in enumerate(self.blocks):
            q = o.blk[l]
            out += [(blk.norm1.weight, q[0]), (blk.norm1.bias, q[1]), (blk.attn.qkv.weight, q[2]),
                    (blk.attn.proj.weight, q[4]), (blk.attn.proj.bias, q[5]), (blk.norm2.weight, q[6]), (blk.norm2.bias, q[7]),
                    (blk.mlp.fc1.weight, q[8]), (blk.mlp.fc1.bias, q[9]), (blk.mlp.fc2.weight, q[10]), (blk.mlp.fc2.bias, q[11]),
                    (blk.attn_skip_gating, o.skip[l][0]), (blk.mlp_skip_gating, o.skip[l][1])]
        out += [(self.norm.weight, o.norm_w), (self.norm.bias, o.norm_b), (self.head.weight, o.head_w), (self.head.bias, o.head_b),
                (self.block_skip_gating, o.gate)]
        if self.gumbel is not None:
            out += [(self.gumbel.weight, o.gumbel_w), (self.gumbel.bias, o.gumbel_b)]
        return out + self._front_slots()

    def _flatten(self, device):
        pads = [(p, p.data.detach().clone()) for p, *_ in self._padded()]
        super()._flatten(device)
        for (p, off, rows, dim, dimp), (_, val) in zip(self._padded(), pads):
            view = self._flat[off:off + rows * dimp].view(rows, dimp)
            view[:, :dim].copy_(val.to(device=device, dtype=torch.float32))
            p.data = view[:, :dim]
            p.grad = None
        self._front_shadow = None
        self._front_bufs = {}

    def grad_views(self, patch_mode2=False):
        super().grad_views(patch_mode2)
        self.pos_embed.grad = None                                  # requires_grad False (t2t_vit.py:119)
        for name in ("attention1", "attention2"):
            getattr(self.tokens_to_token, name).w.grad = None
        for p, off, rows, dim, dimp in self._padded():
            g = self._flat_grad[off:off + rows * dimp].view(rows, dimp)[:, :dim]
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def no_weight_decay(self):
        return {"cls_token"}                                        # t2t_vit.py:153-155

    # -- tokens-to-token module --------------------------------------------------------------------------
    def _dt(self):
        return ops.UVC_F32 if self.precision == "fp32" else ops.UVC_BF16

    def _tdt(self):
        return torch.float32 if self.precision == "fp32" else torch.bfloat16

    def _fp(self, off, *shape):
        n = int(np.prod(shape))
        return self._flat[off:off + n].view(*shape)

    def _fg(self, off, *shape):
        n = int(np.prod(shape))
        return self._flat_grad[off:off + n].view(*shape)

    def _front_weights(self):
        """T-typed copies W and W^T of the front end's Linear weights (refreshed with the engine's shadows)."""
        if self._front_shadow is None:
            dev, tdt = self._flat.device, self._tdt()
            sh = {}
            for name in ("attention1", "attention2"):
                dimp = self._front[name]["dimp"]
                for key, (r, c) in (("kqv", (192, dimp)), ("proj", (64, 64)), ("fc1", (64, 64)), ("fc2", (64, 64))):
                    sh[name + "." + key] = (torch.empty(r, c, device=dev, dtype=tdt), torch.empty(c, r, device=dev, dtype=tdt))
            D = self.embed_dim
            sh["project"] = (torch.empty(D, 576, device=dev, dtype=tdt), torch.empty(576, D, device=dev, dtype=tdt))
            self._front_shadow = sh
        return self._front_shadow

    def _refresh_front_shadows(self):
        sh, dt = self._front_weights(), self._dt()
        for name in ("attention1", "attention2"):
            f = self._front[name]
            for key, (r, c) in (("kqv", (192, f["dimp"])), ("proj", (64, 64)), ("fc1", (64, 64)), ("fc2", (64, 64))):
                w, wt = sh[name + "." + key]
                ops.cast_transpose(self._fp(f[key + "_w"], r, c), r, c, w, wt, dt)
        w, wt = sh["project"]
        ops.cast_transpose(self._fp(self._front["project_w"], self.embed_dim, 576), self.embed_dim, 576, w, wt, dt)

    def _front_buffers(self, B, training):
        key = (B, bool(training))
        if key in self._front_bufs:
            return self._front_bufs[key]
        dev, tdt = self._flat.device, self._tdt()
        S = self._cfg.img_size
        side = [S // 4, S // 8, S // 16]
        bufs = dict(side=side)
        tn_bytes = 0
        for i, name in enumerate(("attention1", "attention2")):
            f = self._front[name]
            T = side[i] * side[i]
            M = B * T
            sp = ops.performer_splits(B, T)
            b = dict(T=T, M=M, xn=torch.empty(M, f["dimp"], device=dev, dtype=tdt), mean1=torch.empty(M, device=dev), rstd1=torch.empty(M, device=dev),
                     kqv=torch.empty(M, 192, device=dev), part=torch.empty(B * sp * 65 * 32, device=dev), kptv=torch.empty(B, 65, 32, device=dev),
                     att=torch.empty(M, 64, device=dev, dtype=tdt), x1=torch.empty(M, 64, device=dev), h=torch.empty(M, 64, device=dev, dtype=tdt),
                     mean2=torch.empty(M, device=dev), rstd2=torch.empty(M, device=dev), u=torch.empty(M, 64, device=dev, dtype=tdt),
                     out=torch.empty(M, 64, device=dev))
            if training:
                # gradient streams in the operand type T (bf16 in the throughput mode, like the engine's dL/dx streams); sums are float32
                b.update(gp=torch.empty(M, 64, device=dev, dtype=tdt), dout=torch.empty(M, 64, device=dev, dtype=tdt), da=torch.empty(M, 64, device=dev, dtype=tdt),
                         dh=torch.empty(M, 64, device=dev, dtype=tdt), dx1=torch.empty(M, 64, device=dev, dtype=tdt), datt=torch.empty(M, 64, device=dev, dtype=tdt),
                         dkqv=torch.empty(M, 192, device=dev, dtype=tdt), dkptv=torch.empty(B, 65, 32, device=dev), dxn=torch.empty(M, f["dimp"], device=dev, dtype=tdt),
                         ln2_partial=torch.empty(ops.layernorm_bwd_blocks(M) * (2 * 64 + 2), device=dev),
                         ln1_partial=torch.empty(ops.unfold_bwd_blocks(M) * 2 * f["dim"], device=dev),
                         dxu=torch.empty(M, f["dim"], device=dev) if i == 1 else None)
                for (m_, n1, n2) in ((M, 64, 64), (M, 192, f["dimp"])):
                    tn_bytes = max(tn_bytes, ops.gemm_tn_workspace_bytes(m_, n1, n2))
            bufs[name] = b
        M3 = B * side[2] * side[2]
        bufs["tok_u"] = torch.empty(M3, 576, device=dev, dtype=tdt)
        if training:
            bufs["dxu3"] = torch.empty(M3, 576, device=dev, dtype=tdt)
            tn_bytes = max(tn_bytes, ops.gemm_tn_workspace_bytes(M3, self.embed_dim, 576))
            bufs["tn_ws"] = torch.empty(tn_bytes, device=dev, dtype=torch.uint8)
            bufs["cs_partial"] = torch.empty(ops.colsum_blocks(M3) * self.embed_dim, device=dev)
        self._front_bufs = {k: v for k, v in self._front_bufs.items() if k[1] != bool(training)}
        self._front_bufs[key] = bufs
        return bufs

    def _stage_struct(self, name, src, strides, B, C_, H, W, k, s, p, b, bufs, training, need_dx):
        """uvc_t2t_stage of one Token_performer stage (the whole stage is ONE call across the boundary, include/uvc_t2t.h): pointers into the flat
        parameter / gradient buffers, the weight shadows and the stage's buffers -- all of them live as long as `b` does."""
        key = "_stage_struct"
        st = b.get(key)
        f, sh = self._front[name], self._front_weights()
        fp, fg = self._fp, self._fg
        # the struct caches raw device pointers: it is valid as long as the tensors behind them are the same allocations (and the precision /
        # mode the same) -- checked by address instead of trusting that every re-allocation also drops the cache
        w0 = sh[name + ".kqv"][0]
        sig = (self._flat.data_ptr(), self._flat_grad.data_ptr() if training and self._flat_grad is not None else 0, w0.data_ptr(),
               bufs["tn_ws"].data_ptr() if training and bufs is not None and "tn_ws" in bufs else 0, self._dt(), int(training))
        if st is not None and b.get("_stage_sig") != sig:
            st = None
        b["_stage_sig"] = sig
        if st is None:
            st = L.uvc_t2t_stage()
            st.B, st.C, st.H, st.W, st.k, st.s, st.p = B, C_, H, W, k, s, p
            st.T, st.dim, st.dimp, st.dtype, st.training, st.eps = b["T"], f["dim"], f["dimp"], self._dt(), int(training), LN_EPS
            for n_, t_ in (("norm1_w", fp(f["norm1_w"], f["dim"])), ("norm1_b", fp(f["norm1_b"], f["dim"])), ("kqv_b", fp(f["kqv_b"], 192)), ("w", fp(f["w"], 32, 64)),
                           ("proj_b", fp(f["proj_b"], 64)), ("norm2_w", fp(f["norm2_w"], 64)), ("norm2_b", fp(f["norm2_b"], 64)), ("fc1_b", fp(f["fc1_b"], 64)),
                           ("fc2_b", fp(f["fc2_b"], 64))):
                setattr(st, n_, L.ptr(t_))
            for key_ in ("kqv", "proj", "fc1", "fc2"):
                w_, wt_ = sh[name + "." + key_]
                setattr(st, key_ + "_w", L.ptr(w_))
                setattr(st, key_ + "_wt", L.ptr(wt_))
            for n_ in ("xn", "mean1", "rstd1", "kqv", "part", "kptv", "att", "x1", "h", "mean2", "rstd2", "u", "out"):
                setattr(st, n_, L.ptr(b[n_]))
            if training:
                for n_, t_ in (("g_norm1_w", fg(f["norm1_w"], f["dim"])), ("g_norm1_b", fg(f["norm1_b"], f["dim"])), ("g_kqv_w", fg(f["kqv_w"], 192, f["dimp"])),
                               ("g_kqv_b", fg(f["kqv_b"], 192)), ("g_proj_w", fg(f["proj_w"], 64, 64)), ("g_proj_b", fg(f["proj_b"], 64)), ("g_norm2_w", fg(f["norm2_w"], 64)),
                               ("g_norm2_b", fg(f["norm2_b"], 64)), ("g_fc1_w", fg(f["fc1_w"], 64, 64)), ("g_fc1_b", fg(f["fc1_b"], 64)), ("g_fc2_w", fg(f["fc2_w"], 64, 64)),
                               ("g_fc2_b", fg(f["fc2_b"], 64))):
                    setattr(st, n_, L.ptr(t_))
                for n_ in ("gp", "dout", "da", "dh", "dx1", "datt", "dkqv", "dkptv", "dxn", "ln2_partial", "ln1_partial"):
                    setattr(st, n_, L.ptr(b[n_]))
                st.dxu = L.ptr(b["dxu"]) if b["dxu"] is not None else None
                st.tn_ws, st.tn_ws_bytes = L.ptr(bufs["tn_ws"]), bufs["tn_ws"].numel()
            b[key] = st
        st.src = L.ptr(src)                      # the image of this batch / the previous stage's output
        st.sb, st.sc, st.sh, st.sw = strides
        st.beta = 1.0 if self.grad_accumulate else 0.0
        st.need_dx = int(bool(need_dx))
        return st

    def _performer_forward(self, name, src, strides, B, C_, H, W, k, s, p, b, training, bufs=None):
        """One Token_performer stage (token_performer.py:45-69) on the soft split of `src`; result in b['out'] [B*T, 64]."""
        if self.front_in_c:
            st = self._stage_struct(name, src, strides, B, C_, H, W, k, s, p, b, bufs, training, False)
            L.check(L.lib().uvc_t2t_stage_forward(C.byref(st), L.cur_stream()), "uvc_t2t_stage_forward")
            return
        f, dt, sh = self._front[name], self._dt(), self._front_weights()
        M, T = b["M"], b["T"]
        fp = self._fp
        ops.unfold_ln_fwd(src, strides, B, C_, H, W, k, s, p, b["xn"], dt, gamma=fp(f["norm1_w"], f["dim"]), beta=fp(f["norm1_b"], f["dim"]),
                          mean=b["mean1"], rstd=b["rstd1"], eps=LN_EPS)
        ops.gemm_nt(b["xn"], sh[name + ".kqv"][0], b["kqv"], dtype=dt, epilogue=ops.EPI_BIAS, bias=fp(f["kqv_b"], 192))
        ops.performer_fwd(b["kqv"], fp(f["w"], 32, 64), b["part"], b["kptv"], b["att"], B, T, dt)
        # y = v + proj(att): v is columns 128..191 of kqv
        ops.gemm_nt(b["att"], sh[name + ".proj"][0], b["x1"], dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=fp(f["proj_b"], 64), R=b["kqv"].view(-1)[128:], ldr=192)
        ops.layernorm_fwd(b["x1"], fp(f["norm2_w"], 64), fp(f["norm2_b"], 64), b["h"], b["mean2"], b["rstd2"], M, 64, dt, eps=LN_EPS)
        if training:
            ops.gemm_nt(b["h"], sh[name + ".fc1"][0], b["gp"], dtype=dt, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=fp(f["fc1_b"], 64), C2=b["u"])   # C = GELU'(a), C2 = GELU(a)
        else:
            ops.gemm_nt(b["h"], sh[name + ".fc1"][0], b["u"], dtype=dt, epilogue=ops.EPI_BIAS_GELU_OUT, bias=fp(f["fc1_b"], 64))
        ops.gemm_nt(b["u"], sh[name + ".fc2"][0], b["out"], dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=fp(f["fc2_b"], 64), R=b["x1"])

    def _front_end_forward(self, x, B, training):
        """T2T_module.forward (t2t_vit.py:84-105): writes the [B*P, D] token embedding into the engine's workspace."""
        if not (self.frozen_weights and self._front_fresh):
            self._refresh_front_shadows()
            self._front_fresh = True
        bufs = self._front_buffers(B, training)
        S = self._cfg.img_size
        s1, s2, s3 = bufs["side"]
        dt = self._dt()
        b1, b2 = bufs["attention1"], bufs["attention2"]
        self._performer_forward("attention1", x, (3 * S * S, S * S, S, 1), B, 3, S, S, 7, 4, 2, b1, training, bufs)
        tok = lambda side: (side * side * 64, 1, side * 64, 64)          # token-major [B, side*side, 64] read as [B, 64, side, side]
        self._performer_forward("attention2", b1["out"], tok(s1), B, 64, s1, s1, 3, 2, 1, b2, training, bufs)
        ops.unfold_ln_fwd(b2["out"], tok(s2), B, 64, s2, s2, 3, 2, 1, bufs["tok_u"], dt)
        pe = self._ws_view(B, training, "pe")
        ops.gemm_nt(bufs["tok_u"], self._front_weights()["project"][0], pe, dtype=dt, epilogue=ops.EPI_BIAS, bias=self._fp(self._front["project_b"], self.embed_dim))
        self._front_state = dict(B=B, x=x) if training else None
        return True

    def _performer_backward(self, name, src, strides, B, C_, H, W, k, s, p, b, bufs, need_dx):
        """Backward of one stage from b['dout'] [B*T, 64]; leaves d(unfolded input) in b['dxu'] when the source needs it."""
        if self.front_in_c:
            st = self._stage_struct(name, src, strides, B, C_, H, W, k, s, p, b, bufs, True, need_dx)
            L.check(L.lib().uvc_t2t_stage_backward(C.byref(st), L.cur_stream()), "uvc_t2t_stage_backward")
            return
        f, dt, sh = self._front[name], self._dt(), self._front_weights()
        M, T = b["M"], b["T"]
        fp, fg = self._fp, self._fg
        beta = 1.0 if self.grad_accumulate else 0.0
        ws = bufs["tn_ws"]
        dout = b["dout"]
        ops.gemm_tn(dout, b["u"], fg(f["fc2_w"], 64, 64), ws, dtype=dt, beta=beta, colsum_out=fg(f["fc2_b"], 64))
        ops.gemm_nt(dout, sh[name + ".fc2"][1], b["da"], dtype=dt, epilogue=ops.EPI_MUL_AUX, aux=b["gp"])
        ops.gemm_tn(b["da"], b["h"], fg(f["fc1_w"], 64, 64), ws, dtype=dt, beta=beta, colsum_out=fg(f["fc1_b"], 64))
        ops.gemm_nt(b["da"], sh[name + ".fc1"][1], b["dh"], dtype=dt)
        ops.layernorm_bwd(b["dh"], b["x1"], fp(f["norm2_w"], 64), b["mean2"], b["rstd2"], b["dx1"], b["ln2_partial"], fg(f["norm2_w"], 64), fg(f["norm2_b"], 64),
                          M, 64, dt, add1=dout, beta_acc=beta, eps=LN_EPS)
        ops.gemm_tn(b["dx1"], b["att"], fg(f["proj_w"], 64, 64), ws, dtype=dt, beta=beta, colsum_out=fg(f["proj_b"], 64))
        ops.gemm_nt(b["dx1"], sh[name + ".proj"][1], b["datt"], dtype=dt)
        ops.performer_bwd(b["kqv"], fp(f["w"], 32, 64), b["part"], b["kptv"], b["datt"], b["dkqv"], b["dkptv"], B, T, dt, dskip=b["dx1"])
        ops.gemm_tn(b["dkqv"], b["xn"], fg(f["kqv_w"], 192, f["dimp"]), ws, dtype=dt, beta=beta, colsum_out=fg(f["kqv_b"], 192))
        ops.gemm_nt(b["dkqv"], sh[name + ".kqv"][1], b["dxn"], dtype=dt)
        ops.unfold_ln_bwd(src, strides, B, C_, H, W, k, s, p, b["dxn"], dt, gamma=fp(f["norm1_w"], f["dim"]), mean=b["mean1"], rstd=b["rstd1"],
                          partial=b["ln1_partial"], dgamma=fg(f["norm1_w"], f["dim"]), dbeta=fg(f["norm1_b"], f["dim"]), dxu=b["dxu"] if need_dx else None,
                          beta_acc=beta, eps=LN_EPS, dxu_tap_major=need_dx)     # [rows][9 taps][64]: whole channel rows for the fold below

    def _front_end_backward(self, st):
        fs = st["front"]
        B, x = fs["B"], fs["x"]
        bufs = self._front_buffers(B, True)
        S, D, dt = self._cfg.img_size, self.embed_dim, self._dt()
        s1, s2, s3 = bufs["side"]
        b1, b2 = bufs["attention1"], bufs["attention2"]
        beta = 1.0 if self.grad_accumulate else 0.0
        # pos_embed is not trainable (t2t_vit.py:119): the engine's token-assembly backward wrote its gradient, drop it
        self._flat_grad[self._off.pos_embed:self._off.pos_embed + self.pos_embed.numel()].zero_()
        dpe = self._ws_view(B, True, "dpe")
        ops.gemm_tn(dpe, bufs["tok_u"], self._fg(self._front["project_w"], D, 576), bufs["tn_ws"], dtype=dt, beta=beta,
                    colsum_out=self._fg(self._front["project_b"], D))
        ops.gemm_nt(dpe, self._front_weights()["project"][1], bufs["dxu3"], dtype=dt)
        ops.fold_tokens(bufs["dxu3"], b2["dout"], B, 64, s2, s2, 3, 2, 1, dt)
        tok = lambda side: (side * side * 64, 1, side * 64, 64)
        self._performer_backward("attention2", b1["out"], tok(s1), B, 64, s1, s1, 3, 2, 1, b2, bufs, True)
        ops.fold_tokens(b2["dxu"], b1["dout"], B, 64, s1, s1, 3, 2, 1, dt, tap_major=True)
        self._performer_backward("attention1", x, (3 * S * S, S * S, S, 1), B, 3, S, S, 7, 4, 2, b1, bufs, False)

    def mark_weights_changed(self):
        super().mark_weights_changed()
        self._front_fresh = False

    _front_fresh = False

    # -- reference API ----------------------------------------------------------------------------------
    def macs(self, B):
        """MAC bookkeeping of the reference forward (token_performer.py:54-69, transformer_block.py:27-75)."""
        c = self._cfg
        S = c.img_size
        N, D, Fh, H = (S // 16) ** 2 + 1, c.embed_dim, c.hidden, c.num_heads
        embed = 0
        for T, dim in (((S // 4) ** 2, 147), ((S // 8) ** 2, 576)):
            emb, m = 64, 32
            embed += B * (T * dim * 3 * emb + 2 * (T * emb + emb * T * emb) + T * m + T * emb * m + T * m * emb + T * emb * emb)
            embed += B * (T * emb * emb + emb * emb * emb)
        blk = [B * 3 * D * N * D, N * B * H * N * 64, N * B * H * N * 64, B * N * D * D, Fh * B * N * D, D * B * N * Fh]
        return embed, [list(blk) for _ in range(c.depth)]

    def forward(self, x, tau=-1, number=0.9):
        """``tau`` / ``number`` are what joint_train.py:410,1012 pass.  With ``enable_patch_gating=2`` and ``tau > 0`` the tokens are
        gated as the module docstring defines; otherwise both are ignored, as the reference's forward_features ignores them."""
        if self.enable_jumping:
            raise NotImplementedError("enable_jumping is off on the UVC hot path")
        macs = self.macs(x.shape[0])
        if self.gumbel is None:
            tau = -1
        if not self.enable_block_gating:
            run = self._hard_run_blocks()
            macs = (macs[0], [m if run[l] else [] for l, m in enumerate(macs[1])])
        if self.training and torch.is_grad_enabled():
            from .model_distilled import _VitFunction
            out = _VitFunction.apply(self, x, self.cls_token, tau, number)
            return (out, out), macs                                  # t2t_vit.py:205-206
        o, _ = self._run_forward(x, tau, number, training=False)
        if self.training:
            return (o, o), macs
        return o, macs


def t2t_vit_14(pretrained=False, **kwargs):
    """t2t_vit.py:244-249."""
    if pretrained:
        raise NotImplementedError("no network in this image: load a state_dict instead")
    return T2T_ViT(tokens_type="performer", embed_dim=384, depth=14, num_heads=6, mlp_ratio=3., **kwargs)
