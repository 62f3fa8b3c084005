"""Oracle: T2T-ViT (tokens-to-token front end + plain transformer blocks), restated functionally on a state dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows the reference's
``UVC/T2TViT/models/t2t_vit.py`` (T2T_module.forward :84-105, T2T_ViT.forward_features :168-200, forward :202-208),
``token_performer.py`` (prm_exp :31-43, single_attn :45-62, forward :64-69) and ``transformer_block.py``
(Mlp :25-40, Attention :54-77, Block :100-112, get_sinusoid_encoding :115-125); backward is torch autograd on CPU.

Pinning: the UNGATED forward (token-to-token module output, logits, the MAC table) is pinned to the reference's own
modules run in the build container (tests/golden/make_t2t_golden.py -> tests/golden/t2t_*.npz), and its backward by the
Stage-2 step fixture (make_t2t_stage2_golden.py -> t2t_stage2_micro.npz: the reference's autograd, Performer dropout p = 0).  The reference's GATED
T2T forward raises as shipped (``F`` never imported, ``self.gumbel_hard`` never assigned), so the gated
branch below restates the lines as written (:181-189, identical to model_distilled.py:480-500) and its parity is UNPINNED.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class T2TConfig:
    """t2t_vit_14 (t2t_vit.py:244-249): embed 384, depth 14, heads 6, mlp_ratio 3, token_dim 64."""
    img_size: int = 224
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 384
    depth: int = 14
    num_heads: int = 6
    mlp_ratio: float = 3.0
    token_dim: int = 64
    kernel_ratio: float = 0.5

    @property
    def num_patches(self) -> int:
        return (self.img_size // 16) ** 2                      # t2t_vit.py:82

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1

    @property
    def hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def m(self) -> int:
        return int(self.token_dim * self.kernel_ratio)         # token_performer.py:27


LN_EPS = 1e-5          # nn.LayerNorm default (t2t_vit.py:111, token_performer.py:16-17)
PRM_EPS = 1e-8         # token_performer.py:18


def sinusoid_encoding(n_position: int, d_hid: int) -> torch.Tensor:
    """transformer_block.py:115-125: float64 table, sin on even / cos on odd columns, cast to float32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)
    table = pos / np.power(10000.0, 2 * (j // 2) / d_hid)[None, :]
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.from_numpy(table.astype(np.float32)).unsqueeze(0)


def param_shapes(cfg: T2TConfig, enable_patch_gating: int = 0) -> Dict[str, tuple]:
    """``enable_patch_gating == 2`` adds the token scorer ``gumbel = Linear(D -> 1)`` that DeiT's patch gating uses
    (model_distilled.py:419) -- T2T_ViT has no such module in the reference; see ``forward``."""
    td, D, Fh, m = cfg.token_dim, cfg.embed_dim, cfg.hidden, cfg.m
    s: Dict[str, tuple] = {}
    for name, dim in (("attention1", cfg.in_chans * 49), ("attention2", td * 9)):
        p = f"tokens_to_token.{name}."
        s[p + "w"] = (m, td)
        s[p + "kqv.weight"] = (3 * td, dim); s[p + "kqv.bias"] = (3 * td,)
        s[p + "proj.weight"] = (td, td); s[p + "proj.bias"] = (td,)
        s[p + "norm1.weight"] = (dim,); s[p + "norm1.bias"] = (dim,)
        s[p + "norm2.weight"] = (td,); s[p + "norm2.bias"] = (td,)
        s[p + "mlp.0.weight"] = (td, td); s[p + "mlp.0.bias"] = (td,)
        s[p + "mlp.2.weight"] = (td, td); s[p + "mlp.2.bias"] = (td,)
    s["tokens_to_token.project.weight"] = (D, td * 9); s["tokens_to_token.project.bias"] = (D,)
    s["cls_token"] = (1, 1, D)
    s["pos_embed"] = (1, cfg.seq_len, D)
    s["block_skip_gating"] = (cfg.depth, 2)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s[b + "attn_skip_gating"] = (2,); s[b + "mlp_skip_gating"] = (2,)
        s[b + "norm1.weight"] = (D,); s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)                                   # qkv_bias=False (t2t_vit.py:109)
        s[b + "attn.proj.weight"] = (D, D); s[b + "attn.proj.bias"] = (D,)
        s[b + "norm2.weight"] = (D,); s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (Fh, D); s[b + "mlp.fc1.bias"] = (Fh,)
        s[b + "mlp.fc2.weight"] = (D, Fh); s[b + "mlp.fc2.bias"] = (D,)
    s["norm.weight"] = (D,); s["norm.bias"] = (D,)
    s["head.weight"] = (cfg.num_classes, D); s["head.bias"] = (cfg.num_classes,)
    if enable_patch_gating == 2:                        # appended last: the other tensors keep their RandomState draws
        s["gumbel.weight"] = (1, D); s["gumbel.bias"] = (1,)
    return s


def init_params_numpy(cfg: T2TConfig, seed: int, std: float = 0.02, weight_gain: float = 1.0,
                      enable_patch_gating: int = 0) -> Dict[str, torch.Tensor]:
    """Portable deterministic weights (numpy's frozen RandomState stream): clipped normal std .02 * gain for Linear
    weights, small normal biases, LN 1 +- .1, gates [-1, 1], the sinusoid table for pos_embed; the Performer's
    random-feature matrix ``w`` is a plain normal scaled to the row norm sqrt(m) of the reference's
    orthogonal_ * sqrt(m) init (token_performer.py:28-29; values differ, the forward only reads them)."""
    rs = np.random.RandomState(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in param_shapes(cfg, enable_patch_gating).items():
        if name == "block_skip_gating":
            a = np.tile(np.array([-1.0, 1.0], dtype=np.float32), (cfg.depth, 1))
        elif name.endswith("skip_gating"):
            a = np.array([-1.0, 1.0], dtype=np.float32)
        elif name == "pos_embed":
            out[name] = sinusoid_encoding(cfg.seq_len, cfg.embed_dim)
            continue
        elif name.endswith(".w"):
            a = (rs.standard_normal(shp) * math.sqrt(cfg.m / cfg.token_dim)).astype(np.float32)
        elif "norm" in name and name.endswith("weight"):
            a = (1.0 + 0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif name.endswith("bias"):
            a = (0.02 * rs.standard_normal(shp)).astype(np.float32)
        else:
            g = weight_gain if name.endswith("weight") else 1.0
            a = np.clip(rs.standard_normal(shp), -2.0, 2.0).astype(np.float32) * np.float32(std * g)
        out[name] = torch.from_numpy(np.ascontiguousarray(a.reshape(shp)))
    return out


# ---- tokens-to-token ------------------------------------------------------------------------------------
def soft_split(x: torch.Tensor, k: int, s: int, p: int) -> torch.Tensor:
    """nn.Unfold(k, stride s, padding p)(x).transpose(1, 2): [B, C, H, W] -> [B, L, C*k*k], feature index c*k*k + ki*k + kj
    (t2t_vit.py:86,93,100)."""
    return F.unfold(x, (k, k), stride=(s, s), padding=(p, p)).transpose(1, 2)


def prm_exp(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """token_performer.py:31-43: exp(w x - |x|^2 / 2) / sqrt(m)."""
    m = w.shape[0]
    xd = (x * x).sum(dim=-1, keepdim=True) / 2
    wtx = torch.einsum("bti,mi->btm", x.float(), w)
    return torch.exp(wtx - xd) / math.sqrt(m)


def linear_attention(kqv: torch.Tensor, w: torch.Tensor):
    """token_performer.py:46-50: kqv [B, T, 3*emb] split as k | q | v -> (y [B, T, emb], v)."""
    emb = kqv.shape[-1] // 3
    k, q, v = torch.split(kqv, emb, dim=-1)
    kp, qp = prm_exp(k, w), prm_exp(q, w)
    D = torch.einsum("bti,bi->bt", qp, kp.sum(dim=1)).unsqueeze(2)
    kptv = torch.einsum("bin,bim->bnm", v.float(), kp)
    y = torch.einsum("bti,bni->btn", qp, kptv) / (D + PRM_EPS)
    return y, v


def performer(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor):
    """Token_performer.forward (token_performer.py:45-69).  Returns (tokens [B, T, emb], macs)."""
    emb = sd[pre + "proj.weight"].shape[0]
    w = sd[pre + "w"]
    xn = F.layer_norm(x, (x.shape[-1],), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    kqv = F.linear(xn, sd[pre + "kqv.weight"], sd[pre + "kqv.bias"])
    y, v = linear_attention(kqv, w)
    y = v + F.linear(y, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    B, T, dim = x.shape
    m = w.shape[0]
    attn_macs = B * (T * dim * 3 * emb + 2 * (T * emb + emb * T * emb) + T * m + T * emb * m + T * m * emb + T * emb * emb)
    h = F.layer_norm(y, (emb,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)
    h = F.linear(F.gelu(F.linear(h, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])), sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"])
    out = y + h
    mlp_macs = B * (T * emb * emb + emb * emb * emb)            # as written at token_performer.py:68
    return out, attn_macs + mlp_macs


def t2t_module(sd: Dict[str, torch.Tensor], x: torch.Tensor, taps: Optional[dict] = None):
    """T2T_module.forward (t2t_vit.py:84-105): [B, 3, S, S] -> ([B, (S/16)^2, D], macs1 + macs2)."""
    p = "tokens_to_token."
    x = soft_split(x, 7, 4, 2)
    x, macs1 = performer(sd, p + "attention1.", x)
    if taps is not None:
        taps["attention1"] = x
    B, T, C = x.shape
    side = int(np.sqrt(T))
    x = soft_split(x.transpose(1, 2).reshape(B, C, side, side), 3, 2, 1)
    x, macs2 = performer(sd, p + "attention2.", x)
    if taps is not None:
        taps["attention2"] = x
    B, T, C = x.shape
    side = int(np.sqrt(T))
    x = soft_split(x.transpose(1, 2).reshape(B, C, side, side), 3, 2, 1)
    x = F.linear(x, sd[p + "project.weight"], sd[p + "project.bias"])
    return x, macs1 + macs2


def block(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, num_heads: int):
    """Block.forward with Attention / Mlp (transformer_block.py:54-77,25-40,100-112).  Returns (x, 6 MAC entries)."""
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    qkv = F.linear(h, sd[pre + "attn.qkv.weight"], sd.get(pre + "attn.qkv.bias"))
    qkv = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (C // num_heads) ** -0.5).softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    o = F.linear(o, sd[pre + "attn.proj.weight"], sd[pre + "attn.proj.bias"])
    macs = [B * 3 * C * N * C, N * B * num_heads * N * (C // num_heads), N * B * num_heads * N * (C // num_heads), B * N * C * C]
    x = x + o
    h = F.layer_norm(x, (C,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)
    Fh = sd[pre + "mlp.fc1.weight"].shape[0]
    h = F.linear(F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    macs += [Fh * B * N * C, C * B * N * Fh]
    return x + h, macs


def forward(sd: Dict[str, torch.Tensor], cfg: T2TConfig, x: torch.Tensor, gate_d: Optional[torch.Tensor] = None,
            taps: Optional[dict] = None, patch: Optional[dict] = None):
    """T2T_ViT.forward (t2t_vit.py:168-208).  ``gate_d`` [L, 2] = the per-block distributions when block gating is
    enabled (:181-189: x = d1 * blk(x) + d0 * x); None = the hard skip on the gate logits (:192-194).
    ``patch`` = dict(tau, ratio, e [B, P] Exp(1) draws[, record]): Gumbel top-k patch gating on the tokens the
    tokens-to-token module produced.  The reference's T2T forward_features has NO such step although BASELINE config 5 /
    joint_train.py:404-410 call the model with (tau, ratio); it is DEFINED here by transplanting
    model_distilled.py:446-456 verbatim: scorer Linear(D -> 1), log_softmax, Gumbel top-k with k = int(ratio * P),
    straight-through mask, mask[:, 0] = 1, tokens multiplied (not removed) before cls / pos are added.  UNPINNED.
    Returns (logits, (macs_embed, macs_list))."""
    B = x.shape[0]
    tok, macs_embed = t2t_module(sd, x, taps)
    if patch is not None:
        from . import vit as V
        k = int(patch["ratio"] * tok.shape[1])
        scores = F.linear(tok, sd["gumbel.weight"], sd["gumbel.bias"]).reshape(B, -1)
        mask, index = V.patch_topk_mask(scores, patch["e"], k, patch["tau"])
        if patch.get("record") is not None:
            patch["record"]["patch_index"] = index
            patch["record"]["patch_mask"] = mask.detach()
        tok = tok * mask.unsqueeze(-1)
    if taps is not None:
        taps["tokens"] = tok
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), tok), dim=1) + sd["pos_embed"]
    macs_list: List[list] = []
    g = sd["block_skip_gating"]
    for i in range(cfg.depth):
        pre = f"blocks.{i}."
        if gate_d is not None:
            t, macs = block(sd, pre, x, cfg.num_heads)
            x = gate_d[i, 1] * t + gate_d[i, 0] * x
            macs_list.append(macs)
        else:
            macs = []
            if g[i, 1] > g[i, 0]:
                x, macs = block(sd, pre, x, cfg.num_heads)
            macs_list.append(macs)
    x = F.layer_norm(x, (cfg.embed_dim,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    logits = F.linear(x[:, 0], sd["head.weight"], sd["head.bias"])
    return logits, (macs_embed, macs_list)


def forward_flags(params: Dict[str, torch.Tensor], cfg: T2TConfig, flags, x: torch.Tensor, tau: float = -1.0, ratio: float = 0.9,
                  exp_draws: Optional[list] = None, record: Optional[dict] = None):
    """``forward`` behind the call signature of oracle/vit.py:forward (what oracle/step.py:stage1_step drives): the per-block
    gate distributions come from oracle/vit.py:block_distrib (model_distilled.py:480-488 == t2t_vit.py:181-185) with one
    Exp(1) draw [2] per block; ``tau > 0`` switches on the patch gating defined in ``forward`` (its [B, P] draw is consumed
    first, DeiT's RNG order).  Training returns ((logits, logits), macs) (t2t_vit.py:205-206), eval (logits, macs).
    UNPINNED for the gated cases."""
    from . import vit as V
    gate_d = None
    draws = list(exp_draws) if exp_draws is not None else []
    patch = None
    if tau > 0:
        patch = dict(tau=float(tau), ratio=float(ratio), e=draws.pop(0), record=record)
    if flags.enable_block_gating:
        rows = []
        for i in range(cfg.depth):
            e = draws.pop(0) if (flags.use_gumbel == 1 and not flags.enable_warmup) else None
            rows.append(V.block_distrib(params["block_skip_gating"][i], flags, e))
        gate_d = torch.stack(rows)
        if record is not None:
            record["distribs"] = gate_d.detach()
    logits, macs = forward(params, cfg, x, gate_d=gate_d, patch=patch)
    if flags.training:
        return (logits, logits), macs
    return logits, macs
