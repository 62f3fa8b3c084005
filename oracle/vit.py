"""Oracle: DeiT (DistilledVisionTransformer) forward, restated functionally on a state dict.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
``UVC/models/model_distilled.py`` of the reference; backward is torch autograd on CPU, as in
the reference.  All randomness is an explicit input: ``exp_draws`` holds the
``Tensor.exponential_`` samples in the order the reference consumes them
(SURVEY.md §8c note 3): [patch gating E[B,P]] then one E[2] per block.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VitConfig:
    """Dims of ``DistilledVisionTransformer`` (model_distilled.py:258-261,391)."""
    img_size: int = 224
    patch_size: int = 16
    in_chans: int = 3
    num_classes: int = 1000
    embed_dim: int = 192
    depth: int = 12
    num_heads: int = 3
    mlp_ratio: float = 4.0
    enable_dist: int = 0

    @property
    def num_patches(self) -> int:
        return (self.img_size // self.patch_size) ** 2

    @property
    def num_tokens(self) -> int:
        return 2 if self.enable_dist else 1

    @property
    def seq_len(self) -> int:
        return self.num_patches + self.num_tokens

    @property
    def hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


# models/configs.py:112-165 (dims only)
CONFIGS = {
    "deit_tiny_patch16_224": dict(embed_dim=192, depth=12, num_heads=3, mlp_ratio=4.0),
    "deit_small_patch16_224": dict(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0),
    "deit_base_patch16_224": dict(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0),
}


@dataclass
class GateFlags:
    """Mutable attributes the driver pokes on the model (uvc_optimizer.py:204-210,
    joint_train.py:348-360)."""
    enable_block_gating: int = 0
    enable_patch_gating: int = 0
    enable_jumping: int = 0
    use_gumbel: int = 0
    eps: float = 0.1
    enable_warmup: int = 0
    gumbel_hard: bool = True
    patch_hard: bool = False
    training: bool = True


def param_shapes(cfg: VitConfig, enable_patch_gating: int = 0) -> "Dict[str, Tuple[int, ...]]":
    """state_dict parameter keys and shapes, in the reference's registration order
    (model_distilled.py:272-306,393-417; SURVEY.md §5 checkpoint row)."""
    D, Fh, C = cfg.embed_dim, cfg.hidden, cfg.num_classes
    shapes: Dict[str, Tuple[int, ...]] = {}
    shapes["cls_token"] = (1, 1, D)
    shapes["pos_embed"] = (1, cfg.seq_len, D)
    if cfg.enable_dist:                 # registered after pos_embed (model_distilled.py:277 is None, :393 re-registers)
        shapes["dist_token"] = (1, 1, D)
    shapes["block_skip_gating"] = (cfg.depth, 2)
    if enable_patch_gating == 1:
        shapes["patch_gating"] = (1, cfg.num_patches, 1)
    shapes["patch_embed.proj.weight"] = (D, cfg.in_chans, cfg.patch_size, cfg.patch_size)
    shapes["patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        shapes[p + "attn_skip_gating"] = (2,)
        shapes[p + "mlp_skip_gating"] = (2,)
        shapes[p + "norm1.weight"] = (D,)
        shapes[p + "norm1.bias"] = (D,)
        shapes[p + "attn.qkv.weight"] = (3 * D, D)
        shapes[p + "attn.qkv.bias"] = (3 * D,)
        shapes[p + "attn.proj.weight"] = (D, D)
        shapes[p + "attn.proj.bias"] = (D,)
        shapes[p + "norm2.weight"] = (D,)
        shapes[p + "norm2.bias"] = (D,)
        shapes[p + "mlp.fc1.weight"] = (Fh, D)
        shapes[p + "mlp.fc1.bias"] = (Fh,)
        shapes[p + "mlp.fc2.weight"] = (D, Fh)
        shapes[p + "mlp.fc2.bias"] = (D,)
    shapes["norm.weight"] = (D,)
    shapes["norm.bias"] = (D,)
    shapes["head.weight"] = (C, D)
    shapes["head.bias"] = (C,)
    if cfg.enable_dist:
        shapes["head_dist.weight"] = (C, D)
        shapes["head_dist.bias"] = (C,)
    shapes["gumbel.weight"] = (1, D)
    shapes["gumbel.bias"] = (1,)
    return shapes


def gumbel_from_exp(e: torch.Tensor) -> torch.Tensor:
    """G = -log(E), E ~ Exp(1)  (torch F.gumbel_softmax; model_distilled.py:40)."""
    return -e.log()


def block_distrib(g_i: torch.Tensor, flags: GateFlags, e: Optional[torch.Tensor]) -> torch.Tensor:
    """Per-block gate 2-vector (model_distilled.py:480-488)."""
    if flags.enable_warmup:
        return torch.ones(2, dtype=g_i.dtype) * 0.5
    if flags.use_gumbel == 1:
        u = (g_i + gumbel_from_exp(e)) / 0.5
        y_soft = u.softmax(-1)
        if flags.gumbel_hard:
            idx = y_soft.argmax(-1)
            y_hard = torch.zeros_like(y_soft)
            y_hard[idx] = 1.0
            return y_hard - y_soft.detach() + y_soft
        return y_soft
    d1 = g_i[1] ** 2 / (g_i[1] ** 2 + flags.eps)
    return torch.stack([1 - d1, d1])


def patch_topk_mask(scores: torch.Tensor, e: torch.Tensor, k: int, tau: float):
    """Gumbel top-k token mask with straight-through (model_distilled.py:36-63,446-456).
    scores [B,P] are the raw Linear(D->1) outputs.  Returns (mask[B,P] with STE, hard index set)."""
    logits = F.log_softmax(scores, dim=-1)
    u = (logits + gumbel_from_exp(e)) / tau
    y_soft = u.softmax(-1)
    index = y_soft.topk(k, dim=-1)[1]
    y_hard = torch.zeros_like(y_soft)
    y_hard.scatter_(1, index, 1.0)
    ret = y_hard - y_soft.detach() + y_soft
    ret = ret.clone()
    ret[:, 0] = 1.0
    return ret, index


def mac_table(cfg: VitConfig, B: int = 1):
    """The reference's MAC bookkeeping (model_distilled.py:115,121,177,182,185,189,460)."""
    N, D, Fh, H = cfg.seq_len, cfg.embed_dim, cfg.hidden, cfg.num_heads
    hd = cfg.head_dim
    embed = B * cfg.num_patches * D * cfg.patch_size * cfg.patch_size * cfg.in_chans
    blk = [B * 3 * D * N * D,          # qkv
           N * B * H * N * hd,         # q k^T
           N * B * H * N * hd,         # p v
           B * N * D * D,              # proj
           Fh * B * N * D,             # fc1
           D * B * N * Fh]             # fc2
    return embed, [list(blk) for _ in range(cfg.depth)]


def forward(params: Dict[str, torch.Tensor], cfg: VitConfig, flags: GateFlags, x: torch.Tensor,
            tau: float = -1.0, ratio: float = 0.9,
            exp_draws: Optional[List[torch.Tensor]] = None, record: Optional[dict] = None):
    """model(x, tau, number) -> ((logits, logits_dist), (macs_embed, macs_list)) in train mode,
    ((x + x_dist)/2, macs) in eval mode (model_distilled.py:429-531)."""
    draws = list(exp_draws) if exp_draws is not None else []
    B = x.shape[0]
    D, H, hd, N = cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.seq_len
    scale = hd ** -0.5
    t = F.conv2d(x, params["patch_embed.proj.weight"], params["patch_embed.proj.bias"], stride=cfg.patch_size)
    t = t.flatten(2).transpose(1, 2)                                   # :151
    if flags.enable_patch_gating == 1:                                 # :434-444
        pg = torch.sigmoid(params["patch_gating"])
        if flags.patch_hard:
            m = (pg >= 0.5).to(t.dtype).clone()
            m[:, 0] = 1
            t = t * m
        else:
            t = t * pg
    if tau > 0:                                                        # :446-456
        k = int(ratio * t.shape[1])
        scores = F.linear(t, params["gumbel.weight"], params["gumbel.bias"]).reshape(B, -1)
        mask, index = patch_topk_mask(scores, draws.pop(0), k, tau)
        if record is not None:
            record["patch_index"] = index
            record["patch_mask"] = mask.detach()
        t = t * mask.unsqueeze(-1)
    toks = [params["cls_token"].expand(B, -1, -1)]
    if cfg.enable_dist:
        toks.append(params["dist_token"].expand(B, -1, -1))
    h = torch.cat(toks + [t], dim=1) + params["pos_embed"]            # :462-471
    distribs = []
    skipped = []
    accum = 0
    for i in range(cfg.depth):
        p = f"blocks.{i}."

        def blk(z):
            a = F.layer_norm(z, (D,), params[p + "norm1.weight"], params[p + "norm1.bias"], 1e-6)
            qkv = F.linear(a, params[p + "attn.qkv.weight"], params[p + "attn.qkv.bias"])
            qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
            q, k_, v = qkv[0], qkv[1], qkv[2]
            att = ((q @ k_.transpose(-2, -1)) * scale).softmax(dim=-1)
            o = (att @ v).transpose(1, 2).reshape(B, N, D)
            z = z + F.linear(o, params[p + "attn.proj.weight"], params[p + "attn.proj.bias"])
            m = F.layer_norm(z, (D,), params[p + "norm2.weight"], params[p + "norm2.bias"], 1e-6)
            m = F.gelu(F.linear(m, params[p + "mlp.fc1.weight"], params[p + "mlp.fc1.bias"]))
            return z + F.linear(m, params[p + "mlp.fc2.weight"], params[p + "mlp.fc2.bias"])

        if flags.enable_block_gating:                                  # :479-494
            need_draw = (not flags.enable_warmup) and flags.use_gumbel == 1
            d = block_distrib(params["block_skip_gating"][i], flags, draws.pop(0) if need_draw else None)
            distribs.append(d.detach().clone())
            h = d[1] * blk(h) + d[0] * h
        else:                                                          # :496-500
            g = params["block_skip_gating"][i]
            if g[1] > g[0]:
                h = blk(h)
            else:
                skipped.append(i)                                      # tmp_macs stays [] (:477,500)
        accum = accum + h
    if flags.enable_jumping:
        h = accum
    h = F.layer_norm(h, (D,), params["norm.weight"], params["norm.bias"], 1e-6)
    o = F.linear(h[:, 0], params["head.weight"], params["head.bias"])
    if cfg.enable_dist:
        od = F.linear(h[:, 1], params["head_dist.weight"], params["head_dist.bias"])
    else:
        od = o
    if record is not None:
        record["distribs"] = distribs
    macs = mac_table(cfg, B)
    if skipped:
        macs = (macs[0], [[] if i in skipped else m for i, m in enumerate(macs[1])])
    if flags.training:
        return (o, od), macs
    return (o + od) / 2, macs


def init_params_numpy(cfg: VitConfig, seed: int, enable_patch_gating: int = 0, std: float = 0.02,
                      weight_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Portable deterministic weights for fixtures/benchmarks: numpy's frozen legacy
    ``RandomState`` stream (identical on every machine, unlike torch.manual_seed inits).
    Distribution follows model_distilled.py:65-97,311-318,416 (normal std .02 clipped at 2 std,
    zero biases, LN 1/0, gates [-1,1]); conv gets the same normal instead of torch's
    kaiming-uniform default.  ``weight_gain`` scales Linear weights (used to make fixtures with
    non-degenerate attention)."""
    import numpy as np
    rs = np.random.RandomState(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shp in param_shapes(cfg, enable_patch_gating).items():
        if name == "block_skip_gating":
            a = np.tile(np.array([-1.0, 1.0], dtype=np.float32), (cfg.depth, 1))
        elif name.endswith("skip_gating"):
            a = np.array([-1.0, 1.0], dtype=np.float32)
        elif name == "patch_gating":
            a = np.full(shp, 3.0, dtype=np.float32)
        elif ("norm" in name and name.endswith("weight")):
            a = (1.0 + 0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif name.endswith("bias"):
            a = (0.02 * rs.standard_normal(shp)).astype(np.float32)
        else:
            g = weight_gain if name.endswith("weight") else 1.0
            a = np.clip(rs.standard_normal(shp), -2.0, 2.0).astype(np.float32) * np.float32(std * g)
        out[name] = torch.from_numpy(np.ascontiguousarray(a.reshape(shp)))
    return out
