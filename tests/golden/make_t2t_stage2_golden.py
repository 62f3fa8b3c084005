#!/usr/bin/env python3
"""Generate the T2T-ViT Stage-2 fixture(s) (t2t_stage2_*.npz) by running the REFERENCE's own T2T_ViT, DistillationLoss and
torch autograd / AdamW through the loop body of post_train.py:341-377 (restated, post_train.py itself cannot be imported).

Build container only (needs /root/reference).  Usage: python tests/golden/make_t2t_stage2_golden.py [scenario ...]

This is the pin for the BACKWARD of the tokens-to-token module (soft split, Performer linear attention, LayerNorms,
project): the gradients below come from the reference's modules.  Harness changes, all stated here:
  * the Performer's Dropout(0.1) layers (token_performer.py:13,24) are set to p = 0 -- they draw from the global RNG in
    train mode; the engine does not apply them (NOTEBOOK 10b);
  * block_skip_gating is replaced by a real [L, 2] parameter before loading (the reference's rows alias one storage);
  * timm's add_weight_decay / cosine epoch schedule are restated as in make_stage2_golden.py (parity for those unpinned).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
import scenarios as SC  # noqa: E402
import t2t_scenarios as TS  # noqa: E402
from oracle import t2t as OT  # noqa: E402  (portable weight recipe only)
from oracle import stage2 as O2  # noqa: E402  (timm restatements: decay groups + epoch lr)

ref_shim.install()
from T2TViT.models.t2t_vit import T2T_ViT  # noqa: E402
from utils.losses import DistillationLoss  # noqa: E402


class SoftTargetCrossEntropy(nn.Module):
    def forward(self, x, target):
        return torch.sum(-target * torch.nn.functional.log_softmax(x, dim=-1), dim=-1).mean()


def build_ref(cfg, sd):
    model = T2T_ViT(img_size=cfg.img_size, tokens_type="performer", num_classes=cfg.num_classes, embed_dim=cfg.embed_dim, depth=cfg.depth,
                    num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, token_dim=cfg.token_dim)
    model.block_skip_gating = nn.Parameter(torch.zeros(cfg.depth, 2))
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
    return model


def run(name):
    r = TS.stage2_recipe(name)
    cfg = OT.T2TConfig(**r["model_cfg"])
    L, H, hd, F = cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden
    params = OT.init_params_numpy(cfg, r["seed"], weight_gain=r["weight_gain"])
    tparams = OT.init_params_numpy(cfg, r["seed"] + 500, weight_gain=r["weight_gain"])
    keep_proj, keep_hidden, gate = SC.stage2_masks(r, L, H, hd, F)
    x_all, y_all = TS.stage1_inputs(r)
    model = build_ref(cfg, params)
    for _, p in model.named_modules():                                                              # post_train.py:155-157
        if hasattr(p, "weight"):
            p.register_buffer("mask", torch.ones_like(p.weight))
    state = {k: v.clone() for k, v in params.items()}
    state["block_skip_gating"] = torch.from_numpy(gate)
    for k, v in model.state_dict().items():
        if k.endswith(".mask"):
            state[k] = torch.ones_like(v)
    for l in range(L):
        state[f"blocks.{l}.attn.proj.mask"] = torch.from_numpy(keep_proj[l])[None, :].expand(cfg.embed_dim, -1).clone()
        state[f"blocks.{l}.mlp.fc2.mask"] = torch.from_numpy(keep_hidden[l])[None, :].expand(cfg.embed_dim, -1).clone()
        state[f"blocks.{l}.mlp.fc1.mask"] = torch.from_numpy(keep_hidden[l])[:, None].expand(-1, cfg.embed_dim).clone()
    model.load_state_dict(state)
    teacher = build_ref(cfg, tparams)
    teacher.load_state_dict(tparams, strict=True)
    teacher.eval()
    criterion = DistillationLoss(SoftTargetCrossEntropy(), teacher, r["distillation_type"], r["distillation_alpha"], r["distillation_tau"])
    hp = O2.Stage2Hyper(learning_rate=r["learning_rate"], train_batch_size=r["batch"], weight_decay=r["weight_decay"],
                        max_grad_norm=r["max_grad_norm"], epochs=r["epochs"], warmup_epochs=r["warmup_epochs"],
                        warmup_lr=r["warmup_lr"], min_lr=r["min_lr"], decay_rate=r["decay_rate"], opt_eps=r["opt_eps"])
    model.block_skip_gating.requires_grad = False                                                   # :313
    named = {n: p for n, p in model.named_parameters() if p.requires_grad}                          # add_weight_decay skips frozen tensors
    wd_of = O2.weight_decay_groups(named, r["weight_decay"], tuple(model.no_weight_decay()))
    groups = [dict(params=[p for n, p in named.items() if wd_of[n] == 0.0], weight_decay=0.0),
              dict(params=[p for n, p in named.items() if wd_of[n] != 0.0], weight_decay=r["weight_decay"])]
    optimizer = torch.optim.AdamW(groups, lr=hp.lr, weight_decay=0.0, eps=r["opt_eps"])
    model.zero_grad()
    model.train()
    names = [k for k, _ in model.named_parameters()]
    out = dict(param_names=np.array(names), no_decay=np.array([n for n in named if wd_of[n] == 0.0]),
               frozen=np.array([n for n, p in model.named_parameters() if not p.requires_grad]),
               state_dict_keys=np.array(list(model.state_dict().keys())))
    for step in range(r["steps"]):
        epoch = r["epoch_of_step"][step]
        lr = O2.cosine_epoch_lr(epoch, hp.lr, hp.epochs, hp.min_lr, hp.warmup_epochs, hp.warmup_lr, hp.decay_rate)
        for g in optimizer.param_groups:
            g["lr"] = lr
        x = torch.from_numpy(x_all[step]); y = torch.from_numpy(y_all[step])
        for _, mod in model.named_modules():                                                        # :343-346
            if hasattr(mod, "mask"):
                mod.weight.data *= mod.mask
        outputs, flops_list = model(x)                                                              # :363
        loss = criterion(x, outputs, y)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), r["max_grad_norm"])              # :377
        pre = f"step{step}."
        out[pre + "grad_abs_sum"] = np.array([float(p.grad.double().abs().sum()) if p.grad is not None else np.nan
                                              for _, p in model.named_parameters()])
        t2t = model.tokens_to_token
        out[pre + "g_kqv1"] = t2t.attention1.kqv.weight.grad[:8].numpy().copy()                     # rows of the deepest gradients
        out[pre + "g_norm1_1"] = t2t.attention1.norm1.weight.grad.numpy().copy()
        out[pre + "g_proj2"] = t2t.attention2.proj.weight.grad.numpy().copy()
        out[pre + "g_project_b"] = t2t.project.bias.grad.numpy().copy()
        optimizer.step()
        optimizer.zero_grad()
        out[pre + "lr"] = np.float64(lr)
        out[pre + "loss"] = np.float64(loss.item())
        out[pre + "logits"] = outputs[0].detach().numpy()
        out[pre + "grad_norm"] = np.float64(float(gnorm))
        out[pre + "blocks_run"] = np.array([int(len(b) > 0) for b in flops_list[1]], dtype=np.int64)
        out[pre + "param_abs_sum"] = np.array([float(p.data.double().abs().sum()) for _, p in model.named_parameters()])
        out[pre + "kqv1_row0"] = t2t.attention1.kqv.weight.data[0].numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB) loss0={out['step0.loss']:.6f} gnorm0={out['step0.grad_norm']:.4f} "
          f"blocks_run={out['step0.blocks_run'].tolist()} frozen={list(out['frozen'])}")


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(TS.STAGE2)):
        run(n)
