#!/usr/bin/env python3
"""Generate the Stage-2 fixtures (stage2_*.npz) by running the REFERENCE's own model and loss.

Build container only (needs /root/reference).  Usage: python tests/golden/make_stage2_golden.py [scenario ...]

post_train.py itself cannot be imported (apex, timm, tensorboard), so its 35-line loop body (:341-377) and the
set-up around it (:149-157 model + masks, :297-313 lr scaling / optimiser / gate freeze) are re-stated here over
the reference's DistilledVisionTransformer and DistillationLoss and torch.optim.AdamW.  timm (0.3.2) is absent:
its add_weight_decay grouping and cosine epoch schedule are restated (oracle/stage2.py says which parity claims
that leaves unpinned).
"""
from __future__ import annotations

import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
import scenarios as SC  # noqa: E402
from oracle import vit as OV  # noqa: E402  (portable weight recipe only)
from oracle import stage2 as O2  # noqa: E402  (timm restatements: decay groups + epoch lr)

ref_shim.install()
from models.model_distilled import DistilledVisionTransformer  # noqa: E402
from utils.losses import DistillationLoss  # noqa: E402


class SoftTargetCrossEntropy(nn.Module):
    """timm.loss.SoftTargetCrossEntropy (absent from the image; one-liner)."""

    def forward(self, x, target):
        return torch.sum(-target * torch.nn.functional.log_softmax(x, dim=-1), dim=-1).mean()


def run(name):
    r = SC.stage2_recipe(name)
    m = r["model_cfg"]
    cfg = OV.VitConfig(img_size=m["img_size"], patch_size=m["patch_size"], num_classes=m["num_classes"],
                       embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"], mlp_ratio=m["mlp_ratio"],
                       enable_dist=m["enable_dist"])
    L, H, hd, F = cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden
    params = OV.init_params_numpy(cfg, r["seed"], 0, weight_gain=m["weight_gain"])
    tparams = OV.init_params_numpy(cfg, r["seed"] + 500, 0, weight_gain=m["weight_gain"])
    keep_proj, keep_hidden, gate = SC.stage2_masks(r, L, H, hd, F)
    x_all, y_all = SC.make_inputs(r)
    kw = dict(patch_size=m["patch_size"], embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"],
              mlp_ratio=m["mlp_ratio"], qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_rate=0,
              img_size=m["img_size"], num_classes=m["num_classes"])
    torch.manual_seed(r["seed"])
    model = DistilledVisionTransformer(enable_dist=m["enable_dist"], gumbel_hard=True, **kw)      # post_train.py:149-154
    for _, p in model.named_modules():                                                              # :155-157
        if hasattr(p, "weight"):
            p.register_buffer("mask", torch.ones_like(p.weight))
    # the Stage-1 checkpoint: weights, masks, gate logits (strict load, :683)
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
    teacher = None
    if r["distillation_type"] != "none":
        teacher = DistilledVisionTransformer(enable_dist=m["enable_dist"], **kw)
        teacher.load_state_dict(tparams, strict=False)
        teacher.eval()
    criterion = DistillationLoss(SoftTargetCrossEntropy(), teacher, r["distillation_type"], r["distillation_alpha"],
                                 r["distillation_tau"])
    hp = O2.Stage2Hyper(learning_rate=r["learning_rate"], train_batch_size=r["batch"], weight_decay=r["weight_decay"],
                        max_grad_norm=r["max_grad_norm"], epochs=r["epochs"], warmup_epochs=r["warmup_epochs"],
                        warmup_lr=r["warmup_lr"], min_lr=r["min_lr"], decay_rate=r["decay_rate"], opt_eps=r["opt_eps"])
    named = dict(model.named_parameters())
    wd_of = O2.weight_decay_groups(named, r["weight_decay"])                                        # create_optimizer, :299
    groups = [dict(params=[p for n, p in named.items() if wd_of[n] == 0.0], weight_decay=0.0),
              dict(params=[p for n, p in named.items() if wd_of[n] != 0.0], weight_decay=r["weight_decay"])]
    optimizer = torch.optim.AdamW(groups, lr=hp.lr, weight_decay=0.0, eps=r["opt_eps"])
    model.block_skip_gating.requires_grad = False                                                   # :313
    model.zero_grad()
    model.train()
    names = [k for k, _ in model.named_parameters()]
    out = dict(param_names=np.array(names), no_decay=np.array([n for n in names if wd_of[n] == 0.0]),
               state_dict_keys=np.array(list(model.state_dict().keys())),
               mask_count=np.float64(float(sum(p.mask.sum() for _, p in model.named_modules() if hasattr(p, "mask")) / 1e6)))
    for step in range(r["steps"]):
        epoch = r["epoch_of_step"][step]
        lr = O2.cosine_epoch_lr(epoch, hp.lr, hp.epochs, hp.min_lr, hp.warmup_epochs, hp.warmup_lr, hp.decay_rate)
        for g in optimizer.param_groups:                                                            # scheduler.step(epoch), :339
            g["lr"] = lr
        x = torch.from_numpy(x_all[step]); y = torch.from_numpy(y_all[step])
        for _, mod in model.named_modules():                                                        # :343-346
            if hasattr(mod, "mask"):
                mod.weight.data *= mod.mask
        outputs, flops_list = model(x)                                                              # :363
        loss = criterion(x, outputs, y)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), r["max_grad_norm"])              # :377
        gsum = np.array([float(p.grad.double().abs().sum()) if p.grad is not None else np.nan
                         for _, p in model.named_parameters()])
        optimizer.step()
        optimizer.zero_grad()
        pre = f"step{step}."
        out[pre + "lr"] = np.float64(lr)
        out[pre + "loss"] = np.float64(loss.item())
        out[pre + "logits"] = outputs[0].detach().numpy()
        out[pre + "logits_dist"] = outputs[1].detach().numpy()
        out[pre + "grad_norm"] = np.float64(float(gnorm))
        out[pre + "grad_abs_sum"] = gsum
        out[pre + "blocks_run"] = np.array([int(len(b) > 0) for b in flops_list[1]], dtype=np.int64)
        out[pre + "param_sum"] = np.array([float(p.data.double().sum()) for _, p in model.named_parameters()])
        out[pre + "param_abs_sum"] = np.array([float(p.data.double().abs().sum()) for _, p in model.named_parameters()])
        out[pre + "proj_0_row0"] = model.blocks[0].attn.proj.weight.data[0].numpy().copy()
        out[pre + "fc1_last_col0"] = model.blocks[L - 1].mlp.fc1.weight.data[:, 0].numpy().copy()
        out[pre + "pos_embed_tok0"] = model.pos_embed.data[0, 0].numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB) loss0={out['step0.loss']:.6f} gnorm0={out['step0.grad_norm']:.4f} "
          f"masks={out['mask_count']:.6f}M blocks_run={out['step0.blocks_run'].tolist()}")


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(SC.STAGE2)):
        run(n)
