"""The Stage-1 step body (UVC/joint_train.py:395-450) written against the PRODUCT API (uvc_amd), fed
with a golden scenario's inputs and the reference's recorded RNG draws.  Used by the GPU parity tests,
__graft_entry__.smoke() and bench.py."""
from __future__ import annotations

from argparse import Namespace

import numpy as np
import torch

import scenarios as SC
from helpers import initial_params, load_golden, split_draws, vit_config


def make_args(r, cfg):
    return Namespace(eps_decay=r["eps_decay"], enable_patch_gating=r["enable_patch_gating"], enable_part_gating=0,
                     enable_block_gating=r["enable_block_gating"], head_size=cfg.head_dim, num_heads=cfg.num_heads,
                     flops_with_mhsa=1, use_gumbel=r["use_gumbel"], enable_jumping=0, eps=r["eps"], enable_warmup=r["warmup"],
                     soptim="sgd", roptim="sgd", slr=r["slr"], rlr=r["rlr"], glr=r["glr"], zlr_schedule_list=[int(r["zlr"])],
                     ylr=r["ylr"], plr=r["plr"], budget=r["budget"], sl2wd=r["sl2wd"], gating_weight=r["gating_weight"],
                     z_grad_clip=r["z_grad_clip"], gating_interval=r["gating_interval"], patch_ratio=r["patch_ratio"])


class Stage1Run:
    """Everything joint_train.main()/train() set up before the loop (joint_train.py:122-171,948-1026)."""

    def __init__(self, name_or_recipe, precision="fp32", batch=None):
        from uvc_amd.joint_train import get_uvc_layers, register_masks
        from uvc_amd.losses import DistillationLoss, SoftTargetCrossEntropy
        from uvc_amd.model_distilled import DistilledVisionTransformer
        from uvc_amd.optim import FusedAdamW
        from uvc_amd.scheduler import WarmupCosineSchedule
        from uvc_amd.uvc_optimizer import build_minimax_model
        from uvc_amd.uvc_utils import prune_w_mask
        r = SC.recipe(name_or_recipe) if isinstance(name_or_recipe, str) else name_or_recipe
        self.r = r
        cfg, params, teacher_params = initial_params(r)
        self.cfg = cfg
        m = r["model_cfg"]
        kw = dict(img_size=m["img_size"], patch_size=m["patch_size"], num_classes=m["num_classes"], embed_dim=m["embed_dim"],
                  depth=m["depth"], num_heads=m["num_heads"], mlp_ratio=m["mlp_ratio"], qkv_bias=True, drop_rate=0,
                  precision=precision)
        model = DistilledVisionTransformer(enable_dist=m["enable_dist"], gumbel_hard=False,
                                           enable_patch_gating=r["enable_patch_gating"], **kw)    # joint_train.py:135-140
        model.load_state_dict(params, strict=False)
        register_masks(model)
        teacher = DistilledVisionTransformer(enable_dist=m["enable_dist"], **kw)                    # :957-961
        teacher.load_state_dict({k: v for k, v in teacher_params.items() if k != "patch_gating"}, strict=False)
        teacher.eval()
        teacher.frozen_weights = True
        self.model, self.teacher = model, teacher
        self.criterion = DistillationLoss(SoftTargetCrossEntropy(), teacher, "soft", r["distillation_alpha"], r["distillation_tau"])
        self.args = make_args(r, cfg)
        names, layers, ldict = get_uvc_layers(model)
        self.layers = layers
        with torch.no_grad():                                                                         # :1010-1012
            model.eval()
            _, flops_list = model(torch.ones(1, 3, cfg.img_size, cfg.img_size, device="cuda"), number=r["patch_ratio"])
        self.flops_list = flops_list
        (self.minimax, self.dual_opt, self.s_opt, self.r_opt, self.g_opt) = build_minimax_model(
            model, names, layers, ldict, self.args, flops_list)
        s0, r0, y0, p0, z0 = SC.initial_state(r, cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden)
        mm = self.minimax
        mm.s.data.copy_(torch.from_numpy(s0)); mm.r.data.copy_(torch.from_numpy(r0))
        mm.y.data.copy_(torch.from_numpy(y0)); mm.p.data.copy_(torch.from_numpy(p0)); mm.z.data.fill_(float(z0))
        self.optimizer = FusedAdamW(model, lr=r["learning_rate"], weight_decay=r["weight_decay"])
        self.scheduler = WarmupCosineSchedule(self.optimizer, warmup_steps=r["warmup_steps"], t_total=r["t_total"])
        model.train()
        if r["warmup"]:                                                                               # :344-351
            model.enable_warmup = 1
            model.block_skip_gating.requires_grad = False
            for g in self.optimizer.param_groups:
                g["lr"] = r["warmup_lr"]
        else:
            model.enable_warmup = 0
            model.block_skip_gating.requires_grad = True
        prune_w_mask(mm, self.optimizer)                                                              # :377
        self.global_step = 0
        self.gating_grad_list = []

    def inject_draws(self, model_draws, e1, e2):
        """Feed the reference's recorded Exp(1) draws: the student consumes one [L,2] block-gate draw."""
        L = self.cfg.depth
        dev = "cuda"
        if model_draws:
            stacked = torch.stack([d.reshape(2) for d in model_draws[-L:]]).to(dev)
            self.model.exp_source = lambda shape, t=stacked: t
        q = [t.to(dev) for t in (e1, e2) if t is not None]
        self.minimax.exp_source = lambda shape, q=q: q.pop(0)

    def step(self, x, y):
        """joint_train.py:395-450 (after mixup)."""
        from uvc_amd.optim import clip_grad_norm_
        from uvc_amd.uvc_optimizer import uvc_optimizer
        r = self.r
        tau = r["patch_tau"] if r["enable_patch_gating"] == 2 else -1
        outputs, _ = self.model(x, tau, r["patch_ratio"])
        loss = self.criterion(x, outputs, y)
        loss.backward()
        gnorm = clip_grad_norm_(self.model, r["max_grad_norm"])
        self.optimizer.step()
        self.scheduler.step()
        self.global_step += 1
        self.minimax.update_gating()
        cur, s, rr, g, self.gating_grad_list = uvc_optimizer(
            self.optimizer, self.minimax, self.s_opt, self.r_opt, self.g_opt, self.dual_opt, self.args, {}, [],
            self.flops_list, r["z_grad_clip"], self.global_step, r["gating_interval"], self.gating_grad_list)
        out = dict(loss=loss.detach(), outputs=outputs, gnorm=gnorm, cur=cur, s=s, r=rr, g=g)
        return out
