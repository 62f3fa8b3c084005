"""Builds the PRODUCT's Stage1Trainer (uvc_amd/stage1.py) for a golden scenario: the scenario's
portable weights / initial primal-dual state, and the reference's recorded RNG draws injected into the
model's and the minimax object's ``exp_source`` hooks."""
from __future__ import annotations

import torch

import scenarios as SC
from helpers import initial_params


def recipe_args(r, precision):
    from uvc_amd.stage1 import default_args
    m = r["model_cfg"]
    return default_args(
        model_type="scenario", model_cfg=dict(patch_size=m["patch_size"], embed_dim=m["embed_dim"], depth=m["depth"],
                                              num_heads=m["num_heads"], mlp_ratio=m["mlp_ratio"]),
        img_size=m["img_size"], num_classes=m["num_classes"], enable_deit=m["enable_dist"], precision=precision,
        learning_rate=r["learning_rate"], weight_decay=r["weight_decay"], max_grad_norm=r["max_grad_norm"],
        warmup_steps=r["warmup_steps"], steps_per_epoch=r["t_total"], num_epochs=1, warmup_lr=r["warmup_lr"],
        distillation_alpha=r["distillation_alpha"], distillation_tau=r["distillation_tau"],
        enable_patch_gating=r["enable_patch_gating"], patch_ratio=r["patch_ratio"], budget=r["budget"], slr=r["slr"],
        rlr=r["rlr"], glr=r["glr"], ylr=r["ylr"], plr=r["plr"], zlr_schedule_list=str(int(r["zlr"])), sl2wd=r["sl2wd"],
        z_grad_clip=r["z_grad_clip"], gating_interval=r["gating_interval"], gating_weight=r["gating_weight"],
        use_gumbel=r["use_gumbel"], enable_block_gating=r["enable_block_gating"], eps=r["eps"], eps_decay=r["eps_decay"],
        enable_warmup=r["warmup"], warmup_epochs=1 if r["warmup"] else 0)


class Stage1Run:
    def __init__(self, name_or_recipe, precision="fp32"):
        from uvc_amd.stage1 import Stage1Trainer
        r = SC.recipe(name_or_recipe) if isinstance(name_or_recipe, str) else name_or_recipe
        self.r = r
        self.cfg, params, teacher_params = initial_params(r)
        cfg = self.cfg
        tr = Stage1Trainer(recipe_args(r, precision), student_state=params,
                           teacher_state={k: v for k, v in teacher_params.items() if k != "patch_gating"})
        self.trainer = tr
        self.model, self.teacher, self.minimax, self.optimizer = tr.model, tr.teacher, tr.minimax, tr.optimizer
        self.layers = tr.uvc_layers
        s0, r0, y0, p0, z0 = SC.initial_state(r, cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden)
        mm = self.minimax
        mm.s.data.copy_(torch.from_numpy(s0)); mm.r.data.copy_(torch.from_numpy(r0))
        mm.y.data.copy_(torch.from_numpy(y0)); mm.p.data.copy_(torch.from_numpy(p0)); mm.z.data.fill_(float(z0))
        # epoch header (joint_train.py:335-386) without the eps decay the 1-epoch fixtures never reach
        tr.gating_grad_list = []
        if r["warmup"]:
            tr.model.enable_warmup = 1
            tr.model.block_skip_gating.requires_grad = False
            for g in tr.optimizer.param_groups:
                g["lr"] = r["warmup_lr"]
        else:
            tr.model.enable_warmup = 0
            tr.model.block_skip_gating.requires_grad = True
        from uvc_amd.uvc_utils import prune_w_mask
        prune_w_mask(mm, tr.optimizer)

    def inject_draws(self, model_draws, e1, e2):
        """The student consumes its L block-gate draws as one [L,2] tensor (one launch); the patch-gating
        draw [B,P] (mode 2) comes first in the reference's order."""
        L = self.cfg.depth
        md = list(model_draws)
        by_shape = {}
        if self.r["enable_patch_gating"] == 2 and md:
            pd = md.pop(0).cuda()
            by_shape[tuple(pd.shape)] = pd
        if md:
            by_shape[(L, 2)] = torch.stack([d.reshape(2) for d in md[-L:]]).cuda()
        self.model.exp_source = lambda shape, t=by_shape: t[tuple(shape)]
        q = [t.cuda() for t in (e1, e2) if t is not None]
        self.minimax.exp_source = lambda shape, q=q: q.pop(0)

    def step(self, x, y):
        r = self.r
        tau = r["patch_tau"] if r["enable_patch_gating"] == 2 else -1
        return self.trainer.step(x, y, tau=tau, zero_grad=False)
