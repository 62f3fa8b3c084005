"""Shared test plumbing: load golden fixtures, build the oracle state for a scenario, split the
recorded Exp(1) draws in the order the reference consumed them (SURVEY.md §8c note 3)."""
from __future__ import annotations

import os

import numpy as np
import torch

import scenarios as SC
from oracle import step as OS
from oracle import uvc as OU
from oracle import vit as OV

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def vit_config(r) -> OV.VitConfig:
    m = r["model_cfg"]
    return OV.VitConfig(img_size=m["img_size"], patch_size=m["patch_size"], num_classes=m["num_classes"],
                        embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"],
                        mlp_ratio=m["mlp_ratio"], enable_dist=m["enable_dist"])


def uvc_hyper(r) -> OU.UvcHyper:
    return OU.UvcHyper(budget=r["budget"], slr=r["slr"], rlr=r["rlr"], glr=r["glr"], ylr=r["ylr"], plr=r["plr"],
                       zlr=float(r["zlr"]), sl2wd=r["sl2wd"], z_grad_clip=r["z_grad_clip"],
                       gating_interval=r["gating_interval"], gating_weight=r["gating_weight"],
                       use_gumbel=r["use_gumbel"], enable_block_gating=r["enable_block_gating"],
                       eps_decay=r["eps_decay"])


def train_hyper(r) -> OS.TrainHyper:
    return OS.TrainHyper(learning_rate=r["learning_rate"], weight_decay=r["weight_decay"],
                         max_grad_norm=r["max_grad_norm"], warmup_steps=r["warmup_steps"], t_total=r["t_total"],
                         warmup_lr=r["warmup_lr"], distillation_alpha=r["distillation_alpha"],
                         distillation_tau=r["distillation_tau"], enable_patch_gating=r["enable_patch_gating"],
                         patch_ratio=r["patch_ratio"],
                         patch_tau=r["patch_tau"] if r["enable_patch_gating"] == 2 else -1.0)


def student_flags(r) -> OV.GateFlags:
    # joint_train.py:135-140 (gumbel_hard=False) + uvc_optimizer.py:204-210 + epoch header :344-360
    return OV.GateFlags(enable_block_gating=r["enable_block_gating"], enable_patch_gating=r["enable_patch_gating"],
                        use_gumbel=r["use_gumbel"], eps=r["eps"], enable_warmup=r["warmup"], gumbel_hard=False,
                        training=True)


def initial_params(r):
    cfg = vit_config(r)
    m = r["model_cfg"]
    params = OV.init_params_numpy(cfg, r["seed"], r["enable_patch_gating"], weight_gain=m["weight_gain"])
    teacher = OV.init_params_numpy(cfg, r["seed"] + 500, 0, weight_gain=m["weight_gain"])
    return cfg, params, teacher


def build_oracle(name):
    return build_oracle_from_recipe(SC.recipe(name))


def build_oracle_from_recipe(r):
    cfg, params, teacher = initial_params(r)
    if r["enable_patch_gating"] == 1:
        # UVC_CP_MiniMax replaces model.patch_gating by its own 3*ones parameter (uvc_utils.py:152,286-288)
        params["patch_gating"] = torch.full((1, cfg.num_patches, 1), 3.0)
    embed, macs = OV.mac_table(cfg, 1)
    st = OU.UvcState.create(cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden, embed, macs, eps=r["eps"])
    s0, r0, y0, p0, z0 = SC.initial_state(r, cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden)
    st.s, st.r = torch.from_numpy(s0.copy()), torch.from_numpy(r0.copy())
    st.y, st.p, st.z = torch.from_numpy(y0.copy()), torch.from_numpy(p0.copy()), torch.tensor(float(z0))
    S = OS.Stage1(cfg=cfg, flags=student_flags(r), params=params, teacher=teacher, st=st, hp=uvc_hyper(r),
                  th=train_hyper(r))
    if r["warmup"]:
        S.lr = r["warmup_lr"]           # joint_train.py:350-351
    return r, S


def split_draws(r, gold, step, L):
    """Recorded draws of one step -> (model draws list, e1, e2)."""
    n = int(gold[f"step{step}.n_draws"])
    draws = [torch.from_numpy(gold[f"step{step}.draw{i}"]) for i in range(n)]
    model = []
    if r["enable_patch_gating"] == 2:
        model.append(draws.pop(0))
    gum = r["enable_block_gating"] and r["use_gumbel"]
    if gum and not r["warmup"]:
        for _ in range(L):
            model.append(draws.pop(0))
    e1 = draws.pop(0) if gum else None
    e2 = draws.pop(0) if (gum and not r["warmup"]) else None
    assert not draws, f"unconsumed draws: {len(draws)}"
    return model, e1, e2


# ----------------------------------------------------------------------------------------------------- Stage-2
def stage2_state(r):
    """The Stage-1 checkpoint a Stage-2 scenario starts from, as a state_dict of torch tensors (weights from the
    portable recipe, structured masks, gate logits), plus the teacher's."""
    cfg = vit_config(r)
    m = r["model_cfg"]
    params = OV.init_params_numpy(cfg, r["seed"], 0, weight_gain=m["weight_gain"])
    teacher = OV.init_params_numpy(cfg, r["seed"] + 500, 0, weight_gain=m["weight_gain"])
    keep_proj, keep_hidden, gate = SC.stage2_masks(r, cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden)
    params["block_skip_gating"] = torch.from_numpy(gate.copy())
    D = cfg.embed_dim
    masks = {}
    for l in range(cfg.depth):
        masks[f"blocks.{l}.attn.proj.weight"] = torch.from_numpy(keep_proj[l])[None, :].expand(D, -1).clone()
        masks[f"blocks.{l}.mlp.fc2.weight"] = torch.from_numpy(keep_hidden[l])[None, :].expand(D, -1).clone()
        masks[f"blocks.{l}.mlp.fc1.weight"] = torch.from_numpy(keep_hidden[l])[:, None].expand(-1, D).clone()
    return cfg, params, masks, teacher


def stage2_hyper(r):
    from oracle import stage2 as O2
    return O2.Stage2Hyper(learning_rate=r["learning_rate"], train_batch_size=r["batch"], weight_decay=r["weight_decay"],
                          max_grad_norm=r["max_grad_norm"], epochs=r["epochs"], warmup_epochs=r["warmup_epochs"],
                          warmup_lr=r["warmup_lr"], min_lr=r["min_lr"], decay_rate=r["decay_rate"], opt_eps=r["opt_eps"],
                          distillation_type=r["distillation_type"], distillation_alpha=r["distillation_alpha"],
                          distillation_tau=r["distillation_tau"])


def build_oracle_stage2(name):
    from oracle import stage2 as O2
    r = SC.stage2_recipe(name)
    cfg, params, masks, teacher = stage2_state(r)
    S = O2.Stage2(cfg=cfg, params=params, masks=masks, teacher=teacher if r["distillation_type"] != "none" else None,
                  hp=stage2_hyper(r))
    return r, S
