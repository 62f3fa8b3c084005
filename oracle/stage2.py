"""Oracle: one Stage-2 masked fine-tune step (post_train.py:341-377) and its set-up (:289-313).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Pinning: the model / loss / autograd / AdamW arithmetic is pinned by tests/golden/stage2_*.npz, which
tests/golden/make_stage2_golden.py generates from the reference's own DistilledVisionTransformer and
DistillationLoss with torch.optim.AdamW.  Two pieces come from timm (pinned 0.3.2,
Baseline_pruning/requirements.txt:3), which is NOT in the image and therefore restated from its published
behaviour in both the generator and here -- parity for exactly these two is UNPINNED:
  * timm.optim.create_optimizer -> add_weight_decay: no decay for 1-D tensors, names ending in ".bias" and
    model.no_weight_decay() = {pos_embed, cls_token, dist_token} (model_distilled.py:330-331);
  * timm.scheduler.CosineLRScheduler as create_scheduler builds it for --sched cosine (per-epoch values).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Tuple

import torch

from . import step as S1
from . import vit as V

NO_DECAY_NAMES = ("pos_embed", "cls_token", "dist_token")          # model_distilled.py:330-331


def weight_decay_groups(params: Dict[str, torch.Tensor], weight_decay: float, no_decay_names=NO_DECAY_NAMES) -> Dict[str, float]:
    """timm.optim.optim_factory.add_weight_decay (0.3.2) as a name -> decay table (``no_decay_names`` = model.no_weight_decay():
    DeiT's by default; T2T-ViT's is {cls_token}, t2t_vit.py:153-155)."""
    return {n: (0.0 if (p.ndim == 1 or n.endswith(".bias") or n in no_decay_names) else weight_decay)
            for n, p in params.items()}


def cosine_epoch_lr(t: int, base_lr: float, t_initial: int, lr_min: float, warmup_t: int, warmup_lr_init: float,
                    decay_rate: float = 0.1, cycle_limit: int = 1) -> float:
    """timm.scheduler.CosineLRScheduler._get_lr (0.3.2; t_mul = 1, no warm-up prefix, no noise)."""
    if t < warmup_t:
        return warmup_lr_init + t * (base_lr - warmup_lr_init) / warmup_t
    i = t // t_initial
    t_curr = t - t_initial * i
    gamma = decay_rate ** i
    lo, hi = lr_min * gamma, base_lr * gamma
    if cycle_limit == 0 or i < cycle_limit:
        return lo + 0.5 * (hi - lo) * (1 + math.cos(math.pi * t_curr / t_initial))
    return lr_min


@dataclass
class Stage2Hyper:
    """post_train.py argparse defaults (:432-493, :557-559) that reach the step."""
    learning_rate: float = 1e-4
    train_batch_size: int = 64
    world_size: int = 1
    weight_decay: float = 0.05
    max_grad_norm: float = 1.0
    epochs: int = 100
    warmup_epochs: int = 5
    warmup_lr: float = 1e-6
    min_lr: float = 1e-5
    decay_rate: float = 0.1
    opt_eps: float = 1e-8
    distillation_type: str = "none"
    distillation_alpha: float = 0.5
    distillation_tau: float = 1.0

    @property
    def lr(self) -> float:                                        # :297-298
        return self.learning_rate * self.train_batch_size * self.world_size / 512.0


@dataclass
class Stage2:
    cfg: V.VitConfig
    params: Dict[str, torch.Tensor]                               # student, updated in place
    masks: Dict[str, torch.Tensor]                                # "<module>.weight" -> mask (missing = ones)
    teacher: Optional[Dict[str, torch.Tensor]]
    hp: Stage2Hyper
    opt: S1.AdamWState = None
    wd_of: Dict[str, float] = field(default_factory=dict)
    cur_lr: float = 0.0
    global_step: int = 0
    fwd: Optional[Callable] = None            # None = DeiT (oracle/vit.py:forward); oracle/t2t.py:forward_flags for T2T-ViT
    frozen: Tuple[str, ...] = ()              # requires_grad False parameters (T2T: pos_embed, the Performer random features)
    no_decay_names: Tuple[str, ...] = NO_DECAY_NAMES

    def __post_init__(self):
        if self.opt is None:
            self.opt = S1.AdamWState(lr0=self.hp.lr, wd=self.hp.weight_decay, eps=self.hp.opt_eps)
        self.wd_of = weight_decay_groups(self.params, self.hp.weight_decay, self.no_decay_names)
        self.cur_lr = self.hp.warmup_lr if self.hp.warmup_epochs else self.hp.lr

    def begin_epoch(self, epoch: int):                            # scheduler.step(epoch), :339
        h = self.hp
        self.cur_lr = cosine_epoch_lr(epoch, h.lr, h.epochs, h.min_lr, h.warmup_epochs, h.warmup_lr, h.decay_rate)


def student_flags() -> V.GateFlags:
    """post_train.py:149-154: default constructor flags -> enable_block_gating=0 (hard skip), train mode."""
    return V.GateFlags(enable_block_gating=0, training=True)


def stage2_step(S: Stage2, x: torch.Tensor, y_soft: torch.Tensor, out: Optional[dict] = None):
    """post_train.py:341-377 after mixup."""
    with torch.no_grad():
        for n, m in S.masks.items():                              # :343-346
            S.params[n].mul_(m)
    fwd = S.fwd if S.fwd is not None else V.forward
    for k, p in S.params.items():
        p.requires_grad_(k not in S.frozen)
        p.grad = None
    S.params["block_skip_gating"].requires_grad_(False)           # :313,331
    (o, od), _ = fwd(S.params, S.cfg, student_flags(), x)         # :363
    tl = None
    if S.hp.distillation_type != "none":
        with torch.no_grad():
            tl, _ = fwd(S.teacher, S.cfg, S1.teacher_flags(), x)
    loss = S1.distillation_loss(o, od, y_soft, tl, S.hp.distillation_type, S.hp.distillation_alpha, S.hp.distillation_tau)
    loss.backward()
    grads = {k: p.grad for k, p in S.params.items()}
    with torch.no_grad():
        live = [g for g in grads.values() if g is not None]
        gnorm = S1.clip_grad_norm(live, S.hp.max_grad_norm)       # :377
        for p in S.params.values():
            p.requires_grad_(False)
        S1.adamw_step(S.opt, S.params, grads, S.cur_lr, wd_of=S.wd_of)
        S.global_step += 1
    if out is not None:
        out.update(loss=loss.detach(), logits=o.detach(), logits_dist=od.detach(), teacher_logits=tl, grad_norm=gnorm,
                   grads=grads)
    return loss.item()
