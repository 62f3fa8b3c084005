"""Oracle: one full Stage-1 step (joint_train.py:395-450) = student forward/backward + teacher
forward + distillation loss + global-norm clip + AdamW + warmup-cosine + uvc_optimizer.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import uvc as U
from . import vit as V


# ----------------------------------------------------------------------------- loss
def distillation_loss(o, o_kd, labels_soft, teacher_logits, kind="soft", alpha=0.1, T=1.0):
    """DistillationLoss.forward (utils/losses.py:25-65) over timm SoftTargetCrossEntropy
    (joint_train.py:940): base = mean_b sum_c -y log_softmax(o)."""
    base = torch.sum(-labels_soft * F.log_softmax(o, dim=-1), dim=-1).mean()
    if kind == "none":
        return base
    if kind == "soft":
        kd = F.kl_div(F.log_softmax(o_kd / T, dim=1), F.log_softmax(teacher_logits / T, dim=1),
                      reduction="sum", log_target=True) * (T * T) / o_kd.numel()
    else:
        kd = F.cross_entropy(o_kd, teacher_logits.argmax(dim=1))
    return base * (1 - alpha) + kd * alpha


# ----------------------------------------------------------------------------- schedule / optimiser
def warmup_cosine_lambda(step: int, warmup_steps: int, t_total: int, cycles: float = 0.5) -> float:
    """WarmupCosineSchedule.lr_lambda (utils/scheduler.py:58-63)."""
    if step < warmup_steps:
        return float(step) / float(max(1.0, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, t_total - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * progress)))


def clip_grad_norm(grads: List[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (joint_train.py:428): L2 norm of per-tensor L2 norms."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


@dataclass
class AdamWState:
    lr0: float = 1e-4
    wd: float = 0.05
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    m: Dict[str, torch.Tensor] = field(default_factory=dict)
    v: Dict[str, torch.Tensor] = field(default_factory=dict)
    t: Dict[str, int] = field(default_factory=dict)


def adamw_step(opt: AdamWState, params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]], lr: float,
               wd_of: Optional[Dict[str, float]] = None):
    """torch.optim.AdamW(lr, betas=(.9,.999), eps=1e-8, weight_decay=.05) (joint_train.py:271,429):
    decoupled decay on every parameter that HAS a gradient; parameters whose grad is None are
    skipped entirely (gumbel.*, attn/mlp_skip_gating, block_skip_gating during warm-up)."""
    for name, p in params.items():
        g = grads.get(name)
        if g is None:
            continue
        if name not in opt.m:
            opt.m[name] = torch.zeros_like(p)
            opt.v[name] = torch.zeros_like(p)
            opt.t[name] = 0
        opt.t[name] += 1
        t = opt.t[name]
        p.mul_(1 - lr * (opt.wd if wd_of is None else wd_of[name]))      # wd_of: per-parameter groups (Stage-2)
        opt.m[name].lerp_(g, 1 - opt.b1)
        opt.v[name].mul_(opt.b2).addcmul_(g, g, value=1 - opt.b2)
        bc1 = 1 - opt.b1 ** t
        bc2 = 1 - opt.b2 ** t
        denom = (opt.v[name].sqrt() / math.sqrt(bc2)).add_(opt.eps)
        p.addcdiv_(opt.m[name], denom, value=-(lr / bc1))


# ----------------------------------------------------------------------------- the step
@dataclass
class TrainHyper:
    """The model-side flags of the README command (run_uvc_train.sh:4-38, joint_train.py:684-879)."""
    learning_rate: float = 1e-4
    weight_decay: float = 0.05
    max_grad_norm: float = 1.0
    warmup_steps: int = 500
    t_total: int = 150150
    warmup_lr: float = 1e-4
    distillation_type: str = "soft"
    distillation_alpha: float = 0.1
    distillation_tau: float = 1.0
    enable_patch_gating: int = 0
    patch_ratio: float = 0.9
    patch_tau: float = -1.0


@dataclass
class Stage1:
    cfg: V.VitConfig
    flags: V.GateFlags
    params: Dict[str, torch.Tensor]          # student, updated in place
    teacher: Dict[str, torch.Tensor]
    st: U.UvcState
    hp: U.UvcHyper
    th: TrainHyper
    opt: AdamWState = None
    sched_step: int = 0                      # LambdaLR.last_epoch
    lr: float = 0.0                          # optimizer.param_groups[0]['lr']
    global_step: int = 0
    fwd: Optional[Callable] = None           # model forward with V.forward's signature (None = DeiT; oracle/t2t.py:forward_flags for T2T-ViT)
    frozen: Tuple[str, ...] = ()             # requires_grad False parameters (T2T: pos_embed, the Performer random features)

    def __post_init__(self):
        if self.opt is None:
            self.opt = AdamWState(lr0=self.th.learning_rate, wd=self.th.weight_decay)
        self.lr = self.th.learning_rate * warmup_cosine_lambda(0, self.th.warmup_steps, self.th.t_total)

    def w1(self):
        return [self.params[f"blocks.{i}.attn.proj.weight"] for i in range(self.cfg.depth)]

    def w3(self):
        return [self.params[f"blocks.{i}.mlp.fc2.weight"] for i in range(self.cfg.depth)]


def teacher_flags() -> V.GateFlags:
    """Teacher = same class with default ctor flags, eval() (joint_train.py:957-981)."""
    return V.GateFlags(enable_block_gating=0, training=False)


def stage1_step(S: Stage1, x: torch.Tensor, y_soft: torch.Tensor, exp_model: List[torch.Tensor],
                e1: torch.Tensor, e2: Optional[torch.Tensor], out: Optional[dict] = None):
    """joint_train.py:395-450 after mixup.  exp_model: Exp(1) draws consumed by the student forward;
    e1/e2: draws of the two resource evaluations inside uvc_optimizer."""
    fwd = S.fwd if S.fwd is not None else V.forward
    for k, p in S.params.items():
        p.requires_grad_(k not in S.frozen)
        p.grad = None
    gate_trainable = not S.flags.enable_warmup          # joint_train.py:349,358
    S.params["block_skip_gating"].requires_grad_(bool(gate_trainable))
    rec: dict = {}
    (o, od), _ = fwd(S.params, S.cfg, S.flags, x, tau=S.th.patch_tau, ratio=S.th.patch_ratio,
                     exp_draws=exp_model, record=rec)
    with torch.no_grad():
        tl, _ = fwd(S.teacher, S.cfg, teacher_flags(), x)
    loss = distillation_loss(o, od, y_soft, tl, S.th.distillation_type, S.th.distillation_alpha,
                             S.th.distillation_tau)
    loss.backward()
    grads = {k: p.grad for k, p in S.params.items()}
    with torch.no_grad():
        live = [g for g in grads.values() if g is not None]
        gnorm = clip_grad_norm(live, S.th.max_grad_norm)                      # :428
        for p in S.params.values():
            p.requires_grad_(False)
        adamw_step(S.opt, S.params, grads, S.lr)                              # :429
        S.sched_step += 1                                                     # :430
        S.lr = S.th.learning_rate * warmup_cosine_lambda(S.sched_step, S.th.warmup_steps, S.th.t_total)
        S.global_step += 1                                                    # :434
        g = S.params["block_skip_gating"]
        cur = U.uvc_update(S.st, S.hp, S.w1(), S.w3(), S.lr, g, grads.get("block_skip_gating"),
                           e1, e2, S.flags.enable_warmup, S.global_step)      # :444
    if out is not None:
        out.update(loss=loss.detach(), logits=o.detach(), logits_dist=od.detach(), teacher_logits=tl,
                   grad_norm=gnorm, cur_resource=cur, grads=grads, distribs=rec.get("distribs"),
                   patch_index=rec.get("patch_index"))
    return loss.item(), cur
