"""Oracle: the UVC primal-dual engine (scores, least-k selection, proximal shrink, FLOPs
resource model with analytic gradients, primal/dual SGD steps, mask writer).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows ``UVC/uvc_utils.py`` and
``UVC/uvc_optimizer.py`` of the reference; every function cites its lines.  Where the
reference uses torch autograd on a handful of scalars this file writes the derivative out, so
that the HIP kernel (uvc_amd/csrc/uvc_engine.hip) can be compared term by term.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch


# ----------------------------------------------------------------------------- scores
def scores_w1(W: torch.Tensor, H: int, hd: int):
    """weight_list_to_scores(layer,"W1") (uvc_utils.py:56-69): per-column and per-head sums of
    squares of attn.proj.weight[D_out, D_in].  float64 accumulation, one rounding to float32
    (see oracle/__init__.py for why)."""
    col = (W.double() ** 2).sum(0)                                  # [D_in]
    s1 = col.view(H, hd)
    return s1.float(), s1.sum(1).float()


def scores_w3(W: torch.Tensor):
    """weight_list_to_scores(layer,"W3") (uvc_utils.py:71-73): per-column sums of squares of
    mlp.fc2.weight[D, F]."""
    return (W.double() ** 2).sum(0).float()


def least_k(scores: torch.Tensor, k: int):
    """Index set of the k smallest scores == torch.topk(scores, k, largest=False)[1]
    (uvc_utils.py:81,238,328,334,343,387,390,398,422) with ties broken by lowest index.
    Returns (bool mask[n], sorted values ascending[n])."""
    n = scores.numel()
    k = max(0, min(int(k), n))
    order = torch.argsort(scores, stable=True)
    mask = torch.zeros(n, dtype=torch.bool)
    mask[order[:k]] = True
    return mask, scores[order]


def least_sum_and_next(scores: torch.Tensor, k: int):
    """LeastSsum (uvc_utils.py:75-92): forward value = sum of the k smallest, backward factor =
    (k+1)-th smallest, or max if k+1 > n.  Sum in float64, rounded once."""
    n = scores.numel()
    srt = torch.sort(scores)[0]
    if k + 1 <= n:
        return srt[:k].double().sum().float(), srt[k].clone()
    return srt.double().sum().float(), srt[-1].clone()


# ----------------------------------------------------------------------------- state
@dataclass
class UvcHyper:
    """Flags that matter to uvc_optimizer (joint_train.py:747-853; README command
    run_uvc_train.sh:4-38 gives the non-default values)."""
    budget: float = 0.5
    slr: float = 0.02
    rlr: float = 0.02
    glr: float = 0.1
    ylr: float = 1e-4
    plr: float = 1e-4
    zlr: float = 1.0                 # int(zlr_schedule_list[0]); the scheduler is a no-op (Q3)
    sl2wd: float = 0.0
    z_grad_clip: float = 0.5
    gating_interval: int = 50
    gating_weight: float = 5e-4
    use_gumbel: int = 1
    enable_block_gating: int = 1
    eps_decay: float = 0.92


@dataclass
class UvcState:
    """UVC_CP_MiniMax (uvc_utils.py:129-169) minus the module plumbing."""
    L: int
    H: int
    hd: int
    F: int
    s: torch.Tensor = None           # [L,2]
    r: torch.Tensor = None           # [L,H]
    y: torch.Tensor = None           # [L,2]
    p: torch.Tensor = None           # [L,H]
    z: torch.Tensor = None           # scalar
    s_ub: torch.Tensor = None
    r_ub: torch.Tensor = None
    embed_macs: float = 0.0
    total_macs: torch.Tensor = None  # [L,6] float32 (torch.Tensor(total_macs), uvc_utils.py:413)
    resource_ub: float = 0.0
    gate_momentum: Optional[torch.Tensor] = None   # SGD momentum buffer of block_skip_gating
    gating_grad_list: List[torch.Tensor] = field(default_factory=list)
    eps: float = 0.1

    @staticmethod
    def create(L, H, hd, F, embed_macs, macs_list, z_init=1e-3, y_init=1e-3, p_init=1e-3, eps=0.1):
        st = UvcState(L=L, H=H, hd=hd, F=F)
        st.s = torch.zeros(L, 2)
        st.r = torch.zeros(L, H)
        st.y = torch.full((L, 2), y_init)
        st.p = torch.full((L, H), p_init)
        st.z = torch.tensor(float(z_init))
        st.s_ub = torch.zeros(L, 2)
        st.s_ub[:, 0] = H                                         # uvc_utils.py:163
        st.s_ub[:, 1] = F                                         # :164
        st.r_ub = torch.full((L, H), float(hd))                    # :167
        st.embed_macs = float(embed_macs)
        st.total_macs = torch.Tensor(macs_list)
        # calc_flops(..., full_model_flops=None) (uvc_utils.py:467-470; uvc_optimizer.py:178-187)
        st.resource_ub = float((embed_macs + torch.Tensor(macs_list).sum()) * 2)
        st.eps = eps
        return st


# ----------------------------------------------------------------------------- prox / masks
def prox_w(st: UvcState, W1: List[torch.Tensor], W3: List[torch.Tensor], lr: float):
    """prox_w (uvc_utils.py:315-345), in place on the weight tensors.  lr is the AdamW group-0
    lr *after* scheduler.step() (Q12)."""
    cs, cr = st.s.ceil(), st.r.ceil()
    for l, W in enumerate(W1):
        s1, s2 = scores_w1(W, st.H, st.hd)
        for h in range(st.H):
            m, _ = least_k(s1[h], int(cr[l, h].item()))
            cols = torch.nonzero(m).flatten() + h * st.hd
            W[:, cols] /= (1.0 + 2.0 * lr * st.p[l, h].item())     # :329-330
        mh, _ = least_k(s2, int(cs[l, 0].item()))                  # pre-prox scores2 (:322,334)
        for h in torch.nonzero(mh).flatten().tolist():
            W[:, h * st.hd:(h + 1) * st.hd] /= (1.0 + 2.0 * lr * st.y[l, 0].item())   # :336-337
    for l, W in enumerate(W3):
        m, _ = least_k(scores_w3(W), int(cs[l, 1].item()))
        cols = torch.nonzero(m).flatten()
        W[:, cols] /= (1.0 + 2.0 * lr * st.y[l, 1].item())         # :344-345


def prune_masks(st: UvcState, W1: List[torch.Tensor], W3: List[torch.Tensor], prev_fc1: Optional[List[torch.Tensor]] = None):
    """prune_w_mask (uvc_utils.py:376-401).  Returns per layer (mask_proj[D,D], mask_fc2[D,F],
    mask_fc1[F,D]) as float 0/1 plus the index sets (proj column keep-mask[D], fc2 keep-mask[F]).
    The reference resets the proj / fc2 masks to 1 on every call (:382,393) but only writes zeros into the fc1 mask
    (:401): ``prev_fc1`` = the fc1 masks before this call (None = all ones, the first call)."""
    cs, cr = st.s.ceil(), st.r.ceil()
    out = []
    for l in range(st.L):
        D = W1[l].shape[0]
        s1, s2 = scores_w1(W1[l], st.H, st.hd)
        keep1 = torch.ones(D, dtype=torch.bool)
        for h in range(st.H):
            m, _ = least_k(s1[h], int(cr[l, h].item()))
            keep1[h * st.hd:(h + 1) * st.hd] &= ~m
        mh, _ = least_k(s2, int(cs[l, 0].item()))
        for h in torch.nonzero(mh).flatten().tolist():
            keep1[h * st.hd:(h + 1) * st.hd] = False
        m3, _ = least_k(scores_w3(W3[l]), int(cs[l, 1].item()))
        keep3 = ~m3
        mask_proj = keep1.float().unsqueeze(0).expand(D, D).contiguous()
        mask_fc2 = keep3.float().unsqueeze(0).expand(D, st.F).contiguous()
        mask_fc1 = keep3.float().unsqueeze(1).expand(st.F, D).contiguous()   # :401
        if prev_fc1 is not None:
            mask_fc1 = mask_fc1 * prev_fc1[l]
        out.append((mask_proj, mask_fc2, mask_fc1, keep1, keep3))
    return out


# ----------------------------------------------------------------------------- resource model
def gate_d1(g: torch.Tensor, e: Optional[torch.Tensor], use_gumbel: int, hard: bool, eps: float):
    """distrib1 of calc_flops (uvc_utils.py:443-449).  Returns (d1[L], dd1_dg[L,2])."""
    if g is None:
        return None, None
    if use_gumbel:
        u = (g + (-e.log())) / 0.5
        ysoft = u.softmax(1)
        d0, d1 = ysoft[:, 0], ysoft[:, 1]
        dd = torch.stack([-d0 * d1 / 0.5, d0 * d1 / 0.5], dim=1)   # softmax jvp of component 1, /tau
        if hard:
            d1 = (ysoft[:, 1] > ysoft[:, 0]).float()   # argmax one-hot (ties -> index 0)
        return d1, dd
    tmp = g ** 2
    d1 = (tmp / (tmp + eps))[:, 1]
    dd = torch.zeros_like(g)
    dd[:, 1] = 2 * g[:, 1] * eps / (tmp[:, 1] + eps) ** 2
    return d1, dd


def resource(st: UvcState, scores2_post: List[torch.Tensor], g: Optional[torch.Tensor],
             e: Optional[torch.Tensor], hp: UvcHyper, hard: bool = False, want_grad: bool = False):
    """calc_flops with full_model_flops set (uvc_utils.py:409-462) on ceil(s), ceil(r).
    Returns R (float32 scalar) and, if want_grad, dR/ds[L,2], dR/dr[L,H], dR/dg[L,2]
    (straight-through ceil; torch clamp passes the gradient on the closed interval)."""
    cs, cr = st.s.ceil(), st.r.ceil()
    L, H, hd = st.L, st.H, st.hd
    s_raw = (st.s_ub - cs) / st.s_ub
    s_ratio = s_raw.clamp(0.0, 1.0)
    Dsum = st.r_ub.sum(1)                                          # [L] == embed_dim
    attn_proj = Dsum.clone()
    notleast = torch.ones(L, H, dtype=torch.bool)
    for l in range(L):
        mh, _ = least_k(scores2_post[l], int(cs[l, 0].item()))
        notleast[l] = ~mh
        attn_proj[l] -= cs[l, 0] * hd                              # :425
        for h in range(H):
            if notleast[l, h]:
                attn_proj[l] -= cr[l, h]                           # :433
    r_raw = attn_proj / Dsum
    r_ratio = r_raw.clamp(0.0, 1.0)
    d1, dd1_dg = gate_d1(g, e, hp.use_gumbel, hard, st.eps)
    tm = st.total_macs if d1 is None else (st.total_macs.transpose(0, 1) * d1).transpose(0, 1)
    macs = st.embed_macs \
        + (tm[:, 0] * s_ratio[:, 0]).sum() + (tm[:, 1] * s_ratio[:, 0]).sum() \
        + (tm[:, 2] * r_ratio).sum() + (tm[:, 3] * r_ratio).sum() \
        + (tm[:, 4] * s_ratio[:, 1]).sum() + (tm[:, 5] * s_ratio[:, 1]).sum()
    R = macs * 2 / st.resource_ub
    if not want_grad:
        return R
    c = torch.tensor(2.0) / st.resource_ub
    in_s = ((s_raw >= 0) & (s_raw <= 1)).float()
    in_r = ((r_raw >= 0) & (r_raw <= 1)).float()
    dR_dsr0 = c * (tm[:, 0] + tm[:, 1])
    dR_dsr1 = c * (tm[:, 4] + tm[:, 5])
    dR_drr = c * (tm[:, 2] + tm[:, 3])
    gs = torch.zeros(L, 2)
    gs[:, 0] = dR_dsr0 * in_s[:, 0] * (-1.0 / st.s_ub[:, 0]) + dR_drr * in_r * (-float(hd) / Dsum)
    gs[:, 1] = dR_dsr1 * in_s[:, 1] * (-1.0 / st.s_ub[:, 1])
    gr = (dR_drr * in_r * (-1.0 / Dsum)).unsqueeze(1) * notleast.float()
    gg = None
    if g is not None:
        A = st.total_macs[:, 0] * s_ratio[:, 0] + st.total_macs[:, 1] * s_ratio[:, 0] \
            + st.total_macs[:, 2] * r_ratio + st.total_macs[:, 3] * r_ratio \
            + st.total_macs[:, 4] * s_ratio[:, 1] + st.total_macs[:, 5] * s_ratio[:, 1]
        gg = (c * A).unsqueeze(1) * dd1_dg
    return R, gs, gr, gg


# ----------------------------------------------------------------------------- one update
def uvc_update(st: UvcState, hp: UvcHyper, W1: List[torch.Tensor], W3: List[torch.Tensor], lr: float,
               g: Optional[torch.Tensor], g_grad: Optional[torch.Tensor], e1: Optional[torch.Tensor],
               e2: Optional[torch.Tensor], enable_warmup: int, global_step: int):
    """uvc_optimizer (uvc_optimizer.py:37-144).  Mutates st, W1/W3 (prox) and g (gating SGD) in
    place.  ``g`` is block_skip_gating.data, ``g_grad`` its task-loss gradient (None in warm-up).
    e1/e2 are the Exp(1) draws [L,2] of the two resource evaluations (srloss2, zloss)."""
    L, H = st.L, st.H
    s_max = (st.s_ub - 1 - 1e-8).clamp(min=0.0)                    # :38-39
    r_max = (st.r_ub - 1 - 1e-8).clamp(min=0.0)
    prox_w(st, W1, W3, lr)                                         # :42
    post = [scores_w1(W, st.H, st.hd) for W in W1]
    s1p = [a for a, _ in post]
    s2p = [b for _, b in post]
    s3p = [scores_w3(W) for W in W3]
    cs, cr = st.s.ceil(), st.r.ceil()
    gate = g if hp.enable_block_gating else None
    R, gs2, gr2, gg2 = resource(st, s2p, gate, e1, hp, hard=False, want_grad=True)
    diff = R - hp.budget
    cur_resource = diff.item() + hp.budget                          # :49
    inside = float(-hp.z_grad_clip <= diff.item() <= hp.z_grad_clip)   # clamp grad, :50
    if enable_warmup:
        return cur_resource                                         # :52-58
    # ---- primal gradients (:63-87)
    gs1 = torch.zeros(L, 2)
    gr1 = torch.zeros(L, H)
    for l in range(L):
        gs1[l, 0] = st.y[l, 0] * least_sum_and_next(s2p[l], int(cs[l, 0].item()))[1]   # uvc_utils.py:184,189
        gs1[l, 1] = st.y[l, 1] * least_sum_and_next(s3p[l], int(cs[l, 1].item()))[1]   # :194,199
        for h in range(H):
            gr1[l, h] = st.p[l, h] * least_sum_and_next(s1p[l][h], int(cr[l, h].item()))[1]   # :212,215
    gs1 = gs1 + hp.sl2wd * (st.s / st.s_ub)
    gr1 = gr1 + hp.sl2wd * (st.r / st.r_ub)
    s_grad = gs1 + st.z * (gs2 * inside)
    r_grad = gr1 + st.z * (gr2 * inside)
    # ---- gating (:89-98)
    if gate is not None:
        gg = g_grad + st.z * hp.gating_weight * (gg2 * inside)
        st.gating_grad_list.append(gg.unsqueeze(0) * (global_step % hp.gating_interval))
        if (global_step + 1) % hp.gating_interval == 0:
            grad = torch.cat(st.gating_grad_list).mean(0)
            d_p = grad + 1e-4 * g                                   # SGD wd 1e-4 (uvc_optimizer.py:252-255)
            if st.gate_momentum is None:
                st.gate_momentum = d_p.clone()
            else:
                st.gate_momentum.mul_(0.9).add_(d_p)
            g.add_(st.gate_momentum, alpha=-hp.glr)
            st.gating_grad_list = []
    # ---- box-projected SGD on s (:100-110) and r (:113-123)
    for var, grad, vmax, lr_ in ((st.s, s_grad, s_max, hp.slr), (st.r, r_grad, r_max, hp.rlr)):
        over = var >= vmax
        under = var <= 0
        grad[over] = grad[over].clamp(min=0.0)
        grad[under] = grad[under].clamp(max=0.0)
        total = grad.abs().max()
        coef = (1.0 / (total + 1e-6)).clamp(max=1.0)                # clip_grad_norm_(.,1.0,inf)
        grad = grad * coef
        var.add_(grad, alpha=-lr_)
        var.clamp_(min=0.0)
        var[over] = vmax[over]
    # ---- dual ascent (:126-135) with the UPDATED s, r (uvc_utils.py:231-269)
    cs, cr = st.s.ceil(), st.r.ceil()
    ns = torch.zeros(L, 2)
    nr = torch.zeros(L, H)
    for l in range(L):
        ns[l, 0] = least_sum_and_next(s2p[l], int(cs[l, 0].item()))[0] if cs[l, 0] > 0 else 0.0
        ns[l, 1] = least_sum_and_next(s3p[l], int(cs[l, 1].item()))[0] if cs[l, 1] > 0 else 0.0
        for h in range(H):
            nr[l, h] = least_sum_and_next(s1p[l][h], int(cr[l, h].item()))[0] if cr[l, h] > 0 else 0.0
    R2 = resource(st, s2p, gate, e2, hp, hard=False)
    st.y.add_(ns, alpha=hp.ylr)
    st.p.add_(nr, alpha=hp.plr)
    st.z.add_(R2 - hp.budget, alpha=hp.zlr)
    st.y.clamp_(min=0.0)                                            # proj_dual, uvc_utils.py:403-406
    st.p.clamp_(min=0.0)
    st.z.clamp_(min=0.0)
    return cur_resource
