"""Fused global-norm clip + AdamW over the model's flat parameter buffer (mirror of
``torch.nn.utils.clip_grad_norm_`` + ``torch.optim.AdamW`` as used at joint_train.py:271,428-429).

``FusedAdamW`` is a ``torch.optim.Optimizer`` (so LambdaLR schedulers and ``param_groups[0]['lr']``
readers such as prox_w work unchanged); its ``step()`` is two HIP launches over the always-active
segment plus tiny launches for conditionally-active tensors (block_skip_gating after warm-up,
gumbel.* / patch_gating with patch gating), each with its own step count like torch's per-parameter
state.  Parameters whose gradient is None are skipped entirely (no decay), as in torch.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import ops


def clip_grad_norm_(model_or_params, max_norm: float):
    """Drop-in for ``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)`` on a uvc_amd model:
    computes the global L2 norm of all live gradients on the device and ARMS the clip; the scaling is
    applied inside ``FusedAdamW.step()`` (same arithmetic, one pass less over the gradients), which also
    rescales block_skip_gating.grad in place because uvc_optimizer reads it afterwards.  Returns the
    (lazy, device) total norm.

    LIFETIME of the returned tensor: it is a 0-dim VIEW into a ring of 64 slots, not a fresh tensor as torch returns -- it holds this
    call's norm until the 64th later call of clip_grad_norm_ on the same model overwrites the slot.  A caller that keeps norms longer
    (a per-epoch log read lazily) takes ``float(norm)`` or ``norm.clone()``.  The trainers' ``step()`` return this view as ``gnorm``."""
    model = model_or_params if hasattr(model_or_params, "_flat_grad") else getattr(model_or_params, "_uvc_model", None)
    if model is None:
        raise L.UvcHipError("clip_grad_norm_: pass the uvc_amd model (or the generator from model.parameters() wrapped by FusedAdamW)")
    segs = _live_segments(model)
    st = _clip_state(model)
    st["slot"] = (st["slot"] + 1) % st["ring"].shape[0]
    st["sq"] = st["ring"][st["slot"]]              # [sum of squares, its square root]: a fresh slot per call, so the returned norm stays valid
    first = True                                    # for the next 63 steps without a copy (the reference returns a new tensor per call)
    for off, n in segs:
        ops.grad_sqnorm(model._flat_grad[off:off + n], st["partial"], st["sq"], accumulate=not first)
        first = False
    st["armed"] = float(max_norm)
    return st["sq"][1]                              # written by the reduction kernel: no sqrt launch


def _clip_state(model):
    if not hasattr(model, "_clip"):
        dev = model._flat.device
        ring = torch.zeros(64, 2, device=dev)
        model._clip = dict(partial=torch.empty(1024, device=dev), ring=ring, slot=0, sq=ring[0], armed=None,
                           gnorm=torch.zeros(1, device=dev), zero=torch.zeros(2, device=dev))
    return model._clip


def _small_tensors(model):
    """(name, parameter, offset) of the conditionally-active tensors."""
    o = model._off
    out = [("gate", model.block_skip_gating, o.gate)]
    if getattr(model, "gumbel", None) is not None:
        out += [("gumbel_w", model.gumbel.weight, o.gumbel_w), ("gumbel_b", model.gumbel.bias, o.gumbel_b)]
    if model.patch_gating is not None:
        out.append(("patch_gating", model.patch_gating, o.patch_gating))
    return out


def _live_segments(model):
    segs = [(0, model._off.n_main)] + list(model._extra_live_segments())
    for _, p, off in _small_tensors(model):
        if p.grad is not None:
            segs.append((off, p.numel()))
    return segs


NO_DECAY, DECAY, FROZEN = 0, 1, 2      # uvc_adamw_args.flags bits


class FusedAdamW(torch.optim.Optimizer):
    """``filter_bias_and_bn=True`` reproduces timm's ``add_weight_decay`` grouping used by Stage-2
    (post_train.py:299 -> timm.optim.create_optimizer): 1-D tensors, ``*.bias`` and the names in
    ``model.no_weight_decay()`` get no decay.  Parameters of hard-skipped blocks (no gradient) are frozen
    through the same per-element flag array; without either, no flag array is used."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, max_grad_norm=None,
                 filter_bias_and_bn=False):
        if not hasattr(model, "_flat"):
            raise L.UvcHipError("FusedAdamW needs a uvc_amd DistilledVisionTransformer (flat parameter buffer)")
        model._check_flat()
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.max_grad_norm = max_grad_norm
        dev = model._flat.device
        n = model.n_flat
        self.exp_avg = torch.zeros(n, device=dev)
        self.exp_avg_sq = torch.zeros(n, device=dev)
        self.steps = {"main": 0}
        self.filter_bias_and_bn = bool(filter_bias_and_bn and weight_decay)
        self._flags = None
        self._flags_key = None

    def _element_flags(self):
        """uint8 flag per element of the flat buffer (or None when every element decays and trains)."""
        m = self.model
        skipped = tuple(m.skipped_block_ranges()) + tuple(m._frozen_ranges())
        if not self.filter_bias_and_bn and not skipped:
            return None
        if self._flags is None or self._flags_key != skipped:
            fl = torch.full((m.n_flat,), DECAY, dtype=torch.uint8)
            if self.filter_bias_and_bn:
                skip = m.no_weight_decay() if hasattr(m, "no_weight_decay") else set()
                off_of = {id(p): off for p, off in m._slots()}
                for name, p in m.named_parameters():
                    if (p.ndim == 1 or name.endswith(".bias") or name in skip) and id(p) in off_of:
                        off = off_of[id(p)]
                        fl[off:off + p.numel()] = NO_DECAY
            for off, k in skipped:
                fl[off:off + k] |= FROZEN
            self._flags, self._flags_key = fl.to(m._flat.device), skipped
        return self._flags

    def zero_grad(self, set_to_none: bool = True):
        """Gradients are overwritten by the next backward (beta = 0), so nothing has to be cleared; keeps
        torch's contract that conditionally-active tensors read None again."""
        m = self.model
        if m.grad_accumulate:
            m._flat_grad.zero_()
        for _, p, _ in _small_tensors(m):
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        m = self.model
        g = self.param_groups[0]
        st = _clip_state(m)
        max_norm = st["armed"] if st["armed"] is not None else self.max_grad_norm
        if max_norm is None:
            max_norm = float("inf")
            st["sq"] = st["zero"]                  # a scratch slot of its own: the ring slot a caller may still hold a norm in is not touched
            st["sq"].zero_()
        elif st["armed"] is None:
            clip_grad_norm_(m, max_norm)
        st["armed"] = None
        b1, b2 = g["betas"]
        common = dict(lr=float(g["lr"]), beta1=b1, beta2=b2, eps=g["eps"], weight_decay=g["weight_decay"], max_norm=float(max_norm))
        self.steps["main"] += 1
        n = m._off.n_main
        flags = self._element_flags()
        ops.adamw_step(m._flat[:n], m._flat_grad[:n], self.exp_avg[:n], self.exp_avg_sq[:n], st["sq"], step=self.steps["main"],
                       gnorm_out=st["gnorm"], flags=flags[:n] if flags is not None else None, **common)
        for off, k in m._extra_live_segments():
            ops.adamw_step(m._flat[off:off + k], m._flat_grad[off:off + k], self.exp_avg[off:off + k], self.exp_avg_sq[off:off + k], st["sq"],
                           step=self.steps["main"], flags=flags[off:off + k] if flags is not None else None, **common)
        for name, p, off in _small_tensors(m):
            if p.grad is None:
                continue
            self.steps[name] = self.steps.get(name, 0) + 1
            k = p.numel()
            ops.adamw_step(m._flat[off:off + k], m._flat_grad[off:off + k], self.exp_avg[off:off + k],
                           self.exp_avg_sq[off:off + k], st["sq"], step=self.steps[name],
                           flags=flags[off:off + k] if flags is not None else None, **common)
            if name == "gate":
                m._run_block_host = None            # the hard-skip decision reads these logits
            if name == "gate" and max_norm != float("inf"):
                ops.scale_by_clip(m._flat_grad[off:off + k], st["sq"], float(max_norm))   # uvc_optimizer.py:90 reads the clipped grad
        m.mark_weights_changed()
        return None


def create_optimizer(args, model, filter_bias_and_bn=True):
    """timm.optim.create_optimizer(args, model) as Stage-2 calls it (post_train.py:299; --opt adamw, :456-467): AdamW
    with lr=args.lr, eps=args.opt_eps, betas=args.opt_betas (if given), weight decay only on the >=2-D weights."""
    opt = str(getattr(args, "opt", "adamw")).lower()
    if opt != "adamw":
        raise NotImplementedError(f"--opt {opt}: the HIP optimizer is AdamW (the reference's Stage-2 default)")
    kw = dict(lr=args.lr, weight_decay=args.weight_decay)
    if getattr(args, "opt_eps", None) is not None:
        kw["eps"] = args.opt_eps
    if getattr(args, "opt_betas", None) is not None:
        kw["betas"] = tuple(args.opt_betas)
    return FusedAdamW(model, filter_bias_and_bn=filter_bias_and_bn, **kw)
