"""MI355X mirror of the reference's ``UVC/uvc_optimizer.py``: ``build_minimax_model`` and
``uvc_optimizer`` keep the reference's signatures (uvc_optimizer.py:164,37) so a
``joint_train.py``-style driver calls them unchanged; the work is four HIP launches
(scores -> rank -> prox+scores -> rank) plus one single-workgroup scalar kernel, with no
device->host synchronisation on the step path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .uvc_utils import DeviceValue, UVC_CP_MiniMax, prox_w


def uvc_optimizer(optimizer, minimax_model, s_optimizer, r_optimizer, gating_optimizer, dual_optimizer, args, infos,
                  save_budgets, flops_list, z_grad_clip, global_step, gating_interval, gating_grad_list):
    """One primal-dual update (uvc_optimizer.py:37-144).  Returns
    ``(cur_resource, s, r, gating, gating_grad_list)`` like the reference; the first four are lazy
    host views (DeviceValue) that synchronise only when read."""
    mm: UVC_CP_MiniMax = minimax_model
    if len(gating_grad_list) == 0 and mm.gating_list_len != 0:
        mm.reset_gating_list()                      # the driver cleared the list (joint_train.py:337)
    prox_w(mm, optimizer)                           # :42 (+ scores/ranks of the shrunk weights)
    warm = int(bool(mm.model.enable_warmup))
    hp = mm.hyper(args, s_optimizer, r_optimizer, gating_optimizer, dual_optimizer, z_grad_clip, gating_interval)
    gate = mm.block_skip_gating
    gate_grad = None
    if gate is not None and not warm:
        if gating_optimizer is None:
            hp.enable_block_gating = 0
        else:
            gate_grad = gate.grad
            if gate_grad is None:
                raise L.UvcHipError("block_skip_gating.grad is None outside warm-up (uvc_optimizer.py:90 would raise too)")
            L.require_cuda(gate_grad)
            gate_grad = gate_grad.contiguous()
    e1 = e2 = None
    if gate is not None and hp.use_gumbel:          # Gumbel draws of srloss2 (:48) and zloss (:126)
        e1 = mm.exp_source((mm.n_layers, 2))
        if not warm:
            e2 = mm.exp_source((mm.n_layers, 2))
    st = mm._state(gate_grad)
    L.check(L.lib().uvc_dual_step(C.byref(st), mm.dims, hp, L.ptr(e1), L.ptr(e2), warm, int(global_step),
                                  L.cur_stream()), "uvc_dual_step")
    if not warm and gate is not None and gating_optimizer is not None:
        gating_grad_list.append(None)               # length mirrors the reference's list
        if (global_step + 1) % gating_interval == 0:
            gating_grad_list = []
        mm.gating_list_len = len(gating_grad_list)
    snap = mm._flat.clone()                         # s,r,y,p,z and the reported resource of THIS step (one device-side copy, no sync)
    n2, nH = mm.n_layers * 2, mm.n_layers * mm.num_heads
    cur = DeviceValue(snap[snap.numel() - 4:snap.numel() - 3], scalar=True)
    s_v = DeviceValue(snap[0:n2].view(mm.n_layers, 2))
    r_v = DeviceValue(snap[n2:n2 + nH].view(mm.n_layers, mm.num_heads))
    g_v = DeviceValue(gate.detach().clone()) if gate is not None else None
    return cur, s_v, r_v, g_v, gating_grad_list


def uvc_optimizer_gating(*a, **k):
    """The reference's enable_pruning=0 path raises TypeError (10 parameters called with 14,
    uvc_optimizer.py:148 vs joint_train.py:444).  Not a working configuration."""
    raise TypeError("uvc_optimizer_gating() takes 10 positional arguments but 14 were given "
                    "(enable_pruning=0 is broken in the reference as shipped)")


def build_minimax_model(model, layer_names, uvc_layers, uvc_layers_dict, args, flops_list, vanilla=False):
    """uvc_optimizer.py:164-268: the minimax object + the four small optimisers.  The optimisers are
    real torch.optim.SGD objects (so PresetLRScheduler and drivers can poke param_groups); their
    learning rates are read by uvc_optimizer, the update itself runs in uvc_dual_step."""
    if not args.flops_with_mhsa:
        raise NotImplementedError("flops_with_mhsa=0 (legacy flops2, uvc_utils.py:95-125) is not on the hot path")
    minimax_model = UVC_CP_MiniMax(model, resource_fn=None, uvc_layers=uvc_layers, uvc_layers_dict=uvc_layers_dict,
                                   head_size=args.head_size, num_heads=args.num_heads, flops_list=flops_list, args=args)
    m = minimax_model.model                         # :204-210
    m.enable_block_gating = args.enable_block_gating
    m.enable_part_gating = args.enable_part_gating
    m.enable_patch_gating = args.enable_patch_gating
    m.enable_jumping = args.enable_jumping
    m.use_gumbel = args.use_gumbel
    m.eps = args.eps
    m.enable_warmpup = args.enable_warmup           # (sic) the reference's attribute typo is kept
    print(f"** Initial FLOP size: {minimax_model.resource_ub/1e6:.2f}M")
    if vanilla:
        return minimax_model
    for which in (args.soptim, args.roptim):
        if which != "sgd":
            raise NotImplementedError(f"soptim/roptim={which}: only 'sgd' (the README configuration) runs on the HIP path")
    s_optimizer = torch.optim.SGD([minimax_model.s], args.slr, momentum=0.0, weight_decay=0.0)
    r_optimizer = torch.optim.SGD([minimax_model.r], args.rlr, momentum=0.0, weight_decay=0.0)
    gating_optimizer = None
    if args.enable_block_gating:
        gating_optimizer = torch.optim.SGD([minimax_model.block_skip_gating], args.glr, momentum=0.9, weight_decay=1e-4)
    dual_optimizer = torch.optim.SGD([{'params': minimax_model.z, 'lr': args.zlr_schedule_list[0]},
                                      {'params': minimax_model.y, 'lr': args.ylr},
                                      {'params': minimax_model.p, 'lr': args.plr}], 1.0, momentum=0.0, weight_decay=0.0)
    return minimax_model, dual_optimizer, s_optimizer, r_optimizer, gating_optimizer
