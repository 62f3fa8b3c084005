"""MI355X mirror of the reference's ``UVC/uvc_utils.py``: same names, argument meaning and error
behaviour, but every tensor op is a HIP kernel of libuvc_hip.so (include/uvc_engine.h) and the
whole primal/dual state stays on the device.

Replaces (reference file:line):
  UVC_CP_MiniMax            uvc_utils.py:129-308
  weight_list_to_scores     uvc_utils.py:54-73     -> uvc_scores   (one pass over all layers)
  prox_w                    uvc_utils.py:315-345   -> uvc_prox
  prune_w_mask              uvc_utils.py:376-401   -> uvc_write_masks
  proj_dual                 uvc_utils.py:403-406   (folded into uvc_dual_step; kept as a function)
  calc_flops/run_resource_fn uvc_utils.py:220-229,409-471 -> uvc_resource / uvc_dual_step
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch
from torch import nn
from torch.nn import Parameter

from . import _lib as L


class DeviceValue:
    """Lazy host view of a device tensor: converts (and synchronises) only when read, so the
    per-step ``.cpu().numpy()`` of the reference (uvc_optimizer.py:138-144) leaves the step path."""

    def __init__(self, t: torch.Tensor, scalar: bool = False):
        self._t = t
        self._scalar = scalar
        self._np = None

    def numpy(self):
        if self._np is None:
            self._np = self._t.detach().cpu().numpy()
        return self._np

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def tolist(self):
        return self.numpy().tolist()

    def __getitem__(self, i):
        return self.numpy()[i]

    def __len__(self):
        return len(self.numpy())

    def __float__(self):
        return float(self.numpy().reshape(-1)[0])

    def item(self):
        return float(self)

    def __format__(self, spec):
        return format(float(self), spec) if self._scalar else format(self.numpy(), spec)

    def __repr__(self):
        return repr(float(self)) if self._scalar else repr(self.numpy())

    def __mul__(self, o):
        return float(self) * o if self._scalar else self.numpy() * o

    __rmul__ = __mul__

    def __add__(self, o):
        return float(self) + o if self._scalar else self.numpy() + o

    __radd__ = __add__

    @property
    def shape(self):
        return tuple(self._t.shape)


def _ptr_table(tensors, device):
    return torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=device)


class UVC_CP_MiniMax(nn.Module):
    """Same constructor and attributes as the reference (uvc_utils.py:129-169).  ``resource_fn`` is
    accepted for signature compatibility and ignored: the FLOPs model runs in uvc_resource /
    uvc_dual_step (flops_with_mhsa=1 path, uvc_utils.py:409-462)."""

    def __init__(self, model, resource_fn, uvc_layers, uvc_layers_dict, head_size, num_heads, flops_list,
                 z_init=1e-3, y_init=1e-3, p_init=1e-3, args=None):
        super().__init__()
        self.model = model
        self.uvc_layers = uvc_layers
        self.uvc_layers_dict = uvc_layers_dict
        self.head_size = head_size
        n_layers = len(self.uvc_layers["W1"])
        self.n_layers = n_layers
        self.eps_decay = args.eps_decay
        self.num_heads = num_heads
        self.flops_list = flops_list
        self.args = args
        w1 = self.uvc_layers["W1"][0].weight
        L.require_cuda(w1)
        dev = w1.device
        self.device_ = dev
        D = self.uvc_layers["W1"][0].in_features
        F = self.uvc_layers["W3"][0].in_features
        if D != num_heads * head_size:
            raise ValueError(f"attn.proj in_features {D} != num_heads*head_size {num_heads}*{head_size}")
        self.dims = L.uvc_dims(n_layers, num_heads, head_size, D, F)
        H = num_heads
        # one contiguous block: s[L,2] r[L,H] y[L,2] p[L,H] z[1] out[4]  (a single async D2H -- and the per-step snapshot, one copy -- reads all;
        # out = what uvc_dual_step reports: [0] the current resource)
        n = n_layers * (4 + 2 * H) + 1
        self._flat = torch.zeros(n + 4, device=dev, dtype=torch.float32)
        self._out = self._flat[n:n + 4]
        o = 0

        def take(shape):
            nonlocal o
            k = int(np.prod(shape)) if shape else 1
            v = self._flat[o:o + k].view(shape)
            o += k
            return v

        self.s = Parameter(take((n_layers, 2)))
        self.r = Parameter(take((n_layers, H)))
        self.y = Parameter(take((n_layers, 2)))
        self.p = Parameter(take((n_layers, H)))
        self.z = Parameter(take(()))
        self.y.data.fill_(y_init)
        self.p.data.fill_(p_init)
        self.z.data.fill_(float(z_init))
        self.resource_fn = resource_fn

        self.enable_patch_gating = args.enable_patch_gating
        self.patch_gating = Parameter(3 * torch.ones(1, model.patch_embed.grid_size[0] * model.patch_embed.grid_size[1], 1, device=dev)) \
            if self.enable_patch_gating == 1 else None
        self.update_patch()
        self.enable_part_gating = args.enable_part_gating
        self.enable_block_gating = args.enable_block_gating
        self.update_gating()

        self.s_ub = torch.zeros_like(self.s.data)
        self.s_ub[:, 0] = num_heads
        self.s_ub[:, 1] = F
        self.r_ub = torch.zeros_like(self.r.data)
        self.r_ub[:, :] = head_size

        # FLOPs table (uvc_utils.py:410-413; uvc_optimizer.py:177-189)
        embed_macs, total_macs = flops_list
        self.total_macs = torch.tensor(np.asarray(total_macs, dtype=np.float32), device=dev)
        self.embed_macs = float(embed_macs)
        tm = torch.tensor(np.asarray(total_macs, dtype=np.float32))      # float32 sum, as torch.Tensor(list).sum()
        self.resource_ub = float((float(embed_macs) + tm.sum()) * 2)

        # scratch of the score/rank kernels
        f32 = dict(device=dev, dtype=torch.float32)
        i32 = dict(device=dev, dtype=torch.int32)
        self._ws64 = torch.empty(n_layers * (D + F), device=dev, dtype=torch.float64)
        self._sc = [torch.empty(n_layers, D, **f32), torch.empty(n_layers, H, **f32), torch.empty(n_layers, F, **f32)]
        self._rk = [torch.empty(n_layers, D, **i32), torch.empty(n_layers, H, **i32), torch.empty(n_layers, F, **i32)]
        self._gate_momentum = torch.zeros(n_layers, 2, **f32)
        self._gate_gsum = torch.zeros(n_layers, 2, **f32)
        self._gate_counters = torch.zeros(2, **i32)
        self._res_out = torch.zeros(1, **f32)
        self._tables = None
        self._table_key = None
        self.gating_list_len = 0
        # source of the Exp(1) draws behind the Gumbel noise; tests inject the reference's draws here
        self.exp_source = lambda shape: torch.empty(shape, device=dev, dtype=torch.float32).exponential_()

    # -- reference API ---------------------------------------------------------------------
    def ceiled_s(self):
        return self.s.detach().ceil()

    def ceiled_r(self):
        return self.r.detach().ceil()

    def update_gating(self):                                       # uvc_utils.py:273-284
        self.block_skip_gating = self.model.block_skip_gating if self.enable_block_gating else None
        self.attn_skip_gating = [] if self.enable_part_gating else None
        self.mlp_skip_gating = [] if self.enable_part_gating else None
        if self.enable_part_gating:
            for name, p in self.model.named_parameters():
                if "attn_skip_gating" in name:
                    self.attn_skip_gating.append(p)
                if "mlp_skip_gating" in name:
                    self.mlp_skip_gating.append(p)

    def update_patch(self):                                        # :286-288
        if self.enable_patch_gating == 1:
            self.model.patch_gating = self.patch_gating

    def update_eps(self):                                          # :290-293
        if not self.model.enable_warmup:
            print(f"[EPS update] {self.model.eps} =====> {self.model.eps * self.eps_decay} ")
            self.model.eps = self.model.eps * self.eps_decay

    def run_resource_fn(self, gumbel_hard=False):                  # :220-224
        """FLOPs ratio at ceil(s), ceil(r) and the current gates, on the scores of the CURRENT weights."""
        self.refresh_scores()
        hp = self.hyper(self.args)
        e = None
        if self.block_skip_gating is not None and hp.use_gumbel:
            e = self.exp_source((self.n_layers, 2))
        st = self._state(None)
        L.check(L.lib().uvc_resource(C.byref(st), self.dims, hp, L.ptr(e), int(bool(gumbel_hard)),
                                     L.ptr(self._res_out), L.cur_stream()), "uvc_resource")
        return DeviceValue(self._res_out.clone(), scalar=True)

    def srloss2(self, budget):
        return float(self.run_resource_fn()) - budget

    # -- device plumbing ---------------------------------------------------------------------
    def tables(self):
        """Device pointer tables of W1/W3 weights and W1/W3/W2 masks (rebuilt if a tensor moved)."""
        w1 = [m.weight for m in self.uvc_layers["W1"]]
        w3 = [m.weight for m in self.uvc_layers["W3"]]
        key = tuple(t.data_ptr() for t in w1 + w3)
        if key != self._table_key:
            for t in w1 + w3:
                L.require_cuda(t)
                if not t.is_contiguous() or t.dtype != torch.float32:
                    raise L.UvcHipError("UVC layers must be contiguous float32 weights")
            self._tables = (_ptr_table(w1, self.device_), _ptr_table(w3, self.device_))
            self._table_key = key
        return self._tables

    def hyper(self, args, s_opt=None, r_opt=None, g_opt=None, dual_opt=None, z_grad_clip=None,
              gating_interval=None) -> "L.uvc_hyper":
        hp = L.uvc_hyper()
        hp.budget = float(args.budget)
        hp.slr = float(s_opt.param_groups[0]["lr"]) if s_opt is not None else float(getattr(args, "slr", 0.02))
        hp.rlr = float(r_opt.param_groups[0]["lr"]) if r_opt is not None else float(getattr(args, "rlr", 0.02))
        hp.glr = float(g_opt.param_groups[0]["lr"]) if g_opt is not None else float(getattr(args, "glr", 1e-3))
        if dual_opt is not None:                                   # uvc_optimizer.py:261-266
            hp.zlr = float(dual_opt.param_groups[0]["lr"])
            hp.ylr = float(dual_opt.param_groups[1]["lr"])
            hp.plr = float(dual_opt.param_groups[2]["lr"])
        else:
            hp.zlr, hp.ylr, hp.plr = 1.0, float(getattr(args, "ylr", 1e-4)), float(getattr(args, "plr", 1e-4))
        hp.sl2wd = float(getattr(args, "sl2wd", 0.0))
        hp.z_grad_clip = float(z_grad_clip if z_grad_clip is not None else getattr(args, "z_grad_clip", 0.5))
        hp.gating_weight = float(getattr(args, "gating_weight", 5.0))
        hp.eps = float(self.model.eps)
        hp.gating_interval = int(gating_interval if gating_interval is not None else getattr(args, "gating_interval", 100))
        hp.use_gumbel = int(getattr(args, "use_gumbel", 1))
        hp.enable_block_gating = int(bool(self.enable_block_gating))
        return hp

    def _state(self, gate_grad: Optional[torch.Tensor]) -> "L.uvc_state":
        st = L.uvc_state()
        st.s, st.r, st.y, st.p, st.z = (L.ptr(self.s.data), L.ptr(self.r.data), L.ptr(self.y.data),
                                        L.ptr(self.p.data), L.ptr(self.z.data))
        g = self.block_skip_gating
        if g is not None:
            L.require_cuda(g)
            if not g.is_contiguous():
                raise L.UvcHipError("block_skip_gating must be contiguous")
        st.gate = L.ptr(g.data) if g is not None else None
        st.gate_grad = L.ptr(gate_grad) if gate_grad is not None else None
        st.gate_momentum = L.ptr(self._gate_momentum)
        st.gate_gsum = L.ptr(self._gate_gsum)
        st.gate_counters = L.ptr(self._gate_counters)
        st.total_macs = L.ptr(self.total_macs)
        st.embed_macs = self.embed_macs
        st.resource_ub = self.resource_ub
        st.scores1, st.scores2, st.scores3 = (L.ptr(t) for t in self._sc)
        st.rank1, st.rankh, st.rank3 = (L.ptr(t) for t in self._rk)
        st.out = L.ptr(self._out)
        return st

    def refresh_scores(self):
        """scores + ranks of the current weights (weight_list_to_scores + topk order)."""
        t1, t3 = self.tables()
        lib, stream = L.lib(), L.cur_stream()
        L.check(lib.uvc_scores(L.ptr(t1), L.ptr(t3), self.dims, L.ptr(self._ws64), *(L.ptr(t) for t in self._sc),
                               stream), "uvc_scores")
        L.check(lib.uvc_rank(*(L.ptr(t) for t in self._sc), self.dims, *(L.ptr(t) for t in self._rk), stream),
                "uvc_rank")

    def reset_gating_list(self):
        """gating_grad_list = [] at every epoch start (joint_train.py:337)."""
        self._gate_gsum.zero_()
        self._gate_counters[0:1].zero_()
        self.gating_list_len = 0


def weight_list_to_scores(layer, layer_group_name, head_size=None):
    """uvc_utils.py:54-73 for ONE layer (diagnostics/tests; the step path scores all layers in one
    launch through UVC_CP_MiniMax.refresh_scores)."""
    W = layer.weight.data
    L.require_cuda(W)
    dev = W.device
    Dout = W.shape[0]
    if layer_group_name == "W1":
        D = W.shape[1]
        H = D // head_size
        dummy = torch.zeros(Dout, 64, device=dev)
        dims = L.uvc_dims(1, H, head_size, D, 64)
        if Dout != D:
            raise ValueError("W1 must be square")
        t1, t3 = _ptr_table([W], dev), _ptr_table([dummy], dev)
    elif layer_group_name == "W3":
        F = W.shape[1]
        dummy = torch.zeros(Dout, Dout, device=dev)
        hs = 64 if Dout % 64 == 0 else Dout
        dims = L.uvc_dims(1, Dout // hs, hs, Dout, F)
        t1, t3 = _ptr_table([dummy], dev), _ptr_table([W], dev)
    else:
        raise ValueError(layer_group_name)
    ws = torch.empty(dims.D + dims.F, device=dev, dtype=torch.float64)
    s1 = torch.empty(dims.D, device=dev)
    s2 = torch.empty(dims.H, device=dev)
    s3 = torch.empty(dims.F, device=dev)
    L.check(L.lib().uvc_scores(L.ptr(t1), L.ptr(t3), dims, L.ptr(ws), L.ptr(s1), L.ptr(s2), L.ptr(s3),
                               L.cur_stream()), "uvc_scores")
    if layer_group_name == "W1":
        return s1.view(dims.H, head_size), s2
    return s3


def prox_w(minimax_model: UVC_CP_MiniMax, optimizer):
    """uvc_utils.py:315-345.  In place on attn.proj.weight / mlp.fc2.weight (the tensors AdamW and
    the gradient all-reduce hold).  Leaves the scores and ranks of the SHRUNK weights in the
    minimax object (the reference recomputes them in sloss1/rloss1/calc_flops)."""
    mm = minimax_model
    lr = float(optimizer.param_groups[0]["lr"])
    mm.refresh_scores()
    t1, t3 = mm.tables()
    lib, stream = L.lib(), L.cur_stream()
    L.check(lib.uvc_prox(L.ptr(t1), L.ptr(t3), mm.dims, *(L.ptr(t) for t in mm._rk), L.ptr(mm.s.data), L.ptr(mm.r.data),
                         L.ptr(mm.y.data), L.ptr(mm.p.data), lr, L.ptr(mm._ws64), *(L.ptr(t) for t in mm._sc), stream),
            "uvc_prox")
    L.check(lib.uvc_rank(*(L.ptr(t) for t in mm._sc), mm.dims, *(L.ptr(t) for t in mm._rk), stream), "uvc_rank")


def prune_w_mask(minimax_model: UVC_CP_MiniMax, optimizer=None):
    """uvc_utils.py:376-401: writes attn.proj.mask, mlp.fc2.mask and mlp.fc1.mask."""
    mm = minimax_model
    dev = mm.device_
    masks = []
    for grp in ("W1", "W3", "W2"):
        ms = [m.mask for m in mm.uvc_layers[grp]]
        for t in ms:
            L.require_cuda(t)
            if not t.is_contiguous() or t.dtype != torch.float32:
                raise L.UvcHipError("mask buffers must be contiguous float32")
        masks.append(_ptr_table(ms, dev))
    mm.refresh_scores()
    L.check(L.lib().uvc_write_masks(L.ptr(masks[0]), L.ptr(masks[1]), L.ptr(masks[2]), mm.dims,
                                    *(L.ptr(t) for t in mm._rk), L.ptr(mm.s.data), L.ptr(mm.r.data), L.cur_stream()),
            "uvc_write_masks")


def proj_dual(minimax_model):
    """uvc_utils.py:403-406 (uvc_dual_step already projects; kept for callers that use it directly)."""
    minimax_model.y.data.clamp_(min=0.0)
    minimax_model.p.data.clamp_(min=0.0)
    minimax_model.z.data.clamp_(min=0.0)
