"""Data parallelism for the Stage-1 step: one process per GPU, RCCL all-reduce (torch.distributed
backend "nccl" on ROCm) of the flat gradient buffer in reverse-layer buckets on a side stream, so
the reduction of the last blocks overlaps the backward of the first ones.

Replaces apex ``DistributedDataParallel(model, message_size=250000000,
gradient_predivide_factor=world_size, delay_allreduce=True)`` (UVC/joint_train.py:293): mean of the
gradients over ranks, parameters broadcast from rank 0 at construction.  The reference does one
flat all-reduce after backward with no overlap; here the backward sequencer is cut at bucket
boundaries (uvc_vit_io.stage_begin/end) and each finished bucket is handed to RCCL immediately.
The UVC primal/dual state is replicated (every rank runs uvc_optimizer on identical weights and
identical noise, as in the reference); the FLOPs-budget dual scalar z rides in the tail bucket
so replicas cannot drift.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class FlatGradReducer:
    """Mean all-reduce of slices of one flat buffer, bucket by bucket.  Device-agnostic (the CPU/gloo
    tests drive it with CPU tensors); on a GPU the collectives run on a dedicated stream."""

    def __init__(self, flat: torch.Tensor, buckets: Sequence[Sequence[Tuple[int, int]]], process_group=None):
        self.flat = flat
        self.buckets = [list(b) for b in buckets]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.on_gpu = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.on_gpu else None
        self.avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.pending = []
        self._exposed = []

    def launch(self, i: int):
        """Start the all-reduce of bucket i (its gradients are complete on the current stream)."""
        if self.world == 1:
            return
        if self.on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            ctx = torch.cuda.stream(self.stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for off, n in self.buckets[i]:
                t = self.flat[off:off + n]
                if self.avg:
                    self.pending.append(dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                    t.div_(self.world)

    def finish(self):
        """Make the current stream wait for every launched bucket.  On a GPU the wait is bracketed by two events on the current
        stream: their distance is the time the compute stream stood still for communication (the EXPOSED part of the all-reduce),
        read back lazily by ``exposed_ms``."""
        if self.world == 1:
            return
        cur = torch.cuda.current_stream(self.flat.device) if self.on_gpu else None
        if self.on_gpu and self.time_exposed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.on_gpu:
            cur.wait_stream(self.stream)
            if self.time_exposed:
                e1.record(cur)
                self._exposed.append((e0, e1))
                if len(self._exposed) > 256:
                    self._exposed = self._exposed[-256:]

    time_exposed = True

    def exposed_ms(self):
        """Mean exposed-communication time per finish() over the recorded window (synchronises)."""
        ev, self._exposed = self._exposed, []
        if not ev:
            return 0.0
        ev[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


BUCKET_BYTES = 32 << 20      # target bucket size when the count is derived from the model size (Base: 346 MB -> 11 buckets)


def bucket_plan(off, depth: int, n_extra: int, num_buckets: Optional[int] = 4, n_flat: Optional[int] = None):
    """Backward stage boundaries and the flat-buffer ranges that are final after each of them.
    Stages: 0 = heads + final norm, 1..L = blocks L-1..0, L+1/L+2 = embedding (uvc_vit.h).  The flat
    layout is in forward order, so a bucket is a contiguous [block k .. previous bucket) range.
    ``num_buckets=None`` derives the number of block buckets from the gradient bytes (BUCKET_BYTES each, at least 4, at most one
    per block).  The last block bucket (blocks 0..) goes out as soon as block 0 is done, i.e. BEFORE the token-assembly / patch-
    embedding backward, and only the embedding tensors + the small conditionally-trained tensors + the dual scalar slot wait for
    the end of the backward: the exposed tail is ~0.6 MB for DeiT-Tiny instead of three blocks + embeddings."""
    if num_buckets is None:
        num_buckets = max(4, -(-(4 * off.n_main) // BUCKET_BYTES))
    num_buckets = max(1, min(num_buckets, depth))
    per = -(-depth // num_buckets)
    plan = []
    hi = off.n_main
    l = depth
    while l > 0:
        lo_blk = max(0, l - per)
        stage_end = depth - lo_blk + 1           # stages < stage_end are done once block lo_blk is done
        lo = off.blk[lo_blk][0]
        plan.append((stage_end, [(lo, hi - lo)]))
        hi = lo
        l = lo_blk
    # tail: embedding + the small conditionally-active tensors (+ the dual scalar slot)
    n_flat = off.n_total if n_flat is None else n_flat      # T2T-ViT: the tokens-to-token parameters sit behind the engine's layout
    plan.append((depth + 3, [(0, hi), (off.n_main, n_flat - off.n_main + n_extra)]))
    return plan


class DistributedDataParallel(torch.nn.Module):
    """Same constructor keywords as the apex class the reference uses; ``module`` must be a uvc_amd
    DistilledVisionTransformer."""

    def __init__(self, module, message_size=250000000, gradient_predivide_factor=1.0, delay_allreduce=False,
                 num_buckets=None, process_group=None, dual_scalar: Optional[torch.Tensor] = None):
        super().__init__()
        if not hasattr(module, "_flat"):
            raise TypeError("uvc_amd.ddp.DistributedDataParallel wraps a uvc_amd DistilledVisionTransformer")
        self.module = module
        module._check_flat()
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world > 1:
            dist.broadcast(module._flat, src=0, group=process_group)       # apex DDP ctor behaviour
            module.mark_weights_changed()
        self.dual_scalar = dual_scalar
        plan = bucket_plan(module._off, module._cfg.depth, module.N_EXTRA, 1 if delay_allreduce else num_buckets, module.n_flat)
        self.stage_ends = [p[0] for p in plan]
        self.reducer = FlatGradReducer(module._flat_grad, [p[1] for p in plan], process_group)
        object.__setattr__(module, "_ddp", self)      # plain attribute: registering it as a sub-module would create a module cycle

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def exposed_comm_ms(self):
        return self.reducer.exposed_ms()

    # called by the model's backward between stages
    def pack_dual(self):
        if self.dual_scalar is not None:
            self.module._flat_grad[self.module.n_flat] = self.dual_scalar.detach()

    def unpack_dual(self):
        if self.dual_scalar is not None and self.world > 1:
            self.dual_scalar.data.copy_(self.module._flat_grad[self.module.n_flat])
