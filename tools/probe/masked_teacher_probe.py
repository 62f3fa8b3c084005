"""Probe (r6): the teacher's forward a batch AHEAD on its OWN CUs for the whole step -- the teacher's stream pinned to TEACHER_CUS CUs by a CU mask
(uvc_stream_create_masked: an equal share of every XCD), the student's two streams pinned to the other 256 - TEACHER_CUS, every persistent kernel sized for its
share (uvc_set_cu_budget) -- against the product's schedule (teacher and student alternate whole-chip kernels; the next batch's teacher behind the backward).
The r5 version of this probe had the budgets but no masks: workgroups of the two shares landed on each other's CUs and the step LOST 0.15 ms.
    TEACHER_CUS=64 python tools/probe/masked_teacher_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from uvc_amd import _lib as L  # noqa: E402
from uvc_amd.stage1 import Stage1Trainer, default_args  # noqa: E402

model = os.environ.get("STEP_MODEL", "deit_tiny_patch16_224")
batch = int(os.environ.get("STEP_BATCH", "512"))
shares = [int(v) for v in os.environ.get("TEACHER_CUS", "64").split(",")]
a = default_args(model_type=model, precision="bf16", train_batch_size=batch, local_rank=0)
tr = Stage1Trainer(a, device="cuda:0", distributed=False)
bench.pruned_state(tr)
tr.begin_epoch(a.warmup_epochs + 1)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(batch, 3, a.img_size, a.img_size, device="cuda", generator=g)
y = torch.softmax(torch.randn(batch, a.num_classes, device="cuda", generator=g), -1)
crit = tr.criterion
dev = x.device


def timed(fn, n=60, warm=15):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        out = fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, float(out["loss"])


pending = []


def ahead():
    # the teacher's forward for the NEXT step's batch starts now (on the teacher's own stream); this step's loss takes the one started a step ago
    crit.prefetch(x)
    pending.append(crit._pref)
    crit._pref = pending.pop(0) if len(pending) > 1 else None
    a.overlap_teacher = 0
    return tr.step(x, y)


def product():
    return tr.step(x, y, next_x=x)


plain_side, plain_wgrad = crit._side, tr.model._wgrad_stream
for r in range(2):
    a.overlap_teacher = 1
    tr.model.cu_budget = 0; crit.teacher_model.cu_budget = 0
    crit._side, tr.model._wgrad_stream = plain_side, plain_wgrad
    pending.clear(); crit._pref = None
    ms, loss = timed(product)
    plain_side, plain_wgrad = crit._side, tr.model._wgrad_stream
    print("product (whole-chip kernels alternate; next batch's teacher behind the backward): %.3f ms  loss %.5f" % (ms, loss), flush=True)
    for tc in shares:
        sc = 256 - tc
        t_stream = L.masked_stream(dev, 0, tc)
        s_main, s_side = L.masked_stream(dev, tc, sc), L.masked_stream(dev, tc, sc)
        crit.teacher_model.cu_budget = tc
        tr.model.cu_budget = sc
        crit._side, tr.model._wgrad_stream = t_stream, s_side
        pending.clear(); crit._pref = None
        s_main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_main):
            ms, loss = timed(ahead)
        torch.cuda.current_stream().wait_stream(s_main)
        torch.cuda.synchronize()
        print("teacher a batch ahead pinned to %3d CUs, student pinned to the other %3d: %.3f ms  loss %.5f" % (tc, sc, ms, loss), flush=True)
        # the same shares WITHOUT masks (r5's experiment): budgets only
        crit._side, tr.model._wgrad_stream = plain_side, plain_wgrad
        pending.clear(); crit._pref = None
        ms, loss = timed(ahead)
        print("teacher a batch ahead on %3d CUs, student on %3d, budgets only (no masks): %.3f ms  loss %.5f" % (tc, sc, ms, loss), flush=True)
