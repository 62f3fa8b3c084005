"""Probe: teacher under the student's forward as today, but each on a SHARE of the chip during the forward (backward on the whole chip).
    PYTHONPATH=. python tools/forward_shares_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uvc_amd.stage1 import Stage1Trainer, default_args  # noqa: E402

model = os.environ.get("STEP_MODEL", "deit_tiny_patch16_224")
batch = int(os.environ.get("STEP_BATCH", "512"))
a = default_args(model_type=model, precision="bf16", train_batch_size=batch, local_rank=0)
tr = Stage1Trainer(a, device="cuda:0", distributed=False)
bench.pruned_state(tr)
tr.begin_epoch(a.warmup_epochs + 1)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(batch, 3, a.img_size, a.img_size, device="cuda", generator=g)
y = torch.softmax(torch.randn(batch, a.num_classes, device="cuda", generator=g), -1)
crit = tr.criterion


def timed(fn, n=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shares = [(0, 0)] + [tuple(int(v) for v in s.split(":")) for s in os.environ.get("SHARES", "96:160,112:144,128:128,80:176,96:176,112:160,128:144,128:256,96:256").split(",")]
for r in range(2):
    for t, s in shares:
        crit.teacher_model.cu_budget = t
        tr.model.cu_budget = s
        tr.model.cu_budget_bwd = 0
        print("teacher on %3d CUs, student's forward on %3d (backward: whole chip): %.3f ms" % (t or 256, s or 256, timed(lambda: tr.step(x, y))), flush=True)
