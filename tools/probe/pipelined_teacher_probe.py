"""Probe: the teacher's forward a batch AHEAD, on its own share of the chip for the whole step (teacher model on TEACHER_CUS, student on STUDENT_CUS),
against the teacher under the student's forward.   TEACHER_CUS=64 STUDENT_CUS=192 python tools/pipelined_teacher_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from uvc_amd import _lib as L  # noqa: E402
from uvc_amd.stage1 import Stage1Trainer, default_args  # noqa: E402

model = os.environ.get("STEP_MODEL", "deit_tiny_patch16_224")
batch = int(os.environ.get("STEP_BATCH", "512"))
tc, sc = int(os.environ.get("TEACHER_CUS", "64")), int(os.environ.get("STUDENT_CUS", "192"))
a = default_args(model_type=model, precision="bf16", train_batch_size=batch, local_rank=0)
tr = Stage1Trainer(a, device="cuda:0", distributed=False)
bench.pruned_state(tr)
tr.begin_epoch(a.warmup_epochs + 1)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(batch, 3, a.img_size, a.img_size, device="cuda", generator=g)
y = torch.softmax(torch.randn(batch, a.num_classes, device="cuda", generator=g), -1)
crit = tr.criterion


def timed(fn, n=60, warm=15):
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


def plain():
    return tr.step(x, y)


pending = []


def ahead():
    # the teacher's forward for the NEXT step's batch starts now; this step's loss takes the one started a step ago
    crit.prefetch(x)
    pending.append(crit._pref)
    crit._pref = pending.pop(0) if len(pending) > 1 else None
    a.overlap_teacher = 0
    return tr.step(x, y)


for r in range(2):
    a.overlap_teacher = 1
    tr.model.cu_budget = 0
    crit.teacher_model.cu_budget = 0
    pending.clear(); crit._pref = None
    print("teacher under the student's forward, whole chip each: %.3f ms" % timed(plain), flush=True)
    crit.teacher_model.cu_budget = tc
    tr.model.cu_budget = sc
    pending.clear(); crit._pref = None
    print("teacher a batch ahead on %d CUs, student on %d: %.3f ms" % (tc, sc, timed(ahead)), flush=True)
    torch.cuda.synchronize()
