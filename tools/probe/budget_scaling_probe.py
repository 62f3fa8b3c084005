"""Probe: the student's step WITHOUT any teacher work (a fixed teacher output is handed to the loss) at several CU budgets, and the teacher's forward alone at several.
    PYTHONPATH=. python tools/budget_scaling_probe.py"""
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


with torch.no_grad():
    tout, _ = crit.teacher_model(x)
ev = torch.cuda.Event()
ev.record()
a.overlap_teacher = 0


def student_only():
    crit._pref = (x.data_ptr(), x._version, tout, ev, tuple(x.shape))
    return tr.step(x, y)


def teacher_only():
    with torch.no_grad():
        return crit.teacher_model(x)


for b in (0, 224, 208, 192, 176, 160, 128):
    tr.model.cu_budget = b
    print("student's step without teacher work, budget %3d CUs: %.3f ms" % (b or 256, timed(student_only)), flush=True)
tr.model.cu_budget = 0
for b in (0, 128, 96, 80, 72, 64, 48, 32):
    crit.teacher_model.cu_budget = b
    print("teacher's forward alone, budget %3d CUs: %.3f ms" % (b or 256, timed(teacher_only)), flush=True)
