"""Is the main stream idle at the step boundary while the host enqueues the teacher forward?  Event A: end of step n (main stream);
event B: main stream right behind the host's enqueue of the teacher forward of step n + 1 (nothing else lies between A and B on that
stream, so elapsed(A, B) is time the main stream had nothing to run).  No profiler involved.
    STEP_MODEL=t2t_vit_14 STEP_BATCH=128 python tools/host_gap.py"""
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
evB = []
orig = tr.criterion.prefetch


def prefetch(inp):
    orig(inp)
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    evB.append(e)


tr.criterion.prefetch = prefetch
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
evB.clear()
evA = []
N = 30
for i in range(N):
    tr.step(x, y)
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    evA.append(e)
torch.cuda.synchronize()
gaps = sorted(evA[i].elapsed_time(evB[i + 1]) for i in range(N - 1))
steps = sorted(evA[i].elapsed_time(evA[i + 1]) for i in range(N - 1))
print("%s batch %d: main stream idle at the step boundary: median %.3f ms (p10 %.3f, p90 %.3f) of a %.2f ms step" % (
    model, batch, gaps[len(gaps) // 2], gaps[len(gaps) // 10], gaps[9 * len(gaps) // 10], steps[len(steps) // 2]))
