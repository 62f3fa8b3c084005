"""Does the host run ahead of the GPU?  N steps are enqueued without any synchronisation: wall time of the enqueue loop alone against the
time until the GPU has finished them.  (If tr.step() blocks somewhere, the loop takes as long as the GPU.)
    STEP_MODEL=t2t_vit_14 STEP_BATCH=128 python tools/host_ahead.py"""
import os
import sys
import time
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
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
per = []
for i in range(N):
    s = time.perf_counter()
    tr.step(x, y)
    per.append(time.perf_counter() - s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s batch %d: enqueue loop %.2f ms per step (min %.2f, max %.2f), GPU done after %.2f ms per step" % (
    model, batch, (t1 - t0) / N * 1e3, min(per) * 1e3, max(per) * 1e3, (t2 - t0) / N * 1e3))
