#!/usr/bin/env python3
"""Which ATen operators launch kernels inside one UVC-train step, and from which line of uvc_amd/ (VERDICT r3 next #9).
    STEP_MODEL=deit_tiny_patch16_224 STEP_BATCH=512 python tools/aten_in_step.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in ("view", "select", "slice", "detach", "alias", "as_strided", "empty", "_unsafe_view", "reshape", "expand", "t.default", "permute", "transpose", "squeeze", "unsqueeze", "is_same_size", "stride", "sym_")):
            fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack() if "uvc_amd" in f.filename][-3:]
            print("dispatch", name, "<-", " | ".join(fr))
        return func(*args, **(kwargs or {}))


with Spy():
    tr.step(x, y)
torch.cuda.synchronize()
n = 0
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0 and not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        frames = [f for f in ev.stack if "uvc_amd" in f or "tools/" in f or "bench.py" in f][:3]
        print(f"{ev.name:28s} device {ev.device_time_total:7.1f} us  shapes-free  <- " + " | ".join(frames))
        n += 1
print(f"{n} ATen operators with device time in one step")
