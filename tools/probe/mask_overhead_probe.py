"""Probe (r6): what a CU mask itself costs.  The product's step with every persistent kernel sized for N CUs (uvc_set_cu_budget), its three streams (a) unpinned,
(b) pinned to the same N CUs by a CU mask.  N = 256 masks nothing but still takes the masked-queue path.     python tools/probe/mask_overhead_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from uvc_amd import _lib as L  # noqa: E402
from uvc_amd.stage1 import Stage1Trainer, default_args  # noqa: E402

batch = 512
a = default_args(model_type="deit_tiny_patch16_224", precision="bf16", train_batch_size=batch, local_rank=0)
tr = Stage1Trainer(a, device="cuda:0", distributed=False)
bench.pruned_state(tr)
tr.begin_epoch(a.warmup_epochs + 1)
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(batch, 3, a.img_size, a.img_size, device="cuda", generator=g)
y = torch.softmax(torch.randn(batch, a.num_classes, device="cuda", generator=g), -1)
crit = tr.criterion
dev = x.device


def timed(n=60, warm=15):
    for _ in range(warm):
        tr.step(x, y, next_x=x)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        tr.step(x, y, next_x=x)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tr.step(x, y, next_x=x)
plain_side, plain_wgrad = crit._side, tr.model._wgrad_stream
for r in range(2):
    for n in (256, 224, 192):
        tr.model.cu_budget = crit.teacher_model.cu_budget = 0 if n == 256 else n
        crit._side, tr.model._wgrad_stream = plain_side, plain_wgrad
        crit._pref = None
        a_ms = timed()
        first = 256 - n
        m_main, m_side, m_t = (L.masked_stream(dev, first, n) for _ in range(3))
        crit._side, tr.model._wgrad_stream = m_t, m_side
        crit._pref = None
        m_main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(m_main):
            b_ms = timed()
        torch.cuda.current_stream().wait_stream(m_main)
        torch.cuda.synchronize()
        print("every kernel sized for %3d CUs: streams unpinned %.3f ms, pinned to those CUs by a mask %.3f ms" % (n, a_ms, b_ms), flush=True)
