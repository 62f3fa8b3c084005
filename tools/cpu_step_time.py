#!/usr/bin/env python3
"""Host-side cost of one Stage-1 step: wall time of the enqueue-only call sequence (no device sync inside), by phase.
If it approaches the device step time the step is launch-bound.  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd.stage1 import Stage1Trainer, default_args
from uvc_amd.optim import clip_grad_norm_
from uvc_amd.uvc_optimizer import uvc_optimizer

MODEL = os.environ.get("STEP_MODEL", "deit_tiny_patch16_224")
BATCH = int(os.environ.get("STEP_BATCH", 512))
a = default_args(model_type=MODEL, train_batch_size=BATCH, **({"enable_patch_gating": 2} if "t2t" in MODEL else {}))
tr = Stage1Trainer(a)
tr.begin_epoch(a.warmup_epochs + 1)
x = torch.randn(BATCH, 3, 224, 224, device="cuda")
y = torch.softmax(torch.randn(BATCH, 1000, device="cuda"), -1)
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
ph = dict(prefetch=0.0, forward=0.0, loss=0.0, backward=0.0, clip_adamw=0.0, sched=0.0, uvc=0.0)
N = 20
t_all = time.perf_counter()
for _ in range(N):
    t = time.perf_counter(); tr.criterion.prefetch(x); ph["prefetch"] += time.perf_counter() - t
    t = time.perf_counter(); outputs, _ = tr.model(x, tr.get_tau(), a.patch_ratio); ph["forward"] += time.perf_counter() - t
    t = time.perf_counter(); loss = tr.criterion(x, outputs, y); ph["loss"] += time.perf_counter() - t
    t = time.perf_counter(); loss.backward(); ph["backward"] += time.perf_counter() - t
    t = time.perf_counter(); clip_grad_norm_(tr.model, a.max_grad_norm); tr.optimizer.step(); ph["clip_adamw"] += time.perf_counter() - t
    t = time.perf_counter(); tr.scheduler.step(); tr.global_step += 1; tr.zlr_scheduler(tr.dual_opt, tr.epoch, "zlr"); tr.minimax.update_gating(); ph["sched"] += time.perf_counter() - t
    t = time.perf_counter()
    cur, s, r, g, tr.gating_grad_list = uvc_optimizer(tr.optimizer, tr.minimax, tr.s_opt, tr.r_opt, tr.g_opt, tr.dual_opt, a, {"global_step": tr.global_step},
                                                      [], tr.flops_list, a.z_grad_clip, tr.global_step, a.gating_interval, tr.gating_grad_list)
    tr.optimizer.zero_grad(); ph["uvc"] += time.perf_counter() - t
host = (time.perf_counter() - t_all) / N
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / N
print("%s batch %d: host enqueue ms/step %.3f   device-inclusive ms/step %.3f" % (MODEL, BATCH, host * 1e3, total * 1e3))
for k, v in ph.items():
    print("  %-12s %.3f ms" % (k, v / N * 1e3))
