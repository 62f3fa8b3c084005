#!/usr/bin/env python3
"""Is the main stream held up at the step boundary while the teacher forward (side stream) runs?  Times, with events on the main
stream, a trivial kernel enqueued right after the teacher prefetch.  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd.stage1 import Stage1Trainer, default_args
from uvc_amd.optim import clip_grad_norm_
from uvc_amd.uvc_optimizer import uvc_optimizer

a = default_args(train_batch_size=512)
tr = Stage1Trainer(a)
tr.begin_epoch(a.warmup_epochs + 1)
x = torch.randn(512, 3, 224, 224, device="cuda")
y = torch.softmax(torch.randn(512, 1000, device="cuda"), -1)
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
dummy = torch.zeros(64, device="cuda")
N = 12
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
for i in range(N):
    e = ev[i]
    e[0].record()                                   # end of the previous step on the main stream
    tr.criterion.prefetch(x)
    dummy.add_(1.0)                                 # a trivial kernel on the main stream, right behind the prefetch
    e[1].record()
    outputs, _ = tr.model(x, tr.get_tau(), a.patch_ratio)
    e[2].record()
    loss = tr.criterion(x, outputs, y)
    loss.backward()
    clip_grad_norm_(tr.model, a.max_grad_norm); tr.optimizer.step()
    tr.scheduler.step(); tr.global_step += 1; tr.zlr_scheduler(tr.dual_opt, tr.epoch, "zlr"); tr.minimax.update_gating()
    cur, s, r, g, tr.gating_grad_list = uvc_optimizer(tr.optimizer, tr.minimax, tr.s_opt, tr.r_opt, tr.g_opt, tr.dual_opt, a, {"global_step": tr.global_step},
                                                      [], tr.flops_list, a.z_grad_clip, tr.global_step, a.gating_interval, tr.gating_grad_list)
    tr.optimizer.zero_grad()
    e[3].record()
torch.cuda.synchronize()
for i in range(2, N):
    e = ev[i]
    print("step %2d: boundary->dummy %.3f ms   forward %.3f ms   rest %.3f ms   total %.3f" % (i, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]), e[0].elapsed_time(e[3])))
