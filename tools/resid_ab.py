#!/usr/bin/env python3
"""Loss trajectory of the Stage-1 trainer over a few hundred optimiser steps, bf16 residual stream (the throughput mode's default) against
float32 residual rows (`precision="bf16_f32resid"`) and against the float32-exact mode, same weights / data / noise keys (the bf16
residual stream had only 2-step evidence).  Synthetic images and soft labels drawn once per step from a seeded generator; the distillation
teacher is a fixed random-init network, so there IS a signal to fit (the student moves towards the teacher's logits and the soft labels).
    python tools/resid_ab.py [steps] [batch] [model_type]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
model = sys.argv[3] if len(sys.argv) > 3 else "deit_tiny_patch16_224"
from uvc_amd.stage1 import Stage1Trainer, default_args

curves = {}
for prec in ("bf16", "bf16_f32resid", "fp32"):
    torch.manual_seed(730)
    with contextlib.redirect_stdout(sys.stderr):
        a = default_args(model_type=model, precision=prec, train_batch_size=batch, warmup_epochs=0, steps_per_epoch=steps, num_epochs=1, warmup_steps=20, learning_rate=5e-4)
        tr = Stage1Trainer(a, device="cuda")
        tr.begin_epoch(1)
    g = torch.Generator(device="cuda").manual_seed(11)
    losses = []
    for i in range(steps):
        x = torch.randn(batch, 3, 224, 224, device="cuda", generator=g)
        y = torch.softmax(2.0 * torch.randn(batch, 1000, device="cuda", generator=g), -1)
        with contextlib.redirect_stdout(sys.stderr):
            out = tr.step(x, y)
        losses.append(out["loss"])
    curves[prec] = [float(l) for l in losses]
    del tr
    torch.cuda.empty_cache()
print(f"# {model}, batch {batch}, {steps} UVC-train steps from the same init, data and noise keys; loss (mean over the 10 steps ending at the step)")
print(f"{'step':>6s} " + " ".join(f"{p:>15s}" for p in curves))
for s in list(range(9, steps, max(10, steps // 15))) + [steps - 1]:
    print(f"{s + 1:6d} " + " ".join(f"{sum(curves[p][max(0, s - 9):s + 1]) / len(curves[p][max(0, s - 9):s + 1]):15.5f}" for p in curves))
ref = curves["fp32"]
for p in ("bf16", "bf16_f32resid"):
    d = [abs(a - b) for a, b in zip(curves[p], ref)]
    print(f"{p}: max |loss - fp32 loss| over the run {max(d):.5f}, over the last 50 steps {max(d[-50:]):.5f}, final loss {curves[p][-1]:.5f} (fp32 {ref[-1]:.5f})")
