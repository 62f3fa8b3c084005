#!/usr/bin/env python3
"""Weight-gradient GEMMs of the DeiT-Tiny step (batch 512) as the step launches them, by kernel variant: 0 = picked by shape (ring kernel k_gemm_tn_dma for the
192 x 256 / 256 x 192 tiles, two-group k_gemm_tn8p for 192 x 192), 1 = the two-group schedule also for 192 x 256 / 256 x 192, 2 = the ring kernel everywhere.
r5 moved these GEMMs to 128 workgroups (half the chip) where the per-workgroup pipeline, not HBM, bounds them: the r4 comparison of the two schedules was made
at 256 workgroups.  Alone (back to back) and beside a streaming kernel on a second stream (fc1-like traffic), HIP events.
    python tools/tn_variants.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
D, F, N = 192, 768, 197
VARIANTS = (0, 1, 2)
if len(sys.argv) > 2 and sys.argv[2] == "base":       # DeiT-Base widths: 256 x 256 tiles (variant 0) against 128 x 256 (variant 3)
    D, F, N, VARIANTS = 768, 3072, 198, (0, 3)
M = B * N
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
gD, hF, xD, q3 = rn(M, D), rn(M, F), rn(M, D), rn(M, 3 * D)
ws = torch.empty(max(ops.gemm_tn_workspace_bytes(M, F, D), ops.gemm_tn_workspace_bytes(M, D, F), ops.gemm_tn_workspace_bytes(M, 3 * D, D)) // 4, device=dev)
shapes = {f"dW2 [{D} x {F}]": (gD, hF, torch.empty(D, F, device=dev)), f"dW1 [{F} x {D}]": (hF, xD, torch.empty(F, D, device=dev)),
          f"dWqkv [{3 * D} x {D}]": (q3, xD, torch.empty(3 * D, D, device=dev)), f"dWproj [{D} x {D}]": (gD, xD, torch.empty(D, D, device=dev))}


def t(fn, iters=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


ref = {}
print(f"# D = {D}, batch {B}: M = {M}; us per launch (GEMM + reduce), back to back on one stream")
for name, (A, Bm, C) in shapes.items():
    row = []
    for v in VARIANTS:
        us = t(lambda: ops.gemm_tn(A, Bm, C, ws, dtype=ops.UVC_BF16, variant=v))
        torch.cuda.synchronize()
        if v == 0:
            ref[name] = C.clone()
        same = torch.equal(ref[name], C)
        row.append(f"variant {v}: {us:6.1f}{'' if same else ' (DIFFERENT BITS)'}")
    print(f"{name:20s} " + "   ".join(row))
