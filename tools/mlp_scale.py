#!/usr/bin/env python3
"""Fused inference MLP timing against the number of row blocks (occupancy check): time should stay flat up to the number of
workgroup slots of the chip and step up beyond."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd import ops
D, F_ = 192, 768
g = torch.Generator(device="cuda").manual_seed(1)
gamma, beta = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
W1, b1 = (torch.randn(F_, D, device="cuda", generator=g) * 0.06).bfloat16(), torch.zeros(F_, device="cuda")
W2, b2 = (torch.randn(D, F_, device="cuda", generator=g) * 0.04).bfloat16(), torch.zeros(D, device="cuda")
for nb in (128, 256, 384, 512, 640, 768, 788, 1024, 1536):
    M = nb * 128
    x = torch.randn(M, D, device="cuda", generator=g)
    out = torch.empty_like(x)
    for _ in range(3):
        ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out)
    e1.record(); e1.synchronize()
    print(f"{nb:5d} blocks of 128 rows: {e0.elapsed_time(e1) / 20 * 1000:7.1f} us")
