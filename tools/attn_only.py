#!/usr/bin/env python3
"""Runs only the attention forward a few times at the bench shape (for rocprofv3 --pmc passes).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd import ops
B, N, H = 512, 197, 3
qkv = (torch.randn(B * N, 3 * H * 64, device="cuda")).to(torch.bfloat16)
o = torch.empty(B * N, H * 64, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * H * N, device="cuda")
for _ in range(5):
    ops.attention_fwd(qkv, o, lse, B, N, H, ops.UVC_BF16)
torch.cuda.synchronize()
