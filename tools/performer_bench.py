#!/usr/bin/env python3
"""Stand-alone timing of the Performer linear-attention kernels at the T2T-ViT-14 stage-1 shape (batch 128, 3136 tokens).
UVC_LIB=<path> selects another build of the library for A/B runs.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd import _lib
if os.environ.get("UVC_LIB"):
    _lib.LIB_PATH = os.environ["UVC_LIB"]
from uvc_amd import ops
B, T = 128, 3136
kqv = torch.randn(B * T, 192, device='cuda') * 0.5
w = torch.randn(32, 64, device='cuda') * 0.7
S = 8
part = torch.empty(B * S * 65 * 32, device='cuda'); kptv = torch.empty(B, 65, 32, device='cuda'); dkptv = torch.empty_like(kptv)
att = torch.empty(B * T, 64, device='cuda', dtype=torch.bfloat16); datt = torch.randn(B * T, 64, device='cuda').to(torch.bfloat16)
dkqv = torch.empty(B * T, 192, device='cuda', dtype=torch.bfloat16)


def t(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(_lib.LIB_PATH)
print("fwd  us %.1f" % t(lambda: ops.performer_fwd(kqv, w, part, kptv, att, B, T, 1)))
print("bwd  us %.1f" % t(lambda: ops.performer_bwd(kqv, w, part, kptv, datt, dkqv, dkptv, B, T, 1, dskip=datt)))
