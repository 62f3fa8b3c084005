"""Attention backward at the bench shape B 512, N 197, H 3, bf16: the dq + dk/dv pair (variant 1) and the one-pass persistent kernel
(variant 2), alternating on one box.   python tools/attn_bwd_time.py [B] [H]"""
import sys
import torch
from uvc_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 197
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
ops.attention_fwd(qkv, o, lse, B, N, H, 1)
dqkv = torch.empty(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
delta = torch.empty(B, H, N, device="cuda")


def t(variant, it=30):
    for _ in range(5):
        ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1, variant=variant)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1, variant=variant)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


u = B * N * D * 2
for rnd in range(3):
    for v, name in ((1, "pair    "), (2, "one-pass")):
        us = t(v)
        print("B %d H %d  %s %7.1f us   %.0f GB/s of 8 u   checksum %.6e" % (B, H, name, us, 8 * u / us * 1e-3, float(dqkv.float().abs().sum())), flush=True)
