"""Attention forward + the dq / dk-dv pair at a shape: python tools/attn_fwd_time.py [B] [H] [N]"""
import sys
import torch
from uvc_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(sys.argv[3]) if len(sys.argv) > 3 else 197
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
dqkv = torch.empty(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
delta = torch.empty(B, H, N, device="cuda")


def t(fn, it=40):
    for _ in range(8):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for r in range(3):
    print("B %d H %d N %d  forward %6.1f us   pair %6.1f us   checksum %.6e %.6e" % (
        B, H, N, t(lambda: ops.attention_fwd(qkv, o, lse, B, N, H, 1)),
        t(lambda: ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1, variant=1)), float(o.float().abs().sum()), float(dqkv.float().abs().sum())), flush=True)
