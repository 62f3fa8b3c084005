"""Attention backward (dQ kernel + dK / dV kernel) at the bench shape B 512, N 197, H 3, bf16."""
import torch
from uvc_amd import ops
B, N, H = 512, 197, 3
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
ops.attention_fwd(qkv, o, lse, B, N, H, 1)
dqkv = torch.empty(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
delta = torch.empty(B, H, N, device="cuda")


def t(it=30):
    for _ in range(5):
        ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


print("attention backward %.1f us   checksum %.6e" % (t(), float(dqkv.float().abs().sum())))
