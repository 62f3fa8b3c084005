"""qkv Linear + attention forward: the two kernels against the fused one (uvc_qkv_attention_fwd), DeiT-Tiny shape.  python tools/qkv_attn_time.py [B] [grid]"""
import sys
import torch
from uvc_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, D = 197, 64 * H
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(B * N, D, device="cuda", generator=g).bfloat16()
W = (torch.randn(3 * D, D, device="cuda", generator=g) * 0.08).bfloat16()
bias = torch.randn(3 * D, device="cuda", generator=g) * 0.1
qkv = torch.empty(B * N, 3 * D, device="cuda", dtype=torch.bfloat16)
o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")


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


def two():
    ops.gemm_nt(h, W, qkv, dtype=1, epilogue=ops.EPI_BIAS, bias=bias)
    ops.attention_fwd(qkv.view(B, N, 3 * D), o, lse, B, N, H, 1)


u = B * N * D * 2 / 1e6
for r in range(3):
    print("B %d grid %d: gemm + attention %6.1f us (8 u = %.0f MB)   fused, qkv stored %6.1f us (5 u)   fused, not stored %6.1f us (2 u)" % (
        B, grid, t(two), 8 * u, t(lambda: ops.qkv_attention_fwd(h, W, bias, o, lse, B, N, H, 1, qkv=qkv.view(B, N, 3 * D), grid=grid)),
        t(lambda: ops.qkv_attention_fwd(h, W, bias, o, lse, B, N, H, 1, grid=grid))), flush=True)
