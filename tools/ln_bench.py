"""Stand-alone LayerNorm forward / backward at the widths the fused epilogues do not cover (bf16 rows, bf16 gradient stream)."""
import sys
import torch
from uvc_amd import ops

BF16 = 1


def run(rows, D, it=40):
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, dy, add1 = r(rows, D).bfloat16(), r(rows, D).bfloat16(), r(rows, D).bfloat16()
    gamma, beta = r(D), r(D)
    y = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    dx = torch.empty_like(x)
    partial = torch.empty(ops.layernorm_bwd_blocks(rows) * (2 * D + 2), device="cuda")
    dgamma, dbeta = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    a1 = torch.ones(1, device="cuda")

    def t(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    f = t(lambda: ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, BF16))
    b = t(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, partial, dgamma, dbeta, rows, D, BF16, add1=add1, a1=a1))
    fb, bb = rows * D * 2 * 2, rows * D * 2 * 4
    print("rows %6d D %3d: fwd %6.1f us (%.2f TB/s)   bwd + reduce %6.1f us (%.2f TB/s)" % (rows, D, f, fb / f / 1e6, b, bb / b / 1e6))


for rows, D in ((50432, 384), (25216, 384), (100864, 192), (25216, 768), (12608, 384)):
    run(rows, D)
