"""D = 384: uvc_gemm_nt_lnbwd (row-tile kernel) against the unfused pair uvc_gemm_nt + uvc_layernorm_bwd at DeiT-Small / T2T shapes."""
import torch
from uvc_amd import ops

BF16 = 1
D = 384


def t(fn, it=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M, K in ((50432, 1536), (50432, 1152), (25216, 1152), (25216, 384)):
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    A, Wt = r(M, K).bfloat16(), (r(D, K) * 0.04).bfloat16()
    x, add1 = r(M, D).bfloat16(), r(M, D).bfloat16()
    gamma = 1 + 0.1 * r(D)
    mean, rstd = x.float().mean(1), torch.rsqrt(x.float().var(1, unbiased=False) + 1e-6)
    dx, dyb = torch.empty_like(x), torch.empty_like(x)
    nb = max(ops.layernorm_bwd_blocks(M), 272)
    part = torch.empty(nb * (2 * D + 2), device="cuda")
    dg, db = torch.empty(D, device="cuda"), torch.empty(D, device="cuda")
    a1 = torch.ones(1, device="cuda")
    fused = t(lambda: ops.gemm_nt_lnbwd(A, Wt, x, mean, rstd, gamma, dx, part, dg, db, add1=add1, a1=a1))
    gemm = t(lambda: ops.gemm_nt(A, Wt, dyb, dtype=BF16, epilogue=ops.EPI_NONE))
    ln = t(lambda: ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dx, part, dg, db, M, D, BF16, add1=add1, a1=a1))
    print("M %6d K %4d: fused %6.1f us (%.0f TFLOP/s)   GEMM %6.1f + LayerNorm backward %5.1f = %6.1f us" % (M, K, fused, 2.0 * M * K * D / fused / 1e6, gemm, ln, gemm + ln))
