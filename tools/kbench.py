#!/usr/bin/env python3
"""Per-kernel microbenchmarks at the bench.py shapes (DeiT-Tiny, B=512 -> M=100864): HIP-event timing
of each C-ABI kernel, with algorithmic bytes/flops -> GB/s and TFLOP/s.  Development tool."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uvc_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--D", type=int, default=192)
    ap.add_argument("--H", type=int, default=3)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    B, D, H, N = a.B, a.D, a.H, 197
    F, M = 4 * D, a.B * 197
    dev = "cuda"
    bf = torch.bfloat16
    dt = ops.UVC_BF16
    rows = []

    def rec(name, ms, bytes_, flops=0.0):
        rows.append((name, ms, bytes_ / ms / 1e6, flops / ms / 1e9))

    def want(n):
        return (not a.only) or any(k in n for k in a.only.split(","))

    x32 = torch.randn(M, D, device=dev)
    xb = x32.to(bf)
    hF = torch.randn(M, F, device=dev).to(bf)
    g32 = torch.randn(M, D, device=dev)
    g16 = g32.to(bf)               # the backward's dL/dx streams are bf16 in the throughput mode
    bD, bF, b3 = torch.zeros(D, device=dev), torch.zeros(F, device=dev), torch.zeros(3 * D, device=dev)
    Wqkv = (torch.randn(3 * D, D, device=dev) * .02).to(bf)
    Wp = (torch.randn(D, D, device=dev) * .02).to(bf)
    W1 = (torch.randn(F, D, device=dev) * .02).to(bf)
    W2 = (torch.randn(D, F, device=dev) * .02).to(bf)
    gate = torch.tensor([0.3, 0.7], device=dev)
    if want("gemm_nt"):
        qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
        rec("gemm_nt qkv   bias         K=D N=3D", timeit(lambda: ops.gemm_nt(xb, Wqkv, qkv, dtype=dt, epilogue=ops.EPI_BIAS, bias=b3)),
            M * D * 2 + M * 3 * D * 2, 2.0 * M * D * 3 * D)
        o32 = torch.empty(M, D, device=dev)
        rec("gemm_nt proj  bias+resid   K=D N=D", timeit(lambda: ops.gemm_nt(xb, Wp, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=bD, R=x32)),
            M * D * 2 + M * D * 8, 2.0 * M * D * D)
        aa, uu = torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
        rec("gemm_nt fc1   bias+gelu    K=D N=F", timeit(lambda: ops.gemm_nt(xb, W1, aa, dtype=dt, epilogue=ops.EPI_BIAS_GELU, bias=bF, C2=uu)),
            M * D * 2 + 2 * M * F * 2, 2.0 * M * D * F)
        rec("gemm_nt fc1   gelu + gelu'  K=D N=F", timeit(lambda: ops.gemm_nt(xb, W1, aa, dtype=dt, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bF, C2=uu)),
            M * D * 2 + 2 * M * F * 2, 2.0 * M * D * F)
        rec("gemm_nt fc2   resid+gate   K=F N=D", timeit(lambda: ops.gemm_nt(hF, W2, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID_GATE, bias=bD, R=x32, R2=g32, gate=gate)),
            M * F * 2 + M * D * 12, 2.0 * M * D * F)
        dA = torch.empty(M, F, device=dev, dtype=bf)
        rec("gemm_nt dfc2  dgelu        K=D N=F", timeit(lambda: ops.gemm_nt(g16, W1, dA, dtype=dt, epilogue=ops.EPI_DGELU, aux=hF, alpha_ptr=gate)),
            M * D * 4 + 2 * M * F * 2, 2.0 * M * D * F)
        rec("gemm_nt dfc2  x stored g'  K=D N=F", timeit(lambda: ops.gemm_nt(g16, W1, dA, dtype=dt, epilogue=ops.EPI_MUL_AUX, aux=hF, alpha_ptr=gate)),
            M * D * 4 + 2 * M * F * 2, 2.0 * M * D * F)
        dH = torch.empty(M, D, device=dev, dtype=bf)
        rec("gemm_nt dfc1  none         K=F N=D", timeit(lambda: ops.gemm_nt(hF, W2, dH, dtype=dt, epilogue=ops.EPI_NONE)),
            M * F * 2 + M * D * 2, 2.0 * M * D * F)
        rec("gemm_nt dproj none         K=D N=D", timeit(lambda: ops.gemm_nt(g16, Wp, dH, dtype=dt, epilogue=ops.EPI_NONE)),
            M * D * 4 + M * D * 2, 2.0 * M * D * D)
        q3 = torch.randn(M, 3 * D, device=dev).to(bf)
        Wt = (torch.randn(D, 3 * D, device=dev) * .02).to(bf)
        rec("gemm_nt dqkv  none         K=3D N=D", timeit(lambda: ops.gemm_nt(q3, Wt, dH, dtype=dt, epilogue=ops.EPI_NONE)),
            M * 3 * D * 2 + M * D * 2, 2.0 * M * D * 3 * D)
    if want("gemm_tn"):
        ws = torch.empty(max(ops.gemm_tn_workspace_bytes(M, F, D), ops.gemm_tn_workspace_bytes(M, D, F), ops.gemm_tn_workspace_bytes(M, 3 * D, D)) // 4, device=dev)
        C1, C2, C3, C4 = torch.empty(D, F, device=dev), torch.empty(F, D, device=dev), torch.empty(D, D, device=dev), torch.empty(3 * D, D, device=dev)
        q3 = torch.randn(M, 3 * D, device=dev).to(bf)
        rec("gemm_tn dW2         [D,F]", timeit(lambda: ops.gemm_tn(g16, hF, C1, ws, dtype=dt)), M * D * 2 + M * F * 2, 2.0 * M * D * F)
        rec("gemm_tn dW1         [F,D]", timeit(lambda: ops.gemm_tn(hF, xb, C2, ws, dtype=dt)), M * D * 2 + M * F * 2, 2.0 * M * D * F)
        rec("gemm_tn dWp         [D,D]", timeit(lambda: ops.gemm_tn(g16, xb, C3, ws, dtype=dt)), M * D * 4, 2.0 * M * D * D)
        rec("gemm_tn dWqkv       [3D,D]", timeit(lambda: ops.gemm_tn(q3, xb, C4, ws, dtype=dt)), M * D * 8, 2.0 * M * D * 3 * D)
    if want("ln"):
        gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        y = torch.empty(M, D, device=dev, dtype=bf)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        rec("ln_fwd", timeit(lambda: ops.layernorm_fwd(x32, gam, bet, y, mean, rstd, M, D, dt)), M * D * 6)
        dx = torch.empty(M, D, device=dev)
        part = torch.empty(ops.layernorm_bwd_blocks(M) * (2 * D + 2), device=dev)
        dg, db, dots = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(2, device=dev)
        dx16, add16 = torch.empty(M, D, device=dev, dtype=bf), g16.clone()
        rec("ln_bwd +add1 (bf16 streams)", timeit(lambda: ops.layernorm_bwd(xb, x32, gam, mean, rstd, dx16, part, dg, db, M, D, dt, add1=g16, a1=gate[1:])), M * D * 10)
        rec("ln_bwd +add1+add2+dots (bf16 streams)", timeit(lambda: ops.layernorm_bwd(xb, x32, gam, mean, rstd, dx16, part, dg, db, M, D, dt, add1=g16, add2=add16, a2=gate[:1], dots=dots)), M * D * 12)
    if want("lnbwd") and D == 192:
        gam = torch.ones(D, device=dev)
        mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
        part = torch.empty(max(ops.layernorm_bwd_blocks(M), 272) * (2 * D + 2), device=dev)
        dg, db, dots = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(2, device=dev)
        dx16, add16 = torch.empty(M, D, device=dev, dtype=bf), g16.clone()
        W2t = (torch.randn(D, F, device=dev) * .02).to(bf)
        rec("dfc1 + ln2_bwd fused  K=F (A 155 + x 77 + add1 39 + dx 39 MB)", timeit(lambda: ops.gemm_nt_lnbwd(hF, W2t, x32, mean, rstd, gam, dx16, part, dg, db, add1=g16, a1=gate[1:])),
            M * F * 2 + M * D * 8, 2.0 * M * D * F)
        q3 = torch.randn(M, 3 * D, device=dev).to(bf)
        Wqt = (torch.randn(D, 3 * D, device=dev) * .02).to(bf)
        rec("dqkv + ln1_bwd fused  K=3D (A 116 + x 77 + add1 39 + add2 39 + dx 39 MB)", timeit(lambda: ops.gemm_nt_lnbwd(q3, Wqt, x32, mean, rstd, gam, dx16, part, dg, db, add1=g16, add2=add16, a2=gate[:1], dots=dots)),
            M * 3 * D * 2 + M * D * 10, 2.0 * M * D * 3 * D)
    if want("mlp_fused") and D == 192:
        gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        o32 = torch.empty(M, D, device=dev)
        rec("mlp_fused (LN+fc1+GELU+fc2+resid, inference)", timeit(lambda: ops.mlp_fused_fwd(x32, gam, bet, W1, bF, W2, bD, o32)),
            M * D * 8, 4.0 * M * D * F)
        hh, gpp, uu2 = torch.empty(M, D, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        rec("mlp_fused TRAIN (+h, g', u stores, gate mix)", timeit(lambda: ops.mlp_fused_fwd(x32, gam, bet, W1, bF, W2, bD, o32, x_prev=g32, gate=gate, h=hh, mean=mean, rstd=rstd, gp=gpp, u=uu2)),
            M * D * 12 + M * D * 2 + 2 * M * F * 2, 4.0 * M * D * F)
    if want("attn"):
        qkv = torch.randn(B, N, 3 * D, device=dev).to(bf)
        o = torch.empty(B, N, D, device=dev, dtype=bf)
        lse = torch.empty(B, H, N, device=dev)
        fl = 4.0 * B * H * N * N * 64
        rec("attn_fwd", timeit(lambda: ops.attention_fwd(qkv, o, lse, B, N, H, dt)), M * D * 8, fl)
        do = torch.randn(B, N, D, device=dev).to(bf)
        dq = torch.empty(B, N, 3 * D, device=dev, dtype=bf)
        dl = torch.empty(B, H, N, device=dev)
        rec("attn_bwd", timeit(lambda: ops.attention_bwd(qkv, o, lse, do, dq, dl, B, N, H, dt)), M * D * 16, fl * 3.5)
    if want("colsum"):
        part = torch.empty(ops.colsum_blocks(M) * F, device=dev)
        out = torch.empty(F, device=dev)
        rec("colsum bf16 [M,F]", timeit(lambda: ops.colsum(hF, part, out, dt)), M * F * 2)
        rec("colsum f32  [M,D]", timeit(lambda: ops.colsum(g32, part, out[:D], dt)), M * D * 4)
    if want("adamw"):
        n = 5717440
        p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
        v.abs_()
        part, sq = torch.empty(1024, device=dev), torch.zeros(2, device=dev)
        rec("grad_sqnorm", timeit(lambda: ops.grad_sqnorm(g, part, sq)), n * 4)
        rec("adamw", timeit(lambda: ops.adamw_step(p, g, m, v, sq, lr=1e-4, step=3)), n * 28)
    print(f"{'kernel':40s} {'ms':>8s} {'GB/s':>9s} {'TFLOP/s':>8s}")
    for name, ms, gbs, tf in rows:
        print(f"{name:40s} {ms:8.4f} {gbs:9.1f} {tf:8.1f}")


if __name__ == "__main__":
    main()
