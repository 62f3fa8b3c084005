#!/usr/bin/env python3
"""Stand-alone timing of uvc_gemm_nt / uvc_gemm_tn on the wide models' shapes (DeiT-Base batch 128, DeiT-Small batch 256): kernels picked
by shape against the generic tiled kernel (force_generic = 1).  HIP events, back-to-back launches, random operands.
    python tools/gemm_bench.py [--model base|small] [--iters 30]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import ops  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="base")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=0)
    a = ap.parse_args()
    D, F, B = {"base": (768, 3072, 128), "small": (384, 1536, 256), "t2t": (384, 1152, 128), "tiny": (192, 768, 512)}[a.model]
    B = a.batch or B
    M = B * 197
    dev, bf = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(3)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    shapes = [("qkv (bias)", M, 3 * D, D, ops.EPI_BIAS), ("proj (+resid)", M, D, D, ops.EPI_BIAS_RESID), ("fc1 (gelu, gelu')", M, F, D, ops.EPI_BIAS_GELU_GRAD),
              ("fc2 (+resid+gate)", M, D, F, ops.EPI_BIAS_RESID_GATE), ("dfc2 (x aux)", M, F, D, ops.EPI_MUL_AUX), ("dfc1", M, D, F, ops.EPI_NONE),
              ("dqkv", M, D, 3 * D, ops.EPI_NONE), ("dproj", M, D, D, ops.EPI_NONE)]
    print(f"{a.model}: D={D} F={F} batch={B} M={M}")
    print(f"{'gemm_nt':22s} {'M':>7s} {'N':>6s} {'K':>6s} {'by shape us':>12s} {'TF/s':>8s} {'generic us':>11s} {'TF/s':>8s}")
    for name, m, n, k, epi in shapes:
        A, W = (rn(m, k) * 0.5).to(bf), (rn(n, k) * 0.04).to(bf)
        C, C2 = torch.empty(m, n, device=dev, dtype=bf), torch.empty(m, n, device=dev, dtype=bf)
        kw = dict(bias=torch.zeros(n, device=dev))
        if epi in (ops.EPI_BIAS_RESID, ops.EPI_BIAS_RESID_GATE):
            kw.update(R=rn(m, n).to(bf))
        if epi == ops.EPI_BIAS_RESID_GATE:
            kw.update(R2=rn(m, n).to(bf), gate=torch.tensor([0.3, 0.7], device=dev))
        if epi == ops.EPI_MUL_AUX:
            kw = dict(aux=rn(m, n).to(bf))
        if epi == ops.EPI_NONE:
            kw = {}
        if epi == ops.EPI_BIAS_GELU_GRAD:
            kw.update(C2=C2)
        fl = 2.0 * m * n * k
        t = [timeit(lambda fg=fg: ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=epi, force_generic=fg, **kw), a.iters) for fg in (0, 1)]
        print(f"{name:22s} {m:7d} {n:6d} {k:6d} {t[0]:12.1f} {fl / t[0] / 1e6:8.1f} {t[1]:11.1f} {fl / t[1] / 1e6:8.1f}")
    print(f"{'gemm_tn':22s} {'M':>7s} {'N1':>6s} {'N2':>6s} {'us':>12s} {'TF/s':>8s}")
    for name, n1, n2 in [("dW2", D, F), ("dW1", F, D), ("dWqkv", 3 * D, D), ("dWproj", D, D)]:
        Ag, Bg = rn(M, n1).to(bf), rn(M, n2).to(bf)
        Cw = torch.empty(n1, n2, device=dev)
        ws = torch.empty(ops.gemm_tn_workspace_bytes(M, n1, n2) // 4, device=dev)
        t = timeit(lambda: ops.gemm_tn(Ag, Bg, Cw, ws, dtype=ops.UVC_BF16), a.iters)
        t1 = timeit(lambda: ops.gemm_tn(Ag, Bg, Cw, ws, dtype=ops.UVC_BF16, variant=1), a.iters)
        print(f"{name:22s} {M:7d} {n1:6d} {n2:6d} {t:12.1f} {2.0 * M * n1 * n2 / t / 1e6:8.1f}   variant 1: {t1:8.1f} us {2.0 * M * n1 * n2 / t1 / 1e6:8.1f}")


if __name__ == "__main__":
    main()
