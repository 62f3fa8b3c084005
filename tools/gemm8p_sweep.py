#!/usr/bin/env python3
"""k_gemm_nt8p against round 3's kernels on DeiT-Base / Small shapes: every tile height (force_generic = 0x100 | RI), the lock-step
256 x 256 kernel (4), the generic 128 x 128 kernel (1) and the dispatch's own choice (0).  HIP events, interleaved rounds, random data.
    python tools/gemm8p_sweep.py [--model base|small] [--batch B] [--rounds 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import ops  # noqa: E402
from gemm_bench import timeit  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="base")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
D, F, B = {"base": (768, 3072, 128), "small": (384, 1536, 256), "t2t": (384, 1152, 128)}[a.model]
B = a.batch or B
M = B * 197
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
shapes = [("qkv (bias)", M, 3 * D, D, ops.EPI_BIAS), ("proj (+resid)", M, D, D, ops.EPI_BIAS_RESID), ("fc1 (gelu, gelu')", M, F, D, ops.EPI_BIAS_GELU_GRAD),
          ("fc2 (+resid+gate)", M, D, F, ops.EPI_BIAS_RESID_GATE), ("dfc2 (x aux)", M, F, D, ops.EPI_MUL_AUX), ("dfc1", M, D, F, ops.EPI_NONE),
          ("dqkv", M, D, 3 * D, ops.EPI_NONE), ("dproj", M, D, D, ops.EPI_NONE)]
variants = [("auto", 0), ("ri8", 0x108), ("ri6", 0x106), ("ri5", 0x105), ("ri4", 0x104), ("nt256", 4), ("generic", 1)]
print(f"{a.model}: D={D} F={F} batch={B} M={M}; microseconds (min over {a.rounds} interleaved rounds), TFLOP/s of the best")
print(f"{'gemm_nt':20s} {'N':>5s} {'K':>5s} " + " ".join(f"{n:>8s}" for n, _ in variants) + f" {'best TF/s':>10s}")
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
    best = {}
    for _ in range(a.rounds):
        for vn, fg in variants:
            try:
                t = timeit(lambda fg=fg: ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=epi, force_generic=fg, **kw), a.iters)
            except Exception:
                t = float("nan")
            best[vn] = min(best.get(vn, 1e30), t) if t == t else float("nan")
    fl = 2.0 * m * n * k
    tb = min(v for v in best.values() if v == v)
    print(f"{name:20s} {n:5d} {k:5d} " + " ".join(f"{best[vn]:8.1f}" for vn, _ in variants) + f" {fl / tb / 1e6:10.1f}")
