#!/usr/bin/env python3
"""Times the K = 384 streaming GEMMs of DeiT-Small / T2T-ViT-14 (qkv, proj, fc1, dfc2 x GELU', the teacher's fc1) on `k_gemm_ws`:
force_generic 0 = the dispatch's choice (r4: eight waves x 32 columns where N is a multiple of 256), 5 = six waves x 32 columns, 1 = generic 128 x 128 tiles;
checks that the three agree bit for bit.    python tools/gemm_ws384.py [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import timeit  # noqa: E402

Ms = [int(x) for x in sys.argv[1:]] or [50432, 25216]
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
for M in Ms:
    for name, n, epi in [("qkv+bias", 1152, ops.EPI_BIAS), ("proj+resid", 384, ops.EPI_BIAS_RESID), ("fc1+gelu,gelu'", 1536, ops.EPI_BIAS_GELU_GRAD),
                         ("teacher fc1+gelu", 1536, ops.EPI_BIAS_GELU_OUT), ("dfc2 x aux", 1536, ops.EPI_MUL_AUX), ("fc1 T2T (N 1152)", 1152, ops.EPI_BIAS_GELU_GRAD)]:
        k = 384
        A, W = (rn(M, k) * 0.5).to(bf), (rn(n, k) * 0.04).to(bf)
        kw = {}
        if epi != ops.EPI_MUL_AUX:
            kw["bias"] = rn(n) * 0.1
        if epi == ops.EPI_BIAS_RESID:
            kw["R"] = rn(M, n).to(bf)
        if epi == ops.EPI_MUL_AUX:
            kw["aux"] = rn(M, n).to(bf)
        outs, ts = [], []
        for fg in (0, 5, 1):
            C = torch.full((M, n), float("nan"), device=dev, dtype=bf)
            k2 = dict(kw)
            if epi == ops.EPI_BIAS_GELU_GRAD:
                k2["C2"] = torch.full((M, n), float("nan"), device=dev, dtype=bf)
            f = lambda: ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=epi, force_generic=fg, **k2)  # noqa: E731
            f()
            outs.append((C.clone(), None if "C2" not in k2 else k2["C2"].clone()))
            ts.append(min(timeit(f, 20) for _ in range(3)))
        same = all(torch.equal(o[0], outs[2][0]) and (o[1] is None or torch.equal(o[1], outs[2][1])) for o in outs[:2])
        fl = 2.0 * M * n * k
        print(f"M={M:6d} {name:18s} N={n:5d}  dispatch {ts[0]:7.1f} us {fl / ts[0] / 1e6:6.1f} TF | six waves {ts[1]:7.1f} us | generic {ts[2]:7.1f} us | bit-identical: {same}")
