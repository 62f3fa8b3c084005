#!/usr/bin/env python3
"""Times uvc_gemm_nt on a few DeiT-Base shapes for the variants given on the command line (force_generic values): used with
tools/with_lib.py to compare builds of the library.   python tools/gemm8p_one.py 0x108 0x105 4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import timeit  # noqa: E402

fgs = [int(x, 0) for x in sys.argv[1:]] or [0x108]
M = int(os.environ.get("SWEEP_M", 25216))
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
tag = os.environ.get("UVC_LIB", "in-tree").split("libuvc_hip")[-1]
for name, n, k, epi in [("qkv", 2304, 768, ops.EPI_BIAS), ("none2304", 2304, 768, ops.EPI_NONE), ("dfc2xaux", 3072, 768, ops.EPI_MUL_AUX), ("dfc1", 768, 3072, ops.EPI_NONE),
                        ("k6144", 2304, 6144, ops.EPI_NONE)]:
    A, W = (rn(M, k) * 0.5).to(bf), (rn(n, k) * 0.04).to(bf)
    C = torch.empty(M, n, device=dev, dtype=bf)
    kw = {}
    if epi == ops.EPI_BIAS:
        kw = dict(bias=torch.zeros(n, device=dev))
    if epi == ops.EPI_MUL_AUX:
        kw = dict(aux=rn(M, n).to(bf))
    ts = []
    for fg in fgs:
        t = min(timeit(lambda: ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=epi, force_generic=fg, **kw), 20) for _ in range(3))
        ts.append(t)
    fl = 2.0 * M * n * k
    print(f"{tag:12s} {name:10s} N={n:5d} K={k:5d} " + " ".join(f"fg={fg:#x}: {t:7.1f} us {fl / t / 1e6:7.1f} TF" for fg, t in zip(fgs, ts)))
