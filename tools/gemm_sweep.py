#!/usr/bin/env python3
"""K sweep of uvc_gemm_nt at M = 25216: separates the per-k-step cost of a kernel from its fixed cost per tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd import ops
from gemm_bench import timeit

M = int(os.environ.get("SWEEP_M", 25216))
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
for N in (2304, 768):
    for K in (256, 768, 1536, 3072, 6144):
        A, W = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(bf), (torch.randn(N, K, device=dev, generator=g) * 0.04).to(bf)
        C = torch.empty(M, N, device=dev, dtype=bf)
        t = [timeit(lambda fg=fg: ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=ops.EPI_NONE, force_generic=fg), 20) for fg in (0, 1)]
        fl = 2.0 * M * N * K
        print(f"M={M} N={N:5d} K={K:5d}  by-shape {t[0]:8.1f} us {fl / t[0] / 1e6:7.1f} TF/s   generic {t[1]:8.1f} us {fl / t[1] / 1e6:7.1f} TF/s")
