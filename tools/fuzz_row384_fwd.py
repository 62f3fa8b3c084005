#!/usr/bin/env python3
"""Race / edge screen of uvc_gemm_nt with ln_out at N = 384 (k_gemm_row384_lnbwd<.., 1>): random row counts from 4096 up (partial last tiles, fewer tiles
than workgroups, several tiles per workgroup), random K in multiples of 64 from 384, with and without the gate mix and the statistics; every result compared
BIT FOR BIT with the generic kernel + the stand-alone LayerNorm pass, each problem twice.   python tools/fuzz_row384_fwd.py [problems] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import ops  # noqa: E402

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = random.Random(seed)
dev, bf, N = "cuda", torch.bfloat16, 384
g = torch.Generator(device=dev).manual_seed(seed)
bad = 0
for i in range(n):
    M = rng.choice([4096, 4097, 4096 + rng.randrange(1, 40000), 128 * rng.randrange(32, 300), 128 * rng.randrange(32, 300) + rng.randrange(1, 128)])
    K = 64 * rng.randrange(6, 33)
    gate = rng.random() < 0.5
    stats = rng.random() < 0.7
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(bf)
    W = (torch.randn(N, K, device=dev, generator=g) * 0.04).to(bf)
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    R, R2 = torch.randn(M, N, device=dev, generator=g).to(bf), torch.randn(M, N, device=dev, generator=g).to(bf)
    gm, bt = 1 + 0.1 * torch.randn(N, device=dev, generator=g), 0.1 * torch.randn(N, device=dev, generator=g)
    kw = dict(dtype=ops.UVC_BF16, epilogue=ops.EPI_BIAS_RESID_GATE if gate else ops.EPI_BIAS_RESID, bias=bias, R=R)
    if gate:
        kw.update(R2=R2, gate=torch.tensor([rng.random(), rng.random()], device=dev))
    C0, h0 = torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
    m0, r0 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.gemm_nt(A, W, C0, force_generic=1, **kw)
    ops.layernorm_fwd(C0, gm, bt, h0, m0, r0, M, N, ops.UVC_BF16)
    for t in range(2):
        C1, h1 = torch.full_like(C0, float("nan")), torch.full_like(h0, float("nan"))
        m1, r1 = torch.full_like(m0, float("nan")), torch.full_like(r0, float("nan"))
        ops.gemm_nt(A, W, C1, ln_gamma=gm, ln_beta=bt, ln_out=h1, ln_mean=m1 if stats else None, ln_rstd=r1 if stats else None, **kw)
        ok = torch.equal(C1, C0) and torch.equal(h1, h0) and (not stats or (torch.equal(m1, m0) and torch.equal(r1, r0)))
        if not ok:
            bad += 1
            print("MISMATCH", dict(M=M, K=K, gate=gate, stats=stats, trial=t))
print(f"{n} problems x 2, seed {seed}: {bad} mismatches")
