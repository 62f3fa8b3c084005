#!/usr/bin/env python3
"""Race / edge screen of the two-group kernels (k_gemm_nt8p at every tile height, k_gemm_tn8p in its four tile shapes, k_gemm_row384_lnbwd):
random shapes, every run compared bit for bit with the generic kernel (NT) / the ring kernel (TN) and repeated -- a read that beat its LDS-DMA would
show as a mismatch that comes and goes.   python tools/fuzz_wide_gemms.py [seconds]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uvc_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = random.Random(7)
dev, bf = "cuda", torch.bfloat16
t0, n_nt, n_tn, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    # ---- NT
    M = rs.choice([rs.randint(1, 700), rs.randint(700, 9000), rs.randint(9000, 30000)])
    N = 256 * rs.randint(1, 5) + rs.choice([0, 0, 0, 8, 56, 120])
    K = 64 * rs.randint(4, 20)
    ri = rs.choice([8, 6, 5, 4])
    epi = rs.choice([ops.EPI_NONE, ops.EPI_BIAS, ops.EPI_BIAS_RESID, ops.EPI_BIAS_RESID_GATE, ops.EPI_MUL_AUX, ops.EPI_BIAS_GELU_GRAD])
    g = torch.Generator(device=dev).manual_seed(rs.randint(0, 1 << 30))
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(bf); W = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(bf)
    kw = {}
    if epi != ops.EPI_NONE and epi != ops.EPI_MUL_AUX: kw["bias"] = torch.randn(N, device=dev, generator=g)
    if epi in (ops.EPI_BIAS_RESID, ops.EPI_BIAS_RESID_GATE): kw["R"] = torch.randn(M, N, device=dev, generator=g).to(bf)
    if epi == ops.EPI_BIAS_RESID_GATE: kw["R2"] = torch.randn(M, N, device=dev, generator=g).to(bf); kw["gate"] = torch.tensor([0.3, 0.7], device=dev)
    if epi == ops.EPI_MUL_AUX: kw["aux"] = torch.randn(M, N, device=dev, generator=g).to(bf)
    outs = []
    for fg in (1, 0x100 | ri, 0x100 | ri, 4):
        C = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        k2 = dict(kw)
        if epi == ops.EPI_BIAS_GELU_GRAD: k2["C2"] = torch.full((M, N), float("nan"), device=dev, dtype=bf)
        ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=epi, force_generic=fg, **k2)
        outs.append((C, k2.get("C2")))
    for o in outs[1:]:
        if not torch.equal(o[0], outs[0][0]) or (o[1] is not None and not torch.equal(o[1], outs[0][1])):
            bad += 1; print("NT MISMATCH", M, N, K, ri, epi)
    n_nt += 1
    # ---- TN
    M = rs.choice([rs.randint(1, 300), rs.randint(300, 6000), rs.randint(6000, 26000)])
    N1, N2 = rs.choice([(768, 768), (768, 3072), (1024, 768), (384, 1152), (1152, 384), (384, 384), (576, 192), (192, 768), (768, 192), (384, 1536)])
    Ag = torch.randn(M, N1, device=dev, generator=g).to(bf); Bg = torch.randn(M, N2, device=dev, generator=g).to(bf)
    ws = torch.empty(ops.gemm_tn_workspace_bytes(M, N1, N2) // 4, device=dev)
    res = []
    for var in (2, 1, 1, 0):
        Cw = torch.full((N1, N2), float("nan"), device=dev); cs = torch.zeros(N1, device=dev)
        ops.gemm_tn(Ag, Bg, Cw, ws, dtype=ops.UVC_BF16, colsum_out=cs, variant=var)
        res.append((Cw, cs))
    ref = Ag.double().t() @ Bg.double()
    if not torch.allclose(res[0][0].double(), ref, rtol=2e-2, atol=5e-2 * max(1.0, M / 4000.0)): bad += 1; print("TN vs float64", M, N1, N2)
    for r in res[1:]:
        if not torch.equal(r[0], res[0][0]) or not torch.equal(r[1], res[0][1]): bad += 1; print("TN MISMATCH", M, N1, N2)
    n_tn += 1
torch.cuda.synchronize()
print(f"{n_nt} NT problems, {n_tn} TN problems, {bad} mismatches in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
