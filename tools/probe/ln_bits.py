"""Prints a hash of the LayerNorm forward / backward outputs (D = 192 / 384 / 768, bf16 rows) and of the two D = 384 row kernels: run with two builds of the
library (UVC_LIB=...) to show that a change left every bit in place."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from uvc_amd import _lib  # noqa: E402
if os.environ.get("UVC_LIB"):
    _lib.LIB_PATH = os.environ["UVC_LIB"]
from uvc_amd import ops  # noqa: E402


def h(*ts):
    m = hashlib.sha256()
    for t in ts:
        m.update(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return m.hexdigest()[:16]


g = torch.Generator(device="cuda").manual_seed(11)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
bf = torch.bfloat16
for D in (192, 384, 768):
    M = 5000
    x, dy, ad = rn(M, D).to(bf), rn(M, D).to(bf), rn(M, D).to(bf)
    gm, bt = 1 + 0.1 * rn(D), 0.1 * rn(D)
    y, mean, rstd = torch.empty_like(x), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops.layernorm_fwd(x, gm, bt, y, mean, rstd, M, D, ops.UVC_BF16)
    dx = torch.empty_like(x)
    part = torch.empty(max(ops.layernorm_bwd_blocks(M), 272) * (2 * D + 2), device="cuda")
    dg, db, dots = torch.empty(D, device="cuda"), torch.empty(D, device="cuda"), torch.empty(2, device="cuda")
    ops.layernorm_bwd(dy, x, gm, mean, rstd, dx, part, dg, db, M, D, ops.UVC_BF16, add1=ad, dots=dots)
    print("D", D, "fwd", h(y, mean, rstd), "bwd", h(dx, dg, db, dots))
M, K, N = 4096 + 37, 1152, 384
A, W = (rn(M, K) * 0.5).to(bf), (rn(N, K) * 0.04).to(bf)
R, R2 = rn(M, N).to(bf), rn(M, N).to(bf)
C, hh = torch.empty(M, N, device="cuda", dtype=bf), torch.empty(M, N, device="cuda", dtype=bf)
m1, r1 = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
gm, bt = 1 + 0.1 * rn(N), 0.1 * rn(N)
ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=ops.EPI_BIAS_RESID_GATE, bias=rn(N) * 0.1, R=R, R2=R2, gate=torch.tensor([0.3, 0.7], device="cuda"),
            ln_gamma=gm, ln_beta=bt, ln_out=hh, ln_mean=m1, ln_rstd=r1)
print("row384 forward", h(C, hh, m1, r1))
x = rn(M, N).to(bf)
mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
ops.layernorm_fwd(x, gm, bt, torch.empty_like(x), mean, rstd, M, N, ops.UVC_BF16)
Wt = (rn(N, K) * 0.04).to(bf)          # [D, K]: A [M, K] . Wt^T
dx = torch.empty(M, N, device="cuda", dtype=bf)
part = torch.empty(272 * (2 * N + 2), device="cuda")
dg, db, dots = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(2, device="cuda")
ops.gemm_nt_lnbwd(A, Wt, x, mean, rstd, gm, dx, part, dg, db, add1=R, add2=R2, dots=dots)
print("row384 backward", h(dx, dg, db, dots))
