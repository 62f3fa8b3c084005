"""Fused inference MLP, bf16 residual stream: the persistent kernel (M >= 16384) against k_mlp_fused_v3 run on row slices below that
threshold (rows are independent: the concatenation must equal the persistent kernel's output BIT FOR BIT), and both timed."""
import sys
import torch
from uvc_amd import ops

D, F = 192, 768


def make(M, seed=0, gate=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g, device="cuda") * sc
    t = dict(x=(r(M, D) + 0.3).bfloat16(), gamma=1 + 0.1 * r(D), beta=0.1 * r(D), w1=r(F, D, sc=0.05).bfloat16(), b1=0.1 * r(F),
             w2=r(D, F, sc=0.05).bfloat16(), b2=0.1 * r(D), ng=1 + 0.1 * r(D), nb=0.1 * r(D))
    if gate:
        t["x_prev"] = r(M, D).bfloat16()
        t["gate"] = torch.tensor([0.3, 0.7], device="cuda")
    return t


def run(t, lo, hi, out, nh, nm, nr):
    kw = {}
    if "gate" in t:
        kw = dict(x_prev=t["x_prev"][lo:hi], gate=t["gate"])
    ops.mlp_fused_fwd(t["x"][lo:hi], t["gamma"], t["beta"], t["w1"], t["b1"], t["w2"], t["b2"], out[lo:hi], next_gamma=t["ng"], next_beta=t["nb"],
                      next_h=nh[lo:hi], next_mean=nm[lo:hi], next_rstd=nr[lo:hi], **kw)


def bufs(M):
    return (torch.empty(M, D, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16),
            torch.empty(M, device="cuda"), torch.empty(M, device="cuda"))


def check(M, gate=False, step=12800):
    t = make(M, seed=M % 1000, gate=gate)
    a, b = bufs(M), bufs(M)
    for z in a + b:
        z.fill_(7.0)
    run(t, 0, M, *a)
    for lo in range(0, M, step):
        run(t, lo, min(M, lo + step), *b)
    torch.cuda.synchronize()
    ok = all(torch.equal(p, q) for p, q in zip(a, b))
    print("M=%d gate=%d bit-identical=%s  max|out diff|=%g nonfinite=%d" % (M, gate, ok, (a[0].float() - b[0].float()).abs().max().item(),
                                                                        (~torch.isfinite(a[0].float())).sum().item()))
    return ok


def bench(M, slices=False, iters=30):
    t = make(M)
    o = bufs(M)
    step = 12800 if slices else M
    def once():
        for lo in range(0, M, step):
            run(t, lo, min(M, lo + step), *o)
    for _ in range(5):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":          # a few launches of each kernel for a rocprofv3 --pmc pass
        print(bench(int(sys.argv[2]) if len(sys.argv) > 2 else 100864, iters=5), bench(12800, iters=5))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        print("M=100864: %.1f us  M=25216: %.1f us" % (bench(100864), bench(25216)))
        sys.exit(0)
    good = True
    for M, gt in ((100864, False), (100864 + 5, False), (16384, False), (50432, True), (16 * 4133 + 9, False), (197 * 1000, False)):
        good &= check(M, gt)
    print("ALL BIT-IDENTICAL" if good else "MISMATCH")
    for M in (100864, 50432, 25216):
        print("M=%d persistent %.1f us   (v3 on 12800-row slices, launch overhead included: %.1f us)" % (M, bench(M), bench(M, True)))
    sys.exit(0 if good else 1)
