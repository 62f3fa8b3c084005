"""k_mlp_fused_p<TRAIN> (M >= 16384 bf16 rows, training outputs) against k_mlp_fused_v3<TRAIN> on row slices (bit for bit), and its time at the step's shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uvc_amd import ops
bf = torch.bfloat16
dev = "cuda"
def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(*s, device=dev, generator=g) * scale
D, F_ = 192, 768
for M, gated in ((100864, True), (16 * 4133 + 9, False), (100864 + 5, True)):
    x = (rnd(M, D, seed=71) * 1.5 + 0.2).to(bf); xp = rnd(M, D, seed=72).to(bf)
    gamma, beta = rnd(D, seed=73) * 0.2 + 1.0, rnd(D, seed=74) * 0.1
    g2, b2n = rnd(D, seed=75) * 0.3 + 1.0, rnd(D, seed=76) * 0.2
    W1, b1 = rnd(F_, D, seed=77, scale=0.06).to(bf), rnd(F_, seed=78) * 0.1
    W2, b2 = rnd(D, F_, seed=79, scale=0.04).to(bf), rnd(D, seed=80) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev) if gated else None
    def run(step):
        nan = float("nan")
        out, nh, h = (torch.full((M, D), nan, device=dev, dtype=bf) for _ in range(3))
        nm, nr, mean, rstd = (torch.full((M,), nan, device=dev) for _ in range(4))
        BL = bool(os.environ.get("BLOCKED")) and step >= M
        Mp = (M + 15) // 16 * 16
        gp, u = (torch.full((Mp, F_), nan, device=dev, dtype=bf) for _ in range(2))
        for lo in range(0, M, step):
            hi = min(M, lo + step)
            ops.mlp_fused_fwd(x[lo:hi], gamma, beta, W1, b1, W2, b2, out[lo:hi], next_gamma=g2, next_beta=b2n, next_h=nh[lo:hi], next_mean=nm[lo:hi], next_rstd=nr[lo:hi],
                              x_prev=xp[lo:hi] if gated else None, gate=gate, h=h[lo:hi], mean=mean[lo:hi], rstd=rstd[lo:hi], gp=gp[lo:hi] if not BL else gp, u=u[lo:hi] if not BL else u, hidden_blocked=BL)
        if BL:
            gp, u = ops.hidden_blocked_to_rows(gp, M, F_), ops.hidden_blocked_to_rows(u, M, F_)
        return out, nh, nm, nr, h, mean, rstd, gp[:M], u[:M]
    whole = run(M)
    if os.environ.get("TIME_ONLY"):
        sliced = whole
    for rep in range(0 if os.environ.get("TIME_ONLY") else 3):
        again = run(M)
        for name, a, b in zip("out nh nm nr h mean rstd gp u".split(), whole, again):
            if not torch.equal(a, b):
                bad = (a != b) & ~(torch.isnan(a.float()) & torch.isnan(b.float()))
                idx = bad.nonzero()
                print("NOT DETERMINISTIC", name, int(bad.sum()), "first", idx[:6].tolist(), "last", idx[-3:].tolist(), flush=True)
    if not os.environ.get("TIME_ONLY"):
        sliced = run(12800)
    for name, a, b in zip("out nh nm nr h mean rstd gp u".split(), whole, sliced):
        ok = bool(torch.isfinite(a.float()).all()) and torch.equal(a, b)
        print(M, gated, name, "ok" if ok else "MISMATCH %d" % int((a != b).sum()), flush=True)
    if M == 100864:
        out, nh, nm, nr, h, mean, rstd, gp, u = whole
        def t(fn, n=30):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        gpb, ub = (torch.empty((M + 15) // 16 * 16, F_, device=dev, dtype=bf) for _ in range(2))
        tr = lambda: ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out, next_gamma=g2, next_beta=b2n, next_h=nh, next_mean=nm, next_rstd=nr, x_prev=xp, gate=gate, h=h, mean=mean, rstd=rstd, gp=gpb, u=ub, hidden_blocked=bool(os.environ.get("BLOCKED")))
        inf = lambda: ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out, next_gamma=g2, next_beta=b2n, next_h=nh, x_prev=xp, gate=gate)
        print("training form %.1f us, inference form %.1f us" % (t(tr), t(inf)))
        if os.environ.get("TIME_ONLY"):
            break
