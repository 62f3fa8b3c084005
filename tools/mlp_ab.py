#!/usr/bin/env python3
"""A/B helper for the fused inference MLP: `mlp_ab.py save <file>` writes the kernel's outputs for a few shapes,
`mlp_ab.py cmp <a> <b>` compares two such files bit for bit.  Run once with UVC_MLP_OLD=1 and once without."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(path):
    from uvc_amd import ops
    outs = {}
    for M, F_, gated in [(100864, 768, False), (100864, 768, True), (1576, 768, False), (300, 768, True), (4133, 512, False), (515, 128, False), (70000, 256, True)]:
        g = torch.Generator(device="cuda").manual_seed(M + F_)
        D = 192
        x = torch.randn(M, D, device="cuda", generator=g) * 1.5 + 0.2
        xp = torch.randn(M, D, device="cuda", generator=g)
        gamma, beta = torch.randn(D, device="cuda", generator=g) * 0.2 + 1.0, torch.randn(D, device="cuda", generator=g) * 0.1
        W1, b1 = (torch.randn(F_, D, device="cuda", generator=g) * 0.06).bfloat16(), torch.randn(F_, device="cuda", generator=g) * 0.1
        W2, b2 = (torch.randn(D, F_, device="cuda", generator=g) * 0.04).bfloat16(), torch.randn(D, device="cuda", generator=g) * 0.1
        gate = torch.tensor([0.3, 0.7], device="cuda")
        out = torch.full((M, D), float("nan"), device="cuda")
        ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out, x_prev=xp if gated else None, gate=gate if gated else None)
        torch.cuda.synchronize()
        assert torch.isfinite(out).all(), (M, F_)
        outs[f"{M}_{F_}_{int(gated)}"] = out.cpu()
        if M == 100864 and not gated:
            for _ in range(3):
                ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out)
            e1.record(); e1.synchronize()
            print(f"mlp_fused M={M} F={F_}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us  (UVC_MLP_OLD={os.environ.get('UVC_MLP_OLD')})")
    torch.save(outs, path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    ok = True
    for k in A:
        same = torch.equal(A[k], B[k])
        print(k, "bit-identical" if same else f"DIFFERENT max abs {float((A[k] - B[k]).abs().max()):.3e}")
        ok &= same
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if sys.argv[1] == "save":
        run(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
