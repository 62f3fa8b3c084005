#!/usr/bin/env python3
"""Stand-alone timing of the stage-2 soft-split backward of T2T-ViT-14 at batch 128 (3136 -> 784 tokens, 64 channels, k 3, s 2, p 1):
uvc_unfold_ln_bwd + uvc_fold_tokens with dxu in natural and in tap-major column order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import _lib  # noqa: E402
if os.environ.get("UVC_LIB"):
    _lib.LIB_PATH = os.environ["UVC_LIB"]
from uvc_amd import ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import timeit  # noqa: E402

B, C, H, W, k, s, p = 128, 64, 56, 56, 3, 2, 1
dev, bf = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(3)
src = torch.randn(B, H * W, C, device=dev, generator=g)
strides = (H * W * C, 1, W * C, C)
Ho, Wo = ops.unfold_out_hw(H, W, k, s, p)
rows, dim = B * Ho * Wo, C * k * k
gamma, beta = torch.ones(dim, device=dev), torch.zeros(dim, device=dev)
out = torch.empty(rows, dim, device=dev, dtype=bf)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
ops.unfold_ln_fwd(src, strides, B, C, H, W, k, s, p, out, ops.UVC_BF16, gamma=gamma, beta=beta, mean=mean, rstd=rstd)
dy = torch.randn(rows, dim, device=dev, generator=g).to(bf)
partial = torch.empty(ops.unfold_bwd_blocks(rows) * 2 * dim, device=dev)
dgamma, dbeta = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
dxu = torch.empty(rows, dim, device=dev)
dst = torch.empty(B, H * W, C, device=dev, dtype=bf)
print("unfold_ln_fwd            %7.1f us" % timeit(lambda: ops.unfold_ln_fwd(src, strides, B, C, H, W, k, s, p, out, ops.UVC_BF16, gamma=gamma, beta=beta, mean=mean, rstd=rstd), 20))
for tm in (False, True):
    f1 = lambda: ops.unfold_ln_bwd(src, strides, B, C, H, W, k, s, p, dy, ops.UVC_BF16, gamma=gamma, mean=mean, rstd=rstd, partial=partial, dgamma=dgamma, dbeta=dbeta,  # noqa: E731
                                   dxu=dxu, dxu_tap_major=tm)
    f2 = lambda: ops.fold_tokens(dxu, dst, B, C, H, W, k, s, p, ops.UVC_BF16, tap_major=tm)  # noqa: E731
    f1()
    print("tap_major=%d  unfold_ln_bwd %7.1f us   fold %7.1f us   (dxu %.0f MB float32 written and read, dy %.0f MB, tokens %.0f + %.0f MB)" %
          (tm, timeit(f1, 20), timeit(f2, 20), rows * dim * 4 / 1e6, rows * dim * 2 / 1e6, src.numel() * 4 / 1e6, dst.numel() * 2 / 1e6))

# the image split (stage 0): 224 x 224 x 3 -> 3136 tokens of 147 (+13 K padding) features
B, C, H, W, k, s, p = 128, 3, 224, 224, 7, 4, 2
img = torch.randn(B, C, H, W, device=dev, generator=g)
strides = (C * H * W, H * W, W, 1)
Ho, Wo = ops.unfold_out_hw(H, W, k, s, p)
rows, dim, ldo = B * Ho * Wo, C * k * k, 160
gamma, beta = torch.ones(dim, device=dev), torch.zeros(dim, device=dev)
out = torch.empty(rows, ldo, device=dev, dtype=bf)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
f0 = lambda: ops.unfold_ln_fwd(img, strides, B, C, H, W, k, s, p, out, ops.UVC_BF16, gamma=gamma, beta=beta, mean=mean, rstd=rstd)  # noqa: E731
f0()
dy = torch.randn(rows, ldo, device=dev, generator=g).to(bf)
partial = torch.empty(ops.unfold_bwd_blocks(rows) * 2 * dim, device=dev)
dgamma, dbeta = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
f1 = lambda: ops.unfold_ln_bwd(img, strides, B, C, H, W, k, s, p, dy, ops.UVC_BF16, gamma=gamma, mean=mean, rstd=rstd, partial=partial, dgamma=dgamma, dbeta=dbeta)  # noqa: E731
print("image split: forward %7.1f us (%.0f MB)   LayerNorm backward %7.1f us (%.0f MB)" %
      (timeit(f0, 20), (img.numel() * 4 + out.numel() * 2) / 1e6, timeit(f1, 20), (img.numel() * 4 + dy.numel() * 2) / 1e6))
