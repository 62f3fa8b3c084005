import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from uvc_amd import _lib
if os.environ.get("UVC_LIB"): _lib.LIB_PATH = os.environ["UVC_LIB"]
from uvc_amd import ops
from gemm_bench import timeit
B, C, H, W, k, s, p = 128, 3, 224, 224, 7, 4, 2
img = torch.randn(B, C, H, W, device="cuda")
strides = (C * H * W, H * W, W, 1)
rows, dim, ldo = B * 56 * 56, 147, 160
gamma, beta = torch.ones(dim, device="cuda"), torch.zeros(dim, device="cuda")
out = torch.empty(rows, ldo, device="cuda", dtype=torch.bfloat16)
mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
f0 = lambda: ops.unfold_ln_fwd(img, strides, B, C, H, W, k, s, p, out, ops.UVC_BF16, gamma=gamma, beta=beta, mean=mean, rstd=rstd)
f0()
print(os.environ.get("UVC_LIB", "in-tree").split("libuvc_hip")[-1], "image split forward %.1f us" % timeit(f0, 20))
