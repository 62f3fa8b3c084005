"""Reads the s_memtime stamps an instrumented build of k_gemm_nt256 leaves behind (tools/perturb/libuvc_hip_probe.so, built from a
patched copy of gemm.hip; not part of the product)."""
import os, sys, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
shutil.copy(os.path.join(R, "tools/perturb/libuvc_hip_probe.so"), os.path.join(R, "uvc_amd/libuvc_hip.so"))
import torch
from uvc_amd import ops
M, N, K = 25216, int(os.environ.get("TN", 2304)), int(os.environ.get("TK", 768))
dev, bf = "cuda", torch.bfloat16
A, W = (torch.randn(M, K, device=dev) * 0.5).to(bf), (torch.randn(N, K, device=dev) * 0.04).to(bf)
C = torch.empty(M, N, device=dev, dtype=bf)
buf = torch.zeros(256 * 8 * 64 * 6 + 256 * 8 * 8 * 2, device=dev, dtype=torch.int64)
for _ in range(3):
    ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=ops.EPI_NONE, ln_mean=buf.view(torch.float32)[:1] if False else None)
# pass the buffer through ln_mean (unused by this epilogue)
import ctypes as Cc
from uvc_amd import _lib as L
a = L.uvc_gemm_nt_args()
a.A, a.B, a.C = L.ptr(A), L.ptr(W), L.ptr(C)
a.alpha = 1.0; a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.ldr, a.ldaux = M, N, K, K, K, N, N, N
a.dtype, a.a_is_f32, a.c_is_f32, a.epilogue = 1, 0, 0, 0
a.ln_mean = buf.data_ptr()
L.check(L.lib().uvc_gemm_nt(Cc.byref(a), L.cur_stream()), "gemm")
torch.cuda.synchronize()
t = buf[:256 * 8 * 64 * 6].view(256, 8, 64, 6).cpu().numpy()
ep = buf[256 * 8 * 64 * 6:].view(256, 8, 8, 2).cpu().numpy()
nk = K // 64
import numpy as np
for blk in (0, 1, 100, 255):
    for w in (0, 5):
        x = t[blk, w, :nk]
        d = np.stack([x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3], x[:, 5] - x[:, 4]], 1)
        print(f"block {blk} wave {w}: per k-step [read F1 + 32 MFMA F0 | wait F1, DMA | barrier | 32 MFMA F1 + DMA issue + read F0 | wait F0] mean", d[2:].mean(0).round(0), " total/step", (x[1:, 0] - x[:-1, 0])[1:].mean().round(0))
        if blk == 0 and w == 0:
            print(d[:nk])

for blk in (0, 100):
    for w in (0, 5):
        e = ep[blk, w]
        print(f"block {blk} wave {w}: first k-step T0 {t[blk, w, 0, 0]}, loop end {t[blk, w, nk - 1, 5]}; epilogues (start, end) per tile:", [(int(a), int(b), int(b - a)) for a, b in e if a], " gaps between epilogue end and next epilogue start:", [int(e[i + 1][0] - e[i][1]) for i in range(7) if e[i + 1][0]])
