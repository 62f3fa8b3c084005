#!/usr/bin/env python3
"""Reads the s_memtime stamps of the instrumented k_gemm_nt8p copy (tools/perturb/libuvc_hip_stamp.so, built from a patched COPY of gemm.hip;
the product source carries no probe code): per phase [a = before the requests, b = requests issued, c = counted wait done, d = behind the first
barrier, e = fragments there, f = MFMA block issued]; the next phase's a = behind the second barrier.  Last k-step of workgroup 0, waves 0 and 4.
    UVC_LIB=tools/perturb/libuvc_hip_stamp.so python tools/with_lib.py tools/probe/nt8p_stamps.py [ri]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from uvc_amd import ops
ri = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M, N, K = 25216, 2304, 3072
g = torch.Generator(device="cuda").manual_seed(3)
A = (torch.randn(M, K, device="cuda", generator=g) * .5).bfloat16(); W = (torch.randn(N, K, device="cuda", generator=g) * .04).bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(M * N // 2, device="cuda", dtype=torch.int32)       # (C2's nominal size)
for _ in range(3):
    ops.gemm_nt(A, W, C, dtype=ops.UVC_BF16, epilogue=ops.EPI_NONE, force_generic=0x100 | ri, C2=dbg.view(torch.bfloat16).view(M, N))
torch.cuda.synchronize()
d = dbg[:64].cpu().numpy().astype("int64")
for grp in (0, 1):
    t = d[grp * 32: grp * 32 + 24]
    t0 = t[0]
    print(f"group {grp} (wave {4 * grp}): ticks relative to phase 0's start")
    for ph in range(4):
        row = [(int(x) - int(t0)) & 0xffffffff for x in t[ph * 6: ph * 6 + 6]]
        nxt = ((int(t[(ph + 1) * 6]) - int(t0)) & 0xffffffff) if ph < 3 else None
        print(f"  phase {ph}: a {row[0]:6d}  req +{row[1] - row[0]:4d}  vmwait +{row[2] - row[1]:4d}  barrier1 +{row[3] - row[2]:4d}  lgkm +{row[4] - row[3]:4d}  mfma +{row[5] - row[4]:4d}" + (f"  barrier2 +{nxt - row[5]:4d}   phase {nxt - row[0]:5d}" if nxt is not None else ""))
print("offset of group 1 behind group 0 at phase 0:", (int(d[32]) - int(d[0])) & 0xffffffff)
