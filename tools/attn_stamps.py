"""Cycle stamps of k_attn_bwd_one (library built with -DUVC_ATTN_PROBE=5: tools/attn_probes.sh build with PROBES=5): where a compute wave of
workgroup 0 spends a head.   UVC_LIB=tools/perturb/libuvc_hip_attnprobe5.so python tools/with_lib.py tools/attn_stamps.py"""
import numpy as np
import torch
from uvc_amd import ops
B, N, H = 512, 197, 3
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
ops.attention_fwd(qkv, o, lse, B, N, H, 1)
dqkv = torch.empty(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
delta = torch.zeros(B, H, N, device="cuda")
for _ in range(3):
    ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1, variant=2)
torch.cuda.synchronize()
st = delta.view(torch.int32).flatten()[: 8 * 16 * 16 * 8].cpu().numpy().astype(np.int64).reshape(8, 16, 16, 8) & 0xFFFFFFFF
NT = 13
nheads = 6
d = lambda a, b: (a - b) & 0xFFFFFFFF
print("per step, cycles (mean over heads 1..%d, steps; by wave):  batch1+S/dP | exp | counters+hand-on | products | step" % (nheads - 1))
for w in range(NT):
    s = st[1:nheads, w, :NT]
    ph = [d(s[..., k + 1], s[..., k]) for k in range(4)]
    print("wave %2d  %6.0f %6.0f %6.0f %6.0f   %7.0f" % (w, *[p.mean() for p in ph], d(s[..., 4], s[..., 0]).mean()))
print("batch 1 until the tile of the step before is taken in and acknowledged | consumed wait before the hand-on")
for w in range(NT):
    s = st[1:nheads, w, 1:NT]
    print("wave %2d  %6.0f %6.0f" % (w, d(s[..., 5], s[..., 0]).mean(), d(s[..., 6], s[..., 2]).mean()))
print("per head, cycles (by wave): steps | wait A | staging | wait B | head total  (start->start of the next)")
for w in range(NT):
    hd = st[:nheads, w, 15]
    tot = d(st[1:nheads, w, 15, 0], st[:nheads - 1, w, 15, 0]).mean()
    print("wave %2d  %7.0f %7.0f %7.0f %7.0f   %8.0f" % (w, d(hd[1:, 1], hd[1:, 0]).mean(), d(hd[1:, 2], hd[1:, 1]).mean(), d(hd[1:, 3], hd[1:, 2]).mean(),
                                                       d(hd[1:, 4], hd[1:, 3]).mean(), tot))
s = st[1:nheads, :NT, :NT]
print("step length by step index (mean over waves, heads):", " ".join("%d" % d(s[:, :, i, 4], s[:, :, i, 0]).mean() for i in range(NT)))
print("gap between steps (end -> next start):", " ".join("%d" % d(s[:, :, i + 1, 0], s[:, :, i, 4]).mean() for i in range(NT - 1)))
