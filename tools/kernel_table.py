#!/usr/bin/env python3
"""The kernels of one DeiT Stage-1 step (bf16 mode) as stand-alone launches at the step's shapes.

Shared by bench.py (which times every entry with HIP events, multiplies by the launches per step and names the entry with the
largest total as `roofline.kernel`) and tools/kbench.py (development print-out).  Each entry:
    key      short name used in bench.py's JSON and profiles/*_pmc_traffic.json
    rocprof  substring of the kernel name in a rocprofv3 kernel trace (profiles/*_kernel_stats_*.csv)
    calls    launches per step (student forward + teacher forward + backward, L blocks)
    bytes    ALGORITHMIC HBM bytes of one launch (every operand once)
    flops    MFMA flops of one launch (0 for non-GEMM kernels)
    fn       closure that enqueues one launch on the current stream

Only shapes with a dedicated streaming kernel are listed in full (DeiT-Tiny widths); for other widths the list is the same set of
library calls and `rocprof` is a generic prefix.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uvc_amd import ops  # noqa: E402


def build(B, D=192, H=3, L=12, N=197, with_teacher=True, tail=True, F=None, resid_f32=None):
    """tail=True (the engine's default, uvc_vit_io.full_tail = 0): the last block runs everything behind its qkv projection on the
    B class-token rows only, so the full-row kernels of that part launch L - 1 times per pass and the token-query attention once;
    the proj / MLP GEMMs on B rows are a few microseconds each and are not listed."""
    F, M = (F or 4 * D), B * N
    if resid_f32 is None:           # the engine's default (uvc_vit_cfg.resid_f32): bf16 residual-stream rows unless UVC_RESID_F32=1
        resid_f32 = os.environ.get("UVC_RESID_F32", "0") not in ("", "0")
    Lf = L - 1 if tail else L          # launches per pass of the full-row kernels behind the qkv projection
    dev = "cuda"
    bf = torch.bfloat16
    dt = ops.UVC_BF16
    T = 1 if with_teacher else 0
    tiny = D == 192
    rows = []

    # share of the algorithmic bytes that are WRITES, by key prefix: what HBM delivers depends on the mix (hbm_ceiling_gbs below)
    WFRAC = {"ln_fwd": 1 / 3, "qkv": 0.75, "attn_fwd": 0.25, "attn_tok_fwd": 0.0, "attn_tok_bwd": 0.6, "proj+resid+norm2": 0.5, "proj+resid": 0.4,
             "fc1": 8 / 9, "fc2+resid+gate+norm1": 3 / 11, "fc2+resid+gate": 0.2, "teacher mlp_fused+norm1": 0.6, "teacher mlp_fused": 0.5,
             "dfc2": 4 / 9, "dfc1+ln2_bwd": 1 / 8, "dqkv+ln1_bwd": 1 / 8, "dfc1": 0.2, "dqkv": 0.25, "ln_bwd": 0.2, "dproj": 0.5, "attn_bwd": 0.375,
             "dW": 0.15, "clip+adamw": 0.45}

    def add(key, rocprof, calls, nbytes, flops, fn, wfrac=None):
        wf = wfrac if wfrac is not None else next((v for k, v in WFRAC.items() if key.startswith(k)), None)
        rows.append(dict(key=key, rocprof=rocprof, calls=calls, bytes=int(nbytes), flops=float(flops), fn=fn, wfrac=wf))

    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    x32 = rn(M, D)
    xb = x32.to(bf)
    hF = rn(M, F).to(bf)
    g32 = rn(M, D)
    g16 = g32.to(bf)
    bD, bF, b3 = torch.zeros(D, device=dev), torch.zeros(F, device=dev), torch.zeros(3 * D, device=dev)
    Wqkv, Wp = (rn(3 * D, D) * .02).to(bf), (rn(D, D) * .02).to(bf)
    W1, W2 = (rn(F, D) * .02).to(bf), (rn(D, F) * .02).to(bf)
    gate = torch.tensor([0.3, 0.7], device=dev)
    gam, bet = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    u = 2 * M * D                     # bytes of one bf16 [M, D] tensor
    ru = 2 * u if resid_f32 else u    # bytes of one [M, D] tensor of residual-stream rows (x_l, x1: float32 in rounds 1-2, bf16 now)
    rdt = torch.float32 if resid_f32 else bf
    xr, gr_ = x32.to(rdt), g32.to(rdt)      # residual-stream operands (R, R2, LayerNorm inputs)

    # ---- forward (student; the teacher repeats LayerNorm1, qkv, attention, proj and runs the fused MLP)
    y = torch.empty(M, D, device=dev, dtype=bf)
    m_, r_ = torch.empty(M, device=dev), torch.empty(M, device=dev)
    # DeiT-Tiny width (uvc_vit_io.fuse_next_ln): norm1 of blocks 1 .. L-1 is written by the kernel that produces their input rows
    # (fc2 + residual + gate mix of the student, the fused MLP of the teacher) and the student's norm2 by attn.proj + residual, so only
    # block 0's norm1 is a pass of its own (the final norms and the last block's token-row norm2 run on B rows: not listed)
    # D = 384 (r4, M >= 4096): the same through k_gemm_row384_lnbwd<.., 1> -- fc2 and attn.proj of the student AND of the teacher write the next LayerNorm
    fuse_ln = tiny or (D == 384 and M >= 4096)
    row384 = fuse_ln and not tiny
    n_ln1 = 1 if fuse_ln else L
    add("ln_fwd", "k_ln_fwd_v", (1 + T) * n_ln1 + (0 if fuse_ln else Lf), ru + u, 0, lambda: ops.layernorm_fwd(xr, gam, bet, y, m_, r_, M, D, dt), wfrac=u / (ru + u))
    qkv = torch.empty(M, 3 * D, device=dev, dtype=bf)
    qkv3 = rn(B, N, 3 * D).to(bf)
    o = torch.empty(B, N, D, device=dev, dtype=bf)
    lse = torch.empty(B, H, N, device=dev)
    afl = 4.0 * B * H * N * N * 64
    # r5: where uvc_qkv_attention_fwd exists (DeiT-Tiny's shape) the qkv Linear and the attention forward of every block but the last are ONE kernel:
    # h in, o out, qkv written only by the student (the backward reads it); the last block keeps the qkv GEMM (its attention runs on the token rows)
    fused_qa = tail and D == 192 and ops.qkv_attention_supported(B, N, H, D, dt)      # (the engine's condition: vit_engine.hip)
    if fused_qa:
        add("qkv+attn_fwd (student)", "k_qkv_attn_fwd<6, true", Lf, 5 * u, 2.0 * M * D * 3 * D + afl,
            lambda: ops.qkv_attention_fwd(xb, Wqkv, b3, o, lse, B, N, H, dt, qkv=qkv.view(B, N, 3 * D)), wfrac=0.8)
        if T:
            add("qkv+attn_fwd (teacher)", "k_qkv_attn_fwd<6, false", Lf, 2 * u, 2.0 * M * D * 3 * D + afl,
                lambda: ops.qkv_attention_fwd(xb, Wqkv, b3, o, lse, B, N, H, dt), wfrac=0.5)
    add("qkv", "k_gemm_ws<unsigned short, unsigned short, 1" if tiny else "k_gemm", (1 + T) * (L - Lf if fused_qa else L), 4 * u, 2.0 * M * D * 3 * D,
        lambda: ops.gemm_nt(xb, Wqkv, qkv, dtype=dt, epilogue=ops.EPI_BIAS, bias=b3))
    if not fused_qa:
        add("attn_fwd", "k_attn_fwd", (1 + T) * Lf, 4 * u, afl, lambda: ops.attention_fwd(qkv3, o, lse, B, N, H, dt))
    if tail:
        oc, doc = torch.empty(B, 1, D, device=dev, dtype=bf), rn(B, 1, D).to(bf)
        dq_t = torch.empty(B, N, 3 * D, device=dev, dtype=bf)
        add("attn_tok_fwd (last block)", "k_attn_tok_fwd", 1 + T, 2 * u, 0, lambda: ops.attention_tok_fwd(qkv3, oc, B, N, H, 1, dt))
        add("attn_tok_bwd (last block)", "k_attn_tok_bwd", 1, 5 * u, 0, lambda: ops.attention_tok_bwd(qkv3, oc, doc, dq_t, B, N, H, 1, dt))
    o32 = torch.empty(M, D, device=dev, dtype=rdt)      # output rows of the residual stream
    # student at DeiT-Tiny width: norm2 leaves with attn.proj's rows (the teacher's fused MLP normalises its rows itself)
    if fuse_ln:
        add("proj+resid+norm2", "k_gemm_row384_lnbwd<true, 1>" if row384 else "k_gemm_wsn16_dma<3, 6", Lf * (1 + T if row384 else 1), 2 * u + 2 * ru, 2.0 * M * D * D,
            lambda: ops.gemm_nt(xb, Wp, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=bD, R=xr, ln_gamma=gam, ln_beta=bet, ln_out=y, ln_mean=m_, ln_rstd=r_),
            wfrac=(ru + u) / (2 * u + 2 * ru))
    if not fuse_ln or (T and tiny):
        add("proj+resid", "k_gemm_wsn16_dma<3, 6, false" if tiny else "k_gemm", (T if fuse_ln else 1 + T) * Lf, u + 2 * ru, 2.0 * M * D * D,
            lambda: ops.gemm_nt(xb, Wp, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=bD, R=xr), wfrac=ru / (u + 2 * ru))
    aa, uu = torch.empty(M, F, device=dev, dtype=bf), torch.empty(M, F, device=dev, dtype=bf)
    # r6: where the one-byte GELU' code exists (DeiT-Tiny's width; the engine's condition, vit_engine.hip gp_q8) fc1 writes GELU(a) in bf16 and GELU'(a) as bytes:
    # h2 in (1 u) + 4 u + 2 u out; the dgrad of fc2 reads the bytes
    q8 = ops.gemm_nt_q8_supported(M, F, D, dt) and os.environ.get("UVC_GELU_GRAD_BF16", "0") in ("", "0")
    aq = torch.empty(M, F, device=dev, dtype=torch.uint8)
    if q8:
        add("fc1+gelu,gelu'", "k_gemm_ws<unsigned short, unsigned short, 9", Lf, 7 * u, 2.0 * M * D * F,
            lambda: ops.gemm_nt(xb, W1, aq, dtype=dt, epilogue=ops.EPI_BIAS_GELU_GRAD_Q8, bias=bF, C2=uu), wfrac=6 / 7)
    else:
        add("fc1+gelu,gelu'", "k_gemm_ws<unsigned short, unsigned short, 7" if tiny else "k_gemm", Lf, 9 * u, 2.0 * M * D * F,
            lambda: ops.gemm_nt(xb, W1, aa, dtype=dt, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bF, C2=uu))
    if fuse_ln:
        add("fc2+resid+gate+norm1", "k_gemm_row384_lnbwd<true, 1>" if row384 else "k_gemm_wsn16_dma<4", Lf, 5 * u + 3 * ru, 2.0 * M * D * F,
            lambda: ops.gemm_nt(hF, W2, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID_GATE, bias=bD, R=xr, R2=gr_, gate=gate,
                                ln_gamma=gam, ln_beta=bet, ln_out=y, ln_mean=m_, ln_rstd=r_), wfrac=(ru + u) / (5 * u + 3 * ru))
    else:
        add("fc2+resid+gate", "k_gemm", Lf, 4 * u + 3 * ru, 2.0 * M * D * F,
            lambda: ops.gemm_nt(hF, W2, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID_GATE, bias=bD, R=xr, R2=gr_, gate=gate), wfrac=ru / (4 * u + 3 * ru))
    if with_teacher and not tiny:
        # wider models: the teacher's MLP is two GEMMs (the fused kernel exists for D = 192 only): fc1 + GELU (one output), fc2 + residual
        add("teacher fc1+gelu", "k_gemm", Lf, 5 * u, 2.0 * M * D * F,
            lambda: ops.gemm_nt(xb, W1, uu, dtype=dt, epilogue=ops.EPI_BIAS_GELU_OUT, bias=bF), wfrac=0.8)
        if row384:
            add("teacher fc2+resid+norm1", "k_gemm_row384_lnbwd<true, 1>", Lf, 5 * u + 2 * ru, 2.0 * M * D * F,
                lambda: ops.gemm_nt(hF, W2, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=bD, R=xr, ln_gamma=gam, ln_beta=bet, ln_out=y), wfrac=(ru + u) / (5 * u + 2 * ru))
        else:
            add("teacher fc2+resid", "k_gemm", Lf, 4 * u + 2 * ru, 2.0 * M * D * F,
                lambda: ops.gemm_nt(hF, W2, o32, dtype=dt, epilogue=ops.EPI_BIAS_RESID, bias=bD, R=xr), wfrac=ru / (4 * u + 2 * ru))
    if tiny and with_teacher:
        add("teacher mlp_fused+norm1", "k_mlp_fused", Lf, u + 2 * ru, 4.0 * M * D * F,
            lambda: ops.mlp_fused_fwd(xr, gam, bet, W1, bF, W2, bD, o32, next_gamma=gam, next_beta=bet, next_h=y), wfrac=(ru + u) / (u + 2 * ru))
    # ---- backward, main stream
    dA = torch.empty(M, F, device=dev, dtype=bf)
    if q8:
        aq.random_(0, 256)
        add("dfc2 x gelu'", "k_gemm_ws<unsigned short, unsigned short, 10", Lf, 7 * u, 2.0 * M * D * F,
            lambda: ops.gemm_nt(g16, W1, dA, dtype=dt, epilogue=ops.EPI_MUL_AUX_Q8, aux=aq, alpha_ptr=gate), wfrac=4 / 7)
    else:
        add("dfc2 x gelu'", "k_gemm_ws<unsigned short, unsigned short, 8" if tiny else "k_gemm", Lf, 9 * u, 2.0 * M * D * F,
            lambda: ops.gemm_nt(g16, W1, dA, dtype=dt, epilogue=ops.EPI_MUL_AUX, aux=hF, alpha_ptr=gate))
    dx16, add16 = torch.empty(M, D, device=dev, dtype=bf), g16.clone()
    part = torch.empty(max(ops.layernorm_bwd_blocks(M), 272) * (2 * D + 2), device=dev)
    dg, db, dots = torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(2, device=dev)
    q3 = rn(M, 3 * D).to(bf)
    if ops.gemm_lnbwd_supported(M, D, F, dt):
        W2t, Wqt = (rn(D, F) * .02).to(bf), (rn(D, 3 * D) * .02).to(bf)
        add("dfc1+ln2_bwd", "k_gemm_wsn_lnbwd_dma<24", Lf, 6 * u + ru, 2.0 * M * D * F,
            lambda: ops.gemm_nt_lnbwd(hF, W2t, xr, mean, rstd, gam, dx16, part, dg, db, add1=g16, a1=gate[1:]), wfrac=u / (6 * u + ru))
        add("dqkv+ln1_bwd", "k_gemm_wsn_lnbwd_dma<18", L, 6 * u + ru, 2.0 * M * D * 3 * D,
            lambda: ops.gemm_nt_lnbwd(q3, Wqt, xr, mean, rstd, gam, dx16, part, dg, db, add1=g16, add2=add16, a2=gate[:1], dots=dots), wfrac=u / (6 * u + ru))
    else:
        dH = torch.empty(M, D, device=dev, dtype=bf)
        Wt = (rn(D, 3 * D) * .02).to(bf)
        add("dfc1", "k_gemm", Lf, 5 * u, 2.0 * M * D * F, lambda: ops.gemm_nt(hF, W2, dH, dtype=dt, epilogue=ops.EPI_NONE))
        add("dqkv", "k_gemm", L, 4 * u, 2.0 * M * D * 3 * D, lambda: ops.gemm_nt(q3, Wt, dH, dtype=dt, epilogue=ops.EPI_NONE))
        add("ln_bwd", "k_ln_bwd_v", L + Lf, 3.5 * u + ru, 0,
            lambda: ops.layernorm_bwd(xb, xr, gam, mean, rstd, dx16, part, dg, db, M, D, dt, add1=g16, a1=gate[1:]))
    dH2 = torch.empty(M, D, device=dev, dtype=bf)
    add("dproj", "k_gemm_ws<unsigned short, unsigned short, 0" if tiny else "k_gemm", Lf, 2 * u, 2.0 * M * D * D,
        lambda: ops.gemm_nt(g16, Wp, dH2, dtype=dt, epilogue=ops.EPI_NONE))
    do = rn(B, N, D).to(bf)
    dq = torch.empty(B, N, 3 * D, device=dev, dtype=bf)
    dl = torch.empty(B, H, N, device=dev)
    # algorithmic bytes: every operand ONCE -- q, k, v, dO, O read (5 u), dq, dk, dv written (3 u).  r5: at N = 193 .. 200 in bf16 this is ONE kernel
    # (k_attn_bwd_one: S and dP once per tile pair, every operand crosses HBM once); the dq + dk/dv pair (other shapes, float32) moved 12 u
    # (PMC: 470 MB against 310 algorithmic at DeiT-Tiny batch 512).  The launch is the one the step makes (variant 0: at H <= 3 on 7/8 of the CUs)
    add("attn_bwd", "k_attn_bwd", Lf, 8 * u, afl * 2.5, lambda: ops.attention_bwd(qkv3, o, lse, do, dq, dl, B, N, H, dt))
    # ---- backward, weight-gradient stream (each entry = the split-M GEMM + its fixed-order reduction)
    ws = torch.empty(max(ops.gemm_tn_workspace_bytes(M, F, D), ops.gemm_tn_workspace_bytes(M, D, F), ops.gemm_tn_workspace_bytes(M, 3 * D, D), ops.gemm_tn_workspace_bytes(M, D, D)) // 4, device=dev)
    C1, C2, C3, C4 = torch.empty(D, F, device=dev), torch.empty(F, D, device=dev), torch.empty(D, D, device=dev), torch.empty(3 * D, D, device=dev)
    add("dW2 (+reduce)", "k_gemm_tn8p<6, 4" if tiny else "k_gemm_tn", Lf, 5 * u, 2.0 * M * D * F, lambda: ops.gemm_tn(g16, hF, C1, ws, dtype=dt))
    add("dW1 (+reduce)", "k_gemm_tn8p<8, 3" if tiny else "k_gemm_tn", Lf, 5 * u, 2.0 * M * D * F, lambda: ops.gemm_tn(hF, xb, C2, ws, dtype=dt))
    add("dWproj (+reduce)", "k_gemm_tn_dma<96, 192" if tiny else "k_gemm_tn", Lf, 2 * u, 2.0 * M * D * D, lambda: ops.gemm_tn(g16, xb, C3, ws, dtype=dt))
    add("dWqkv (+reduce)", "k_gemm_tn8p<6, 3" if tiny else "k_gemm_tn", L, 4 * u, 2.0 * M * D * 3 * D, lambda: ops.gemm_tn(q3, xb, C4, ws, dtype=dt))
    # ---- optimiser
    n = 5717440 if tiny else L * (4 * D * D + 2 * D * F)
    p, gr, m, v = (rn(n) for _ in range(4))
    v.abs_()
    pp, sq = torch.empty(1024, device=dev), torch.zeros(2, device=dev)
    add("clip+adamw", "k_adamw", 1, n * 28, 0, lambda: (ops.grad_sqnorm(gr, pp, sq), ops.adamw_step(p, gr, m, v, sq, lr=1e-4, step=3)))
    return rows


def timeit(fn, iters=20, warm=3):
    st = torch.cuda.current_stream()
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


# What HBM delivers by read : write mix on buffers beyond the Infinity Cache (tools/probe/hbm_mix_probe.hip, profiles/r2h_hbm_mix_probe.txt:
# flat 16-byte streams): write share of the bytes -> GB/s, linear in between
HBM_MIX_GBS = [(0.0, 5700.0), (0.2, 4850.0), (0.5, 5300.0), (0.8, 5050.0), (1.0, 4200.0)]


def hbm_ceiling_gbs(wfrac):
    if wfrac is None:
        return None
    for (x0, y0), (x1, y1) in zip(HBM_MIX_GBS, HBM_MIX_GBS[1:]):
        if wfrac <= x1:
            return round(y0 + (y1 - y0) * (wfrac - x0) / (x1 - x0), 0)
    return HBM_MIX_GBS[-1][1]


def measure(rows, iters=20):
    """HIP-event average launch time of every entry on the current stream (back-to-back launches)."""
    out = []
    for r in rows:
        ms = timeit(r["fn"], iters)
        out.append(dict(key=r["key"], rocprof=r["rocprof"], calls=r["calls"], us=round(ms * 1e3, 2), us_per_step=round(ms * 1e3 * r["calls"], 1),
                        bytes=r["bytes"], gbs=round(r["bytes"] / ms / 1e6, 1), tflops=round(r["flops"] / ms / 1e9, 1),
                        write_share=None if r.get("wfrac") is None else round(r["wfrac"], 3), hbm_mix_ceiling_gbs=hbm_ceiling_gbs(r.get("wfrac"))))
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    a = ap.parse_args()
    res = measure(build(a.B))
    print(f"{'kernel':24s} {'calls':>5s} {'us':>8s} {'us/step':>9s} {'GB/s':>8s} {'TFLOP/s':>8s}")
    for r in sorted(res, key=lambda r: -r["us_per_step"]):
        print(f"{r['key']:24s} {r['calls']:5d} {r['us']:8.1f} {r['us_per_step']:9.1f} {r['gbs']:8.1f} {r['tflops']:8.1f}")
    print("sum of stand-alone kernel time per step: %.2f ms" % (sum(r["us_per_step"] for r in res) / 1e3))
