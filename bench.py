#!/usr/bin/env python3
"""Benchmark of the UVC Stage-1 step on MI355X (BASELINE.json metric: images/sec, DeiT-Tiny,
budget 0.5, per-GPU batch 512, bf16 MFMA compute, 1/2/4/8 GPUs weak scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = the full UVC-train step of joint_train.py:395-450 on one batch of synthetic
ImageNet-shaped inputs already resident in HBM: student forward + teacher forward + distillation
loss + student backward + gradient all-reduce + clip + AdamW + scheduler + uvc_optimizer (prox, scores,
ranks, primal/dual update).  Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for how
`roofline` and `cpu_baseline` are defined.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

_real_stdout = sys.stdout

# algorithmic FLOPs per image per step = 2*(3E + 4*L*Bk + 4*C)
GFLOP_PER_IMG = {"deit_tiny_patch16_224": 9.972, "deit_small_patch16_224": 36.675, "deit_base_patch16_224": 140.28,
                 "t2t_vit_14": 37.42}      # T2T: 2*(3E + 4*L*Bk + 4*C) with E = 256,647,680; the tokens-to-token dgrad (<= 0.51) not counted
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE config 2)")
    p.add_argument("--model_type", default="deit_tiny_patch16_224")
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--stage", type=int, default=1, choices=[1, 2],
                   help="1 = the headline Stage-1 UVC-train step; 2 = the Stage-2 masked fine-tune step (SURVEY §8 f-1)")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--phase", default="train", choices=["train", "warmup"],
                   help="stage 1: 'train' = the UVC-train step the metric is quoted on; 'warmup' = the warm-up-phase step (gates fixed at .5/.5, "
                        "gate logits frozen, lr = warmup_lr), reported for reference (SURVEY 8d)")
    p.add_argument("--compact_mlp", type=int, default=1, help="stage 2: skip pruned MLP hidden units (0 = dense masked computation)")
    p.add_argument("--serialize", type=int, default=0, help="diagnostic bit mask: 1 = weight gradients on the main stream, 2 = teacher forward on the main stream (3 = no overlap at all)")
    p.add_argument("--cpu_steps", type=int, default=6)
    return p.parse_args()


def pruned_state(tr, seed=731):
    """Non-trivial primal/dual start so prox / rank / mask kernels do real work."""
    import numpy as np
    mm = tr.minimax
    L, H, F = mm.n_layers, mm.num_heads, mm.dims.F
    rs = np.random.RandomState(seed)
    s = np.zeros((L, 2), np.float32)
    s[:, 0] = rs.uniform(0, 0.6 * (H - 1) + 0.3, L)
    s[:, 1] = rs.uniform(0, 0.5 * F, L)
    r = rs.uniform(0, 30.0, (L, H)).astype(np.float32)
    mm.s.data.copy_(torch.from_numpy(s)); mm.r.data.copy_(torch.from_numpy(r))
    mm.y.data.fill_(1.0); mm.p.data.fill_(1.0); mm.z.data.fill_(2.0)


def stage2_checkpoint_state(model, seed=732, skip_blocks=(4, 9)):
    """A finished-Stage-1-like state for the Stage-2 bench: structured masks at roughly the budget-0.5 operating point
    (one head of three dropped + a quarter of the remaining proj input columns, half of the MLP hidden units) and
    two hard-skipped blocks."""
    import numpy as np
    rs = np.random.RandomState(seed)
    cfg = model._cfg
    D, F, H = cfg.embed_dim, cfg.hidden, cfg.num_heads
    hd = D // H
    dev = model._flat.device
    with torch.no_grad():
        for l, blk in enumerate(model.blocks):
            kp = np.ones(D, np.float32)
            dead = rs.randint(0, H)
            kp[dead * hd:(dead + 1) * hd] = 0
            for h in range(H):
                if h != dead:
                    kp[h * hd + rs.choice(hd, size=hd // 4, replace=False)] = 0
            kh = np.ones(F, np.float32)
            kh[rs.choice(F, size=F // 2, replace=False)] = 0
            kp, kh = torch.from_numpy(kp).to(dev), torch.from_numpy(kh).to(dev)
            blk.attn.proj.mask.copy_(kp[None, :].expand(D, -1))
            blk.mlp.fc2.mask.copy_(kh[None, :].expand(D, -1))
            blk.mlp.fc1.mask.copy_(kh[:, None].expand(-1, D))
        g = torch.tensor([-1.0, 1.0]).repeat(cfg.depth, 1)
        for b in skip_blocks:
            g[b] = torch.tensor([1.0, -1.0])
        model.block_skip_gating.data.copy_(g.to(dev))


def kernel_roofline(tr, args, iters=30):
    """The largest GEMM kernel of the student's forward/backward chain: fc1 with its fused epilogue (bias, GELU and GELU' -- the
    two [M, F] outputs the backward needs), as the step launches it: algorithmic bytes of one launch / its average duration
    measured with HIP events on the launch stream."""
    from uvc_amd import ops
    m = tr.model
    cfg = m._cfg
    B = args.batch
    N = (cfg.img_size // cfg.patch_size) ** 2 + cfg.ntok
    M, D, F = B * N, cfg.embed_dim, cfg.hidden
    dt = torch.bfloat16 if args.precision == "bf16" else torch.float32
    dev = m._flat.device
    A = torch.randn(M, D, device=dev).to(dt)
    W = (torch.randn(F, D, device=dev) * 0.02).to(dt)
    bias = torch.zeros(F, device=dev)
    a_out, u_out = torch.empty(M, F, device=dev, dtype=dt), torch.empty(M, F, device=dev, dtype=dt)
    dtype = ops.UVC_BF16 if args.precision == "bf16" else ops.UVC_F32
    for _ in range(3):
        ops.gemm_nt(A, W, a_out, dtype=dtype, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bias, C2=u_out)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        ops.gemm_nt(A, W, a_out, dtype=dtype, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bias, C2=u_out)
    e1.record(st)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * M * D * F
    esz = 2 if args.precision == "bf16" else 4
    bytes_alg = (M * D + F * D + 2 * M * F) * esz          # A once + W once + the two outputs (GELU'(a), GELU(a)) once
    gbs = bytes_alg / (ms * 1e-3) / 1e9
    tf = flops / (ms * 1e-3) / 1e12
    mfma_peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3
    # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/
    # r1i_pmc_hbm_traffic_kbench.csv); only valid for the profiled configuration
    traffic = 352572549 if (args.precision == "bf16" and args.batch == 512 and args.model_type == "deit_tiny_patch16_224") else None
    # intensity 2*M*D*F / bytes = 85 flop/B << the ~400 flop/B ridge: this kernel's roofline is HBM
    kname = "k_gemm_ws<bf16,bf16,EPI_BIAS_GELU_GRAD,6>" if D == 192 else "uvc_gemm_nt"
    return {"bound": "hbm", "kernel": kname + " (mlp.fc1 + bias + GELU and GELU', M=%d K=%d N=%d)" % (M, D, F),
            "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": traffic, "algorithmic_bytes": bytes_alg, "launch_ms": round(ms, 4),
            "mfma_tflops": round(tf, 2), "mfma_frac": round(tf / mfma_peak, 4)}


def cpu_baseline(args):
    """The oracle (a PyTorch-CPU restatement of the reference step, pinned to the reference's golden
    vectors) timed on the host cores on a bounded sample: DeiT-Tiny, batch 8, a few steps."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import scenarios as SC
    from helpers import build_oracle, load_golden, split_draws
    from oracle import step as OS
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    name = "tiny8_pruned"
    gold = load_golden(name)
    r, S = build_oracle(name)
    x_all, y_all = SC.make_inputs(r)
    md, e1, e2 = split_draws(r, gold, 0, S.cfg.depth)
    x, y = torch.from_numpy(x_all[0]), torch.from_numpy(y_all[0])
    OS.stage1_step(S, x, y, list(md), e1, e2)          # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < args.cpu_steps and time.perf_counter() - t0 < 40:
        OS.stage1_step(S, x, y, list(md), e1, e2)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(8 * n / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle Stage-1 step, DeiT-Tiny batch 8, {n} steps, torch CPU fp32 {threads} threads"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    torch.manual_seed(730)
    if args.stage == 2:
        from uvc_amd.post_train import Stage2Trainer, default_args
        a = default_args(model_type=args.model_type, precision=args.precision, train_batch_size=args.batch, local_rank=local)
        tr = Stage2Trainer(a, device=f"cuda:{local}", distributed=world > 1, world_size=world)
        stage2_checkpoint_state(tr.model)
        tr.mlp_widths = tr.model.set_mlp_compaction() if args.compact_mlp else tr.model.set_mlp_compaction(False)
        tr.begin_epoch(a.warmup_epochs + 1)
    else:
        from uvc_amd.stage1 import Stage1Trainer, default_args
        a = default_args(model_type=args.model_type, precision=args.precision, train_batch_size=args.batch, local_rank=local)
        tr = Stage1Trainer(a, device=f"cuda:{local}", distributed=world > 1)
        pruned_state(tr)
        tr.begin_epoch(a.warmup_epochs + 1 if args.phase == "train" else 1)      # UVC-train phase (post warm-up) is the metric, SURVEY.md §8d
    if args.serialize & 1:
        tr.model.two_stream_backward = False
    if args.serialize & 2:
        a.overlap_teacher = 0
    dev = torch.device("cuda", local)
    g = torch.Generator(device=dev).manual_seed(730 + rank)
    x = torch.randn(args.batch, 3, a.img_size, a.img_size, device=dev, generator=g)
    y = torch.softmax(torch.randn(args.batch, a.num_classes, device=dev, generator=g), -1)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        tr.step(x, y)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.step(x, y)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss = float(out["loss"])
    if rank == 0:
        imgs = world * args.batch * args.steps / dt
        gf = GFLOP_PER_IMG.get(args.model_type)
        tiny = args.model_type == "deit_tiny_patch16_224"
        metric = (("images/sec UVC Stage-1 step, DeiT-Tiny budget=0.5" if tiny else f"images/sec UVC Stage-1 step, {args.model_type} budget=0.5 (not the headline model)")
                  if args.phase == "train" else
                  "images/sec UVC Stage-1 WARM-UP-phase step, DeiT-Tiny (for reference, not the headline)") if args.stage == 1 else \
                 "images/sec UVC Stage-2 masked fine-tune step, DeiT-Tiny (SURVEY 8 f-1, not the headline)"
        line = {"metric": metric, "value": round(imgs, 1), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
                "config": {"workload": (f"{args.model_type} Stage-1 UVC-train step, budget 0.5, per-GPU batch {args.batch}, "
                                        f"224x224x3 synthetic, soft distillation alpha 0.1, block gating on") if args.stage == 1 else
                                       (f"{args.model_type} Stage-2 masked fine-tune step, per-GPU batch {args.batch}, masks at the "
                                        f"budget-0.5 operating point, 2 of 12 blocks hard-skipped, soft distillation alpha 0.1"),
                           "global_batch": world * args.batch, "parallelism": f"dp{world}"},
                "step_tflops_per_gpu": round(imgs / world * gf / 1e3, 2) if gf else None,
                "step_frac_of_bf16_mfma_peak": round(imgs / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4) if gf else None,
                "final_loss": round(loss, 4)}
        if args.stage == 1:
            line["cur_resource"] = round(float(out["cur"]), 4)
        else:   # Stage-2 FLOPs per image: teacher 1x forward, student (fwd + bwd) only over the blocks that run
            line.pop("step_tflops_per_gpu"); line.pop("step_frac_of_bf16_mfma_peak")
        line["roofline"] = kernel_roofline(tr, args)
        if world == 1 and not args.no_cpu_baseline and args.stage == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), file=_real_stdout, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    # stdout carries exactly ONE line, the JSON record: the library's progress prints (FLOP size, eps updates, gating banners)
    # are the reference's own messages and go to stderr here
    import contextlib
    _real_stdout = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        main()
