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
# Of the last block only the class-token row reaches the head, so the engine runs that block's attention output, proj, LayerNorm2, MLP
# and their backward on B rows instead of B*N (uvc_vit_io.full_tail = 0, NOTEBOOK.md section 5b): the same loss, logits and gradients
# with 4 * (2 N^2 D + N D^2 + 2 N D F) * (1 - 1/N) fewer multiply-adds per image (student + teacher forward, 2x backward).  The
# TFLOP/s figures of the JSON line use the EXECUTED count; `--full_tail 1` runs every row as the reference does.
def executed_gflop_per_img(model_type, full_tail):
    gf = GFLOP_PER_IMG.get(model_type)
    dims = {"deit_tiny_patch16_224": (192, 197, 768), "deit_small_patch16_224": (384, 197, 1536), "deit_base_patch16_224": (768, 197, 3072),
            "t2t_vit_14": (384, 197, 1152)}.get(model_type)
    if gf is None or dims is None or full_tail:
        return gf
    D, N, F = dims
    saved_macs = 4.0 * (2.0 * N * N * D + N * D * D + 2.0 * N * D * F) * (1.0 - 1.0 / N)
    return gf - 2.0 * saved_macs / 1e9


PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)       # >= 20 warm-up iterations discarded, >= 100 timed
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--batch", type=int, default=512, help="per-GPU batch (BASELINE config 2)")
    p.add_argument("--model_type", default="deit_tiny_patch16_224")
    p.add_argument("--precision", default="bf16", choices=["bf16", "bf16_f32resid", "fp32"],
                   help="bf16 = the throughput mode (bf16 operands AND bf16 residual-stream rows); bf16_f32resid = rounds 1-2's float32 residual rows (A/B)")
    p.add_argument("--stage", type=int, default=1, choices=[1, 2],
                   help="1 = the headline Stage-1 UVC-train step; 2 = the Stage-2 masked fine-tune step (SURVEY §8 f-1)")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_mode_legs", action="store_true", help="skip the short fp32 / bf16_f32resid legs behind the timed region (--no_cpu_baseline, the A/B scripts' switch, skips them too)")
    p.add_argument("--no_next_batch", action="store_true", help="A/B: do not hand the next batch to Stage1Trainer.step (its teacher forward then starts with the step)")
    p.add_argument("--phase", default="train", choices=["train", "warmup"],
                   help="stage 1: 'train' = the UVC-train step the metric is quoted on; 'warmup' = the warm-up-phase step (gates fixed at .5/.5, "
                        "gate logits frozen, lr = warmup_lr), reported for reference (SURVEY 8d)")
    p.add_argument("--compact_mlp", type=int, default=1, help="stage 2: skip pruned MLP hidden units (0 = dense masked computation)")
    p.add_argument("--serialize", type=int, default=0, help="diagnostic bit mask: 1 = weight gradients on the main stream, 2 = teacher forward on the main stream (3 = no overlap at all)")
    p.add_argument("--cpu_steps", type=int, default=10, help="timed CPU-baseline steps at batch 8 after 3 warm-ups; the median is reported (BASELINE.md section 3)")
    # the reference's own switches (run_uvc_train.sh:4-38, joint_train.py:684-879), so that BASELINE.json's configs 3 / 4 / 5 are benched AS STATED:
    p.add_argument("--budget", type=float, default=0.5, help="FLOPs budget (config 3: 0.58)")
    p.add_argument("--enable_deit", type=int, default=0, help="1: the distillation token and the second head (config 4: N = 198)")
    p.add_argument("--enable_patch_gating", type=int, default=0, help="1: Gumbel top-k patch gating (config 5)")
    p.add_argument("--patch_ratio", type=float, default=0.9)
    p.add_argument("--enable_block_gating", type=int, default=1)
    p.add_argument("--full_tail", type=int, default=0, help="1: the last block computes all B*N rows like the reference (default: its token rows only; same outputs)")
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


def _git_head():
    try:
        import subprocess
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:        # a snapshot without .git (the GPU box): the commit the snapshot was taken from, if the sender left it behind
        return open(os.path.join(ROOT, ".commit_for_profiles")).read().strip() or None
    except Exception:
        return None


def _profile_file(kind, pattern, args=None):
    """The committed profile file of `kind` that the JSON line may cite.  profiles/MANIFEST.json (written by tools/profiles_manifest.py when a round's
    profile run is copied into profiles/) NAMES the current file of every kind together with the model / batch it was measured on; without a manifest the
    newest file by modification time is taken.  (Round 5 picked `sorted(glob)[-1]`: lexicographically r5zz > r5end, so the line cited a superseded pass.)
    A PMC pass is a measurement of ONE model's kernels: with `args` given, a file measured on another model or batch is not returned (the Tiny
    pass's mfma_busy_frac appeared in the DeiT-Small / Base / T2T lines, looked up by kernel key)."""
    import glob
    man = {}
    try:
        man = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
    except Exception:
        pass
    ent = man.get(kind)
    path = None
    if ent and os.path.exists(os.path.join(ROOT, "profiles", ent["file"])):
        path = os.path.join(ROOT, "profiles", ent["file"])
        model, batch = ent.get("model", "deit_tiny_patch16_224"), ent.get("batch", 512)
    else:
        files = glob.glob(os.path.join(ROOT, "profiles", pattern))
        if files:
            path = max(files, key=os.path.getmtime)
            model, batch = "deit_tiny_patch16_224", 512          # tools/kernel_table.py's and bench.py's defaults, which every un-manifested pass ran
    if path is None:
        return None
    if args is not None and (model != args.model_type or int(batch) != int(args.batch) or args.enable_deit or args.enable_patch_gating):
        return None
    return path


def pmc_traffic(key, args=None):
    """HBM bytes per launch of kernel `key` from the current PMC pass committed under profiles/ (tools/pmc_traffic.py writes
    profiles/<round>_pmc_traffic.json from two rocprofv3 --pmc runs, FETCH_SIZE and WRITE_SIZE, with the guide's gfx950 correction).
    None when no pass covers the kernel (or the pass is another model's): the number is a measurement, never a constant in this file."""
    f = _profile_file("pmc_traffic", "*_pmc_traffic.json", args)
    if f is None:
        return None
    try:
        d = json.load(open(f))
    except Exception:
        return None
    ent = d.get("kernels", {}).get(key)
    return dict(bytes=int(ent["hbm_bytes"]), source=os.path.basename(f), commit=d.get("commit")) if ent else None


def pmc_mfma(args=None):
    """Matrix-pipe utilisation per kernel key from the current rocprofv3 SQ pass committed under profiles/ (tools/pmc_mfma.py:
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)); {} when no pass of THIS model is committed."""
    f = _profile_file("pmc_mfma", "*_pmc_mfma.json", args)
    if f is None:
        return {}, None
    try:
        d = json.load(open(f))
    except Exception:
        return {}, None
    return d.get("kernels", {}), f"{os.path.basename(f)} @ {d.get('commit')}"


def in_step_top(args=None):
    """Row 1 of the current committed in-step rocprofv3 table (profiles/*_kernel_stats_uvc_train_steps_only.csv): the kernel with the
    largest summed duration INSIDE the step.  Reported next to `roofline` because the two rankings differ by construction: the step
    runs two or three streams that time-slice one memory system, so in-step durations are inflated by whatever ran beside the
    kernel (k_tn_reduce: ~10 us alone, 64 us beside the dgrad stream) and do not add up to the step time, while the stand-alone
    launch times do (their sum per step equals the measured step time to ~2 %: `kernel_ms_per_step_standalone_sum`)."""
    import csv
    f = _profile_file("steps_only", "*_kernel_stats_uvc_train_steps_only.csv", args)
    if f is None:
        return None
    rows = [r for r in csv.reader(l for l in open(f) if not l.startswith("#"))]
    if len(rows) < 2:
        return None
    r = rows[1]
    return {"kernel": r[0], "calls_per_step": float(r[1]), "avg_us_in_step": float(r[2]), "ms_per_step_in_step": float(r[3]),
            "percent_of_kernel_time": float(r[4]), "source": "profiles/" + os.path.basename(f)}


def kernel_table(args):
    """Every kernel of the step as a stand-alone launch at the step's shapes (tools/kernel_table.py): HIP-event average of
    back-to-back launches on the launch stream x launches per step.  `roofline` is the entry with the LARGEST TOTAL TIME PER STEP --
    the dominant kernel, not the most flattering one; the ten largest go into `top_kernels`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_table as KT
    # T2T-ViT-14: the 14 blocks (D = 384, mlp_ratio 3) -- the tokens-to-token front end is not in the stand-alone table, its kernels
    # are in the rocprofv3 tables under profiles/
    D, H, L, F = {"deit_tiny_patch16_224": (192, 3, 12, 768), "deit_small_patch16_224": (384, 6, 12, 1536), "deit_base_patch16_224": (768, 12, 12, 3072),
                  "t2t_vit_14": (384, 6, 14, 1152)}[args.model_type]
    rows = KT.measure(KT.build(args.batch, D=D, H=H, L=L, N=198 if args.enable_deit else 197, F=F, tail=not args.full_tail,
                               resid_f32=(args.precision == "bf16_f32resid") or None), iters=20)
    rows.sort(key=lambda r: -r["us_per_step"])
    top = rows[0]
    ridge = PEAK_BF16_TFLOPS * 1e3 / PEAK_HBM_GBS          # flop per byte where the two roofs meet (312)
    for r in rows:
        inten = (r["tflops"] * 1e12) / (r["gbs"] * 1e9) if r["gbs"] else 0.0
        r["bound"] = "mfma" if inten > ridge else "hbm"
        r["frac"] = round((r["tflops"] / PEAK_BF16_TFLOPS) if r["bound"] == "mfma" else (r["gbs"] / PEAK_HBM_GBS), 4)
    tr = pmc_traffic(top["key"], args)
    roof = {"bound": top["bound"], "kernel": f"{top['key']} ({top['rocprof']})", "calls_per_step": top["calls"],
            "achieved": top["gbs"] if top["bound"] == "hbm" else top["tflops"], "peak": PEAK_HBM_GBS if top["bound"] == "hbm" else PEAK_BF16_TFLOPS,
            "unit": "GB/s" if top["bound"] == "hbm" else "TFLOP/s", "frac": top["frac"], "traffic": tr["bytes"] if tr else None,
            "traffic_source": (f"{tr['source']} @ {tr['commit']}" if tr else None), "algorithmic_bytes": top["bytes"],
            "launch_us": top["us"], "us_per_step": top["us_per_step"], "mfma_tflops": top["tflops"],
            "selection": "largest stand-alone launch time x launches per step among the step's kernels (tools/kernel_table.py)"}
    for r in rows:       # against what HBM delivers for the kernel's read : write mix beyond the Infinity Cache (profiles/r2h_hbm_mix_probe.txt)
        r["frac_of_mix_ceiling"] = round(r["gbs"] / r["hbm_mix_ceiling_gbs"], 4) if r.get("hbm_mix_ceiling_gbs") and r["bound"] == "hbm" else None
    roof["hbm_mix_ceiling"] = top.get("hbm_mix_ceiling_gbs")
    roof["frac_of_mix_ceiling"] = top.get("frac_of_mix_ceiling")
    mf, mf_src = pmc_mfma(args)
    for r in rows:
        e = mf.get(r["key"])
        r["mfma_busy_frac"] = e.get("mfma_busy_frac") if e else None
    roof["mfma_busy_frac"] = top.get("mfma_busy_frac")
    roof["mfma_busy_source"] = mf_src
    keep = ("key", "calls", "us", "us_per_step", "bytes", "gbs", "tflops", "bound", "frac", "write_share", "hbm_mix_ceiling_gbs", "frac_of_mix_ceiling",
            "mfma_busy_frac")
    return roof, [{k: r[k] for k in keep} for r in rows[:10]], round(sum(r["us_per_step"] for r in rows) / 1e3, 2)


def cpu_baseline(args):
    """The oracle (a PyTorch-CPU restatement of the reference step, pinned to the reference's golden vectors) timed on the host cores
    on a bounded sample: DeiT-Tiny, batch 8 (BASELINE config 1) and batch 64, a few steps each, with the time inside the
    uvc_optimizer restatement split out (the reference spends 74-93 % of it in weight_list_to_scores)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import scenarios as SC
    from helpers import build_oracle_from_recipe, load_golden, split_draws
    from oracle import step as OS
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    gold = load_golden("tiny8_pruned")
    acc = {"uvc": 0.0}
    orig = OS.U.uvc_update

    def timed(*a, **k):
        t = time.perf_counter()
        r = orig(*a, **k)
        acc["uvc"] += time.perf_counter() - t
        return r

    OS.U.uvc_update = timed
    points = []
    try:
        for batch, warm, max_steps, budget_s in ((8, 3, max(10, args.cpu_steps), 40.0), (64, 1, 3, 25.0)):
            r = SC.recipe("tiny8_pruned")
            r["batch"], r["steps"] = batch, 1
            r0, S = build_oracle_from_recipe(r)
            x_all, y_all = SC.make_inputs(r0)
            md, e1, e2 = split_draws(SC.recipe("tiny8_pruned"), gold, 0, S.cfg.depth)
            x, y = torch.from_numpy(x_all[0]), torch.from_numpy(y_all[0])
            for _ in range(warm):
                OS.stage1_step(S, x, y, list(md), e1, e2)      # warm-ups (3 at batch 8)
            acc["uvc"] = 0.0
            t0 = time.perf_counter()
            per = []
            while len(per) < max_steps and (time.perf_counter() - t0 < budget_s or len(per) < 3):
                ts = time.perf_counter()
                OS.stage1_step(S, x, y, list(md), e1, e2)
                per.append(time.perf_counter() - ts)
            n, dt = len(per), sum(per)
            med = sorted(per)[n // 2]                            # the MEDIAN step is the value (the mean wandered 12.3 -> 10.9 img/s between rounds on identical code)
            points.append({"batch": batch, "warmup_steps": warm, "steps": n, "images_per_sec": round(batch / med, 3), "s_per_step_median": round(med, 3),
                           "s_per_step_mean": round(dt / n, 3), "s_per_step_min": round(min(per), 3), "s_per_step_max": round(max(per), 3),
                           "uvc_update_s_per_step": round(acc["uvc"] / n, 3), "model_part_s_per_step": round((dt - acc["uvc"]) / n, 3)})
    finally:
        OS.U.uvc_update = orig
    b8 = points[0]
    return {"value": b8["images_per_sec"], "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle Stage-1 step (student fwd/bwd + teacher fwd + loss + clip + AdamW + uvc_optimizer), DeiT-Tiny, torch CPU fp32 {threads} threads: "
                      f"batch 8, 3 warm-ups + {b8['steps']} timed steps, median (value); batch 64, 1 + {points[1]['steps']} steps", "points": points}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # One rank per GPU over RCCL ("nccl") is the contract.  UVC_BENCH_BACKEND=gloo with UVC_BENCH_SHARE_DEVICE=1 puts every rank on
    # device 0 over gloo (RCCL refuses two ranks on one device): no scaling number, but the multi-rank branches of this file -- bucketed
    # all-reduce overlapped with the backward, MAX over ranks, ranks_seen, exposed_allreduce_ms_per_step -- run on a one-GPU box
    # (tests/test_ddp_gpu.py::test_bench_two_ranks_on_one_device).
    backend = os.environ.get("UVC_BENCH_BACKEND", "nccl")
    if os.environ.get("UVC_BENCH_SHARE_DEVICE", "0") not in ("", "0"):
        local = 0
    torch.cuda.set_device(local)
    pinned = None
    if world > 1:
        # one slice of the host cores per rank (the enqueue thread of a rank must not migrate or share a core with another rank's: with 8
        # ranks on one host the ~3 ms of enqueue per 12-ms step is the first thing that can surface), and as many intra-op threads
        ncpu = os.cpu_count() or 1
        per = max(1, ncpu // world)
        try:
            avail = sorted(os.sched_getaffinity(0))
            per = max(1, len(avail) // world)
            mine = avail[(int(os.environ.get("LOCAL_RANK", 0)) * per) % len(avail):][:per] or avail
            os.sched_setaffinity(0, mine)
            pinned = [mine[0], mine[-1]]
        except (AttributeError, OSError):
            pass
        torch.set_num_threads(max(1, min(per, 8)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    torch.manual_seed(730)
    if args.stage == 2:
        from uvc_amd.post_train import Stage2Trainer, default_args
        a = default_args(model_type=args.model_type, precision=args.precision, train_batch_size=args.batch, local_rank=local)
        tr = Stage2Trainer(a, device=f"cuda:{local}", distributed=world > 1, world_size=world)
        stage2_checkpoint_state(tr.model)
        tr.mlp_widths = tr.model.set_mlp_compaction() if args.compact_mlp else tr.model.set_mlp_compaction(False)
        # (the masks were set after the trainer derived its tables from the checkpoint: derive the pruned-head table again)
        tr.head_keep = tr.model.set_head_skipping() if args.compact_mlp else tr.model.set_head_skipping(False)
        tr.model.skip_pruned_head_grads = bool(args.compact_mlp) and tr.head_keep is not None
        tr.begin_epoch(a.warmup_epochs + 1)
    else:
        from uvc_amd.stage1 import Stage1Trainer, default_args
        a = default_args(model_type=args.model_type, precision=args.precision, train_batch_size=args.batch, local_rank=local, budget=args.budget,
                         enable_deit=args.enable_deit, enable_patch_gating=args.enable_patch_gating, patch_ratio=args.patch_ratio,
                         enable_block_gating=args.enable_block_gating)
        tr = Stage1Trainer(a, device=f"cuda:{local}", distributed=world > 1)
        pruned_state(tr)
        tr.begin_epoch(a.warmup_epochs + 1 if args.phase == "train" else 1)      # UVC-train phase (post warm-up) is the metric, SURVEY.md §8d
    if args.full_tail:
        tr.model.full_tail = True
        if getattr(tr, "teacher", None) is not None:
            tr.teacher.full_tail = True
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

    # next_x: the trainer starts the NEXT step's teacher forward behind this step's backward (under the optimizer / UVC tail) when the caller already holds the
    # next batch, as a prefetching loader does (Stage1Trainer.step; uvc_amd.cli reads one batch ahead).  The last timed step promises nothing, so the K timed
    # steps contain exactly K teacher forwards: the first one's was started by the last warm-up step, the others' inside the region.
    nx = dict(next_x=x) if args.stage == 1 and not args.no_next_batch else {}
    for _ in range(args.warmup):
        tr.step(x, y, **nx)
    sync()
    # K timed steps between two barrier + synchronize points (the contract's wall clock); an event at every step boundary on the
    # launch stream gives the per-step device times (median / p10 / p90) without a host sync inside the timed region
    st = torch.cuda.current_stream()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record(st)
    host = []
    for i in range(args.steps):
        h0 = time.perf_counter()
        out = tr.step(x, y, **(nx if i + 1 < args.steps else {}))
        evs[i + 1].record(st)
        host.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0                 # the host has enqueued everything; the GPU is still running if the host runs ahead
    sync()
    dt = time.perf_counter() - t0
    host_ms = sorted(host)[len(host) // 2] * 1e3
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]  # noqa: E731
    exposed_ms = None
    ranks_seen, backend_name = (dist.get_world_size(), dist.get_backend()) if world > 1 else (1, None)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        exposed_ms = tr.ddp.exposed_comm_ms() if tr.ddp is not None else None
        hm = torch.tensor([host_ms, t_enq], device=dev, dtype=torch.float64)
        dist.all_reduce(hm, op=dist.ReduceOp.MAX)
        host_ms, t_enq = float(hm[0]), float(hm[1])
        # the ranks other than 0 are DONE here: rank 0's stand-alone kernel table below must not keep N - 1 GPUs waiting on a barrier
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
    loss = float(out["loss"])
    if rank == 0:
        imgs = world * args.batch * args.steps / dt
        gf = executed_gflop_per_img(args.model_type, bool(args.full_tail))
        tiny = args.model_type == "deit_tiny_patch16_224"
        headline = tiny and args.budget == 0.5 and not args.enable_deit and not args.enable_patch_gating and args.enable_block_gating
        opts = (f"budget={args.budget:g}" + (", distillation token (enable_deit)" if args.enable_deit else "")
                + (f", patch gating (ratio {args.patch_ratio:g})" if args.enable_patch_gating else "")
                + (", block gating" if args.enable_block_gating else ", no block gating"))
        metric = (("images/sec UVC Stage-1 step, DeiT-Tiny budget=0.5" if headline else f"images/sec UVC Stage-1 step, {args.model_type} {opts} (not the headline configuration)")
                  if args.phase == "train" else
                  "images/sec UVC Stage-1 WARM-UP-phase step, DeiT-Tiny (for reference, not the headline)") if args.stage == 1 else \
                 "images/sec UVC Stage-2 masked fine-tune step, DeiT-Tiny (SURVEY 8 f-1, not the headline)"
        line = {"metric": metric, "value": round(imgs, 1), "unit": "images/sec",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                "ms_per_step_device": {"median": round(pct(0.5), 3), "p10": round(pct(0.1), 3), "p90": round(pct(0.9), 3)},
                "ranks_seen": ranks_seen, "backend": backend_name,
                # host side: MEDIAN wall time of one tr.step() call (enqueue only, nothing synchronises inside the timed region), MAX over ranks: what a
                # step costs the host when it is not held back; and the enqueue loop's share of the timed region -- that one INCLUDES the time the runtime
                # blocks the launching thread on a full queue (a host that runs ahead is throttled to the GPU's pace: a share near 1 with a median far
                # below ms_per_step means "ran ahead until the queue was full", not "host-bound")
                "host_enqueue_ms_per_step": round(host_ms, 3), "host_enqueue_loop_share_of_wall": round(t_enq / dt, 3), "host_cores_pinned": pinned,
                # every timed step runs one teacher forward; with next_x it is the NEXT step's batch, started behind this step's backward (Stage1Trainer.step)
                "teacher_forward": "next batch's, started behind the backward (K per K timed steps)" if nx else "this batch's, started with the step",
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16" if args.precision.startswith("bf16") else "f32", "data": "synthetic",
                "config": {"workload": (f"{args.model_type} Stage-1 UVC-train step, budget {args.budget:g}, per-GPU batch {args.batch}, "
                                        f"224x224x3 synthetic, soft distillation alpha 0.1, block gating {'on' if args.enable_block_gating else 'off'}"
                                        + (", distillation token (N = 198, two heads)" if args.enable_deit else "")
                                        + (f", patch gating on (patch_ratio {args.patch_ratio:g})" if args.enable_patch_gating else "")) if args.stage == 1 else
                                       (f"{args.model_type} Stage-2 masked fine-tune step, per-GPU batch {args.batch}, masks at the "
                                        f"budget-0.5 operating point, 2 of 12 blocks hard-skipped, soft distillation alpha 0.1"),
                           "global_batch": world * args.batch, "parallelism": f"dp{world}"},
                "step_tflops_per_gpu": round(imgs / world * gf / 1e3, 2) if gf else None,
                "step_frac_of_bf16_mfma_peak": round(imgs / world * gf / 1e3 / PEAK_BF16_TFLOPS, 4) if gf else None,
                "residual_stream": "float32" if (args.precision != "bf16" or os.environ.get("UVC_RESID_F32", "0") not in ("", "0")) else "bf16",
                # what fc1 leaves for the backward (r6): GELU(a) in bf16 (an MFMA operand of fc2 and dW2) and GELU'(a) as one byte per activation where the streaming
                # kernel of DeiT-Tiny's width runs (uvc_vit_io.gelu_grad_bf16 = 0; include/uvc_kernels.h: uniform code over [-0.13, 1.13], |error| <= 2.47e-3)
                "hidden_tensors": ("GELU(a) bf16 + GELU'(a) one byte (uniform code, |err| <= 2.47e-3)" if (args.precision == "bf16" and tiny and args.batch * 197 >= 4096
                                    and os.environ.get("UVC_GELU_GRAD_BF16", "0") in ("", "0")) else "GELU(a), GELU'(a) in the compute dtype"),
                "final_loss": round(loss, 4)}
        if args.stage == 1:
            line["cur_resource"] = round(float(out["cur"]), 4)
            line["gflop_per_image"] = {"model": GFLOP_PER_IMG.get(args.model_type), "executed": round(gf, 3) if gf else None,
                                       "note": "executed < model: the last block's rows that never reach the head are not computed (full_tail=0); TFLOP/s uses executed"}
        else:   # Stage-2 FLOPs per image: teacher 1x forward, student (fwd + bwd) only over the blocks that run
            line.pop("step_tflops_per_gpu"); line.pop("step_frac_of_bf16_mfma_peak")
        if exposed_ms is not None:
            line["exposed_allreduce_ms_per_step"] = round(exposed_ms, 3)       # main stream stalled in reducer.finish(), event-timed
        if args.stage == 1 and args.phase == "train" and world == 1 and not args.full_tail:
            # the same step with every row of the last block computed, as the reference executes it (identical outputs; A/B of the
            # dead-row elimination): 5 + 30 steps
            models = [tr.model] + ([tr.teacher] if getattr(tr, "teacher", None) is not None else [])
            for m_ in models:
                m_.full_tail = True
            for _ in range(5):
                tr.step(x, y)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for _ in range(30):
                tr.step(x, y)
            torch.cuda.synchronize()
            line["images_per_sec_all_rows_of_last_block"] = round(30 * args.batch / (time.perf_counter() - tw), 1)
            for m_ in models:
                m_.full_tail = False
        if args.stage == 1 and args.phase == "train" and world == 1:
            # the warm-up-phase step for reference: gates fixed at .5/.5, gate logits frozen, uvc_optimizer returns early
            tr.begin_epoch(1)
            for _ in range(5):
                tr.step(x, y)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for _ in range(20):
                tr.step(x, y)
            torch.cuda.synchronize()
            line["warmup_phase_images_per_sec"] = round(20 * args.batch / (time.perf_counter() - tw), 1)
        if args.stage == 1 and args.precision.startswith("bf16") and args.model_type in GFLOP_PER_IMG:
            del tr, out
            torch.cuda.empty_cache()
            roof, top, total_ms = kernel_table(args)
            line["roofline"] = roof
            line["top_kernels"] = top
            line["kernel_ms_per_step_standalone_sum"] = total_ms
            line["in_step_top_kernel"] = in_step_top(args)
        else:
            line["roofline"] = None
        if args.stage == 1 and args.phase == "train" and world == 1 and args.precision == "bf16" and not (args.no_mode_legs or args.no_cpu_baseline):
            # the same step in the two parity-grade modes, outside the timed region (VERDICT r5 #4a): `fp32` is the mode the reference goldens are held to at
            # 1e-3 (exact-float32 MFMA, float32 activations), `bf16_f32resid` keeps float32 residual-stream rows under bf16 operands
            try:
                del tr, out
            except NameError:
                pass
            torch.cuda.empty_cache()
            for prec, key in (("bf16_f32resid", "images_per_sec_bf16_f32resid"), ("fp32", "images_per_sec_fp32_mode")):
                from uvc_amd.stage1 import Stage1Trainer, default_args
                torch.manual_seed(730)
                a2 = default_args(model_type=args.model_type, precision=prec, train_batch_size=args.batch, local_rank=local, budget=args.budget,
                                  enable_deit=args.enable_deit, enable_patch_gating=args.enable_patch_gating, patch_ratio=args.patch_ratio,
                                  enable_block_gating=args.enable_block_gating)
                t2 = Stage1Trainer(a2, device=f"cuda:{local}", distributed=False)
                pruned_state(t2)
                t2.begin_epoch(a2.warmup_epochs + 1)
                for _ in range(3):
                    t2.step(x, y, **nx)
                torch.cuda.synchronize()
                tw = time.perf_counter()
                n2 = 12 if prec == "fp32" else 30
                for i in range(n2):
                    t2.step(x, y, **(nx if i + 1 < n2 else {}))
                torch.cuda.synchronize()
                line[key] = round(n2 * args.batch / (time.perf_counter() - tw), 1)
                del t2
                torch.cuda.empty_cache()
            line["parity_modes_note"] = ("fp32: tests hold it to the reference goldens at 1e-3; bf16_f32resid / bf16 (the headline): 2e-2 on loss / logits, "
                                         "<= 2.5 % per gradient tensor against float32 autograd; 3 warm-up + 12 / 30 timed steps each, same batch and state")
        line["commit"] = _git_head()
        if world == 1 and not args.no_cpu_baseline and args.stage == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), file=_real_stdout, flush=True)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed():
    """`--gpus N` is the number of ranks that RUN (the reference launches one process per GPU: run_uvc_train.sh:1-3).
    Started under torch.distributed.run (WORLD_SIZE set), WORLD_SIZE must equal --gpus, otherwise the line would carry a wrong
    n_gpus: exit 2.  Started as plain `python bench.py --gpus N` with N > 1, this process becomes the launcher: it re-executes
    itself under `torch.distributed.run --nproc-per-node N` on 127.0.0.1 and returns the children's exit code."""
    args = parse()
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}: refusing to print a line with a wrong n_gpus", file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    share = os.environ.get("UVC_BENCH_SHARE_DEVICE", "0") not in ("", "0")
    if not share and torch.cuda.device_count() < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        sys.exit(2)
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    launch_ranks_if_needed()
    # stdout carries exactly ONE line, the JSON record: the library's progress prints (FLOP size, eps updates, gating banners)
    # are the reference's own messages and go to stderr here
    import contextlib
    _real_stdout = sys.stdout
    with contextlib.redirect_stdout(sys.stderr):
        main()
