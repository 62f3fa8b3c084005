"""Stage-1 command line (mirror of the argparse surface of ``UVC/joint_train.py:684-879`` and of the
epoch loop :330-514) on the MI355X engine.

    python -m torch.distributed.run --nproc-per-node N -m uvc_amd.cli --model_type deit_tiny_patch16_224 \\
        --budget 0.5 --enable_pruning 1 --enable_block_gating 1 --enable_patch_gating 0 ...

Flag names and defaults are the reference's.  Data: the image has no torchvision/timm, so the loop
runs on synthetic ImageNet-shaped batches (``--synthetic 1``, the default); the (x, y_soft) contract
after mixup is what the step consumes, so a real loader + Mixup plugs in at
``iterate_batches``.  Checkpoints ({output_dir}/{name}/{model_type}_{epoch}.pth.tar = bare state_dict
incl. masks) and the s_/r_/gating_ JSON side logs keep the reference layout.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .joint_train import count_mask, save_model
from .stage1 import CONFIGS, Stage1Trainer
from .uvc_utils import prune_w_mask


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    a("--name", default="debug"); a("--dataset", choices=["cifar10", "cifar100", "imagenet"], default="imagenet")
    a("--data_dir", default="/ssd1/shixing/imagenet2012"); a("--num_workers", default=4, type=int)
    a("--model_type", choices=list(CONFIGS) + ["t2t_vit_14", "custom", "custom_t2t"], default="deit_tiny_patch16_224")
    a("--model_path", default=None); a("--pretrained_dir", type=str, default="../ViT-pytorch/pretrain/ViT-B_16.npz"); a("--pretrained", type=int, default=1)
    a("--output_dir", default="../result/output/uvc_train", type=str); a("--img_size", default=224, type=int)
    a("--train_batch_size", default=1024, type=int); a("--eval_batch_size", default=64, type=int)
    a("--eval_every", default=1000, type=int); a("--learning_rate", default=1e-4, type=float)
    a("--weight_decay", default=0.05, type=float); a("--num_steps", default=10000, type=int)
    a("--num_epochs", default=20, type=int); a("--decay_type", choices=["cosine", "linear"], default="cosine")
    a("--warmup_steps", default=500, type=int); a("--max_grad_norm", default=1.0, type=float)
    a("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0))); a("--seed", type=int, default=42)
    a("--gradient_accumulation_steps", type=int, default=1)
    a("--fp16", action="store_true"); a("--fp16_opt_level", type=str, default="O2"); a("--loss_scale", type=float, default=0)
    a("--uvc_train", action="store_true", default=True); a("--soptim", default="sgd"); a("--roptim", default="sgd")
    a("--zlr_schedule_list", default="10,20,30,40,50", type=str)
    a("--ylr", default=1e-4, type=float); a("--plr", default=1e-4, type=float); a("--slr", default=0.02, type=float)
    a("--rlr", default=0.02, type=float); a("--glr", default=1e-3, type=float); a("--log_interval", default=2000, type=int)
    a("--save_budgets", default="0.6, 0.5, 0.4"); a("--budget", default=0.5); a("--sl2wd", default=0.0, type=float)
    a("--verbose", default=True, action="store_true")
    a("--mixup", type=float, default=0.8); a("--cutmix", type=float, default=1.0)
    a("--cutmix-minmax", type=float, nargs="+", default=None); a("--mixup-prob", type=float, default=0.8)
    a("--mixup-switch-prob", type=float, default=0.5); a("--mixup-mode", type=str, default="batch")
    a("--teacher-model", default=None, type=str); a("--teacher-path", type=str, default=None)
    a("--distillation-type", default="hard", choices=["none", "soft", "hard"], type=str)
    a("--distillation-alpha", default=0.5, type=float); a("--distillation-tau", default=1.0, type=float)
    a("--smoothing", type=float, default=0.1)
    a("--post_learning_rate", default=1e-3, type=float); a("--post_weight_decay", default=0.05, type=float)
    a("--post_num_epochs", default=100, type=int)
    a("--use_distribute", default=1, type=int); a("--enable_writer", default=0, type=int)
    a("--flops_with_mhsa", type=int, default=1); a("--enable_block_gating", type=int, default=1)
    a("--enable_part_gating", type=int, default=0); a("--enable_jumping", type=int, default=0)
    a("--enable_deit", type=int, default=0); a("--enable_pruning", type=int, default=1)
    a("--enable_patch_gating", type=int, default=2); a("--patch_ratio", type=float, default=0.9)
    a("--z_grad_clip", default=0.5, type=float); a("--gating_interval", default=100, type=int)
    a("--gating_weight", default=5, type=float); a("--patch_weight", default=5, type=float)
    a("--patch_l1_weight", default=0.01, type=float); a("--patchlr", default=0.01, type=float)
    a("--patchloss", default="l1", type=str); a("--use_gumbel", default=1, type=int)
    a("--eps", default=0.1, type=float); a("--eps_decay", default=0.92, type=float)
    a("--enable_warmup", default=1, type=int); a("--warmup_epochs", default=5, type=int)
    a("--warmup_lr", default=1e-4, type=float); a("--warmup_reset", default=0, type=int)
    a("--gpu_num", type=str, default="0, 1")
    # engine-specific (not in the reference)
    a("--precision", default="bf16", choices=["bf16", "bf16_f32resid", "fp32"],
      help="bf16 MFMA (throughput; bf16_f32resid keeps float32 residual-stream rows) or exact float32 MFMA (parity)")
    a("--synthetic", type=int, default=1, help="synthetic ImageNet-shaped batches (no torchvision in this image)")
    a("--steps_per_epoch", type=int, default=5005, help="len(train_loader) for synthetic data (ImageNet @256 = 5005)")
    a("--num_classes", type=int, default=1000)
    a("--resume", type=str, default=None, help="engine training state written by --save_state (complete: s r y p z, AdamW, schedule)")
    a("--save_state", type=int, default=1, help="also write <name>/<model>_state_<epoch>.pth.tar (resumable) next to the reference-format checkpoint")
    a("--model_cfg", type=str, default=None, help='with --model_type custom / custom_t2t: JSON dims, e.g. {"patch_size":16,"embed_dim":128,"depth":2,"num_heads":2}')
    a("--eval_steps", type=int, default=2, help="synthetic validation batches per epoch (valid(), joint_train.py:199-246)")
    return p


def build_mixup(args):
    """joint_train.py:922-933."""
    from .mixup import Mixup
    mixup_active = args.mixup > 0 or args.cutmix > 0. or args.cutmix_minmax is not None
    if not mixup_active:
        return None
    return Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, cutmix_minmax=args.cutmix_minmax, prob=args.mixup_prob,
                 switch_prob=args.mixup_switch_prob, mode=args.mixup_mode, label_smoothing=args.smoothing, num_classes=args.num_classes)


def iterate_batches(args, device, rank, mixup_fn=None, epoch=0):
    """Synthetic ImageNet-shaped batches with hard labels, passed through the on-device Mixup / CutMix exactly where the
    reference calls mixup_fn (joint_train.py:399-409: odd batches lose their last sample first); without mixup the
    labels become smoothed one-hot soft targets.  Seeded per (rank, epoch): every epoch sees fresh batches, like a
    DistributedSampler after set_epoch."""
    g = torch.Generator(device=device).manual_seed(args.seed + 1000 * rank + 1000003 * epoch)
    for _ in range(args.steps_per_epoch):
        x = torch.randn(args.train_batch_size, 3, args.img_size, args.img_size, device=device, generator=g)
        t = torch.randint(0, args.num_classes, (args.train_batch_size,), device=device, generator=g)
        if len(x) % 2 != 0:
            x, t = x[:-1].contiguous(), t[:-1]
        if mixup_fn is not None:
            x, y = mixup_fn(x, t)
        else:
            off = args.smoothing / args.num_classes
            y = torch.full((len(x), args.num_classes), off, device=device).scatter_(1, t.view(-1, 1), 1.0 - args.smoothing + off)
        yield x, y


def iterate_eval_batches(args, device, rank):
    """Synthetic stand-in for the test loader of valid() (joint_train.py:199-246): (x, hard labels), eval_batch_size each."""
    g = torch.Generator(device=device).manual_seed(args.seed + 77 + 1000 * rank)
    for _ in range(args.eval_steps):
        x = torch.randn(args.eval_batch_size, 3, args.img_size, args.img_size, device=device, generator=g)
        t = torch.randint(0, args.num_classes, (args.eval_batch_size,), device=device, generator=g)
        yield x, t


def append_json(path, step, value):
    """joint_train.py:466-486: `data.update({global_step: value}); json.dump` -- int keys become strings on disk."""
    with open(path, "r+") as f:
        data = json.load(f)
        data.update({str(step): value})
        f.seek(0)
        json.dump(data, f)
        f.truncate()


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.fp16:
        raise NotImplementedError("--fp16 (apex amp) is not reproduced: bf16 MFMA with float32 master weights needs no loss scaling")
    if not args.synthetic:
        raise NotImplementedError("real-data loading needs torchvision/timm (absent here); plug a loader into iterate_batches")
    if args.model_type.startswith("custom"):
        if not args.model_cfg:
            raise SystemExit("--model_type custom needs --model_cfg '<json>'")
        args.model_cfg = json.loads(args.model_cfg)
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    torch.cuda.set_device(args.local_rank)
    device = torch.device("cuda", args.local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
    torch.manual_seed(args.seed)                      # same seed on every rank (joint_train.py:191-196,914)
    args.train_batch_size = args.train_batch_size // args.gradient_accumulation_steps
    tr = Stage1Trainer(args, device=device, distributed=world > 1)
    np.random.seed(args.seed)                         # Mixup draws from numpy's global RNG (set_seed, joint_train.py:191-196)
    mixup_fn = build_mixup(args)
    if rank == 0:
        print(f"mixup active: {mixup_fn is not None}")
    out_dir = os.path.join(args.output_dir, args.name)
    stamp = time.strftime("%Y-%m-%d-%H:%M:%S", time.localtime())
    logs = {k: os.path.join(out_dir, f"{k}_{stamp}.json") for k in ("s", "r", "gating")}
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        for f in logs.values():
            json.dump({}, open(f, "w"))
        print("***** [Stage 1] Training with ADMM *****")
        print(f"  Instantaneous batch size per GPU = {args.train_batch_size}")
        print(f"  Total train batch size (w. parallel, distributed & accumulation) = {args.train_batch_size * args.gradient_accumulation_steps * world}")
    first_epoch = 1
    if args.resume:
        tr.load_state_dict(torch.load(args.resume, map_location=device))
        first_epoch = tr.epoch + 1
        if rank == 0:
            print(f"resumed from {args.resume}: epoch {tr.epoch} done, global step {tr.global_step}")
    for epoch in range(first_epoch, args.num_epochs + 2):       # `while epoch <= num_epochs: epoch += 1` (:331-335)
        tr.begin_epoch(epoch)
        stage = "Warm Up" if tr.model.enable_warmup else "UVC Train"
        remained = float(count_mask(tr.model))
        if rank == 0:
            print("=" * 60)
            print(f"Start [Epoch {epoch}] at Stage {stage}")
            print(f"[Initial Sparsity|Epoch {epoch}] Parameter size: {remained:.2f}M / {float(args.total_param):.2f}M = {remained / float(args.total_param) * 100:.2f}%")
        t0 = time.time()
        for step, ((x, y), next_x) in enumerate(tr.lookahead(iterate_batches(args, device, rank, mixup_fn, epoch))):
            out = tr.step(x, y, next_x=next_x)
            if not out["stepped"]:                      # gradient accumulation: not an optimiser step yet (:417)
                continue
            gs = tr.global_step
            if rank == 0 and gs % args.log_interval == 0:
                print(f"{stage} [{epoch} Epochs] [{gs} / {tr.t_total} Steps] [LR: {tr.scheduler.get_last_lr()[0]:.6f} | "
                      f"Loss: {float(out['loss']):.3f}] resource {float(out['cur']):.4f}  {(step + 1) * args.train_batch_size * world / (time.time() - t0):.0f} img/s")
                if epoch > args.warmup_epochs:                                                   # :464-486
                    append_json(logs["s"], gs, out["s"].tolist())
                    append_json(logs["r"], gs, out["r"].tolist())
                    if args.enable_block_gating and out["g"] is not None:
                        append_json(logs["gating"], gs, out["g"].tolist())
        if rank == 0:
            print("*" * 60)
            print("Epoch finished, begin validating ...")
        val = tr.validate(iterate_eval_batches(args, device, rank))                              # :498
        if rank == 0:
            print(f"Validation Results\nGlobal Steps: {tr.global_step}\nValid Loss: {val['loss']:2.5f}\nValid Accuracy: {val['top1']:2.5f}")
        tr.check_replicas()                                                                      # ranks still bit-identical (raises otherwise)
        prune_w_mask(tr.minimax, tr.optimizer)                                                   # :500
        remained = float(count_mask(tr.model))
        save_model(args, tr.model, tr.minimax, epoch)                                            # :502
        # :509.  The reference evaluates the two resource samples on rank 0 only, which advances rank 0's RNG by two Gumbel
        # draws and silently desynchronises the replicas' gate noise; here every rank draws, rank 0 prints.
        hard = bool(tr.model.enable_warmup)
        expect_f, real_f = float(tr.minimax.run_resource_fn(hard)), float(tr.minimax.run_resource_fn(gumbel_hard=True))
        if rank == 0:
            print(f"[Validation Sparsity|Step {tr.global_step}|Epoch {epoch}]")
            print(f"Parameter size: {remained:.2f}M / {float(args.total_param):.2f}M = {remained / float(args.total_param) * 100:.2f}%")
            print(f"Expectation FLOPs: {expect_f * 100}%", f"Real FLOPs: {real_f * 100}%")
        if args.save_state and rank == 0:               # last: the saved RNG state is the one the next epoch starts from
            tmp = os.path.join(out_dir, f"{args.model_type}_state_{epoch}.pth.tar.tmp")
            torch.save(tr.state_dict(), tmp)
            os.replace(tmp, tmp[:-4])
    if world > 1:
        dist.destroy_process_group()
    return tr


if __name__ == "__main__":
    main()
