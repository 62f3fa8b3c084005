"""Stage-2 masked fine-tune on the MI355X engine (mirror of ``UVC/post_train.py``: ``setup`` :135-186, the
checkpoint hand-over :676-683 and the loop body of ``post_training`` :270-403).

Stage-2 consumes the Stage-1 checkpoint (bare state_dict with ``mask`` buffers and the learned
``block_skip_gating`` logits) and fine-tunes the pruned network: every step starts with
``weight *= mask`` for every module that carries a mask, the forward hard-skips the blocks whose gate
logits say so (model_distilled.py:496-500, in training mode too), the gate logits are frozen, and the
optimiser / schedule come from timm's factories (AdamW with no decay on 1-D tensors, biases and the
tokens; cosine schedule stepped once per epoch on a learning rate scaled by batch * world / 512).

``Stage2Trainer`` is what the CLI (``python -m uvc_amd.post_train``), ``bench.py --stage 2`` and the
parity tests drive.  No CPU fallback: everything numeric is a kernel of libuvc_hip.so.
"""
from __future__ import annotations

import argparse
import json
import os
import time
from argparse import Namespace

import torch

from .ddp import DistributedDataParallel
from .joint_train import count_mask, save_model
from .losses import DistillationLoss, SoftTargetCrossEntropy, unit_gradient
from .model_distilled import DistilledVisionTransformer
from .optim import clip_grad_norm_, create_optimizer
from .scheduler import create_scheduler
from .stage1 import CONFIGS, Stage1Trainer


def default_args(**over) -> Namespace:
    """argparse defaults of post_train.py:412-600 overlaid with run_post_train.sh; keyword arguments override."""
    a = dict(model_type="deit_tiny_patch16_224", img_size=224, num_classes=1000, train_batch_size=128, learning_rate=1e-4,
             weight_decay=0.05, epochs=120, max_grad_norm=1.0, gradient_accumulation_steps=1, seed=42,
             opt="adamw", opt_eps=1e-8, opt_betas=None, momentum=0.9, sched="cosine", lr_noise=None, warmup_lr=1e-6,
             min_lr=1e-5, decay_epochs=30, warmup_epochs=5, cooldown_epochs=10, patience_epochs=10, decay_rate=0.1,
             distillation_type="soft", distillation_alpha=0.1, distillation_tau=1.0, enable_deit=0, local_rank=-1,
             precision="bf16", output_dir="output", name="post_train", steps_per_epoch=5005, compact_mlp=1, compact_multiple=256)
    a.update(over)
    return Namespace(**a)


def register_masks(model):
    """post_train.py:155-157: every module with a ``weight`` gets a ``mask`` buffer of ones."""
    for _, m in model.named_modules():
        if hasattr(m, "weight") and not hasattr(m, "mask"):
            m.register_buffer("mask", torch.ones_like(m.weight))


def setup(args, device="cuda", model_cfg=None):
    """post_train.py:135-186 for the DeiT family: the student as Stage-2 builds it (default gate flags, i.e. hard
    block skip; ``gumbel_hard=True``) with mask buffers registered."""
    if "t2t" in args.model_type:                                    # post_train.py:165-167: t2t_vit_14() with the default flags
        from .stage1 import T2T_CONFIGS
        from .t2t_vit import T2T_ViT
        cfg = dict(T2T_CONFIGS[args.model_type]) if args.model_type in T2T_CONFIGS else dict(model_cfg or args.model_cfg)
        if args.enable_deit:
            raise NotImplementedError("T2T-ViT has no distillation token")
        kw = dict(embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=cfg.get("mlp_ratio", 3.0),
                  img_size=args.img_size, num_classes=args.num_classes, precision=args.precision, device=device)
        model = T2T_ViT(**kw)
        register_masks(model)
        return args, model, kw
    cfg = dict(CONFIGS[args.model_type]) if args.model_type in CONFIGS else dict(model_cfg or args.model_cfg)
    kw = dict(patch_size=cfg["patch_size"], embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"],
              mlp_ratio=cfg.get("mlp_ratio", 4), qkv_bias=True, drop_rate=0, img_size=args.img_size,
              num_classes=args.num_classes, precision=args.precision, device=device)
    model = DistilledVisionTransformer(enable_dist=args.enable_deit, gumbel_hard=True, **kw)
    register_masks(model)
    return args, model, kw


class Stage2Trainer:
    def __init__(self, args: Namespace, device="cuda", checkpoint=None, teacher_state=None, distributed=False, world_size=1):
        self.args = args
        args, model, kw = setup(args, device)
        teacher = None
        if args.distillation_type != "none":                                                        # :636-666
            if "t2t" in args.model_type:                                                            # :659-660
                from .t2t_vit import T2T_ViT
                teacher = T2T_ViT(**kw)
            else:
                teacher = DistilledVisionTransformer(enable_dist=args.enable_deit, **kw)
            if teacher_state is not None:
                teacher.load_state_dict(teacher_state, strict=False)
            teacher.eval()
            teacher.frozen_weights = True
        self.criterion = DistillationLoss(SoftTargetCrossEntropy(), teacher, args.distillation_type,
                                          args.distillation_alpha, args.distillation_tau)          # :668-671
        if checkpoint is not None:                                                                  # :676-683
            model.load_state_dict(checkpoint)       # `hasattr(checkpoint, 'args')` is never true for a dict: bare state_dict
        self.model, self.teacher = model, teacher
        self.total_param = count_mask(model)
        # structured sparsity: MLP hidden units whose fc1 row and fc2 column are masked out are skipped, not multiplied
        self.mlp_widths = model.set_mlp_compaction(multiple=getattr(args, "compact_multiple", 256)) if getattr(args, "compact_mlp", 1) else None
        self.head_keep = model.set_head_skipping() if getattr(args, "compact_mlp", 1) else None     # eval forwards skip pruned heads ...
        # ... and training BACKWARDS skip their dq / dk / dv: step() multiplies the weights by the masks before every forward (:343-346), so
        # dL/d(attention output) of a head whose 64 attn.proj input columns are masked is exactly zero (the forward keeps the head: the
        # reference's clip norm sees dW_proj of the masked columns, which needs its output)
        model.skip_pruned_head_grads = model._head_keep is not None
        # post_training(): DDP, scaled learning rate, timm optimiser + schedule (:289-301)
        self.ddp = DistributedDataParallel(model, message_size=250000000, gradient_predivide_factor=1.0) if distributed else None
        args.train_batch_size = args.train_batch_size // args.gradient_accumulation_steps
        args.lr = args.learning_rate * args.train_batch_size * world_size / 512.0
        self.optimizer = create_optimizer(args, model)
        self.scheduler, self.num_epochs = create_scheduler(args, self.optimizer)
        model.block_skip_gating.requires_grad = False                                               # :313
        model.train()
        self.accum = max(1, int(getattr(args, "gradient_accumulation_steps", 1)))                  # :365-378
        model.grad_accumulate = self.accum > 1
        self._micro = 0
        self.global_step = 0
        self.epoch = 0

    def begin_epoch(self, epoch: int):
        """post_train.py:326-339."""
        self.epoch = epoch
        self._micro = 0          # `(step + 1) % k` on the epoch's loader index (:351,372): an accumulation window never straddles epochs
        self.model.train()
        self.model.block_skip_gating.requires_grad = False
        self.scheduler.step(epoch)

    def step(self, x, y, zero_grad=True, next_x=None):
        """post_train.py:341-377 after the odd-batch trim and mixup: mask, forward (hard block skip), loss, backward,
        clip, AdamW.  ``next_x``: as Stage1Trainer.step."""
        a = self.args
        self.model.apply_masks()                                                                    # :343-346
        overlap = bool(getattr(a, "overlap_teacher", 1))
        if overlap and not self.criterion.has_prefetch(x):
            self.criterion.prefetch(x)
        outputs, _ = self.model(x)                                                                  # :363
        loss = self.criterion(x, outputs, y)
        if self.accum > 1:
            loss = loss / self.accum                                                                # :365-366
        loss.backward(unit_gradient(loss.device))
        if overlap and next_x is not None:
            self.criterion.prefetch(next_x)
        self._micro += 1
        if self._micro % self.accum != 0:                                                           # :372: the backward added into .grad
            return dict(loss=loss.detach() * self.accum, outputs=outputs, stepped=False)
        gnorm = clip_grad_norm_(self.model, a.max_grad_norm)                                        # :377
        self.optimizer.step()
        self.global_step += 1
        if zero_grad:
            self.optimizer.zero_grad()
        return dict(loss=loss.detach() * self.accum if self.accum > 1 else loss.detach(), outputs=outputs, gnorm=gnorm, stepped=True)


    # resumable state (the reference saves only the best model's bare state_dict, post_train.py:395-397)
    def state_dict(self):
        o = self.optimizer
        return dict(format="uvc_amd.stage2.v1", model=self.model.state_dict(),
                    adamw=dict(exp_avg=o.exp_avg.clone(), exp_avg_sq=o.exp_avg_sq.clone(), steps=dict(o.steps), lr=o.param_groups[0]["lr"]),
                    progress=dict(global_step=self.global_step, epoch=self.epoch))

    def load_state_dict(self, sd):
        if sd.get("format") != "uvc_amd.stage2.v1":
            raise ValueError("not a uvc_amd Stage-2 training state")
        self.model.load_state_dict(sd["model"])      # (re-derives the pruned-head table and the MLP compaction from the loaded masks)
        if getattr(self.args, "compact_mlp", 1):
            self.head_keep = self.model.set_head_skipping()
            self.mlp_widths = self.model.set_mlp_compaction(multiple=getattr(self.args, "compact_multiple", 256))
            self.model.skip_pruned_head_grads = self.model._head_keep is not None
        a, o = sd["adamw"], self.optimizer
        o.exp_avg.copy_(a["exp_avg"]); o.exp_avg_sq.copy_(a["exp_avg_sq"]); o.steps = dict(a["steps"]); o.param_groups[0]["lr"] = a["lr"]
        self.global_step, self.epoch = int(sd["progress"]["global_step"]), int(sd["progress"]["epoch"])


def post_training(trainer: Stage2Trainer, batches, epochs=None, valid_fn=None, log=print):
    """The epoch loop of post_train.py:326-403 over an iterable factory ``batches(epoch)`` of device (x, y) pairs
    (mixup already applied); ``valid_fn(model) -> accuracy`` drives the save-best policy (:393-399)."""
    a = trainer.args
    best_acc = 0.0
    for epoch in range(epochs if epochs is not None else a.epochs):
        trainer.begin_epoch(epoch)
        t0 = time.time()
        last = None
        def trimmed():
            for x, y in batches(epoch):
                if len(x) % 2 != 0:                                                                 # :348-350
                    x, y = x[:-1], y[:-1]
                yield x, y
        # one batch read ahead (as the reference's prefetching loader holds it): the frozen teacher's forward for the NEXT batch starts behind this
        # step's backward (Stage2Trainer.step's next_x; same results, tests/test_stage2_gpu.py)
        for (x, y), next_x in Stage1Trainer.lookahead(trimmed()):
            last = trainer.step(x, y, next_x=next_x)
        lr = trainer.scheduler.get_epoch_values(epoch)[0]
        if last is not None:
            log(f"[Stage 2] epoch {epoch} steps {trainer.global_step} lr {lr:.6g} loss {float(last['loss']):.4f} "
                f"({time.time() - t0:.1f}s)")
        if valid_fn is not None and a.local_rank in (-1, 0):
            acc = valid_fn(trainer.model)
            if best_acc < acc:
                save_model(a, trainer.model, None, trainer.global_step, barrier=False)
                best_acc = acc
            trainer.model.train()
    return best_acc


def main(argv=None):
    """Synthetic-data Stage-2 run (the image has no dataset): loads a Stage-1 checkpoint and fine-tunes it."""
    p = argparse.ArgumentParser(description="UVC Stage-2 masked fine-tune on MI355X (synthetic data)")
    d = default_args()
    for k, v in vars(d).items():
        if v is None or isinstance(v, (list, tuple)):
            p.add_argument("--" + k, default=v)
        else:
            p.add_argument("--" + k, type=type(v), default=v)
    p.add_argument("--checkpoint_dir", type=str, default=None, help="Stage-1 checkpoint (bare state_dict)")
    p.add_argument("--steps", type=int, default=20, help="steps per synthetic epoch")
    p.add_argument("--model_cfg", type=str, default=None, help="JSON dims for a --model_type outside models/configs.py (tests)")
    p.add_argument("--eval_steps", type=int, default=2, help="synthetic validation batches per epoch (valid(), post_train.py:188-234)")
    p.add_argument("--eval_batch_size", type=int, default=64)
    args = p.parse_args(argv)
    if args.model_cfg:
        args.model_cfg = json.loads(args.model_cfg)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl")
        args.local_rank = local
    ck = torch.load(args.checkpoint_dir, map_location="cpu") if args.checkpoint_dir else None
    tr = Stage2Trainer(args, device=f"cuda:{local}", checkpoint=ck, distributed=world > 1, world_size=world)
    dev = torch.device("cuda", local)
    g = torch.Generator(device=dev).manual_seed(args.seed + rank)

    def batches(epoch):
        for _ in range(args.steps):
            x = torch.randn(args.train_batch_size, 3, args.img_size, args.img_size, device=dev, generator=g)
            y = torch.softmax(torch.randn(args.train_batch_size, args.num_classes, device=dev, generator=g), -1)
            yield x, y

    @torch.no_grad()
    def valid_fn(model):
        """valid() of post_train.py:188-234 on synthetic (x, hard label) batches: eval-mode logits, top-1 in percent."""
        model.eval()
        ge = torch.Generator(device=dev).manual_seed(args.seed + 77)
        hit = n = 0
        for _ in range(args.eval_steps):
            x = torch.randn(args.eval_batch_size, 3, args.img_size, args.img_size, device=dev, generator=ge)
            t = torch.randint(0, args.num_classes, (args.eval_batch_size,), device=dev, generator=ge)
            logits, _ = model(x)
            hit += int((logits.argmax(dim=1) == t).sum())
            n += len(t)
        from .model_distilled import drop_shared_patches
        drop_shared_patches()
        return 100.0 * (hit + 1e-3) / max(n, 1)          # + epsilon: the first epoch always beats best_acc = 0 and saves (:393-397)

    best = post_training(tr, batches, epochs=args.epochs, valid_fn=valid_fn if args.eval_steps > 0 else None,
                         log=print if rank == 0 else (lambda *_: None))
    if rank == 0:
        print(json.dumps(dict(steps=tr.global_step, masked_params_M=float(tr.total_param), best_acc=best)))
    if world > 1:
        torch.distributed.destroy_process_group()
    return tr


if __name__ == "__main__":
    main()
