"""Stage-1 set-up and step body (mirror of ``UVC/joint_train.py`` ``main()`` :881-1026 and the loop body
:395-450) on the MI355X engine.  ``Stage1Trainer`` is what the CLI, ``bench.py`` and the parity tests
drive; it composes the drop-in pieces (model, DistillationLoss, FusedAdamW, schedulers,
build_minimax_model / uvc_optimizer, DDP) exactly in the reference's order.
"""
from __future__ import annotations

from argparse import Namespace
from typing import Optional

import torch

from . import _lib as L
from .ddp import DistributedDataParallel
from .joint_train import count_mask, get_uvc_layers, register_masks
from .losses import DistillationLoss, SoftTargetCrossEntropy, unit_gradient
from .model_distilled import DistilledVisionTransformer
from .optim import FusedAdamW, clip_grad_norm_
from .scheduler import PresetLRScheduler, WarmupCosineSchedule, WarmupLinearSchedule
from .uvc_optimizer import build_minimax_model, uvc_optimizer
from .uvc_utils import prune_w_mask

# models/configs.py:112-165 -- dims of the DeiT family the reference instantiates
CONFIGS = {
    "deit_tiny_patch16_224": dict(patch_size=16, embed_dim=192, depth=12, num_heads=3),
    "deit_small_patch16_224": dict(patch_size=16, embed_dim=384, depth=12, num_heads=6),
    "deit_base_patch16_224": dict(patch_size=16, embed_dim=768, depth=12, num_heads=12),
}


# T2TViT/models/t2t_vit.py:244-249 (models/configs.py:159-165); BASELINE config 5
T2T_CONFIGS = {"t2t_vit_14": dict(embed_dim=384, depth=14, num_heads=6, mlp_ratio=3.0)}


def default_args(**over) -> Namespace:
    """The argparse defaults of joint_train.py:684-879 overlaid with the README command
    (run_uvc_train.sh:4-38); keyword arguments override."""
    a = dict(model_type="deit_tiny_patch16_224", img_size=224, num_classes=1000, train_batch_size=512, learning_rate=1e-4,
             weight_decay=0.05, num_epochs=30, decay_type="cosine", warmup_steps=500, max_grad_norm=1.0,
             gradient_accumulation_steps=1, seed=730, uvc_train=True, soptim="sgd", roptim="sgd",
             zlr_schedule_list="1,5,9,13,17", ylr=1e-4, plr=1e-4, slr=0.02, rlr=0.02, glr=0.1, log_interval=1000,
             budget=0.5, sl2wd=0.0, distillation_type="soft", distillation_alpha=0.1, distillation_tau=1.0,
             flops_with_mhsa=1, enable_block_gating=1, enable_part_gating=0, enable_jumping=0, enable_deit=0,
             enable_pruning=1, enable_patch_gating=0, patch_ratio=0.9, z_grad_clip=0.5, gating_interval=50,
             gating_weight=5e-4, use_gumbel=1, eps=0.1, eps_decay=0.92, enable_warmup=1, warmup_epochs=5, warmup_lr=1e-4,
             warmup_reset=0, local_rank=-1, steps_per_epoch=5005, precision="bf16", output_dir="output", name="debug")
    a.update(over)
    return Namespace(**a)


def _bits_fingerprint(t, chunk=1 << 22):
    """64-bit hash of a float32 tensor's bit pattern: sum_i (i mod 65521 + 1) * bits_i in wrapping int64 arithmetic, chunked so that
    the temporaries stay at 2 x chunk x 8 bytes whatever the model size."""
    v = t.detach().reshape(-1).view(torch.int32)
    acc = torch.zeros((), dtype=torch.int64, device=v.device)
    for o in range(0, v.numel(), chunk):
        c = v[o:o + chunk].to(torch.int64)
        w = (torch.arange(o, o + c.numel(), device=v.device, dtype=torch.int64) % 65521) + 1
        acc += (c * w).sum()
    return acc


class Stage1Trainer:
    def __init__(self, args: Namespace, device="cuda", student_state=None, teacher_state=None, distributed=False):
        if not args.enable_pruning:
            raise TypeError("enable_pruning=0 raises in the reference (uvc_optimizer_gating signature, SURVEY.md Q7)")
        self.args = args
        t2t = "t2t" in args.model_type                                      # joint_train.py:143-145
        if t2t:
            cfg = dict(T2T_CONFIGS[args.model_type]) if args.model_type in T2T_CONFIGS else dict(args.model_cfg)
        else:
            cfg = dict(CONFIGS[args.model_type]) if args.model_type in CONFIGS else dict(args.model_cfg)
        args.head_size = cfg["embed_dim"] // cfg["num_heads"]              # joint_train.py:883-885
        args.num_heads = cfg["num_heads"]
        args.budget = float(args.budget)
        if t2t:
            # The reference builds `t2t_vit_14()` with default flags and then calls it as model(x, tau, ratio), which raises
            # (SURVEY Q8).  Defined here: the same model with DeiT's gate flags (block gating as written at t2t_vit.py:181-189).
            from .t2t_vit import T2T_ViT
            if args.enable_deit or args.enable_patch_gating == 1:
                raise NotImplementedError("T2T-ViT has no distillation token and no patch-gating mode 1 (t2t_vit.py:168-200); "
                                          "enable_patch_gating=2 is defined in uvc_amd/t2t_vit.py")
            kw = dict(embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=cfg.get("mlp_ratio", 3.0),
                      img_size=args.img_size, num_classes=args.num_classes, precision=args.precision, device=device)
            model = T2T_ViT(gumbel_hard=False, enable_patch_gating=args.enable_patch_gating, **kw)
        else:
            kw = dict(patch_size=cfg["patch_size"], embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"],
                      mlp_ratio=cfg.get("mlp_ratio", 4), qkv_bias=True, drop_rate=0, img_size=args.img_size,
                      num_classes=args.num_classes, precision=args.precision, device=device)
            model = DistilledVisionTransformer(enable_dist=args.enable_deit, gumbel_hard=False,
                                               enable_patch_gating=args.enable_patch_gating, **kw)      # :135-140
        if student_state is not None:
            model.load_state_dict(student_state, strict=False)
        register_masks(model)                                                                       # :169-171
        args.total_param = count_mask(model)
        teacher = None
        if args.distillation_type != "none":                                                        # :948-981
            if t2t:
                from .t2t_vit import T2T_ViT
                teacher = T2T_ViT(**kw)                                                             # :963-964
            else:
                teacher = DistilledVisionTransformer(enable_dist=args.enable_deit, **kw)
            src = teacher_state if teacher_state is not None else {k: v for k, v in model.state_dict().items()
                                                                   if not k.endswith(".mask") and k != "patch_gating"}
            teacher.load_state_dict(src, strict=False)
            teacher.eval()
            teacher.frozen_weights = True
        self.model, self.teacher = model, teacher
        self.criterion = DistillationLoss(SoftTargetCrossEntropy(), teacher, args.distillation_type,
                                          args.distillation_alpha, args.distillation_tau)          # :986-988
        if isinstance(args.zlr_schedule_list, str):                                                 # :999-1005
            lst = [int(v) for v in args.zlr_schedule_list.split(",")]
            gap = args.num_epochs // len(lst)
            args.zlr_schedule = {i * gap: v for i, v in enumerate(lst)}
            args.zlr_schedule_list = lst
        names, self.uvc_layers, ldict = get_uvc_layers(model, args)                                 # :1007
        with torch.no_grad():                                                                       # :1010-1012
            model.eval()
            _, flops_list = model(torch.ones(1, 3, args.img_size, args.img_size, device=device), number=args.patch_ratio)
        self.flops_list = flops_list
        (self.minimax, self.dual_opt, self.s_opt, self.r_opt, self.g_opt) = build_minimax_model(
            model, names, self.uvc_layers, ldict, args, flops_list)                                 # :1014
        prune_w_mask(self.minimax)                                                                  # :1026
        # train(): optimiser, schedule, DDP (:271-295)
        self.accum = max(1, int(getattr(args, "gradient_accumulation_steps", 1)))       # :403-426
        model.grad_accumulate = self.accum > 1
        self._micro = 0
        self._build_optimizer()
        self.zlr_scheduler = PresetLRScheduler(getattr(args, "zlr_schedule", {}))
        self.ddp = DistributedDataParallel(model, message_size=250000000, gradient_predivide_factor=1.0,
                                           dual_scalar=self.minimax.z) if distributed else None
        model.train()
        self.global_step = 0
        self.epoch = 0
        self.gating_grad_list = []
        # Gumbel noise of the gates / resource samples / patch top-k: keyed by (seed, optimiser step, call number), see ops.KeyedExpSource
        from .ops import KeyedExpSource
        self.noise = KeyedExpSource(getattr(args, "seed", 0), model._flat.device)
        model.exp_source = self.noise
        self.minimax.exp_source = self.noise

    def _build_optimizer(self):
        """AdamW over every parameter + the warm-up-cosine / linear schedule over len(loader) * num_epochs optimiser steps
        (joint_train.py:271-276; rebuilt by --warmup_reset at the first UVC-train epoch, :357-365)."""
        a = self.args
        self.optimizer = FusedAdamW(self.model, lr=a.learning_rate, weight_decay=a.weight_decay)
        self.t_total = a.steps_per_epoch * a.num_epochs
        sched = WarmupCosineSchedule if a.decay_type == "cosine" else WarmupLinearSchedule
        self.scheduler = sched(self.optimizer, warmup_steps=a.warmup_steps, t_total=self.t_total)

    # -- epoch header (joint_train.py:335-386)
    def begin_epoch(self, epoch: int):
        a, mm = self.args, self.minimax
        self.epoch = epoch
        self.gating_grad_list = []
        # the reference counts accumulation windows with the loader index of the epoch, `(step + 1) % k` (:423): a window never straddles
        # an epoch boundary; gradients of an unfinished window stay in .grad (zero_grad sits inside the `if`) and join the next one
        # (its noise draws continue under the same step number, KeyedExpSource.begin_step; `or`: a flag restored by load_state_dict from a checkpoint taken
        #  AFTER a begin_epoch -- micro already 0 -- survives the begin_epoch of the resumed run, so the site numbering continues as in the uninterrupted run)
        self._window_carried = bool(getattr(self, "_window_carried", False)) or (self._micro % self.accum != 0)
        self._micro = 0
        # The reference keys the warm-up phase on the epoch alone (:343): --enable_warmup never reaches the model
        # (`enable_warmpup` typo in build_minimax_model's kwargs), so `--enable_warmup 0 --warmup_epochs 5` still warms up.
        if epoch <= a.warmup_epochs:
            mm.model.enable_warmup = 1
            mm.model.block_skip_gating.requires_grad = False
            for g in self.optimizer.param_groups:
                g["lr"] = a.warmup_lr
        else:
            mm.model.enable_warmup = 0
            a.enable_warmup = 0
            mm.model.block_skip_gating.requires_grad = True
            if epoch == a.warmup_epochs + 1 and getattr(a, "warmup_reset", 0):              # :357-365
                if getattr(a, "local_rank", -1) in (-1, 0):
                    print(" Reset the Optimizer and Learning rate scheduler")
                self._build_optimizer()
        prune_w_mask(mm, self.optimizer)
        if not mm.model.enable_warmup:
            mm.update_eps()

    def get_tau(self):
        if self.args.enable_patch_gating == 2:                                                      # :83-85,404-407
            return 0.1 + (10 - 0.1) * self.global_step / self.t_total
        return -1

    # -- one loader iteration (joint_train.py:395-450), x / y already mixed.  With --gradient_accumulation_steps k only every
    # k-th call is an optimiser step (:417-426): the others return after the backward, which ADDS into the flat gradient
    # buffer (model.grad_accumulate), and their result carries `stepped=False`.
    def step(self, x, y, tau=None, zero_grad=True, next_x=None):
        """One training step on the batch (x, y).  ``next_x`` (optional): the NEXT step's input batch, if the caller already holds it (a prefetching loader does;
        ``lookahead`` below wraps one): the frozen teacher's forward for it is started on the side stream as soon as this step's backward is enqueued, so it runs
        under the optimizer / UVC tail of small launches, where the chip is otherwise nearly idle, instead of in front of the next step's student forward (the two
        forwards are whole-chip kernels that alternate).  Results do not depend on it: the next step picks the forward up only for that very tensor."""
        a = self.args
        if self._micro % self.accum == 0:
            self.noise.begin_step(self.global_step, resume_window=getattr(self, "_window_carried", False))
            self._window_carried = False
        overlap = bool(getattr(a, "overlap_teacher", 1))
        if overlap and not self.criterion.has_prefetch(x):
            self.criterion.prefetch(x)              # teacher forward on a side stream, under the student forward
        outputs, _ = self.model(x, self.get_tau() if tau is None else tau, a.patch_ratio)
        loss = self.criterion(x, outputs, y)
        if self.accum > 1:
            loss = loss / self.accum                                                                # :413-414
        loss.backward(unit_gradient(loss.device))       # d(loss) = 1 without a ones_like fill or a multiply by it (losses.unit_gradient)
        if overlap and next_x is not None:
            # (enqueued behind the LOSS instead, the teacher's whole-chip kernels alternate with the backward's: 11.28 against 11.14 ms, profiles/r5zz_ab_next_teacher_at.txt)
            self.criterion.prefetch(next_x)
        self._micro += 1
        if self._micro % self.accum != 0:                                                           # :417
            return dict(loss=loss.detach() * self.accum, outputs=outputs, stepped=False)
        gnorm = clip_grad_norm_(self.model, a.max_grad_norm)
        self.optimizer.step()
        self.scheduler.step()
        self.global_step += 1
        if not self.minimax.model.enable_warmup:
            self.zlr_scheduler(self.dual_opt, self.epoch, "zlr")
        self.minimax.update_gating()
        cur, s, r, g, self.gating_grad_list = uvc_optimizer(
            self.optimizer, self.minimax, self.s_opt, self.r_opt, self.g_opt, self.dual_opt, a, {"global_step": self.global_step},
            [], self.flops_list, a.z_grad_clip, self.global_step, a.gating_interval, self.gating_grad_list)
        if zero_grad:
            self.optimizer.zero_grad()
        return dict(loss=loss.detach() * self.accum if self.accum > 1 else loss.detach(), outputs=outputs, gnorm=gnorm, cur=cur, s=s, r=r, g=g,
                    stepped=True)

    @staticmethod
    def lookahead(batches):
        """(x, y) batches -> ((x, y), next_x or None): what ``step(..., next_x=)`` wants, one batch read ahead."""
        it = iter(batches)
        try:
            cur = next(it)
        except StopIteration:
            return
        for nxt in it:
            yield cur, nxt[0]
            cur = nxt
        yield cur, None

    def check_replicas(self):
        """Data-parallel invariant: every rank holds bit-identical parameters and primal / dual state (the reference assumes it and
        cannot tell when it breaks).  One all-gather of two 64-bit hashes of the raw float32 BIT PATTERNS (position-weighted integer sums in
        chunks: no float64 copies of the parameters, and a test of bit identity rather than of sums); raises on the first divergence.
        No-op on one rank."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return True
        mm = self.minimax
        state = torch.cat([mm.s.data.flatten(), mm.r.data.flatten(), mm.y.data.flatten(), mm.p.data.flatten(), mm.z.data.flatten()]).contiguous()
        fp = torch.stack([_bits_fingerprint(self.model._flat), _bits_fingerprint(state)])
        allfp = [torch.empty_like(fp) for _ in range(dist.get_world_size())]
        dist.all_gather(allfp, fp)
        if not all(torch.equal(allfp[0], t) for t in allfp):
            raise RuntimeError(f"data-parallel replicas diverged at step {self.global_step}: fingerprints {[t.tolist() for t in allfp]}")
        return True

    # -- valid() (joint_train.py:199-246): eval-mode forward, CrossEntropy against hard labels, top-1
    @torch.no_grad()
    def validate(self, batches):
        """``batches`` yields (x, hard labels int64).  Mirrors valid(): model.eval(), tau = 1 with patch-gating mode 2 else -1
        (:216-219), logits = model(x, tau, patch_ratio)[0] (block gating still draws its Gumbel noise, as in the reference),
        mean of the per-batch CrossEntropy and top-1 accuracy in percent; the model goes back to train() afterwards (:513)."""
        a, model = self.args, self.model
        model.eval()
        tau = 1 if a.enable_patch_gating == 2 else -1
        loss_sum = torch.zeros((), device=self.model._flat.device, dtype=torch.float64)
        hit = torch.zeros((), device=self.model._flat.device, dtype=torch.float64)
        nb, n = 0, 0
        for x, t in batches:
            logits, _ = model(x, tau, a.patch_ratio)
            lse = torch.logsumexp(logits.double(), dim=1)
            loss_sum += (lse - logits.double().gather(1, t.view(-1, 1)).squeeze(1)).mean()
            hit += (logits.argmax(dim=1) == t).sum()
            nb += 1
            n += x.shape[0]
        model.train()
        from .model_distilled import drop_shared_patches
        drop_shared_patches()                 # a pass by one model alone leaves a pending patch-row entry behind
        return dict(loss=float(loss_sum) / max(nb, 1), top1=100.0 * float(hit) / max(n, 1), images=n)

    # -- resumable training state.  The reference's checkpoint is the bare model state_dict and it cannot
    # resume Stage-1: s, r, y, p, z, eps, the optimiser moments and the schedule position are lost (Q10).  save_model keeps
    # writing that file unchanged (Stage-2 loads it); this is the engine's own, separate, complete state.
    def state_dict(self):
        mm, opt = self.minimax, self.optimizer
        return dict(
            format="uvc_amd.stage1.v1",
            # numerics the run was made with: the bf16 mode rounds the residual stream to bf16 unless resid_f32
            numerics=dict(precision=str(self.args.precision), residual_stream="float32" if (getattr(self.model, "precision", "") == "fp32" or getattr(self.model, "resid_f32", False)) else "bf16"),
            model=self.model.state_dict(),
            uvc=dict(s=mm.s.data.clone(), r=mm.r.data.clone(), y=mm.y.data.clone(), p=mm.p.data.clone(), z=mm.z.data.clone(),
                     gate_momentum=mm._gate_momentum.clone(), gate_gsum=mm._gate_gsum.clone(), gate_counters=mm._gate_counters.clone(),
                     gating_list_len=mm.gating_list_len, eps=float(self.model.eps),
                     patch_gating=None if mm.patch_gating is None else mm.patch_gating.data.clone()),
            adamw=dict(exp_avg=opt.exp_avg.clone(), exp_avg_sq=opt.exp_avg_sq.clone(), steps=dict(opt.steps),
                       lr=opt.param_groups[0]["lr"]),
            scheduler=self.scheduler.state_dict(),
            lrs=dict(s=self.s_opt.param_groups[0]["lr"], r=self.r_opt.param_groups[0]["lr"],
                     g=None if self.g_opt is None else self.g_opt.param_groups[0]["lr"],
                     dual=[g["lr"] for g in self.dual_opt.param_groups]),
            progress=dict(global_step=self.global_step, epoch=self.epoch, gating_grad_list_len=len(self.gating_grad_list),
                          enable_warmup=int(self.model.enable_warmup), args_enable_warmup=int(self.args.enable_warmup), micro=self._micro,
                          noise_step=int(self.noise.step), noise_site=int(self.noise.site), window_carried=int(getattr(self, "_window_carried", False))),
            # Gumbel / mixup draws continue where they stopped: torch CPU + this device's generator, numpy's global RNG
            rng=dict(torch_cpu=torch.get_rng_state(), torch_cuda=torch.cuda.get_rng_state(self.model._flat.device),
                     numpy=_np_rng_state()),
        )

    def load_state_dict(self, sd):
        if sd.get("format") != "uvc_amd.stage1.v1":
            raise ValueError("not a uvc_amd Stage-1 training state (the reference's checkpoint is the bare model state_dict: "
                             "load that with model.load_state_dict)")
        mm, opt = self.minimax, self.optimizer
        self.model.load_state_dict(sd["model"])
        u = sd["uvc"]
        for k in ("s", "r", "y", "p", "z"):
            getattr(mm, k).data.copy_(u[k])
        mm._gate_momentum.copy_(u["gate_momentum"]); mm._gate_gsum.copy_(u["gate_gsum"]); mm._gate_counters.copy_(u["gate_counters"])
        mm.gating_list_len = int(u["gating_list_len"])
        self.model.eps = float(u["eps"])
        if u.get("patch_gating") is not None and mm.patch_gating is not None:
            mm.patch_gating.data.copy_(u["patch_gating"])
        a = sd["adamw"]
        opt.exp_avg.copy_(a["exp_avg"]); opt.exp_avg_sq.copy_(a["exp_avg_sq"]); opt.steps = dict(a["steps"])
        self.scheduler.load_state_dict(sd["scheduler"])
        opt.param_groups[0]["lr"] = a["lr"]
        l = sd["lrs"]
        self.s_opt.param_groups[0]["lr"] = l["s"]; self.r_opt.param_groups[0]["lr"] = l["r"]
        if self.g_opt is not None and l["g"] is not None:
            self.g_opt.param_groups[0]["lr"] = l["g"]
        for g, v in zip(self.dual_opt.param_groups, l["dual"]):
            g["lr"] = v
        pr = sd["progress"]
        self.global_step, self.epoch = int(pr["global_step"]), int(pr["epoch"])
        self.gating_grad_list = [None] * int(pr["gating_grad_list_len"])
        self.model.enable_warmup = int(pr["enable_warmup"])
        self.args.enable_warmup = int(pr["args_enable_warmup"])
        self.model.block_skip_gating.requires_grad = not self.model.enable_warmup
        self._micro = int(pr.get("micro", 0))
        self.noise.step, self.noise.site = int(pr.get("noise_step", -1)), int(pr.get("noise_site", 0))
        self._window_carried = bool(pr.get("window_carried", 0))
        rng = sd.get("rng")
        if rng is not None:
            torch.set_rng_state(rng["torch_cpu"].cpu())
            torch.cuda.set_rng_state(rng["torch_cuda"].cpu(), self.model._flat.device)
            _np_set_rng_state(rng["numpy"])
        self.model.mark_weights_changed()


def _np_rng_state():
    import numpy as np
    k, keys, pos, has_gauss, cached = np.random.get_state()
    return dict(kind=k, keys=torch.from_numpy(keys.astype("int64")), pos=int(pos), has_gauss=int(has_gauss), cached=float(cached))


def _np_set_rng_state(st):
    import numpy as np
    np.random.set_state((st["kind"], st["keys"].cpu().numpy().astype("uint32"), st["pos"], st["has_gauss"], st["cached"]))
