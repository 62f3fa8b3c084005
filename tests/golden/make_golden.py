#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference); the GPU box only reads the .npz /
.json files this writes.  Usage:  python tests/golden/make_golden.py [scenario ...]

For every scenario of scenarios.py it
  * builds the inputs from the portable recipe (no reference involvement),
  * instantiates the reference's DistilledVisionTransformer / DistillationLoss /
    build_minimax_model / uvc_optimizer / prune_w_mask through ref_shim.py,
  * re-states the 40-line step body of UVC/joint_train.py:395-450 and get_uvc_layers
    (:530-564) -- joint_train.py itself cannot be imported (apex, timm.data, stty),
  * records every Tensor.exponential_ draw the reference consumed and the outputs named in
    SURVEY.md §8c (loss, logits, grad-norm, cur_resource, s r y p z, gate logits, per-block
    distrib, per-parameter checksums, post-prox W1/W3 slices, mask index sets, count_mask,
    final FLOPs ratio),
  * asserts that every least-k selection the reference made has a margin far above float32
    reduction noise, so the stored index sets do not depend on the machine's summation order.
"""
from __future__ import annotations

import json
import os
import sys
from argparse import Namespace
from functools import partial

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
import scenarios as SC  # noqa: E402
from oracle import vit as OV  # noqa: E402  (portable weight recipe only)
from oracle import uvc as OU  # noqa: E402  (fp64 scores for the margin check only)

ref_shim.install()
from models.model_distilled import DistilledVisionTransformer  # noqa: E402
from utils.losses import DistillationLoss  # noqa: E402
from utils.scheduler import WarmupCosineSchedule  # noqa: E402
import uvc_utils  # noqa: E402
import uvc_optimizer as uvc_opt_mod  # noqa: E402


class SoftTargetCrossEntropy(nn.Module):
    """timm.loss.SoftTargetCrossEntropy (absent from the image; one-liner)."""

    def forward(self, x, target):
        return torch.sum(-target * torch.nn.functional.log_softmax(x, dim=-1), dim=-1).mean()


def get_uvc_layers(model):
    """Re-statement of joint_train.py:530-564 (module-name matching order)."""
    uvc_layers = {"W1": [], "W2": [], "W3": []}
    layer_names = {None: None}
    for name, m in model.named_modules():
        if hasattr(m, "in_features"):
            if "mlp.fc2" in name or "attn.proj" in name:
                layer_names[m] = name
                (uvc_layers["W1"] if "attn.proj" in name else uvc_layers["W3"]).append(m)
                m.uvc_s = 0
            if "mlp.fc1" in name:
                layer_names[m] = name
                uvc_layers["W2"].append(m)
                m.uvc_s = 0
    d = {"s_dict": {}, "r_dict": {}}
    for i, m in enumerate(uvc_layers["W1"]):
        d["s_dict"][m] = [i, 0]
        d["r_dict"][m] = i
    for i, m in enumerate(uvc_layers["W3"]):
        d["s_dict"][m] = [i, 1]
    return layer_names, uvc_layers, d


def count_mask(model):
    total = 0
    for _, p in model.named_modules():
        if hasattr(p, "mask"):
            total += p.mask.sum()
    return total / 1e6


def build_model(m, r, student=True):
    kw = dict(patch_size=m["patch_size"], embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"],
              mlp_ratio=m["mlp_ratio"], qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), drop_rate=0,
              img_size=m["img_size"], num_classes=m["num_classes"])
    if student:   # joint_train.py:135-140
        return DistilledVisionTransformer(enable_dist=m["enable_dist"], gumbel_hard=False, **kw)
    return DistilledVisionTransformer(enable_dist=m["enable_dist"], **kw)   # :957-961


def margins_ok(minimax, tag, min_margin=3e-6):
    """Every topk boundary used by prox/mask/loss: (1) the reference's own float32 scores and the
    oracle's float64-accumulated scores must select the SAME index sets, and (2) the boundary
    must be separated by more than float32 reduction noise, so that this holds on any machine."""
    worst = np.inf
    s = minimax.s.data.ceil()
    rr = minimax.r.data.ceil()
    H, hd = minimax.num_heads, minimax.head_size
    for l, (w1, w3) in enumerate(zip(minimax.uvc_layers["W1"], minimax.uvc_layers["W3"])):
        s1, s2 = OU.scores_w1(w1.weight.data, H, hd)
        s3 = OU.scores_w3(w3.weight.data)
        f1, f2 = uvc_utils.weight_list_to_scores(w1, "W1", hd)
        f3 = uvc_utils.weight_list_to_scores(w3, "W3")
        for sc, ref, k in [(s2, f2, int(s[l, 0])), (s3, f3, int(s[l, 1]))] + \
                          [(s1[h], f1[h], int(rr[l, h])) for h in range(H)]:
            for kk in (k, k + 1):                    # the set boundary and the (k+1)-th value
                if 0 < kk < sc.numel():
                    a = set(torch.topk(ref, kk, largest=False)[1].tolist())
                    b = set(torch.nonzero(OU.least_k(sc, kk)[0]).flatten().tolist())
                    assert a == b, f"{tag}: reference fp32 scores and fp64 scores select different sets"
                    srt = torch.sort(sc.double())[0]
                    gap = (srt[kk] - srt[kk - 1]).item() / max(srt[kk].item(), 1e-30)
                    worst = min(worst, gap)
    assert worst > min_margin, f"{tag}: selection margin {worst:.2e} too close to float32 reduction noise; change the seed"
    return worst


def run_scenario(name):
    r = SC.recipe(name)
    m = r["model_cfg"]
    cfg = OV.VitConfig(img_size=m["img_size"], patch_size=m["patch_size"], num_classes=m["num_classes"],
                       embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"],
                       mlp_ratio=m["mlp_ratio"], enable_dist=m["enable_dist"])
    L, H, hd, F = cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden
    params = OV.init_params_numpy(cfg, r["seed"], r["enable_patch_gating"], weight_gain=m["weight_gain"])
    tparams = OV.init_params_numpy(cfg, r["seed"] + 500, 0, weight_gain=m["weight_gain"])
    x_all, y_all = SC.make_inputs(r)

    torch.manual_seed(r["seed"])
    model = build_model(m, r, student=True)
    if r["enable_patch_gating"] == 1:
        model.patch_gating = nn.Parameter(torch.zeros(1, cfg.num_patches, 1))
    missing = model.load_state_dict(params, strict=False)
    assert not missing.unexpected_keys and all("mask" in k for k in missing.missing_keys), missing
    for _, p in model.named_modules():                      # joint_train.py:169-171
        if hasattr(p, "weight"):
            p.register_buffer("mask", torch.ones_like(p.weight))
    total_param = float(count_mask(model))
    teacher = build_model(m, r, student=False)
    teacher.load_state_dict({k: v for k, v in tparams.items() if k != "patch_gating"}, strict=False)
    teacher.eval()
    criterion = DistillationLoss(SoftTargetCrossEntropy(), teacher, "soft", r["distillation_alpha"],
                                 r["distillation_tau"])
    args = Namespace(eps_decay=r["eps_decay"], enable_patch_gating=r["enable_patch_gating"], enable_part_gating=0,
                     enable_block_gating=r["enable_block_gating"], head_size=hd, num_heads=H, flops_with_mhsa=1,
                     use_gumbel=r["use_gumbel"], enable_jumping=0, eps=r["eps"], enable_warmup=r["warmup"],
                     soptim="sgd", roptim="sgd", slr=r["slr"], rlr=r["rlr"], glr=r["glr"],
                     zlr_schedule_list=[int(r["zlr"])], ylr=r["ylr"], plr=r["plr"], budget=r["budget"],
                     sl2wd=r["sl2wd"], gating_weight=r["gating_weight"], patch_ratio=r["patch_ratio"])
    layer_names, uvc_layers, uvc_layers_dict = get_uvc_layers(model)
    with torch.no_grad():                                    # joint_train.py:1010-1012
        model.eval()
        _, flops_list = model(torch.ones(1, 3, cfg.img_size, cfg.img_size), number=r["patch_ratio"])
    minimax, dual_opt, s_opt, r_opt, gating_opt = uvc_opt_mod.build_minimax_model(
        model, layer_names, uvc_layers, uvc_layers_dict, args, flops_list)
    s0, r0, y0, p0, z0 = SC.initial_state(r, L, H, hd, F)
    minimax.s.data.copy_(torch.from_numpy(s0)); minimax.r.data.copy_(torch.from_numpy(r0))
    minimax.y.data.copy_(torch.from_numpy(y0)); minimax.p.data.copy_(torch.from_numpy(p0))
    minimax.z.data.fill_(float(z0))
    optimizer = torch.optim.AdamW(model.parameters(), lr=r["learning_rate"], weight_decay=r["weight_decay"])
    scheduler = WarmupCosineSchedule(optimizer, warmup_steps=r["warmup_steps"], t_total=r["t_total"])
    model.train()
    out = dict(embed_macs=np.int64(flops_list[0]), macs_list=np.array(flops_list[1], dtype=np.int64),
               resource_ub=np.float64(minimax.resource_fn.__closure__ and 0.0), total_param=np.float64(total_param))
    out["resource_ub"] = np.float64(float(uvc_utils.calc_flops(
        torch.zeros(L, 2), torch.zeros(L, H), uvc_layers_dict, uvc_layers, hd, None, None, flops_list,
        (None, None, None), 0, None)))
    # epoch header (joint_train.py:337-386)
    gating_grad_list = []
    if r["warmup"]:
        minimax.model.enable_warmup = 1
        minimax.model.block_skip_gating.requires_grad = False
        for g in optimizer.param_groups:
            g["lr"] = r["warmup_lr"]
    else:
        minimax.model.enable_warmup = 0
        minimax.model.block_skip_gating.requires_grad = True
    mm_ = r.get("min_margin", 3e-6)
    margins = [margins_ok(minimax, name + ":start", mm_)]
    uvc_utils.prune_w_mask(minimax, optimizer)
    out["mask0_count"] = np.float64(float(count_mask(model)))
    global_step = 0
    names = [k for k, _ in model.named_parameters()]
    out["param_names"] = np.array(names)
    with ref_shim.RecordExponential() as rec:
        for step in range(r["steps"]):
            n0 = len(rec.draws)
            x = torch.from_numpy(x_all[step]); y = torch.from_numpy(y_all[step])
            tau = r["patch_tau"] if r["enable_patch_gating"] == 2 else -1
            outputs, _ = model(x, tau, r["patch_ratio"])
            loss = criterion(x, outputs, y)
            loss.backward()
            gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), r["max_grad_norm"])
            gsum = np.array([float(p.grad.double().abs().sum()) if p.grad is not None else np.nan
                             for _, p in model.named_parameters()])
            optimizer.step()
            scheduler.step()
            global_step += 1
            minimax.update_gating()
            margins.append(margins_ok(minimax, f"{name}:pre-prox{step}", mm_))
            cur, s_np, r_np, g_np, gating_grad_list = uvc_opt_mod.uvc_optimizer(
                optimizer, minimax, s_opt, r_opt, gating_opt, dual_opt, args, {}, [], flops_list,
                r["z_grad_clip"], global_step, r["gating_interval"], gating_grad_list)
            margins.append(margins_ok(minimax, f"{name}:post{step}", mm_))
            optimizer.zero_grad()
            draws = rec.draws[n0:]
            pre = f"step{step}."
            out[pre + "n_draws"] = np.int64(len(draws))
            for i, d in enumerate(draws):
                out[pre + f"draw{i}"] = d.numpy()
            out[pre + "loss"] = np.float64(loss.item())
            out[pre + "logits"] = outputs[0].detach().numpy()
            out[pre + "logits_dist"] = outputs[1].detach().numpy()
            out[pre + "grad_norm"] = np.float64(float(gnorm))
            out[pre + "grad_abs_sum"] = gsum
            out[pre + "cur_resource"] = np.float64(cur)
            out[pre + "lr"] = np.float64(optimizer.param_groups[0]["lr"])
            out[pre + "s"] = minimax.s.data.numpy().copy()
            out[pre + "r"] = minimax.r.data.numpy().copy()
            out[pre + "y"] = minimax.y.data.numpy().copy()
            out[pre + "p"] = minimax.p.data.numpy().copy()
            out[pre + "z"] = np.float64(minimax.z.item())
            out[pre + "gating"] = model.block_skip_gating.data.numpy().copy()
            out[pre + "param_sum"] = np.array([float(p.data.double().sum()) for _, p in model.named_parameters()])
            out[pre + "param_abs_sum"] = np.array([float(p.data.double().abs().sum()) for _, p in model.named_parameters()])
            out[pre + "w1_0_row0"] = uvc_layers["W1"][0].weight.data[0].numpy().copy()
            out[pre + "w3_0_row0"] = uvc_layers["W3"][0].weight.data[0].numpy().copy()
        # epoch end (joint_train.py:500-509)
        uvc_utils.prune_w_mask(minimax, optimizer)
        out["mask_count"] = np.float64(float(count_mask(model)))
        for l in range(L):
            out[f"keep_proj.{l}"] = np.packbits(uvc_layers["W1"][l].mask.data[0].numpy().astype(np.uint8))
            out[f"keep_fc2.{l}"] = np.packbits(uvc_layers["W3"][l].mask.data[0].numpy().astype(np.uint8))
            assert torch.equal(uvc_layers["W2"][l].mask.data[:, 0], uvc_layers["W3"][l].mask.data[0])
            assert bool((uvc_layers["W1"][l].mask.data == uvc_layers["W1"][l].mask.data[0:1]).all())
        n0 = len(rec.draws)
        out["real_flops"] = np.float64(float(minimax.run_resource_fn(gumbel_hard=True)))
        out["expect_flops"] = np.float64(float(minimax.run_resource_fn(gumbel_hard=False)))
        for i, d in enumerate(rec.draws[n0:]):
            out[f"final.draw{i}"] = d.numpy()
    out["min_margin"] = np.float64(min(margins))
    keys = sorted(model.state_dict().keys())
    out["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    out["state_dict_shapes"] = np.array([json.dumps(list(v.shape)) for v in model.state_dict().values()])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)  loss0={out['step0.loss']:.6f} "
          f"cur0={out['step0.cur_resource']:.6f} mask={out['mask_count']:.6f}/{total_param:.6f} margin={min(margins):.2e}")


if __name__ == "__main__":
    todo = sys.argv[1:] or list(SC.SCENARIOS)
    for n in todo:
        run_scenario(n)
