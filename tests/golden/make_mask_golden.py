#!/usr/bin/env python3
"""Fixture for the mask life cycle of prune_w_mask (uvc_utils.py:376-401) across calls whose index sets are NOT monotone.

Runs only in the build container (needs /root/reference).  The reference resets the attn.proj (W1) and mlp.fc2 (W3) masks
to 1 on every call but only ever WRITES ZEROS into the mlp.fc1 (W2) mask (:401), so the fc1 mask is the sticky union of every
pruned set seen so far.  This drives the reference's own prune_w_mask through three primal states (pruned -> fewer pruned ->
different sets) on the micro model and records the three mask families after each call.

    python tests/golden/make_mask_golden.py        -> tests/golden/mask_sticky_micro.npz
"""
from __future__ import annotations

import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (installs the shim, imports the reference modules)
import scenarios as SC  # noqa: E402
from oracle import vit as OV  # noqa: E402


def states(r, L, H, hd, F):
    """Three primal states: the scenario's 'pruned' start, then s / r scaled down (sets shrink), then a fresh draw."""
    s0, r0, *_ = SC.initial_state(r, L, H, hd, F)
    rs = np.random.RandomState(r["seed"] + 4000)
    s1, r1 = (s0 * np.array([1.0, 0.4], np.float32)).astype(np.float32), (r0 * 0.5).astype(np.float32)
    s2 = s0.copy(); s2[:, 1] = rs.uniform(0, 0.5 * F, L); s2[:, 0] = rs.uniform(0, 0.6 * (H - 1) + 0.3, L)
    r2 = rs.uniform(0, 30.0, (L, H)).astype(np.float32)
    return [(s0, r0), (s1, r1), (s2.astype(np.float32), r2)]


def main(name="micro_pruned"):
    r = SC.recipe(name)
    m = r["model_cfg"]
    cfg = OV.VitConfig(img_size=m["img_size"], patch_size=m["patch_size"], num_classes=m["num_classes"], embed_dim=m["embed_dim"],
                       depth=m["depth"], num_heads=m["num_heads"], mlp_ratio=m["mlp_ratio"], enable_dist=m["enable_dist"])
    L, H, hd, F = cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden
    params = OV.init_params_numpy(cfg, r["seed"], 0, weight_gain=m["weight_gain"])
    model = MG.build_model(m, r, student=True)
    model.load_state_dict(params, strict=False)
    for _, p in model.named_modules():
        if hasattr(p, "weight"):
            p.register_buffer("mask", torch.ones_like(p.weight))
    args = Namespace(eps_decay=r["eps_decay"], enable_patch_gating=0, enable_part_gating=0, enable_block_gating=1, head_size=hd, num_heads=H,
                     flops_with_mhsa=1, use_gumbel=1, enable_jumping=0, eps=r["eps"], enable_warmup=0, soptim="sgd", roptim="sgd", slr=r["slr"],
                     rlr=r["rlr"], glr=r["glr"], zlr_schedule_list=[1], ylr=r["ylr"], plr=r["plr"], budget=r["budget"], sl2wd=0.0,
                     gating_weight=r["gating_weight"], patch_ratio=0.9)
    names, layers, ldict = MG.get_uvc_layers(model)
    with torch.no_grad():
        model.eval()
        _, flops = model(torch.ones(1, 3, cfg.img_size, cfg.img_size), number=0.9)
    minimax, *_ = MG.uvc_opt_mod.build_minimax_model(model, names, layers, ldict, args, flops)
    out = dict(scenario=np.array(name))
    for i, (s, rr) in enumerate(states(r, L, H, hd, F)):
        minimax.s.data.copy_(torch.from_numpy(s)); minimax.r.data.copy_(torch.from_numpy(rr))
        MG.margins_ok(minimax, f"mask_sticky:{i}")
        MG.uvc_utils.prune_w_mask(minimax, None)
        out[f"call{i}.s"], out[f"call{i}.r"] = s, rr
        out[f"call{i}.count"] = np.float64(float(MG.count_mask(model)))
        for l in range(L):
            w1, w2, w3 = layers["W1"][l].mask.data, layers["W2"][l].mask.data, layers["W3"][l].mask.data
            assert bool((w1 == w1[0:1]).all()) and bool((w3 == w3[0:1]).all()) and bool((w2 == w2[:, 0:1]).all())
            out[f"call{i}.keep_proj.{l}"] = np.packbits(w1[0].numpy().astype(np.uint8))
            out[f"call{i}.keep_fc2.{l}"] = np.packbits(w3[0].numpy().astype(np.uint8))
            out[f"call{i}.keep_fc1.{l}"] = np.packbits(w2[:, 0].numpy().astype(np.uint8))
    # the point of the fixture: after call 1 some fc1 rows are masked although their fc2 column no longer is
    diff = sum(int((np.unpackbits(out[f"call1.keep_fc1.{l}"]) != np.unpackbits(out[f"call1.keep_fc2.{l}"])).sum()) for l in range(L))
    assert diff > 0, "states do not exercise the sticky fc1 mask"
    path = os.path.join(HERE, "mask_sticky_micro.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {diff} fc1 rows stay masked after their fc2 column was released")


if __name__ == "__main__":
    main()
