"""Scenario recipes shared by the golden generator (make_golden.py, runs the REFERENCE) and the
tests (run the oracle / the HIP path).  Pure data + portable generators: nothing here reads
/root/reference, so it is usable on the GPU box.

Inputs are generated from numpy's frozen legacy RandomState stream so that every machine
builds bit-identical weights/images/targets from the recipe; only the reference's OUTPUTS
(and the RNG draws it consumed) live in the .npz fixtures.
"""
from __future__ import annotations

import numpy as np

# name -> recipe
SCENARIOS = {
    # micro model: every code path, small enough that fixtures + CPU tests are instant
    "micro_warmup": dict(model="micro", batch=4, steps=2, warmup=1, state="zero", seed=11),
    "micro_train": dict(model="micro", batch=4, steps=3, warmup=0, state="zero", seed=12,
                        gating_interval=2),
    "micro_pruned": dict(model="micro", batch=4, steps=4, warmup=0, state="pruned", seed=13,
                         gating_interval=3, warmup_steps=2),
    "micro_clip": dict(model="micro", batch=4, steps=2, warmup=0, state="pruned", seed=14,
                       z_grad_clip=0.01, gating_interval=2, warmup_steps=1),
    "micro_bounds": dict(model="micro", batch=4, steps=3, warmup=0, state="bounds", seed=15,
                         gating_interval=2, sl2wd=0.01),
    "micro_softl0": dict(model="micro", batch=4, steps=2, warmup=0, state="pruned", seed=16,
                         use_gumbel=0, gating_interval=2),
    "micro_deit": dict(model="micro_dist", batch=4, steps=2, warmup=0, state="pruned", seed=17,
                       gating_interval=2),
    "micro_patch2": dict(model="micro", batch=4, steps=2, warmup=0, state="pruned", seed=28,
                         gating_interval=2, enable_patch_gating=2, patch_tau=0.7),
    "micro_patch1": dict(model="micro", batch=4, steps=2, warmup=0, state="pruned", seed=19,
                         gating_interval=2, enable_patch_gating=1),
    # BASELINE config 1: DeiT-Tiny patch16 224, batch 8, budget .5, 2 Stage-1 steps on CPU
    "tiny8_train": dict(model="deit_tiny", batch=8, steps=2, warmup=0, state="zero", seed=730),
    "tiny8_pruned": dict(model="deit_tiny", batch=8, steps=2, warmup=0, state="pruned", seed=731,
                         gating_interval=2, warmup_steps=1),
    # BASELINE configs 3 / 4 at a CPU-sized batch: DeiT-Small budget .58; DeiT-Base with the distillation token
    "small2_pruned": dict(model="deit_small", batch=2, steps=1, warmup=0, state="pruned", seed=735, budget=0.58,
                          gating_interval=2, warmup_steps=1),
    # 3072 fc2 columns per layer: the densest score spectrum of the fixtures, so the margin floor is lower here;
    # make_golden.py still asserts that the reference's float32 and the float64 scores pick identical sets
    "base2_deit": dict(model="deit_base_dist", batch=2, steps=1, warmup=0, state="pruned", seed=734,
                       gating_interval=2, warmup_steps=1, min_margin=4e-7),
}

MODELS = {
    "micro": dict(img_size=64, patch_size=16, num_classes=16, embed_dim=128, depth=2, num_heads=2,
                  mlp_ratio=4.0, enable_dist=0, weight_gain=3.0),
    "micro_dist": dict(img_size=64, patch_size=16, num_classes=16, embed_dim=128, depth=2, num_heads=2,
                       mlp_ratio=4.0, enable_dist=1, weight_gain=3.0),
    "deit_tiny": dict(img_size=224, patch_size=16, num_classes=1000, embed_dim=192, depth=12,
                      num_heads=3, mlp_ratio=4.0, enable_dist=0, weight_gain=2.0),
    "deit_small": dict(img_size=224, patch_size=16, num_classes=1000, embed_dim=384, depth=12,
                       num_heads=6, mlp_ratio=4.0, enable_dist=0, weight_gain=1.5),
    "deit_base_dist": dict(img_size=224, patch_size=16, num_classes=1000, embed_dim=768, depth=12,
                           num_heads=12, mlp_ratio=4.0, enable_dist=1, weight_gain=1.0),
}

# README command (run_uvc_train.sh:4-38) hyper-parameters
DEFAULTS = dict(budget=0.5, slr=0.02, rlr=0.02, glr=0.1, ylr=1e-4, plr=1e-4, zlr=1, sl2wd=0.0,
                z_grad_clip=0.5, gating_interval=50, gating_weight=5e-4, use_gumbel=1,
                enable_block_gating=1, eps=0.1, eps_decay=0.92, learning_rate=1e-4, weight_decay=0.05,
                max_grad_norm=1.0, warmup_steps=500, t_total=150150, warmup_lr=1e-4,
                distillation_alpha=0.1, distillation_tau=1.0, enable_patch_gating=0, patch_ratio=0.9,
                patch_tau=-1.0)


def recipe(name: str) -> dict:
    r = dict(DEFAULTS)
    r.update(SCENARIOS[name])
    r["name"] = name
    r["model_cfg"] = dict(MODELS[r["model"]])
    return r


def make_inputs(r: dict):
    """x[steps,B,3,S,S] ~ N(0,1) float32, y_soft[steps,B,C] = softmax(2*N(0,1)) (stand-in for
    mixup+smoothing targets; the parity boundary starts after mixup_fn)."""
    m = r["model_cfg"]
    rs = np.random.RandomState(r["seed"] + 1000)
    B, S, C = r["batch"], m["img_size"], m["num_classes"]
    x = rs.standard_normal((r["steps"], B, 3, S, S)).astype(np.float32)
    logits = 2.0 * rs.standard_normal((r["steps"], B, C))
    e = np.exp(logits - logits.max(-1, keepdims=True))
    y = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return x, y


def initial_state(r: dict, L: int, H: int, hd: int, F: int):
    """Primal/dual start.  'zero' = the reference's init (uvc_utils.py:141-148).  'pruned' =
    the non-trivial state of SURVEY.md §8c so prox/mask/dual kernels do real work.
    'bounds' = entries sitting on both box bounds."""
    rs = np.random.RandomState(r["seed"] + 2000)
    s = np.zeros((L, 2), np.float32)
    rr = np.zeros((L, H), np.float32)
    y = np.full((L, 2), 1e-3, np.float32)
    p = np.full((L, H), 1e-3, np.float32)
    z = np.float32(1e-3)
    if r["state"] in ("pruned", "bounds"):
        s[:, 0] = rs.uniform(0, 0.6 * (H - 1) + 0.3, L)
        s[:, 1] = rs.uniform(0, 0.5 * F, L)
        rr[:] = rs.uniform(0, 30.0, (L, H))
        y[:] = rs.uniform(0.5, 3.0, (L, 2))
        p[:] = rs.uniform(0.5, 3.0, (L, H))
        z = np.float32(2.0)
    if r["state"] == "bounds":
        s[0, 0] = H - 1           # == s_max
        s[0, 1] = 0.0
        s[-1, 1] = F - 1
        rr[0, 0] = hd - 1
        rr[-1, -1] = 0.0
        y[0, 0] = 0.0
    return s, rr, y, p, z


# ----------------------------------------------------------------------------------------------------
# Stage-2 (post_train.py) scenarios: masked fine-tune steps from a pruned state with hard-skipped blocks.
STAGE2 = {
    "stage2_micro": dict(model="micro", batch=4, steps=3, seed=41, skip_blocks=[], epoch_of_step=[0, 1, 1]),
    "stage2_micro_skip": dict(model="micro", batch=4, steps=3, seed=42, skip_blocks=[1], epoch_of_step=[1, 1, 2]),
    "stage2_micro_none": dict(model="micro", batch=4, steps=2, seed=43, skip_blocks=[0], epoch_of_step=[2, 2],
                              distillation_type="none"),
    "stage2_micro_deit": dict(model="micro_dist", batch=4, steps=2, seed=44, skip_blocks=[], epoch_of_step=[1, 3]),
    # BASELINE config 1 shape: DeiT-Tiny, batch 8, two blocks skipped
    "stage2_tiny8": dict(model="deit_tiny", batch=8, steps=2, seed=740, skip_blocks=[3, 7], epoch_of_step=[1, 1]),
}

# run_post_train.sh + post_train.py defaults; learning_rate is chosen so the batch-scaled lr (lr*batch/512) is ~1e-3
# and three steps move the weights measurably; one warm-up epoch so both schedule branches are hit
STAGE2_DEFAULTS = dict(learning_rate=0.064, weight_decay=0.05, max_grad_norm=1.0, epochs=10, warmup_epochs=1,
                       warmup_lr=1e-6, min_lr=1e-5, decay_rate=0.1, opt_eps=1e-8, distillation_type="soft",
                       distillation_alpha=0.1, distillation_tau=1.0)


def stage2_recipe(name: str) -> dict:
    r = dict(STAGE2_DEFAULTS)
    r.update(STAGE2[name])
    r["name"] = name
    r["model_cfg"] = dict(MODELS[r["model"]])
    return r


def stage2_masks(r: dict, L: int, H: int, hd: int, F: int):
    """Structured masks as prune_w_mask leaves them (uvc_utils.py:376-401) and the gate logits of a finished
    Stage-1: per layer a 0/1 keep-vector over attn.proj input columns (whole heads + columns inside the kept heads)
    and over the MLP hidden units (fc2 input columns = fc1 output rows); block_skip_gating = [1,-1] for skipped
    blocks, [-1,1] otherwise."""
    rs = np.random.RandomState(r["seed"] + 3000)
    keep_proj = np.ones((L, H * hd), np.float32)
    keep_hidden = np.ones((L, F), np.float32)
    for l in range(L):
        dead_heads = rs.choice(H, size=rs.randint(0, H), replace=False)
        for h in range(H):
            if h in dead_heads:
                keep_proj[l, h * hd:(h + 1) * hd] = 0
            else:
                k = rs.randint(0, hd // 2)
                keep_proj[l, h * hd + rs.choice(hd, size=k, replace=False)] = 0
        keep_hidden[l, rs.choice(F, size=rs.randint(1, F // 2), replace=False)] = 0
    gate = np.tile(np.array([-1.0, 1.0], np.float32), (L, 1))
    for b in r["skip_blocks"]:
        gate[b] = [1.0, -1.0]
    return keep_proj, keep_hidden, gate
