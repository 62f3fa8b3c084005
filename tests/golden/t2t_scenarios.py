"""T2T-ViT fixture recipes shared by make_t2t_golden.py (runs the REFERENCE) and the tests.  Pure data + portable
generators (numpy's frozen RandomState): nothing here reads /root/reference."""
from __future__ import annotations

import numpy as np

SCENARIOS = {
    # micro: 64x64 image -> 256 -> 64 -> 16 tokens, two blocks
    "t2t_micro": dict(model="micro", batch=3, seed=41, weight_gain=3.0),
    "t2t_micro_skip": dict(model="micro3", batch=2, seed=42, weight_gain=3.0, skip_blocks=[1]),
    # BASELINE config 5's model at batch 1
    "t2t_14_b1": dict(model="t2t_vit_14", batch=1, seed=43, weight_gain=2.0, store_stages=False),
}
MODELS = {
    "micro": dict(img_size=64, num_classes=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=3.0, token_dim=64),
    "micro3": dict(img_size=64, num_classes=16, embed_dim=128, depth=3, num_heads=2, mlp_ratio=3.0, token_dim=64),
    "t2t_vit_14": dict(img_size=224, num_classes=1000, embed_dim=384, depth=14, num_heads=6, mlp_ratio=3.0, token_dim=64),
}


def recipe(name):
    r = dict(SCENARIOS[name])
    r["name"] = name
    r["model_cfg"] = dict(MODELS[r["model"]])
    return r


def make_input(r):
    m = r["model_cfg"]
    rs = np.random.RandomState(r["seed"] + 1000)
    return rs.standard_normal((r["batch"], 3, m["img_size"], m["img_size"])).astype(np.float32)


# Stage-1 steps on T2T-ViT.  The reference cannot run these: oracle (oracle/step.py with oracle/t2t.py's forward)
# against the HIP path, no reference fixture.
STAGE1 = {
    "t2t_micro_train": dict(model="micro", batch=4, steps=3, warmup=0, state="pruned", seed=51, gating_interval=2, warmup_steps=1, weight_gain=3.0),
    "t2t_micro_warmup": dict(model="micro", batch=4, steps=2, warmup=1, state="zero", seed=52, weight_gain=3.0),
    "t2t_micro_softl0": dict(model="micro", batch=4, steps=2, warmup=0, state="pruned", seed=53, use_gumbel=0, gating_interval=2, weight_gain=3.0),
    # BASELINE config 5's flags: patch gating (Gumbel top-k, mode 2) + block gating; 16 tokens, k = int(.9 * 16) = 14
    "t2t_micro_patch2": dict(model="micro", batch=4, steps=3, warmup=0, state="pruned", seed=54, gating_interval=2, warmup_steps=1, weight_gain=3.0,
                             enable_patch_gating=2, patch_tau=0.7),
}


def stage1_recipe(name):
    import scenarios as SC
    r = dict(SC.DEFAULTS)
    r.update(STAGE1[name])
    r["name"] = name
    r["model_cfg"] = dict(MODELS[r["model"]])
    return r


def stage1_inputs(r):
    m = r["model_cfg"]
    rs = np.random.RandomState(r["seed"] + 1000)
    x = rs.standard_normal((r["steps"], r["batch"], 3, m["img_size"], m["img_size"])).astype(np.float32)
    z = 2.0 * rs.standard_normal((r["steps"], r["batch"], m["num_classes"]))
    e = np.exp(z - z.max(-1, keepdims=True))
    y = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    return x, y


def stage1_draws(r, depth):
    """Exp(1) draws per step: `depth` block-gate draws [2] for the student forward, then e1 / e2 [depth, 2] of the two resource
    evaluations inside uvc_optimizer (SURVEY 8c note 3)."""
    rs = np.random.RandomState(r["seed"] + 2000)
    rp = np.random.RandomState(r["seed"] + 2500)          # patch-gating draws [B, P] come first in the model's list (DeiT's RNG order)
    P = (r["model_cfg"]["img_size"] // 16) ** 2
    out = []
    for _ in range(r["steps"]):
        md = [rs.exponential(size=2).astype(np.float32) for _ in range(depth)]
        if r.get("enable_patch_gating", 0) == 2:
            md = [rp.exponential(size=(r["batch"], P)).astype(np.float32)] + md
        e1 = rs.exponential(size=(depth, 2)).astype(np.float32)
        e2 = rs.exponential(size=(depth, 2)).astype(np.float32)
        out.append((md, e1, e2))
    return out


# Stage-2 masked fine-tune steps on T2T-ViT: the reference CAN run these (post_train.py:165-167 builds t2t_vit_14() with the
# default flags = hard block skip, and calls model(x)); make_t2t_stage2_golden.py runs its own T2T_ViT, loss and autograd
# with the Performer's Dropout(0.1) layers set to p = 0 (RNG-free fixture; the engine does not apply them, NOTEBOOK 10b).
STAGE2 = {
    "t2t_stage2_micro": dict(model="micro3", batch=4, steps=2, seed=61, skip_blocks=[1], epoch_of_step=[1, 2], weight_gain=3.0),
}


def stage2_recipe(name):
    import scenarios as SC
    r = dict(SC.STAGE2_DEFAULTS)
    r.update(STAGE2[name])
    r["name"] = name
    r["model_cfg"] = dict(MODELS[r["model"]])
    return r
