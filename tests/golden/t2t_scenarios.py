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
