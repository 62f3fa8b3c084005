#!/usr/bin/env python3
"""Generate the T2T-ViT fixtures (t2t_*.npz) by running the REFERENCE's own T2T_ViT (UVC/T2TViT/models/t2t_vit.py) on
the portable weight / input recipes of t2t_scenarios.py.  Runs only in the build container (needs /root/reference);
the GPU box reads the .npz files.  Usage:  python tests/golden/make_t2t_golden.py [scenario ...]

Only the UNGATED forward exists in the reference (the gated one raises): the fixtures hold the token-to-token
module's output, the two Performer stages' outputs, the eval-mode logits and the MAC table.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shim  # noqa: E402
import t2t_scenarios as TS  # noqa: E402
from oracle import t2t as OT  # noqa: E402  (portable weight recipe only)

ref_shim.install()
from T2TViT.models.t2t_vit import T2T_ViT  # noqa: E402


def run(name):
    r = TS.recipe(name)
    cfg = OT.T2TConfig(**r["model_cfg"])
    sd = OT.init_params_numpy(cfg, r["seed"], weight_gain=r["weight_gain"])
    for i in r.get("skip_blocks", []):
        sd["block_skip_gating"][i] = torch.tensor([1.0, -1.0])
    x = torch.from_numpy(TS.make_input(r))
    model = T2T_ViT(img_size=cfg.img_size, tokens_type="performer", num_classes=cfg.num_classes, embed_dim=cfg.embed_dim,
                    depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, token_dim=cfg.token_dim)
    # t2t_vit.py:139 builds block_skip_gating with .expand(): all L rows alias ONE [2] storage, so copy_ into it raises;
    # the harness replaces it with a real [L, 2] parameter before loading
    model.block_skip_gating = torch.nn.Parameter(torch.zeros(cfg.depth, 2))
    model.load_state_dict(sd, strict=True)
    taps = {}
    t2t = model.tokens_to_token
    h1 = t2t.attention1.register_forward_hook(lambda m, i, o: taps.__setitem__("attention1", o[0].detach()))
    h2 = t2t.attention2.register_forward_hook(lambda m, i, o: taps.__setitem__("attention2", o[0].detach()))
    h3 = t2t.register_forward_hook(lambda m, i, o: taps.__setitem__("tokens", o[0].detach()))
    model.eval()
    with torch.no_grad():
        logits, (macs_embed, macs_list) = model(x)
    for h in (h1, h2, h3):
        h.remove()
    # training mode returns (x, x) (:205-206) and switches on the Performer's Dropout(0.1) layers (token_performer.py:13,24):
    # RNG-dependent, so only the eval forward is stored
    out = dict(logits=logits.numpy(), tokens=taps["tokens"].numpy(),
               macs_embed=np.int64(macs_embed), macs_list=np.array([m if m else [0] * 6 for m in macs_list], dtype=np.int64),
               state_dict_keys=np.array(list(model.state_dict().keys())),
               state_dict_shapes=np.array([str(list(v.shape)) for v in model.state_dict().values()]))
    if r.get("store_stages", True):
        out["attention1"] = taps["attention1"].numpy()
        out["attention2"] = taps["attention2"].numpy()
    else:                                            # full-size: checksums + a slice
        for k in ("attention1", "attention2"):
            t = taps[k]
            out[k + "_abs_sum"] = np.float64(t.double().abs().sum())
            out[k + "_head"] = t[:, :64].numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path)/1024:.1f} KiB) macs_embed={macs_embed} block macs={macs_list[0] or macs_list[-1]} "
          f"|logits|={float(logits.abs().mean()):.4f}")


if __name__ == "__main__":
    for n in sys.argv[1:] or list(TS.SCENARIOS):
        run(n)
