import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import torch, numpy as np
from test_t2t_model_gpu import build
from oracle import t2t as OT
for name in ("t2t_micro", "t2t_micro_skip"):
    r, cfg, sd, m, x, g = build(name, "fp32")
    m.eval()
    with torch.no_grad():
        le, _ = m(x.cuda())
    tok_e = m._ws_view(x.shape[0], False, "pe").clone()
    m.train()
    (lt, _), _ = m(x.cuda())
    tok_t = m._ws_view(x.shape[0], True, "pe").clone()
    print(name, "tok diff", float((tok_e - tok_t).abs().max()), "logit diff", float((le - lt).abs().max()))
    with torch.no_grad():
        (lt2, _), _ = m(x.cuda())
    print("  train-mode no_grad logits diff vs eval", float((le - lt2).abs().max()))
