"""CPU: bench.py's provenance logic and the numerics claim behind the one-byte GELU'(a) code (no GPU, no library call).

* `profiles/MANIFEST.json` names the committed PMC / in-step profile files a bench line may cite and the model they were measured on: the default
  configuration gets them, any other model / batch / switch gets None (`sorted(glob)` picked a superseded file and Tiny's MFMA
  numbers appeared on the DeiT-Small / Base / T2T lines).
* The uniform 8-bit code of GELU'(a) (include/uvc_kernels.h: UVC_Q8_LO, UVC_Q8_STEP) is as accurate as a bf16 GELU'(a) where it matters: |error| <=
  STEP / 2 everywhere, below bf16's spacing on [0.5, 2), and the relative L2 error it puts into dA = (g W2) o GELU'(a) stays within 1.7 x of what the
  bf16 tensor puts there (tools/gelu_grad_codes.py is the full study; this is its assertion)."""
import json
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**over):
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    for k, v in over.items():
        setattr(a, k, v)
    return bench, a


def test_manifest_names_existing_files_of_the_headline_configuration():
    man = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
    for kind in ("pmc_traffic", "pmc_mfma", "steps_only"):
        ent = man[kind]
        assert os.path.exists(os.path.join(ROOT, "profiles", ent["file"])), ent
        assert ent["model"] == "deit_tiny_patch16_224" and ent["batch"] == 512
    assert len({man[k]["tag"] for k in man}) == 1, "the three files of a line come from one profile run"


def test_profile_numbers_are_attached_to_the_model_they_were_measured_on_only():
    bench, a = _args()
    man = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
    tr = bench.pmc_traffic("attn_bwd", a)
    assert tr is not None and tr["source"] == man["pmc_traffic"]["file"] and tr["bytes"] > 3e8
    mf, src = bench.pmc_mfma(a)
    assert mf and src.startswith(man["pmc_mfma"]["file"])
    top = bench.in_step_top(a)
    assert top is not None and top["source"].endswith(man["steps_only"]["file"])
    for over in (dict(model_type="deit_base_patch16_224", batch=128), dict(model_type="deit_small_patch16_224", batch=256), dict(batch=256),
                 dict(enable_deit=1), dict(enable_patch_gating=2)):
        _, b = _args(**over)
        assert bench.pmc_traffic("attn_bwd", b) is None and bench.pmc_mfma(b) == ({}, None) and bench.in_step_top(b) is None, over


def test_one_byte_gelu_grad_code_is_as_accurate_as_bf16_where_it_matters():
    from uvc_amd._lib import Q8_LO, Q8_STEP
    hdr = open(os.path.join(ROOT, "include", "uvc_kernels.h")).read()
    assert "#define UVC_Q8_LO (-0.13f)" in hdr and "#define UVC_Q8_STEP (1.26f / 255.0f)" in hdr and abs(Q8_LO + 0.13) < 1e-12 and abs(Q8_STEP - 1.26 / 255) < 1e-12
    torch.manual_seed(0)
    x = torch.linspace(-8.0, 8.0, 400001, dtype=torch.float64)
    gp = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    assert Q8_LO < float(gp.min()) and float(gp.max()) < Q8_LO + 255 * Q8_STEP            # the code's range covers GELU' ([-0.1290, 1.1290])
    code = torch.clamp(torch.floor((gp - Q8_LO) / Q8_STEP + 0.5), 0, 255)
    dec = code * Q8_STEP + Q8_LO
    assert float((dec - gp).abs().max()) <= Q8_STEP / 2 + 1e-12
    b16 = gp.float().bfloat16().double()
    big = gp >= 0.5
    assert float((dec - gp).abs()[big].max()) < float((b16 - gp).abs()[big].max())        # finer than bf16 where GELU' is large
    for sigma in (0.3, 1.0, 2.0):                                                            # fresh-init, trained, wide pre-activations
        a = torch.randn(2048, 768, dtype=torch.float64) * sigma
        g = torch.randn(2048, 768, dtype=torch.float64)
        gpa = 0.5 * (1 + torch.erf(a / math.sqrt(2))) + a * torch.exp(-0.5 * a * a) / math.sqrt(2 * math.pi)
        ref = g * gpa
        e8 = float((g * (torch.clamp(torch.floor((gpa - Q8_LO) / Q8_STEP + 0.5), 0, 255) * Q8_STEP + Q8_LO) - ref).norm() / ref.norm())
        e16 = float((g * gpa.float().bfloat16().double() - ref).norm() / ref.norm())
        assert e8 <= 3e-3 and e8 <= 1.7 * e16, (sigma, e8, e16)
