"""GPU: one FULL UVC-train step of DeiT-Small and of DeiT-Base (+ distillation token) against the oracle, at sizes that select the
kernels their PRODUCTION batches run on.

The reference goldens of BASELINE configs 3 / 4 (`small2_pruned`, `base2_deit`) run at batch 2 (M = 394 / 396 token rows), where every
GEMM is the generic 128 x 128 kernel and every LayerNorm a pass of its own.  The bench shapes run elsewhere:

  * DeiT-Small (D = 384), M >= 4096: qkv / proj / fc1 / dfc2 on the weights-stationary `k_gemm_ws<.., 12, 6, 2>` (ws384_ok), the
    dgrads of fc1 / qkv fused with the LayerNorm backward on the 128 x 384 row tile (`k_gemm_row384_lnbwd`) -- selected by the row
    count alone, so the test simply runs at batch 24 (M = 4728);
  * DeiT-Base (D = 768): `k_gemm_nt256` (256 x 256 tiles, LDS-DMA) for the shapes with >= 640 tiles, i.e. batch >= 94.  The oracle on
    the host would need minutes there, so the test runs batch 12 with `uvc_vit_io.force_generic = 3`: the wide-tile kernels wherever
    the shape admits them, whatever the row / tile count (the weight gradients take their 256 x 256 tiles by shape at any batch).

Same weights / primal-dual state / Exp(1) draws as the fixtures (those do not depend on the batch), inputs from the same frozen numpy
stream, expected values from oracle.step.stage1_step on the host (pinned to the reference on exactly these scenarios at batch 2 by
tests/test_oracle_golden.py).  Checked as in tests/test_streaming_batch_gpu.py: loss, logits, clip norm, EVERY parameter's gradient
as a relative L2 error (bound 2.5 %), s r y p z, gate logits, resource, mask index sets bit-exact -- plus the same step on the generic
kernels as an A/B.  tools/perturb_demo.sh shows the test failing when `k_gemm_nt256` or `k_gemm_row384_lnbwd` is scaled by 1.02."""
import copy
import os

import numpy as np
import pytest
import torch

import scenarios as SC
from helpers import build_oracle_from_recipe, load_golden, split_draws
from oracle import step as OS
from stage1_driver import Stage1Run
from test_streaming_batch_gpu import TOL_GRAD_BF16, _check_against_oracle, _masks, _oracle_keep, _rel

pytestmark = pytest.mark.gpu

# scenario -> (batch, force_generic of the "production kernels" run)
CASES = {"small2_pruned": (24, 0), "base2_deit": (12, 3)}


def _recipe(name):
    r = copy.deepcopy(SC.recipe(name))
    r["batch"], r["steps"] = CASES[name][0], 1
    return r


_ORACLE = {}


def _oracle_step(name):
    if name not in _ORACLE:
        r, S = build_oracle_from_recipe(_recipe(name))
        gold = load_golden(name)
        x_all, y_all = SC.make_inputs(r)
        md, e1, e2 = split_draws(r, gold, 0, S.cfg.depth)
        torch.set_num_threads(min(64, os.cpu_count() or 1))
        out = {}
        OS.stage1_step(S, torch.from_numpy(x_all[0]), torch.from_numpy(y_all[0]), md, e1, e2, out)
        _ORACLE[name] = (S, out)
    return _ORACLE[name]


def _hip_step(name, precision, force_generic=0, fuse_next_ln=True):
    r = _recipe(name)
    gold = load_golden(name)
    run = Stage1Run(r, precision=precision)
    for m in (run.model, run.teacher):
        m.force_generic = force_generic
        m.fuse_next_ln = fuse_next_ln
    x_all, y_all = SC.make_inputs(r)
    md, e1, e2 = split_draws(r, gold, 0, run.cfg.depth)
    run.inject_draws(md, e1, e2)
    out = run.step(torch.from_numpy(x_all[0]).cuda(), torch.from_numpy(y_all[0]).cuda())
    torch.cuda.synchronize()
    grads = {n: (None if p.grad is None else p.grad.detach().float().cpu().clone()) for n, p in run.model.named_parameters()}
    return run, out, grads


def _check(name, run, out, grads, tol_grad, tol_out, tol_state, what):
    S, o = _oracle_step(name)
    worst = _check_against_oracle(run, out, grads, S, o, tol_grad=tol_grad, tol_out=tol_out, tol_state=tol_state, what=what)
    if run.cfg.enable_dist:
        lg, ref = out["outputs"][1].detach().float().cpu(), o["logits_dist"]
        assert _rel(lg, ref) <= tol_out, (what, "logits_dist", _rel(lg, ref))
    for l, ((k1, k3), (r1, r3)) in enumerate(zip(_masks(run), _oracle_keep(S))):
        assert torch.equal(k1.bool(), r1.bool()) and torch.equal(k3.bool(), r3.bool()), f"{what}: mask index set of layer {l} differs from the oracle"
    print("per-tensor gradient error vs oracle, %s: max %.4f (%s), median %.4f; loss %.6f vs %.6f" %
          (what, max(worst.values()), max(worst, key=worst.get), float(np.median(list(worst.values()))), float(out["loss"]), float(o["loss"])))
    return worst


@pytest.mark.parametrize("name", list(CASES))
def test_wide_model_step_matches_oracle_on_production_kernels(name):
    """bf16 throughput mode on the kernels the production batch selects (see the module text)."""
    run, out, grads = _hip_step(name, "bf16", force_generic=CASES[name][1])
    _check(name, run, out, grads, TOL_GRAD_BF16, 2e-2, 2e-2, f"{name} bf16 production kernels")


@pytest.mark.parametrize("name", list(CASES))
def test_wide_model_production_and_generic_kernels_agree_per_tensor(name):
    """A/B inside the bf16 mode: the production kernels against the generic tiled kernel + stand-alone LayerNorm passes + the unfused
    dgrad / LayerNorm-backward pair; both against the oracle, and against each other per tensor."""
    run_p, out_p, g_p = _hip_step(name, "bf16", force_generic=CASES[name][1])
    run_g, out_g, g_g = _hip_step(name, "bf16", force_generic=1, fuse_next_ln=False)
    _check(name, run_g, out_g, g_g, TOL_GRAD_BF16, 2e-2, 2e-2, f"{name} bf16 generic kernels")
    diff = {n: _rel(g_p[n], g_g[n]) for n in g_p if g_p[n] is not None}
    bad = {k: round(v, 5) for k, v in diff.items() if v > (2e-2 if k != "block_skip_gating" else 8e-2)}
    assert not bad, (name, "production vs generic kernels", bad)
    assert abs(float(out_p["loss"]) - float(out_g["loss"])) <= 2e-3 * abs(float(out_g["loss"]))
    for (a1, a3), (b1, b3) in zip(_masks(run_p), _masks(run_g)):
        assert torch.equal(a1, b1) and torch.equal(a3, b3)
    print("%s production vs generic, per-tensor gradient difference: max %.5f (%s)" % (name, max(diff.values()), max(diff, key=diff.get)))


@pytest.mark.parametrize("name", list(CASES))
def test_wide_model_step_fp32_matches_oracle(name):
    """The float32-exact mode at the same batch: the 1e-3 bar of north_star, masks bit-exact."""
    run, out, grads = _hip_step(name, "fp32")
    _check(name, run, out, grads, 3e-3, 1e-3, 1e-3, f"{name} fp32")
