"""GPU: the full Stage-1 step on the HIP path (exact-float32 MFMA mode) against the golden vectors
captured from the reference's own modules: loss / logits / gradients / s r y p z / gate logits /
FLOPs ratio within 1e-3 relative (BASELINE.json north_star), mask index sets bit-exact.  The bf16
throughput mode is checked against the same goldens at the looser tolerance stated below."""
import numpy as np
import pytest
import torch

import scenarios as SC
from helpers import load_golden, split_draws
from stage1_driver import Stage1Run

pytestmark = pytest.mark.gpu

CORE = ["micro_warmup", "micro_train", "micro_pruned", "micro_clip", "micro_bounds", "micro_softl0", "micro_deit",
        "micro_patch1", "micro_patch2", "tiny8_train", "tiny8_pruned", "small2_pruned", "base2_deit"]


def close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b); tol = atol + rtol * np.abs(b)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} vs tol {tol.flat[err.argmax()]:.3e} (ref {b.flat[err.argmax()]:.4e})"


def run_scenario(name, precision, rtol, check_masks=True):
    from uvc_amd.joint_train import count_mask
    from uvc_amd.uvc_utils import prune_w_mask
    gold = load_golden(name)
    run = Stage1Run(name, precision=precision)
    r, cfg = run.r, run.cfg
    x_all, y_all = SC.make_inputs(r)
    names = [str(n) for n in gold["param_names"]]
    pmap = dict(run.model.named_parameters())
    assert list(pmap.keys()) == names, "named_parameters order differs from the reference"
    close(float(count_mask(run.model)), gold["mask0_count"], 1e-6, 0, "count_mask at start")
    for step in range(r["steps"]):
        md, e1, e2 = split_draws(r, gold, step, cfg.depth)
        run.inject_draws(md, e1, e2)
        out = run.step(torch.from_numpy(x_all[step]).cuda(), torch.from_numpy(y_all[step]).cuda())
        pre = f"step{step}."
        close(float(out["loss"]), gold[pre + "loss"], rtol, 1e-6, pre + "loss")
        close(out["outputs"][0].detach().cpu().numpy(), gold[pre + "logits"], rtol, 3e-4 if precision == "fp32" else 3e-2, pre + "logits")
        close(out["outputs"][1].detach().cpu().numpy(), gold[pre + "logits_dist"], rtol, 3e-4 if precision == "fp32" else 3e-2, pre + "logits_dist")
        gn = float(out["gnorm"])
        close(gn, gold[pre + "grad_norm"], rtol if precision == "fp32" else 3e-2, 0, pre + "grad_norm")
        close(float(out["cur"]), gold[pre + "cur_resource"], 1e-4, 0, pre + "cur_resource")
        close(run.optimizer.param_groups[0]["lr"], gold[pre + "lr"], 1e-9, 0, pre + "lr")
        st = 1e-3 if precision == "fp32" else 2e-2
        close(out["s"].numpy(), gold[pre + "s"], st, 1e-6, pre + "s")
        close(out["r"].numpy(), gold[pre + "r"], st, 1e-6, pre + "r")
        close(run.minimax.y.data.cpu().numpy(), gold[pre + "y"], st, 1e-7, pre + "y")
        close(run.minimax.p.data.cpu().numpy(), gold[pre + "p"], st, 1e-7, pre + "p")
        close(float(run.minimax.z), gold[pre + "z"], 1e-4, 0, pre + "z")
        close(out["g"].numpy(), gold[pre + "gating"], st, 1e-6, pre + "gating")
        # gradients: the reference's were recorded after clip_grad_norm_ (in place)
        coef = min(1.0, r["max_grad_norm"] / (gn + 1e-6))
        ref = gold[pre + "grad_abs_sum"]
        got = []
        for n in names:
            p = pmap[n]
            if p.grad is None:
                got.append(np.nan)
            else:
                c = 1.0 if n == "block_skip_gating" else coef     # the gate's grad is already clipped in place
                got.append(float(p.grad.double().abs().sum()) * c)
        got = np.array(got)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), [n for n, a, b in zip(names, got, ref) if np.isnan(a) != np.isnan(b)]
        ok = ~np.isnan(ref)
        # bf16 mode: the gate-logit gradient is d0/tau * (<gA, x_{l+1}> - <gA, x_l>), a difference of O(1) dot products of a
        # bf16 gradient stream that leaves ~1e-3: its absolute floor is the rounding of the stream (9 % against 6 % depending on
        # which kernels produced the last block's stream), not a relative quantity
        gate_atol = np.where(np.array(names)[ok] == "block_skip_gating", 0.0 if precision == "fp32" else 3e-4, 0.0)
        close(got[ok], ref[ok], 3e-3 if precision == "fp32" else 4e-2, 1e-6 + gate_atol, pre + "grad_abs_sum")
        psum = np.array([float(pmap[n].data.double().abs().sum()) for n in names])
        # AdamW's first steps move every element by ~lr * g/(|g|+eps): elements with |g| ~ eps amplify
        # 1e-7-level gradient differences, hence 1e-4 (fp32) / 1e-3 (bf16) on the per-tensor checksums
        close(psum, gold[pre + "param_abs_sum"], 1e-4 if precision == "fp32" else 1e-3, 0, pre + "param_abs_sum")
        wt = (1e-4, 1e-7) if precision == "fp32" else (1e-3, 2.0 * r["learning_rate"])   # bf16: an AdamW step is ~lr per element
        close(run.layers["W1"][0].weight.data[0].cpu().numpy(), gold[pre + "w1_0_row0"], *wt, pre + "w1 row")
        close(run.layers["W3"][0].weight.data[0].cpu().numpy(), gold[pre + "w3_0_row0"], *wt, pre + "w3 row")
        run.optimizer.zero_grad()
    prune_w_mask(run.minimax, run.optimizer)
    if check_masks:
        for l in range(cfg.depth):
            kp = np.packbits(run.layers["W1"][l].mask[0].cpu().numpy().astype(np.uint8))
            kf = np.packbits(run.layers["W3"][l].mask[0].cpu().numpy().astype(np.uint8))
            assert np.array_equal(kp, gold[f"keep_proj.{l}"]), f"proj mask layer {l} differs from the reference"
            assert np.array_equal(kf, gold[f"keep_fc2.{l}"]), f"fc2 mask layer {l} differs from the reference"
            assert torch.equal(run.layers["W2"][l].mask[:, 0], run.layers["W3"][l].mask[0])
        close(float(count_mask(run.model)), gold["mask_count"], 1e-6, 0, "count_mask")
    if r["use_gumbel"] and r["enable_block_gating"]:
        q = [torch.from_numpy(gold["final.draw0"]).cuda(), torch.from_numpy(gold["final.draw1"]).cuda()]
        run.minimax.exp_source = lambda shape: q.pop(0)
    close(float(run.minimax.run_resource_fn(gumbel_hard=True)), gold["real_flops"], 1e-4, 0, "real flops")
    close(float(run.minimax.run_resource_fn(gumbel_hard=False)), gold["expect_flops"], 1e-4, 0, "expected flops")
    return run


@pytest.mark.parametrize("name", CORE)
def test_stage1_fp32_matches_reference_golden(name):
    run_scenario(name, "fp32", 1e-3)


@pytest.mark.parametrize("name", ["micro_train", "micro_pruned", "micro_deit", "micro_patch2", "tiny8_train", "tiny8_pruned", "small2_pruned",
                                  "base2_deit"])
def test_stage1_bf16_matches_reference_golden(name):
    """bf16 operands (8 mantissa bits) with float32 accumulation and float32 master weights: loss and
    logits within 2e-2; the UVC state and the mask index sets do not depend on the activations' precision
    beyond the gate gradient, so masks stay bit-exact on these fixtures."""
    run_scenario(name, "bf16", 2e-2)


def test_checkpoint_layout_roundtrip(tmp_path):
    """save_model writes the bare state_dict incl. masks (joint_train.py:107-119); a fresh model loads it strictly."""
    from argparse import Namespace
    from uvc_amd.joint_train import register_masks, save_model
    from uvc_amd.model_distilled import DistilledVisionTransformer
    gold = load_golden("tiny8_train")
    run = Stage1Run("micro_train", precision="fp32")
    keys = list(run.model.state_dict().keys())
    args = Namespace(output_dir=str(tmp_path), name="t", model_type="deit_micro", local_rank=0)
    path = save_model(args, run.model, run.minimax, 1)
    sd = torch.load(path, map_location="cpu")
    assert list(sd.keys()) == keys
    m = run.r["model_cfg"]
    fresh = DistilledVisionTransformer(enable_dist=0, img_size=m["img_size"], patch_size=16, num_classes=m["num_classes"],
                                       embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"], precision="fp32")
    register_masks(fresh)
    fresh.load_state_dict(sd, strict=True)
    tiny = DistilledVisionTransformer(enable_dist=0, embed_dim=192, depth=12, num_heads=3, precision="fp32")
    register_masks(tiny)
    assert list(tiny.state_dict().keys()) == [str(k) for k in gold["state_dict_keys"]]


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 3e-2)])
def test_eval_forward_with_hard_block_skip_matches_oracle(precision, tol):
    """valid()/teacher path (joint_train.py:199-246): eval mode, enable_block_gating=0 -> a block runs only if its
    gate logit g1 > g0 (model_distilled.py:496-500); output = (head + head_dist)/2."""
    from helpers import initial_params
    from oracle import vit as OV
    from uvc_amd.model_distilled import DistilledVisionTransformer
    r = SC.recipe("micro_deit")
    cfg, params, _ = initial_params(r)
    params["block_skip_gating"] = torch.tensor([[1.0, -1.0], [-1.0, 1.0]])      # block 0 skipped, block 1 runs
    m = r["model_cfg"]
    model = DistilledVisionTransformer(enable_dist=1, img_size=m["img_size"], patch_size=16, num_classes=m["num_classes"],
                                       embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"], precision=precision)
    model.load_state_dict(params, strict=False)
    model.eval()
    x, _ = SC.make_inputs(r)
    x0 = torch.from_numpy(x[0])
    with torch.no_grad():
        out, macs = model(x0.cuda())
    ref, rmacs = OV.forward(params, cfg, OV.GateFlags(enable_block_gating=0, training=False), x0)
    assert macs[0] == rmacs[0] and macs[1] == rmacs[1]
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=tol, atol=tol)


def test_bucketed_backward_is_identical_to_the_single_call_backward():
    """The DDP path cuts the backward at gradient-bucket boundaries (uvc_vit_io.stage_begin/end).  With the
    collectives stubbed out (one process) the staged backward must reproduce the one-call backward bit for bit,
    and wrapping must not disturb state_dict()."""
    from uvc_amd.ddp import DistributedDataParallel
    grads = []
    for staged in (False, True):
        run = Stage1Run("micro_pruned", precision="fp32")
        gold = load_golden("micro_pruned")
        if staged:
            keys = list(run.model.state_dict().keys())
            ddp = DistributedDataParallel(run.model, num_buckets=2, dual_scalar=run.minimax.z)
            ddp.world = 2                      # force the staged path; the reducer itself stays a no-op (world 1)
            assert ddp.stage_ends == [2, 3, run.cfg.depth + 3]         # block 1 | block 0 (before the embedding backward) | embedding + small tensors
            assert list(run.model.state_dict().keys()) == keys
        r = run.r
        x_all, y_all = SC.make_inputs(r)
        md, e1, e2 = split_draws(r, gold, 0, run.cfg.depth)
        run.inject_draws(md, e1, e2)
        out = run.step(torch.from_numpy(x_all[0]).cuda(), torch.from_numpy(y_all[0]).cuda())
        grads.append((run.model._flat_grad[:run.model._off.n_total].clone(), float(out["loss"]), float(run.minimax.z)))
    assert grads[0][1] == grads[1][1] and grads[0][2] == grads[1][2]
    assert torch.equal(grads[0][0], grads[1][0])


def test_training_state_resume_is_bit_identical(tmp_path):
    """the engine's own training state (model + masks, s r y p z, gate momentum / accumulators, eps, AdamW
    moments and step counts, schedule position) restores a fresh trainer so that the next step is bit-identical to the
    uninterrupted run."""
    name = "micro_pruned"
    gold = load_golden(name)

    def steps(run, lo, hi):
        r = run.r
        x_all, y_all = SC.make_inputs(r)
        out = None
        for st in range(lo, hi):
            md, e1, e2 = split_draws(r, gold, st, run.cfg.depth)
            run.inject_draws(md, e1, e2)
            out = run.step(torch.from_numpy(x_all[st]).cuda(), torch.from_numpy(y_all[st]).cuda())
            run.optimizer.zero_grad()
        return out

    a = Stage1Run(name, precision="fp32")
    out_a = steps(a, 0, 4)
    b = Stage1Run(name, precision="fp32")
    steps(b, 0, 3)
    path = str(tmp_path / "state.pth.tar")
    torch.save(b.trainer.state_dict(), path)
    c = Stage1Run(name, precision="fp32")                 # fresh process-equivalent: new model, optimiser, minimax
    c.trainer.load_state_dict(torch.load(path, map_location="cuda"))
    out_c = steps(c, 3, 4)
    assert float(out_a["loss"]) == float(out_c["loss"]) and float(out_a["cur"]) == float(out_c["cur"])
    assert torch.equal(a.model._flat, c.model._flat)
    for k in ("s", "r", "y", "p", "z"):
        assert torch.equal(getattr(a.minimax, k).data, getattr(c.minimax, k).data), k
    assert torch.equal(a.optimizer.exp_avg, c.optimizer.exp_avg) and a.optimizer.steps == c.optimizer.steps
    assert a.trainer.global_step == c.trainer.global_step == 4
    assert a.optimizer.param_groups[0]["lr"] == c.optimizer.param_groups[0]["lr"]
    with pytest.raises(ValueError):
        c.trainer.load_state_dict(a.model.state_dict())   # the reference-format checkpoint is not a training state
