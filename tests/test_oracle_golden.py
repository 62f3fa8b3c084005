"""CPU: the oracle (oracle/) against the golden vectors captured from the reference's own
modules (tests/golden/make_golden.py) and against the reference's known answers
(SURVEY.md §4).  This is what pins the oracle; the GPU tests then compare HIP against it."""
import numpy as np
import pytest
import torch

import scenarios as SC
from helpers import build_oracle, load_golden, split_draws
from oracle import step as OS
from oracle import uvc as OU
from oracle import vit as OV

RTOL = 1e-3   # BASELINE.json north_star: 1e-3 rel-tol fp32; masks bit-exact


def _close(a, b, rtol=RTOL, atol=1e-6, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} (tol {tol.flat[err.argmax()]:.3e}) at {np.unravel_index(err.argmax(), err.shape)}"


@pytest.mark.parametrize("name", list(SC.SCENARIOS))
def test_oracle_matches_reference_golden(name):
    torch.set_num_threads(4)
    gold = load_golden(name)
    r, S = build_oracle(name)
    cfg = S.cfg
    x_all, y_all = SC.make_inputs(r)
    names = [str(n) for n in gold["param_names"]]
    # start-of-epoch masks at the initial state (joint_train.py:377)
    m0 = OU.prune_masks(S.st, S.w1(), S.w3())
    for step in range(r["steps"]):
        model_draws, e1, e2 = split_draws(r, gold, step, cfg.depth)
        out = {}
        OS.stage1_step(S, torch.from_numpy(x_all[step]), torch.from_numpy(y_all[step]), model_draws, e1, e2, out)
        pre = f"step{step}."
        _close(out["loss"].item(), gold[pre + "loss"], what=pre + "loss")
        _close(out["logits"].numpy(), gold[pre + "logits"], atol=1e-5, what=pre + "logits")
        _close(out["logits_dist"].numpy(), gold[pre + "logits_dist"], atol=1e-5, what=pre + "logits_dist")
        _close(float(out["grad_norm"]), gold[pre + "grad_norm"], what=pre + "grad_norm")
        _close(out["cur_resource"], gold[pre + "cur_resource"], rtol=1e-5, what=pre + "cur_resource")
        _close(S.lr, gold[pre + "lr"], rtol=1e-9, atol=0, what=pre + "lr")
        for k, v in (("s", S.st.s), ("r", S.st.r), ("y", S.st.y), ("p", S.st.p)):
            _close(v.numpy(), gold[pre + k], rtol=1e-4, atol=1e-7, what=pre + k)
        _close(S.st.z.item(), gold[pre + "z"], rtol=1e-5, what=pre + "z")
        _close(S.params["block_skip_gating"].numpy(), gold[pre + "gating"], rtol=1e-4, what=pre + "gating")
        gabs = np.array([float(out["grads"][n].double().abs().sum()) if out["grads"].get(n) is not None else np.nan
                         for n in names])
        ref = gold[pre + "grad_abs_sum"]
        assert np.array_equal(np.isnan(gabs), np.isnan(ref)), "set of parameters without gradient differs"
        ok = ~np.isnan(ref)
        _close(gabs[ok], ref[ok], rtol=2e-3, atol=1e-7, what=pre + "grad_abs_sum")
        psum = np.array([float(S.params[n].double().abs().sum()) for n in names])
        _close(psum, gold[pre + "param_abs_sum"], rtol=1e-5, what=pre + "param_abs_sum")
        _close(S.params[f"blocks.0.attn.proj.weight"][0].numpy(), gold[pre + "w1_0_row0"], rtol=1e-4, atol=1e-7,
               what=pre + "w1 row")
        _close(S.params[f"blocks.0.mlp.fc2.weight"][0].numpy(), gold[pre + "w3_0_row0"], rtol=1e-4, atol=1e-7,
               what=pre + "w3 row")
    # end-of-epoch masks: index sets bit-exact (joint_train.py:500)
    masks = OU.prune_masks(S.st, S.w1(), S.w3())
    total_other = 0.0
    for l, (mp, mf2, mf1, keep1, keep3) in enumerate(masks):
        assert np.array_equal(np.packbits(keep1.numpy().astype(np.uint8)), gold[f"keep_proj.{l}"]), f"proj mask layer {l}"
        assert np.array_equal(np.packbits(keep3.numpy().astype(np.uint8)), gold[f"keep_fc2.{l}"]), f"fc2 mask layer {l}"
    # count_mask (joint_train.py:182-188): every weighted module has a mask; only W1/W2/W3 masks change
    full = float(gold["total_param"])
    removed = sum(float((1 - mp).sum() + (1 - mf2).sum() + (1 - mf1).sum()) for mp, mf2, mf1, _, _ in masks) / 1e6
    _close(full - removed, gold["mask_count"], rtol=1e-6, what="count_mask")
    # final FLOPs ratio ("Real FLOPs", joint_train.py:509)
    if r["use_gumbel"] and r["enable_block_gating"]:
        eh, es = torch.from_numpy(gold["final.draw0"]), torch.from_numpy(gold["final.draw1"])
    else:
        eh = es = None
    s2 = [OU.scores_w1(W, S.st.H, S.st.hd)[1] for W in S.w1()]
    g = S.params["block_skip_gating"] if r["enable_block_gating"] else None
    _close(float(OU.resource(S.st, s2, g, eh, S.hp, hard=True)), gold["real_flops"], rtol=1e-5, what="real flops")
    _close(float(OU.resource(S.st, s2, g, es, S.hp, hard=False)), gold["expect_flops"], rtol=1e-5, what="expected flops")


def test_known_answers_deit_tiny():
    """2506.98 M FLOPs and the per-block MAC list (log/deit-tiny-log.log:7,9), 5.6529 M
    mask sum (:2,31) -- all also re-derived from the reference itself into tiny8_train.npz."""
    cfg = OV.VitConfig()
    embed, macs = OV.mac_table(cfg, 1)
    assert embed == 28901376
    assert macs[0] == [21786624, 7451328, 7451328, 7262208, 29048832, 29048832]
    st = OU.UvcState.create(12, 3, 64, 768, embed, macs)
    assert st.resource_ub == 2506982400.0
    gold = load_golden("tiny8_train")
    assert int(gold["embed_macs"]) == embed and gold["macs_list"].tolist() == macs
    assert float(gold["resource_ub"]) == st.resource_ub
    shapes = OV.param_shapes(cfg)
    # every module with a .weight gets a mask of the weight's shape (joint_train.py:169-171)
    total = sum(int(np.prod(s)) for n, s in shapes.items() if n.endswith(".weight")) / 1e6
    assert abs(total - 5.652864) < 1e-9 and abs(float(gold["total_param"]) - total) < 1e-6


@pytest.mark.parametrize("name,nkeys", [("tiny8_train", 255), ("micro_deit", None), ("micro_patch1", None)])
def test_state_dict_layout_matches_reference(name, nkeys):
    """Checkpoint layout = bare state_dict incl. mask buffers: key order and shapes."""
    import json
    from helpers import vit_config
    gold = load_golden(name)
    r = SC.recipe(name)
    keys = [str(k) for k in gold["state_dict_keys"]]
    shapes = OV.param_shapes(vit_config(r), r["enable_patch_gating"])
    pkeys = [k for k in keys if not k.endswith(".mask")]
    assert pkeys == list(shapes.keys())
    for k, sh in zip(keys, gold["state_dict_shapes"]):
        sh = tuple(json.loads(str(sh)))
        if k.endswith(".mask"):
            assert sh == tuple(shapes[k[:-5] + ".weight"])
        else:
            assert sh == tuple(shapes[k])
    if nkeys:
        assert len(keys) == nkeys


def test_zlr_schedule_and_eps_decay_known_answers():
    """joint_train.py:999-1005 with --num_epochs 30 --zlr_schedule_list 1,5,9,13,17 ->
    {0:1, 6:5, 12:9, 18:13, 24:17} (log/deit-tiny-log.log:12-13); eps 0.1 -> 0.092 -> 0.08464
    (log :142,166; uvc_utils.py:290-293)."""
    lst = "1,5,9,13,17".split(",")
    gap = 30 // len(lst)
    assert {i * gap: int(v) for i, v in enumerate(lst)} == {0: 1, 6: 5, 12: 9, 18: 13, 24: 17}
    eps = 0.1
    seq = []
    for _ in range(2):
        eps = eps * 0.92
        seq.append(eps)
    assert abs(seq[0] - 0.092) < 1e-12 and abs(seq[1] - 0.08464) < 1e-12


def test_fc1_mask_is_the_sticky_union_of_pruned_sets():
    """prune_w_mask resets the proj / fc2 masks on every call but only writes zeros into the fc1 mask (uvc_utils.py:382,393,401):
    fixture from the reference's own prune_w_mask over three non-monotone primal states (tests/golden/make_mask_golden.py)."""
    gold = load_golden("mask_sticky_micro")
    r, S = build_oracle(str(gold["scenario"]))
    prev = None
    for i in range(3):
        S.st.s, S.st.r = torch.from_numpy(gold[f"call{i}.s"].copy()), torch.from_numpy(gold[f"call{i}.r"].copy())
        masks = OU.prune_masks(S.st, S.w1(), S.w3(), prev_fc1=prev)
        prev = [m[2] for m in masks]
        for l, (mp, mf2, mf1, keep1, keep3) in enumerate(masks):
            assert np.array_equal(np.packbits(keep1.numpy().astype(np.uint8)), gold[f"call{i}.keep_proj.{l}"])
            assert np.array_equal(np.packbits(keep3.numpy().astype(np.uint8)), gold[f"call{i}.keep_fc2.{l}"])
            assert bool((mf1 == mf1[:, 0:1]).all())
            assert np.array_equal(np.packbits(mf1[:, 0].numpy().astype(np.uint8)), gold[f"call{i}.keep_fc1.{l}"]), (i, l)
    # the fixture exercises the difference: after call 1 fc1 rows stay masked whose fc2 column was released
    assert any(not np.array_equal(gold[f"call1.keep_fc1.{l}"], gold[f"call1.keep_fc2.{l}"]) for l in range(S.cfg.depth))
