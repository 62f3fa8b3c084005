"""CPU: the Stage-2 oracle (oracle/stage2.py) against the fixtures captured from the reference's own model and
loss (tests/golden/make_stage2_golden.py), plus the known answers of the two timm restatements."""
import math

import numpy as np
import pytest
import torch

import scenarios as SC
from helpers import build_oracle_stage2, load_golden
from oracle import stage2 as O2
from test_oracle_golden import _close

# one AdamW step moves an element by up to lr (1e-3 / 5e-4 here) and the direction m/(sqrt(v)+eps) of elements whose
# gradient is ~eps is ill-conditioned: weight rows are compared to 0.2 % of a step
ROW_ATOL = 2e-6


@pytest.mark.parametrize("name", list(SC.STAGE2))
def test_stage2_oracle_matches_reference_golden(name):
    torch.set_num_threads(4)
    gold = load_golden(name)
    r, S = build_oracle_stage2(name)
    x_all, y_all = SC.make_inputs(r)
    names = [str(n) for n in gold["param_names"]]
    assert sorted(n for n in names if S.wd_of[n] == 0.0) == sorted(str(n) for n in gold["no_decay"])
    L = S.cfg.depth
    for step in range(r["steps"]):
        S.begin_epoch(r["epoch_of_step"][step])
        out = {}
        O2.stage2_step(S, torch.from_numpy(x_all[step]), torch.from_numpy(y_all[step]), out)
        pre = f"step{step}."
        _close(S.cur_lr, gold[pre + "lr"], rtol=1e-12, atol=0, what=pre + "lr")
        _close(out["loss"].item(), gold[pre + "loss"], what=pre + "loss")
        _close(out["logits"].numpy(), gold[pre + "logits"], atol=1e-5, what=pre + "logits")
        _close(out["logits_dist"].numpy(), gold[pre + "logits_dist"], atol=1e-5, what=pre + "logits_dist")
        _close(float(out["grad_norm"]), gold[pre + "grad_norm"], what=pre + "grad_norm")
        gabs = np.array([float(out["grads"][n].double().abs().sum()) if out["grads"].get(n) is not None else np.nan for n in names])
        ref = gold[pre + "grad_abs_sum"]
        assert np.array_equal(np.isnan(gabs), np.isnan(ref)), "set of parameters without gradient differs"
        ok = ~np.isnan(ref)
        _close(gabs[ok], ref[ok], rtol=2e-3, atol=1e-7, what=pre + "grad_abs_sum")
        # parameters of hard-skipped blocks get no gradient and are not touched by AdamW (no decay either)
        for b in r["skip_blocks"]:
            assert all(np.isnan(ref[i]) for i, n in enumerate(names) if n.startswith(f"blocks.{b}."))
        assert gold[pre + "blocks_run"].tolist() == [int(b not in r["skip_blocks"]) for b in range(L)]
        psum = np.array([float(S.params[n].double().abs().sum()) for n in names])
        _close(psum, gold[pre + "param_abs_sum"], rtol=1e-5, what=pre + "param_abs_sum")
        _close(S.params["blocks.0.attn.proj.weight"][0].numpy(), gold[pre + "proj_0_row0"], rtol=1e-4, atol=ROW_ATOL, what=pre + "proj row")
        _close(S.params[f"blocks.{L - 1}.mlp.fc1.weight"][:, 0].numpy(), gold[pre + "fc1_last_col0"], rtol=1e-4, atol=ROW_ATOL,
               what=pre + "fc1 col")
        _close(S.params["pos_embed"][0, 0].numpy(), gold[pre + "pos_embed_tok0"], rtol=1e-4, atol=ROW_ATOL, what=pre + "pos_embed")


def test_stage2_checkpoint_keys_match_stage1():
    """post_train.py:683 loads the Stage-1 checkpoint strictly: same keys, same order (patch gating off)."""
    s1 = [str(k) for k in load_golden("micro_pruned")["state_dict_keys"]]
    s2 = [str(k) for k in load_golden("stage2_micro")["state_dict_keys"]]
    assert s1 == s2


def test_timm_cosine_schedule_known_values():
    """timm CosineLRScheduler (0.3.2) for create_scheduler's arguments: linear warm-up from warmup_lr over
    warmup_epochs, then lr_min + (base - lr_min)/2 * (1 + cos(pi t / T)) with t the absolute epoch."""
    f = lambda t: O2.cosine_epoch_lr(t, 5e-5, 120, 1e-5, 5, 1e-6, 0.1)
    assert f(0) == 1e-6
    assert abs(f(3) - (1e-6 + 3 * (5e-5 - 1e-6) / 5)) < 1e-18
    assert abs(f(5) - (1e-5 + 0.5 * 4e-5 * (1 + math.cos(math.pi * 5 / 120)))) < 1e-18
    assert abs(f(60) - 3e-5) < 1e-12
    assert f(120) == 1e-5 and f(500) == 1e-5            # past the single cycle: lr_min
    # run_post_train.sh: lr 1e-4, batch 128, 2 GPUs -> 5e-5 (post_train.py:297)
    assert abs(O2.Stage2Hyper(learning_rate=1e-4, train_batch_size=128, world_size=2).lr - 5e-5) < 1e-18


def test_timm_weight_decay_groups():
    r, S = build_oracle_stage2("stage2_micro_deit")
    nd = {n for n, w in S.wd_of.items() if w == 0.0}
    assert {"pos_embed", "cls_token", "dist_token", "norm.weight", "norm.bias", "head.bias", "blocks.0.attn.qkv.bias",
            "blocks.1.norm2.weight", "patch_embed.proj.bias", "gumbel.bias", "blocks.0.attn_skip_gating"} <= nd
    assert not ({"head.weight", "patch_embed.proj.weight", "blocks.0.mlp.fc1.weight", "block_skip_gating", "gumbel.weight"} & nd)
