"""GPU: the HIP primal-dual engine (through the C-ABI, via the uvc_amd mirrors of
uvc_utils.py / uvc_optimizer.py) against the oracle on the same seeded inputs; bit-exact for
index sets / masks / shrunk weights, 1e-4 relative for the float state."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import scenarios as SC
from oracle import uvc as OU
from oracle import vit as OV
from stub_model import StubModel

pytestmark = pytest.mark.gpu


def get_uvc_layers(model):
    from uvc_amd.joint_train import get_uvc_layers as g
    return g(model)


def make_args(r, H, hd):
    return Namespace(eps_decay=r["eps_decay"], enable_patch_gating=0, enable_part_gating=0,
                     enable_block_gating=r["enable_block_gating"], head_size=hd, num_heads=H, flops_with_mhsa=1,
                     use_gumbel=r["use_gumbel"], enable_jumping=0, eps=r["eps"], enable_warmup=r["warmup"], soptim="sgd",
                     roptim="sgd", slr=r["slr"], rlr=r["rlr"], glr=r["glr"], zlr_schedule_list=[int(r["zlr"])],
                     ylr=r["ylr"], plr=r["plr"], budget=r["budget"], sl2wd=r["sl2wd"], gating_weight=r["gating_weight"],
                     z_grad_clip=r["z_grad_clip"], gating_interval=r["gating_interval"])


def build_pair(L, H, hd, Fh, seed, r):
    """Identical weights/state on the CPU (oracle) and the GPU (product)."""
    from uvc_amd.uvc_optimizer import build_minimax_model
    D = H * hd
    rs = np.random.RandomState(seed)
    model = StubModel(L, D, Fh)
    W1c, W3c = [], []
    for blk in model.blocks:
        w1 = torch.from_numpy((rs.standard_normal((D, D)) * 0.05).astype(np.float32))
        w3 = torch.from_numpy((rs.standard_normal((D, Fh)) * 0.05).astype(np.float32))
        blk.attn.proj.weight.data.copy_(w1)
        blk.mlp.fc2.weight.data.copy_(w3)
        W1c.append(w1.clone()); W3c.append(w3.clone())
    model = model.cuda()
    model.eps = r["eps"]
    model.enable_warmup = r["warmup"]
    cfg = OV.VitConfig(embed_dim=D, depth=L, num_heads=H, mlp_ratio=Fh / D)
    embed, macs = OV.mac_table(cfg, 1)
    names, layers, ldict = get_uvc_layers(model)
    args = make_args(r, H, hd)
    mm, dual_opt, s_opt, r_opt, g_opt = build_minimax_model(model, names, layers, ldict, args, (embed, macs))
    st = OU.UvcState.create(L, H, hd, Fh, embed, macs, eps=r["eps"])
    s0, r0, y0, p0, z0 = SC.initial_state(dict(r, seed=seed), L, H, hd, Fh)
    st.s, st.r = torch.from_numpy(s0.copy()), torch.from_numpy(r0.copy())
    st.y, st.p, st.z = torch.from_numpy(y0.copy()), torch.from_numpy(p0.copy()), torch.tensor(float(z0))
    mm.s.data.copy_(st.s); mm.r.data.copy_(st.r); mm.y.data.copy_(st.y); mm.p.data.copy_(st.p); mm.z.data.fill_(float(z0))
    return model, mm, (dual_opt, s_opt, r_opt, g_opt), args, st, W1c, W3c


CASES = [
    ("micro", 2, 2, 64, 512), ("tiny", 12, 3, 64, 768), ("base", 12, 12, 64, 3072), ("t2t", 14, 6, 64, 1152),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("state,use_gumbel,clip", [("pruned", 1, 0.5), ("bounds", 1, 0.01), ("pruned", 0, 0.5), ("zero", 1, 0.5)])
def test_uvc_optimizer_matches_oracle(case, state, use_gumbel, clip):
    from uvc_amd.uvc_optimizer import uvc_optimizer
    from uvc_amd.uvc_utils import prune_w_mask
    _, L, H, hd, Fh = case
    r = dict(SC.DEFAULTS, warmup=0, state=state, use_gumbel=use_gumbel, z_grad_clip=clip, gating_interval=2, sl2wd=0.01)
    seed = 100 + L + H
    model, mm, (dual_opt, s_opt, r_opt, g_opt), args, st, W1c, W3c = build_pair(L, H, hd, Fh, seed, r)
    hp = OU.UvcHyper(budget=r["budget"], slr=r["slr"], rlr=r["rlr"], glr=r["glr"], ylr=r["ylr"], plr=r["plr"],
                     zlr=float(r["zlr"]), sl2wd=r["sl2wd"], z_grad_clip=clip, gating_interval=2,
                     gating_weight=r["gating_weight"], use_gumbel=use_gumbel, enable_block_gating=1)
    rs = np.random.RandomState(7)
    g_cpu = torch.tensor([-1.0, 1.0]).expand(L, 2).contiguous().clone()

    class Opt:
        param_groups = [dict(lr=3e-3)]

    ggl = []
    for step in range(1, 4):
        e1 = torch.from_numpy(rs.exponential(size=(L, 2)).astype(np.float32))
        e2 = torch.from_numpy(rs.exponential(size=(L, 2)).astype(np.float32))
        gg = torch.from_numpy((rs.standard_normal((L, 2)) * 0.1).astype(np.float32))
        draws = [e1.cuda(), e2.cuda()] if use_gumbel else []
        mm.exp_source = lambda shape, d=draws: d.pop(0)
        model.block_skip_gating.grad = gg.cuda()
        cur, s_v, r_v, g_v, ggl = uvc_optimizer(Opt, mm, s_opt, r_opt, g_opt, dual_opt, args, {}, [], None, clip, step, 2, ggl)
        cur_o = OU.uvc_update(st, hp, W1c, W3c, 3e-3, g_cpu, gg, e1 if use_gumbel else None, e2 if use_gumbel else None, 0, step)
        assert abs(float(cur) - cur_o) <= 1e-5 * abs(cur_o) + 1e-6
        for name, a, b in (("s", s_v.numpy(), st.s), ("r", r_v.numpy(), st.r), ("y", mm.y.data.cpu().numpy(), st.y),
                           ("p", mm.p.data.cpu().numpy(), st.p), ("z", mm.z.data.cpu().numpy(), st.z),
                           ("g", g_v.numpy(), g_cpu)):
            np.testing.assert_allclose(a, b.numpy(), rtol=1e-4, atol=1e-7, err_msg=f"{name} step {step}")
        # shrunk weights: same index sets and correctly-rounded float32 division -> bit-exact
        for l, blk in enumerate(model.blocks):
            assert torch.equal(blk.attn.proj.weight.data.cpu(), W1c[l]), f"W1 layer {l} step {step}"
            assert torch.equal(blk.mlp.fc2.weight.data.cpu(), W3c[l]), f"W3 layer {l} step {step}"
        # emulate the AdamW update between steps
        for l, blk in enumerate(model.blocks):
            d1 = torch.from_numpy((rs.standard_normal(W1c[l].shape) * 1e-3).astype(np.float32))
            d3 = torch.from_numpy((rs.standard_normal(W3c[l].shape) * 1e-3).astype(np.float32))
            W1c[l] += d1; W3c[l] += d3
            blk.attn.proj.weight.data.copy_(W1c[l]); blk.mlp.fc2.weight.data.copy_(W3c[l])
    # masks: bit-exact index sets, fc1 rows follow fc2 columns, count = sum of ceil()s
    prune_w_mask(mm)
    masks = OU.prune_masks(st, W1c, W3c)
    for l, blk in enumerate(model.blocks):
        mp, mf2, mf1, keep1, keep3 = masks[l]
        assert torch.equal(blk.attn.proj.mask.cpu(), mp)
        assert torch.equal(blk.mlp.fc2.mask.cpu(), mf2)
        assert torch.equal(blk.mlp.fc1.mask.cpu(), mf1)
        assert int((~keep3).sum()) == int(st.s[l, 1].ceil())
    # final FLOPs ratio (hard gates)
    e = torch.from_numpy(rs.exponential(size=(L, 2)).astype(np.float32))
    mm.exp_source = lambda shape: e.cuda()
    s2 = [OU.scores_w1(W, H, hd)[1] for W in W1c]
    ref = float(OU.resource(st, s2, g_cpu, e if use_gumbel else None, hp, hard=True))
    assert abs(float(mm.run_resource_fn(gumbel_hard=True)) - ref) <= 1e-5 * abs(ref)


def test_scores_and_ranks_full_size_properties():
    """DeiT-Base sizes: scores equal the float64-accumulated column sums, ranks are permutations,
    prox is idempotent in its index sets when lr = 0 (weights untouched)."""
    from uvc_amd.uvc_utils import prox_w
    r = dict(SC.DEFAULTS, warmup=0, state="pruned")
    model, mm, _, args, st, W1c, W3c = build_pair(12, 12, 64, 3072, 5, r)
    mm.refresh_scores()
    for l in range(12):
        s1, s2 = OU.scores_w1(W1c[l], 12, 64)
        assert torch.equal(mm._sc[0][l].cpu().view(12, 64), s1)
        assert torch.equal(mm._sc[1][l].cpu(), s2)
        assert torch.equal(mm._sc[2][l].cpu(), OU.scores_w3(W3c[l]))
        assert sorted(mm._rk[2][l].cpu().tolist()) == list(range(3072))
        order = torch.argsort(OU.scores_w3(W3c[l]), stable=True)
        rk = torch.empty(3072, dtype=torch.int64); rk[order] = torch.arange(3072)
        assert torch.equal(mm._rk[2][l].cpu().long(), rk)

    class Opt0:
        param_groups = [dict(lr=0.0)]

    prox_w(mm, Opt0)
    for l, blk in enumerate(model.blocks):
        assert torch.equal(blk.attn.proj.weight.data.cpu(), W1c[l])
        assert torch.equal(blk.mlp.fc2.weight.data.cpu(), W3c[l])


def test_warmup_returns_early():
    from uvc_amd.uvc_optimizer import uvc_optimizer
    r = dict(SC.DEFAULTS, warmup=1, state="pruned")
    model, mm, (dual_opt, s_opt, r_opt, g_opt), args, st, W1c, W3c = build_pair(2, 2, 64, 512, 9, r)
    before = mm._flat.clone()
    e1 = torch.from_numpy(np.random.RandomState(3).exponential(size=(2, 2)).astype(np.float32))
    mm.exp_source = lambda shape: e1.cuda()

    class Opt:
        param_groups = [dict(lr=1e-3)]

    cur, s_v, r_v, g_v, ggl = uvc_optimizer(Opt, mm, s_opt, r_opt, g_opt, dual_opt, args, {}, [], None, 0.5, 1, 50, [])
    hp = OU.UvcHyper()
    g_cpu = torch.tensor([-1.0, 1.0]).expand(2, 2).contiguous().clone()
    cur_o = OU.uvc_update(st, hp, W1c, W3c, 1e-3, g_cpu, None, e1, None, 1, 1)
    assert abs(float(cur) - cur_o) < 1e-5
    assert torch.equal(mm._flat[:-4], before[:-4])   # no primal/dual update in warm-up (the last four floats are the step's report: resource ...)
    assert torch.equal(model.blocks[0].attn.proj.weight.data.cpu(), W1c[0])   # but prox ran (:42)


def test_fc1_mask_is_sticky_like_the_reference():
    """k_masks against the fixture of the REFERENCE's prune_w_mask over three non-monotone primal states
    (tests/golden/mask_sticky_micro.npz): proj / fc2 masks are rebuilt on every call, the fc1 mask only accumulates zeros
    (uvc_utils.py:382,393,401) -- index sets bit-exact, and count_mask equals the reference's after every call."""
    from helpers import load_golden
    from stage1_driver import Stage1Run
    from uvc_amd.joint_train import count_mask
    from uvc_amd.uvc_utils import prune_w_mask
    gold = load_golden("mask_sticky_micro")
    run = Stage1Run(str(gold["scenario"]), precision="fp32")
    for grp in ("W1", "W2", "W3"):                   # Stage1Run already pruned once at state 0: start from clean masks
        for m in run.layers[grp]:
            m.mask.fill_(1.0)
    mm = run.minimax
    for i in range(3):
        mm.s.data.copy_(torch.from_numpy(gold[f"call{i}.s"])); mm.r.data.copy_(torch.from_numpy(gold[f"call{i}.r"]))
        prune_w_mask(mm, run.optimizer)
        for l in range(run.cfg.depth):
            w1, w2, w3 = (run.layers[g][l].mask.cpu() for g in ("W1", "W2", "W3"))
            assert bool((w1 == w1[0:1]).all()) and bool((w3 == w3[0:1]).all()) and bool((w2 == w2[:, 0:1]).all())
            assert np.array_equal(np.packbits(w1[0].numpy().astype(np.uint8)), gold[f"call{i}.keep_proj.{l}"]), (i, l)
            assert np.array_equal(np.packbits(w3[0].numpy().astype(np.uint8)), gold[f"call{i}.keep_fc2.{l}"]), (i, l)
            assert np.array_equal(np.packbits(w2[:, 0].numpy().astype(np.uint8)), gold[f"call{i}.keep_fc1.{l}"]), (i, l)
        assert abs(float(count_mask(run.model)) - float(gold[f"call{i}.count"])) < 1e-6
