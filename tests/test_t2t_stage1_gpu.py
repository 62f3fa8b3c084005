"""GPU: Stage-1 UVC steps on T2T-ViT (BASELINE config 5's model family at micro size) -- the product's Stage1Trainer with
the HIP T2T_ViT against the oracle step (oracle/step.py) driven with oracle/t2t.py's forward on CPU, same weights, inputs
and Exp(1) draws.  The reference cannot run this path, so the parity here is to the oracle's restatement of
the lines as written (gated T2T forward UNPINNED); the ungated forward underneath is pinned by test_t2t_model_gpu.py."""
import numpy as np
import pytest
import torch

import scenarios as SC
import t2t_scenarios as TS
from helpers import train_hyper, uvc_hyper, student_flags
from oracle import step as OS
from oracle import t2t as OT
from oracle import uvc as OU

pytestmark = pytest.mark.gpu


def close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b); tol = atol + rtol * np.abs(b)
    assert np.all(err <= tol), f"{what}: max err {err.max():.3e} vs tol {tol.flat[err.argmax()]:.3e} (ref {b.flat[err.argmax()]:.4e})"


def build(name, precision):
    from uvc_amd.stage1 import Stage1Trainer, default_args
    from uvc_amd.uvc_utils import prune_w_mask
    r = TS.stage1_recipe(name)
    m = r["model_cfg"]
    cfg = OT.T2TConfig(**m)
    params = OT.init_params_numpy(cfg, r["seed"], weight_gain=r["weight_gain"], enable_patch_gating=r["enable_patch_gating"])
    teacher = OT.init_params_numpy(cfg, r["seed"] + 500, weight_gain=r["weight_gain"])
    # ---- oracle
    with torch.no_grad():
        _, (embed, macs) = OT.forward(params, cfg, torch.ones(1, 3, cfg.img_size, cfg.img_size))
    st = OU.UvcState.create(cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden, embed, macs, eps=r["eps"])
    s0, r0, y0, p0, z0 = SC.initial_state(r, cfg.depth, cfg.num_heads, cfg.head_dim, cfg.hidden)
    st.s, st.r = torch.from_numpy(s0.copy()), torch.from_numpy(r0.copy())
    st.y, st.p, st.z = torch.from_numpy(y0.copy()), torch.from_numpy(p0.copy()), torch.tensor(float(z0))
    frozen = ("pos_embed", "tokens_to_token.attention1.w", "tokens_to_token.attention2.w") + \
        tuple(f"blocks.{i}.{k}" for i in range(cfg.depth) for k in ("attn_skip_gating", "mlp_skip_gating"))
    S = OS.Stage1(cfg=cfg, flags=student_flags(r), params={k: v.clone() for k, v in params.items()}, teacher=teacher, st=st, hp=uvc_hyper(r),
                  th=train_hyper(r), fwd=OT.forward_flags, frozen=frozen)
    if r["warmup"]:
        S.lr = r["warmup_lr"]
    # ---- product
    a = default_args(
        model_type="t2t_scenario", model_cfg=dict(embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"], mlp_ratio=m["mlp_ratio"]),
        img_size=m["img_size"], num_classes=m["num_classes"], enable_deit=0, precision=precision, learning_rate=r["learning_rate"],
        weight_decay=r["weight_decay"], max_grad_norm=r["max_grad_norm"], warmup_steps=r["warmup_steps"], steps_per_epoch=r["t_total"], num_epochs=1,
        warmup_lr=r["warmup_lr"], distillation_alpha=r["distillation_alpha"], distillation_tau=r["distillation_tau"], enable_patch_gating=r["enable_patch_gating"],
        patch_ratio=r["patch_ratio"], budget=r["budget"], slr=r["slr"], rlr=r["rlr"], glr=r["glr"], ylr=r["ylr"], plr=r["plr"],
        zlr_schedule_list=str(int(r["zlr"])), sl2wd=r["sl2wd"], z_grad_clip=r["z_grad_clip"], gating_interval=r["gating_interval"],
        gating_weight=r["gating_weight"], use_gumbel=r["use_gumbel"], enable_block_gating=r["enable_block_gating"], eps=r["eps"],
        eps_decay=r["eps_decay"], enable_warmup=r["warmup"], warmup_epochs=1 if r["warmup"] else 0)
    tr = Stage1Trainer(a, student_state=params, teacher_state=teacher)
    mm = tr.minimax
    mm.s.data.copy_(torch.from_numpy(s0)); mm.r.data.copy_(torch.from_numpy(r0))
    mm.y.data.copy_(torch.from_numpy(y0)); mm.p.data.copy_(torch.from_numpy(p0)); mm.z.data.fill_(float(z0))
    tr.gating_grad_list = []
    if r["warmup"]:
        tr.model.enable_warmup = 1
        tr.model.block_skip_gating.requires_grad = False
        for g in tr.optimizer.param_groups:
            g["lr"] = r["warmup_lr"]
    else:
        tr.model.enable_warmup = 0
        tr.model.block_skip_gating.requires_grad = True
    prune_w_mask(mm, tr.optimizer)
    r["_fc1_masks0"] = [m[2] for m in OU.prune_masks(S.st, S.w1(), S.w3())]       # the start-of-epoch call above (fc1 masks are sticky)
    return r, cfg, S, tr


def run(name, precision):
    r, cfg, S, tr = build(name, precision)
    f32 = precision == "fp32"
    rt = 1e-3 if f32 else 3e-2
    x_all, y_all = TS.stage1_inputs(r)
    draws = TS.stage1_draws(r, cfg.depth)
    gum = r["enable_block_gating"] and r["use_gumbel"]
    for step in range(r["steps"]):
        md, e1, e2 = draws[step]
        md_t = [torch.from_numpy(d) for d in md] if (gum and not r["warmup"]) else []
        e1_t = torch.from_numpy(e1) if gum else None
        e2_t = torch.from_numpy(e2) if (gum and not r["warmup"]) else None
        x, y = torch.from_numpy(x_all[step]), torch.from_numpy(y_all[step])
        ref = {}
        OS.stage1_step(S, x, y, list(md_t), e1_t, e2_t, out=ref)
        if md_t:
            by_shape = {}
            blocks = list(md_t)
            if r["enable_patch_gating"] == 2:
                pd = blocks.pop(0)
                by_shape[tuple(pd.shape)] = pd.cuda()
            by_shape[(cfg.depth, 2)] = torch.stack(blocks).cuda()
            tr.model.exp_source = lambda shape, t=by_shape: t[tuple(shape)]
        q = [t.cuda() for t in (e1_t, e2_t) if t is not None]
        tr.minimax.exp_source = lambda shape, q=q: q.pop(0)
        out = tr.step(x.cuda(), y.cuda(), tau=r["patch_tau"] if r["enable_patch_gating"] == 2 else -1, zero_grad=False)
        pre = f"step{step} "
        if r["enable_patch_gating"] == 2:           # the kept-token index sets: bit-exact
            hard = torch.zeros(r["batch"], cfg.num_patches).scatter_(1, ref["patch_index"], 1.0) > 0.5
            hard[:, 0] = True
            assert torch.equal(tr.model.last_patch_mask.cpu() > 0.5, hard), pre + "patch index sets"
        close(float(out["loss"]), float(ref["loss"]), rt, 1e-6, pre + "loss")
        close(out["outputs"][0].detach().cpu().numpy(), ref["logits"].numpy(), rt, 3e-4 if f32 else 5e-2, pre + "logits")
        close(float(out["gnorm"]), float(ref["grad_norm"]), rt if f32 else 5e-2, 0, pre + "grad_norm")
        close(float(out["cur"]), float(ref["cur_resource"]), 1e-4, 0, pre + "cur_resource")
        stt = 1e-3 if f32 else 2e-2
        close(out["s"].numpy(), S.st.s.numpy(), stt, 1e-6, pre + "s")
        close(out["r"].numpy(), S.st.r.numpy(), stt, 1e-6, pre + "r")
        close(tr.minimax.y.data.cpu().numpy(), S.st.y.numpy(), stt, 1e-7, pre + "y")
        close(tr.minimax.p.data.cpu().numpy(), S.st.p.numpy(), stt, 1e-7, pre + "p")
        close(float(tr.minimax.z), float(S.st.z), 1e-4, 0, pre + "z")
        close(out["g"].numpy(), S.params["block_skip_gating"].detach().numpy(), stt, 1e-6, pre + "gating")
        named = dict(tr.model.named_parameters())
        # the oracle's clip scaled its gradients in place; the product arms the clip and applies it inside the fused AdamW
        coef = min(1.0, r["max_grad_norm"] / (float(out["gnorm"]) + 1e-6))
        for k, gref in ref["grads"].items():
            got = named[k].grad
            if gref is None:
                assert got is None, k
                continue
            assert got is not None, k
            if k == "block_skip_gating":
                continue
            sc = float(gref.abs().max()) + 1e-12
            aerr = float((got.cpu() * coef - gref).abs().max())
            # + 1e-7 absolute: d(gumbel.bias) is exactly 0 in real arithmetic (log_softmax is shift invariant), both sides hold rounding noise
            assert aerr < (3e-3 if f32 else 8e-2) * sc + 1e-7, (pre, k, aerr / sc)
        for k, v in S.params.items():                       # weights after clip + AdamW + prox
            a_, b_ = float(named[k].data.double().abs().sum()), float(v.detach().double().abs().sum())
            assert abs(a_ - b_) <= (1e-4 if f32 else 2e-3) * abs(b_) + 1e-6, (pre, k, a_, b_)
        tr.optimizer.zero_grad()
    return r, cfg, S, tr


@pytest.mark.parametrize("name", list(TS.STAGE1))
def test_t2t_stage1_fp32_matches_oracle(name):
    run(name, "fp32")


@pytest.mark.parametrize("name", ["t2t_micro_train"])
def test_t2t_stage1_bf16_matches_oracle(name):
    run(name, "bf16")


def test_t2t_stage1_masks_bit_exact_vs_oracle():
    """Mask index sets after the steps: bit-exact with the oracle's (least-k on the same scores)."""
    from uvc_amd.uvc_utils import prune_w_mask
    r, cfg, S, tr = run("t2t_micro_train", "fp32")
    prune_w_mask(tr.minimax, tr.optimizer)
    masks = OU.prune_masks(S.st, S.w1(), S.w3(), prev_fc1=r["_fc1_masks0"])
    for l in range(cfg.depth):
        assert torch.equal(tr.uvc_layers["W1"][l].mask[0].cpu().bool(), masks[l][3])
        assert torch.equal(tr.uvc_layers["W3"][l].mask[0].cpu().bool(), masks[l][4])
        assert torch.equal(tr.uvc_layers["W2"][l].mask.cpu(), masks[l][2])          # union of the start and end sets (uvc_utils.py:401)


def test_t2t_14_stage1_step_runs():
    """BASELINE config 5's model: one post-warm-up Stage-1 step of t2t_vit_14 at batch 4 (bf16)."""
    from uvc_amd.stage1 import Stage1Trainer, default_args
    a = default_args(model_type="t2t_vit_14", precision="bf16", train_batch_size=4)
    tr = Stage1Trainer(a)
    tr.begin_epoch(a.warmup_epochs + 1)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4, 3, 224, 224, device="cuda", generator=g)
    y = torch.softmax(torch.randn(4, 1000, device="cuda", generator=g), -1)
    out = tr.step(x, y)
    assert np.isfinite(float(out["loss"])) and np.isfinite(float(out["gnorm"])) and 0 < float(out["cur"]) <= 1.0 + 1e-6
    assert abs(float(tr.flops_list[0]) - 256647680) < 1 and tr.flops_list[1][0] == [87146496, 14902656, 14902656, 29048832, 87146496, 87146496]


def test_t2t_14_config5_patch_and_block_gating_forward_matches_oracle():
    """BASELINE config 5 as written: t2t_vit_14 with enable_patch_gating=2 (P = 196, k = 176) AND block gating, float32 mode,
    against the oracle's forward on CPU with the same weights and Exp(1) draws: kept-token index sets bit-exact, logits
    within 1e-3.  (The reference's T2T forward has no patch-gating step: semantics defined in uvc_amd/t2t_vit.py.)"""
    from uvc_amd.stage1 import Stage1Trainer, default_args
    from oracle import vit as OV
    B, tau, ratio = 2, 0.7, 0.9
    cfg = OT.T2TConfig()
    params = OT.init_params_numpy(cfg, 77, weight_gain=2.0, enable_patch_gating=2)
    a = default_args(model_type="t2t_vit_14", precision="fp32", train_batch_size=B, enable_patch_gating=2, enable_block_gating=1)
    tr = Stage1Trainer(a, student_state=params)
    tr.begin_epoch(a.warmup_epochs + 1)
    rs = np.random.RandomState(78)
    x = torch.from_numpy(rs.standard_normal((B, 3, 224, 224)).astype(np.float32))
    ep = torch.from_numpy(rs.exponential(size=(B, cfg.num_patches)).astype(np.float32))
    eb = [torch.from_numpy(rs.exponential(size=2).astype(np.float32)) for _ in range(cfg.depth)]
    by_shape = {(B, cfg.num_patches): ep.cuda(), (cfg.depth, 2): torch.stack(eb).cuda()}
    tr.model.exp_source = lambda shape, t=by_shape: t[tuple(shape)]
    tr.model.train()
    with torch.no_grad():
        (logits, _), macs = tr.model(x.cuda(), tau, ratio)
    flags = OV.GateFlags(enable_block_gating=1, use_gumbel=1, eps=a.eps, enable_warmup=0, gumbel_hard=False, training=True)
    rec = {}
    with torch.no_grad():
        (ref, _), rmacs = OT.forward_flags(params, cfg, flags, x, tau=tau, ratio=ratio, exp_draws=[ep] + eb, record=rec)
    hard = torch.zeros(B, cfg.num_patches).scatter_(1, rec["patch_index"], 1.0) > 0.5
    hard[:, 0] = True
    got = tr.model.last_patch_mask.cpu() > 0.5
    assert int(got.sum()) in range(B * 175, B * 176 + 1 + B) and torch.equal(got, hard), "kept-token index sets differ from the oracle"
    close(logits.cpu().numpy(), ref.numpy(), 1e-3, 3e-4, "logits")
    assert macs[0] == rmacs[0] and macs[1] == rmacs[1]


def test_t2t_14_config5_stage1_step_bf16():
    """BASELINE config 5 in the throughput mode: one post-warm-up Stage-1 step of t2t_vit_14 with patch + block gating at
    batch 8, tau from the schedule of joint_train.py:404-407; the scorer receives a gradient, 176 (or 175 + token 0) tokens
    are kept per image, and the step is deterministic."""
    from uvc_amd.stage1 import Stage1Trainer, default_args

    def one():
        torch.manual_seed(5)
        a = default_args(model_type="t2t_vit_14", precision="bf16", train_batch_size=8, enable_patch_gating=2)
        tr = Stage1Trainer(a)
        tr.begin_epoch(a.warmup_epochs + 1)
        tr.global_step = 1000
        g = torch.Generator(device="cuda").manual_seed(3)
        x = torch.randn(8, 3, 224, 224, device="cuda", generator=g)
        y = torch.softmax(torch.randn(8, 1000, device="cuda", generator=g), -1)
        out = tr.step(x, y, zero_grad=False)
        return tr, out

    tr, out = one()
    assert abs(tr.get_tau() - (0.1 + 9.9 * 1001 / tr.t_total)) < 1e-12
    assert np.isfinite(float(out["loss"])) and np.isfinite(float(out["gnorm"])) and 0 < float(out["cur"]) <= 1.0 + 1e-6
    kept = (tr.model.last_patch_mask > 0.5).sum(1).cpu()
    assert bool(((kept == 176) | (kept == 177)).all()), kept
    gw = tr.model.gumbel.weight.grad
    assert gw is not None and float(gw.abs().sum()) > 0 and tr.model.gumbel.bias.grad is not None
    tr2, out2 = one()
    assert float(out["loss"]) == float(out2["loss"]) and torch.equal(tr.model._flat, tr2.model._flat)


# ---- Stage-2 masked fine-tune step on T2T-ViT against the fixture from the REFERENCE's own T2T_ViT + autograd
# (tests/golden/make_t2t_stage2_golden.py; Performer dropout p = 0): pins the HIP backward of the tokens-to-token module
def build_stage2(name, precision):
    import os
    from uvc_amd.post_train import Stage2Trainer, default_args, setup
    from test_oracle_t2t import build_stage2_oracle
    r, cfg, S, masks, teacher = build_stage2_oracle(name)
    m = r["model_cfg"]
    args = default_args(model_type="t2t_scenario", model_cfg=dict(embed_dim=m["embed_dim"], depth=m["depth"], num_heads=m["num_heads"], mlp_ratio=m["mlp_ratio"]),
                        img_size=m["img_size"], num_classes=m["num_classes"], enable_deit=0, precision=precision, train_batch_size=r["batch"],
                        learning_rate=r["learning_rate"], weight_decay=r["weight_decay"], max_grad_norm=r["max_grad_norm"], epochs=r["epochs"],
                        warmup_epochs=r["warmup_epochs"], warmup_lr=r["warmup_lr"], min_lr=r["min_lr"], decay_rate=r["decay_rate"], opt_eps=r["opt_eps"],
                        distillation_type=r["distillation_type"], distillation_alpha=r["distillation_alpha"], distillation_tau=r["distillation_tau"],
                        compact_mlp=1, compact_multiple=64)
    _, probe, _ = setup(default_args(**vars(args)), device="cuda")
    state = {k: v.detach().cpu().clone() for k, v in probe.state_dict().items()}
    for k, v in S.params.items():
        state[k] = v.clone()
    for k, v in masks.items():
        state[k[:-len("weight")] + "mask"] = v.clone()
    del probe
    tr = Stage2Trainer(args, checkpoint=state, teacher_state=teacher)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    return r, cfg, tr, g


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_t2t_stage2_matches_reference_golden(precision):
    name = "t2t_stage2_micro"
    r, cfg, tr, g = build_stage2(name, precision)
    f32 = precision == "fp32"
    rt = 1e-3 if f32 else 3e-2
    model = tr.model
    names = [str(n) for n in g["param_names"]]
    pmap = dict(model.named_parameters())
    assert list(pmap.keys()) == names, "named_parameters order differs from the reference"
    assert list(model.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    x_all, y_all = TS.stage1_inputs(r)
    t2t = model.tokens_to_token
    for step in range(r["steps"]):
        tr.begin_epoch(r["epoch_of_step"][step])
        out = tr.step(torch.from_numpy(x_all[step]).cuda(), torch.from_numpy(y_all[step]).cuda(), zero_grad=False)
        pre = f"step{step}."
        close(tr.optimizer.param_groups[0]["lr"], g[pre + "lr"], 1e-12, 0, pre + "lr")
        close(float(out["loss"]), g[pre + "loss"], rt, 1e-6, pre + "loss")
        close(out["outputs"][0].detach().cpu().numpy(), g[pre + "logits"], rt, 3e-4 if f32 else 5e-2, pre + "logits")
        gn = float(out["gnorm"])
        close(gn, g[pre + "grad_norm"], rt if f32 else 5e-2, 0, pre + "grad_norm")
        coef = min(1.0, r["max_grad_norm"] / (gn + 1e-6))
        ref = g[pre + "grad_abs_sum"]
        got = np.array([np.nan if pmap[n].grad is None else float(pmap[n].grad.double().abs().sum()) * coef for n in names])
        assert np.array_equal(np.isnan(got), np.isnan(ref)), [n for n, a, b in zip(names, got, ref) if np.isnan(a) != np.isnan(b)]
        ok = ~np.isnan(ref)
        close(got[ok], ref[ok], 3e-3 if f32 else 8e-2, 1e-6, pre + "grad_abs_sum")
        ga = 1e-6 if f32 else 2e-3
        gr = 3e-3 if f32 else 8e-2
        close(t2t.attention1.kqv.weight.grad[:8].cpu().numpy() * coef, g[pre + "g_kqv1"], gr, ga, pre + "d kqv1")
        close(t2t.attention1.norm1.weight.grad.cpu().numpy() * coef, g[pre + "g_norm1_1"], gr, ga, pre + "d norm1")
        close(t2t.attention2.proj.weight.grad.cpu().numpy() * coef, g[pre + "g_proj2"], gr, ga, pre + "d proj2")
        close(t2t.project.bias.grad.cpu().numpy() * coef, g[pre + "g_project_b"], gr, ga, pre + "d project.bias")
        if f32:
            psum = np.array([float(pmap[n].data.double().abs().sum()) for n in names])
            close(psum, g[pre + "param_abs_sum"], 1e-4, 0, pre + "param_abs_sum")
            close(t2t.attention1.kqv.weight.data[0].cpu().numpy(), g[pre + "kqv1_row0"], 1e-3, 2e-6, pre + "kqv1 row")
        tr.optimizer.zero_grad()
