"""GPU: uvc_amd.t2t_vit.T2T_ViT against (a) the fixtures the REFERENCE's T2T_ViT produced (ungated forward: token
embedding, logits, MAC table, state_dict layout -- tests/golden/t2t_*.npz) and (b) the T2T oracle's autograd on CPU for the
training path the reference does not have in runnable form (gated forward, every gradient)."""
import os

import numpy as np
import pytest
import torch

import t2t_scenarios as TS
from oracle import t2t as OT

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def build(name, precision, **kw):
    from uvc_amd.t2t_vit import T2T_ViT
    r = TS.recipe(name)
    cfg = OT.T2TConfig(**r["model_cfg"])
    sd = OT.init_params_numpy(cfg, r["seed"], weight_gain=r["weight_gain"])
    for i in r.get("skip_blocks", []):
        sd["block_skip_gating"][i] = torch.tensor([1.0, -1.0])
    m = T2T_ViT(img_size=cfg.img_size, num_classes=cfg.num_classes, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                mlp_ratio=cfg.mlp_ratio, token_dim=cfg.token_dim, precision=precision, **kw)
    m.load_state_dict(sd, strict=True)
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return r, cfg, sd, m, torch.from_numpy(TS.make_input(r)), g


@pytest.mark.parametrize("name", list(TS.SCENARIOS))
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_matches_reference_fixture(name, precision):
    r, cfg, sd, m, x, g = build(name, precision)
    m.eval()
    with torch.no_grad():
        logits, (macs_embed, macs_list) = m(x.cuda())
    assert int(macs_embed) == int(g["macs_embed"])
    assert np.array_equal(np.array([q if q else [0] * 6 for q in macs_list], dtype=np.int64), g["macs_list"])
    tok = m._ws_view(x.shape[0], False, "pe").float().cpu().numpy().reshape(g["tokens"].shape)
    # fp32: float32 MFMA GEMMs + float32 VALU attention, only summation order differs.  bf16: operands rounded to 2^-9,
    # float32 accumulation and residual stream; the logits of the random-init net are O(0.5)
    t = dict(rtol=1e-3, atol=2e-5) if precision == "fp32" else dict(rtol=5e-2, atol=6e-2)
    np.testing.assert_allclose(tok, g["tokens"], **t)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], **(dict(rtol=1e-3, atol=5e-5) if precision == "fp32" else dict(rtol=5e-2, atol=8e-2)))


def test_state_dict_layout_matches_reference():
    r, cfg, sd, m, x, g = build("t2t_micro", "fp32")
    keys = list(m.state_dict().keys())
    assert keys == list(g["state_dict_keys"])
    for k, s in zip(keys, g["state_dict_shapes"]):
        assert str(list(m.state_dict()[k].shape)) == str(s), k
    back = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(back[k].cpu(), v), k


def _oracle_grads(sd, cfg, x, gate_d, dlogits):
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and k not in ("pos_embed",) and not k.endswith(".w")) for k, v in sd.items()}
    logits, _ = OT.forward(p, cfg, x, gate_d=gate_d)
    (logits * dlogits).sum().backward()
    return logits.detach(), {k: v.grad for k, v in p.items() if v.grad is not None}


@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_training_step_gradients_match_oracle(gated, precision):
    name = "t2t_micro" if gated else "t2t_micro_skip"
    r, cfg, sd, m, x, g = build(name, precision, enable_block_gating=int(gated), use_gumbel=0, enable_warmup=False)
    m.train()
    gate_d = None
    (out, out2), _ = m(x.cuda())
    assert out is out2 or torch.equal(out, out2)
    dl = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)) * 0.1
    out.backward(dl.cuda())
    if gated:
        gate_d = m.last_distrib.cpu()
    ref_logits, ref = _oracle_grads(sd, cfg, x, gate_d, dl)
    f32 = precision == "fp32"
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_logits.numpy(), **(dict(rtol=1e-3, atol=5e-5) if f32 else dict(rtol=5e-2, atol=8e-2)))
    named = dict(m.named_parameters())
    skipped = {i for i in r.get("skip_blocks", [])}
    checked = 0
    for k, gr in ref.items():
        if k == "block_skip_gating":
            continue
        if any(k.startswith(f"blocks.{i}.") for i in skipped):
            assert named[k].grad is None
            continue
        got = named[k].grad
        if k.endswith("skip_gating"):
            assert got is None
            continue
        assert got is not None, k
        scale = float(gr.abs().max()) + 1e-12
        err = float((got.cpu() - gr).abs().max()) / scale
        assert err < (2e-3 if f32 else 6e-2), (k, err)
        checked += 1
    assert checked > 40
    assert m.pos_embed.grad is None and m.tokens_to_token.attention1.w.grad is None
    # the flat gradient buffer carries nothing for the frozen tensors (the global-norm clip runs over it)
    for off, n in m._frozen_ranges():
        assert float(m._flat_grad[off:off + n].abs().max()) == 0.0


def test_training_forward_is_deterministic_and_optimizer_steps():
    from uvc_amd.optim import FusedAdamW, clip_grad_norm_
    r, cfg, sd, m, x, g = build("t2t_micro", "bf16")
    opt = FusedAdamW(m, lr=1e-3, weight_decay=0.05)
    m.train()
    xs = x.cuda()
    before = m._flat.clone()
    outs = []
    for _ in range(2):
        m.load_state_dict(sd)
        (o, _), _ = m(xs)
        o.backward(torch.ones_like(o) * 0.01)
        outs.append((o.detach().clone(), m._flat_grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    gn = clip_grad_norm_(m, 1.0)
    opt.step()
    assert float(gn) > 0
    after = m._flat
    for off, n in m._frozen_ranges():
        assert torch.equal(after[off:off + n], before[off:off + n])
    f = m._front["attention1"]
    pad = after[f["kqv_w"]:f["kqv_w"] + 192 * 160].view(192, 160)[:, 147:]
    assert float(pad.abs().max()) == 0.0                                    # K padding of attention1.kqv.weight stays zero
    assert not torch.equal(m.tokens_to_token.attention1.kqv.weight.data, sd["tokens_to_token.attention1.kqv.weight"].cuda())


def test_t2t_14_forward_is_batch_independent_and_deterministic():
    """Size-independent property at BASELINE config 5's model size: an image's logits are bit-identical whether it is
    processed in a batch of 2 or of 24 (fixed-order sums everywhere, token-tile splits independent of the batch), and
    two training forward/backward passes give bit-identical gradients."""
    from uvc_amd.t2t_vit import t2t_vit_14
    torch.manual_seed(3)
    m = t2t_vit_14(precision="bf16")
    m.eval()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(24, 3, 224, 224, device="cuda", generator=g)      # 24 images: the streaming K = 384 GEMMs; 2 images: the generic ones
    with torch.no_grad():
        big, _ = m(x)
        small, _ = m(x[:2].contiguous())
    assert torch.equal(big[:2], small)
    m.train()
    outs = []
    for _ in range(2):
        (o, _), _ = m(x[:4].contiguous())
        o.backward(torch.ones_like(o) * 0.01)
        outs.append((o.detach().clone(), m._flat_grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert bool(torch.isfinite(outs[0][1]).all())


def test_t2t_14_full_size_training_gradients_match_oracle():
    """BASELINE config 5's model at a batch large enough (24 images = 4728 token rows) to select the streaming K = 384 GEMMs, the
    7-way token splits of the Performer sums (3136 tokens) and the padded 147 -> 160 soft split: logits and gradients of a
    training forward / backward in the bf16 throughput mode against the oracle's float32 autograd on the host."""
    from uvc_amd.t2t_vit import t2t_vit_14
    cfg = OT.T2TConfig()
    sd = OT.init_params_numpy(cfg, 77, weight_gain=2.0)
    m = t2t_vit_14(precision="bf16")
    m.load_state_dict(sd, strict=True)
    m.train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(24, 3, 224, 224, generator=g)
    dl = torch.randn(24, 1000, generator=g) * 0.05
    (out, _), _ = m(x.cuda())
    out.backward(dl.cuda())
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref_logits, ref = _oracle_grads(sd, cfg, x, None, dl)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_logits.numpy(), rtol=5e-2, atol=8e-2)
    named = dict(m.named_parameters())
    worst = {}
    for k in ("tokens_to_token.attention1.kqv.weight", "tokens_to_token.attention1.norm1.weight", "tokens_to_token.attention2.kqv.weight",
              "tokens_to_token.attention2.mlp.2.weight", "tokens_to_token.project.weight", "blocks.0.attn.qkv.weight", "blocks.0.mlp.fc1.weight",
              "blocks.7.attn.proj.weight", "blocks.13.mlp.fc2.weight", "head.weight", "cls_token", "norm.weight"):
        gr, got = ref[k], named[k].grad
        assert got is not None, k
        worst[k] = float((got.cpu() - gr).abs().max()) / (float(gr.abs().max()) + 1e-12)
    assert max(worst.values()) < 8e-2, worst


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_front_end_stage_in_c_equals_the_python_sequence(precision):
    """uvc_t2t_stage_forward / _backward (one C call per Token_performer stage) issue the launches the Python sequence issues: logits, the token
    embedding and every gradient are bit-identical; twice in a row with gradient accumulation on the second pass (beta = 1)."""
    res = []
    for in_c in (True, False):
        r, cfg, sd, m, x, g = build("t2t_micro", precision, enable_block_gating=1, use_gumbel=0, enable_warmup=False)
        m.front_in_c = in_c
        m.train()
        outs = []
        for rep in range(2):
            m.grad_accumulate = rep == 1
            (out, _), _ = m(x.cuda())
            dl = torch.randn(out.shape, generator=torch.Generator().manual_seed(5 + rep)) * 0.1
            out.backward(dl.cuda())
            outs.append(out.detach().clone())
        torch.cuda.synchronize()
        res.append((outs, m._flat_grad.clone(), m._ws_view(x.shape[0], True, "pe").clone()))
        m.eval()
        with torch.no_grad():
            res[-1] = res[-1] + (m(x.cuda())[0].clone(),)
    (o_c, g_c, pe_c, ev_c), (o_p, g_p, pe_p, ev_p) = res
    assert all(torch.equal(a, b) for a, b in zip(o_c, o_p))
    assert torch.equal(pe_c, pe_p) and torch.equal(ev_c, ev_p)
    assert torch.equal(g_c, g_p)
    assert float(g_c.abs().sum()) > 0
