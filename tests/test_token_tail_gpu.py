"""GPU: the last block's tail on the class / distillation token rows only (uvc_amd/csrc/token_tail.hip, uvc_vit_io.full_tail).

Only rows 0 .. ntok-1 of the last block's output reach the head (UVC/models/model_distilled.py:507-526), so the engine computes that
block's attention output, proj, LayerNorm2, MLP and their backward on those rows alone.  Checked here:
  * the token-query attention kernels against float64 math (forward, and the FULL dqkv of the backward with a gradient that is zero
    off the token rows), ragged N, one and two tokens, both precisions, head skipping;
  * the row gather / scatter;
  * the engine with full_tail = 0 (default) against full_tail = 1 (every row, as the reference executes it): logits, loss and EVERY
    gradient agree to rounding (float32 mode 1e-5 relative to the gradient scale; bf16 mode within bf16 noise), for DeiT with one and
    two tokens, with soft block gating (UVC-train), warm-up and hard block skip, and for the no-grad teacher forward.
The reference goldens (tests/test_stage1_gpu.py, test_stage2_gpu.py, ...) run with the default, i.e. they pin the token-row path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1


def dev():
    return torch.device("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, device="cuda", generator=g) * scale


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("B,N,H,ntok", [(3, 197, 3, 1), (2, 198, 2, 2), (2, 17, 1, 1), (1, 256, 2, 2), (4, 5, 1, 2)])
def test_token_attention_matches_float64(dtype, B, N, H, ntok):
    from uvc_amd import ops
    D = H * 64
    T = torch.float32 if dtype == F32 else torch.bfloat16
    qkv = rnd(B, N, 3 * D, seed=31).to(T)
    dout = rnd(B, ntok, D, seed=32).to(T)
    o = torch.full((B, ntok, D), float("nan"), device=dev(), dtype=T)
    ops.attention_tok_fwd(qkv, o, B, N, H, ntok, dtype)
    x = qkv.double().requires_grad_(True)
    q, k, v = x.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * 64 ** -0.5
    full = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, D)
    ref = full[:, :ntok]
    t = dict(rtol=1e-4, atol=1e-5) if dtype == F32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(o.double(), ref, **t)
    dqkv = torch.full((B, N, 3 * D), float("nan"), device=dev(), dtype=T)
    ops.attention_tok_bwd(qkv, o, dout, dqkv, B, N, H, ntok, dtype)
    g = torch.zeros(B, N, D, device=dev(), dtype=torch.float64)
    g[:, :ntok] = dout.double()
    full.backward(g)
    tb = dict(rtol=2e-4, atol=2e-5) if dtype == F32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dqkv.double(), x.grad, **tb)
    dq = dqkv.view(B, N, 3, D)[:, ntok:, 0]
    assert float(dq.float().abs().sum()) == 0.0          # queries that never reach the head get exact zeros
    # the full-row kernel on the same operands agrees on the token rows
    o_full = torch.empty(B, N, D, device=dev(), dtype=T)
    lse = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv, o_full, lse, B, N, H, dtype)
    torch.testing.assert_close(o.float(), o_full[:, :ntok].float(), **(dict(rtol=1e-4, atol=1e-5) if dtype == F32 else dict(rtol=3e-2, atol=3e-2)))


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_token_attention_head_keep(dtype):
    from uvc_amd import ops
    B, N, H, ntok = 2, 197, 3, 1
    T = torch.float32 if dtype == F32 else torch.bfloat16
    qkv = rnd(B, N, 3 * H * 64, seed=33).to(T)
    a = torch.empty(B, ntok, H * 64, device=dev(), dtype=T)
    b = torch.full((B, ntok, H * 64), float("nan"), device=dev(), dtype=T)
    ops.attention_tok_fwd(qkv, a, B, N, H, ntok, dtype)
    ops.attention_tok_fwd(qkv, b, B, N, H, ntok, dtype, head_keep=torch.tensor([1, 0, 1], device=dev(), dtype=torch.int32))
    a, b = a.view(B, ntok, H, 64), b.view(B, ntok, H, 64)
    assert torch.equal(a[:, :, 0], b[:, :, 0]) and torch.equal(a[:, :, 2], b[:, :, 2])
    assert float(b[:, :, 1].float().abs().sum()) == 0.0


def test_copy_row_groups_gathers_and_scatters_token_rows():
    from uvc_amd import ops
    B, N, ntok, D = 5, 197, 2, 192
    x = rnd(B, N, D, seed=34)
    c = torch.empty(B, ntok, D, device=dev())
    ops.copy_row_groups(x, c, B, ntok * D * 4, N * D * 4, ntok * D * 4)
    assert torch.equal(c, x[:, :ntok])
    y = torch.zeros(B, N, D, device=dev(), dtype=torch.bfloat16)
    cb = c.bfloat16()
    ops.copy_row_groups(cb, y, B, ntok * D * 2, ntok * D * 2, N * D * 2)
    assert torch.equal(y[:, :ntok], cb) and float(y[:, ntok:].float().abs().sum()) == 0.0


def _trainer(precision, batch, **over):
    from uvc_amd.stage1 import Stage1Trainer, default_args
    torch.manual_seed(730)
    a = default_args(precision=precision, train_batch_size=batch, **over)
    tr = Stage1Trainer(a, device="cuda")
    mm = tr.minimax
    L, H, F = mm.n_layers, mm.num_heads, mm.dims.F
    rs = np.random.RandomState(731)
    s = np.zeros((L, 2), np.float32); s[:, 0] = rs.uniform(0, 0.6 * (H - 1) + 0.3, L); s[:, 1] = rs.uniform(0, 0.5 * F, L)
    mm.s.data.copy_(torch.from_numpy(s)); mm.r.data.copy_(torch.from_numpy(rs.uniform(0, 30.0, (L, H)).astype(np.float32)))
    return tr


def _fwd_bwd(tr, x, y, full_tail):
    m = tr.model
    m.full_tail = full_tail
    tr.teacher.full_tail = full_tail
    L = m._cfg.depth
    e = torch.empty(L, 2, device="cuda").exponential_(generator=torch.Generator(device="cuda").manual_seed(9))
    m.exp_source = lambda shape, e=e: e.clone()
    m._flat_grad.zero_()
    outputs, _ = m(x, -1, tr.args.patch_ratio)
    loss = tr.criterion(x, outputs, y)
    loss.backward()
    torch.cuda.synchronize()
    n = m._off.n_total
    with torch.no_grad():
        teacher = tr.teacher(x)
        teacher = teacher[0] if isinstance(teacher, (tuple, list)) else teacher
    return [o.detach().clone() for o in outputs if o is not None], float(loss.detach()), m._flat_grad[:n].clone(), teacher.clone()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["tiny_uvc_train", "tiny_warmup", "tiny_deit_tokens", "tiny_patch_gating"])
def test_token_row_tail_equals_the_full_rows(precision, case):
    """Same weights, inputs and gate noise: the token-row tail (default) and the reference's full-row execution give the same logits,
    loss and gradients (every parameter, incl. the last block's and the gate logits)."""
    over = dict()
    if case == "tiny_deit_tokens":
        over = dict(enable_deit=1)
    if case == "tiny_patch_gating":
        over = dict(enable_patch_gating=2)
    B = 16
    tr = _trainer(precision, B, **over)
    tr.begin_epoch(1 if case == "tiny_warmup" else tr.args.warmup_epochs + 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    y = torch.softmax(torch.randn(B, 1000, device="cuda", generator=g), -1)
    tr.model.train()
    if case == "tiny_patch_gating":
        src = tr.model.exp_source
    lo_f, loss_f, g_f, t_f = _fwd_bwd(tr, x, y, True)
    lo_t, loss_t, g_t, t_t = _fwd_bwd(tr, x, y, False)
    assert torch.isfinite(g_t).all() and float(g_t.abs().max()) > 0
    if precision == "fp32":
        for a, b in zip(lo_t, lo_f):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(t_t, t_f, rtol=1e-4, atol=1e-5)
        assert abs(loss_t - loss_f) <= 1e-5 * abs(loss_f)
        scale = float(g_f.abs().max())
        assert float((g_t - g_f).abs().max()) <= 2e-5 * scale, float((g_t - g_f).abs().max()) / scale
    else:
        for a, b in zip(lo_t, lo_f):
            torch.testing.assert_close(a, b, rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(t_t, t_f, rtol=3e-2, atol=3e-2)
        assert abs(loss_t - loss_f) <= 2e-3 * abs(loss_f)
        # bf16 operands round differently on the two paths: compare in the norm
        assert float((g_t - g_f).norm()) <= 3e-2 * float(g_f.norm()), float((g_t - g_f).norm()) / float(g_f.norm())


def test_token_row_tail_with_hard_block_skip_fp32():
    """Stage-2 style execution (no gate distribution, run_block list): the last block THAT RUNS gets the token-row tail."""
    from uvc_amd.model_distilled import DistilledVisionTransformer
    outs = {}
    for full in (True, False):
        torch.manual_seed(3)
        m = DistilledVisionTransformer(0, enable_block_gating=0, embed_dim=192, depth=12, num_heads=3, precision="fp32").cuda()
        m.full_tail = full
        with torch.no_grad():
            m.block_skip_gating.data[:, 0] = 0.0
            m.block_skip_gating.data[:, 1] = 1.0
            m.block_skip_gating.data[11] = torch.tensor([1.0, 0.0])     # last block skipped: block 10 is the last that runs
            m.block_skip_gating.data[4] = torch.tensor([1.0, 0.0])
        g = torch.Generator(device="cuda").manual_seed(6)
        x = torch.randn(4, 3, 224, 224, device="cuda", generator=g)
        m.train()
        out = m(x, -1, 0.9)
        while isinstance(out, (tuple, list)):
            out = out[0]
        logits = out
        logits.square().mean().backward()
        torch.cuda.synchronize()
        grad = m._flat_grad[:m._off.n_total].clone()
        assert float(grad.abs().max()) > 0
        outs[full] = (logits.detach().clone(), grad)
    torch.testing.assert_close(outs[False][0], outs[True][0], rtol=1e-4, atol=1e-5)
    scale = float(outs[True][1].abs().max())
    assert float((outs[False][1] - outs[True][1]).abs().max()) <= 2e-5 * scale


def test_fused_training_mlp_in_the_engine_matches_the_three_kernels():
    """uvc_vit_io.fused_train_mlp = 1 (opt-in): LayerNorm2 + fc1 + GELU / GELU' + fc2 + gate mix of every block as one kernel that also
    stores the backward's operands -- same logits, loss and gradients as the three-kernel forward up to bf16 rounding order."""
    B = 16
    tr = _trainer("bf16", B)
    tr.begin_epoch(tr.args.warmup_epochs + 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    y = torch.softmax(torch.randn(B, 1000, device="cuda", generator=g), -1)
    tr.model.train()
    res = {}
    for fused in (False, True):
        tr.model.fused_train_mlp = fused
        res[fused] = _fwd_bwd(tr, x, y, False)
    (lo_a, loss_a, g_a, _), (lo_b, loss_b, g_b, _) = res[False], res[True]
    for a, b in zip(lo_a, lo_b):
        torch.testing.assert_close(a, b, rtol=3e-2, atol=3e-2)
    assert abs(loss_a - loss_b) <= 2e-3 * abs(loss_a)
    assert float((g_a - g_b).norm()) <= 3e-2 * float(g_a.norm()), float((g_a - g_b).norm()) / float(g_a.norm())


def test_shared_patch_rows_give_identical_results():
    """Student and teacher take the batch's patch rows from one uvc_patchify (uvc_vit_io.patches_in, model_distilled._shared_patches):
    bit-identical logits, loss, gradients and teacher output to each model rearranging the batch itself; an entry is consumed once."""
    import uvc_amd.model_distilled as MD
    B = 8
    tr = _trainer("bf16", B)
    tr.begin_epoch(tr.args.warmup_epochs + 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    y = torch.softmax(torch.randn(B, 1000, device="cuda", generator=g), -1)
    tr.model.train()
    res = {}
    old = MD._SHARE_PATCHES
    try:
        for share in (False, True):
            MD._SHARE_PATCHES = share
            MD._PATCH_SHARE.update(key=None, buf=None, ev=None, owner=None)
            m = tr.model
            m._flat_grad.zero_()
            e = torch.empty(m._cfg.depth, 2, device="cuda").exponential_(generator=torch.Generator(device="cuda").manual_seed(9))
            m.exp_source = lambda shape, e=e: e.clone()
            tr.criterion.prefetch(x)                                  # teacher first, on its side stream (as Stage1Trainer.step does)
            outputs, _ = m(x, -1, tr.args.patch_ratio)
            loss = tr.criterion(x, outputs, y)
            loss.backward()
            torch.cuda.synchronize()
            if share:
                assert MD._PATCH_SHARE["key"] is None                 # produced by the teacher, consumed by the student
            res[share] = (outputs[0].detach().clone(), float(loss.detach()), m._flat_grad[:m._off.n_total].clone())
    finally:
        MD._SHARE_PATCHES = old
    assert torch.equal(res[True][0], res[False][0]) and res[True][1] == res[False][1] and torch.equal(res[True][2], res[False][2])


@pytest.mark.parametrize("fused_train", [False, True])
def test_norm1_written_by_the_producer_of_its_rows_matches_the_stand_alone_pass(fused_train):
    """uvc_vit_io.fuse_next_ln (default 1): the kernel that produces a block's output rows -- uvc_mlp_fused_fwd in the teacher / eval forward
    and in the opt-in fused training forward, the fc2 + residual + gate-mix GEMM in the default training forward -- also writes norm1 of
    the next block (and its statistics for the backward) and that block's stand-alone LayerNorm pass is skipped.  Same float32 formula on
    the same rows (another summation order inside a row): logits, loss, gradients and the teacher's output agree with fuse_next_ln = 0
    to bf16 rounding."""
    B = 16
    tr = _trainer("bf16", B)
    tr.begin_epoch(tr.args.warmup_epochs + 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, 3, 224, 224, device="cuda", generator=g)
    y = torch.softmax(torch.randn(B, 1000, device="cuda", generator=g), -1)
    tr.model.train()
    tr.model.fused_train_mlp = fused_train
    res = {}
    for fuse in (False, True):
        tr.model.fuse_next_ln = fuse
        tr.teacher.fuse_next_ln = fuse
        res[fuse] = _fwd_bwd(tr, x, y, False)
    (lo_a, loss_a, g_a, t_a), (lo_b, loss_b, g_b, t_b) = res[False], res[True]
    for a, b in zip(lo_a, lo_b):
        torch.testing.assert_close(a, b, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(t_a, t_b, rtol=3e-2, atol=3e-2)
    assert abs(loss_a - loss_b) <= 2e-3 * abs(loss_a)
    assert float((g_a - g_b).norm()) <= 3e-2 * float(g_a.norm()), float((g_a - g_b).norm()) / float(g_a.norm())
