"""GPU: the tokens-to-token kernels (include/uvc_t2t.h) through the C-ABI against the T2T oracle's float32 ops on CPU
(oracle/t2t.py: soft_split = F.unfold, F.layer_norm, linear_attention; backward = torch autograd).  float32 streams at
tight tolerance, bf16 streams at the tolerance bf16 storage allows (stated per test)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import t2t as OT

pytestmark = pytest.mark.gpu
F32, BF16 = 0, 1
LN_EPS = 1e-5


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def cu(t):
    return t.cuda().contiguous()


CASES = [  # (name, B, C, H, W, k, s, p, token_major)
    ("image", 2, 3, 64, 64, 7, 4, 2, False),
    ("image_ragged", 3, 3, 36, 36, 7, 4, 2, False),
    ("tokens", 2, 64, 16, 16, 3, 2, 1, True),
    ("tokens_odd", 1, 64, 7, 7, 3, 2, 1, True),
    ("tokens_k2", 2, 64, 10, 10, 2, 2, 0, True),          # four taps, no padding: the run-time tap paths of the gather and of the tap-major fold
]


def make_src(B, C, H, W, token_major, seed):
    x = rnd(B, C, H, W, seed=seed)                                   # NCHW view of the data
    if token_major:
        src = x.permute(0, 2, 3, 1).contiguous()                     # [B, H*W, C] storage
        strides = (H * W * C, 1, W * C, C)
    else:
        src = x.contiguous()
        strides = (C * H * W, H * W, W, 1)
    return x, src, strides


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("ln", [True, False])
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_unfold_ln_fwd(case, ln, dtype):
    from uvc_amd import ops
    _, B, C, H, W, k, s, p, tm = case
    x, src, strides = make_src(B, C, H, W, tm, 1)
    dim = C * k * k
    ldo = -(-dim // 32) * 32
    gamma, beta = 1 + 0.1 * rnd(dim, seed=2), 0.1 * rnd(dim, seed=3)
    ref = OT.soft_split(x, k, s, p)
    if ln:
        ref = F.layer_norm(ref, (dim,), gamma, beta, LN_EPS)
    rows = ref.shape[0] * ref.shape[1]
    out = torch.full((rows, ldo), 7.0, device="cuda", dtype=torch.float32 if dtype == F32 else torch.bfloat16)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.unfold_ln_fwd(cu(src), strides, B, C, H, W, k, s, p, out, dtype, gamma=cu(gamma) if ln else None, beta=cu(beta) if ln else None,
                      mean=mean if ln else None, rstd=rstd if ln else None)
    got = out.float().cpu()
    t = dict(rtol=1e-5, atol=1e-5) if dtype == F32 else dict(rtol=1e-2, atol=1e-2)
    np.testing.assert_allclose(got[:, :dim].numpy(), ref.reshape(rows, dim).numpy(), **t)
    assert float(got[:, dim:].abs().max()) == 0.0 if ldo > dim else True          # K padding written as zeros
    if ln:
        u = OT.soft_split(x, k, s, p).reshape(rows, dim)
        np.testing.assert_allclose(mean.cpu().numpy(), u.mean(1).numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rstd.cpu().numpy(), (1.0 / torch.sqrt(u.var(1, unbiased=False) + LN_EPS)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_unfold_ln_bwd_and_fold(case, dtype):
    """dgamma / dbeta, and d(source) = fold(dxu): the adjoint of soft split + LayerNorm (autograd on the oracle ops)."""
    from uvc_amd import ops
    _, B, C, H, W, k, s, p, tm = case
    x, src, strides = make_src(B, C, H, W, tm, 4)
    dim = C * k * k
    ldo = -(-dim // 32) * 32
    gamma, beta = (1 + 0.1 * rnd(dim, seed=5)).requires_grad_(), (0.1 * rnd(dim, seed=6)).requires_grad_()
    xr = x.clone().requires_grad_()
    y = F.layer_norm(OT.soft_split(xr, k, s, p), (dim,), gamma, beta, LN_EPS)
    rows = y.shape[0] * y.shape[1]
    dy = rnd(rows, dim, seed=7)
    if dtype == BF16:
        dy = dy.to(torch.bfloat16).float()
    (y.reshape(rows, dim) * dy).sum().backward()
    Ho, Wo = ops.unfold_out_hw(H, W, k, s, p)
    tdt = torch.float32 if dtype == F32 else torch.bfloat16
    out = torch.empty(rows, ldo, device="cuda", dtype=tdt)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    g_, b_ = cu(gamma.detach()), cu(beta.detach())
    srcd = cu(src)
    ops.unfold_ln_fwd(srcd, strides, B, C, H, W, k, s, p, out, dtype, gamma=g_, beta=b_, mean=mean, rstd=rstd)
    dyp = torch.zeros(rows, ldo, device="cuda", dtype=tdt)
    dyp[:, :dim] = dy.cuda().to(tdt)
    partial = torch.empty(ops.unfold_bwd_blocks(rows) * 2 * dim, device="cuda")
    dgamma, dbeta = torch.full((dim,), 3.0, device="cuda"), torch.full((dim,), 3.0, device="cuda")
    dxu = torch.empty(rows, dim, device="cuda")
    ops.unfold_ln_bwd(srcd, strides, B, C, H, W, k, s, p, dyp, dtype, gamma=g_, mean=mean, rstd=rstd, partial=partial, dgamma=dgamma, dbeta=dbeta, dxu=dxu)
    np.testing.assert_allclose(dgamma.cpu().numpy(), gamma.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dbeta.cpu().numpy(), beta.grad.numpy(), rtol=2e-4, atol=2e-4)
    # accumulate form: beta_acc = 1 adds to the previous value
    ops.unfold_ln_bwd(srcd, strides, B, C, H, W, k, s, p, dyp, dtype, gamma=g_, mean=mean, rstd=rstd, partial=partial, dgamma=dgamma, dbeta=dbeta, dxu=dxu, beta_acc=1.0)
    np.testing.assert_allclose(dgamma.cpu().numpy(), 2 * gamma.grad.numpy(), rtol=2e-4, atol=4e-4)
    if tm:                                                           # fold back onto the token map
        dst = torch.empty(B, H * W, C, device="cuda")
        ops.fold_tokens(dxu, dst, B, C, H, W, k, s, p, dtype)
        ref = xr.grad.permute(0, 2, 3, 1).reshape(B, H * W, C)
        np.testing.assert_allclose(dst.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)
        if C == 64:                                                  # tap-major dxu ([rows][k*k][C]) is the same numbers re-ordered, and folds to the same map
            dxt = torch.empty(rows, dim, device="cuda")
            ops.unfold_ln_bwd(srcd, strides, B, C, H, W, k, s, p, dyp, dtype, gamma=g_, mean=mean, rstd=rstd, partial=partial, dgamma=dgamma, dbeta=dbeta, dxu=dxt,
                              dxu_tap_major=True)
            assert torch.equal(dxt.view(rows, k * k, C), dxu.view(rows, C, k * k).transpose(1, 2))
            dst2 = torch.empty(B, H * W, C, device="cuda")
            ops.fold_tokens(dxt, dst2, B, C, H, W, k, s, p, dtype, tap_major=True)
            assert torch.equal(dst2, dst)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_fold_is_adjoint_of_soft_split(dtype):
    """<unfold(x), g> == <x, fold(g)> for the plain soft split that feeds `project` (t2t_vit.py:100-103)."""
    from uvc_amd import ops
    B, C, H, W, k, s, p = 2, 64, 8, 8, 3, 2, 1
    tdt = torch.float32 if dtype == F32 else torch.bfloat16
    g = rnd(B * 16, C * 9, seed=8).to(tdt)
    dst = torch.empty(B, H * W, C, device="cuda")
    ops.fold_tokens(cu(g), dst, B, C, H, W, k, s, p, dtype)
    ref = F.fold(g.float().reshape(B, 16, C * 9).transpose(1, 2), (H, W), (k, k), stride=s, padding=p)      # [B, C, H, W]
    np.testing.assert_allclose(dst.cpu().numpy(), ref.permute(0, 2, 3, 1).reshape(B, H * W, C).numpy(), rtol=1e-5, atol=1e-5)


PERF = [(3, 200), (2, 3136), (5, 64), (1, 784)]


def perf_inputs(B, T, seed):
    kqv = rnd(B, T, 192, seed=seed, scale=0.5)
    w = rnd(32, 64, seed=seed + 1, scale=0.7)
    return kqv, w


@pytest.mark.parametrize("B,T", PERF)
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_performer_fwd(B, T, dtype):
    from uvc_amd import ops
    kqv, w = perf_inputs(B, T, 10)
    y, _ = OT.linear_attention(kqv, w)
    S = ops.performer_splits(B, T)
    part = torch.empty(B * S * 65 * 32, device="cuda")
    kptv = torch.empty(B, 65, 32, device="cuda")
    att = torch.empty(B * T, 64, device="cuda", dtype=torch.float32 if dtype == F32 else torch.bfloat16)
    ops.performer_fwd(cu(kqv.reshape(B * T, 192)), cu(w), part, kptv, att, B, T, dtype)
    k, q, v = torch.split(kqv, 64, dim=-1)
    kp = OT.prm_exp(k, w)
    np.testing.assert_allclose(kptv[:, :64].cpu().numpy(), torch.einsum("bin,bim->bnm", v, kp).numpy(), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(kptv[:, 64].cpu().numpy(), kp.sum(1).numpy(), rtol=2e-5, atol=1e-5)
    t = dict(rtol=2e-5, atol=1e-5) if dtype == F32 else dict(rtol=1e-2, atol=1e-2)    # bf16: output storage rounding 2^-9
    np.testing.assert_allclose(att.float().cpu().numpy(), y.reshape(B * T, 64).numpy(), **t)


@pytest.mark.parametrize("B,T", PERF)
@pytest.mark.parametrize("dtype", [F32, BF16])
def test_performer_bwd(B, T, dtype):
    from uvc_amd import ops
    kqv, w = perf_inputs(B, T, 20)
    tdt = torch.float32 if dtype == F32 else torch.bfloat16
    datt = rnd(B, T, 64, seed=22).to(tdt).float()
    dskip = rnd(B, T, 64, seed=23).to(tdt).float()
    kr = kqv.clone().requires_grad_()
    y, v = OT.linear_attention(kr, w)
    ((y * datt).sum() + (v * dskip).sum()).backward()
    S = ops.performer_splits(B, T)
    part = torch.empty(B * S * 65 * 32, device="cuda")
    kptv, dkptv = torch.empty(B, 65, 32, device="cuda"), torch.empty(B, 65, 32, device="cuda")
    att = torch.empty(B * T, 64, device="cuda", dtype=tdt)
    kd, wd = cu(kqv.reshape(B * T, 192)), cu(w)
    ops.performer_fwd(kd, wd, part, kptv, att, B, T, dtype)
    dkqv = torch.full((B * T, 192), 9.0, device="cuda", dtype=tdt)
    ops.performer_bwd(kd, wd, part, kptv, cu(datt.reshape(B * T, 64).to(tdt)), dkqv, dkptv, B, T, dtype, dskip=cu(dskip.reshape(B * T, 64).to(tdt)))
    ref = kr.grad.reshape(B * T, 192).numpy()
    got = dkqv.float().cpu().numpy()
    scale = float(np.abs(ref).max())
    t = dict(rtol=2e-4, atol=2e-5 * scale) if dtype == F32 else dict(rtol=2e-2, atol=1e-2 * scale)
    np.testing.assert_allclose(got, ref, **t)
    # without the skip gradient dv is the attention's alone
    ops.performer_bwd(kd, wd, part, kptv, cu(datt.reshape(B * T, 64).to(tdt)), dkqv, dkptv, B, T, dtype)
    np.testing.assert_allclose(dkqv.float().cpu().numpy()[:, 128:], ref[:, 128:] - dskip.reshape(B * T, 64).numpy(), **t)


def test_performer_is_deterministic():
    from uvc_amd import ops
    B, T = 4, 3136
    kqv, w = perf_inputs(B, T, 30)
    kd, wd = cu(kqv.reshape(B * T, 192)), cu(w)
    S = ops.performer_splits(B, T)
    outs = []
    for _ in range(2):
        part = torch.empty(B * S * 65 * 32, device="cuda")
        kptv = torch.empty(B, 65, 32, device="cuda")
        att = torch.empty(B * T, 64, device="cuda")
        ops.performer_fwd(kd, wd, part, kptv, att, B, T, F32)
        outs.append((kptv.clone(), att.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
