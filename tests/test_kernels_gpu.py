"""GPU: each floating-point HIP kernel (through the C-ABI) against a plain PyTorch float32/float64
reference of the same op, float32-exact mode at tight tolerance and bf16 mode at the tolerance the
bf16 operands allow (stated per test).  Ragged / edge shapes included."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

F32, BF16 = 0, 1


def dev():
    return torch.device("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def to_t(x, dtype):
    return x if dtype == F32 else x.to(torch.bfloat16)


def tol(dtype):
    # float32 MFMA is an exact fmaf chain: only summation order differs from torch.  bf16 operands
    # carry 2^-9 relative rounding each; accumulation is float32.
    return dict(rtol=2e-5, atol=2e-5) if dtype == F32 else dict(rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N,K", [(1576, 576, 192), (100, 1000, 192), (130, 10, 128), (257, 192, 768), (64, 64, 64)])
def test_gemm_nt_bias(dtype, M, N, K):
    from uvc_amd import ops
    A, B, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3)
    At, Bt = to_t(A, dtype), to_t(B, dtype)
    ref = At.double() @ Bt.double().t() + bias.double()
    for cf32 in ([True] if dtype == F32 else [True, False]):
        Cc = torch.empty(M, N, device=dev(), dtype=torch.float32 if cf32 else torch.bfloat16)
        ops.gemm_nt(At, Bt, Cc, dtype=dtype, epilogue=ops.EPI_BIAS, bias=bias)
        t = tol(dtype) if cf32 else dict(rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(Cc.double(), ref, **t)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_gemm_nt_f32_source_and_epilogues(dtype):
    from uvc_amd import ops
    M, N, K = 333, 192, 768
    A, B, bias = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.05), rnd(N, seed=6)
    R, R2 = rnd(M, N, seed=7), rnd(M, N, seed=8)
    gate = torch.tensor([0.3, 0.7], device=dev())
    Bt = to_t(B, dtype)
    Aeff = A if dtype == F32 else A.to(torch.bfloat16)
    lin = Aeff.double() @ Bt.double().t() + bias.double()
    # residual (float32 A source converted while staging)
    Cc = torch.empty(M, N, device=dev())
    ops.gemm_nt(A, Bt, Cc, dtype=dtype, epilogue=ops.EPI_BIAS_RESID, bias=bias, R=R)
    torch.testing.assert_close(Cc.double(), lin + R.double(), **tol(dtype))
    ops.gemm_nt(A, Bt, Cc, dtype=dtype, epilogue=ops.EPI_BIAS_RESID_GATE, bias=bias, R=R, R2=R2, gate=gate)
    torch.testing.assert_close(Cc.double(), 0.7 * (lin + R.double()) + 0.3 * R2.double(), **tol(dtype))
    # GELU pair
    T = ops.tdtype(dtype)
    Ca, Cu = torch.empty(M, N, device=dev(), dtype=T), torch.empty(M, N, device=dev(), dtype=T)
    ops.gemm_nt(to_t(A, dtype), Bt, Ca, dtype=dtype, epilogue=ops.EPI_BIAS_GELU, bias=bias, C2=Cu)
    torch.testing.assert_close(Ca.double(), lin, **tol(dtype))
    torch.testing.assert_close(Cu.double(), F.gelu(lin), **tol(dtype))
    # dGELU with device alpha
    aux = to_t(rnd(M, N, seed=9), dtype)
    alpha = torch.tensor([0.6], device=dev())
    Cd = torch.empty(M, N, device=dev(), dtype=T)
    ops.gemm_nt(A, Bt, Cd, dtype=dtype, epilogue=ops.EPI_DGELU, aux=aux, alpha_ptr=alpha)
    x = aux.double().requires_grad_(True)
    F.gelu(x).sum().backward()
    torch.testing.assert_close(Cd.double(), 0.6 * (Aeff.double() @ Bt.double().t()) * x.grad, **tol(dtype))
    # training forward of fc1: C = GELU'(a), C2 = GELU(a); and its backward partner C = alpha*acc * aux
    Cg, Cu2 = torch.empty(M, N, device=dev(), dtype=T), torch.empty(M, N, device=dev(), dtype=T)
    ops.gemm_nt(to_t(A, dtype), Bt, Cg, dtype=dtype, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bias, C2=Cu2)
    xl = lin.clone().requires_grad_(True)
    F.gelu(xl).sum().backward()
    torch.testing.assert_close(Cu2.double(), F.gelu(lin), **tol(dtype))
    torch.testing.assert_close(Cg.double(), xl.grad, **tol(dtype))
    Cm = torch.empty(M, N, device=dev(), dtype=T)
    ops.gemm_nt(A, Bt, Cm, dtype=dtype, epilogue=ops.EPI_MUL_AUX, aux=aux, alpha_ptr=alpha)
    torch.testing.assert_close(Cm.double(), 0.6 * (Aeff.double() @ Bt.double().t()) * aux.double(), **tol(dtype))


@pytest.mark.parametrize("K,N", [(192, 192), (192, 576), (192, 768), (128, 128), (128, 512), (384, 384), (384, 1152), (384, 1536)])
def test_gemm_nt_streaming_kernel_matches_generic(K, N):
    """The weights-stationary streaming kernel (M >= 4096, K in {128,192}, N % 64 == 0; K = 384 with N % 192 == 0 and bf16 A) must agree with the
    generic tiled kernel bit-for-bit up to fp32 summation order, on every epilogue, incl. a ragged last M tile."""
    from uvc_amd import ops
    M = 4096 + 37
    A32, W = rnd(M, K, seed=81), rnd(N, K, seed=82, scale=0.05).to(torch.bfloat16)
    A = A32.to(torch.bfloat16)
    bias, R, R2 = rnd(N, seed=83), rnd(M, N, seed=84), rnd(M, N, seed=85)
    aux = rnd(M, N, seed=86).to(torch.bfloat16)
    gate = torch.tensor([0.25, 0.75], device=dev())
    alpha = torch.tensor([0.6], device=dev())
    ref = A.double() @ W.double().t()
    cases = [
        (A, torch.bfloat16, dict(epilogue=ops.EPI_BIAS, bias=bias)),
        (A, torch.float32, dict(epilogue=ops.EPI_BIAS_RESID, bias=bias, R=R)),
        (A, torch.float32, dict(epilogue=ops.EPI_BIAS_RESID_GATE, bias=bias, R=R, R2=R2, gate=gate)),
        (A32, torch.bfloat16, dict(epilogue=ops.EPI_DGELU, aux=aux, alpha_ptr=alpha)),
        (A32, torch.bfloat16, dict(epilogue=ops.EPI_NONE)),
        (A, torch.bfloat16, dict(epilogue=ops.EPI_MUL_AUX, aux=aux, alpha_ptr=alpha)),
    ]
    for Ain, cdt, kw in cases:
        C1, C2 = torch.empty(M, N, device=dev(), dtype=cdt), torch.empty(M, N, device=dev(), dtype=cdt)
        ops.gemm_nt(Ain, W, C1, dtype=BF16, **kw)
        ops.gemm_nt(Ain, W, C2, dtype=BF16, force_generic=True, **kw)
        torch.testing.assert_close(C1.float(), C2.float(), rtol=1e-2 if cdt == torch.bfloat16 else 1e-5, atol=1e-2 if cdt == torch.bfloat16 else 1e-4)
    Ca, Cu = torch.empty(M, N, device=dev(), dtype=torch.bfloat16), torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, W, Ca, dtype=BF16, epilogue=ops.EPI_BIAS_GELU, bias=bias, C2=Cu)
    torch.testing.assert_close(Ca.double(), ref + bias.double(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(Cu.double(), F.gelu(ref + bias.double()), rtol=2e-2, atol=2e-2)
    Cg = torch.empty_like(Ca)
    ops.gemm_nt(A, W, Cg, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bias, C2=Cu)
    xl = (ref + bias.double()).requires_grad_(True)
    F.gelu(xl).sum().backward()
    torch.testing.assert_close(Cg.double(), xl.grad, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(Cu.double(), F.gelu(ref + bias.double()), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M", [4096 + 21, 4096 + 16, 16 * 1031])
@pytest.mark.parametrize("K", [768, 576])
def test_gemm_nt_ksplit_streaming_kernel_matches_generic(K, M):
    """The column-sliced weights-stationary kernel (N == 192, K in {576, 768}, M >= 4096; float32 and bf16 outputs) against the generic kernel.
    M % 16 == 0 with K = 768 and a residual epilogue selects the LDS-DMA ring variant (fc2 of DeiT-Tiny); a ragged M the register-staged one."""
    from uvc_amd import ops
    N = 192
    A, W = rnd(M, K, seed=91).to(torch.bfloat16), rnd(N, K, seed=92, scale=0.03).to(torch.bfloat16)
    bias, R, R2 = rnd(N, seed=93), rnd(M, N, seed=94), rnd(M, N, seed=95)
    gate = torch.tensor([0.25, 0.75], device=dev())
    ref = A.double() @ W.double().t()
    for cdt, kw in ((torch.float32, dict(epilogue=ops.EPI_BIAS_RESID_GATE, bias=bias, R=R, R2=R2, gate=gate)),
                    (torch.float32, dict(epilogue=ops.EPI_BIAS_RESID, bias=bias, R=R)),
                    (torch.float32, dict(epilogue=ops.EPI_NONE)), (torch.float32, dict(epilogue=ops.EPI_BIAS, bias=bias)),
                    (torch.bfloat16, dict(epilogue=ops.EPI_NONE)), (torch.bfloat16, dict(epilogue=ops.EPI_BIAS, bias=bias))):
        C1, C2 = torch.empty(M, N, device=dev(), dtype=cdt), torch.empty(M, N, device=dev(), dtype=cdt)
        for _ in range(3):                      # repeated: a packed-math hazard once showed up as run-to-run differences
            ops.gemm_nt(A, W, C1, dtype=BF16, **kw)
        ops.gemm_nt(A, W, C2, dtype=BF16, force_generic=True, **kw)
        torch.testing.assert_close(C1.float(), C2.float(), rtol=1e-2 if cdt == torch.bfloat16 else 1e-5, atol=1e-2 if cdt == torch.bfloat16 else 2e-4)
    C1 = torch.empty(M, N, device=dev())
    ops.gemm_nt(A, W, C1, dtype=BF16, epilogue=ops.EPI_BIAS_RESID, bias=bias, R=R)
    torch.testing.assert_close(C1.double(), ref + bias.double() + R.double(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("M,N1,N2", [(1576, 576, 192), (403, 192, 768), (3001, 768, 192), (64, 1000, 192), (5000, 128, 128), (40, 8, 16), (4133, 192, 192),
                                     (2500, 3072, 768), (1210, 768, 3072), (999, 2304, 768), (25216, 768, 768), (70, 1024, 768),
                                     (3000, 384, 1536), (2999, 1536, 384), (5001, 384, 1152), (3000, 1152, 384), (2000, 384, 384)])
                                     # the last five: 256 x 256 tiles (DeiT-Base; r4: k_gemm_tn8p -- splits that are not whole 64-row k-steps, a split of one
                                     # k-step, the full batch-128 row count with 28 splits of 9 tiles)
def test_gemm_tn(dtype, M, N1, N2):
    from uvc_amd import ops
    A, B = rnd(M, N1, seed=11), rnd(M, N2, seed=12)
    Cc = rnd(N1, N2, seed=13)
    C0 = Cc.clone()
    ws = torch.empty(ops.gemm_tn_workspace_bytes(M, N1, N2) // 4, device=dev())
    alpha = torch.tensor([0.5], device=dev())
    for a_f32 in ([True] if dtype == F32 else [True, False]):
        At = A if a_f32 else A.to(torch.bfloat16)
        Bt = to_t(B, dtype)
        Aeff = A.to(torch.bfloat16) if dtype == BF16 else A
        Cc.copy_(C0)
        cs = torch.ones(N1, device=dev())
        ops.gemm_tn(At, Bt, Cc, ws, dtype=dtype, alpha=2.0, alpha_ptr=alpha, beta=1.0, colsum_out=cs)
        ref = C0.double() + Aeff.double().t() @ Bt.double()
        # (float32: the rounding of an M-term sum of O(1) products grows with M -- the bound is the 4 k-row one scaled)
        t = dict(rtol=2e-5, atol=2e-4 * max(1.0, M / 4000.0)) if dtype == F32 else dict(rtol=2e-2, atol=5e-2)
        torch.testing.assert_close(Cc.double(), ref, **t)
        torch.testing.assert_close(cs.double(), 1 + Aeff.double().sum(0), rtol=1e-4, atol=2e-3 * max(1.0, M / 4000.0))   # fused bias gradient
    # determinism: two runs bit-identical
    C1, C2 = torch.empty(N1, N2, device=dev()), torch.empty(N1, N2, device=dev())
    ops.gemm_tn(to_t(A, dtype), to_t(B, dtype), C1, ws, dtype=dtype)
    ops.gemm_tn(to_t(A, dtype), to_t(B, dtype), C2, ws, dtype=dtype)
    assert torch.equal(C1, C2)
    if dtype == BF16:
        # variant 1: the two-group schedule (k_gemm_tn8p) on the 192-wide tiles too: same k order per accumulator as the ring kernel -> same bits
        C3, cs0, cs1 = torch.empty(N1, N2, device=dev()), torch.zeros(N1, device=dev()), torch.zeros(N1, device=dev())
        ops.gemm_tn(to_t(A, dtype), to_t(B, dtype), C1, ws, dtype=dtype, colsum_out=cs0, variant=2)
        ops.gemm_tn(to_t(A, dtype), to_t(B, dtype), C3, ws, dtype=dtype, colsum_out=cs1, variant=1)
        assert torch.equal(C1, C3) and torch.equal(cs0, cs1)
        ops.gemm_tn(to_t(A, dtype), to_t(B, dtype), C3, ws, dtype=dtype, colsum_out=cs1, variant=0)
        assert torch.equal(C1, C3) and torch.equal(cs0, cs1)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (3, 198, 2), (2, 17, 2), (1, 64, 1), (2, 5, 1)])
def test_attention_fwd_bwd(dtype, B, N, H):
    from uvc_amd import ops
    D = H * 64
    T = ops.tdtype(dtype)
    qkv = to_t(rnd(B, N, 3 * D, seed=21), dtype)
    dout = to_t(rnd(B, N, D, seed=22), dtype)
    o = torch.empty(B, N, D, device=dev(), dtype=T)
    lse = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv, o, lse, B, N, H, dtype)
    x = qkv.double().requires_grad_(True)
    q, k, v = x.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * 64 ** -0.5
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, D)
    t = dict(rtol=1e-4, atol=1e-5) if dtype == F32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(o.double(), ref, **t)
    torch.testing.assert_close(lse.double(), torch.logsumexp(s, -1), rtol=1e-4 if dtype == F32 else 2e-2, atol=1e-4 if dtype == F32 else 2e-2)
    dqkv = torch.full((B, N, 3 * D), float("nan"), device=dev(), dtype=T)
    delta = torch.empty(B, H, N, device=dev())
    ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, dtype)
    ref.backward(dout.double())
    tb = dict(rtol=2e-4, atol=2e-5) if dtype == F32 else dict(rtol=5e-2, atol=6e-2)
    torch.testing.assert_close(dqkv.double(), x.grad, **tb)


@pytest.mark.parametrize("B,N,H,grid,bias", [(2, 197, 3, 0, True), (7, 197, 3, 3, True), (5, 198, 3, 2, False), (3, 193, 3, 0, True), (2, 208, 3, 1, True),
                                             (3, 197, 6, 0, True), (5, 197, 6, 4, False), (2, 198, 6, 3, True)])
def test_qkv_attention_forward_fused_equals_the_two_kernels(B, N, H, grid, bias):
    """uvc_qkv_attention_fwd (the qkv Linear + attention forward as one persistent kernel, r5) against uvc_gemm_nt (bias epilogue) + uvc_attention_fwd on the
    same inputs: o, lse and the stored qkv BIT for bit (same accumulation chain, same rounding points), with and without storing qkv, several images per
    workgroup (grid < B), both ends of the N range; and against float64."""
    from uvc_amd import ops
    D = 64 * H                              # D = 192: one work item per image; D = 384: two groups of three heads per image, 32-row weight chunks
    assert ops.qkv_attention_supported(B, N, H, D, BF16) and not ops.qkv_attention_supported(B, 64, H, D, BF16) and not ops.qkv_attention_supported(B, N, 12, 768, BF16)
    h = to_t(rnd(B * N, D, seed=41), BF16)
    W = to_t(rnd(3 * D, D, seed=42) * 0.08, BF16)
    bvec = rnd(3 * D, seed=43) * 0.1 if bias else None
    qkv0 = torch.empty(B * N, 3 * D, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(h, W, qkv0, dtype=BF16, epilogue=ops.EPI_BIAS if bias else ops.EPI_NONE, bias=bvec)
    o0 = torch.empty(B, N, D, device=dev(), dtype=torch.bfloat16)
    lse0 = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv0.view(B, N, 3 * D), o0, lse0, B, N, H, BF16)
    for store in (True, False):
        qkv1 = torch.full((B, N, 3 * D), float("nan"), device=dev(), dtype=torch.bfloat16) if store else None
        o1 = torch.full((B, N, D), float("nan"), device=dev(), dtype=torch.bfloat16)
        lse1 = torch.full((B, H, N), float("nan"), device=dev())
        ops.qkv_attention_fwd(h, W, bvec, o1, lse1, B, N, H, BF16, qkv=qkv1, grid=grid)
        if store:
            assert torch.equal(qkv1.view(B * N, 3 * D), qkv0)
        assert torch.equal(o1, o0) and torch.equal(lse1, lse0)
    x = (h.double() @ W.double().t() + (bvec.double() if bias else 0)).view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s_ = (x[0] @ x[1].transpose(-2, -1)) * 64 ** -0.5
    ref = (s_.softmax(-1) @ x[2]).transpose(1, 2).reshape(B, N, D)
    torch.testing.assert_close(o1.double(), ref, rtol=4e-2, atol=4e-2)


@pytest.mark.parametrize("B,N,H,grid", [(2, 197, 3, 0), (7, 197, 3, 4), (3, 198, 2, 2), (5, 193, 1, 2), (2, 200, 2, 1)])
def test_attention_backward_one_pass(B, N, H, grid):
    """The one-pass persistent backward (uvc_attn_args.variant 2: k_attn_bwd_one, r5) against float64 and against the dq + dk/dv pair
    (variant 1); several heads per workgroup (grid < B * H, uneven head counts per workgroup), both ends of its N range; bit-identical
    repeat; variant 0 is the pair below 1024 heads (test_fullsize_gpu.py runs the step at 1536)."""
    from uvc_amd import ops
    D = H * 64
    qkv = to_t(rnd(B, N, 3 * D, seed=31), BF16)
    dout = to_t(rnd(B, N, D, seed=32), BF16)
    o = torch.empty(B, N, D, device=dev(), dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv, o, lse, B, N, H, BF16)
    x = qkv.double().requires_grad_(True)
    q, k, v = x.reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * 64 ** -0.5
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, D)
    ref.backward(dout.double())
    outs = []
    for variant, gr in ((1, 0), (2, grid), (2, grid), (0, 0)):
        dqkv = torch.full((B, N, 3 * D), float("nan"), device=dev(), dtype=torch.bfloat16)
        delta = torch.full((B, H, N), float("nan"), device=dev())
        ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, BF16, variant=variant, grid=gr)
        torch.testing.assert_close(dqkv.double(), x.grad, rtol=5e-2, atol=6e-2)
        torch.testing.assert_close(delta.double(), (dout.double() * o.double()).view(B, N, H, 64).sum(-1).permute(0, 2, 1), rtol=1e-3, atol=1e-3)
        outs.append(dqkv)
    assert torch.equal(outs[1], outs[2])                       # repeat: same bits
    assert torch.equal(outs[0], outs[3])                       # variant 0: below 1024 heads the pair (a persistent workgroup wants >= 4 heads)
    # closer to the pair than the bf16 tolerance against float64: the two differ by the rounding of P / dS only
    e_pair = (outs[0].double() - x.grad).abs().max().item()
    e_one = (outs[1].double() - x.grad).abs().max().item()
    assert e_one <= 1.5 * e_pair + 1e-3, (e_one, e_pair)
    err = lambda t: ((t.double() - x.grad).norm() / x.grad.norm()).item()
    assert err(outs[1]) <= 1.1 * err(outs[0]) + 1e-4, (err(outs[1]), err(outs[0]))


def test_attention_backward_one_pass_refuses_other_shapes():
    from uvc_amd import ops
    from uvc_amd._lib import UvcHipError
    B, N, H = 1, 64, 1
    qkv = to_t(rnd(B, N, 3 * 64, seed=33), BF16)
    o = torch.zeros(B, N, 64, device=dev(), dtype=torch.bfloat16)
    lse = torch.zeros(B, H, N, device=dev())
    with pytest.raises(UvcHipError):
        ops.attention_bwd(qkv, o, lse, o.clone(), torch.empty_like(qkv), torch.empty_like(lse), B, N, H, BF16, variant=2)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attention_backward_head_keep(dtype):
    """uvc_attention_bwd with head_keep: heads marked 0 get dq = dk = dv = 0 without being computed -- what the full computation gives
    when dout is zero on those heads' columns (Stage-2: masked attn.proj input columns); the other heads are untouched."""
    from uvc_amd import ops
    B, N, H = 3, 197, 3
    T = ops.tdtype(dtype)
    qkv = to_t(rnd(B, N, 3 * H * 64, seed=151), dtype)
    dout = to_t(rnd(B, N, H * 64, seed=152), dtype)
    dout.view(B, N, H, 64)[:, :, 1] = 0                    # head 1 is the pruned one
    o = torch.empty(B, N, H * 64, device=dev(), dtype=T)
    lse = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv, o, lse, B, N, H, dtype)
    keep = torch.tensor([1, 0, 1], device=dev(), dtype=torch.int32)
    outs = []
    for hk in (None, keep):
        dqkv = torch.full((B, N, 3 * H * 64), float("nan"), device=dev(), dtype=T)
        delta = torch.empty(B, H, N, device=dev())
        ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, dtype, head_keep=hk)
        outs.append(dqkv)
    assert torch.equal(outs[0], outs[1])
    assert float(outs[1].view(B, N, 3, H, 64)[:, :, :, 1].abs().max()) == 0.0
    assert float(outs[1].view(B, N, 3, H, 64)[:, :, :, 0].abs().max()) > 0.0


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_attention_forward_head_keep(dtype):
    """Inference-only head skipping: heads marked 0 get an all-zero output slice, the others are untouched."""
    from uvc_amd import ops
    B, N, H = 3, 197, 3
    T = ops.tdtype(dtype)
    qkv = to_t(rnd(B, N, 3 * H * 64, seed=141), dtype)
    o_full, o_skip = torch.empty(B, N, H * 64, device=dev(), dtype=T), torch.full((B, N, H * 64), float("nan"), device=dev(), dtype=T)
    lse = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv, o_full, lse, B, N, H, dtype)
    keep = torch.tensor([1, 0, 1], device=dev(), dtype=torch.int32)
    ops.attention_fwd(qkv, o_skip, lse, B, N, H, dtype, head_keep=keep)
    o_full, o_skip = o_full.view(B, N, H, 64), o_skip.view(B, N, H, 64)
    assert torch.equal(o_skip[:, :, 0], o_full[:, :, 0]) and torch.equal(o_skip[:, :, 2], o_full[:, :, 2])
    assert float(o_skip[:, :, 1].float().abs().sum()) == 0.0


def test_attention_softmax_extremes_f32():
    """Rows dominated by one key (large logits) and identical keys: no NaN/inf, matches float64."""
    from uvc_amd import ops
    B, N, H = 1, 40, 1
    qkv = rnd(B, N, 192, seed=23)
    qkv[0, 3, :64] *= 40.0
    qkv[0, :, 64:128] = qkv[0, 0, 64:128]          # all keys identical -> uniform attention
    qkv[0, 5, 64:128] *= 30.0
    o = torch.empty(B, N, 64, device=dev()); lse = torch.empty(B, H, N, device=dev())
    ops.attention_fwd(qkv, o, lse, B, N, H, F32)
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-2, -1)) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(B, N, 64)
    assert torch.isfinite(o).all()
    torch.testing.assert_close(o.double(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dtype", [F32, BF16])
@pytest.mark.parametrize("rows,D", [(1576, 192), (333, 384), (130, 768), (7, 128)])
def test_layernorm_fwd_bwd(dtype, rows, D):
    from uvc_amd import ops
    T = ops.tdtype(dtype)
    x, gamma, beta = rnd(rows, D, seed=31) * 2 + 0.5, rnd(D, seed=32) * 0.2 + 1.0, rnd(D, seed=33) * 0.1
    y = torch.empty(rows, D, device=dev(), dtype=T)
    mean, rstd = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, dtype)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.layer_norm(xd, (D,), gd, bd, 1e-6)
    torch.testing.assert_close(y.double(), ref, **(dict(rtol=1e-5, atol=1e-5) if dtype == F32 else dict(rtol=1e-2, atol=1e-2)))
    dy = to_t(rnd(rows, D, seed=34), dtype)
    add1, add2 = rnd(rows, D, seed=35), rnd(rows, D, seed=36)
    a1, a2 = torch.tensor([0.7], device=dev()), torch.tensor([0.3], device=dev())
    dx = torch.empty(rows, D, device=dev())
    partial = torch.empty(ops.layernorm_bwd_blocks(rows) * (2 * D + 2), device=dev())
    dg, db, dots = torch.ones(D, device=dev()), torch.ones(D, device=dev()), torch.empty(2, device=dev())
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, partial, dg, db, rows, D, dtype, add1=add1, a1=a1, add2=add2, a2=a2,
                      dots=dots, beta_acc=1.0)
    ref.backward(dy.double())
    dxr = xd.grad + 0.7 * add1.double() + 0.3 * add2.double()
    torch.testing.assert_close(dx.double(), dxr, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dg.double(), 1 + gd.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.double(), 1 + bd.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dots.double(), torch.stack([(dxr * x.double()).sum(), (add2.double() * x.double()).sum()]),
                               rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("rows,D", [(1576, 192), (333, 384)])
def test_layernorm_bwd_bf16_gradient_stream(rows, D):
    """The throughput mode keeps dL/dx (dx, add1, add2) in bf16: loads are widened, all sums and the gate dots
    stay float32, only the stored dx is rounded."""
    from uvc_amd import ops
    x, gamma, beta = rnd(rows, D, seed=31) * 2 + 0.5, rnd(D, seed=32) * 0.2 + 1.0, rnd(D, seed=33) * 0.1
    y = torch.empty(rows, D, device=dev(), dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, BF16)
    dy = to_t(rnd(rows, D, seed=34), BF16)
    add1, add2 = rnd(rows, D, seed=35).bfloat16(), rnd(rows, D, seed=36).bfloat16()
    a1, a2 = torch.tensor([0.7], device=dev()), torch.tensor([0.3], device=dev())
    dx = torch.empty(rows, D, device=dev(), dtype=torch.bfloat16)
    partial = torch.empty(ops.layernorm_bwd_blocks(rows) * (2 * D + 2), device=dev())
    dg, db, dots = torch.zeros(D, device=dev()), torch.zeros(D, device=dev()), torch.empty(2, device=dev())
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, partial, dg, db, rows, D, BF16, add1=add1, a1=a1, add2=add2, a2=a2, dots=dots)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    F.layer_norm(xd, (D,), gd, bd, 1e-6).backward(dy.double())
    dxr = xd.grad + 0.7 * add1.double() + 0.3 * add2.double()
    torch.testing.assert_close(dx.double(), dxr, rtol=8e-3, atol=8e-3)              # one bf16 rounding of the result
    torch.testing.assert_close(dg.double(), gd.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dots.double(), torch.stack([(dxr * x.double()).sum(), (add2.double() * x.double()).sum()]),
                               rtol=1e-4, atol=1e-2)                                # from the unrounded values
    with pytest.raises(Exception):
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, partial, dg, db, rows, D, BF16, add1=add1.float())


def test_layernorm_bwd_deferred_batched_reduction():
    """defer_reduce leaves the per-block partials in a private region; uvc_layernorm_bwd_reduce_batch finishes several calls
    in one launch and must give exactly the results of the immediate path (same fixed summation order is not required,
    equality to 1e-6 is)."""
    import ctypes as C
    from uvc_amd import _lib as L, ops
    rows, D = 1576, 192
    outs = []
    items = (L.uvc_ln_reduce_item * 2)()
    keep = []
    for k in range(2):
        x, gamma = rnd(rows, D, seed=131 + k) * 2, rnd(D, seed=133 + k) * 0.2 + 1.0
        y = torch.empty(rows, D, device=dev(), dtype=torch.bfloat16)
        mean, rstd = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
        ops.layernorm_fwd(x, gamma, rnd(D, seed=135), y, mean, rstd, rows, D, BF16)
        dy = to_t(rnd(rows, D, seed=137 + k), BF16)
        dx = torch.empty(rows, D, device=dev())
        part = torch.empty(ops.layernorm_bwd_blocks(rows) * (2 * D + 2), device=dev())
        dg, db, dots = torch.zeros(D, device=dev()), torch.zeros(D, device=dev()), torch.zeros(2, device=dev())
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, part, dg, db, rows, D, BF16, dots=dots)
        dg2, db2, dots2 = torch.zeros(D, device=dev()), torch.zeros(D, device=dev()), torch.zeros(2, device=dev())
        part2 = torch.empty_like(part)
        a = ops._ln_args(x, gamma, None, rows, D, BF16, 1, None)
        a.mean, a.rstd, a.dy, a.dx = L.ptr(mean), L.ptr(rstd), L.ptr(dy), L.ptr(dx)
        a.partial, a.dgamma, a.dbeta, a.dots = L.ptr(part2), L.ptr(dg2), L.ptr(db2), L.ptr(dots2)
        a.dy_is_f32, a.defer_reduce = 0, 1
        L.check(L.lib().uvc_layernorm_bwd(C.byref(a), L.cur_stream()), "uvc_layernorm_bwd")
        assert float(dg2.abs().sum()) == 0.0                       # nothing reduced yet
        items[k].partial, items[k].dgamma, items[k].dbeta, items[k].dots = L.ptr(part2), L.ptr(dg2), L.ptr(db2), L.ptr(dots2)
        items[k].nblocks = int(L.lib().uvc_layernorm_bwd_nblocks(rows))
        outs.append((dg, db, dots, dg2, db2, dots2))
        keep.append((x, gamma, y, mean, rstd, dy, dx, part, part2))
    L.check(L.lib().uvc_layernorm_bwd_reduce_batch(items, 2, D, 0.0, L.cur_stream()), "uvc_layernorm_bwd_reduce_batch")
    for dg, db, dots, dg2, db2, dots2 in outs:
        torch.testing.assert_close(dg2, dg, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(db2, db, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dots2, dots, rtol=1e-5, atol=1e-3)


def test_layernorm_class_token_rows():
    """Final norm on the class/dist-token rows only (model_distilled.py:507-508): strided rows."""
    from uvc_amd import ops
    B, N, D = 5, 198, 192
    x, gamma, beta = rnd(B, N, D, seed=37), rnd(D, seed=38) + 1, rnd(D, seed=39)
    for rpg in (1, 2):
        rows = B * rpg
        y = torch.empty(rows, D, device=dev()); mean = torch.empty(rows, device=dev()); rstd = torch.empty(rows, device=dev())
        ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, F32, rows_per_group=rpg, group_stride=N * D)
        ref = F.layer_norm(x[:, :rpg].double(), (D,), gamma.double(), beta.double(), 1e-6).reshape(rows, D)
        torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=1e-5)
        dy = rnd(rows, D, seed=40)
        dx = torch.zeros(B, N, D, device=dev())
        partial = torch.empty(ops.layernorm_bwd_blocks(rows) * (2 * D + 2), device=dev())
        dg, db, dots = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.empty(2, device=dev())
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, partial, dg, db, rows, D, F32, dots=dots, rows_per_group=rpg,
                          group_stride=N * D)
        xd = x.double().requires_grad_(True)
        F.layer_norm(xd[:, :rpg], (D,), gamma.double(), beta.double(), 1e-6).backward(dy.double().reshape(B, rpg, D))
        torch.testing.assert_close(dx.double(), xd.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,C,same", [(8, 1000, True), (5, 10, True), (6, 1000, False)])
def test_distill_loss(B, C, same):
    from uvc_amd import ops
    o, y, t = rnd(B, C, seed=41) * 2, F.softmax(rnd(B, C, seed=42) * 2, -1), rnd(B, C, seed=43) * 2
    okd = o if same else rnd(B, C, seed=44)
    loss = torch.empty(1, device=dev()); d_o = torch.empty(B, C, device=dev())
    d_k = d_o if same else torch.empty(B, C, device=dev())
    scratch = torch.empty(B, device=dev())
    alpha, tau = 0.1, 2.0
    ops.distill_loss(o, okd, y, t, loss, d_o, d_k, scratch, alpha, tau, kind=1)
    od = o.double().requires_grad_(True)
    kd_in = od if same else okd.double().requires_grad_(True)
    base = torch.sum(-y.double() * F.log_softmax(od, -1), -1).mean()
    kd = F.kl_div(F.log_softmax(kd_in / tau, 1), F.log_softmax(t.double() / tau, 1), reduction="sum", log_target=True) * tau * tau / (B * C)
    ref = base * (1 - alpha) + kd * alpha
    ref.backward()
    torch.testing.assert_close(loss.double()[0], ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d_o.double(), od.grad, rtol=1e-4, atol=1e-7)
    if not same:
        torch.testing.assert_close(d_k.double(), kd_in.grad, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("B,C,same", [(8, 1000, True), (5, 1000, False), (4, 16, False)])
def test_distill_loss_hard(B, C, same):
    """distillation_type='hard' (utils/losses.py:61-62, the argparse default joint_train.py:781): CE against the teacher's
    argmax class; a tie in the teacher row goes to the first class (torch.argmax)."""
    from uvc_amd import ops
    o, y, t = rnd(B, C, seed=45) * 2, F.softmax(rnd(B, C, seed=46) * 2, -1), rnd(B, C, seed=47) * 2
    t[0, 7] = t[0, 3] = t[0].max() + 1.0                   # exact tie: class 3 wins
    okd = o if same else rnd(B, C, seed=48)
    loss = torch.empty(1, device=dev()); d_o = torch.empty(B, C, device=dev())
    d_k = d_o if same else torch.empty(B, C, device=dev())
    scratch = torch.empty(B, device=dev())
    alpha = 0.5
    ops.distill_loss(o, okd, y, t, loss, d_o, d_k, scratch, alpha, 3.0, kind=2)
    od = o.double().requires_grad_(True)
    kd_in = od if same else okd.double().requires_grad_(True)
    base = torch.sum(-y.double() * F.log_softmax(od, -1), -1).mean()
    target = t.argmax(dim=1)
    assert int(target[0]) == 3
    ref = base * (1 - alpha) + F.cross_entropy(kd_in, target) * alpha
    ref.backward()
    torch.testing.assert_close(loss.double()[0], ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d_o.double(), od.grad, rtol=1e-4, atol=1e-7)
    if not same:
        torch.testing.assert_close(d_k.double(), kd_in.grad, rtol=1e-4, atol=1e-8)


def test_clip_adamw_matches_torch():
    from uvc_amd import ops
    n = 100003
    p0, g0 = rnd(n, seed=51), rnd(n, seed=52) * 0.01
    P = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([P], lr=3e-4, weight_decay=0.05)
    p, m, v = p0.clone(), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    shadow = torch.empty(n, device=dev(), dtype=torch.bfloat16)
    partial, sq, gn = torch.empty(1024, device=dev()), torch.empty(2, device=dev()), torch.empty(1, device=dev())
    for step in range(1, 4):
        g = g0 * step
        P.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([P], 1.0)
        opt.step()
        ops.grad_sqnorm(g, partial, sq)
        torch.testing.assert_close(sq[1], tn, rtol=1e-5, atol=0)            # the norm clip_grad_norm_ returns, left by the reduction itself
        ops.adamw_step(p, g, m, v, sq, lr=3e-4, step=step, p_shadow=shadow, gnorm_out=gn)
        torch.testing.assert_close(gn[0], tn, rtol=1e-5, atol=0)
        torch.testing.assert_close(p, P.data, rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(shadow.float(), P.data.to(torch.bfloat16).float(), rtol=1e-2, atol=1e-3)
    gg = g0.clone()
    ops.scale_by_clip(gg, sq, 1.0)
    torch.testing.assert_close(gg, g0 * min(1.0, 1.0 / (float(sq[0].sqrt()) + 1e-6)), rtol=1e-6, atol=0)


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_patchify_assemble_colsum_transpose(dtype):
    from uvc_amd import ops
    B, S, P, D, ntok = 3, 64, 16, 128, 2
    T = ops.tdtype(dtype)
    x = rnd(B, 3, S, S, seed=61)
    npatch = (S // P) ** 2
    out = torch.empty(B * npatch, 3 * P * P, device=dev(), dtype=T)
    ops.patchify(x, out, P, dtype)
    ref = F.unfold(x, P, stride=P).transpose(1, 2).reshape(B * npatch, -1)
    torch.testing.assert_close(out.float(), ref.to(T).float(), rtol=0, atol=0)
    pe, cls, dist, pos = rnd(B, npatch, D, seed=62), rnd(D, seed=63), rnd(D, seed=64), rnd(npatch + ntok, D, seed=65)
    mask = (rnd(B, npatch, seed=66) > 0).float()
    tok = torch.empty(B, npatch + ntok, D, device=dev())
    ops.assemble_tokens(pe, cls, dist, pos, mask, tok, B, npatch, D, ntok)
    reft = torch.cat([cls.expand(B, 1, D), dist.expand(B, 1, D), pe * mask.unsqueeze(-1)], 1) + pos
    torch.testing.assert_close(tok, reft, rtol=0, atol=0)
    dtok = rnd(B, npatch + ntok, D, seed=67)
    dpe = torch.empty(B, npatch, D, device=dev(), dtype=T)
    dpos, dcls, ddist, dmask = (torch.empty(npatch + ntok, D, device=dev()), torch.empty(D, device=dev()),
                                torch.empty(D, device=dev()), torch.empty(B, npatch, device=dev()))
    ops.assemble_tokens_bwd(dtok, pe, mask, dpe, dpos, dcls, ddist, dmask, B, npatch, D, ntok, dtype)
    torch.testing.assert_close(dpe.float(), (dtok[:, ntok:] * mask.unsqueeze(-1)).to(T).float(), rtol=0, atol=0)
    torch.testing.assert_close(dpos, dtok.sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dcls, dtok[:, 0].sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ddist, dtok[:, 1].sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dmask, (dtok[:, ntok:] * pe).sum(-1), rtol=1e-4, atol=1e-4)
    if dtype == BF16:          # the backward's bf16 gradient stream as input: same sums over the rounded values
        d16 = dtok.bfloat16()
        ops.assemble_tokens_bwd(d16, pe, mask, dpe, dpos, dcls, ddist, dmask, B, npatch, D, ntok, dtype)
        torch.testing.assert_close(dpe.float(), d16[:, ntok:].float() * mask.unsqueeze(-1), rtol=0, atol=0)
        torch.testing.assert_close(dpos, d16.float().sum(0), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dcls, d16[:, 0].float().sum(0), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dmask, (d16[:, ntok:].float() * pe).sum(-1), rtol=1e-4, atol=1e-4)
    for M, N in ((1576, 576), (300, 10), (5, 1000)):
        X = to_t(rnd(M, N, seed=68), dtype)
        partial = torch.empty(ops.colsum_blocks(M) * N, device=dev())
        o = torch.ones(N, device=dev())
        ops.colsum(X, partial, o, dtype, alpha=2.0, beta=1.0)
        torch.testing.assert_close(o.double(), 1 + 2 * X.double().sum(0), rtol=1e-4, atol=1e-3)
    W = rnd(192, 576, seed=69)
    wc, wt = torch.empty(192, 576, device=dev(), dtype=T), torch.empty(576, 192, device=dev(), dtype=T)
    ops.cast_transpose(W, 192, 576, wc, wt, dtype)
    assert torch.equal(wc, W.to(T)) and torch.equal(wt, W.to(T).t().contiguous())


def test_gate_distrib_and_grad():
    from uvc_amd import ops
    Lb = 12
    g = rnd(Lb, 2, seed=71)
    e = torch.empty(Lb, 2, device=dev()).exponential_()
    d = torch.empty(Lb, 2, device=dev())
    ops.gate_distrib(g, e, d, Lb, 1, 0.1)
    ref = ((g + (-e.log())) / 0.5).softmax(-1)
    torch.testing.assert_close(d, ref, rtol=1e-5, atol=1e-6)
    ops.gate_distrib(g, e, d, Lb, 0, 0.1)
    assert torch.equal(d, torch.full_like(d, 0.5))
    ops.gate_distrib(g, e, d, Lb, 2, 0.1)
    t = g[:, 1] ** 2
    torch.testing.assert_close(d[:, 1], t / (t + 0.1), rtol=1e-6, atol=1e-7)
    # gradient identities: out = d1*x2 + d0*x ; A = <gA,out>, B = <gA,x>
    for mode in (1, 2):
        gd = g.double().requires_grad_(True)
        if mode == 1:
            dd = ((gd + (-e.double().log())) / 0.5).softmax(-1)
        else:
            d1 = gd[:, 1] ** 2 / (gd[:, 1] ** 2 + 0.1)
            dd = torch.stack([1 - d1, d1], 1)
        gx2, gx = rnd(Lb, seed=72).double(), rnd(Lb, seed=73).double()       # <gA,x2>, <gA,x> per block
        (dd[:, 1] * gx2 + dd[:, 0] * gx).sum().backward()
        A = (dd[:, 1] * gx2 + dd[:, 0] * gx).detach()
        dots = torch.zeros(Lb + 1, 2, dtype=torch.float64, device=dev())
        dots[1:, 0] = A            # <gA_l, out_l> is left by block l+1's norm1 backward (row l+1, col 0)
        dots[:Lb, 1] = gx          # <gA_l, x_l> by block l's own (row l, col 1)
        dots = dots.float().contiguous()
        ops.gate_distrib(g, e, d, Lb, mode, 0.1)
        dg = torch.empty(Lb, 2, device=dev())
        ops.gate_grad(g, d, dots, dg, Lb, mode, 0.1)
        torch.testing.assert_close(dg.double(), gd.grad, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("M,F_", [(1576, 768), (300, 768), (4096 + 37, 768), (515, 128)])
def test_fused_inference_mlp(M, F_):
    """uvc_mlp_fused_fwd = x + fc2(GELU(fc1(LayerNorm(x)))) with the hidden activation kept in registers, against float64
    math on the same bf16-rounded weights (LayerNorm output and GELU output are rounded to bf16 as in the unfused path),
    and against the unfused kernel sequence."""
    from uvc_amd import ops
    D = 192
    x = rnd(M, D, seed=101) * 1.5 + 0.2
    gamma, beta = rnd(D, seed=102) * 0.2 + 1.0, rnd(D, seed=103) * 0.1
    W1, b1 = rnd(F_, D, seed=104, scale=0.06).bfloat16(), rnd(F_, seed=105) * 0.1
    W2, b2 = rnd(D, F_, seed=106, scale=0.04).bfloat16(), rnd(D, seed=107) * 0.1
    out = torch.full((M, D), float("nan"), device=dev())
    ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out)
    xd = x.double()
    h = F.layer_norm(xd, (D,), gamma.double(), beta.double(), 1e-6).bfloat16().double()
    u = F.gelu(h @ W1.double().t() + b1.double()).bfloat16().double()
    ref = xd + u @ W2.double().t() + b2.double()
    torch.testing.assert_close(out.double(), ref, rtol=2e-3, atol=6e-3)
    # the unfused sequence of the same library
    hb = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(x, gamma, beta, hb, mean, rstd, M, D, BF16)
    ub = torch.empty(M, F_, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(hb, W1, ub, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_OUT, bias=b1)
    o2 = torch.empty(M, D, device=dev())
    ops.gemm_nt(ub, W2, o2, dtype=BF16, epilogue=ops.EPI_BIAS_RESID, bias=b2, R=x)
    torch.testing.assert_close(out, o2, rtol=2e-3, atol=6e-3)
    for _ in range(2):                       # deterministic
        o3 = torch.empty_like(out)
        ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, o3)
        assert torch.equal(o3, out)


@pytest.mark.parametrize("M,F_,gated", [(1576, 768, True), (300, 768, False), (4096 + 37, 768, True), (515, 128, False)])
def test_fused_training_mlp(M, F_, gated):
    """The training form of uvc_mlp_fused_fwd: out = d1 * (x + mlp(LN(x))) + d0 * x_prev in one kernel that also stores what a
    backward reads -- LayerNorm(x) (bf16), mean / rstd, GELU'(a), GELU(a) (bf16 [M, F]) -- against float64 math on the same
    bf16-rounded operands and against the three kernels it would replace; deterministic; rows do not depend on the batch.
    (r2d kernel: 218 us against 187 us for the three kernels at batch 512, so the engine keeps the unfused training forward by default
    -- uvc_vit_io.fused_train_mlp opts in; NOTEBOOK section 11.)"""
    from uvc_amd import ops
    D = 192
    x = rnd(M, D, seed=111) * 1.5 + 0.2
    xp = rnd(M, D, seed=118)
    gamma, beta = rnd(D, seed=112) * 0.2 + 1.0, rnd(D, seed=113) * 0.1
    W1, b1 = rnd(F_, D, seed=114, scale=0.06).bfloat16(), rnd(F_, seed=115) * 0.1
    W2, b2 = rnd(D, F_, seed=116, scale=0.04).bfloat16(), rnd(D, seed=117) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev()) if gated else None

    def run(xs, xps):
        m = xs.shape[0]
        o = torch.full((m, D), float("nan"), device=dev())
        h = torch.empty(m, D, device=dev(), dtype=torch.bfloat16)
        mean, rstd = torch.empty(m, device=dev()), torch.empty(m, device=dev())
        gp, u = torch.empty(m, F_, device=dev(), dtype=torch.bfloat16), torch.empty(m, F_, device=dev(), dtype=torch.bfloat16)
        ops.mlp_fused_fwd(xs, gamma, beta, W1, b1, W2, b2, o, x_prev=xps if gated else None, gate=gate, h=h, mean=mean, rstd=rstd, gp=gp, u=u)
        return o, h, mean, rstd, gp, u

    out, h, mean, rstd, gp, u = run(x, xp)
    xd = x.double()
    hr = F.layer_norm(xd, (D,), gamma.double(), beta.double(), 1e-6)
    torch.testing.assert_close(h.double(), hr, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean.double(), xd.mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rstd.double(), torch.rsqrt(xd.var(1, unbiased=False) + 1e-6), rtol=1e-5, atol=0)
    pre = (h.double() @ W1.double().t() + b1.double()).requires_grad_(True)          # from the kernel's own bf16 LayerNorm output
    ur = F.gelu(pre)
    ur.sum().backward()
    torch.testing.assert_close(u.double(), ur.detach(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(gp.double(), pre.grad, rtol=1e-2, atol=1e-2)
    ref = xd + u.double() @ W2.double().t() + b2.double()                              # from the kernel's own bf16 GELU output
    if gated:
        ref = 0.7 * ref + 0.3 * xp.double()
    torch.testing.assert_close(out.double(), ref, rtol=2e-4, atol=2e-4)
    again = run(x, xp)
    assert all(torch.equal(p, q) for p, q in zip((out, h, mean, rstd, gp, u), again)), "not deterministic"
    part = run(x[:100].contiguous(), xp[:100].contiguous())
    assert all(torch.equal(p[:100], q) for p, q in zip((out, h, mean, rstd, gp, u), part)), "rows depend on the batch"
    o_inf = torch.empty(M, D, device=dev())                                              # the inference form computes the same output
    ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, o_inf, x_prev=xp if gated else None, gate=gate)
    # (the inference form is its own kernel since r2d -- biases as initial accumulators, another summation order: same math, not bits)
    torch.testing.assert_close(o_inf, out, rtol=2e-3, atol=6e-3)


@pytest.mark.parametrize("M,gated,train", [(1576, False, False), (4096 + 37, True, True), (300, False, True), (515, True, False)])
def test_fused_mlp_writes_the_next_blocks_norm1(M, gated, train):
    """uvc_mlp_fused_fwd with next_h: the rows it produces leave a second time as LayerNorm(out; next_gamma, next_beta) in bf16 (the
    next block's norm1, model_distilled.py:241) with their statistics -- against float64 LayerNorm of the kernel's own output rows
    and against uvc_layernorm_fwd on them; `out` itself must not change by a bit when the extra output is requested."""
    from uvc_amd import ops
    D, F_ = 192, 768
    x = rnd(M, D, seed=121) * 1.5 + 0.2
    xp = rnd(M, D, seed=128)
    gamma, beta = rnd(D, seed=122) * 0.2 + 1.0, rnd(D, seed=123) * 0.1
    g2, b2n = rnd(D, seed=129) * 0.3 + 1.0, rnd(D, seed=130) * 0.2
    W1, b1 = rnd(F_, D, seed=124, scale=0.06).bfloat16(), rnd(F_, seed=125) * 0.1
    W2, b2 = rnd(D, F_, seed=126, scale=0.04).bfloat16(), rnd(D, seed=127) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev()) if gated else None
    kw = dict(x_prev=xp if gated else None, gate=gate)
    if train:
        kw.update(h=torch.empty(M, D, device=dev(), dtype=torch.bfloat16), mean=torch.empty(M, device=dev()), rstd=torch.empty(M, device=dev()),
                  gp=torch.empty(M, F_, device=dev(), dtype=torch.bfloat16), u=torch.empty(M, F_, device=dev(), dtype=torch.bfloat16))
    plain = torch.empty(M, D, device=dev())
    ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, plain, **kw)
    out = torch.full((M, D), float("nan"), device=dev())
    nh = torch.full((M, D), float("nan"), device=dev(), dtype=torch.bfloat16)
    nm, nr = torch.full((M,), float("nan"), device=dev()), torch.full((M,), float("nan"), device=dev())
    ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out, next_gamma=g2, next_beta=b2n, next_h=nh, next_mean=nm, next_rstd=nr, **kw)
    assert torch.equal(out, plain)
    od = out.double()
    ref = F.layer_norm(od, (D,), g2.double(), b2n.double(), 1e-6)
    torch.testing.assert_close(nh.double(), ref, rtol=8e-3, atol=8e-3)                     # bf16 rounding of the result
    torch.testing.assert_close(nm.double(), od.mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(nr.double(), torch.rsqrt(od.var(1, unbiased=False) + 1e-6), rtol=1e-5, atol=0)
    hb = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(out, g2, b2n, hb, mean, rstd, M, D, BF16)
    torch.testing.assert_close(nm, mean, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(nr, rstd, rtol=2e-6, atol=0)
    # same float32 formula, possibly another summation order inside a row: at most one bf16 ulp apart, and almost nowhere
    diff = (nh.float() - hb.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(hb.float().abs().max()) + 1e-6
    assert float((diff > 0).float().mean()) < 0.02
    nh2 = torch.empty_like(nh)                                                              # statistics optional
    ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, torch.empty_like(out), next_gamma=g2, next_beta=b2n, next_h=nh2, **kw)
    assert torch.equal(nh2, nh)


@pytest.mark.parametrize("M,gated,stats,K", [(4096, False, True, 768), (4096 + 16 * 37, True, True, 768), (100864, True, False, 768),
                                             (4096 + 16 * 5, False, True, 192), (100864, False, True, 192), (4096, False, False, 192),
                                             (1576, True, True, 768), (197, False, True, 192), (16, False, True, 768), (8 * 197 + 3, False, True, 192),
                                             (4096 + 21, False, True, 512), (4096, True, True, 256), (1576, False, False, 256)])
def test_fc2_residual_epilogue_writes_the_next_blocks_norm1(M, gated, stats, K):
    """uvc_gemm_nt with ln_out (fc2 + bias + residual [+ gate mix] at K = 768 -> N = 192 writing the next block's norm1; attn.proj + bias +
    residual at K = 192 writing norm2, same kernel on a seven-stage ring): C must not change by a bit, ln_out =
    LayerNorm(C rows) in bf16 with the row statistics combined across the 12 column-slice waves -- against float64 LayerNorm of C and
    against uvc_layernorm_fwd on C; deterministic; refused (UVC_ERR_UNSUPPORTED) where the kernel does not run."""
    from uvc_amd import ops, _lib as L
    D = 192
    A = (rnd(M, K, seed=141) * 0.5).bfloat16()
    W, bias = rnd(D, K, seed=142, scale=0.05).bfloat16(), rnd(D, seed=143) * 0.1
    R, R2 = rnd(M, D, seed=144) * 1.5 + 0.3, rnd(M, D, seed=145)
    R[5] += 40.0                                                                            # a row far from zero mean: centred statistics
    g2, b2n = rnd(D, seed=146) * 0.3 + 1.0, rnd(D, seed=147) * 0.2
    gate = torch.tensor([0.25, 0.75], device=dev()) if gated else None
    epi = ops.EPI_BIAS_RESID_GATE if gated else ops.EPI_BIAS_RESID
    kw = dict(dtype=BF16, epilogue=epi, bias=bias, R=R, R2=R2 if gated else None, gate=gate)
    assert L.lib().uvc_gemm_nt_ln_supported(M, D, K, BF16, epi) == 1
    plain = torch.empty(M, D, device=dev())
    ops.gemm_nt(A, W, plain, **kw)
    out = torch.full((M, D), float("nan"), device=dev())
    nh = torch.full((M, D), float("nan"), device=dev(), dtype=torch.bfloat16)
    nm = torch.full((M,), float("nan"), device=dev()) if stats else None
    nr = torch.full((M,), float("nan"), device=dev()) if stats else None
    ops.gemm_nt(A, W, out, ln_gamma=g2, ln_beta=b2n, ln_out=nh, ln_mean=nm, ln_rstd=nr, **kw)
    assert torch.equal(out, plain)
    od = out.double()
    ref = F.layer_norm(od, (D,), g2.double(), b2n.double(), 1e-6)
    torch.testing.assert_close(nh.double(), ref, rtol=8e-3, atol=8e-3)
    hb = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(out, g2, b2n, hb, mean, rstd, M, D, BF16)
    if stats:
        torch.testing.assert_close(nm.double(), od.mean(1), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(nr.double(), torch.rsqrt(od.var(1, unbiased=False) + 1e-6), rtol=1e-5, atol=0)
        torch.testing.assert_close(nm, mean, rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(nr, rstd, rtol=2e-6, atol=0)
    diff = (nh.float() - hb.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(hb.float().abs().max()) + 1e-6
    assert float((diff > 0).float().mean()) < 0.02
    nh2 = torch.empty_like(nh)
    ops.gemm_nt(A, W, torch.empty_like(out), ln_gamma=g2, ln_beta=b2n, ln_out=nh2, **kw)
    assert torch.equal(nh2, nh)
    # rows do not depend on how many rows there are (any M from 16 up runs this kernel; a ragged last tile overlaps its neighbour)
    m2 = max(16, M // 3 + 5)
    kw2 = dict(kw, R=R[:m2].contiguous(), R2=R2[:m2].contiguous() if gated else None)
    o3, nh3 = torch.empty(m2, D, device=dev()), torch.empty(m2, D, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A[:m2].contiguous(), W, o3, ln_gamma=g2, ln_beta=b2n, ln_out=nh3, **kw2)
    assert torch.equal(o3, out[:m2]) and torch.equal(nh3, nh[:m2])
    # not this kernel's shape: refused, nothing silently skipped
    assert L.lib().uvc_gemm_nt_ln_supported(M, D, 576, BF16, epi) == 0
    if K == 768:
        with pytest.raises(RuntimeError):
            ops.gemm_nt(A[:, :576].contiguous(), W[:, :576].contiguous(), out, ln_gamma=g2, ln_beta=b2n, ln_out=nh2, **kw)


# ---- patch-gating Gumbel top-k at the production shape
@pytest.mark.parametrize("B,P,k,tau", [(64, 196, 176, 0.7), (512, 196, 176, 0.1), (96, 196, 176, 10.0), (3, 16, 14, 1.0)])
def test_patch_topk_mask_production_shape_bit_exact_indices(B, P, k, tau):
    """ops.patch_topk_mask + backward against oracle/vit.py:patch_topk_mask (model_distilled.py:36-63,446-456) at P = 196,
    k = int(0.9 * 196) = 176, over the tau range of joint_train.py:404-407 (0.1 ... 10): hard index sets BIT-EXACT for every
    image whose k-th and (k+1)-th y_soft differ in the reference arithmetic, the straight-through mask / y_soft within
    float32 rounding, d(scores) against float64 autograd.  Row 0 carries a constructed near-tie at the k-th boundary: the
    (k+1)-th largest u sits 64 float32 ulps (>= 2e-5) below the k-th -- resolvable by the 1-ulp expf / logf the kernel uses,
    the regime where a fast-math exp / log chain starts to blur neighbours.  At tau = 0.1 the tail of y_soft underflows
    (u spans > 87 nats): some images tie at the boundary in the reference itself (torch.topk's choice among equal values is
    unspecified) or decide it between subnormals; for those rows the selection must be A valid top-k of the kernel's own
    y_soft (ties to the lower index), every other row is bit-exact."""
    from oracle import vit as OV
    from uvc_amd import ops
    g = torch.Generator().manual_seed(1234 + P + B)
    scores = torch.randn(B, P, generator=g) * 1.5
    e = torch.empty(B, P).exponential_(generator=g)
    logp = F.log_softmax(scores[0].double(), -1)
    u = (logp - e[0].double().log()) / tau
    order = torch.argsort(u, descending=True)
    a, b = int(order[k - 1]), int(order[k])
    gap = max(2e-5, 64.0 * abs(float(u[a])) * 2.0 ** -23)          # 64 float32 ulps of u: resolvable, but only just
    e[0, b] = torch.exp(-((u[a] - gap) * tau - logp[b])).float()
    ref_mask, ref_index = OV.patch_topk_mask(scores.clone(), e, k, tau)
    y_ref = ((F.log_softmax(scores, -1) - e.log()) / tau).softmax(-1)          # the reference's float32 y_soft
    ys_sorted = y_ref.sort(-1, descending=True)[0]
    # decisive rows: the boundary pair differs AND sits in the normal float32 range (a subnormal y keeps only a few bits, so
    # a 1-ulp difference between two correct expf implementations legitimately reorders the tail)
    distinct = (ys_sorted[:, k - 1] > ys_sorted[:, k]) & (ys_sorted[:, k] > 1e-35)
    if tau >= 0.5:
        assert bool(distinct.all()) and a in ref_index[0].tolist() and b not in ref_index[0].tolist()
    sc, ed = scores.to(dev()), e.to(dev())
    mask, ys, ps = (torch.empty(B, P, device=dev()) for _ in range(3))
    ops.patch_topk_mask(sc, ed, mask, ys, ps, B, P, k, float(tau))
    hard = torch.zeros(B, P).scatter_(1, ref_index, 1.0) > 0.5
    hard[:, 0] = True                                  # the observable mask: token 0 is forced to 1 after the scatter (:453)
    ysc = ys.cpu()
    sel = mask.cpu() > 0.5                             # (1 - y) + y vs (0 - y) + y
    n_sel = sel[:, 1:].sum(1)
    assert bool(((n_sel == k) | (n_sel == k - 1)).all()), "k tokens per image (k - 1 besides token 0 when it ranks in the top k)"
    rows = torch.nonzero(distinct).flatten()
    bad = [int(r) for r in rows if not torch.equal(sel[r], hard[r])]
    assert not bad, f"index sets differ from the reference in rows {bad[:8]}"
    for r in torch.nonzero(~distinct).flatten():       # boundary ties in the reference: a valid top-k, ties to the lower index
        yr, sr = ysc[r, 1:], sel[r, 1:]
        assert float(yr[sr].min()) >= float(yr[~sr].max())
        tied = torch.nonzero(yr == yr[sr].min()).flatten()
        chosen = [int(i) for i in tied if sr[i]]
        assert chosen == [int(i) for i in tied[:len(chosen)]]
    m_ref = ref_mask.detach()
    torch.testing.assert_close(mask.cpu()[distinct], m_ref[distinct], rtol=0, atol=2e-6)
    s64 = scores.double().requires_grad_(True)
    y64 = ((F.log_softmax(s64, -1) - e.double().log()) / tau).softmax(-1)
    torch.testing.assert_close(ysc.double(), y64.detach(), rtol=1e-4 if tau < 0.5 else 2e-5, atol=1e-30)
    # backward: d(mask) -> d(scores); token 0 has no gradient (mask[:, 0] = 1 overwrites it)
    dmask = torch.randn(B, P, generator=g)
    dm = dmask.clone().double()
    dm[:, 0] = 0
    (y64 * dm).sum().backward()
    ds = torch.empty(B, P, device=dev())
    ops.patch_topk_mask_bwd(dmask.to(dev()), ys, ps, ds, B, P, float(tau))
    torch.testing.assert_close(ds.cpu().double(), s64.grad, rtol=5e-4, atol=3e-6 / min(tau, 1.0))


def test_patch_topk_mask_exact_ties_pick_the_lower_index():
    """Exactly equal scores and draws: y ties bit-for-bit and the lower token index wins (the reference's torch.topk leaves
    tie order unspecified)."""
    from uvc_amd import ops
    B, P, k = 2, 196, 176
    sc = torch.zeros(B, P, device=dev())
    e = torch.full((B, P), 0.5, device=dev())
    mask, ys, ps = (torch.empty(B, P, device=dev()) for _ in range(3))
    ops.patch_topk_mask(sc, e, mask, ys, ps, B, P, k, 1.0)
    hard = mask.cpu() > 0.5
    assert hard[:, :k].all() and not hard[:, k:].any()


def test_bf16_mode_gelu_approximant_error_bounds():
    """The bf16 mode evaluates GELU as x / (1 + exp(-x (c0 + c1 x^2 + c2 x^4))) and GELU' = Phi~ + x phi (common.h): against the
    exact erf form of nn.GELU (model_distilled.py:108,118) |GELU error| <= 7.4e-4 * max(|GELU|, 2e-3) -- below 2^-10 relative
    wherever |GELU| >= 2e-3 -- and |GELU' error| <= 8.5e-5, i.e. far inside the 2^-9 rounding of the bf16 value it is stored as.
    Driven through the GEMM epilogues with an identity weight (bf16 inputs pass through the MFMA exactly), float32 outputs."""
    from uvc_amd import ops
    K = 64
    xs = torch.cat([torch.linspace(-9.0, 9.0, 64 * 511), torch.tensor([-60.0, 60.0, -1e4, 1e4, 0.0, -0.0] + [0.0] * 58)]).to(torch.bfloat16)
    A = xs.reshape(-1, K).to(dev())
    W = torch.eye(K, device=dev(), dtype=torch.bfloat16)
    bias = torch.zeros(K, device=dev())
    M = A.shape[0]
    u = torch.empty(M, K, device=dev())
    ops.gemm_nt(A, W, u, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_OUT, bias=bias)
    g, u2 = torch.empty(M, K, device=dev()), torch.empty(M, K, device=dev())
    ops.gemm_nt(A, W, g, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bias, C2=u2)
    x = A.double().cpu()
    ref = F.gelu(x)
    phi = torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    dref = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * phi
    err = (u.double().cpu() - ref).abs()
    assert bool((err <= 7.6e-4 * ref.abs().clamp_min(2e-3)).all()), float((err / ref.abs().clamp_min(2e-3)).max())
    assert torch.equal(u, u2)
    assert float((g.double().cpu() - dref).abs().max()) <= 8.5e-5
    assert torch.isfinite(u).all() and torch.isfinite(g).all()
    assert float(u.reshape(-1)[64 * 511 + 3]) == 1e4 or abs(float(u.reshape(-1)[64 * 511 + 3]) - 9984.0) < 1.0    # saturates to x, not to 0


@pytest.mark.parametrize("K", [768, 576])
@pytest.mark.parametrize("with_add2", [False, True])
@pytest.mark.parametrize("M", [4096 + 53, 4096 + 48, 16 * 1031])     # ragged M: register-staged kernel; M % 16 == 0: the LDS-DMA ring (260 and 1031 tiles: more than one per workgroup)
def test_gemm_nt_lnbwd_matches_the_unfused_pair(K, with_add2, M):
    """uvc_gemm_nt_lnbwd (dgrad GEMM with the LayerNorm backward as its epilogue) against float64 math and against the two
    kernels it replaces (uvc_gemm_nt -> bf16 dy -> uvc_layernorm_bwd): dx, dgamma, dbeta, the two gate dot products; ragged M;
    dx aliasing add2 (the engine updates gA in place); two runs bit-identical."""
    from uvc_amd import ops
    D = 192
    A = rnd(M, K, seed=201).to(torch.bfloat16)
    Wt = rnd(D, K, seed=202, scale=0.05).to(torch.bfloat16)
    x = rnd(M, D, seed=203) * 1.5 + 0.3
    gamma = 1.0 + 0.2 * rnd(D, seed=204)
    add1 = rnd(M, D, seed=205).to(torch.bfloat16)
    add2 = rnd(M, D, seed=206).to(torch.bfloat16) if with_add2 else None
    a1 = torch.tensor([0.7], device=dev())
    a2 = torch.tensor([0.3], device=dev()) if with_add2 else None
    mean = x.mean(1)
    rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6)
    # float64 reference
    dy = A.double() @ Wt.double().t()
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    gy = dy * gamma.double()
    ref = rstd.double()[:, None] * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True)) + 0.7 * add1.double()
    if with_add2:
        ref = ref + 0.3 * add2.double()
    ref_dg, ref_db = (dy * xh).sum(0), dy.sum(0)
    nb = max(ops.layernorm_bwd_blocks(M), 256 + 16)
    outs = []
    for rep in range(2):
        dx = add2.clone() if with_add2 else torch.empty(M, D, device=dev(), dtype=torch.bfloat16)     # in place over add2
        part = torch.empty(nb * (2 * D + 2), device=dev())
        dg, db, dots = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.zeros(2, device=dev())
        ops.gemm_nt_lnbwd(A, Wt, x, mean, rstd, gamma, dx, part, dg, db, add1=add1, a1=a1, add2=dx if with_add2 else None, a2=a2, dots=dots)
        outs.append((dx.clone(), dg.clone(), db.clone(), dots.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "not deterministic"
    dx, dg, db, dots = outs[0]
    torch.testing.assert_close(dx.double(), ref, rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(dg.double(), ref_dg, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(db.double(), ref_db, rtol=2e-3, atol=2e-2)
    # <dx, x> from the unrounded dx against the bf16-rounded one: a random walk of M*D roundings of relative size 2^-9
    torch.testing.assert_close(dots[0].double(), (dx.double() * x.double()).sum(), rtol=2e-3, atol=0.5 + 0.01 * math.sqrt(M * D))
    if with_add2:
        torch.testing.assert_close(dots[1].double(), (add2.double() * x.double()).sum(), rtol=1e-4, atol=0.1)
    # the unfused pair
    dyb = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, Wt, dyb, dtype=BF16, epilogue=ops.EPI_NONE)
    dx2 = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    dg2, db2, dots2 = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.zeros(2, device=dev())
    part = torch.empty(nb * (2 * D + 2), device=dev())
    ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dx2, part, dg2, db2, M, D, BF16, add1=add1, a1=a1, add2=add2, a2=a2, dots=dots2 if with_add2 else None)
    # the pair rounds dy to bf16 between its two kernels (relative 2^-9 per element, ~0.3 absolute on a 4149-row sum of O(1) terms)
    torch.testing.assert_close(dx.float(), dx2.float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dg, dg2, rtol=5e-3, atol=1.0)
    torch.testing.assert_close(db, db2, rtol=5e-3, atol=1.0)
    assert float((dg.double() - ref_dg).abs().max()) <= float((dg2.double() - ref_dg).abs().max()) + 1e-3     # the fused kernel is the more exact one


def test_keyed_exp_noise_is_a_pure_function_of_its_key():
    """uvc_exp_noise / ops.KeyedExpSource: Exp(1) statistics, strictly positive, reproducible from (seed, step, site) alone and
    independent across steps / sites / seeds -- what lets data-parallel replicas draw identical Gumbel noise without a shared RNG."""
    from uvc_amd import ops
    a = ops.KeyedExpSource(730, dev())
    a.begin_step(7)
    x0, x1 = a((12, 2)), a((512, 196))
    b = ops.KeyedExpSource(730, dev())
    b.begin_step(7)
    y0, y1 = b((12, 2)), b((512, 196))
    assert torch.equal(x0, y0) and torch.equal(x1, y1)
    assert not torch.equal(x1.flatten()[:24].reshape(12, 2), x0)                      # another site: another stream
    b.begin_step(7)                                                                     # the same step number again REPLAYS the key ...
    assert torch.equal(b((12, 2)), x0)
    b.begin_step(7, resume_window=True)                                                 # ... unless the caller says the window is carried over
    assert torch.equal(b((512, 196)), x1)
    b.begin_step(8)
    z0 = b((12, 2))
    assert not torch.equal(z0, x0)                                                      # another step
    c = ops.KeyedExpSource(731, dev())
    c.begin_step(7)
    assert not torch.equal(c((12, 2)), x0)                                              # another seed
    big = ops.KeyedExpSource(1, dev())
    big.begin_step(0)
    e = big((4_000_000,)).double()
    assert float(e.min()) > 0 and torch.isfinite(e).all()
    assert abs(float(e.mean()) - 1.0) < 3e-3 and abs(float(e.var()) - 1.0) < 1e-2
    # Exp(1) quantiles: P(E > t) = exp(-t)
    for t in (0.1, 1.0, 3.0):
        assert abs(float((e > t).double().mean()) - math.exp(-t)) < 2e-3
    # lag-1 correlation of consecutive elements ~ 0
    assert abs(float(((e[1:] - 1) * (e[:-1] - 1)).mean())) < 3e-3


# ---- the 256 x 256-tile LDS-DMA NT kernel of the wide models (DeiT-Small / Base, T2T-ViT: K >= 256 against N >= 256)
@pytest.mark.parametrize("M,N,K", [(25216, 2304, 768), (25216, 3072, 768), (64 * 256 + 37, 2560, 512), (50432, 1000, 256), (40000, 1152, 1152),
                                   (25216, 768, 3072)])
def test_gemm_nt256_matches_generic_kernel_and_float64(M, N, K):
    """uvc_gemm_nt on the shapes k_gemm_nt256 takes (>= 640 tiles of 256 x 256; the last shape stays on the 128 x 128 kernel), every epilogue,
    bf16 and float32 outputs: bit-identical to the generic 128 x 128 kernel (same k-ordered accumulation chain per element, same epilogue
    arithmetic: force_generic = 1), ragged M and N included (rows / columns past the edge are read as zeros through the buffer descriptor
    and never stored), more than one tile per persistent workgroup, and within bf16 tolerance of float64."""
    from uvc_amd import ops
    A = (rnd(M, K, seed=301) * 0.5).bfloat16()
    W = rnd(N, K, seed=302, scale=0.04).bfloat16()
    bias = rnd(N, seed=303) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev())
    aux = rnd(M, N, seed=304).bfloat16()
    ref = A.double() @ W.double().t()
    for cdt in (torch.bfloat16, torch.float32):
        R, R2 = rnd(M, N, seed=305).to(cdt), rnd(M, N, seed=306).to(cdt)
        cases = [(ops.EPI_NONE, {}), (ops.EPI_BIAS, dict(bias=bias)), (ops.EPI_BIAS_RESID, dict(bias=bias, R=R)),
                 (ops.EPI_BIAS_RESID_GATE, dict(bias=bias, R=R, R2=R2, gate=gate)), (ops.EPI_MUL_AUX, dict(aux=aux)),
                 (ops.EPI_BIAS_GELU_OUT, dict(bias=bias))]
        if cdt == torch.bfloat16:
            cases.append((ops.EPI_BIAS_GELU_GRAD, dict(bias=bias, C2=True)))
        for epi, kw in cases:
            outs = []
            for fg in (0, 1):
                C1 = torch.full((M, N), float("nan"), device=dev(), dtype=cdt)
                k2 = dict(kw)
                if k2.get("C2") is True:
                    k2["C2"] = torch.full((M, N), float("nan"), device=dev(), dtype=cdt)
                ops.gemm_nt(A, W, C1, dtype=BF16, epilogue=epi, force_generic=fg, **k2)
                outs.append((C1, k2.get("C2")))
            assert torch.equal(outs[0][0], outs[1][0]), (epi, cdt)
            if outs[0][1] is not None:
                assert torch.equal(outs[0][1], outs[1][1]), (epi, cdt, "C2")
            if epi == ops.EPI_BIAS:
                torch.testing.assert_close(outs[0][0].double(), ref + bias.double(), rtol=2e-2, atol=2e-2)
            if epi == ops.EPI_NONE:
                torch.testing.assert_close(outs[0][0].double(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("ri", [8, 6, 5, 4])
@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (2364, 2304, 768), (160 * 9 + 77, 768, 256), (97, 512, 320), (8192 + 19, 1000, 1536)])
def test_gemm_nt8p_every_tile_height_matches_generic_kernel(M, N, K, ri):
    """k_gemm_nt8p (two wave groups half a phase apart, four-slot LDS ring with counted waits) at every tile height it is built for
    (32 * ri rows x 256 columns; force_generic = 0x100 | ri picks it at any size): bit-identical to the generic 128 x 128 kernel for
    every epilogue -- ragged M and N, fewer rows than one tile, an odd number of k-steps per tile (K = 320: the buffer parity flips from
    tile to tile), more tiles than workgroups, several tiles per persistent workgroup; twice in a row (a race between the LDS-DMA ring
    and the fragment reads would show as run-to-run differences)."""
    from uvc_amd import ops
    A = (rnd(M, K, seed=311) * 0.5).bfloat16()
    W = rnd(N, K, seed=312, scale=0.04).bfloat16()
    bias = rnd(N, seed=313) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev())
    aux = rnd(M, N, seed=314).bfloat16()
    R, R2 = rnd(M, N, seed=315).bfloat16(), rnd(M, N, seed=316).bfloat16()
    cases = [(ops.EPI_NONE, {}), (ops.EPI_BIAS, dict(bias=bias)), (ops.EPI_BIAS_RESID, dict(bias=bias, R=R)),
             (ops.EPI_BIAS_RESID_GATE, dict(bias=bias, R=R, R2=R2, gate=gate)), (ops.EPI_MUL_AUX, dict(aux=aux)),
             (ops.EPI_BIAS_GELU_OUT, dict(bias=bias)), (ops.EPI_BIAS_GELU_GRAD, dict(bias=bias, C2=True))]
    for epi, kw in cases:
        outs = []
        for fg in (0x100 | ri, 1, 0x100 | ri):
            C1 = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
            k2 = dict(kw)
            if k2.get("C2") is True:
                k2["C2"] = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
            ops.gemm_nt(A, W, C1, dtype=BF16, epilogue=epi, force_generic=fg, **k2)
            outs.append((C1, k2.get("C2")))
        for o in (outs[0], outs[2]):
            assert torch.equal(o[0], outs[1][0]), (epi, ri)
            if o[1] is not None:
                assert torch.equal(o[1], outs[1][1]), (epi, ri, "C2")


@pytest.mark.parametrize("dtype", [F32, BF16])
def test_cast_transpose_multi_all_shadows_in_one_launch(dtype):
    """uvc_cast_transpose_multi (the per-step refresh of every weight shadow): 64 x 64 tiles with 16-byte loads / 8-byte bf16 stores where
    the matrix allows it, the scalar path for shapes and offsets that do not (R or C not a multiple of 4, odd offsets), both against torch."""
    import ctypes as C
    from uvc_amd import _lib as L, ops
    shapes = [(576, 192), (192, 192), (768, 192), (192, 768), (1000, 192), (192, 768), (70, 36), (37, 50), (64, 64), (130, 4), (8, 260)]
    T = ops.tdtype(dtype)
    n = len(shapes)
    srcs, ws, wts, off, so = [], [], [], 0, 0
    for i, (R, Cc) in enumerate(shapes):
        if i == 7:
            off += 1                                       # an odd source offset: scalar path
        srcs.append(off); off += R * Cc
        ws.append(so if i % 3 != 2 else -1); so += R * Cc
        wts.append(so if i % 4 != 3 else -1); so += R * Cc
    params = rnd(off + 3, seed=91)
    shadow = torch.full((so,), 7.0, device=dev(), dtype=T)
    arr64 = lambda v: (C.c_int64 * n)(*v)
    arr32 = lambda v: (C.c_int32 * n)(*v)
    L.check(L.lib().uvc_cast_transpose_multi(L.ptr(params), L.ptr(shadow), n, arr64(srcs), arr32([s[0] for s in shapes]), arr32([s[1] for s in shapes]),
                                             arr64(ws), arr64(wts), dtype, L.cur_stream()), "uvc_cast_transpose_multi")
    torch.cuda.synchronize()
    for i, (R, Cc) in enumerate(shapes):
        W = params[srcs[i]:srcs[i] + R * Cc].view(R, Cc)
        if ws[i] >= 0:
            assert torch.equal(shadow[ws[i]:ws[i] + R * Cc].view(R, Cc), W.to(T)), ("w", i, R, Cc)
        if wts[i] >= 0:
            assert torch.equal(shadow[wts[i]:wts[i] + R * Cc].view(Cc, R), W.t().contiguous().to(T)), ("wt", i, R, Cc)


@pytest.mark.parametrize("K,with_add2,x_bf16", [(1536, False, True), (1152, True, True), (384, True, False), (1152, False, False)])
@pytest.mark.parametrize("M", [128 * 40 + 53, 128 * 300, 50432])     # ragged last tile; more tiles than workgroups; DeiT-Small at batch 256
def test_gemm_nt_lnbwd_row_tile_d384_equals_the_unfused_pair(K, with_add2, x_bf16, M):
    """D = 384 (DeiT-Small, T2T-ViT-14): uvc_gemm_nt_lnbwd runs k_gemm_row384_lnbwd -- k_gemm_nt256's main loop on a 128 x 384 row tile, the
    tile rounded to bf16 in LDS (what the GEMM of the unfused pair stores) and k_ln_bwd_v's row arithmetic over it.  dx must equal
    uvc_gemm_nt -> uvc_layernorm_bwd BIT FOR BIT; dgamma / dbeta / dots are other partitions of the same sums; two runs bit-identical;
    dx in place over add2."""
    from uvc_amd import ops
    D = 384
    A = rnd(M, K, seed=211).to(torch.bfloat16)
    Wt = rnd(D, K, seed=212, scale=0.04).to(torch.bfloat16)
    x = rnd(M, D, seed=213) * 1.5 + 0.3
    if x_bf16:
        x = x.to(torch.bfloat16)
    gamma = 1.0 + 0.2 * rnd(D, seed=214)
    add1 = rnd(M, D, seed=215).to(torch.bfloat16)
    add2 = rnd(M, D, seed=216).to(torch.bfloat16) if with_add2 else None
    a1 = torch.tensor([0.7], device=dev())
    a2 = torch.tensor([0.3], device=dev()) if with_add2 else None
    mean = x.float().mean(1)
    rstd = torch.rsqrt(x.float().var(1, unbiased=False) + 1e-6)
    assert ops.gemm_lnbwd_supported(M, D, K, BF16)
    nb = max(ops.layernorm_bwd_blocks(M), 256 + 16)
    outs = []
    for rep in range(2):
        dx = add2.clone() if with_add2 else torch.full((M, D), float("nan"), device=dev(), dtype=torch.bfloat16)
        part = torch.full((nb * (2 * D + 2),), float("nan"), device=dev())
        dg, db, dots = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.zeros(2, device=dev())
        ops.gemm_nt_lnbwd(A, Wt, x, mean, rstd, gamma, dx, part, dg, db, add1=add1, a1=a1, add2=dx if with_add2 else None, a2=a2, dots=dots)
        outs.append((dx.clone(), dg.clone(), db.clone(), dots.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), "not deterministic"
    dx, dg, db, dots = outs[0]
    dyb = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, Wt, dyb, dtype=BF16, epilogue=ops.EPI_NONE)
    dx2 = torch.empty(M, D, device=dev(), dtype=torch.bfloat16)
    dg2, db2, dots2 = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.zeros(2, device=dev())
    part = torch.empty(nb * (2 * D + 2), device=dev())
    ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dx2, part, dg2, db2, M, D, BF16, add1=add1, a1=a1, add2=add2, a2=a2, dots=dots2 if with_add2 else None)
    assert bool(torch.isfinite(dx.float()).all())
    assert torch.equal(dx, dx2)
    torch.testing.assert_close(dg, dg2, rtol=1e-4, atol=1e-2 + 1e-5 * M)
    torch.testing.assert_close(db, db2, rtol=1e-4, atol=1e-2 + 1e-5 * M)
    if with_add2:
        torch.testing.assert_close(dots, dots2, rtol=1e-4, atol=0.5)


@pytest.mark.parametrize("M,K", [(4096 + 37, 1536), (25216, 1152), (4096, 768), (50432, 1536), (50432, 384), (4096 + 129, 384)])
@pytest.mark.parametrize("gate", [False, True])
def test_gemm_nt_row384_forward_with_layernorm_is_bit_identical_to_the_unfused_pair(M, K, gate):
    """uvc_gemm_nt with ln_out at N = 384 (k_gemm_row384_lnbwd<.., 1>: fc2 + bias + residual [+ gate mix] of DeiT-Small / T2T-ViT writing the next
    block's norm1, attn.proj + bias + residual writing norm2, from the same launch): C, the LayerNorm rows and the statistics equal the generic kernel + the stand-alone LayerNorm pass BIT FOR BIT
    (so whether a batch takes the fused form cannot show in its results) -- ragged M (a partial last row tile), an odd and an even number of k-steps,
    more tiles than workgroups; inference form (no statistics wanted) included; twice in a row."""
    from uvc_amd import ops
    N = 384
    A = (rnd(M, K, seed=411) * 0.5).bfloat16()
    W = rnd(N, K, seed=412, scale=0.04).bfloat16()
    bias = rnd(N, seed=413) * 0.1
    R, R2 = rnd(M, N, seed=414).bfloat16(), rnd(M, N, seed=415).bfloat16()
    gm, bt = 1 + 0.1 * rnd(N, seed=416), 0.1 * rnd(N, seed=417)
    gate_t = torch.tensor([0.3, 0.7], device=dev())
    epi = ops.EPI_BIAS_RESID_GATE if gate else ops.EPI_BIAS_RESID
    kw = dict(dtype=BF16, epilogue=epi, bias=bias, R=R)
    if gate:
        kw.update(R2=R2, gate=gate_t)
    from uvc_amd import _lib as L
    assert L.lib().uvc_gemm_nt_ln_supported(M, N, K, BF16, epi) == 1
    C0 = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, W, C0, force_generic=1, **kw)
    h0 = torch.empty_like(C0)
    m0, r0 = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(C0, gm, bt, h0, m0, r0, M, N, BF16)
    for trial in range(2):
        C1 = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
        h1 = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
        m1, r1 = torch.full((M,), float("nan"), device=dev()), torch.full((M,), float("nan"), device=dev())
        ops.gemm_nt(A, W, C1, ln_gamma=gm, ln_beta=bt, ln_out=h1, ln_mean=m1, ln_rstd=r1, **kw)
        assert torch.equal(C1, C0), (trial, "C")
        assert torch.equal(h1, h0), (trial, "LayerNorm rows")
        assert torch.equal(m1, m0) and torch.equal(r1, r0), (trial, "statistics")
    h2 = torch.full((M, N), float("nan"), device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, W, torch.empty_like(C0), ln_gamma=gm, ln_beta=bt, ln_out=h2, **kw)          # no statistics (no-grad forward)
    assert torch.equal(h2, h0)


@pytest.mark.parametrize("M", [4096 + 37, 16 * 1031])
def test_gelu_grad_one_byte_code_round_trip(M):
    """r6: GELU'(a) of fc1 as ONE byte per activation (UVC_EPI_BIAS_GELU_GRAD_Q8 -> UVC_EPI_MUL_AUX_Q8; include/uvc_kernels.h).
      * the byte decodes to GELU'(a) within STEP / 2 = 2.47e-3 of the float64 value EVERYWHERE (bf16's own spacing on [1, 2) is 7.8e-3), the code is
        exactly the header's formula applied to the kernel's own bf16-mode GELU' (codes differ from it by at most one step at a rounding boundary), and
        GELU(a) (C2) is bit-identical to the two-byte epilogue's;
      * the backward's product (alpha * acc) * decode(code) equals the float64 product with the DECODED multiplier to bf16 rounding, and agrees with the
        two-byte path (bf16 GELU') to the sum of the two codes' errors;
      * ragged M, several tiles per workgroup; unsupported shapes are refused loudly."""
    from uvc_amd import ops
    K, N = 192, 768
    A = rnd(M, K, seed=91).to(torch.bfloat16)
    W = rnd(N, K, seed=92, scale=0.09).to(torch.bfloat16)           # pre-activations with a spread of ~1.2: both tails of GELU' are visited
    bias = rnd(N, seed=93, scale=0.3)
    a64 = A.double() @ W.double().t() + bias.double()
    gp64 = 0.5 * (1 + torch.erf(a64 / math.sqrt(2))) + a64 * torch.exp(-0.5 * a64 * a64) / math.sqrt(2 * math.pi)
    assert ops.gemm_nt_q8_supported(M, N, K, BF16)
    q = torch.full((M, N), 77, device=dev(), dtype=torch.uint8)
    u8 = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, W, q, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_GRAD_Q8, bias=bias, C2=u8)
    g16, u16 = torch.empty(M, N, device=dev(), dtype=torch.bfloat16), torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(A, W, g16, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_GRAD, bias=bias, C2=u16)
    assert torch.equal(u8, u16)
    dec = q.double() * ops.Q8_STEP + ops.Q8_LO
    err = (dec - gp64).abs()
    assert float(err.max()) <= ops.Q8_STEP / 2 + 2e-4, float(err.max())       # (+ the bf16-mode GELU' approximation, <= 8.3e-5, and float32 rounding of the code)
    assert float(err.pow(2).mean().sqrt()) <= 1.6e-3
    assert int(q.min()) >= 0 and int(q.max()) <= 255 and int(q.max()) - int(q.min()) > 200           # the code's range is used
    # the two-byte tensor of the same launch shape, coded on the host with the header's formula: same codes but for values on a rounding boundary
    host = torch.clamp(torch.floor((g16.double() - ops.Q8_LO) / ops.Q8_STEP + 0.5), 0, 255)
    d = (host - q.double()).abs()
    assert float(d.max()) <= 1.0 and float((d > 0).double().mean()) <= 0.25            # (g16 is itself rounded to bf16: up to a step apart where it crosses a boundary)
    # ---- the backward's product
    G = rnd(M, K, seed=94).to(torch.bfloat16)
    W2t = rnd(N, K, seed=95, scale=0.05).to(torch.bfloat16)
    alpha = torch.tensor([0.6], device=dev())
    dA8, dA16 = torch.empty(M, N, device=dev(), dtype=torch.bfloat16), torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ops.gemm_nt(G, W2t, dA8, dtype=BF16, epilogue=ops.EPI_MUL_AUX_Q8, aux=q, alpha_ptr=alpha)
    ops.gemm_nt(G, W2t, dA16, dtype=BF16, epilogue=ops.EPI_MUL_AUX, aux=g16, alpha_ptr=alpha)
    acc = 0.6 * (G.double() @ W2t.double().t())
    ref8 = acc * dec
    torch.testing.assert_close(dA8.double(), ref8, rtol=2 ** -8, atol=1e-6)              # one bf16 rounding of the exact product with the decoded multiplier
    ref = acc * gp64
    e8, e16 = float((dA8.double() - ref).norm() / ref.norm()), float((dA16.double() - ref).norm() / ref.norm())
    assert e8 <= 4e-3 and e16 <= 4e-3 and e8 <= 1.6 * e16 + 5e-4, (e8, e16)            # as accurate as the two-byte path (CPU study: 2.0e-3 - 2.6e-3 against 1.6e-3 - 2.0e-3)
    # determinism
    q2 = torch.empty_like(q)
    ops.gemm_nt(A, W, q2, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_GRAD_Q8, bias=bias, C2=u8)
    assert torch.equal(q, q2)


def test_gelu_grad_one_byte_code_is_refused_where_it_does_not_exist():
    from uvc_amd import _lib as L
    from uvc_amd import ops
    A = rnd(4096, 384, seed=96).to(torch.bfloat16)
    W = rnd(1536, 384, seed=97, scale=0.05).to(torch.bfloat16)
    q = torch.empty(4096, 1536, device=dev(), dtype=torch.uint8)
    u = torch.empty(4096, 1536, device=dev(), dtype=torch.bfloat16)
    assert not ops.gemm_nt_q8_supported(4096, 1536, 384, BF16) and not ops.gemm_nt_q8_supported(2048, 768, 192, BF16) and not ops.gemm_nt_q8_supported(8192, 768, 192, F32)
    with pytest.raises(L.UvcHipError, match="one-byte"):
        ops.gemm_nt(A, W, q, dtype=BF16, epilogue=ops.EPI_BIAS_GELU_GRAD_Q8, bias=torch.zeros(1536, device=dev()), C2=u)
    A2 = rnd(8192, 192, seed=98).to(torch.bfloat16)
    W2 = rnd(768, 192, seed=99, scale=0.05).to(torch.bfloat16)
    with pytest.raises(L.UvcHipError, match="one-byte"):       # the generic kernel has no such epilogue
        ops.gemm_nt(A2, W2, torch.empty(8192, 768, device=dev(), dtype=torch.bfloat16), dtype=BF16, epilogue=ops.EPI_MUL_AUX_Q8,
                    aux=torch.zeros(8192, 768, device=dev(), dtype=torch.uint8), force_generic=True)
