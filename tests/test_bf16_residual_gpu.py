"""GPU: the bf16 residual stream of the throughput mode (uvc_vit_cfg.resid_f32 = 0).

The rows every block reads and writes (x_l, x1: model_distilled.py:240,244,493) are stored as bf16; all arithmetic on them -- the
residual additions, the gate mix, LayerNorm statistics, the LayerNorm backward -- stays float32 on the loaded values, and a row is
rounded ONCE, where it is stored.  So every kernel with bf16 rows must equal, BIT FOR BIT, the float32-row kernel of rounds 1-2 fed
with the same (bf16-representable) rows and rounded to bf16 afterwards: that is what these tests assert, kernel by kernel, on the
shapes that select each implementation (generic tiled kernel, register-staged streaming kernels, LDS-DMA rings).  Where a kernel
also writes the next LayerNorm (ln_out / next_h) the norm is taken of the ROUNDED rows (what the consumers of the stream read):
checked against uvc_layernorm_fwd on the stored rows and against float64.  The engine-level check (a full step against the oracle
per tensor, both stream types) is tests/test_streaming_batch_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16 = 1
bf = torch.bfloat16


def dev():
    return torch.device("cuda")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


@pytest.mark.parametrize("rows,D", [(1576, 192), (4096 + 5, 192), (333, 384), (130, 768)])
def test_layernorm_on_bf16_rows_equals_the_float32_row_kernel(rows, D):
    from uvc_amd import ops
    xb = (rnd(rows, D, seed=1) * 1.5 + 0.3).to(bf)
    x32 = xb.float()
    gamma, beta = rnd(D, seed=2) * 0.2 + 1.0, rnd(D, seed=3) * 0.1
    outs = []
    for x in (xb, x32):
        y = torch.empty(rows, D, device=dev(), dtype=bf)
        mean, rstd = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
        ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, BF16)
        dy = rnd(rows, D, seed=4).to(bf)
        add1 = rnd(rows, D, seed=5).to(bf)
        a1 = torch.tensor([0.6], device=dev())
        dx = torch.empty(rows, D, device=dev(), dtype=bf)
        part = torch.empty(ops.layernorm_bwd_blocks(rows) * (2 * D + 2), device=dev())
        dg, db, dots = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.zeros(2, device=dev())
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, part, dg, db, rows, D, BF16, add1=add1, a1=a1, add2=add1, a2=a1, dots=dots)
        outs.append((y, mean, rstd, dx, dg, db, dots))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = F.layer_norm(x32.double(), (D,), gamma.double(), beta.double(), 1e-6)
    torch.testing.assert_close(outs[0][0].double(), ref, rtol=8e-3, atol=8e-3)


def test_layernorm_on_bf16_rows_strided_token_rows():
    """The final norm reads the class-token rows of a [B, N, D] stream (rows_per_group / group_stride addressing)."""
    from uvc_amd import ops
    B, N, D, ntok = 6, 197, 192, 1
    xb = rnd(B, N, D, seed=7).to(bf)
    gamma, beta = rnd(D, seed=8) * 0.2 + 1.0, rnd(D, seed=9) * 0.1
    y = torch.empty(B * ntok, D, device=dev(), dtype=bf)
    mean, rstd = torch.empty(B * ntok, device=dev()), torch.empty(B * ntok, device=dev())
    ops.layernorm_fwd(xb, gamma, beta, y, mean, rstd, B * ntok, D, BF16, rows_per_group=ntok, group_stride=N * D)
    ref = F.layer_norm(xb[:, :ntok].double().reshape(B * ntok, D), (D,), gamma.double(), beta.double(), 1e-6)
    torch.testing.assert_close(y.double(), ref, rtol=8e-3, atol=8e-3)


def test_assemble_tokens_writes_bf16_rows():
    from uvc_amd import ops
    B, P, D = 5, 196, 192
    pe, cls, pos = rnd(B, P, D, seed=11), rnd(D, seed=12), rnd(P + 1, D, seed=13)
    mask = (rnd(B, P, seed=14) > 0).float()
    t32 = torch.empty(B, P + 1, D, device=dev())
    t16 = torch.empty(B, P + 1, D, device=dev(), dtype=bf)
    ops.assemble_tokens(pe, cls, None, pos, mask, t32, B, P, D, 1)
    ops.assemble_tokens(pe, cls, None, pos, mask, t16, B, P, D, 1)
    assert torch.equal(t16, t32.to(bf))


# generic kernel (M < 4096), k_gemm_ws / k_gemm_wsn16 register-staged (M >= 4096, M % 16 != 0), the LDS-DMA rings (M % 16 == 0)
@pytest.mark.parametrize("M", [1576, 4096 + 21, 4096 + 16 * 7, 100864])
@pytest.mark.parametrize("K,gated", [(192, False), (768, True), (768, False), (512, True), (256, False)])
def test_residual_epilogues_on_bf16_rows_are_the_rounded_float32_row_results(M, K, gated):
    from uvc_amd import ops
    if M == 100864 and K not in (192, 768):
        pytest.skip("full size on the two production shapes only")
    D = 192
    A = (rnd(M, K, seed=21) * 0.5).to(bf)
    W, bias = rnd(D, K, seed=22, scale=0.05).to(bf), rnd(D, seed=23) * 0.1
    Rb, R2b = (rnd(M, D, seed=24) * 1.5 + 0.3).to(bf), rnd(M, D, seed=25).to(bf)
    gate = torch.tensor([0.25, 0.75], device=dev()) if gated else None
    epi = ops.EPI_BIAS_RESID_GATE if gated else ops.EPI_BIAS_RESID
    c32 = torch.empty(M, D, device=dev())
    ops.gemm_nt(A, W, c32, dtype=BF16, epilogue=epi, bias=bias, R=Rb.float(), R2=R2b.float() if gated else None, gate=gate)
    for fg in (0, 1, 2):
        c16 = torch.full((M, D), float("nan"), device=dev(), dtype=bf)
        ops.gemm_nt(A, W, c16, dtype=BF16, epilogue=epi, bias=bias, R=Rb, R2=R2b if gated else None, gate=gate, force_generic=fg)
        assert torch.equal(c16, c32.to(bf)), f"force_generic={fg}"
    with pytest.raises(RuntimeError):                      # the residual operands have C's element type
        ops.gemm_nt(A, W, torch.empty(M, D, device=dev(), dtype=bf), dtype=BF16, epilogue=epi, bias=bias, R=Rb.float(), R2=R2b.float() if gated else None, gate=gate)


@pytest.mark.parametrize("M,K,gated", [(4096, 768, True), (4096 + 16 * 37, 768, False), (100864, 768, True), (100864, 192, False), (1576, 192, False),
                                       (197, 768, True), (8 * 197 + 3, 192, False), (4096 + 21, 512, True), (1576, 256, False)])
def test_next_layernorm_of_bf16_rows_is_taken_of_the_stored_rows(M, K, gated):
    """ln_out with a bf16 C: C unchanged by the extra output, ln_out = LayerNorm(stored bf16 rows) -- against uvc_layernorm_fwd on C
    (same float32 formula, another summation order inside a row: within one bf16 ulp, almost nowhere different) and float64;
    rows independent of M."""
    from uvc_amd import ops
    D = 192
    A = (rnd(M, K, seed=31) * 0.5).to(bf)
    W, bias = rnd(D, K, seed=32, scale=0.05).to(bf), rnd(D, seed=33) * 0.1
    Rb, R2b = (rnd(M, D, seed=34) * 1.5 + 0.3).to(bf), rnd(M, D, seed=35).to(bf)
    Rb[5] += 40.0
    g2, b2n = rnd(D, seed=36) * 0.3 + 1.0, rnd(D, seed=37) * 0.2
    gate = torch.tensor([0.25, 0.75], device=dev()) if gated else None
    epi = ops.EPI_BIAS_RESID_GATE if gated else ops.EPI_BIAS_RESID
    kw = dict(dtype=BF16, epilogue=epi, bias=bias, R=Rb, R2=R2b if gated else None, gate=gate)
    plain = torch.empty(M, D, device=dev(), dtype=bf)
    ops.gemm_nt(A, W, plain, **kw)
    out = torch.full((M, D), float("nan"), device=dev(), dtype=bf)
    nh = torch.full((M, D), float("nan"), device=dev(), dtype=bf)
    nm, nr = torch.full((M,), float("nan"), device=dev()), torch.full((M,), float("nan"), device=dev())
    ops.gemm_nt(A, W, out, ln_gamma=g2, ln_beta=b2n, ln_out=nh, ln_mean=nm, ln_rstd=nr, **kw)
    assert torch.equal(out, plain)
    od = out.double()
    torch.testing.assert_close(nh.double(), F.layer_norm(od, (D,), g2.double(), b2n.double(), 1e-6), rtol=8e-3, atol=8e-3)
    torch.testing.assert_close(nm.double(), od.mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(nr.double(), torch.rsqrt(od.var(1, unbiased=False) + 1e-6), rtol=1e-5, atol=0)
    hb = torch.empty(M, D, device=dev(), dtype=bf)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(out, g2, b2n, hb, mean, rstd, M, D, BF16)
    torch.testing.assert_close(nm, mean, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(nr, rstd, rtol=2e-6, atol=0)
    diff = (nh.float() - hb.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(hb.float().abs().max()) + 1e-6
    assert float((diff > 0).float().mean()) < 0.02
    m2 = max(16, M // 3 + 5)
    o3, nh3 = torch.empty(m2, D, device=dev(), dtype=bf), torch.empty(m2, D, device=dev(), dtype=bf)
    ops.gemm_nt(A[:m2].contiguous(), W, o3, ln_gamma=g2, ln_beta=b2n, ln_out=nh3, **dict(kw, R=Rb[:m2].contiguous(), R2=R2b[:m2].contiguous() if gated else None))
    assert torch.equal(o3, out[:m2]) and torch.equal(nh3, nh[:m2])


@pytest.mark.parametrize("K", [768, 576])
@pytest.mark.parametrize("with_add2", [False, True])
@pytest.mark.parametrize("M", [4096 + 53, 4096 + 48, 16 * 1031])      # register-staged kernel | the LDS-DMA ring
def test_gemm_nt_lnbwd_on_bf16_rows_equals_the_float32_row_kernel(K, with_add2, M):
    from uvc_amd import ops
    D = 192
    A = rnd(M, K, seed=41).to(bf)
    Wt = rnd(D, K, seed=42, scale=0.05).to(bf)
    xb = (rnd(M, D, seed=43) * 1.5 + 0.3).to(bf)
    gamma = 1.0 + 0.2 * rnd(D, seed=44)
    add1 = rnd(M, D, seed=45).to(bf)
    add2 = rnd(M, D, seed=46).to(bf) if with_add2 else None
    a1 = torch.tensor([0.7], device=dev())
    a2 = torch.tensor([0.3], device=dev()) if with_add2 else None
    mean = xb.float().mean(1)
    rstd = torch.rsqrt(xb.float().var(1, unbiased=False) + 1e-6)
    nb = max(ops.layernorm_bwd_blocks(M), 256 + 16)
    res = []
    for x in (xb, xb.float()):
        for variant in (0, 1):
            dx = add2.clone() if with_add2 else torch.empty(M, D, device=dev(), dtype=bf)
            part = torch.empty(nb * (2 * D + 2), device=dev())
            dg, db, dots = torch.empty(D, device=dev()), torch.empty(D, device=dev()), torch.zeros(2, device=dev())
            ops.gemm_nt_lnbwd(A, Wt, x, mean, rstd, gamma, dx, part, dg, db, add1=add1, a1=a1, add2=dx if with_add2 else None, a2=a2, dots=dots,
                              variant=variant)
            res.append((dx, dg, db, dots))
    # bf16 rows == float32 rows holding the same values, per variant (0: ring where M % 16 == 0, 1: register-staged)
    for v in (0, 1):
        for a, b in zip(res[v], res[2 + v]):
            assert torch.equal(a, b), v


@pytest.mark.parametrize("M,gated,train", [(1576, False, False), (4096 + 37, True, True), (300, False, True), (100864, False, False)])
def test_fused_mlp_on_bf16_rows(M, gated, train):
    """uvc_mlp_fused_fwd with bf16 rows: out = round(the float32-row kernel's result on the same values), bit for bit; next_h is the
    LayerNorm of the rounded rows."""
    from uvc_amd import ops
    D, F_ = 192, 768
    xb = (rnd(M, D, seed=51) * 1.5 + 0.2).to(bf)
    xpb = rnd(M, D, seed=58).to(bf)
    gamma, beta = rnd(D, seed=52) * 0.2 + 1.0, rnd(D, seed=53) * 0.1
    g2, b2n = rnd(D, seed=59) * 0.3 + 1.0, rnd(D, seed=60) * 0.2
    W1, b1 = rnd(F_, D, seed=54, scale=0.06).to(bf), rnd(F_, seed=55) * 0.1
    W2, b2 = rnd(D, F_, seed=56, scale=0.04).to(bf), rnd(D, seed=57) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev()) if gated else None

    def run(x, xp, odt):
        kw = dict(x_prev=xp if gated else None, gate=gate)
        if train:
            kw.update(h=torch.empty(M, D, device=dev(), dtype=bf), mean=torch.empty(M, device=dev()), rstd=torch.empty(M, device=dev()),
                      gp=torch.empty(M, F_, device=dev(), dtype=bf), u=torch.empty(M, F_, device=dev(), dtype=bf))
        out = torch.full((M, D), float("nan"), device=dev(), dtype=odt)
        nh = torch.full((M, D), float("nan"), device=dev(), dtype=bf)
        nm, nr = torch.empty(M, device=dev()), torch.empty(M, device=dev())
        ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, out, next_gamma=g2, next_beta=b2n, next_h=nh, next_mean=nm, next_rstd=nr, **kw)
        return out, nh, nm, nr, kw

    o16, nh, nm, nr, kw16 = run(xb, xpb, bf)
    o32, _, _, _, kw32 = run(xb.float(), xpb.float(), torch.float32)
    assert torch.equal(o16, o32.to(bf))
    if train:
        for k in ("h", "mean", "rstd", "gp", "u"):
            assert torch.equal(kw16[k], kw32[k]), k
    hb = torch.empty(M, D, device=dev(), dtype=bf)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(o16, g2, b2n, hb, mean, rstd, M, D, BF16)
    torch.testing.assert_close(nm, mean, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(nr, rstd, rtol=2e-6, atol=0)
    diff = (nh.float() - hb.float()).abs()
    assert float(diff.max()) <= 2.0 ** -7 * float(hb.float().abs().max()) + 1e-6
    assert float((diff > 0).float().mean()) < 0.02


@pytest.mark.parametrize("M,gated,F_", [(100864, False, 768), (100864 + 5, True, 768), (16384, False, 768), (16 * 4133 + 9, True, 768), (197 * 1000, False, 768),
                                        (50432, False, 256), (50432 + 3, True, 512), (33000, False, 1024), (20000, False, 192)])
def test_persistent_fused_mlp_equals_the_per_workgroup_kernel_on_row_slices(M, gated, F_):
    """M >= 16384 bf16 rows (inference form) run k_mlp_fused_p: one persistent 8-wave workgroup per CU that walks its 16-row tiles in
    passes and prefetches the next pass's rows.  Rows are independent, so k_mlp_fused_v3 on slices below the threshold must give the
    same out / next_h / next_mean / next_rstd BIT FOR BIT (the engine's batch-independence test relies on exactly that)."""
    from uvc_amd import ops
    D = 192                                               # F_: hidden widths of compacted MLPs too (192 < 256: stays on k_mlp_fused_v3)
    x = (rnd(M, D, seed=71) * 1.5 + 0.2).to(bf)
    xp = rnd(M, D, seed=72).to(bf)
    gamma, beta = rnd(D, seed=73) * 0.2 + 1.0, rnd(D, seed=74) * 0.1
    g2, b2n = rnd(D, seed=75) * 0.3 + 1.0, rnd(D, seed=76) * 0.2
    W1, b1 = rnd(F_, D, seed=77, scale=0.06).to(bf), rnd(F_, seed=78) * 0.1
    W2, b2 = rnd(D, F_, seed=79, scale=0.04).to(bf), rnd(D, seed=80) * 0.1
    gate = torch.tensor([0.3, 0.7], device=dev()) if gated else None

    def run(step):
        out = torch.full((M, D), float("nan"), device=dev(), dtype=bf)
        nh = torch.full((M, D), float("nan"), device=dev(), dtype=bf)
        nm, nr = torch.full((M,), float("nan"), device=dev()), torch.full((M,), float("nan"), device=dev())
        for lo in range(0, M, step):
            hi = min(M, lo + step)
            ops.mlp_fused_fwd(x[lo:hi], gamma, beta, W1, b1, W2, b2, out[lo:hi], next_gamma=g2, next_beta=b2n, next_h=nh[lo:hi],
                              next_mean=nm[lo:hi], next_rstd=nr[lo:hi], x_prev=xp[lo:hi] if gated else None, gate=gate)
        return out, nh, nm, nr

    whole, sliced = run(M), run(12800)
    for a, b in zip(whole, sliced):
        assert bool(torch.isfinite(a.float()).all())
        assert torch.equal(a, b)
    # and without the fused next LayerNorm
    o2 = torch.empty(M, D, device=dev(), dtype=bf)
    ops.mlp_fused_fwd(x, gamma, beta, W1, b1, W2, b2, o2, x_prev=xp if gated else None, gate=gate)
    assert torch.equal(o2, whole[0])
