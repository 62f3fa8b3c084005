"""GPU: the two layouts of the training workspace (uvc_vit_workspace_bytes' `training` = 1 / 2, uvc_vit_io.shared_bwd_streams).

Mode 1 (two-stream backward) keeps per-block copies of the backward's streams dL/dx_l, dL/dx1, dA, dqkv so that weight gradients on the side stream
can read a block's streams while the main stream is blocks ahead; mode 2 (no side stream) carves ONE shared set, with dL/dx alternating between two
buffers by the number of blocks that ran.  Checked here: the serialized backward on the shared set gives the same gradients BIT FOR BIT as the
two-stream backward on the per-block copies -- with soft block gating, with hard-skipped blocks (Stage-2 style: the ping-pong must hop over them),
with the distillation token, in both precisions and with a staged backward (DDP's bucket cuts) -- that it is smaller by L x (5 M D + M F) elements,
and that the engine refuses a side stream on it."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(precision, gating, enable_dist=0, depth=12):
    from uvc_amd.model_distilled import DistilledVisionTransformer
    torch.manual_seed(11)
    m = DistilledVisionTransformer(enable_dist, enable_block_gating=gating, embed_dim=192, depth=depth, num_heads=3, precision=precision).cuda()
    m.train()
    if not gating:
        with torch.no_grad():
            m.block_skip_gating.data[:, 0] = 0.0
            m.block_skip_gating.data[:, 1] = 1.0
            for l in (3, 4, 9):                                          # two neighbours and a single one skipped: the ping-pong hops over them
                m.block_skip_gating.data[l] = torch.tensor([1.0, 0.0])
    else:
        m.enable_warmup = 0
        e = torch.empty(depth, 2, device="cuda").exponential_(generator=torch.Generator(device="cuda").manual_seed(9))
        m.exp_source = lambda shape, e=e: e.clone()
    return m


def _grads(m, x, two_stream):
    m.two_stream_backward = two_stream
    m._flat_grad.zero_()
    out = m(x, -1, 0.9)[0]
    logits = [o for o in (out if isinstance(out, (tuple, list)) else [out]) if o is not None]
    loss = sum((o.float() ** 2).mean() for o in {id(o): o for o in logits}.values())
    loss.backward()
    torch.cuda.synchronize()
    return m._flat_grad[:m._off.n_total].clone(), [o.detach().clone() for o in logits]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["soft_gating", "hard_skip", "dist_token"])
def test_shared_backward_streams_equal_the_per_block_copies(precision, case):
    m = _model(precision, gating=int(case != "hard_skip"), enable_dist=int(case == "dist_token"))
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(6, 3, 224, 224, device="cuda", generator=g)
    g2, o2 = _grads(m, x, True)
    g1, o1 = _grads(m, x, False)
    assert float(g2.abs().max()) > 0 and torch.isfinite(g2).all()
    assert all(torch.equal(a, b) for a, b in zip(o1, o2))
    assert torch.equal(g1, g2)
    if case == "hard_skip":                                               # no gradient reaches a skipped block
        o = m._off
        assert float(g1[o.blk[4][0]:o.blk[5][0]].abs().max()) == 0.0


def test_shared_set_is_smaller_and_refuses_a_side_stream():
    from uvc_amd import _lib as L
    from uvc_amd import model_distilled as MD
    m = _model("bf16", gating=1)
    lib = MD._bind()
    B = 8
    n0, n1, n2 = (lib.uvc_vit_workspace_bytes(C.byref(m._cfg), B, mode) for mode in (0, 1, 2))
    cfg = m._cfg
    M, D, F, Lb = B * 197, cfg.embed_dim, cfg.hidden, cfg.depth
    saved = n1 - n2
    # L + 1 dL/dx_l, L dL/dx1, L dA, L dqkv per-block against 2 + 1 + 1 + 1 shared buffers (256-byte aligned each)
    expect = ((Lb + 1 - 2) * M * D + (Lb - 1) * (M * D + M * F + 3 * M * D)) * 2
    assert 0 < n0 < n2 < n1 and abs(saved - expect) <= 256 * 4 * (Lb + 1), (saved, expect)
    # the engine refuses the combination instead of letting the side stream read a buffer the main stream has moved on from
    x = torch.randn(B, 3, 224, 224, device="cuda")
    m.two_stream_backward = False
    out = m(x, -1, 0.9)[0]
    st = m._last
    io = m._io(B, True)
    io.patches_in = L.ptr(st.get("patches"))
    d = torch.zeros(B, cfg.num_classes, device="cuda")
    io.d_logits = L.ptr(d)
    io.gate_d = L.ptr(st["gate_d"])
    io.side_stream = L.side_stream(x.device).cuda_stream
    rc = lib.uvc_vit_backward(C.byref(cfg), C.byref(io), L.cur_stream())
    assert rc != 0 and b"shared_bwd_streams" in L.lib().uvc_last_error()
    # and Python refuses a flag flipped between a forward and its backward
    m.two_stream_backward = True
    with pytest.raises(RuntimeError, match="two_stream_backward"):
        m._run_backward(d, None)
