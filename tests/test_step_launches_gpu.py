"""GPU: the UVC-train step launches no ATen elementwise kernels.

Round 3's step carried five (`(o + od) / 2` of the teacher's eval forward, the ones_like(loss) fill and the multiply by it in the loss backward,
the sqrt of clip_grad_norm_'s return value) plus three device copies for the per-step snapshot.  What is left: two copies (the snapshot of
s r y p z + resource, the gate logits) -- the step's values live on the device until somebody reads them (reference: joint_train.py:404-447,
utils/losses.py:27-62, uvc_optimizer.py:138-144)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_step_launches_no_aten_elementwise_kernels():
    from torch.profiler import ProfilerActivity, profile

    import bench
    from uvc_amd.stage1 import Stage1Trainer, default_args

    a = default_args(model_type="deit_tiny_patch16_224", precision="bf16", train_batch_size=8, local_rank=0)
    tr = Stage1Trainer(a, device="cuda:0", distributed=False)
    bench.pruned_state(tr)
    tr.begin_epoch(a.warmup_epochs + 1)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(8, 3, a.img_size, a.img_size, device="cuda", generator=g)
    y = torch.softmax(torch.randn(8, a.num_classes, device="cuda", generator=g), -1)
    for _ in range(3):
        out = tr.step(x, y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = tr.step(x, y)
        torch.cuda.synchronize()
    ops = [ev.name for ev in prof.events()
           if ev.name.startswith("aten::") and ev.device_time_total > 0 and not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children)]
    assert sorted(ops) == ["aten::copy_", "aten::copy_"], ops
    # the values the removed launches produced are still there: the clip norm is the square root of the summed squares, the loss is finite
    st = tr.model._clip
    torch.testing.assert_close(out["gnorm"], st["sq"][0].sqrt(), rtol=1e-6, atol=0)
    assert torch.isfinite(out["loss"]) and float(out["gnorm"]) > 0


def test_loss_backward_with_and_without_the_unit_gradient_agree():
    """loss.backward(unit_gradient) (the trainers) skips the multiply by d(loss) = 1; a plain loss.backward() and a scaled loss take the general
    path: same gradients (x 1, x 0.5)."""
    from uvc_amd.losses import _LossFunction, unit_gradient

    g = torch.Generator(device="cuda").manual_seed(5)
    o = torch.randn(16, 100, device="cuda", generator=g)
    y = torch.softmax(torch.randn(16, 100, device="cuda", generator=g), -1)
    grads = []
    for mode in ("unit", "plain", "half"):
        oo = o.clone().requires_grad_(True)
        loss = _LossFunction.apply(oo, oo, y, None, 0.0, 1.0, 0)
        if mode == "unit":
            loss.backward(unit_gradient(loss.device))
        elif mode == "plain":
            loss.backward()
        else:
            (loss * 0.5).backward()
        grads.append(oo.grad.clone())
    assert torch.equal(grads[0], grads[1])
    torch.testing.assert_close(grads[2], grads[0] * 0.5, rtol=1e-6, atol=0)


def test_step_with_the_next_batch_starts_its_teacher_forward_early_and_changes_nothing():
    """Stage1Trainer.step(x, y, next_x=...) starts the frozen teacher's forward for the NEXT batch behind this step's backward (it runs under the optimizer / UVC
    tail); the next step picks it up only for that very tensor.  Same losses, same weights, bit for bit, as steps without it -- also when the promise is broken
    (another tensor arrives) -- and exactly one teacher forward per step (the pending one is consumed)."""
    import bench
    from uvc_amd.stage1 import Stage1Trainer, default_args

    g = torch.Generator(device="cuda").manual_seed(3)
    xs = [torch.randn(8, 3, 224, 224, device="cuda", generator=g) for _ in range(4)]
    ys = [torch.softmax(torch.randn(8, 1000, device="cuda", generator=g), -1) for _ in range(4)]

    def run(mode):
        torch.manual_seed(0)                                            # (the trainer draws its initial weights from torch's generator)
        a = default_args(model_type="deit_tiny_patch16_224", precision="bf16", train_batch_size=8, local_rank=0)
        tr = Stage1Trainer(a, device="cuda:0", distributed=False)
        bench.pruned_state(tr)
        tr.begin_epoch(a.warmup_epochs + 1)
        calls = [0]
        fwd = tr.criterion.teacher_model.forward
        tr.criterion.teacher_model.forward = lambda *aa, **kk: (calls.__setitem__(0, calls[0] + 1), fwd(*aa, **kk))[1]
        losses = []
        if mode == "lookahead":
            for (x, y), nx in tr.lookahead(zip(xs, ys)):
                losses.append(tr.step(x, y, next_x=nx)["loss"].clone())
        else:
            for i, (x, y) in enumerate(zip(xs, ys)):
                nx = None if mode == "plain" else xs[0]                 # "broken": always promises batch 0
                losses.append(tr.step(x, y, next_x=nx)["loss"].clone())
        torch.cuda.synchronize()
        return torch.stack(losses), tr.model._flat.clone(), calls[0], tr.criterion._pref

    l0, w0, c0, p0 = run("plain")
    l1, w1, c1, p1 = run("lookahead")
    l2, w2, c2, p2 = run("broken")
    assert torch.equal(l0, l1) and torch.equal(w0, w1) and torch.equal(l0, l2) and torch.equal(w0, w2)
    assert c0 == 4 and c1 == 4 and p0 is None and p1 is None
    assert c2 == 8 and p2 is not None                                   # every promise missed: 4 forwards wasted, results untouched
