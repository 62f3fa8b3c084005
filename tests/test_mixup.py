"""Mixup / CutMix.  CPU: known answers of the oracle restatement of timm's "batch" mode.  GPU: the HIP path
(uvc_amd/mixup.py -> uvc_mixup_batch / uvc_mixup_target) against the oracle, bit for bit, with the same numpy RNG stream."""
import numpy as np
import pytest
import torch

from oracle import mixup as OM


def test_oracle_mixup_known_answers():
    torch.manual_seed(0)
    x = torch.randn(6, 3, 32, 32)
    t = torch.tensor([1, 2, 3, 4, 5, 6])
    # prob 0: untouched images, smoothed one-hot targets
    np.random.seed(1)
    x0 = x.clone()
    xo, y, lam, cm = OM.mixup_batch(x0, t, prob=0.0, num_classes=10, label_smoothing=0.1)
    assert lam == 1.0 and not cm and torch.equal(xo, x)
    assert torch.allclose(y.sum(1), torch.ones(6)) and abs(float(y[0, 1]) - (0.9 + 0.01)) < 1e-7 and abs(float(y[0, 0]) - 0.01) < 1e-7
    # mixup only: convex combination with the flipped batch, targets mixed with the same lambda
    np.random.seed(2)
    x1 = x.clone()
    xo, y, lam, cm = OM.mixup_batch(x1, t, mixup_alpha=0.8, cutmix_alpha=0.0, num_classes=10, label_smoothing=0.0)
    assert 0 < lam < 1 and not cm
    assert torch.allclose(xo, x * lam + x.flip(0) * (1 - lam), atol=1e-6)
    assert abs(float(y[0, 1]) - lam) < 1e-6 and abs(float(y[0, 6]) - (1 - lam)) < 1e-6
    # cutmix only: a box of the flipped batch, lambda corrected to the box area
    np.random.seed(3)
    x2 = x.clone()
    xo, y, lam, cm = OM.mixup_batch(x2, t, mixup_alpha=0.0, cutmix_alpha=1.0, num_classes=10, label_smoothing=0.0)
    assert cm
    changed = (xo != x).any(dim=1).any(dim=0)                      # [H, W] pixels that differ somewhere
    area = int(changed.sum())
    assert abs((1 - lam) * 32 * 32 - area) < 1e-6
    ys, xs = torch.nonzero(changed, as_tuple=True)
    assert area == (ys.max() - ys.min() + 1) * (xs.max() - xs.min() + 1)   # one rectangle
    assert torch.equal(xo[:, :, ys.min():ys.max() + 1, xs.min():xs.max() + 1], x.flip(0)[:, :, ys.min():ys.max() + 1, xs.min():xs.max() + 1])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
def test_hip_mixup_matches_oracle_bit_for_bit(seed):
    from uvc_amd.mixup import Mixup
    B, C, S, NC = 8, 3, 64, 100
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.randn(B, C, S, S, generator=g)
    t = torch.randint(0, NC, (B,), generator=g)
    kw = dict(mixup_alpha=0.8, cutmix_alpha=1.0, prob=0.8, switch_prob=0.5, label_smoothing=0.1, num_classes=NC)
    np.random.seed(seed)
    xr, yr, lam, cm = OM.mixup_batch(x.clone(), t, **kw)
    np.random.seed(seed)
    fn = Mixup(mode="batch", **kw)
    xd, yd = fn(x.clone().cuda(), t.cuda())
    assert torch.equal(xd.cpu(), xr), (lam, cm)
    assert torch.equal(yd.cpu(), yr), (lam, cm)
    assert abs(float(yd.sum()) - B) < 1e-4


@pytest.mark.gpu
def test_hip_mixup_reference_configuration_at_image_size():
    """joint_train.py:924-933 arguments at 224 x 224: both branches occur over a few seeds; odd batches are rejected."""
    from uvc_amd.mixup import Mixup
    fn = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch", label_smoothing=0.1, num_classes=1000)
    seen = set()
    for seed in range(6):
        x = torch.randn(16, 3, 224, 224, generator=torch.Generator().manual_seed(seed))
        t = torch.randint(0, 1000, (16,), generator=torch.Generator().manual_seed(seed + 50))
        np.random.seed(seed)
        xr, yr, lam, cm = OM.mixup_batch(x.clone(), t, prob=1.0)
        np.random.seed(seed)
        xd, yd = fn(x.clone().cuda(), t.cuda())
        assert torch.equal(xd.cpu(), xr) and torch.equal(yd.cpu(), yr)
        seen.add(bool(cm))
    assert seen == {True, False}
    with pytest.raises(AssertionError):
        fn(torch.randn(3, 3, 32, 32).cuda(), torch.zeros(3, dtype=torch.long).cuda())
    with pytest.raises(NotImplementedError):
        Mixup(mode="elem")
