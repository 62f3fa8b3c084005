"""Oracle: timm.data.Mixup in "batch" mode (the step immediately before the hot path: joint_train.py:409,
post_train.py:362) restated with torch on the CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: timm (pinned 0.3.2, Baseline_pruning/requirements.txt:3) is not in the image and the reference holds
no golden vectors for it; this follows timm's published mixup.py (Mixup._params_per_batch, _mix_batch, mixup_target,
rand_bbox, cutmix_bbox_and_lam).  The random draws come from numpy's global RNG in timm's order.
"""
from __future__ import annotations

import numpy as np
import torch


def one_hot(x, num_classes, on_value=1.0, off_value=0.0):
    x = x.long().view(-1, 1)
    return torch.full((x.size()[0], num_classes), off_value).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0):
    off_value = smoothing / num_classes
    on_value = 1.0 - smoothing + off_value
    y1 = one_hot(target, num_classes, on_value=on_value, off_value=off_value)
    y2 = one_hot(target.flip(0), num_classes, on_value=on_value, off_value=off_value)
    return y1 * lam + y2 * (1.0 - lam)


def rand_bbox(img_shape, lam):
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    cy = np.random.randint(0, img_h)
    cx = np.random.randint(0, img_w)
    yl = np.clip(cy - cut_h // 2, 0, img_h); yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w); xh = np.clip(cx + cut_w // 2, 0, img_w)
    return yl, yh, xl, xh


def mixup_batch(x, target, mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000,
                correct_lam=True):
    """One ``mixup_fn(x, target)`` call; x is modified in place like timm does.  Returns (x, y_soft, lam, use_cutmix)."""
    assert len(x) % 2 == 0
    lam, use_cutmix = 1.0, False
    if np.random.rand() < prob:
        if mixup_alpha > 0.0 and cutmix_alpha > 0.0:
            use_cutmix = np.random.rand() < switch_prob
            lam = float(np.random.beta(cutmix_alpha, cutmix_alpha) if use_cutmix else np.random.beta(mixup_alpha, mixup_alpha))
        elif mixup_alpha > 0.0:
            lam = float(np.random.beta(mixup_alpha, mixup_alpha))
        else:
            use_cutmix = True
            lam = float(np.random.beta(cutmix_alpha, cutmix_alpha))
    if lam != 1.0:
        if use_cutmix:
            yl, yh, xl, xh = rand_bbox(x.shape, lam)
            if correct_lam:
                lam = 1.0 - (yh - yl) * (xh - xl) / float(x.shape[-2] * x.shape[-1])
            x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
        else:
            x_flipped = x.flip(0).mul_(1.0 - lam)
            x.mul_(lam).add_(x_flipped)
    return x, mixup_target(target, num_classes, lam, label_smoothing), lam, use_cutmix
