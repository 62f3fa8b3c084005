"""On-device Mixup / CutMix with the interface of ``timm.data.Mixup`` as the reference constructs it
(UVC/joint_train.py:924-933 and UVC/post_train.py:618-621: mixup 0.8, cutmix 1.0, prob, switch_prob 0.5,
mode "batch", label smoothing 0.1) and calls it (``x, y = mixup_fn(x, y)``, joint_train.py:409).

The random decisions (mix or not, mixup vs cutmix, lambda ~ Beta, box centre) are drawn on the host from numpy's
global RNG in the same order as timm, so ``np.random.seed`` reproduces a run; the pixel and target work runs in two HIP
launches, in place.  timm (0.3.2) is not in the image: its published "batch"-mode behaviour is restated in
oracle/mixup.py and checked against this class bit for bit; "pair" / "elem" modes raise.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def rand_bbox(img_shape, lam, margin=0.0, count=None):
    """timm.data.mixup.rand_bbox."""
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y, size=count)
    cx = np.random.randint(0 + margin_x, img_w - margin_x, size=count)
    yl = np.clip(cy - cut_h // 2, 0, img_h)
    yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w)
    xh = np.clip(cx + cut_w // 2, 0, img_w)
    return yl, yh, xl, xh


def rand_bbox_minmax(img_shape, minmax, count=None):
    """timm.data.mixup.rand_bbox_minmax."""
    assert len(minmax) == 2
    img_h, img_w = img_shape[-2:]
    cut_h = np.random.randint(int(img_h * minmax[0]), int(img_h * minmax[1]), size=count)
    cut_w = np.random.randint(int(img_w * minmax[0]), int(img_w * minmax[1]), size=count)
    yl = np.random.randint(0, img_h - cut_h, size=count)
    xl = np.random.randint(0, img_w - cut_w, size=count)
    return yl, yl + cut_h, xl, xl + cut_w


def cutmix_bbox_and_lam(img_shape, lam, ratio_minmax=None, correct_lam=True, count=None):
    """timm.data.mixup.cutmix_bbox_and_lam."""
    if ratio_minmax is not None:
        yl, yu, xl, xu = rand_bbox_minmax(img_shape, ratio_minmax, count=count)
    else:
        yl, yu, xl, xu = rand_bbox(img_shape, lam, count=count)
    if correct_lam or ratio_minmax is not None:
        bbox_area = (yu - yl) * (xu - xl)
        lam = 1.0 - bbox_area / float(img_shape[-2] * img_shape[-1])
    return (yl, yu, xl, xu), lam


class Mixup:
    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch",
                 correct_lam=True, label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha = mixup_alpha
        self.cutmix_alpha = cutmix_alpha
        self.cutmix_minmax = cutmix_minmax
        if self.cutmix_minmax is not None:
            assert len(self.cutmix_minmax) == 2
            self.cutmix_alpha = 1.0
        self.mix_prob = prob
        self.switch_prob = switch_prob
        self.label_smoothing = label_smoothing
        self.num_classes = num_classes
        self.mode = mode
        self.correct_lam = correct_lam
        self.mixup_enabled = True
        if mode != "batch":
            raise NotImplementedError("Mixup mode %r: the reference uses the default 'batch' mode" % (mode,))

    def _params_per_batch(self):
        lam, use_cutmix = 1.0, False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.0:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.0:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
            lam = float(lam_mix)
        return lam, use_cutmix

    def _mix_batch(self, x):
        lam, use_cutmix = self._params_per_batch()
        if lam == 1.0:
            return 1.0
        B, C, H, W = x.shape
        stream = L.cur_stream()
        if use_cutmix:
            (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(x.shape, lam, ratio_minmax=self.cutmix_minmax, correct_lam=self.correct_lam)
            L.check(L.lib().uvc_mixup_batch(L.ptr(x), B, C, H, W, 1.0, 0.0, 1, int(yl), int(yh), int(xl), int(xh), stream), "uvc_mixup_batch")
        else:
            L.check(L.lib().uvc_mixup_batch(L.ptr(x), B, C, H, W, float(np.float32(lam)), float(np.float32(1.0 - lam)), 0, 0, 0, 0, 0, stream),
                    "uvc_mixup_batch")
        return lam

    def __call__(self, x, target):
        assert len(x) % 2 == 0, "Batch size should be even when using this"
        L.require_cuda(x)
        if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4:
            raise L.UvcHipError("Mixup: x must be a contiguous float32 [B, C, H, W] device tensor")
        lam = self._mix_batch(x)
        off_value = self.label_smoothing / self.num_classes
        on_value = 1.0 - self.label_smoothing + off_value
        t = target.to(device=x.device, dtype=torch.int64).contiguous()
        y = torch.empty(x.shape[0], self.num_classes, device=x.device, dtype=torch.float32)
        L.check(L.lib().uvc_mixup_target(L.ptr(t), L.ptr(y), x.shape[0], self.num_classes, float(np.float32(lam)),
                                         float(np.float32(1.0 - lam)), float(np.float32(on_value)), float(np.float32(off_value)),
                                         L.cur_stream()), "uvc_mixup_target")
        return x, y
