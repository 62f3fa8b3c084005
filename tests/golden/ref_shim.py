"""Test-only shim that lets the reference's LEAF modules be imported on CPU in the build container.

It never runs on the GPU box (``/root/reference`` does not exist there) and is used only by
``tests/golden/make_golden.py`` (the committed generator of the fixtures in this directory) and
by the optional ``test_oracle_vs_reference`` tests, which skip when the reference tree is absent.

What it does:
  1. registers stub modules for the third-party imports the reference makes but the image
     lacks (timm pieces, torchvision -- imported-but-unused at UVC/uvc_optimizer.py:12-13);
  2. makes ``.cuda()`` the identity on Tensor / Module (hard-coded ``.cuda()`` calls,
     UVC/uvc_utils.py:162,166,180 ...; UVC/models/model_distilled.py:29,40,480,483);
  3. puts ``/root/reference/UVC`` on ``sys.path``.
No reference source is copied: the reference modules are imported from where they lie.
"""
import os
import sys
import types

REF_ROOT = "/root/reference/UVC"


def available() -> bool:
    return os.path.isdir(REF_ROOT)


def install():
    import torch
    import torch.nn as nn

    if "timm" not in sys.modules:
        def mod(name):
            m = types.ModuleType(name)
            sys.modules[name] = m
            return m

        timm = mod("timm")
        models = mod("timm.models")
        vt = mod("timm.models.vision_transformer")
        reg = mod("timm.models.registry")
        layers = mod("timm.models.layers")
        lhelpers = mod("timm.models.layers.helpers")
        mhelpers = mod("timm.models.helpers")
        timm.models = models
        models.vision_transformer = vt
        models.registry = reg
        models.layers = layers
        models.helpers = mhelpers
        layers.helpers = lhelpers

        vt._cfg = lambda **kw: dict(kw)
        reg.register_model = lambda fn: fn
        layers.trunc_normal_ = nn.init.trunc_normal_

        class DropPath(nn.Identity):  # drop_path is 0 everywhere on this path
            def __init__(self, p=0.0):
                super().__init__()

        layers.DropPath = DropPath

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        layers.to_2tuple = to_2tuple
        lhelpers.to_2tuple = to_2tuple
        mhelpers.load_pretrained = lambda *a, **k: None

        tv = mod("torchvision")
        tv.datasets = mod("torchvision.datasets")
        tv.transforms = mod("torchvision.transforms")

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


class RecordExponential:
    """Context manager recording every ``Tensor.exponential_`` draw (the only RNG the path uses:
    F.gumbel_softmax and the reference's own gumbel_softmax, model_distilled.py:40)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.exponential_
        rec = self

        def patched(t, *a, **k):
            out = rec._orig(t, *a, **k)
            rec.draws.append(out.detach().clone())
            return out

        torch.Tensor.exponential_ = patched
        return self

    def __exit__(self, *exc):
        import torch
        torch.Tensor.exponential_ = self._orig
        return False
