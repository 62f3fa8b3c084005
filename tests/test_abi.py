"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares
(no compute calls without a GPU)."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"^\s*(?:int|int64_t|const char\*)\s+(uvc_\w+)\s*\(", src, flags=re.M):
            names.append(m.group(1))
    return names


def test_library_builds_and_exports_every_declared_symbol():
    from uvc_amd import build
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported by libuvc_hip.so"


def test_binding_table_covers_the_header():
    from uvc_amd import _lib
    assert sorted(_lib.exported_symbols()) == sorted(declared_symbols())


def test_product_path_refuses_cpu_tensors():
    import torch
    from uvc_amd import _lib
    with pytest.raises(_lib.UvcHipError):
        _lib.require_cuda(torch.zeros(3))


def test_models_refuse_to_run_without_a_gpu():
    """No CPU fallback: constructing a model on the CPU raises (DeiT and T2T-ViT mirrors alike)."""
    from uvc_amd import _lib
    from uvc_amd.model_distilled import DistilledVisionTransformer
    from uvc_amd.t2t_vit import T2T_ViT
    with pytest.raises(_lib.UvcHipError):
        DistilledVisionTransformer(0, embed_dim=128, depth=1, num_heads=2, img_size=32, num_classes=8, device="cpu")
    with pytest.raises(_lib.UvcHipError):
        T2T_ViT(img_size=32, embed_dim=128, depth=1, num_heads=2, num_classes=8, device="cpu")


def test_t2t_host_side_tables():
    """Host logic of the T2T mirror that needs no GPU: the sinusoid table equals the oracle's, and the tokens-to-token
    parameter holders carry the reference's names and shapes (T2TViT/models/token_performer.py:8-29, t2t_vit.py:46-82)."""
    import torch
    from oracle import t2t as OT
    from uvc_amd.t2t_vit import T2T_module, get_sinusoid_encoding
    assert torch.equal(get_sinusoid_encoding(197, 384), OT.sinusoid_encoding(197, 384))
    m = T2T_module(img_size=224, embed_dim=384)
    cfg = OT.T2TConfig()
    want = {k[len("tokens_to_token."):]: v for k, v in OT.param_shapes(cfg).items() if k.startswith("tokens_to_token.")}
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want and m.num_patches == 196
