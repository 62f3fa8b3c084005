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


def test_ctypes_mirrors_have_the_layout_of_the_c_structs(tmp_path):
    """Every ctypes.Structure the Python host passes through the C-ABI (uvc_amd/_lib.py, model_distilled.py) against the struct of the same name in
    include/*.h: gcc compiles a probe that prints sizeof and the offsetof of every field the mirror declares -- a field added to a header but not
    to its mirror (or in another place) shifts everything behind it, and nothing but a GPU run would notice."""
    import subprocess
    from uvc_amd import _lib, model_distilled as MD
    mirrors = [getattr(_lib, n) for n in dir(_lib) if n.startswith("uvc_") and isinstance(getattr(_lib, n), type) and issubclass(getattr(_lib, n), ctypes.Structure)]
    mirrors += [getattr(MD, n) for n in ("uvc_vit_cfg", "uvc_vit_offsets", "uvc_vit_shadow_offsets", "uvc_vit_io", "uvc_mlp_compact")]
    assert len(mirrors) >= 15
    lines = ['#include <stdio.h>', '#include <stddef.h>']
    lines += [f'#include "{os.path.basename(h)}"' for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))]
    lines.append("int main(void) {")
    for m in mirrors:
        lines.append(f'  printf("{m.__name__} sizeof %zu\\n", sizeof({m.__name__}));')
        for f in m._fields_:
            lines.append(f'  printf("{m.__name__} {f[0]} %zu\\n", offsetof({m.__name__}, {f[0]}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    r = subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out.splitlines()}
    for m in mirrors:
        assert got[(m.__name__, "sizeof")] == ctypes.sizeof(m), (m.__name__, got[(m.__name__, "sizeof")], ctypes.sizeof(m))
        for f in m._fields_:
            assert got[(m.__name__, f[0])] == getattr(m, f[0]).offset, (m.__name__, f[0])
