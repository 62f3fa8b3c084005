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
