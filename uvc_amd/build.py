"""Build libuvc_hip.so (all HIP kernels + the C-ABI) for gfx950 with hipcc, in-tree.

    python -m uvc_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under uvc_amd/csrc/build/ by mtime.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libuvc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: no SLP vectorisation of scalar float math into packed-fp32 instructions.  hipcc turned the gate mix
# `d1*v + d0*r` into v_pk_fma_f32 with op_sel-swapped halves written in place over its own addend (dest == src2); on
# gfx950 that form returned wrong low halves in lanes 48-63, non-deterministically (k_gemm_wsn<float, GATE, 18>, caught
# by tests/test_kernels_gpu.py).  Plain v_fma_f32 is exact, and the kernels are HBM-bound, so all files are built this way.
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-value", "-Wno-unused-result",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]
# the scalar primal-dual update must round like the reference's separate float32 ops
PER_FILE = {"uvc_engine.hip": ["-ffp-contract=off"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime())):
        return obj, False
    cmd = [HIPCC] + COMMON + PER_FILE.get(src, []) + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"linked {LIB} from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
