"""Timing-only variants of k_mlp_fused_p (results are WRONG by construction): what an iteration of the chunk loop is made of.
Edits COPIES of fused_mlp.hip under /tmp; libraries under tools/perturb/ (git-ignored).
    python tools/probe/mlp_p_variants.py build ; on the GPU box: python tools/probe/mlp_p_variants.py run"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VARIANTS = {
    "prio_hi_loaders": [("  const bool loader = w >= P_NW / 2;\n", "  const bool loader = w >= P_NW / 2;\n  if (loader) __builtin_amdgcn_s_setprio(1);\n", 1)],
    "prio_hi_two_tile": [("  const bool loader = w >= P_NW / 2;\n", "  const bool loader = w >= P_NW / 2;\n  if (!loader) __builtin_amdgcn_s_setprio(1);\n", 1)],
    "nogelu": [("acc[r][t][e0] = Gelu<T>::f(acc[r][t][e0]);\n            acc[r][t][e0 + 1] = Gelu<T>::f(acc[r][t][e0 + 1]);", "", 1),
               ("acc[0][t][e] = Gelu<T>::f(acc[0][t][e]);", "", 1)],
    "nodma": [("      dma(std::integral_constant<int, LD ? P_ND : 1>{}, c1, o_nn, c2, o_nxt);\n", "", 1)],
    "nobarrier": [("      __builtin_amdgcn_s_barrier();\n      __builtin_amdgcn_sched_barrier(0);\n      rotate();", "      __builtin_amdgcn_sched_barrier(0);\n      rotate();", 3)],
    "nomfma": [("acc[r][t] = mma(af, hf[r][ks], ks == 0 ? __builtin_bit_cast(f32x4, bi[t]) : acc[r][t]);", "acc[r][t][0] += __builtin_bit_cast(f32x4, af)[0];", 1),
               ("out[r][j] = mma(af, uf[r], out[r][j]);", "out[r][j][0] += __builtin_bit_cast(f32x4, af)[0];", 1)],
}


def build():
    subprocess.check_call([sys.executable, "-m", "uvc_amd.build"], cwd=R)
    src = open(os.path.join(R, "uvc_amd/csrc/fused_mlp.hip")).read()
    src = src.replace('#include "common.h"', '#include "%s/uvc_amd/csrc/common.h"' % R).replace('#include "../../include/uvc_kernels.h"', '#include "%s/include/uvc_kernels.h"' % R)
    k = src.index("void k_mlp_fused_p(")
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result".split()
    objs = [os.path.join(R, "uvc_amd/csrc/build", f) for f in os.listdir(os.path.join(R, "uvc_amd/csrc/build")) if f.endswith(".o") and f != "fused_mlp.o"]
    os.makedirs("/tmp/mlpvar", exist_ok=True)
    for name, edits in VARIANTS.items():
        body = src[k:]
        for old, new, n in edits:
            assert body.count(old) == n, (name, body.count(old), old[:60])
            body = body.replace(old, new)
        open("/tmp/mlpvar/%s.hip" % name, "w").write(src[:k] + body)
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "/tmp/mlpvar/%s.hip" % name, "-o", "/tmp/mlpvar/%s.o" % name])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "tools/perturb/libuvc_hip_mlp_%s.so" % name)] + objs + ["/tmp/mlpvar/%s.o" % name])
        print("built", name)


def build_head():
    """the committed fused_mlp.hip (git HEAD) as tools/perturb/libuvc_hip_mlp_head.so: same-box A/B against the working tree"""
    src = subprocess.check_output(["git", "show", "HEAD:uvc_amd/csrc/fused_mlp.hip"], cwd=R, text=True)
    src = src.replace('#include "common.h"', '#include "%s/uvc_amd/csrc/common.h"' % R).replace('#include "../../include/uvc_kernels.h"', '#include "%s/include/uvc_kernels.h"' % R)
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result".split()
    objs = [os.path.join(R, "uvc_amd/csrc/build", f) for f in os.listdir(os.path.join(R, "uvc_amd/csrc/build")) if f.endswith(".o") and f != "fused_mlp.o"]
    os.makedirs("/tmp/mlpvar", exist_ok=True)
    open("/tmp/mlpvar/head.hip", "w").write(src)
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "/tmp/mlpvar/head.hip", "-o", "/tmp/mlpvar/head.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "tools/perturb/libuvc_hip_mlp_head.so")] + objs + ["/tmp/mlpvar/head.o"])
    print("built head")


def ab():
    for rep in range(3):
        for name in ("good", "head"):
            env = dict(os.environ, PYTHONPATH=R)
            if name != "good":
                env["UVC_LIB"] = os.path.join(R, "tools/perturb/libuvc_hip_mlp_%s.so" % name)
            out = subprocess.run([sys.executable, os.path.join(R, "tools/with_lib.py"), os.path.join(R, "tools/mlp_persist.py"), "time"], env=env, capture_output=True, text=True, cwd=R)
            print("%-10s %s" % ("tree" if name == "good" else name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]))


def run():
    for name in ["good"] + list(VARIANTS):
        env = dict(os.environ, PYTHONPATH=R)
        if name != "good":
            env["UVC_LIB"] = os.path.join(R, "tools/perturb/libuvc_hip_mlp_%s.so" % name)
        out = subprocess.run([sys.executable, os.path.join(R, "tools/with_lib.py"), os.path.join(R, "tools/mlp_persist.py"), "time"], env=env, capture_output=True, text=True, cwd=R)
        print("%-10s %s" % (name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]))


if __name__ == "__main__":
    {"build": build, "head": build_head, "ab": ab, "run": run}[sys.argv[1]]()
