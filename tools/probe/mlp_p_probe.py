"""s_memtime stamps inside k_mlp_fused_p (tick = shader cycle).  The product source carries no probe code: `build` edits a COPY of
fused_mlp.hip under /tmp and links tools/perturb/libuvc_hip_mlpprobe.so (git-ignored; travels with gpurun); `run` (GPU box) prints where
a wave's cycles go per pass: rows, each chunk iteration, the last fc2, the stores.
    python tools/probe/mlp_p_probe.py build
    gpurun -- 'UVC_LIB=tools/perturb/libuvc_hip_mlpprobe.so python tools/with_lib.py tools/probe/mlp_p_probe.py run'"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NS = 64


def once(s, old, new, count=1):
    assert s.count(old) == count, (s.count(old), old[:80])
    return s.replace(old, new)


def build():
    subprocess.check_call([sys.executable, "-m", "uvc_amd.build"], cwd=R)
    s = open(os.path.join(R, "uvc_amd/csrc/fused_mlp.hip")).read()
    s = once(s, '#include "common.h"', '#include "%s/uvc_amd/csrc/common.h"' % R)
    s = once(s, '#include "../../include/uvc_kernels.h"', '#include "%s/include/uvc_kernels.h"' % R)
    s = once(s, "namespace {\n", "__device__ unsigned long long* g_st;\nextern \"C\" void uvc_mlp_probe_set(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st), &p, sizeof(p)); }\n"
             "#define ST(slot) do { if (lane == 0) g_st[((size_t)blockIdx.x * 8 + w) * %d + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)\nnamespace {\n" % NS)
    k = s.index("void k_mlp_fused_p(")
    head, body = s[:k], s[k:]
    body = once(body, "  const int nch = a.F / V3_FC;\n", "  const int nch = a.F / V3_FC;\n  int pidx = 0;\n  ST(0);\n")
    body = once(body, "rows landed\n", "rows landed\n  ST(1);\n")
    body = once(body, "    auto fc1 = [&](int c) {\n      const unsigned a0", "    ST(2 + pidx * 30);\n    auto fc1 = [&](int c) {\n      const unsigned a0")
    body = once(body, "      rotate();\n    }\n#pragma nounroll\n    for (int c = 1; c < 4; ++c) {", "      rotate();\n      ST(2 + pidx * 30 + 1);\n    }\n#pragma nounroll\n    for (int c = 1; c < 4; ++c) {")
    body = once(body, "      rotate();\n    }\n#pragma nounroll\n    for (int c = 4;", "      rotate();\n      ST(2 + pidx * 30 + 1 + c);\n    }\n#pragma nounroll\n    for (int c = 4;")
    body = once(body, "      rotate();\n    }\n    fc2_gelu(std::true_type{}, std::false_type{});\n", "      rotate();\n      ST(2 + pidx * 30 + 1 + c);\n    }\n    fc2_gelu(std::true_type{}, std::false_type{});\n    ST(2 + pidx * 30 + 25);\n")
    # inside iteration 10 of every pass (the last pass's values stay): after the requests, after fc1, after fc2 + GELU, after the vmcnt wait
    body = once(body, "      request(c);\n      __builtin_amdgcn_sched_barrier(0);\n      fc1(c);\n      fc2_gelu(std::true_type{}, std::true_type{});\n      if (LD && c == 4)",
                "      if (c == 10) ST(59);\n      request(c);\n      __builtin_amdgcn_sched_barrier(0);\n      if (c == 10) ST(60);\n      fc1(c);\n      if (c == 10) ST(61);\n"
                "      fc2_gelu(std::true_type{}, std::true_type{});\n      if (c == 10) ST(62);\n      if (LD && c == 4)")
    body = once(body, "else wait_vm<LD ? P_ND : 1>();\n", "else wait_vm<LD ? P_ND : 1>();\n      if (c == 10) ST(63);\n")
    body = once(body, "    __builtin_amdgcn_sched_barrier(0);\n  };\n\n  for (int p = 0; p < npass; ++p) {", "    __builtin_amdgcn_sched_barrier(0);\n    ST(2 + pidx * 30 + 26);\n    ++pidx;\n  };\n\n  for (int p = 0; p < npass; ++p) {")
    os.makedirs("/tmp/mlpprobe", exist_ok=True)
    open("/tmp/mlpprobe/fused_mlp.hip", "w").write(head + body)
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", "/tmp/mlpprobe/fused_mlp.hip", "-o", "/tmp/mlpprobe/fused_mlp.o"])
    objs = [os.path.join(R, "uvc_amd/csrc/build", f) for f in os.listdir(os.path.join(R, "uvc_amd/csrc/build")) if f.endswith(".o") and f != "fused_mlp.o"]
    os.makedirs(os.path.join(R, "tools/perturb"), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "tools/perturb/libuvc_hip_mlpprobe.so")] + objs + ["/tmp/mlpprobe/fused_mlp.o"])
    print("built tools/perturb/libuvc_hip_mlpprobe.so")


def run(M=100864):
    import ctypes
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(R, "tools"))
    import mlp_persist as MP
    from uvc_amd import _lib as L
    t = MP.make(M)
    o = MP.bufs(M)
    st = torch.zeros(256 * 8 * NS, dtype=torch.int64, device="cuda")
    L.lib().uvc_mlp_probe_set.argtypes = [ctypes.c_void_p]
    L.lib().uvc_mlp_probe_set(ctypes.c_void_p(st.data_ptr()))
    for _ in range(3):
        MP.run(t, 0, M, *o)
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 8, NS).astype(np.float64)
    t0 = s[:, :, 0].min()
    def med(x):
        return "%7.0f" % np.median(x)
    print("ticks (= shader cycles), median over 256 workgroups; waves 0-3 run two tiles per pass, 4-7 one")
    for wv in range(8):
        a = s[:, wv, :]
        line = ["wave %d: start %s  prologue %s |" % (wv, med(a[:, 0] - t0), med(a[:, 1] - a[:, 0]))]
        prev = a[:, 1]
        for p in range(2):
            b = 2 + p * 30
            rows = a[:, b] - prev
            its = [a[:, b + 1 + c] - a[:, b + c] for c in range(24)]
            fin = a[:, b + 25] - a[:, b + 24]
            ep = a[:, b + 26] - a[:, b + 25]
            line.append(" pass %d: rows %s it0 %s it1 %s it2 %s it3-23 mean %s (min %s max %s) last fc2 %s stores %s |" % (
                p, med(rows), med(its[0]), med(its[1]), med(its[2]), med(np.mean(its[3:], axis=0)), med(np.min(its[3:], axis=0)), med(np.max(its[3:], axis=0)), med(fin), med(ep)))
            prev = a[:, b + 26]
        line.append(" total %s" % med(prev - a[:, 0]))
        ref = s[:, 0, 59]
        line.append("\n        iteration 10 of the last pass (times relative to wave 0's loop top): top %s | requests %s fc1 %s fc2+gelu %s vmcnt wait %s -> at barrier %s, released %s" % (
            med(a[:, 59] - ref), med(a[:, 60] - a[:, 59]), med(a[:, 61] - a[:, 60]), med(a[:, 62] - a[:, 61]), med(a[:, 63] - a[:, 62]), med(a[:, 63] - ref), med(a[:, 2 + 30 + 11] - ref)))
        print("".join(line))
    print("kernel span over all waves: %.0f ticks" % (s[:, :, 2 + 30 + 26].max() - t0))


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
