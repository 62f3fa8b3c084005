#!/bin/bash
# What each kernel of the DeiT-Tiny step is worth IN the step: the step timed with that kernel's launches skipped (wrong numbers on purpose, same-box alternating
# A/B against the real library).  The streams overlap, so a kernel's stand-alone time says little about what removing it would return ("What the step
# time is made of"): the teacher's qkv returns its whole stand-alone time, the weight-gradient reduce a sixth of it (tools/bound_probes.sh).
# The variants are built from a sed-edited COPY of vit_engine.hip: a PROBE_* macro in front of the launch helpers, set per variant with -D.
#   here:            tools/marginal_probes.sh build   -> tools/perturb/libuvc_hip_m_<name>.so
#   on the GPU box:  tools/marginal_probes.sh run      -> gpurun_out/marginal_probes.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result -I$R/include -I$R/uvc_amd/csrc"
S='c.io->training'; D='c.d.D'; F='c.d.F'
declare -A V
# (r6: the qkv Linear + attention forward of every block but the last are ONE kernel since r5, k_qkv_attn_fwd)
V[s_qkvattn]="-DPROBE_QA=(io->training)"
V[t_qkvattn]="-DPROBE_QA=(!io->training)"
V[s_proj]="-DPROBE_NT=($S&&N==$D&&K==$D&&epi==UVC_EPI_BIAS_RESID)"
V[t_proj]="-DPROBE_NT=(!$S&&N==$D&&K==$D&&epi==UVC_EPI_BIAS_RESID)"
V[s_fc1]="-DPROBE_NT=(epi==UVC_EPI_BIAS_GELU_GRAD||epi==UVC_EPI_BIAS_GELU_GRAD_Q8)"
V[s_fc2]="-DPROBE_NT=($S&&K==$F&&N==$D&&(epi==UVC_EPI_BIAS_RESID||epi==UVC_EPI_BIAS_RESID_GATE))"
V[t_mlp]="-DPROBE_MLP=(!io->training)"
V[dfc2]="-DPROBE_NT=(epi==UVC_EPI_MUL_AUX||epi==UVC_EPI_MUL_AUX_Q8)"
V[dfc1_ln]="-DPROBE_LNB=(K==$F)"
V[dproj]="-DPROBE_NT=($S&&epi==UVC_EPI_NONE&&N==$D&&K==$D&&M==c.d.M)"
V[attn_bwd]="-DPROBE_ATTN=(bwd)"
V[dqkv_ln]="-DPROBE_LNB=(K==3*$D)"
V[wgrads]="-DPROBE_TN=1"
ORDER="s_qkvattn s_proj s_fc1 s_fc2 t_qkvattn t_proj t_mlp dfc2 dfc1_ln dproj attn_bwd dqkv_ln wgrads"
if [ "${1:-}" = build ]; then
  python -m uvc_amd.build > /dev/null || exit 1
  mkdir -p /tmp/perturb "$R/tools/perturb"
  python3 - "$R/uvc_amd/csrc/vit_engine.hip" /tmp/perturb/vit_probe.hip "$R" <<'PY'
import sys
s = open(sys.argv[1]).read()
R = sys.argv[3]
def ins(after, text):
    global s
    assert s.count(after) == 1, after
    s = s.replace(after, after + text)
s = s.replace('#include "common.h"', '#include "%s/uvc_amd/csrc/common.h"\n#ifndef PROBE_NT\n#define PROBE_NT 0\n#endif\n#ifndef PROBE_TN\n#define PROBE_TN 0\n#endif\n#ifndef PROBE_ATTN\n#define PROBE_ATTN 0\n#endif\n#ifndef PROBE_LNB\n#define PROBE_LNB 0\n#endif\n#ifndef PROBE_MLP\n#define PROBE_MLP 0\n#endif\n#ifndef PROBE_QA\n#define PROBE_QA 0\n#endif' % R)
s = s.replace('#include "../../include/', '#include "%s/include/' % R)
ins('       const float* alpha_ptr = nullptr, int lda = 0, int ldc = 0, const NextLn* ln = nullptr) {\n', '  if (PROBE_NT) return UVC_OK;\n')
ins('       int lda = 0, int ldb = 0, bool scratch = false) {\n', '  if (PROBE_TN) return UVC_OK;\n')
ins('                 const void* add1, const float* a1, const void* add2, const float* a2, float* dots) {\n', '  if (PROBE_LNB) return UVC_OK;\n')
ins('int attn(const Ctx& c, const BlockBufs& b, bool bwd, int layer = -1) {\n', '  if (PROBE_ATTN) return UVC_OK;\n')
assert s.count('      TRY(uvc_qkv_attention_fwd(&qa, c.st));') == 1
s = s.replace('      TRY(uvc_qkv_attention_fwd(&qa, c.st));', '      if (!(PROBE_QA)) TRY(uvc_qkv_attention_fwd(&qa, c.st));')
assert s.count('      TRY(uvc_mlp_fused_fwd(&m, c.st));') == 1
s = s.replace('      TRY(uvc_mlp_fused_fwd(&m, c.st));', '      if (!(PROBE_MLP)) TRY(uvc_mlp_fused_fwd(&m, c.st));')
open(sys.argv[2], 'w').write(s)
PY
  [ $? = 0 ] || exit 1
  objs=$(ls "$R"/uvc_amd/csrc/build/*.o | grep -v '/vit_engine.o$')
  for v in $ORDER; do
    /opt/rocm/bin/hipcc $FLAGS "${V[$v]}" -c /tmp/perturb/vit_probe.hip -o /tmp/perturb/vit_m_$v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/perturb/libuvc_hip_m_$v.so" $objs /tmp/perturb/vit_m_$v.o || exit 1
  done
  ls "$R/tools/perturb" | grep _m_ | wc -l
  exit 0
fi
shift || true                                  # further arguments go to bench.py
OUT=$R/gpurun_out/marginal_probes.txt
mkdir -p "$R/gpurun_out"
cd "$R"
cp uvc_amd/libuvc_hip.so /tmp/good.so
run() { python bench.py --no_cpu_baseline --steps 60 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['cur_resource'])"; }
{
  echo "# DeiT-Tiny batch 512, 60 timed steps per run: step time with one kernel's launches skipped (alternating with the real library on the same box)"
  echo "# variant | ms per step with the real library | with the launches skipped | difference | resource reported (same work?)"
  for v in ${PROBES:-$ORDER}; do
    a=$(run "$@"); cp tools/perturb/libuvc_hip_m_$v.so uvc_amd/libuvc_hip.so; b=$(run "$@"); cp /tmp/good.so uvc_amd/libuvc_hip.so
    a2=$(run "$@"); cp tools/perturb/libuvc_hip_m_$v.so uvc_amd/libuvc_hip.so; b2=$(run "$@"); cp /tmp/good.so uvc_amd/libuvc_hip.so
    python - "$v" "$a" "$b" "$a2" "$b2" <<'PY'
import sys
v = sys.argv[1]
g = [float(sys.argv[i].split()[0]) for i in (2, 4)]
p = [float(sys.argv[i].split()[0]) for i in (3, 5)]
print("%-12s real %.3f %.3f   skipped %.3f %.3f   difference %+.3f ms   resource %s / %s" % (v, g[0], g[1], p[0], p[1], sum(p) / 2 - sum(g) / 2, sys.argv[2].split()[1], sys.argv[3].split()[1]))
PY
  done
} > "$OUT" 2>&1
cp /tmp/good.so uvc_amd/libuvc_hip.so
cat "$OUT"
