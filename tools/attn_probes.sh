#!/bin/bash
# Timing probes of the one-pass attention backward (k_attn_bwd_one): library variants built with -DUVC_ATTN_PROBE=n (see attention.hip), wrong numbers on purpose.
#   here:            tools/attn_probes.sh build        -> tools/perturb/libuvc_hip_attnprobe{1,2,3,4}.so
#   on the GPU box:  tools/attn_probes.sh run           -> gpurun_out/attn_probes.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result -I$R/include -I$R/uvc_amd/csrc"
if [ "${1:-}" = build ]; then
  python -m uvc_amd.build > /dev/null || exit 1
  mkdir -p /tmp/perturb "$R/tools/perturb"
  objs=$(ls "$R"/uvc_amd/csrc/build/*.o | grep -v '/attention.o$')
  for n in ${PROBES:-1 2 3 4}; do
    /opt/rocm/bin/hipcc $FLAGS -DUVC_ATTN_PROBE=$n -c "$R/uvc_amd/csrc/attention.hip" -o /tmp/perturb/attention_p$n.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/perturb/libuvc_hip_attnprobe$n.so" $objs /tmp/perturb/attention_p$n.o || exit 1
  done
  ls -la "$R/tools/perturb"
  exit 0
fi
OUT=$R/gpurun_out/attn_probes.txt
mkdir -p "$R/gpurun_out"
cd "$R"
{
  echo "## library"; PYTHONPATH=. timeout 120 python tools/attn_bwd_time.py
  for n in ${PROBES:-1 2 3 4}; do
    echo "## probe $n"; UVC_LIB=$R/tools/perturb/libuvc_hip_attnprobe$n.so PYTHONPATH=. timeout 120 python tools/with_lib.py tools/attn_bwd_time.py | grep one-pass
  done
} 2>&1 | grep -v amdgpu.ids | tee "$OUT"
