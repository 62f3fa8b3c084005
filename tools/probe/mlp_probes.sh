#!/bin/bash
# Timing probes of k_mlp_fused_p<TRAIN>: library variants built with -DUVC_MLP_PROBE=n (fused_mlp.hip), wrong results on purpose.
#   here: tools/probe/mlp_probes.sh build ; on the GPU box: tools/probe/mlp_probes.sh run
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result -I$R/include -I$R/uvc_amd/csrc"
if [ "${1:-}" = build ]; then
  python -m uvc_amd.build > /dev/null || exit 1
  mkdir -p /tmp/perturb "$R/tools/perturb"
  objs=$(ls "$R"/uvc_amd/csrc/build/*.o | grep -v '/fused_mlp.o$')
  for n in ${PROBES:-1 2 3}; do
    /opt/rocm/bin/hipcc $FLAGS -DUVC_MLP_PROBE=$n -c "$R/uvc_amd/csrc/fused_mlp.hip" -o /tmp/perturb/fused_mlp_p$n.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/perturb/libuvc_hip_mlpprobe$n.so" $objs /tmp/perturb/fused_mlp_p$n.o || exit 1
  done
  exit 0
fi
cd "$R"
{
  echo "## library"; PYTHONPATH=. TIME_ONLY=1 timeout 120 python tools/probe/mlp_p_train_probe.py | grep "form"
  for n in ${PROBES:-1 2 3}; do
    echo "## probe $n"; UVC_LIB=$R/tools/perturb/libuvc_hip_mlpprobe$n.so PYTHONPATH=. TIME_ONLY=1 timeout 120 python tools/with_lib.py tools/probe/mlp_p_train_probe.py | grep "form"
  done
} 2>&1 | grep -v amdgpu.ids | tee "$R/gpurun_out/mlp_probes.txt"
