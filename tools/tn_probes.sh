#!/bin/bash
# What bounds the split-M weight-gradient GEMM k_gemm_tn_dma on 128 workgroups (r6): the library built with -DUVC_TN_PROBE=1 (operand ring only), 2 (compute only),
# 3 (no partial-tile store) from the product source (the switch is a compile-time constant, 0 in the product).  Wrong results on purpose, timing only.
#   here:            tools/tn_probes.sh build   -> tools/perturb/libuvc_hip_tnprobe{1,2,3}.so
#   on the GPU box:  tools/tn_probes.sh run     -> gpurun_out/tn_probes.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result -I$R/include -I$R/uvc_amd/csrc"
if [ "${1:-}" = build ]; then
  python -m uvc_amd.build > /dev/null || exit 1
  mkdir -p /tmp/perturb "$R/tools/perturb"
  for v in 1 2 3; do
    /opt/rocm/bin/hipcc $FLAGS -DUVC_TN_PROBE=$v -c "$R/uvc_amd/csrc/gemm.hip" -o /tmp/perturb/gemm_tnprobe$v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/perturb/libuvc_hip_tnprobe$v.so" $(ls "$R"/uvc_amd/csrc/build/*.o | grep -v '/gemm.o$') /tmp/perturb/gemm_tnprobe$v.o || exit 1
  done
  ls "$R/tools/perturb" | grep tnprobe
  exit 0
fi
cd "$R"
{
  echo "# k_gemm_tn_dma at DeiT-Tiny batch 512 (M = 100 864; 3 tiles x 42 splits = 126 workgroups): us per launch incl. the 10-us reduce, tools/tn_variants.py"
  echo "## product"; python tools/tn_variants.py 2>/dev/null | grep "^dW"
  for v in 1 2 3; do
    echo "## UVC_TN_PROBE=$v (1 = operand ring only, 2 = compute only, 3 = no partial-tile store)"
    UVC_LIB=$R/tools/perturb/libuvc_hip_tnprobe$v.so python tools/with_lib.py tools/tn_variants.py 2>/dev/null | grep "^dW"
  done
} > gpurun_out/tn_probes.txt 2>&1
cat gpurun_out/tn_probes.txt
