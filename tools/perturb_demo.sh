#!/bin/bash
# Shows once that tests/test_streaming_batch_gpu.py goes red when a streaming kernel is wrong.
#   here (build container):  tools/perturb_demo.sh build     -> tools/perturb/libuvc_hip_{ws,lnbwd}.so (git-ignored; they travel with gpurun)
#   on the GPU box:          tools/perturb_demo.sh run       -> gpurun_out/perturb_demo.txt
# The perturbed libraries are built from sed-edited COPIES of gemm.hip under /tmp: the product source carries no test switch.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value -Wno-unused-result -I$R/include -I$R/uvc_amd/csrc"
if [ "${1:-}" = build ]; then
  python -m uvc_amd.build > /dev/null || exit 1
  mkdir -p /tmp/perturb "$R/tools/perturb"
  # 1. k_gemm_ws: every output tile of the K <= 192 streaming GEMM (qkv, fc1, dfc2, dproj) scaled by 1.02
  sed 's|\*reinterpret_cast<f32x4\*>(stg + li \* EPW + j \* 16 + gq \* 4) = c;|*reinterpret_cast<f32x4*>(stg + li * EPW + j * 16 + gq * 4) = c * 1.02f;|' "$R/uvc_amd/csrc/gemm.hip" > /tmp/perturb/gemm_ws.hip
  # 2. k_gemm_wsn_lnbwd_dma: dx of the fused dgrad + LayerNorm backward scaled by 1.02
  sed 's|o\[e\] = rstd \* (c0\[e\] - c1 - xh \* c2);   |o[e] = 1.02f * rstd * (c0[e] - c1 - xh * c2);   |' "$R/uvc_amd/csrc/gemm.hip" > /tmp/perturb/gemm_lnbwd.hip
  for v in ws lnbwd; do
    cmp -s /tmp/perturb/gemm_$v.hip "$R/uvc_amd/csrc/gemm.hip" && { echo "perturbation $v did not apply"; exit 1; }
    sed -i 's|#include "common.h"|#include "'"$R"'/uvc_amd/csrc/common.h"|; s|#include "../../include/uvc_kernels.h"|#include "'"$R"'/include/uvc_kernels.h"|' /tmp/perturb/gemm_$v.hip
    /opt/rocm/bin/hipcc $FLAGS -c /tmp/perturb/gemm_$v.hip -o /tmp/perturb/gemm_$v.o || exit 1
    objs=$(ls "$R"/uvc_amd/csrc/build/*.o | grep -v '/gemm.o$')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/perturb/libuvc_hip_$v.so" $objs /tmp/perturb/gemm_$v.o || exit 1
  done
  ls -la "$R/tools/perturb"
  exit 0
fi
OUT=$R/gpurun_out/perturb_demo.txt
mkdir -p "$R/gpurun_out"
cp "$R/uvc_amd/libuvc_hip.so" /tmp/libuvc_hip_good.so
{
  echo "# tests/test_streaming_batch_gpu.py::test_tiny_step_matches_oracle_at_streaming_batch with one streaming kernel perturbed (tools/perturb_demo.sh)"
  for v in ws lnbwd; do
    cp "$R/tools/perturb/libuvc_hip_$v.so" "$R/uvc_amd/libuvc_hip.so"
    echo; echo "## perturbed: $v (k_gemm_ws output x 1.02 | k_gemm_wsn_lnbwd_dma dx x 1.02) -- expected: FAILED"
    (cd "$R" && python -m pytest tests/test_streaming_batch_gpu.py -q -x -k "matches_oracle_at_streaming_batch and not fp32" 2>&1 | grep -E "AssertionError|passed|failed|assert " | cut -c1-600 | head -8)
  done
  cp /tmp/libuvc_hip_good.so "$R/uvc_amd/libuvc_hip.so"
  echo; echo "## unperturbed library -- expected: passed"
  (cd "$R" && python -m pytest tests/test_streaming_batch_gpu.py -q -x -k "matches_oracle_at_streaming_batch and not fp32" 2>&1 | grep -E "passed|failed")
} > "$OUT" 2>&1
cat "$OUT"
