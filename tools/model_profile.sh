#!/bin/bash
# rocprofv3 kernel trace of a 20-step bench of one model, restricted to whole UVC-train steps (tools/rocprof_steps_only.py):
#   tools/model_profile.sh <model_type> <batch> <tag>   -> gpurun_out/<tag>_kernel_stats_<model>_steps_only.csv  (run on the GPU box from the repo root)
set -u
MODEL=$1; BATCH=$2; TAG=${3:-rX}
R=$(pwd); mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_$MODEL && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/rp_$MODEL -o kt -- python "$R/bench.py" --model_type $MODEL --batch $BATCH --steps 20 --warmup 5 --no_cpu_baseline > /tmp/rp_$MODEL.json 2> /tmp/rp_$MODEL.err
DB=$(find /tmp/rp_$MODEL -name "*.db" | head -1)
python "$R/tools/rocprof_steps_only.py" "$DB" "$R/gpurun_out/${TAG}_kernel_stats_${MODEL}_steps_only.csv" 5 14
