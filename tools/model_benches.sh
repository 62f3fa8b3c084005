#!/bin/bash
# bench.py lines of the other BASELINE models and of Stage-2 (one JSON each, with their own roofline / top_kernels):
#   tools/model_benches.sh <tag>   -> gpurun_out/<tag>/bench_<model>_b<batch>.json, bench_stage2_b512.json
TAG=${1:-rX}; R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
for cfg in "deit_base_patch16_224 128" "deit_base_patch16_224 256" "deit_small_patch16_224 256" "t2t_vit_14 128"; do
  set -- $cfg
  timeout 600 python bench.py --no_cpu_baseline --model_type $1 --batch $2 --steps 40 --warmup 10 > "$OUT/bench_$1_b$2.json" 2> "$OUT/bench_$1_b$2.err"
  python -c "import json,sys; d=json.load(open('$OUT/bench_$1_b$2.json')); print('$1', $2, d['value'], d['ms_per_step'], d['step_frac_of_bf16_mfma_peak'])"
done
# BASELINE.json configs 3 / 4 / 5 AS STATED (r5: bench.py takes the reference's switches): DeiT-Small budget 0.58, DeiT-Base with the distillation
# token (N = 198, two heads), T2T-ViT-14 with patch + block gating
# (config 3 at per-GPU batch 1024, as SURVEY 8(d) restates it, beside the 256 of the earlier rounds' tables)
for cfg in "config3 deit_small_patch16_224 256 --budget 0.58" "config3 deit_small_patch16_224 1024 --budget 0.58" "config4 deit_base_patch16_224 128 --enable_deit 1" "config5 t2t_vit_14 128 --enable_patch_gating 2"; do
  set -- $cfg
  timeout 600 python bench.py --no_cpu_baseline --model_type $2 --batch $3 --steps 40 --warmup 10 ${@:4} > "$OUT/bench_$1_$2_b$3.json" 2> "$OUT/bench_$1_$2_b$3.err"
  python -c "import json,sys; d=json.load(open('$OUT/bench_$1_$2_b$3.json')); print('$1', d['config']['workload'], d['value'], d['ms_per_step'], d['step_frac_of_bf16_mfma_peak'])"
done
timeout 600 python bench.py --stage 2 --no_cpu_baseline --steps 40 --warmup 10 > "$OUT/bench_stage2_b512.json" 2> "$OUT/bench_stage2.err"
python -c "import json; d=json.load(open('$OUT/bench_stage2_b512.json')); print('stage2', d['value'], d['ms_per_step'])"
