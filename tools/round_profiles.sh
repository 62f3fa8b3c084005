#!/bin/bash
# Everything profiles/<tag>_* is made of, in one go on the GPU box (run through gpurun from the repo root):
#   tools/round_profiles.sh r2c
# 1. bench.py with its defaults (100 timed steps, CPU baseline)                 -> gpurun_out/<tag>/bench_steps100_warmup20.json
# 2. rocprofv3 --kernel-trace --stats of a 20-step bench                          -> kernel_stats_bench_steps20_warmup5.csv,
#                                                                                    kernel_stats_uvc_train_steps_only.csv
# 3. two PMC passes (FETCH_SIZE, WRITE_SIZE; kernel trace only) over the table   -> pmc_traffic.json
# The caller copies gpurun_out/<tag>/* into profiles/<tag>_*.
set -u
TAG=${1:-rX}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
COMMIT=$(cat "$R/.commit_for_profiles" 2>/dev/null || echo unknown)
timeout 600 python bench.py > "$OUT/bench_steps100_warmup20.json" 2> "$OUT/bench.err" || echo "bench failed"
tail -c 600 "$OUT/bench_steps100_warmup20.json"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_kt && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_kt -o kt -- python "$R/bench.py" --steps 20 --warmup 5 --no_cpu_baseline > "$OUT/bench_under_rocprof_steps20_warmup5.json" 2> "$OUT/rocprof.err"
DB=$(find /tmp/rp_kt -name "*.db" | head -1)
python "$R/tools/rocprof_summary.py" "$DB" "$OUT/kernel_stats_bench_steps20_warmup5.csv" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no_cpu_baseline (commit $COMMIT); all launches incl. bench.py's stand-alone kernel table"
python "$R/tools/rocprof_steps_only.py" "$DB" "$OUT/kernel_stats_uvc_train_steps_only.csv" 5 24
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rp_$C && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/rp_$C -o p -- python "$R/tools/kernel_table.py" > /dev/null 2> "$OUT/pmc_$C.err"
done
python "$R/tools/pmc_traffic.py" /tmp/rp_FETCH_SIZE /tmp/rp_WRITE_SIZE "$OUT/pmc_traffic.json" "$COMMIT"
# 4. one SQ pass (matrix-pipe utilisation per kernel; kernel trace only)         -> pmc_mfma.json
rm -rf /tmp/rp_MFMA && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d /tmp/rp_MFMA -o p -- python "$R/tools/kernel_table.py" > /dev/null 2> "$OUT/pmc_MFMA.err"
python "$R/tools/pmc_mfma.py" /tmp/rp_MFMA "$OUT/pmc_mfma.json" "$COMMIT" > "$OUT/pmc_mfma.txt" 2>&1; cat "$OUT/pmc_mfma.txt"
for C in FETCH_SIZE WRITE_SIZE; do
  f=$(find /tmp/rp_$C -name "*counter_collection.csv" | head -1)
  python - "$f" "$OUT/pmc_${C}_kernel_table.csv" $C <<'EOF'
import csv, sys
per = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[3]:
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc %s -- python tools/kernel_table.py : mean KiB per dispatch\nkernel,dispatches,mean_kib\n" % sys.argv[3])
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        f.write('"%s",%d,%.1f\n' % (k.replace('"', "'"), len(v), sum(v) / len(v)))
EOF
done
ls -la "$OUT"
