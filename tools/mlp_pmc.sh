#!/bin/bash
# PMC passes over the fused inference MLP kernels (persistent form at M = 100864, k_mlp_fused_v3 at M = 12800); kernel trace only.
R=$(pwd); OUT=$R/gpurun_out/mlp_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/mp_$i
  PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/mp_$i -o p -- python $R/tools/mlp_persist.py pmc > /dev/null 2> $OUT/err_$i.txt
  f=$(find /tmp/mp_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
per = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp_fused" in r["Kernel_Name"]:
        per.setdefault((r["Kernel_Name"][:60], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print("%-62s %-28s n=%d mean=%.0f" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
  tail -2 $OUT/err_$i.txt | grep -i "error\|invalid\|not" | head -2
done 2>&1 | tee $OUT/summary.txt
f=$(find /tmp/mp_1 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee -a $OUT/summary.txt
import csv, sys
per = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "mlp_fused" in r["Kernel_Name"]:
        per.setdefault(r["Kernel_Name"][:60], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, v in per.items():
    print("%-62s n=%d mean %.1f us  min %.1f" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
