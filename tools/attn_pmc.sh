#!/bin/bash
# PMC passes over the attention kernels at the bench shape (B 512, N 197, H 3, bf16); kernel trace only.
R=$(pwd); OUT=$R/gpurun_out/attn_pmc; mkdir -p $OUT
cat > /tmp/attn_run.py <<'PY'
import torch
from uvc_amd import ops
B, N, H = 512, 197, 3
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3 * D, device="cuda", generator=g).bfloat16()
dout = torch.randn(B, N, D, device="cuda", generator=g).bfloat16()
o = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
dqkv = torch.empty(B, N, 3 * D, device="cuda", dtype=torch.bfloat16)
delta = torch.empty(B, H, N, device="cuda")
for _ in range(4):
    ops.attention_fwd(qkv, o, lse, B, N, H, 1)
    ops.attention_bwd(qkv, o, lse, dout, dqkv, delta, B, N, H, 1)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for C in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); rm -rf /tmp/ap_$i
  PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/ap_$i -o p -- python /tmp/attn_run.py > /dev/null 2> $OUT/err_$i.txt
  f=$(find /tmp/ap_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
per = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "attn" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("k_attn_")[1][:12]
        per.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print("%-14s %-28s mean=%.0f" % (k[0], k[1], sum(v) / len(v)))
PY
done 2>&1 | tee $OUT/summary.txt
