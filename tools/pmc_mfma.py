#!/usr/bin/env python3
"""Matrix-pipe utilisation per launch of the step's kernels from ONE rocprofv3 PMC pass over tools/kernel_table.py (kernel trace only,
no other trace domain), written as profiles/<tag>_pmc_mfma.json, which bench.py reads for `mfma_busy_frac` (north_star: "evidenced by
rocprof MFMA utilisation and HBM GB/s").

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
              --output-format csv -d $R/gpurun_out/pmc_m -o m -- python $R/tools/kernel_table.py
    python tools/pmc_mfma.py gpurun_out/pmc_m profiles/r3_pmc_mfma.json [commit]

Units (MI355X_MICROARCH.md, per-instruction constants): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the chip's 1024 SIMDs,
GRBM_GUI_ACTIVE the cycles the launch kept the GPU busy SUMMED OVER THE 8 XCDs (each XCD has its own GRBM: the per-dispatch value is 8 x
the launch's cycles -- checked against the kernel-trace duration: qkv 806 684 / 8 = 100.8 k cycles for 41.1 us = 2.45 GHz),
SQ_WAVE_CYCLES quad-cycles summed over waves.  So
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
is the fraction of matrix-pipe cycles in use over the launch (1.0 = every SIMD's pipe busy every cycle), and
    mfma_flops_frac = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 flop / (GRBM_GUI_ACTIVE * 1024 SIMDs * 1024 flop per cycle and SIMD)
the same from the executed bf16 MFMA operations (one MOP = 512 flop; a SIMD peaks at 1024 bf16 flop per cycle: 2.5 PF / 1024 / 2.4 GHz).
Without GRBM_GUI_ACTIVE in the pass the launch time of the kernel trace x 2.4 GHz stands in."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import NAMES  # noqa: E402

N_SIMD = 1024
N_XCD = 8
CLOCK_HZ = 2.4e9


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = {}      # kernel name -> counter -> [values per dispatch]
    for r in csv.DictReader(open(f[0])):
        per.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = {}
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur.setdefault(r["Kernel_Name"], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return per, dur


def main(d, out, commit=None):
    per, dur = load(d)
    kernels = {}
    for key, subs in NAMES.items():
        tot = {}
        ns = 0.0
        n = 0
        ok = True
        for s in subs:
            names = [k for k in per if s in k]
            if not names:
                continue                  # (a key lists every kernel that may serve it)
            for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE"):
                vals = [v for k in names for v in per[k].get(c, [])]
                if vals:
                    tot[c] = tot.get(c, 0.0) + sum(vals) / len(vals)
                    n = max(n, len(vals))
            dv = [v for k in names for v in dur.get(k, [])]
            if dv:
                ns += sum(dv) / len(dv)
        if not ok or "SQ_VALU_MFMA_BUSY_CYCLES" not in tot:
            continue
        cycles = tot["GRBM_GUI_ACTIVE"] / N_XCD if tot.get("GRBM_GUI_ACTIVE") else (ns * 1e-9 * CLOCK_HZ if ns else None)
        ent = dict(launches=n, mfma_busy_cycles=round(tot["SQ_VALU_MFMA_BUSY_CYCLES"]), gpu_cycles=round(cycles) if cycles else None,
                   cycles_from="GRBM_GUI_ACTIVE / 8 XCDs" if tot.get("GRBM_GUI_ACTIVE") else "kernel trace duration x 2.4 GHz",
                   clock_ghz_under_pmc=round(cycles / ns, 3) if (cycles and ns) else None,
                   launch_us_under_pmc=round(ns / 1e3, 1) if ns else None,
                   sq_busy_cycles=round(tot.get("SQ_BUSY_CYCLES", 0)), sq_wave_quad_cycles=round(tot.get("SQ_WAVE_CYCLES", 0)))
        if cycles:
            ent["mfma_busy_frac"] = round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * N_SIMD), 4)
            if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in tot:
                ent["mfma_mops_bf16"] = round(tot["SQ_INSTS_VALU_MFMA_MOPS_BF16"])
                ent["mfma_flops_frac"] = round(tot["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0 / (cycles * N_SIMD * 1024.0), 4)
        kernels[key] = ent
    json.dump(dict(commit=commit, method="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES "
                                         "SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE over tools/kernel_table.py; mfma_busy_frac = "
                                         "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", kernels=kernels), open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:26s} mfma busy {100 * v.get('mfma_busy_frac', 0):5.1f} %   flops {100 * v.get('mfma_flops_frac', 0):5.1f} % of peak   ({v['launches']} launches)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
