#!/usr/bin/env python3
"""HBM bytes per launch of the step's kernels from two rocprofv3 PMC passes (the guide's recipe: separate runs, --kernel-trace --pmc
FETCH_SIZE and --pmc WRITE_SIZE only), written as profiles/<tag>_pmc_traffic.json, which bench.py reads for `roofline.traffic`.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -o f -- python $R/tools/kernel_table.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -o w -- python $R/tools/kernel_table.py
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r2_pmc_traffic.json [commit]

FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts the 128-byte
requests of a wide coalesced stream at 64 bytes, so hbm_bytes = 2 * FETCH + WRITE (calibrated in round 1 on kernels of known
traffic: profiles/r1_pmc_hbm_traffic_kbench.csv, 1.00-1.01 x algorithmic for every streaming GEMM)."""
import csv
import glob
import json
import os
import sys

# key (tools/kernel_table.py) -> kernel-name substrings whose per-dispatch counters add up to one launch of the entry
NAMES = {
    "ln_fwd": ["k_ln_fwd_v"],
    "qkv": ["k_gemm_ws<unsigned short, unsigned short, 1"],
    "attn_fwd": ["k_attn_fwd"],
    "qkv+attn_fwd (student)": ["k_qkv_attn_fwd<6, true", "k_qkv_attn_fwd<12, true"],
    "qkv+attn_fwd (teacher)": ["k_qkv_attn_fwd<6, false", "k_qkv_attn_fwd<12, false"],
    "proj+resid": ["k_gemm_wsn16_dma<3, 6, false"],
    "proj+resid+norm2": ["k_gemm_wsn16_dma<3, 6, true"],
    "fc1+gelu,gelu'": ["k_gemm_ws<unsigned short, unsigned short, 7", "k_gemm_ws<unsigned short, unsigned short, 9"],      # (9: GELU' as one byte, r6)
    "fc2+resid+gate+norm1": ["k_gemm_wsn16_dma<4"],
    "teacher mlp_fused+norm1": ["k_mlp_fused"],
    "dfc2 x gelu'": ["k_gemm_ws<unsigned short, unsigned short, 8", "k_gemm_ws<unsigned short, unsigned short, 10"],
    "dfc1+ln2_bwd": ["k_gemm_wsn_lnbwd_dma<24"],
    "dqkv+ln1_bwd": ["k_gemm_wsn_lnbwd_dma<18"],
    "dproj": ["k_gemm_ws<unsigned short, unsigned short, 0"],
    "attn_bwd": ["k_attn_bwd_one", "k_attn_bwd_dq", "k_attn_bwd_dkv"],
    "dW2 (+reduce)": ["k_gemm_tn_dma<192, 256", "k_gemm_tn8p<6, 4"],      # (r6: the two-group schedule)
    "dW1 (+reduce)": ["k_gemm_tn_dma<256, 192", "k_gemm_tn8p<8, 3"],
    "dWproj (+reduce)": ["k_gemm_tn<unsigned short", "k_gemm_tn_dma<96, 192"],
    "dWqkv (+reduce)": ["k_gemm_tn_dma<192, 192", "k_gemm_tn8p<6, 3"],
    "clip+adamw": ["k_adamw"],
}


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per


def mean_for(per, sub):
    vals = [v for name, lst in per.items() if sub in name for v in lst]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(fd, wd, out, commit=None):
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    kernels = {}
    for key, subs in NAMES.items():
        f_kib = w_kib = 0.0
        n = 0
        ok = True
        for s in subs:
            a, na = mean_for(fe, s)
            b, nb = mean_for(wr, s)
            if a is None or b is None:
                continue                      # (a key lists every kernel that may serve it: the attention backward is one kernel or a pair)
            f_kib += a; w_kib += b; n = max(n, na)
        ok = n > 0
        if ok:
            kernels[key] = dict(fetch_kib=round(f_kib, 1), write_kib=round(w_kib, 1), hbm_bytes=int((2 * f_kib + w_kib) * 1024), launches=n)
    json.dump(dict(commit=commit, method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/kernel_table.py; "
                                         "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
                   note="wgrad entries: the split-M GEMM only (k_tn_reduce is shared by the four shapes and not attributed)",
                   kernels=kernels), open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:24s} {v['hbm_bytes'] / 1e6:9.1f} MB  ({v['launches']} launches)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
