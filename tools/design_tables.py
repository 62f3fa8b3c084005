#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 5 ("current table") and of the per-model summary, generated from the bench lines committed under
profiles/ -- numbers in DESIGN.md are pasted from this output, not typed.
    python tools/design_tables.py r4e > /tmp/tables.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4e"
P = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")


def load(n):
    try:
        return json.load(open(P(n)))
    except Exception:
        return None


main = load("bench_steps100_warmup20.json")
traffic = (load("pmc_traffic.json") or {}).get("kernels", {})
if main:
    r = main["roofline"]
    print(f"Headline (`profiles/{tag}_bench_steps100_warmup20.json`, commit {main.get('commit')}): **{main['value']:.0f} img/s, {main['ms_per_step']:.2f} ms / step** "
          f"(device median {main['ms_per_step_device']['median']:.2f} ms), {main['step_tflops_per_gpu']:.0f} TFLOP/s of executed flops = "
          f"{100 * main['step_frac_of_bf16_mfma_peak']:.1f} % of the bf16 MFMA peak; every row of the last block computed: "
          f"{main.get('images_per_sec_all_rows_of_last_block')} img/s; sum of the stand-alone kernel times {main['kernel_ms_per_step_standalone_sum']} ms; "
          f"CPU baseline (oracle, {main['cpu_baseline']['cores']} threads) {main['cpu_baseline']['value']} img/s.\n")
    print(f"`roofline`: {r['kernel']}, {r['calls_per_step']} launches x {r['launch_us']:.1f} us; algorithmic {r['algorithmic_bytes'] / 1e6:.1f} MB -> "
          f"{r['achieved']:.0f} GB/s = **{r['frac']:.3f}** of 8 TB/s; PMC traffic {(r['traffic'] or 0) / 1e6:.1f} MB ({r['traffic_source']}); "
          f"MFMA busy {r.get('mfma_busy_frac')}.\n")
    print("| kernel (key of tools/kernel_table.py) | launches | us | us / step | algorithmic MB | GB/s | TFLOP/s | bound | frac of roof | of the mix ceiling | MFMA busy | PMC MB |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for t in main["top_kernels"]:
        tr = traffic.get(t["key"], {}).get("hbm_bytes")
        print(f"| {t['key']} | {t['calls']} | {t['us']:.1f} | {t['us_per_step']:.0f} | {t['bytes'] / 1e6:.0f} | {t['gbs']:.0f} | {t['tflops']:.0f} | {t['bound']} | {t['frac']:.3f} | "
              f"{t.get('frac_of_mix_ceiling') or ''} | {t.get('mfma_busy_frac') or ''} | {'' if tr is None else round(tr / 1e6)} |")
    print()
print("| model (bench.py --model_type, per-GPU batch) | img/s | ms / step | executed TFLOP/s | of bf16 MFMA peak | largest stand-alone kernel (us x launches) |")
print("|---|---|---|---|---|---|")
for n in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
    if n.startswith(f"{tag}_bench_") and n.endswith(".json") and "steps" not in n:
        d = json.load(open(os.path.join(ROOT, "profiles", n)))
        top = (d.get("top_kernels") or [{}])[0]
        print(f"| {n[len(tag) + 7:-5]} | {d['value']:.0f} | {d['ms_per_step']:.2f} | {d.get('step_tflops_per_gpu', '')} | "
              f"{'' if d.get('step_frac_of_bf16_mfma_peak') is None else round(100 * d['step_frac_of_bf16_mfma_peak'], 1)} % | "
              f"{top.get('key', '')} ({top.get('us', '')} x {top.get('calls', '')}) |")
