#!/usr/bin/env python3
"""Markdown rows for DESIGN.md section 6 from a set of evidence files (profiles/<prefix>_* or gpurun_out/prof3_summary/*).
usage: summarize_profiles.py <directory> <file prefix, e.g. r3_ or ''>"""
import csv
import json
import os
import sys

d, pre = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
PEAK = 8000.0


def last_json(name):
    with open(os.path.join(d, pre + name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def kernel_rows(name):
    p = os.path.join(d, pre + name)
    if not os.path.exists(p):
        return []
    return [r for r in csv.DictReader(open(p)) if "fused_resample" in r["Name"] or "generic" in r["Name"]]


print("| workload | frames | kernel ms (hipEvents / rocprofv3 avg [calls, min]) | algorithmic TB/s | % of 8 TB/s | frac_timed |")
print("|---|---|---|---|---|---|")
for w in ("cfg2", "cfg2-alpha", "cfg5", "cfg3-l0", "cfg3-l1", "cfg3-l2", "cfg3-l3", "cfg4-resize", "cfg1-resize", "up2-hermite", "up3-robidoux"):
    try:
        b = last_json(f"bench_{w}.json")
    except FileNotFoundError:
        continue
    r = b["roofline"]
    ks = kernel_rows(f"{w}_kernel_stats.csv")
    prof = "; ".join(f"{float(k['AverageNs']) / 1e6:.4f} [{k['Calls']}, {float(k['MinNs']) / 1e6:.4f}]" for k in ks) or "-"
    print(f"| {w} | {b['config']['frames_per_gpu']} | {r['kernel_ms']:.4f} / {prof} | {r['achieved'] / 1000:.2f} | {100 * r['frac']:.1f} | {r.get('frac_timed', '')} |")
b = last_json("bench_cfg2.json")
r = b["roofline"]
print()
print("headline:", {k: b[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "read probe", r["measured_read_GBps"], "traffic", r["traffic"])
print("cpu_baseline:", b["cpu_baseline"]["value"], b["cpu_baseline"]["single_thread_MPps"], "host_dropin:", b.get("host_dropin"))
if "strong_1024" in b:
    print("strong_1024:", b["strong_1024"]["value"], b["strong_1024"]["ms_per_step"])
for name in ("bench_cfg3_job.json", "bench_strong_1024_1gpu.json"):
    try:
        j = last_json(name)
        print(name, "ms/step", j["ms_per_step"], "value", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"])
    except FileNotFoundError:
        pass
try:
    j = json.load(open(os.path.join(d, pre + "bench_jpeg.json")))
    for k, v in j.items():
        print(k, {a: c for a, c in v.items() if "ms" in a or "GBps" in a or "fused" in a})
    e = json.load(open(os.path.join(d, pre + "bench_entropy.json")))
    print({k: v for k, v in e.items() if not isinstance(v, dict)})
    for k, v in e["roofline"].items():
        print(" ", k, v["ms"], v["frac"])
except FileNotFoundError:
    pass
