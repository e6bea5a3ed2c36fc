#!/usr/bin/env python3
"""Experiment driver (NOT part of the product): bench lib variants interleaved, report kernel ms per repeat.
usage: exp_variants.py [--pattern P] [--reps N] variant_suffix ...   ('' = the product library)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
pattern, reps, env_extra, workload = "random", 3, {}, "cfg2"
vs = [""]
i = 0
while i < len(args):
    if args[i] == "--pattern": pattern = args[i + 1]; i += 2
    elif args[i] == "--reps": reps = int(args[i + 1]); i += 2
    elif args[i] == "--workload": workload = args[i + 1]; i += 2
    elif "=" in args[i] and args[i].split("=")[0].isupper(): vs.append(args[i]); i += 1   # env variant KEY=VAL
    else: vs.append(args[i]); i += 1
res = {v: [] for v in vs}
for r in range(reps):
    for v in vs:
        env = dict(os.environ)
        if "=" in v:
            k, val = v.split("=", 1); env[k] = val
        elif v:
            lib = os.path.join(ROOT, "imageflow_amd", "lib", f"libimageflow_hip{v}.so")
            if not os.path.exists(lib): continue
            env["IFHIP_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3",
                              "--no-cpu-baseline", "--pattern", pattern, "--workload", workload], env=env, capture_output=True, text=True).stdout
        try:
            res[v].append(json.loads(out.strip().splitlines()[-1])["roofline"]["kernel_ms"])
        except Exception as e:
            res[v].append(float("nan"))
for v in vs:
    xs = res[v]
    print(f"variant {v!r:28} {workload} pattern {pattern:8} kernel_ms {xs}  min {min(xs) if xs else None}")
