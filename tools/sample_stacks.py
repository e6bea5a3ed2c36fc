#!/usr/bin/env python3
"""Rank the call stacks tools/bench_abi_jobs.cpp sampled on CPU time (IFHIP_BENCH_SAMPLE=<file>): where the HOST spends its
CPU while jobs run through the libimageflow ABI.

    usage: tools/sample_stacks.py <samples file> [top N]

Each line of the file is one sample, `module+0xoffset` frames, leaf first.  Frames are symbolised with llvm-symbolizer (ROCm's,
/opt/rocm/lib/llvm/bin) where the module has symbols, else left as module+offset.  Three rankings are printed:
  leaf       -- the function the CPU was in;
  library    -- the first frame of each stack that lies in libimageflow_hip.so (which of OUR calls led there);
  entry      -- the outermost exported hip*/hsa_* frame (which runtime entry point the time was spent under);
  whole      -- the most frequent whole stacks.
"""
import collections
import os
import subprocess
import sys

SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def symbolise(frames):
    """{(module, offset)} -> {(module, offset): name}"""
    by_mod = collections.defaultdict(list)
    for m, o in frames:
        by_mod[m].append(o)
    names = {}
    for m, offs in by_mod.items():
        short = os.path.basename(m)
        if m == "?" or not os.path.exists(m) or not os.path.exists(SYMBOLIZER):
            for o in offs:
                names[(m, o)] = f"{short}+{o:#x}"
            continue
        inp = "\n".join(f"{o:#x}" for o in offs) + "\n"
        out = subprocess.run([SYMBOLIZER, f"--obj={m}", "--functions=linkage", "--demangle", "--no-inlines", "--output-style=LLVM"],
                             input=inp, capture_output=True, text=True).stdout
        blocks = [b for b in out.split("\n\n") if b.strip()]
        for o, b in zip(offs, blocks):
            fn = b.strip().splitlines()[0].strip()
            names[(m, o)] = f"{short}!{fn[:90]}" if fn and fn != "??" else f"{short}+{o:#x}"
    return names


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    stacks = []
    for line in open(path):
        fr = []
        for tok in line.strip().split(";"):
            if "+0x" not in tok:
                continue
            m, o = tok.rsplit("+0x", 1)
            fr.append((m, int(o, 16)))
        if fr:
            stacks.append(fr)
    names = symbolise({f for st in stacks for f in st})
    n = len(stacks)
    leaf, ours, entry, whole = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    for st in stacks:
        sym = [names[f] for f in st]
        leaf[sym[0]] += 1
        whole[" <- ".join(x.replace("(anonymous namespace)::", "").split("(")[0][-48:] for x in sym[:12])] += 1
        mine = next((s for s, f in zip(sym, st) if "libimageflow_hip" in f[0]), "(none: runtime's own threads / the harness)")
        ours[mine] += 1
        api = [s for s in sym if "!hip" in s or "!hsa_" in s]
        entry[api[-1] if api else "(no hip*/hsa_* frame)"] += 1
    print(f"{n} samples at 1 kHz of process CPU time = {n / 1000:.2f} CPU-seconds")
    for title, c in (("leaf", leaf), ("first frame in libimageflow_hip.so", ours), ("outermost hip*/hsa_* entry", entry)):
        print(f"-- {title}")
        for k, v in c.most_common(top):
            print(f"  {100.0 * v / n:5.1f} %  {k}")
    print("-- whole stacks (12 frames, leaf first)")
    for k, v in whole.most_common(top):
        print(f"  {100.0 * v / n:5.1f} %  {k}")


if __name__ == "__main__":
    main()
