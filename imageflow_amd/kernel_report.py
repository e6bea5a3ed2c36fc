"""Print VGPR/SGPR/scratch/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys

from . import build as B


def report(src=None, k=4):
    import os
    src = src or os.path.join(B.CSRC, B.FUSED)
    cmd = [B.HIPCC, "-x", "hip", "--offload-arch=gfx950"] + B.COMMON + B.FUSED_FLAGS + [f"-DIFHIP_FUSED_K={k}", "-c", src, "-o", "/dev/null",
                                                                      "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: \s*(Function Name|[A-Za-z ]+\[?[A-Za-z/]*\]?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("ifhip::", "")
        print(f"{name:45s} VGPR {r.get('VGPRs','?'):>4} SGPR {r.get('TotalSGPRs','?'):>4} "
              f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} "
              f"LDS {r.get('LDS Size [bytes/block]','?')}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1].isdigit():
        report(k=int(sys.argv[1]))
    elif len(sys.argv) > 1:
        report(sys.argv[1])
    else:
        for kk in range(1, 9):
            report(k=kk)
