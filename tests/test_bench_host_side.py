"""Host-side pieces of bench.py and of the round-6 tools that need no GPU: the CPU count the baselines run on, the entropy
stage's issue yardstick (symbols from the host scan walk x the committed instruction model), and the enumeration behind DESIGN
section 4.1's "a fifth wave per SIMD adds no working lanes"."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_usable_cpus_is_a_count_this_process_may_use():
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_entropy_issue_yardstick_from_a_file_and_the_committed_model():
    import bench
    f = bench._cfg4_one_file(0)                                    # a 3840x2160 4:2:0 q85 gradient file, as the cfg4 workload writes it
    r = bench.cfg4_entropy_roofline([f], 4, 0.5)                   # "4 such files decoded in 0.5 ms"
    assert r["bound"] == "issue" and r["walks"] == 4 and r["sampled_files"] == 1
    assert 1_200_000 * 4 < r["symbols"] < 1_500_000 * 4 and 0.6 < r["table_reads_per_symbol_with_pair_entries"] < 0.9
    assert r["lane_steps"]["round"] == 2 * r["lane_steps"]["count"] and r["lane_steps"]["write"] == r["symbols"]
    # frac = VALU issue cycles of the walks on 1 024 SIMD-32s at 2.4 GHz / the time given
    per = json.load(open(os.path.join(ROOT, "profiles", "entropy_issue_model.json")))["per_64_lane_steps"]
    cycles = sum(per[k]["valu"] * r["lane_steps"][k] / 64.0 for k in ("round", "count", "write")) * 2.0
    assert abs(r["frac"] - cycles / (1024 * 2.4e9) / 0.5e-3) < 2e-4
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 2e-3
    # the stage's compulsory HBM traffic: the coefficient planes of 4 files + the scans read by three kernels, in the time given
    h = r["hbm"]
    assert 4 * 24_883_200 < h["bytes"] < 4 * 24_883_200 + 3 * 4 * len(f) * 1.1
    assert abs(h["frac"] - h["bytes"] / 8e12 / 0.5e-3) < 2e-4 and abs(h["achieved"] / h["peak"] - h["frac"]) < 2e-3


def test_no_cut_with_the_encode_table_adds_busy_lanes_at_twenty_waves_per_cu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "five_waves_feasibility.py")], capture_output=True, text=True, check=True).stdout
    rows = [json.loads(ln) for ln in out.splitlines()]
    today = {r["shape"]: r["product_today_busy_lanes_per_cu"] for r in rows if "summary" in r}
    assert len(today) == 4
    for r in rows:
        if "summary" in r or r["tables"] != "with the encode table" or r["waves_per_cu"] < 20:
            continue
        assert r["busy_lanes_per_cu"] <= today[r["shape"]] * 1.01, r
    committed = open(os.path.join(ROOT, "profiles", "r6_five_waves_feasibility.jsonl")).read()
    assert committed == out                                        # the file under profiles/ is what the tool prints today
