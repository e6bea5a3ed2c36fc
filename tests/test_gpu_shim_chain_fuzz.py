"""The job interpreter behind the C ABI (csrc/abi_shim.cpp) and the Python node mirrors (imageflow_amd/flow/nodes) are two
statements of flow/nodes/*.rs; a seeded sweep of random node chains and two-input graphs must come out of both with the same
pixels, size and alpha flag (tools/fuzz_shim_chains.py: the long form of this sweep, and what it found in round 6).  The
mirrors themselves are pinned to the oracle in test_node_mirrors.py / test_gpu_bitmap_ops.py / test_gpu_abi_shim.py."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_chains_and_graphs_agree_with_the_mirrors(seed):
    import fuzz_shim_chains as F
    summary, failing = F.sweep(seed, chains=4000)
    assert summary["chains"] == 4000 and summary["graphs"] > 500
    assert summary["disagreements"] == 0, json.dumps(failing[0])[:1500]


def test_concurrent_jobs_agree_with_the_mirrors():
    """the jobs of 32 cases at a time on 4 host threads (one context per job): per-thread streams, the block cache's stream-ordered
    release, JPEG sources of four shared geometries through the decode coalescer"""
    import fuzz_shim_chains as F
    summary, failing = F.sweep(21, chains=2000, threads=4)
    assert summary["chains"] == 2000 and summary["jpeg_sources"] > 500
    assert summary["disagreements"] == 0, json.dumps(failing[0])[:1500]


def test_fill_rect_runs_behind_the_colour_filter_of_the_same_job():
    """round 6: fill_rect was launched on the null stream while the colour filter before it ran on the job's stream -- the
    filter then ran over (part of) the filled rectangle"""
    import fuzz_shim_chains as F
    E = F.environment()
    for k in range(40):
        case = {"size": [180, 120], "alpha": False, "seed": 100 + k, "mark": [4, 4, 1],
                "nodes": [{"color_filter_srgb": {"saturation": 0.13}}, {"fill_rect": {"x1": 3, "y1": 5, "x2": 170, "y2": 110, "color": {"srgb": {"hex": "FFEA9EFF"}}}}]}
        rec = F.run_case(case, E)
        assert rec["ok"], json.dumps(rec)[:800]
