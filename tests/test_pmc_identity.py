"""Which committed PMC summary bench.py may quote for the build in the tree: the one whose device sources are this build's
(csrc_hash) or, for kernels whose gfx950 instruction stream did not change, the one that recorded their kernel_isa hashes.
Needs the built library and llvm-objdump (both present in the build container and on the GPU box); no GPU."""
import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from daccord_amd import build  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(build.LIB) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump")), reason="needs the built library and llvm-objdump")


def test_kernel_isa_hashes_name_every_kernel():
    h = build.kernel_isa_hashes()
    for k in ("k_window_fast<0>", "k_window_fast<7>", "k_window_fast<1>", "k_window_fast<6>", "k_window_fast<2>", "k_window_fast<3>", "k_window_fast<4>", "k_window_long",
              "k_window", "k_classify", "k_trace", "k_trace_wide<4>", "k_trace_wide<8>", "k_vote", "k_prep", "k_prescan"):
        assert k in h and len(h[k]) == 16
    assert len(set(h.values())) == len(h)


def test_bench_quotes_counters_only_for_kernels_it_runs():
    import bench
    cur, isa = build.csrc_hash(), build.kernel_isa_hashes()
    dom = "k_window_fast<1>"
    summaries = {os.path.basename(f): json.load(open(f)) for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))}
    ok = {f: pm for f, pm in summaries.items() if (pm["workload"]["reads"], pm["workload"]["k"]) == (10000, 14) and dom in pm["kernels"] and
          (pm.get("csrc_hash") == cur or pm.get("kernel_isa", {}).get(dom) == isa.get(dom))}
    r = bench.pmc_lookup(dom, 10000, 10000, 20.0, 14)
    assert "pmc_source" in r and not r["pmc_source"].startswith("error")
    if ok:
        assert r["traffic"] > 0 and "NOT measured in this run" in r["pmc_source"]
        # per-kernel numbers only for kernels whose code is the measured one
        pm = ok[r["pmc_source"].split("profiles/")[1].split(",")[0]]
        for kn in r["traffic_all_kernels"]:
            assert pm.get("csrc_hash") == cur or pm["kernel_isa"][kn] == isa[kn]
    else:
        assert "traffic" not in r and r["pmc_source"].startswith("none for this build")
    # a workload nobody measured has no counters
    assert "traffic" not in bench.pmc_lookup(dom, 123, 456, 7.0, 9)


def test_fused_tier0_line_reads_tier0_counters():
    """the bench names the first slot `k_classify+k_window_fast<0>` (one event pair around both launches); the summaries are keyed by
    kernel names: the lookup of that line must go to tier 0's counters (round 5: the line of the final build carried traffic null)"""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"k_window_fast<0>" if dom == "k_classify+k_window_fast<0>" else dom' in src
    a = bench.pmc_lookup("k_window_fast<0>", 10000, 10000, 20.0, 14)
    assert "pmc_source" in a and not a["pmc_source"].startswith("error")
