"""Generates tests/golden/fuzz_wide.json: the ORACLE's digests for the wide random parameter sets (tests/common.py:
random_run_config_wide -- deep piles, trace spacings 126..300 with two byte trace values, foreign error profiles, warped
traces) that tests/test_gpu_fuzz_wide.py runs on the GPU.  The oracle takes 10-60 s per set on the build container's CPUs,
which would be minutes of metered GPU box time per run; so it runs HERE once, and the GPU test only regenerates the
deterministic inputs, runs the HIP path through the C ABI and compares digests.

usage: python tests/golden/make_golden_fuzz.py [seed] [nsets] [nthreads]
"""
import hashlib
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from fuzz_wide_cases import wide_cases, make_wide_case  # noqa: E402
from scale_cases import window_digest  # noqa: E402


def main():
    import pyoracle
    from daccord_amd._structs import default_params
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20260922
    nsets = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    out = {"seed": seed, "nsets": nsets, "sets": []}
    for i, (kw, data, maxin, npl) in enumerate(wide_cases(seed, nsets)):
        t0 = time.time()
        d, prof, ovl, piles, sel, trace = make_wide_case(data, maxin, npl, pyoracle.pile_select)
        p = default_params(**kw)
        O = pyoracle.Oracle(p); O.set_error_profile(*prof); O.load_db(d.bps, d.boff, d.rlen)
        fo, bo = O.run(sel, ovl, trace, trace_bytes=d.trace_bytes, want_windows=True, nthreads=nthreads)
        w = O.windows()
        e = {"i": i, "params": kw, "data": data, "maxinput": maxin, "npiles": int(len(sel)), "nwindows": int(len(w)),
             "nfragments": int(len(fo)), "nbases": int(len(bo)),
             "windows_sha256": window_digest(w), "fasta_sha256": hashlib.sha256(pyoracle.fasta(fo, bo).encode()).hexdigest(),
             "oracle_seconds": round(time.time() - t0, 1)}
        out["sets"].append(e)
        print(i, kw, data, "windows", len(w), "bases", len(bo), "%.1fs" % (time.time() - t0), flush=True)
        with open(os.path.join(HERE, "fuzz_wide.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
