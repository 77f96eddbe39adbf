"""Generates tests/golden/scale_<case>.json: the ORACLE's output at BASELINE scale, frozen as digests.

The oracle is far too slow to run next to the GPU at these sizes inside the metered GPU minutes, and the GPU box has no
/root/reference either way, so the oracle is run HERE (build container, CPU) on the deterministic synthetic inputs and
its output is committed as small fixtures:
  fasta_sha256          SHA-256 of the FASTA text of all selected piles (engine.fasta / pyoracle.fasta are the same format)
  windows_sha256        SHA-256 over the per-window records (pile, y, status, mao, elength, k, filterfreq, minrate, cons)
  pile_sha256[i]        first 12 hex digits of SHA-256 of the FASTA text of pile i alone (well counter 0), to localise a diff
  status/ff histograms  what the case exercises
tests/test_gpu_scale.py regenerates the same synthetic input on the GPU box and compares the HIP path's digests.

usage: python tests/golden/make_golden_scale.py <case> [nthreads]     (cases: see CASES)
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
from scale_cases import CASES, make_case, window_digest, pile_digests  # noqa: E402


def main():
    import pyoracle
    from daccord_amd._structs import default_params
    name = sys.argv[1]
    nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    case = CASES[name]
    d, ovl, piles, sel = make_case(case, pyoracle.pile_select)
    out = {"case": name, "spec": case, "runs": []}
    for kw in case["params"]:
        p = default_params(**kw)
        O = pyoracle.Oracle(p)
        O.set_error_profile(*d.error_profile())
        O.load_db(d.bps, d.boff, d.rlen)
        t0 = time.time()
        fr, ba = O.run(sel, ovl, d.trace, nthreads=nthreads, want_windows=True)
        dt = time.time() - t0
        w = O.windows()
        txt = pyoracle.fasta(fr, ba)
        st = {int(k): int(v) for k, v in zip(*np.unique(w["status"], return_counts=True))}
        okw = w[w["status"] == 1]
        ff = {int(k): int(v) for k, v in zip(*np.unique(okw["filterfreq"], return_counts=True))}
        run = {"params": kw, "npiles": int(len(sel)), "nwindows": int(len(w)), "nfragments": int(len(fr)), "nbases": int(len(ba)),
               "fasta_sha256": hashlib.sha256(txt.encode()).hexdigest(), "windows_sha256": window_digest(w),
               "pile_sha256": pile_digests(fr, ba, sel, pyoracle.fasta), "status_hist": st, "filterfreq_hist": ff,
               "mean_mao": float(w["mao"].mean()) if len(w) else 0.0, "max_mao": int(w["mao"].max()) if len(w) else 0,
               "oracle_seconds": round(dt, 1), "oracle_threads": nthreads}
        out["runs"].append(run)
        print(name, kw, "windows", len(w), "bases", len(ba), "%.1fs" % dt, st, ff, flush=True)
        with open(os.path.join(HERE, "scale_%s.json" % name), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
