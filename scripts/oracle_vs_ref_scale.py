"""oracle/ vs oracle/_ref (the reference's own headers, k16 build) on a slice of a BASELINE-scale case (tests/scale_cases.py):
per-pile FASTA digests of both, compared; also against the committed oracle digests of the case when the slice lies inside it.
usage: python scripts/oracle_vs_ref_scale.py <case> <first> <npiles> [ref_threads=3]      e.g. cfg2 0 100"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import pyoracle, pyref
from daccord_amd._structs import default_params
from scale_cases import CASES, make_case, pile_digests

name, first, npl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rthreads = int(sys.argv[4]) if len(sys.argv) > 4 else 3
case = dict(CASES[name]); case.pop("pile_ranges", None); case["first"] = first; case["npiles"] = npl
d, ovl, piles, sel = make_case(case, pyoracle.pile_select)
bad = 0
for kw in case["params"]:
    p = default_params(**kw)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    R = pyref.Reference(p); R.set_error_profile(*d.error_profile()); R.load_db(d.bps, d.boff, d.rlen)
    t0 = time.time(); fo, bo = O.run(sel, ovl, d.trace, nthreads=os.cpu_count() or 1); t1 = time.time()
    kk = int(kw.get("khigh", kw.get("k", 8)))
    rt = rthreads if kk <= 14 else (3 if kk == 15 else 2)      # 4^k int32 per context: 4 GiB at k = 15, 16 GiB at k = 16
    fr, br = R.run(sel, ovl, d.trace, nthreads=rt); t2 = time.time()
    do, dr = pile_digests(fo, bo, sel, pyoracle.fasta), pile_digests(fr, br, sel, pyoracle.fasta)
    diff = [int(sel[i]["aread"]) for i in range(len(sel)) if do[i] != dr[i]]
    same = pyoracle.fasta(fo, bo) == pyoracle.fasta(fr, br)
    print(json.dumps({"case": name, "params": kw, "first": first, "npiles": npl, "fragments": int(len(fo)), "bases": int(len(bo)), "fasta_identical": bool(same),
                      "piles_that_differ": diff, "oracle_s": round(t1 - t0, 1), "ref_s": round(t2 - t1, 1), "ref_threads": rt,
                      "fasta_sha256": hashlib.sha256(pyoracle.fasta(fo, bo).encode()).hexdigest()}), flush=True)
    bad += (not same)
sys.exit(1 if bad else 0)
