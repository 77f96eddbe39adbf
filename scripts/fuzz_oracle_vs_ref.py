"""Randomized runs of oracle/ (the CPU restatement every parity test of the HIP path is anchored on) against oracle/_ref (the
REFERENCE'S OWN headers compiled against the libmaus2 stand-in, oracle/ref_shim/): same random run parameters, error profiles,
coverages, trace spacings and warped traces as the emulation / GPU fuzzing (tests/common.py).  Compares the model tables bit
for bit and the FASTA of every pile.  k above 12 runs in the k16 build (our factory around the reference's graph template).
usage: python scripts/fuzz_oracle_vs_ref.py <seed> <rounds> [--wide] [--w128] [--warp]"""
import sys, os, time, random
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from daccord_amd.synth import SynthData
from daccord_amd._structs import default_params
import pyoracle, pyref
from common import random_run_config, random_run_config_wide, random_run_config_w128, warp_trace

seed0 = int(sys.argv[1]); nrounds = int(sys.argv[2]); wide = "--wide" in sys.argv
rng = random.Random(seed0)
bad = 0
for r in range(nrounds):
    kw, data, maxin, nplc = (random_run_config_w128 if '--w128' in sys.argv else random_run_config_wide if wide else random_run_config)(rng)
    if "--warp" in sys.argv and not data.get("warp"):
        data["warp"] = (rng.choice([3, 5]), rng.choice([300, 580, 900]), 2000) if data["tspace"] > 125 else (rng.choice([2, 3, 5]), rng.choice([60, 115, 150]))
    try:
        d = SynthData(data["genome_len"], data["nreads"], data["read_len"], **{k: v for k, v in data.items() if k not in ("genome_len", "nreads", "read_len", "profile", "warp")})
        prof = data.get("profile") or d.error_profile()
        ovl, piles = pyoracle.pile_select(d.ovl, d.piles, trace_bytes=d.trace_bytes, maxinput=maxin)
        npl = min(len(piles), nplc)
        trace = d.trace
        if data.get("warp"):
            trace = warp_trace(ovl, piles, d.trace, range(npl), *data["warp"])
        p = default_params(**kw)
        O = pyoracle.Oracle(p); O.set_error_profile(*prof); O.load_db(d.bps, d.boff, d.rlen)
        R = pyref.Reference(p); R.set_error_profile(*prof); R.load_db(d.bps, d.boff, d.rlen)
        teq = np.array_equal(O.tables(), R.tables())
        t0 = time.time()
        fo, bo = O.run(piles[:npl], ovl, trace, trace_bytes=d.trace_bytes, nthreads=4)
        t1 = time.time()
        # (a DebruijnGraph<k> of the reference holds 4^k int32: 1 GiB at k = 14, 16 GiB at k = 16, per context)
        fr, br = R.run(piles[:npl], ovl, trace, trace_bytes=d.trace_bytes, nthreads=(4 if p.khigh <= 13 else (2 if p.khigh == 14 else 1)))
        t2 = time.time()
        ok = teq and pyoracle.fasta(fo, bo) == pyoracle.fasta(fr, br)
        print(("OK  " if ok else "BAD "), r, kw, data, "maxinput", maxin, "piles", npl, "fragments", len(fo), "bases", len(bo), "tables", teq,
              "oracle %.1fs ref %.1fs" % (t1 - t0, t2 - t1), flush=True)
        bad += (not ok)
    except Exception as ex:
        print("EXC ", r, kw, data, maxin, repr(ex)[:300], flush=True)
        bad += 1
print("DONE bad=%d" % bad, flush=True)
sys.exit(1 if bad else 0)
