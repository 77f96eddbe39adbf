"""CPU check without a GPU: the host emulation of the kernels (tests/emul, 1-lane or 64-lane wavefront) against the
committed oracle digests of the BASELINE-scale cases (tests/golden/scale_<case>.json).
usage: emul_scale_parity.py <case> [npiles] [lanes] [run index]"""
import os, sys, json, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import emul_lib
import pyoracle
from daccord_amd import engine
from daccord_amd._structs import default_params
from scale_cases import CASES, make_case, window_digest, pile_digests

name = sys.argv[1]
npiles = int(sys.argv[2]) if len(sys.argv) > 2 else None
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
only = int(sys.argv[4]) if len(sys.argv) > 4 else None
G = json.load(open(os.path.join(ROOT, "tests", "golden", "scale_%s.json" % name)))
case = dict(CASES[name])
d, ovl, piles, sel = make_case(case, pyoracle.pile_select)
if npiles:
    sel = sel[:npiles]
ok = True
for ri, run in enumerate(G["runs"]):
    if only is not None and ri != only:
        continue
    p = default_params(**run["params"])
    E = emul_lib.Emul(p, lanes=lanes); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    t0 = time.time()
    fx, bx = E.run(sel, ovl, d.trace)
    pd = pile_digests(fx, bx, sel, engine.fasta)
    bad = [i for i, (a, b) in enumerate(zip(pd, run["pile_sha256"])) if a != b]
    full = len(sel) == len(run["pile_sha256"])
    wok = (window_digest(E.windows()) == run["windows_sha256"]) if full else None
    print("%s %s: %d piles, tiers (t1,t2,t3,generic) %s, %.1fs, piles differing from the oracle: %d %s, window digest %s"
          % (name, run["params"], len(sel), E.counts(), time.time() - t0, len(bad), bad[:10], wok))
    ok = ok and not bad and wok is not False
sys.exit(0 if ok else 1)
