"""Which windows does the last LDS tier hand to the generic engine, and why (DACC_DEBUG_RETRY=1)."""
import sys, os, collections
os.environ["DACC_DEBUG_RETRY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
k = int(sys.argv[1]) if len(sys.argv) > 1 else 14
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ont = len(sys.argv) > 3 and sys.argv[3] == "ont"          # config 5 error mix (bench.py --ont)
cov = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0   # coverage (bench.py --coverage)
skw = dict(ins_frac=1 / 3.0, del_frac=1 / 3.0, sub_frac=1 / 3.0) if ont else {}
d = SynthData(int(n * 10000 / cov), n, 10000, seed=3, **skw)
ovl, piles = engine.pile_select(d.ovl, d.piles)
E = engine.Engine(default_params(k=k)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
fr, ba = E(piles, ovl, d.trace)
t = E.timing()
r = E.debug_retry()
print("k=%d piles=%d windows=%d window=%.1fms tiers_out=%s tiers_ms=%s" % (k, len(piles), t.nwindows, t.window_ms, list(t.tier_out), [round(x, 1) for x in t.tier_ms]))
print("handed to generic:", len(r))
c = collections.Counter((hex(int(x[1])), int(x[3])) for x in r)
for (f, ff), v in c.most_common(20):
    print("  flags %s ff %d : %d" % (f, ff, v))
print("mao of those:", sorted(int(x[2]) for x in r)[:50])
# (window index, flags, mao, filter frequency) of the first 40: the pile of window i is found through the window counts of the piles
print("windows:", [(int(x[0]), hex(int(x[1])), int(x[2]), int(x[3])) for x in r[:40]])
