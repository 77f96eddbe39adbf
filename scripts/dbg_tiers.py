import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = SynthData(250000, 1000, 5000, seed=3)
ovl, piles = engine.pile_select(d.ovl, d.piles)
E = engine.Engine(default_params(k=k)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
fr, ba = E(piles[:64], ovl, d.trace)
t = E.timing()
w = E.debug_windows()
import hashlib
print("TIERS=%s SCHED=%s k=%d bases=%d md5=%s window=%.1fms tiers_out=%s tiers_ms=%s status=%s" % (os.environ.get("DACC_TIERS"), os.environ.get("DACC_SCHED"), k, len(ba), hashlib.md5(ba).hexdigest()[:8], t.window_ms, list(t.tier_out), [round(x, 1) for x in t.tier_ms], dict(zip(*np.unique(w["status"], return_counts=True)))))
