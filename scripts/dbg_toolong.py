"""Piles 2000..2050 of the default bench data set hold the one window in 10^7 with a B string longer than 64 bases
(generic engine, second stream): result must equal the run with the LDS tiers switched off."""
import sys, os, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
d = SynthData(5000000, 10000, 10000, seed=3)
ovl, piles = engine.pile_select(d.ovl, d.piles)
for nofast in ("0", "1"):
    os.environ["DACC_NOFAST"] = nofast
    E = engine.Engine(default_params(k=14)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    t0 = time.time(); fr, ba = E(piles[2000:2050], ovl, d.trace); t1 = time.time() - t0
    t = E.timing()
    print("NOFAST=%s bases=%d md5=%s wall=%.2fs window=%.1fms tiers_out=%s tiers_ms=%s" % (nofast, len(ba), hashlib.md5(ba).hexdigest()[:8], t1, t.window_ms, list(t.tier_out), [round(x, 1) for x in t.tier_ms]))
    E.close()
