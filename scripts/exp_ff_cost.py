import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
d = SynthData(1000000, 2000, 10000, seed=3)
ovl, piles = engine.pile_select(d.ovl, d.piles)
for kw in (dict(k=14), dict(k=14, maxfilterfreq=1), dict(k=14, minfilterfreq=2)):
    E = engine.Engine(default_params(**kw)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    E(piles, ovl, d.trace); E.rerun(); t = E.timing()
    print(kw, "window %.1f ms tiers %s out %s" % (t.window_ms, [round(x,1) for x in t.tier_ms], list(t.tier_out)))
    E.close()
