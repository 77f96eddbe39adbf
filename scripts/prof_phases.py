"""Per-phase shader-cycle breakdown of the window kernel (needs the -DDACC_PROFILE build: DACC_LIB=...prof.so)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
NAMES = ["gather+strings", "peq+elength", "instances(sort)", "nodes", "successors", "pair-gen (lanes)", "gapfill", "pair-replay (lane 0)",
         "stretches", "cand+tab+stretchfeas", "F trees (lanes)", "R blocks (lanes)", "tail", "cand-errors", "align+emit", "-",
         " candidates+pieces", " loadTab", " feas:ranges", " F:bucket scan+heap fill (lane 0)", " F:drain, pops+extensions (lane 0)"]
EXTRA = {24: "buildSeq cycles (lane 0)", 21: "cut sequences continued", 22: "exact pairs", 23: "serial combines", 25: "pairs", 26: "F batches", 27: "pair rounds", 28: "batch restarts"}
npiles = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = SynthData(250000, 1000, 5000, seed=3)
ovl, piles = engine.pile_select(d.ovl, d.piles)
for k in (8, 14):
    E = engine.Engine(default_params(k=k)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    t0 = time.time(); fr, ba = E(piles[:npiles], ovl, d.trace); t1 = time.time() - t0
    t = E.timing(); pr = E.profile().astype(np.float64)
    w = E.debug_windows()
    print("k=%d piles=%d windows=%d blocks=%d bases=%d wall=%.2fs trace=%.1fms window=%.1fms vote=%.1fms h2d=%.1fms tiers_out=%s tiers_ms=%s" % (
        k, npiles, t.nwindows, t.nblocks, len(ba), t1, t.trace_ms, t.window_ms, t.vote_ms, t.h2d_ms, list(t.tier_out), [round(x, 1) for x in t.tier_ms]))
    print("  status", dict(zip(*np.unique(w["status"], return_counts=True))), "ff", dict(zip(*np.unique(w["filterfreq"][w["status"] == 1], return_counts=True))), "mean mao %.1f" % w["mao"].mean())
    tot = pr[:15].sum() + pr[16:18].sum()
    if tot > 0:
        for i, n in enumerate(NAMES):
            print("  %-16s %6.2f%%  %10.0f cyc/window" % (n, 100 * pr[i] / tot, pr[i] / max(1, t.nwindows)))
        print("  total cyc/window %.0f" % (tot / max(1, t.nwindows)))
        for i, n in EXTRA.items():
            print("  %-22s %12.1f per window" % (n, pr[i] / max(1, t.nwindows)))
        print("  block lifetime: clock64 sum %.3e wall_clock64 sum %.3e (100MHz) max wall %.3f ms; ratio clock/wall %.2f" % (pr[30], pr[31], pr[29]/1e5, pr[30]/max(pr[31],1)))
    E.close()
