"""Hot-spot ledger of the window kernels (round 5): the fine sites of the -DDACC_PROFILE build next to its phase counters.
Needs DACC_LIB=daccord_amd/libdaccord_hip_prof.so.   usage: prof_sites.py [npiles=256] [k=14] [coverage=20]
Piles are the first npiles of BASELINE config 2's generator (10 kb reads, seed 3), so the mix of windows is the bench's."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
PH = ["gather+strings", "peq+elength", "instances(sort)", "nodes", "successors", "pair-gen (lanes)", "gapfill", "pair-replay (lane 0)",
      "stretches", "cand+tab+stretchfeas", "F trees (lanes)", "R blocks (lanes)", "tail", "cand-errors", "align+emit"]
SITES = {1: "combineLane: intervals of a pair (scoreInterval per forward pop + heap)", 2: "combineLane: one pop (sift, next lighter, push, record)",
         3: "replayRound: per live pair up to dispatch", 4: "replayPair: entry -> weight, compare with lightest kept", 5: "offerCandidate: pop of the full candidate heap",
         6: "offerCandidate: stretch count + consensus length (r05b: sequence walk)", 7: "offerCandidate: rare full duplicate comparison (r05b: load words + compare)", 8: "offerCandidate: push (r05b: slot copy + push)",
         9: "replayRound: serial combinePair", 10: "replayRound: exact pair", 11: "F tree: root extensions", 12: "F tree: bucket membership scan (per 64 entries)",
         13: "F tree: bucket heap fill (per 64 entries)", 14: "F tree: pop (ipop + fields)", 15: "F tree: popped path's record (slab load)",
         16: "F tree: one successor stretch (iterator, slab load, extendPath)", 17: "R enum: pop (ipop + fields)", 18: "R enum: accepted-paths check",
         19: "R enum: root extensions", 20: "R enum: one predecessor stretch (iterator, linkOk, slab load, push)", 21: "R blocks: copy + sort + rank",
         22: "F tree finish", 23: "reachability prune", 24: "spillS", 25: "restoreS", 26: "pair-gen: classifyPair", 27: "restoreInstances", 28: "buildInstances",
         29: "saveInstances", 30: "loadHand", 31: "buildInstances: generation", 32: "buildInstances: sort of the last k-mers", 33: "buildInstances: sort of the instances", 34: "materializeKept before restoreS"}
npiles = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 14
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
reads = 10000
d = SynthData(int(reads * 10000 / cov), reads, 10000, seed=3, nthreads=os.cpu_count() or 1, aread_range=(0, npiles))
ovl, piles = engine.pile_select(d.ovl, d.piles)
E = engine.Engine(default_params(k=k)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
fr, ba = E(piles[:npiles], ovl, d.trace)
t = E.timing(); pr = E.profile().astype(np.float64); cyc, vis = E.profile_fine()
cyc = cyc.astype(np.float64); vis = vis.astype(np.float64)
nw = max(1, t.nwindows)
tot = pr[:15].sum() + pr[16:18].sum()
print("k=%d piles=%d windows=%d window=%.1fms tiers_ms=%s tier_out=%s  total %.0f cycles/window (phase probes)" % (k, npiles, t.nwindows, t.window_ms, [round(x, 1) for x in t.tier_ms], list(t.tier_out), tot / nw))
for i, n in enumerate(PH):
    print("  phase %-24s %6.2f%%  %9.0f cyc/window" % (n, 100 * pr[i] / tot, pr[i] / nw))
print("  fine sites (cycles charged once per wavefront visit; %% of the phase total):")
rows = []
for s_, n in sorted(SITES.items()):
    if vis[s_] > 0:
        rows.append((cyc[s_], s_, n))
for c, s_, n in sorted(rows, reverse=True):
    print("  site %2d %-72s %6.2f%%  %9.0f cyc/window  %8.1f visits/window  %7.0f cyc/visit" % (s_, n, 100 * c / tot, c / nw, vis[s_] / nw, c / vis[s_]))
print(json.dumps({"k": k, "npiles": npiles, "windows": int(t.nwindows), "total_cycles_per_window": tot / nw,
                  "phases": {n: pr[i] / nw for i, n in enumerate(PH)}, "sites": {str(s_): {"name": n, "cycles_per_window": cyc[s_] / nw, "visits_per_window": vis[s_] / nw} for s_, n in SITES.items()}}))
E.close()
