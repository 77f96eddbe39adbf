"""w = 56 / 60 / 63: where do the windows finish, and is skipping the first slot's tiers faster?  (consrow cliff)"""
import os, sys, time, hashlib, json
sys.path.insert(0, "/root/repo")
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
d = SynthData(int(1500 * 10000 / 20.0), 1500, 10000, seed=3, nthreads=os.cpu_count() or 1)
ovl, piles = engine.pile_select(d.ovl, d.piles)
for w, a in ((48, 12), (56, 14), (60, 15), (63, 21)):
    for tiers in ("31", "6"):
        os.environ["DACC_TIERS"] = tiers
        E = engine.Engine(default_params(k=14, w=w, a=a)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
        fr, ba = E(piles, ovl, d.trace)
        t0 = time.perf_counter()
        for _ in range(2):
            E.rerun(); fr, ba = E.collect(); t = E.timing()
        dt = (time.perf_counter() - t0) / 2
        print(json.dumps({"w": w, "a": a, "DACC_TIERS": tiers, "ms": round(1e3 * dt, 1), "mbase_s": round(len(ba) / dt / 1e6, 2), "windows": int(t.nwindows), "tier_ms": [round(x, 1) for x in t.tier_ms],
                          "t0": [int(t.tier0_in), int(t.tier0_out)], "t7": [int(t.tier7_in), int(t.tier7_out)], "handed_on": [int(x) for x in t.tier_out], "long": int(t.long_windows),
                          "sha": hashlib.sha256(ba).hexdigest()[:12]}), flush=True)
        E.close()
