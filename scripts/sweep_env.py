"""One process, one synthetic data set, several library settings: for every setting ("K=V,K=V" -- environment knobs the
library reads in dacc_create -- or "lib=<path>" cannot change inside a process, use DACC_LIB for that) create an engine, run the
batch `steps` times resident and print step time, tier times, hand-overs and the FASTA digest.
usage: python scripts/sweep_env.py <reads> <steps> "SETTING" ["SETTING" ...]      ("" = defaults)"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData


def main():
    reads, steps = int(sys.argv[1]), int(sys.argv[2])
    cov = float(os.environ.get("SWEEP_COVERAGE", "20"))
    skw = dict(ins_frac=1 / 3.0, del_frac=1 / 3.0, sub_frac=1 / 3.0) if os.environ.get("SWEEP_ONT") == "1" else {}      # config 5's error mix
    d = SynthData(int(reads * 10000 / cov), reads, 10000, seed=3, nthreads=os.cpu_count() or 1, **skw)
    ovl, piles = engine.pile_select(d.ovl, d.piles)
    keys = set()
    for setting in sys.argv[3:]:
        for k in keys:
            os.environ.pop(k, None)
        for kv in [x for x in setting.split(",") if x]:
            k, v = kv.split("="); os.environ[k] = v; keys.add(k)
        pkw = dict(k=int(os.environ.get("SWEEP_K", "14")))
        if os.environ.get("SWEEP_W"):      # window size / advance (wide windows: SWEEP_W=80 SWEEP_A=20)
            pkw["w"] = int(os.environ["SWEEP_W"]); pkw["a"] = int(os.environ.get("SWEEP_A", str(max(1, pkw["w"] // 4))))
        E = engine.Engine(default_params(**pkw))
        E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
        fr, ba = E(piles, ovl, d.trace)
        t0 = time.perf_counter(); tier = [0.0, 0.0, 0.0]; win = 0.0; trc = 0.0; vot = 0.0
        for _ in range(steps):
            E.rerun(); fr, ba = E.collect(); t = E.timing()
            win += t.window_ms; trc += t.trace_ms; vot += t.vote_ms
            for i in range(3):
                tier[i] += t.tier_ms[i]
        dt = (time.perf_counter() - t0) / steps
        h = hashlib.sha256(); well = 0
        for i in range(0, len(fr), 256):
            h.update(engine.fasta(fr[i:i + 256], ba, start_well=well).encode()); well += len(fr[i:i + 256])
        print(json.dumps({"setting": setting, "ms_per_step": round(1e3 * dt, 2), "mbase_s": round(len(ba) / dt / 1e6, 3),
                          "trace_ms": round(trc / steps, 2), "vote_ms": round(vot / steps, 2), "window_ms": round(win / steps, 2), "tier_ms": [round(x / steps, 2) for x in tier],
                          "handed_on": [int(t.tier_out[i]) for i in range(3)], "t0": [round(float(t.tier0_ms), 1), int(t.tier0_in), int(t.tier0_out)],
                          "t7": [round(float(getattr(t, "tier7_ms", 0.0)), 1), int(getattr(t, "tier7_in", 0)), int(getattr(t, "tier7_out", 0))], "t10": [round(float(getattr(t, "tier10_ms", 0.0)), 1), int(getattr(t, "tier10_out", 0))], "sha": h.hexdigest()[:16]}), flush=True)
        del E


main()
