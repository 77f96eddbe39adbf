"""GPU probe of the wide-window path (w in 64..128: generic engine only) and of the scratch retry in generic-only mode.
Every case runs in its own process with DACC_DEBUG_SYNC=1, so that a device fault names the kernel it happened in and does not
take the other cases with it.  usage: python scripts/gpu_probe_wide.py            (all cases, a summary line per case)
                                       python scripts/gpu_probe_wide.py <case>     (one case, in this process)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    "narrow_fast": (dict(k=8), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.12, seed=3), 4, {}),
    "narrow_generic_only_retry": (dict(w=63, a=20, k=6, maxalign=3), dict(genome_len=60000, nreads=300, read_len=3000, erate=0.28, seed=481075, ins_frac=0.2, del_frac=0.7, sub_frac=0.1), 2, {"DACC_NOFAST": "1"}),
    "w128_a32_k12": (dict(w=128, a=32, k=12), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.15, seed=140), 6, {}),
    "w100_a25_k8": (dict(w=100, a=25, k=8), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.12, seed=108), 6, {}),
    "w65_a16_k8": (dict(w=65, a=16, k=8), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.15, seed=73), 6, {}),
    "w96_full": (dict(w=96, a=24, k=9, producefull=1), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.2, seed=105), 6, {}),
    "w128_a10_k13_14": (dict(w=128, a=10, klow=13, khigh=14), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.12, seed=141), 4, {}),
    "w64_a16_k8": (dict(w=64, a=16, k=8), dict(genome_len=60000, nreads=150, read_len=3000, erate=0.12, seed=72), 6, {}),
}


def run_case(name):
    import pyoracle
    from daccord_amd import engine
    from daccord_amd._structs import default_params
    from daccord_amd.synth import SynthData
    from common import windows_equal, frags_equal
    kw, dk, npiles, _ = CASES[name]
    d = SynthData(dk.pop("genome_len"), dk.pop("nreads"), dk.pop("read_len"), **dk)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    p = default_params(**kw)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[:npiles], ovl, d.trace, nthreads=8, want_windows=True)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    t = time.time()
    fx, bx = E(piles[:npiles], ovl, d.trace)
    dt = time.time() - t
    nwin = len(O.windows())
    bad = windows_equal(O.windows(), E.debug_windows())
    same = frags_equal(fo, bo, fx, bx) and engine.fasta(fx, bx) == pyoracle.fasta(fo, bo)
    fy, by = E(piles[1:npiles], ovl, d.trace)                      # the same context again
    f2, b2 = O.run(piles[1:npiles], ovl, d.trace, nthreads=8)
    again = frags_equal(f2, b2, fy, by)
    print("CASE %s: %s  windows %d (differing %d), bases %d, device %.2f s, second batch %s" % (name, "PASS" if (same and not bad and again) else "FAIL", nwin, len(bad), len(bo), dt, "ok" if again else "DIFFERS"), flush=True)
    return 0 if (same and not bad and again) else 1


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(run_case(sys.argv[1]))
    rc = 0
    only = os.environ.get("PROBE_CASES")
    for name, (_, _, _, env) in CASES.items():
        if only and name not in only.split(","):
            continue
        for sync in (os.environ.get("PROBE_SYNC", "0,1").split(",")):
            e = dict(os.environ); e.update(env); e["DACC_DEBUG_SYNC"] = sync
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), name], env=e, capture_output=True, text=True, timeout=60)
                tail = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(("CASE", "[dacc]")) or "fault" in l.lower() or "error" in l.lower() or "HSA" in l]
                print("== %s (DACC_DEBUG_SYNC=%s): exit %d" % (name, sync, r.returncode)); print("\n".join(tail[-14:]), flush=True)
                rc |= (r.returncode != 0)
            except subprocess.TimeoutExpired:
                print("== %s: TIMEOUT" % name, flush=True); rc = 1
    sys.exit(rc)
