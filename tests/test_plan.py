"""BatchPlan::plan (daccord_amd/csrc/batch_plan.hpp: the host planning of dacc_submit_piles -- device records, window schedule, active
ranges, error keys, offsets, scratch capacities) is a pure function of the batch.  Its outputs are digested for a set of inputs
(narrow / deep / wide-window / two-byte-trace batches, empty and malformed piles, an out-of-range pile) and compared with digests
committed from the serial planner of rounds 1-4 (tests/golden/plan_digests.json, regenerate: python tests/test_plan.py --write):
the threaded planner must reproduce every byte, whatever its number of threads."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import emul_lib  # noqa: E402
from daccord_amd._structs import default_params  # noqa: E402
from daccord_amd.synth import SynthData  # noqa: E402
from daccord_amd import engine  # noqa: E402

GOLD = os.path.join(HERE, "golden", "plan_digests.json")


_CASES = None


def _cases():
    global _CASES
    if _CASES is not None:
        return _CASES
    out = []
    specs = [("narrow_k8", dict(k=8), dict(genome_len=60000, nreads=150, read_len=3000, seed=1), 5000),
             ("narrow_k14_top10", dict(k=14), dict(genome_len=60000, nreads=300, read_len=3000, seed=2, erate=0.2), 10),
             ("deep", dict(k=12, a=20), dict(genome_len=20000, nreads=400, read_len=2000, seed=3), 5000),
             ("two_byte_trace", dict(k=10, tspace=200), dict(genome_len=60000, nreads=120, read_len=3000, seed=4, tspace=200), 5000),
             ("wide_w100", dict(k=9, w=100, a=25), dict(genome_len=60000, nreads=150, read_len=3000, seed=5), 5000),
             ("maxalign_w32", dict(k=8, w=32, a=8, maxalign=6), dict(genome_len=60000, nreads=200, read_len=1500, seed=6, min_overlap=200), 5000)]
    for name, kw, dk, maxin in specs:
        dk = dict(dk)
        d = SynthData(dk.pop("genome_len"), dk.pop("nreads"), dk.pop("read_len"), **dk)
        ovl, piles = engine.pile_select(d.ovl, d.piles, trace_bytes=d.trace_bytes, maxinput=maxin)
        out.append((name, kw, d, ovl.copy(), piles.copy(), d.trace.copy()))
        if name in ("narrow_k8", "two_byte_trace"):
            # malformed / empty piles in the middle of a good batch (each one is dropped and reported, the batch goes on)
            o2, p2, t2 = ovl.copy(), piles.copy(), d.trace.copy()
            p2[1]["novl"] = 0
            f = int(p2[3]["first_ovl"]); o2[f + 1]["aread"] += 1                      # foreign A read
            f = int(p2[5]["first_ovl"]); o2[f + 2]["abpos"], o2[f + 1]["abpos"] = o2[f + 1]["abpos"], o2[f + 2]["abpos"] + 1   # not sorted by abpos
            f = int(p2[7]["first_ovl"]); t2[int(o2[f]["trace_off"]) + 1] += 3         # trace B lengths do not sum up
            f = int(p2[9]["first_ovl"]); o2[f]["tlen"] -= 2                            # trace length does not match
            f = int(p2[11]["first_ovl"]); o2[f]["bepos"] = 10 ** 8                      # B interval beyond the read
            out.append((name + "_malformed", kw, d, o2, p2, t2))
        if name == "narrow_k8":
            p3 = piles.copy(); p3[2]["aread"] = 10 ** 6                                # pile out of range: the whole call fails
            out.append((name + "_pile_out_of_range", kw, d, ovl.copy(), p3, d.trace.copy()))
    _CASES = out
    return out


def _digest(kw, d, ovl, piles, trace):
    p = default_params(**kw)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    L = E.L
    L.emul_plan_digest.restype = C.c_uint64
    L.emul_plan_digest.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_int)]
    pl = np.ascontiguousarray(piles); ov = np.ascontiguousarray(ovl); tr = np.ascontiguousarray(trace)
    rc = C.c_int(0)
    h = L.emul_plan_digest(E.h, pl.ctypes.data, len(pl), ov.ctypes.data, len(ov), tr.ctypes.data, tr.nbytes // d.trace_bytes, d.trace_bytes, C.byref(rc))
    return "%016x" % h, rc.value


def _all(threads=None):
    if threads is not None:
        os.environ["DACC_PLAN_THREADS"] = str(threads)
    else:
        os.environ.pop("DACC_PLAN_THREADS", None)
    return {name: list(_digest(kw, d, ovl, piles, trace)) for name, kw, d, ovl, piles, trace in _cases()}


@pytest.mark.parametrize("threads", [None, 1, 2, 3, 7, 16])
def test_plan_reproduces_the_committed_digests(threads):
    G = json.load(open(GOLD))
    got = _all(threads)
    os.environ.pop("DACC_PLAN_THREADS", None)
    assert set(got) == set(G)
    for k in G:
        assert got[k] == G[k], (k, threads, got[k], G[k])
    assert G["narrow_k8_pile_out_of_range"][1] != 0 and G["narrow_k8_malformed"][1] == 0 and G["narrow_k8"][0] != G["narrow_k8_malformed"][0]


def test_plan_with_sixteen_threads_that_really_run():
    """ the committed cases have 120-400 piles = at most a dozen chunks of 32, and the planner starts one thread per 64 piles, so
    `threads=16` above runs 3-7 of them.  Here a batch is tiled to 2400+ piles (the tiles share the overlap records, which the planner only
    reads) and the piles-per-thread floor is lowered: 16 threads over 75+ chunks against the serial planner, byte for byte."""
    name, kw, d, ovl, piles, trace = [c for c in _cases() if c[0] == "deep"][0]
    reps = (2400 + len(piles) - 1) // len(piles)
    big = np.concatenate([piles] * reps)
    assert len(big) >= 2400
    os.environ["DACC_PLAN_THREADS"] = "1"
    try:
        serial = _digest(kw, d, ovl, big, trace)
        os.environ["DACC_PLAN_THREADS"] = "16"; os.environ["DACC_PLAN_PILES_PER_THREAD"] = "8"
        par16 = _digest(kw, d, ovl, big, trace)
        os.environ["DACC_PLAN_THREADS"] = "-5"      # not a count: treated as unset (hardware concurrency, at most 16)
        unset = _digest(kw, d, ovl, big, trace)
        os.environ["DACC_PLAN_THREADS"] = "100000"      # clamped to 64
        many = _digest(kw, d, ovl, big, trace)
    finally:
        os.environ.pop("DACC_PLAN_THREADS", None); os.environ.pop("DACC_PLAN_PILES_PER_THREAD", None)
    assert serial == par16 == unset == many and serial[1] == 0, (serial, par16, unset, many)


def test_plan_has_no_data_race(tmp_path):
    """the planner's threads under ThreadSanitizer (tests/emul/plan_tsan.cpp): no report, one digest for 1 / 8 / 3 / 16 threads;
    an exception in a worker is rethrown in the caller after the join"""
    import subprocess
    exe = str(tmp_path / "plan_tsan")
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", exe, os.path.join(HERE, "emul", "plan_tsan.cpp")],
                        capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime for this compiler: " + cc.stderr[-300:])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ThreadSanitizer" not in (r.stdout + r.stderr) and "MISMATCH" not in r.stdout, (r.stdout + r.stderr)[-2000:]
    assert len(set(l.split()[-1] for l in r.stdout.splitlines() if l.startswith("threads"))) == 1
    # an exception thrown inside a worker thread reaches the caller (the C ABI turns it into DACC_ENOMEM: capi.hip `guarded`)
    assert r.stdout.count("caught 1") == 2, r.stdout


if __name__ == "__main__":
    if "--write" in sys.argv:
        json.dump(_all(), open(GOLD, "w"), indent=1, sort_keys=True)
        print(open(GOLD).read())
