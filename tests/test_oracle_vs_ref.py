"""oracle/ (the CPU restatement every parity test of the HIP path is anchored on) against oracle/_ref = the REFERENCE'S OWN
hot-path headers (/root/reference/src/HandleContext.hpp, DebruijnGraph.hpp, OffsetLikely.hpp, DotProduct.hpp,
ComputeOffsetLikely.hpp, ... compiled where they lie, unmodified) on a libmaus2 stand-in (oracle/ref_shim/).  This pins the
restatement to the reference source; the libmaus2 primitives (heap sift order, aligner traceback, convolution, binomial)
remain our documented choices on both sides.

The library is built by __graft_entry__.build() / oracle/ref_shim/build.sh where /root/reference exists and travels as a
built file; without it these tests skip.  A larger run (100+ random parameter sets, 100 piles of BASELINE config 2 at k = 14)
is kept as a log under profiles/ (scripts/fuzz_oracle_vs_ref.py, scripts/oracle_vs_ref_scale.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import pyoracle  # noqa: E402
import pyref  # noqa: E402
from daccord_amd._structs import default_params  # noqa: E402

pyref.build()
pytestmark = pytest.mark.skipif(not pyref.available(), reason="oracle/_ref is not built (needs /root/reference, build container only)")


def _pair(p, d):
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    R = pyref.Reference(p); R.set_error_profile(*d.error_profile()); R.load_db(d.bps, d.boff, d.rlen)
    return O, R


@pytest.mark.parametrize("prof", [(0.12, 0.02, 0.85), (0.05, 0.05, 0.85), (0.01, 0.002, 0.98), (0.2, 0.05, 0.7)])
def test_model_tables_are_bit_identical(prof):
    """OffsetLikely (DP, DPnorm, DPnormSquare incl. the fixed point words, supports) and KmerLimit for k = 6..12 and w = 40, 63"""
    for w in (40, 63):
        p = default_params(w=w, klow=6, khigh=12)
        O = pyoracle.Oracle(p); O.set_error_profile(*prof)
        R = pyref.Reference(p); R.set_error_profile(*prof)
        to, tr = O.tables(), R.tables()
        assert len(to) == len(tr) and np.array_equal(to, tr)


@pytest.mark.parametrize("k", [8, 12])
def test_small_data_fasta_identical(small_data, k):
    """the reference's own container (k in [3,12]) on the small data set (all of it and more in profiles/r04_oracle_vs_ref_*.log)"""
    d, ovl, piles = small_data
    p = default_params(k=k)
    O, R = _pair(p, d)
    n = 60 if k == 8 else 20
    fo, bo = O.run(piles[:n], ovl, d.trace, nthreads=4)
    fr, br = R.run(piles[:n], ovl, d.trace, nthreads=4)
    assert len(fo) and pyoracle.fasta(fo, bo) == pyoracle.fasta(fr, br)
    for x, y in zip(fo, fr):
        assert (x["aread"], x["first"], x["last"], x["len"]) == (y["aread"], y["first"], y["last"], y["len"])


def test_k_range_and_options(small_data):
    """-k 8,10 (one graph per k, the lowest error wins), -f, -l, -m, -d, --minfilterfreq / --maxfilterfreq through the reference's handler"""
    d, ovl, piles = small_data
    for kw in (dict(klow=8, khigh=9), dict(k=9, producefull=1, minlen=500), dict(k=8, maxalign=6, minwindowcov=4), dict(k=10, maxfilterfreq=3, minfilterfreq=1, w=32, a=8, eminrate=15)):
        p = default_params(**kw)
        O, R = _pair(p, d)
        fo, bo = O.run(piles[:8], ovl, d.trace, nthreads=4)
        fr, br = R.run(piles[:8], ovl, d.trace, nthreads=4)
        assert pyoracle.fasta(fo, bo) == pyoracle.fasta(fr, br), kw


def test_config2_piles_at_k14():
    """BASELINE config 2 (the bench workload) at k = 14: the reference's graph template through our k <= 16 factory
    (its own container stops at 12); a few piles here, 100 in profiles/r04_oracle_vs_ref_cfg2.log"""
    if not pyref.available(k16=True):
        pytest.skip("k16 build of oracle/_ref missing")
    from scale_cases import CASES, make_case
    case = dict(CASES["cfg2"]); case["first"] = 5000; case["npiles"] = 4
    d, ovl, piles, sel = make_case(case, pyoracle.pile_select)
    p = default_params(k=14)
    O, R = _pair(p, d)
    fo, bo = O.run(sel, ovl, d.trace, nthreads=6)
    fr, br = R.run(sel, ovl, d.trace, nthreads=3)
    assert len(bo) > 30000 and pyoracle.fasta(fo, bo) == pyoracle.fasta(fr, br)


@pytest.mark.parametrize("kw,tspace,maxalign", [(dict(seed=1), 100, 2 ** 64 - 1), (dict(seed=7, ins_frac=1 / 3., del_frac=1 / 3., sub_frac=1 / 3.), 100, 6),
                                                (dict(seed=3, erate=0.08), 200, 2 ** 64 - 1)])
def test_error_profile_estimator_against_the_reference_functions(kw, tspace, maxalign):
    """handleIndelEstimate<8> / handleIndelEstimateDeep<8> of the reference (src/daccord.cpp:271-995, cut out of the driver's
    translation unit at build time) against oracle/o_eprof.hpp: counts, usable / unusable windows, rates, and the window error
    rates of --deepprofileonly.  (This comparison found that the A window always joins the estimator's strings, round 4.)"""
    from daccord_amd.synth import SynthData
    d = SynthData(60000, 120, 3000, tspace=tspace, **kw)
    ovl, piles = pyoracle.select_lowest(d.ovl, d.piles)
    n = 30
    p = default_params(k=8, tspace=tspace)
    O = pyoracle.Oracle(p); O.load_db(d.bps, d.boff, d.rlen)
    R = pyref.Reference(p); R.load_db(d.bps, d.boff, d.rlen)
    co, uo, no, po = O.estimate_profile(piles[:n], ovl, d.trace, trace_bytes=d.trace_bytes, maxalign=maxalign)
    cr, ur, nr, pr = R.estimate_profile(piles[:n], ovl, d.trace, trace_bytes=d.trace_bytes, maxalign=maxalign)
    assert list(co) == list(cr) and (uo, no) == (ur, nr) and po == pr and uo > 50
    do = O.deep_profile(piles[:n], ovl, d.trace, trace_bytes=d.trace_bytes, maxalign=maxalign)
    dr = R.estimate_profile(piles[:n], ovl, d.trace, trace_bytes=d.trace_bytes, maxalign=maxalign, deep=True)[4]
    assert len(do) == len(dr) and np.array_equal(do, dr)


@pytest.mark.parametrize("maxinput", [5000, 40, 7, 1])
def test_estimator_pile_selection_against_the_reference_loop(maxinput):
    """src/daccord.cpp:1712-1742 + :1758 (keep the maxinput lowest error scores, a replacing record takes the slot of the one it
    replaces, sort by abpos), compiled from its lines, against the oracle's and the product's restatements, record for record"""
    from daccord_amd.synth import SynthData
    from daccord_amd import io as dio
    d = SynthData(60000, 200, 3000, seed=4)
    oo, po = pyoracle.select_lowest(d.ovl, d.piles, maxinput=maxinput)
    orf, pr = pyref.select_lowest(d.ovl, d.piles, maxinput=maxinput)
    ox, px = dio.select_lowest(d.ovl, d.piles, maxinput=maxinput)
    assert len(oo) == len(orf) == len(ox) and np.array_equal(po, pr) and np.array_equal(po, px)
    assert oo.tobytes() == orf.tobytes() == ox.tobytes()
    assert len(oo) < len(d.ovl) if maxinput < 20 else True


@pytest.mark.parametrize("kw,erate", [(dict(w=100, a=25, k=8), 0.12), (dict(w=128, a=32, k=12), 0.15), (dict(w=65, a=16, k=8), 0.15), (dict(w=80, a=10, k=10), 0.08)])
def test_wide_windows_against_the_reference_build(kw, erate):
    """-w above 64 is plain reference semantics (src/daccord.cpp:1282-1305 takes any -w): tables and FASTA of the oracle equal the reference build there too"""
    from daccord_amd.synth import SynthData
    d = SynthData(60000, 150, 3000, seed=kw["w"] + kw["k"], erate=erate)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    p = default_params(**kw)
    O, R = _pair(p, d)
    to, tr = O.tables(), R.tables()
    assert len(to) == len(tr) and np.array_equal(to, tr)
    fo, bo = O.run(piles[:5], ovl, d.trace, nthreads=5)
    fr, br = R.run(piles[:5], ovl, d.trace, nthreads=5)
    assert len(bo) > 8000 and pyoracle.fasta(fo, bo) == pyoracle.fasta(fr, br)


def _selection_piles(seed, npiles, tbytes):
    """piles in file order with many ties in score and abpos, trace lengths that make records of 60...400 bytes, sizes from 0 to a few
    64 KiB input blocks (with records straddling the block ends)"""
    from daccord_amd._structs import DaccOverlap, DaccPile
    OVL_DTYPE, PILE_DTYPE = np.dtype(DaccOverlap), np.dtype(DaccPile)
    rng = np.random.default_rng(seed)
    sizes = [0, 1, 2, 3] + [int(x) for x in rng.integers(4, 2500, npiles - 4)]
    ovl = np.zeros(sum(sizes), dtype=OVL_DTYPE); piles = np.zeros(len(sizes), dtype=PILE_DTYPE); o = 0; toff = 0
    for i, n in enumerate(sizes):
        piles[i]["aread"] = i; piles[i]["first_ovl"] = o; piles[i]["novl"] = n
        seg = ovl[o:o + n]
        ab = rng.integers(0, 40, n) * 100                                  # few distinct start positions: ties for the unstable sort
        ln = rng.integers(1, 30, n) * 100
        seg["aread"] = i; seg["bread"] = rng.integers(0, 1000, n); seg["flags"] = rng.integers(0, 2, n)
        seg["abpos"] = ab; seg["aepos"] = ab + ln; seg["bbpos"] = rng.integers(0, 100, n); seg["bepos"] = seg["bbpos"] + ln
        seg["diffs"] = (ln * rng.integers(0, 6, n)) // 20                  # six distinct error rates: ties in the heap
        seg["tlen"] = 2 * rng.integers(10, 181, n) // tbytes
        seg["trace_off"] = toff + np.concatenate([[0], np.cumsum(seg["tlen"][:-1])]) if n else 0
        toff += int(seg["tlen"].sum()); o += n
    return ovl, piles


@pytest.mark.parametrize("maxinput,tbytes", [(5000, 1), (400, 1), (150, 2), (37, 1), (2, 2), (1, 1)])
def test_main_pile_selection_against_the_reference_lines(maxinput, tbytes):
    """src/daccord.cpp:2026-2105 + :2120-2288 (the MAIN path's selection: heap on the error score that evicts the LOWEST score when full,
    survivors copied last-input-block-backwards then earlier blocks forwards, unstable sort by abpos), compiled from its lines, against
    the oracle's restatement (oracle_pile_select) and the product's (dacc_pile_select), record for record.  The block parser under the
    reference's lines is ours (ref_select.cpp: a straddling record belongs to the block its last byte arrives in)."""
    from daccord_amd import engine
    ovl, piles = _selection_piles(11 + maxinput, 40, tbytes)
    oo, po = pyoracle.pile_select(ovl, piles, trace_bytes=tbytes, maxinput=maxinput)
    orf, pr, _ = pyref.pile_select(ovl, piles, trace_bytes=tbytes, maxinput=maxinput)
    ox, px = engine.pile_select(ovl, piles, trace_bytes=tbytes, maxinput=maxinput)
    assert np.array_equal(po, pr) and np.array_equal(po, px)
    assert oo.tobytes() == orf.tobytes() == ox.tobytes()
    assert (len(oo) < len(ovl)) == (maxinput < 2400)
    # piles larger than one input block were among them
    sz = 40 * piles["novl"] + np.array([int(ovl["tlen"][p["first_ovl"]:p["first_ovl"] + p["novl"]].sum()) * tbytes for p in piles])
    assert (sz > 3 * 65536).any() and (sz < 65536).any()


def test_main_pile_selection_vard_formula():
    """--vard: the reference's per-read cap lmaxinput (src/daccord.cpp:2121-2126, computed by its own lines) equals the product's
    (daccord_hip_main.cpp: the same expression), and the selection under it equals the oracle's"""
    ovl, piles = _selection_piles(5, 12, 1)
    rl = [0, 1, 10, 999, 5000, 10000, 12345, 20000, 40000, 7, 9999, 100001]
    for vard, avg in ((10, 10000.0), (3, 7777.5), (1, 1.0)):
        orf, pr, lm = pyref.pile_select(ovl, piles, maxinput=5000, vard=vard, rl=rl, avgreadlength=avg)
        mine = [max(max(2 * vard, 1), int((2.0 * float(vard) * float(r) / float(avg)) + 0.5)) for r in rl]
        assert lm[:len(mine)] == mine[:len(lm)]
        for i, p in enumerate(piles):
            oo, po = pyoracle.pile_select(ovl, piles[i:i + 1], maxinput=mine[i])
            assert oo.tobytes() == orf[pr[i]["first_ovl"]:pr[i]["first_ovl"] + pr[i]["novl"]].tobytes()


def test_option_defaults_are_the_reference_ones():
    """src/daccord.cpp:106-169 (getDefault*), compiled from its lines, against default_params() (the mirror the tests and bench.py use)
    and the usage text of the C++ front end"""
    import re
    d = pyref.defaults()
    p = default_params()
    assert (p.w, p.a, p.klow, p.khigh, p.minwindowcov, p.maxalign, p.eminrate, p.minlen, p.producefull, p.minfilterfreq, p.maxfilterfreq) == \
           (d["w"], d["a"], d["k"], d["k"], d["m"], d["d"], d["e"], d["l"], d["f"], d["minfilterfreq"], d["maxfilterfreq"])
    assert d["D"] == 5000 and d["vard"] == 0 and d["d"] == 2 ** 64 - 1 and d["e"] == 2 ** 64 - 1
    src = open(os.path.join(ROOT, "daccord_amd", "csrc", "daccord_hip_main.cpp")).read()
    m = re.search(r"uint32_t w = (\d+), a = (\d+), m = (\d+); uint64_t d = UINT64_MAX, e = UINT64_MAX, l = (\d+), D = (\d+), vard = (\d+);", src)
    assert m and tuple(int(x) for x in m.groups()) == (d["w"], d["a"], d["m"], d["l"], d["D"], d["vard"])
    m = re.search(r"uint32_t klow = (\d+), khigh = (\d+); int32_t minff = (\d+), maxff = (\d+);", src)
    assert m and tuple(int(x) for x in m.groups()) == (d["k"], d["k"], d["minfilterfreq"], d["maxfilterfreq"])


def test_read_interval_against_the_reference_lines():
    """-J part,parts (how N processes / GPUs share one overlap file) and -I first,last: src/daccord.cpp:1119-1224 + :1227 compiled
    from its lines against the product (dacc_read_interval, which the C++ front end calls) and daccord_amd.shard.shard_range (what
    bench.py shards with) -- values, the empty part, and which texts are refused"""
    from daccord_amd import io as dio
    from daccord_amd.shard import shard_range
    import itertools
    n = 0
    for (lo, hi) in ((0, 9999), (0, 0), (5, 4), (17, 1016), (0, -1), (3, 3), (100, 107)):
        for G in (1, 2, 3, 4, 7, 8, 16, 1000, 20000):
            for g in list(range(min(G, 9))) + [G - 1, G, G + 5]:
                J = "%d,%d" % (g, G)
                r = pyref.read_interval(lo, hi, J=J); x = dio.read_interval(lo, hi, J=J)
                assert r == x, (lo, hi, J, r, x)
                if hi >= lo and 0 <= g:
                    a, b = shard_range(lo, hi + 1, g, G)
                    assert (r[1] - r[0] if r[1] > r[0] else 0) == b - a and (b == a or (a, b) == r), (lo, hi, J, r, (a, b))
                n += 1
        for I in ("0,5", "3,3", "10,5", "-5,100000", "50,60", "9999,9999", "0,-1"):
            assert pyref.read_interval(lo, hi, I=I) == dio.read_interval(lo, hi, I=I), (lo, hi, I)
        # J wins over I (else-if)
        assert pyref.read_interval(lo, hi, J="1,2", I="0,1") == dio.read_interval(lo, hi, J="1,2", I="0,1")
    # the parts of -J g,G cover the A reads exactly once, in order
    for (lo, hi, G) in ((0, 9999, 8), (17, 1016, 7), (0, 6, 8), (3, 3, 2)):
        got = []
        for g in range(G):
            a, b = dio.read_interval(lo, hi, J="%d,%d" % (g, G))
            got += list(range(a, b)) if b > a else []
        assert got == list(range(lo, hi + 1))
    # what is refused, by both: not <int>,<int>; a zero denominator over a non-empty span
    for bad in ("", "1", "1,", ",2", "1;2", "1,2,3", "1,2x", "a,b", "1 ,2", "1,2 "):
        for kw in (dict(J=bad), dict(I=bad)):
            er = ex = None
            try:
                pyref.read_interval(0, 99, **kw)
            except ValueError as e:
                er = str(e)
            try:
                dio.read_interval(0, 99, **kw)
            except ValueError as e:
                ex = str(e)
            assert (er is None) == (ex is None), (kw, er, ex)
    for f in (pyref.read_interval, dio.read_interval):
        with pytest.raises(ValueError):
            f(0, 99, J="0,0")
        assert f(5, 4, J="0,0") == f(5, 4, J="0,1")       # empty span: the denominator is not looked at
    assert n > 500


def test_random_parameter_sets():
    """a few rounds of scripts/fuzz_oracle_vs_ref.py (narrow and wide) inside the CPU suite"""
    for args in (["20260922", "3"], ["22", "1", "--wide"]):      # (k = 14, 9, 8; a two-byte trace set; the k = 14...16 sets are in the profiles/ log)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_oracle_vs_ref.py")] + args, capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0 and "DONE bad=0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_product_never_touches_the_reference_build():
    """nothing under daccord_amd/ names oracle/_ref, the shim or the wrapper"""
    for dp, _, fs in os.walk(os.path.join(ROOT, "daccord_amd")):
        for f in fs:
            if f.endswith((".py", ".hpp", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyref" not in txt and "libdaccord_ref" not in txt and "ref_shim" not in txt, os.path.join(dp, f)
