"""Error profile estimation: the product's host estimator (include/daccord_hip.h: dacc_eprof_*,
daccord_amd/csrc/host_eprof.cpp) against the oracle's restatement of src/daccord.cpp:271-631, 1653-1878 -- counts and
rates bit for bit -- and against the known error mix of the synthetic reads."""
import numpy as np
import pytest
import pyoracle
from daccord_amd import io as dio
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData


@pytest.mark.parametrize("kw,twodb,maxalign", [(dict(seed=1), False, 2 ** 64 - 1), (dict(seed=7, ins_frac=1 / 3., del_frac=1 / 3., sub_frac=1 / 3.), True, 6),
                                               (dict(seed=9, erate=0.08, tspace=64), False, 2 ** 64 - 1)])
def test_estimator_matches_oracle(kw, twodb, maxalign):
    d = SynthData(60000, 120, 3000, **kw)
    tspace = kw.get("tspace", 100)
    op, pp = pyoracle.select_lowest(d.ovl, d.piles), dio.select_lowest(d.ovl, d.piles)
    assert (op[0] == pp[0]).all() and (op[1] == pp[1]).all()
    ovl, piles = pp
    n = 24
    O = pyoracle.Oracle(default_params(k=8, tspace=tspace)); O.load_db(d.bps, d.boff, d.rlen)
    co, uo, no, po = O.estimate_profile(piles[:n], ovl, d.trace, maxalign=maxalign, two_databases=twodb)
    cx, ux, nx, px = dio.estimate_profile(d.bps, d.boff, d.rlen, tspace, piles[:n], ovl, d.trace, maxalign=maxalign, two_databases=twodb, nthreads=3)
    assert list(co) == list(cx) and (uo, no) == (ux, nx)
    assert po == px
    assert ux > 100 and cx[0] > 10000


def test_estimate_is_close_to_the_truth():
    d = SynthData(100000, 200, 5000, seed=3)          # 15 % errors: ins 80 %, del 13.3 %, sub 6.7 %
    ovl, piles = dio.select_lowest(d.ovl, d.piles)
    c, us, un, (p_i, p_d, cor) = dio.estimate_profile(d.bps, d.boff, d.rlen, 100, piles[:40], ovl, d.trace, nthreads=4)
    t_i, t_d, t_cor = d.error_profile()
    assert abs(p_i - t_i) < 0.03 and abs(p_d - t_d) < 0.02 and abs(cor - t_cor) < 0.04, ((p_i, p_d, cor), (t_i, t_d, t_cor))


def test_estimator_with_two_byte_trace_values():
    """tspace 200: two byte trace values (as DALIGNER writes them beyond 125) through the estimator's own block alignments."""
    d = SynthData(60000, 120, 3000, seed=11, tspace=200)
    assert d.trace_bytes == 2
    ovl, piles = dio.select_lowest(d.ovl, d.piles, trace_bytes=2)
    oo, po = pyoracle.select_lowest(d.ovl, d.piles)
    assert (oo == ovl).all() and (po == piles).all()
    n = 16
    O = pyoracle.Oracle(default_params(k=8, tspace=200)); O.load_db(d.bps, d.boff, d.rlen)
    co, uo, no, pro = O.estimate_profile(piles[:n], ovl, d.trace, trace_bytes=2)
    cx, ux, nx, prx = dio.estimate_profile(d.bps, d.boff, d.rlen, 200, piles[:n], ovl, d.trace, trace_bytes=2, nthreads=3)
    assert list(co) == list(cx) and (uo, no) == (ux, nx) and pro == prx
    assert ux > 50


def test_estimator_skips_a_malformed_pile():
    """One pile with a broken record (trace values that do not add up to the B span) must not end the estimation: the
    result equals the estimate over the other piles (the correction path drops only that pile too)."""
    d = SynthData(60000, 120, 3000, seed=1)
    ovl, piles = dio.select_lowest(d.ovl, d.piles)
    n = 12
    bad = ovl.copy()
    z = int(piles[3]["first_ovl"]) + 1
    bad[z]["bepos"] += 7
    good_piles = np.concatenate([piles[:3], piles[4:n]])
    c0, u0, n0, p0 = dio.estimate_profile(d.bps, d.boff, d.rlen, 100, good_piles, ovl, d.trace, nthreads=2)
    c1, u1, n1, p1 = dio.estimate_profile(d.bps, d.boff, d.rlen, 100, piles[:n], bad, d.trace, nthreads=2)
    assert list(c0) == list(c1) and (u0, n0) == (u1, n1) and p0 == p1
    assert dio.estimate_profile.last_skipped == (1, n)      # the skip is reported (dacc_eprof_skipped), not silent
    dio.estimate_profile(d.bps, d.boff, d.rlen, 100, good_piles, ovl, d.trace, nthreads=2)
    assert dio.estimate_profile.last_skipped == (0, n - 1)


def test_deep_profile_matches_oracle_and_cli(tmp_path):
    """--deepprofileonly (src/daccord.cpp:1442-1650, handleIndelEstimateDeep :634-995): round(error rate * (2^32-1)) of every
    estimator window that got a consensus -- product (dacc_eprof_deep) against the oracle's restatement, value for value, and
    the front end's `[deep] <rate> <cumulative fraction>` lines against the same values."""
    import os, subprocess
    d = SynthData(60000, 120, 3000, seed=1)
    ovl, piles = dio.select_lowest(d.ovl, d.piles)
    n = 40
    O = pyoracle.Oracle(default_params(k=8, tspace=100)); O.load_db(d.bps, d.boff, d.rlen)
    want = O.deep_profile(piles[:n], ovl, d.trace)
    got = dio.estimate_profile(d.bps, d.boff, d.rlen, 100, piles[:n], ovl, d.trace, nthreads=3, deep=True)[4]
    assert len(want) > 200 and (np.diff(want.astype(np.int64)) >= 0).all()
    assert len(got) == len(want) and (got == want).all()
    # the front end on files (host code only up to this point: it exits before a device context is created)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "daccord_amd", "daccord_hip")
    db, las = str(tmp_path / "r.db"), str(tmp_path / "r.las")
    dio.write_db(db, d.bps, d.boff, d.rlen); dio.write_las(las, 100, d.ovl, d.trace)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, "daccord_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    p = subprocess.run([exe, "--deepprofileonly", "-V0", "-I0,%d" % (n - 1), las, db], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr[-500:]
    lines = [l.split("\t") for l in p.stdout.splitlines()]
    assert lines and all(l[0] == "[deep]" for l in lines)
    vals, cnts = np.unique(want, return_counts=True)
    assert len(lines) == len(vals)
    cum = np.cumsum(cnts) / float(len(want))
    for (tag, rate, frac), v, c in zip(lines, vals, cum):
        assert float(rate) == float("%g" % (v / 4294967295.0)) and float(frac) == float("%g" % c)
