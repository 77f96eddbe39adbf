"""Boundary behaviour of the C ABI: a pile that cannot be processed is dropped and reported, the batch
goes on (the reference logs the exception of one read and continues, src/daccord.cpp:2464-2478); 2-byte trace values."""
import numpy as np
import pytest
import pyoracle
import emul_lib
from daccord_amd._structs import default_params
from common import frags_equal


def _corrupt(ovl, piles, trace, pi):
    """break the trace of the first overlap of pile pi: its B lengths no longer sum to bepos-bbpos"""
    tr = trace.copy()
    o = ovl[piles[pi]["first_ovl"]]
    tr[o["trace_off"] + 1] = (int(tr[o["trace_off"] + 1]) + 7) % 200
    return tr


def test_malformed_pile_is_dropped_not_fatal_emulation(small_data):
    d, ovl, piles = small_data
    p = default_params(k=8)
    tr = _corrupt(ovl, piles, d.trace, 1)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E.run(piles[:3], ovl, tr)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[[0, 2]], ovl, d.trace, nthreads=2)
    assert frags_equal(fo, bo, fx, bx)
    assert not (fx["aread"] == piles[1]["aread"]).any()


@pytest.mark.gpu
def test_malformed_pile_is_dropped_and_reported(small_data):
    from daccord_amd import engine
    d, ovl, piles = small_data
    p = default_params(k=8, producefull=1)
    tr = _corrupt(ovl, piles, d.trace, 2)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E(piles[:5], ovl, tr)
    st, msgs = E.pile_status()
    assert list(st) == [0, 0, -1, 0, 0]
    assert len(msgs) == 1 and ("read %d" % piles[2]["aread"]) in msgs[0]
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[[0, 1, 3, 4]], ovl, d.trace, nthreads=4)
    assert frags_equal(fo, bo, fx, bx)


@pytest.mark.gpu
def test_two_byte_trace_values(small_data):
    from daccord_amd import engine
    d, ovl, piles = small_data
    p = default_params(k=8)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    f1, b1 = E(piles[:4], ovl, d.trace)
    f2, b2 = E(piles[:4], ovl, d.trace.astype(np.uint16), trace_bytes=2)
    assert frags_equal(f1, b1, f2, b2) and len(b1)
    st, msgs = E.pile_status()
    assert (st == 0).all() and not msgs


def test_monstrous_trace_block_drops_its_pile_only_emulation():
    """Two byte trace values can name a block of thousands of B bases, more than the trace kernels' LDS column stores hold
    (2048 at tspace 300): the plan drops that pile (reported), the batch goes on."""
    from daccord_amd.synth import SynthData
    from common import warp_trace
    d = SynthData(100000, 200, 5000, seed=1, tspace=300)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    tr = warp_trace(ovl, piles, d.trace, [1], every=7, extra=2500, cap=5000)
    assert tr.max() > 2048
    p = default_params(k=8, tspace=300)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E.run(piles[:3], ovl, tr, trace_bytes=2)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[[0, 2]], ovl, d.trace, trace_bytes=2, nthreads=2)
    assert frags_equal(fo, bo, fx, bx)
    assert not (fx["aread"] == piles[1]["aread"]).any()


def test_long_trace_block_at_tspace_126_runs_the_wide_trace_kernel_emulation():
    """A block of more than 928 B bases at tspace <= 128 does not fit the two word trace kernel's 64 column stores: the batch
    runs k_trace_wide<4> (lanes sized to the LDS) and gives the oracle's result."""
    from daccord_amd.synth import SynthData
    from common import warp_trace, windows_equal
    d = SynthData(100000, 200, 5000, seed=1, tspace=126)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    tr = warp_trace(ovl, piles, d.trace, [0, 1], every=6, extra=1100, cap=3000)
    assert tr.max() > 928
    p = default_params(k=8, tspace=126)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E.run(piles[:2], ovl, tr, trace_bytes=2)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[:2], ovl, tr, trace_bytes=2, nthreads=4, want_windows=True)
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fx, bx) and len(bo) > 3000
