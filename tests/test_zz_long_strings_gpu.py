"""GPU parity of what was written after round 2's last GPU run: B window strings beyond 128 / 256 bases in the generic engine
(string stride from the host plan).  Last in the collection order on purpose: what is here had only run
on the 1- and 64-lane emulation (tests/test_emul_parity.py) when it was written -- the long-string tests in round 2 (green on the
device since), the random wide-window sets in round 4 (the wide-window path itself ran on the device: profiles/r04t_*)."""
import numpy as np
import pytest
import pyoracle
from daccord_amd._structs import default_params
from common import frags_equal


@pytest.mark.gpu
def test_window_strings_beyond_128_bases(small_data):
    """Badly aligned trace blocks (100 A bases against more than 200 B bases, w = 63): B window strings of 129..256 bases
    run in the generic engine (LSTR = 256) instead of dropping their pile; same bits as the oracle, no pile reported."""
    from daccord_amd import engine
    from common import warp_trace, windows_equal
    d, ovl, piles = small_data
    tr = warp_trace(ovl, piles, d.trace, [0, 1])
    p = default_params(k=8, w=63, a=16)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E(piles[:2], ovl, tr)
    st, msgs = E.pile_status()
    assert (st == 0).all() and not msgs, msgs
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[:2], ovl, tr, nthreads=8, want_windows=True)
    assert windows_equal(O.windows(), E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx) and len(bo) > 3000


@pytest.mark.gpu
def test_window_strings_beyond_256_bases():
    """Two byte trace values, blocks of 126 A bases against 700 B bases: window strings of more than 256 bases (string
    stride from the host plan, Myers column state in the arena); same bits as the oracle, no pile dropped."""
    from daccord_amd import engine
    from daccord_amd.synth import SynthData
    from common import warp_trace, windows_equal
    d = SynthData(100000, 200, 5000, seed=1, tspace=126)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    tr = warp_trace(ovl, piles, d.trace, [0, 1], every=4, extra=580, cap=2000)
    p = default_params(k=8, w=63, a=16, tspace=126)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E(piles[:2], ovl, tr, trace_bytes=2)
    st, msgs = E.pile_status()
    assert (st == 0).all() and not msgs, msgs
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[:2], ovl, tr, trace_bytes=2, nthreads=8, want_windows=True)
    assert windows_equal(O.windows(), E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx) and len(bo) > 3000


@pytest.mark.gpu
def test_random_wide_window_sets():
    """Random parameter sets with a window of 64 ... 128 bases (tests/common.py:random_run_config_w128; generic engine only): the first
    six of the thirty sets the emulation ran against the oracle in profiles/r04t_cpu_fuzz_emul_w128.log (same seed)."""
    from common import random_run_config_w128
    from test_gpu_parity import _random_configs
    _random_configs(0, 6, seed=4128, gen=random_run_config_w128)
