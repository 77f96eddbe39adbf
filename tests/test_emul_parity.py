"""Kernel LOGIC vs oracle on the CPU: the device headers compiled as a 1-lane wavefront
(tests/emul/emul.cpp, test harness only) must reproduce the oracle bit for bit -- tables, per-window
consensus, fragments.  The real 64-lane gfx950 build is checked by test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest
import pyoracle
import emul_lib
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
from common import windows_equal, frags_equal


def _both(d, ovl, piles, sel, **kw):
    p = default_params(**kw)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[sel], ovl, d.trace, nthreads=4, want_windows=True)
    fe, be = E.run(piles[sel], ovl, d.trace)
    return O, E, (fo, bo), (fe, be)


def test_tables_bit_identical():
    for n, kw in enumerate((dict(k=8), dict(klow=8, khigh=10), dict(k=14, w=32, a=8))):
        p = default_params(**kw)
        O = pyoracle.Oracle(p); E = emul_lib.Emul(p)
        for prof in ((0.12, 0.02, 0.85), (0.05, 0.05, 0.85), (0.01, 0.002, 0.98))[:3 if n == 0 else 1]:
            O.set_error_profile(*prof); E.set_error_profile(*prof)
            assert (O.tables(200) == E.tables(200)).all()


@pytest.mark.parametrize("kw", [dict(k=8), dict(k=14), dict(klow=8, khigh=9), dict(k=8, maxalign=6),
                                dict(k=8, producefull=1), dict(k=10, w=32, a=8), dict(k=8, minlen=3000)])
def test_windows_and_fragments(small_data, kw):
    d, ovl, piles = small_data
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 3), **kw)
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


def test_high_error_exercises_gap_filling_and_failures():
    d = SynthData(60000, 150, 3000, erate=0.30, seed=4)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 3), k=10)
    wo = O.windows()
    assert (wo["status"] == 2).any()                      # failed windows occur
    assert ((wo["status"] == 1) & (wo["filterfreq"] == 0)).any()   # found only after gap filling
    assert windows_equal(wo, E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


def test_empty_and_shallow_piles(small_data):
    d, ovl, piles = small_data
    p = piles[:2].copy()
    p[0]["novl"] = 0                                       # empty pile: no windows, no fragments
    p[1]["novl"] = 1                                       # A + 1 B = depth 2 < -m 3 everywhere
    O, E, (fo, bo), (fe, be) = _both(d, ovl, p, slice(0, 2), k=8)
    assert len(fo) == 0 and frags_equal(fo, bo, fe, be)
    assert windows_equal(O.windows(), E.windows()) == []
    # -f copies the uncorrected read for a pile WITH overlaps only (ita != ite, HandleContext.hpp:2543): the empty pile
    # stays silent, the shallow one comes back in lower case
    O, E, (fo, bo), (fe, be) = _both(d, ovl, p, slice(0, 2), k=8, producefull=1)
    assert len(fo) == 1 and fo[0]["aread"] == p[1]["aread"] and bo == bo.lower() and frags_equal(fo, bo, fe, be)


@pytest.mark.parametrize("dense", ["1", "0"])
def test_capacity_tiers_and_generic_engine_agree(small_data, dense, monkeypatch):
    """k=8 on 20x piles overflows the small LDS layouts for some windows: every capacity tier has to be exercised, and
    the generic engine alone (fast path off) has to give the same bits.  Round 6: what the second slot hands on goes through the
    dense-graph tier (FastTier<10>, two wavefronts per CU) before tier 3 (one per CU); DACC_DENSE_TIER=0 = straight to tier 3."""
    monkeypatch.setenv("DACC_DENSE_TIER", dense)
    d, ovl, piles = small_data
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 4), k=8)
    t1, t2, t3, gen = E.counts()
    assert t1 > 0 and t2 > 0 and ((t3 > 0 and E.count_tier10() == 0) if dense == "0" else E.count_tier10() > 0), (t1, t2, t3, gen, E.count_tier10())
    assert windows_equal(O.windows(), E.windows()) == [] and frags_equal(fo, bo, fe, be)
    G = emul_lib.Emul(default_params(k=8)); G.set_fast(False)
    G.set_error_profile(*d.error_profile()); G.load_db(d.bps, d.boff, d.rlen)
    fg, bg = G.run(piles[0:2], ovl, d.trace)
    assert G.counts()[3] > 0 and G.counts()[0] == 0
    fo2, bo2 = O.run(piles[0:2], ovl, d.trace, nthreads=4)
    assert frags_equal(fo2, bo2, fg, bg)


@pytest.mark.parametrize("hand", ["1", "0"])
def test_hand_over_with_and_without_the_sorted_instances(small_data, hand, monkeypatch):
    """Round 4: a window that overflows a tier's node table leaves its sorted k-mer instances in a hand-over slot and the next
    tier loads them instead of sorting again (DACC_HAND=0: every hand-over restarts from the strings).  Both ways, at a
    size-class threshold that sends everything to tier 0 first (many hand-overs), the bits are the oracle's.
    (k = 10 since round 6: at k = 14 no window of this 10x set overflows tier 0 any more -- its hand-overs had been reverse pool
    overflows, which the chunks of two removed.)"""
    d, ovl, piles = small_data
    monkeypatch.setenv("DACC_HAND", hand); monkeypatch.setenv("DACC_T0INST", "100000")
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 6), k=10)
    assert windows_equal(O.windows(), E.windows()) == [] and frags_equal(fo, bo, fe, be)
    t1, t2, t3, gen = E.counts()
    assert E.count_tier0() > 0 and t1 > 0


def test_no_state_leaks_between_windows(small_data, monkeypatch):
    """The harness fills the LDS image and the engine's members with a poison byte before every window: a window must
    not depend on what an earlier window (or kernel) left behind (such a leak corrupted later windows on the GPU once)."""
    d, ovl, piles = small_data
    monkeypatch.setenv("DACC_EMUL_POISON", "171")
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 3), k=8)
    assert windows_equal(O.windows(), E.windows()) == [] and frags_equal(fo, bo, fe, be)


def test_dense_graph_overflows_the_generic_scratch_and_is_rerun(monkeypatch):
    """k=6 with -d3 and gap filling: the generic engine's first scratch sizes are too small for some windows; the
    windows are run again with grown capacities (same in the library) and must still equal the oracle.
    (DACC_LONG128=0: the windows with strings of 65 ... 128 bases take the second stream as in rounds 3-5 -- tier 5 cannot hold these
    dense graphs and the generic engine gets them; on the default route of round 6 they finish in tiers 6 / 3.)"""
    monkeypatch.setenv("DACC_LONG128", "0")
    d = SynthData(60000, 300, 3000, erate=0.28, seed=481075, ins_frac=0.2, del_frac=0.7, sub_frac=0.1)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 2), k=6, w=63, a=20, maxalign=3)
    assert E.counts()[3] > 0                               # windows with strings > 64 bases: generic engine
    assert windows_equal(O.windows(), E.windows()) == [] and frags_equal(fo, bo, fe, be)


def test_deep_piles_reach_later_tiers_first_and_then_need_the_generic_engine():
    """50x piles, w=56: windows with more strings than tier 1 holds reach tier 2/3 first and many of them then turn out
    to have a string > 64 bases.  The harness mirrors the library's list orchestration (early generic list read once
    after the first tier): every window must be processed by exactly one engine and equal the oracle (a hand-over
    into the early list after it had been read lost such windows on the GPU once)."""
    d = SynthData(30000, 300, 5000, erate=0.08, seed=699273, ins_frac=0.2, del_frac=0.7, sub_frac=0.1, tspace=64, min_overlap=200)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E, (fo, bo), (fe, be) = _both(d, ovl, piles, slice(0, 1), w=56, a=16, klow=7, khigh=9, minfilterfreq=1, tspace=64)
    t1, t2, t3, gen = E.counts()
    assert t3 > 0 and gen > 0 and t1 == 0
    assert windows_equal(O.windows(), E.windows()) == [] and frags_equal(fo, bo, fe, be)


@pytest.mark.parametrize("kw", [dict(k=8), dict(k=14), dict(klow=8, khigh=9, maxalign=6)])
def test_64_lane_wavefront_emulation(small_data, kw):
    """The same kernel headers as a real 64-lane wavefront on the host (coroutine per lane, checked collectives):
    ballot / scan / shuffle / barrier code paths, not only the 1-lane logic."""
    d, ovl, piles = small_data
    p = default_params(**kw)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p, lanes=64); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[3:5], ovl, d.trace, nthreads=4, want_windows=True)
    fe, be = E.run(piles[3:5], ovl, d.trace)
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


@pytest.mark.parametrize("lanes", [1, 64])
def test_deep_batch_starts_in_the_deep_tier(lanes):
    """50x piles at the default window parameters: the plan finds most windows too deep for tier 1 (more than 40 strings
    or 1024 k-mer instances) and starts the batch in the deep tier (FastTier<4>: 96 strings, 2048 instances, small
    graph, three wavefronts per CU); what does not fit there goes on through tiers 2 and 3 as usual.  Same bits as the
    oracle, and no window needs the generic engine only because the pile is deep."""
    d = SynthData(30000, 300, 5000, seed=7)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    p = default_params(k=14)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p, lanes=lanes); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[10:11], ovl, d.trace, nthreads=4, want_windows=True)
    fe, be = E.run(piles[10:11], ovl, d.trace)
    wo = O.windows()
    assert (wo["mao"] > 40).mean() > 0.5
    t1, t2, t3, gen = E.counts()
    assert t1 > 0.7 * len(wo) and gen <= 1, (t1, t2, t3, gen, len(wo))
    assert windows_equal(wo, E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


@pytest.mark.parametrize("tspace", [126, 200, 300])
def test_wide_trace_spacing(tspace):
    """tspace > 125: two byte trace values; > 128: the trace kernel's wide column vectors (4 words up to 256, 8 up to
    512), the reference takes any spacing (daccord.cpp:1375).  Window boundaries, windows and fragments as the oracle's."""
    d = SynthData(60000, 150, 4000, seed=21, tspace=tspace)
    assert d.trace_bytes == 2
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    p = default_params(k=8, tspace=tspace)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[0:3], ovl, d.trace, trace_bytes=2, nthreads=4, want_windows=True)
    fe, be = E.run(piles[0:3], ovl, d.trace, trace_bytes=2)
    assert len(bo) > 3000
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


def test_read_positions_behind_the_table_support(monkeypatch):
    """An error profile much narrower than the data (estimated on clean reads, applied to noisy ones): strings longer than
    the support of the model table put k-mer instances at positions the table has no row for.  They carry no weight
    (getKmerPositionWeight, DebruijnGraph.hpp:3826-3864: `pos < first + size`); the LDS engine used to read whatever lay
    behind its table copy for them.  With a poisoned LDS image the windows must still equal the oracle's."""
    monkeypatch.setenv("DACC_EMUL_POISON", "171")
    d = SynthData(60000, 200, 4000, erate=0.30, seed=32, ins_frac=0.9, del_frac=0.05, sub_frac=0.05)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    prof = (0.01, 0.002, 0.98)
    p = default_params(k=8)
    O = pyoracle.Oracle(p); O.set_error_profile(*prof); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p); E.set_error_profile(*prof); E.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[0:2], ovl, d.trace, nthreads=4, want_windows=True)
    fe, be = E.run(piles[0:2], ovl, d.trace)
    wo = O.windows()
    assert (wo["status"] == 1).sum() > 500
    assert windows_equal(wo, E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


@pytest.mark.parametrize("lanes,route", [(1, "tiers"), (64, "tiers"), (1, "tier5"), (64, "tier5")])
def test_windows_with_long_strings(lanes, route, monkeypatch):
    """B window strings of 65..128 bases (w = 56, insertion-rich reads).  Round 6: tiers 6 and 3 hold such strings (two words per pattern
    mask in the gw layout), the pre-scan puts these windows on the list the second slot reads and nothing of them reaches the second stream.
    DACC_LONG128=0 is the route of rounds 3-5: tier 5 (string stride 128, the LDS of a whole CU) on the second stream, the generic engine
    for what it cannot hold.  Both ways the bits are the oracle's."""
    d = SynthData(100000, 200, 5000, erate=0.25, seed=77, ins_frac=0.9, del_frac=0.05, sub_frac=0.05)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    if route == "tier5":
        monkeypatch.setenv("DACC_LONG128", "0")
    p = default_params(k=10, w=56, a=14)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p, lanes=lanes); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    sel = slice(0, 2 if lanes == 1 else 1)
    fo, bo = O.run(piles[sel], ovl, d.trace, nthreads=4, want_windows=True)
    fe, be = E.run(piles[sel], ovl, d.trace)
    t1, t2, t3, generic = E.counts()
    if route == "tier5":
        assert E.count_long() > 20 and generic <= E.count_long() // 10
    else:
        assert E.count_long() == 0 and t2 + t3 > 20 and generic <= (t2 + t3) // 10, (E.count_long(), E.counts())
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be)


@pytest.mark.parametrize("lanes", [1, 64])
def test_window_strings_beyond_128_bases(lanes):
    """Badly aligned trace blocks (a block of 100 A bases against more than 200 B bases): at w = 63 the windows inside
    have B strings of 129..256 bases, which only the generic engine holds (string stride LSTR = 256, block-wise Myers
    over four words for the candidate errors); the reference has no limit there (std::string windows)."""
    from common import warp_trace
    d = SynthData(100000, 200, 5000, seed=1)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    tr = warp_trace(ovl, piles, d.trace, [0, 1])
    p = default_params(k=8, w=63, a=16)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p, lanes=lanes); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    sel = slice(0, 2 if lanes == 1 else 1)
    fo, bo = O.run(piles[sel], ovl, tr, nthreads=4, want_windows=True)
    fe, be = E.run(piles[sel], ovl, tr)
    assert E.counts()[3] > 5       # the generic engine ran them (tier 5 takes strings of up to 128 bases only; round 6: windows of 65 ... 128 run in tiers 6 / 3)
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be) and len(bo) > 3000


@pytest.mark.parametrize("lanes", [1, 64])
def test_window_strings_beyond_256_bases(lanes):
    """Two byte trace values (tspace 126) with blocks of 126 A bases against 700 B bases: window strings of more than 256
    bases.  The host plan sizes the generic engine's string stride from the trace values (ArenaCaps::lstr) and the
    candidate errors run the block-wise Myers with its column state in the arena."""
    from common import warp_trace
    d = SynthData(100000, 200, 5000, seed=1, tspace=126)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    tr = warp_trace(ovl, piles, d.trace, [0, 1], every=4, extra=580, cap=2000)
    assert tr.dtype == np.uint16 and tr.max() > 600
    p = default_params(k=8, w=63, a=16, tspace=126)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p, lanes=lanes); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    sel = slice(0, 2 if lanes == 1 else 1)
    fo, bo = O.run(piles[sel], ovl, tr, trace_bytes=2, nthreads=4, want_windows=True)
    fe, be = E.run(piles[sel], ovl, tr, trace_bytes=2)
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be) and len(bo) > 3000


def test_long_strings_make_long_stretches():
    """k = 14 on window strings of several hundred bases: an unbranched stretch runs as far as a string does, so the
    enumerations' per-base-length heaps (ArenaCaps::blcap) are sized with the string stride (found by the CPU fuzzing
    with warped traces: flags 0x800 at the former fixed 256)."""
    from common import warp_trace
    d = SynthData(100000, 200, 5000, seed=1, tspace=126)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    tr = warp_trace(ovl, piles, d.trace, [0, 1], every=4, extra=580, cap=2000)
    p = default_params(k=14, w=40, a=20, tspace=126)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(piles[:2], ovl, tr, trace_bytes=2, nthreads=4, want_windows=True)
    fe, be = E.run(piles[:2], ovl, tr, trace_bytes=2)
    assert windows_equal(O.windows(), E.windows()) == []
    assert frags_equal(fo, bo, fe, be) and len(bo) > 3000


def test_twice_split_stretch_stays_in_the_lds_tiers():
    """A (first, last) k-mer pair that splits one stretch at two different nodes needs the MIDDLE piece of that stretch, which
    the gw layout creates together with the candidates' pieces (FastEngine::findCandidatesAndPieces, makePiece) so that it gets its
    feasibility with all other stretches, before the enumeration pools are laid over the node tables.  While the gw tiers handed such windows on instead, 0.044 % of the windows of config 2
    ended in the generic engine and took it longer than all other windows together (profiles/r03e_*): none may get there,
    and the piles must still carry the oracle's digests (tests/golden/scale_cfg2.json, first 60 piles of the bench's data set)."""
    import json, os
    from daccord_amd import engine
    from scale_cases import CASES, make_case, pile_digests
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scale_cfg2.json")))
    d, ovl, piles, sel = make_case(dict(CASES["cfg2"]), pyoracle.pile_select)
    sel = sel[:60]
    run = G["runs"][0]
    E = emul_lib.Emul(default_params(**run["params"])); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E.run(sel, ovl, d.trace)
    assert pile_digests(fx, bx, sel, engine.fasta) == run["pile_sha256"][:60]
    t1, t2, t3, generic = E.counts()
    assert generic == 0 and E.count_tier0() + E.count_tier7() + t1 > 50000, (E.count_tier0(), E.count_tier7(), E.counts())


@pytest.mark.parametrize("lanes,kw,erate", [(1, dict(w=100, a=25, k=8), 0.12), (1, dict(w=128, a=32, k=12), 0.15), (1, dict(w=65, a=16, k=8), 0.15),
                                            (1, dict(w=96, a=24, k=9, producefull=1), 0.2), (1, dict(w=110, a=55, k=10, minwindowcov=4), 0.1),
                                            (1, dict(w=128, a=16, klow=13, khigh=14), 0.12),
                                            (64, dict(w=128, a=64, k=8, maxalign=6), 0.15), (64, dict(w=80, a=10, k=10), 0.08)])
def test_wide_windows(lanes, kw, erate, monkeypatch):
    """-w 65..128 (free in the reference, src/daccord.cpp:1282-1305): the two-word consensus -> A alignment, 640 byte window records with
    16 bit group offsets and the vote over them.  Round 6: w = 64 ... 127 run in the wide LDS tiers (FastTier<8>, then <9>: two-word
    feasibility masks, candidates of up to 128 symbols) in front of the generic engine, which rounds 4-5 ran them in alone and which
    still takes w = 128 and DACC_WIDE_TIER=0.  The oracle equals the reference build at these sizes (tests/test_oracle_vs_ref.py)."""
    generic_only = kw["w"] == 128 or kw.get("a") == 55
    if kw.get("a") == 55:
        monkeypatch.setenv("DACC_WIDE_TIER", "0")
    d = SynthData(60000, 150, 3000, seed=kw["w"] + kw.get("k", 13), erate=erate)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    p = default_params(**kw)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = emul_lib.Emul(p, lanes=lanes); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    n = 2 if "klow" in kw else 4                            # (the k range at w = 128 is the slowest case)
    fo, bo = O.run(piles[:n], ovl, d.trace, nthreads=4, want_windows=True)
    fe, be = E.run(piles[:n], ovl, d.trace)
    wo = O.windows()
    assert (wo["status"] == 1).sum() > 50 and (wo["conslen"] > 64).any()
    assert windows_equal(wo, E.windows()) == []
    assert len(bo) > 1200 * n and frags_equal(fo, bo, fe, be)
    t1, t2, t3, generic = E.counts()
    if generic_only:
        assert (t1, t2, t3) == (0, 0, 0)                    # w = 128 (a model table of 129 rows) and DACC_WIDE_TIER=0: generic engine only
    else:
        assert t1 == 0 and t2 > 0 and t2 + t3 > generic, (t1, t2, t3, generic)      # the wide tiers finish most of the windows (k = 8: dense graphs, two thirds)


def test_wide_window_tables_bit_identical():
    for kw in (dict(k=8, w=100, a=25), dict(klow=10, khigh=12, w=128, a=32)):
        p = default_params(**kw)
        O = pyoracle.Oracle(p); E = emul_lib.Emul(p)
        for prof in ((0.12, 0.02, 0.85), (0.03, 0.01, 0.95)):
            O.set_error_profile(*prof); E.set_error_profile(*prof)
            assert (O.tables(200) == E.tables(200)).all()


def test_contexts_on_concurrent_threads(small_data):
    """bench.py's like-for-like leg: one context per thread, all started together.  The arena guard sink of the emulation build is
    one pointer per process; contexts that reach their sizing carve together used to push into each other's guard list (a double free
    once in some dozen bench runs, round 5) -- tests/emul/emul.cpp now makes the sizing carve exclusive.  Same output as one thread."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    d, ovl, piles = small_data
    p = default_params(k=8)
    def work(sel):
        E = emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
        fe, be = E.run(piles[sel], ovl, d.trace)
        return hashlib.sha256(pyoracle.fasta(fe, be).encode()).hexdigest()
    emul_lib.lib(1)
    sels = [slice(i % 3, i % 3 + 1) for i in range(12)]
    want = [work(s) for s in sels[:3]]
    for _ in range(3):
        with ThreadPoolExecutor(12) as ex:
            got = list(ex.map(work, sels))
        assert got == [want[i % 3] for i in range(12)]
