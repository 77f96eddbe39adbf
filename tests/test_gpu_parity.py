"""Parity tests proper: the gfx950 kernels, called through the C ABI, against the oracle on the same
seeded inputs -- bit-exact per-window consensus and FASTA.  Run with -m gpu on an MI355X."""
import hashlib
import json
import os
import numpy as np
import pytest
import pyoracle
from daccord_amd import engine
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
from common import windows_equal, frags_equal, truth_error

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _pair(d, **kw):
    p = default_params(**kw)
    O = pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    return O, E


def test_library_loaded_is_the_hip_one():
    assert os.path.exists(engine._SO)
    L = engine.lib()
    assert hasattr(L, "dacc_rerun_resident")


def test_tables_bit_identical():
    p = default_params(klow=8, khigh=10)
    O = pyoracle.Oracle(p); E = engine.Engine(p)
    for prof in ((0.12, 0.02, 0.85), (0.05, 0.05, 0.85)):
        O.set_error_profile(*prof); E.set_error_profile(*prof)
        assert (O.tables(200) == E.tables(200)).all()


@pytest.mark.parametrize("kw", [dict(k=8), dict(k=14), dict(klow=8, khigh=9), dict(k=8, maxalign=6),
                                dict(k=8, producefull=1), dict(k=10, w=32, a=8), dict(k=16)])
def test_windows_and_fragments(small_data, kw):
    d, ovl, piles = small_data
    O, E = _pair(d, **kw)
    fo, bo = O.run(piles[:8], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[:8], ovl, d.trace)
    bad = windows_equal(O.windows(), E.debug_windows())
    assert bad == [], (len(bad), bad[:5])
    assert frags_equal(fo, bo, fx, bx)
    assert engine.fasta(fx, bx) == pyoracle.fasta(fo, bo)


def test_golden_fixture(small_data):
    with open(os.path.join(HERE, "golden", "oracle_small.json")) as f:
        G = json.load(f)
    d, ovl, piles = small_data
    E = engine.Engine(default_params(k=G["k"])); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E(piles[:G["npiles"]], ovl, d.trace)
    assert hashlib.sha256(engine.fasta(fx, bx).encode()).hexdigest() == G["fasta_sha256"]


def test_high_error_gap_filling_and_failures():
    d = SynthData(60000, 150, 3000, erate=0.30, seed=4)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E = _pair(d, k=10)
    fo, bo = O.run(piles[:12], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[:12], ovl, d.trace)
    wo = O.windows()
    assert (wo["status"] == 2).any() and ((wo["status"] == 1) & (wo["filterfreq"] == 0)).any()
    assert windows_equal(wo, E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx)


@pytest.mark.parametrize("kw", [dict(k=8), dict(klow=8, khigh=10), dict(k=12, producefull=1), dict(k=14), dict(k=16, w=48, a=12)])
def test_hip_equals_the_reference_build(small_data, kw):
    """The HIP path against oracle/_ref directly: the reference's OWN HandleContext.hpp / DebruijnGraph.hpp / OffsetLikely.hpp ...
    compiled (in the build container, unmodified) against the libmaus2 stand-in; the built library travels with the snapshot, the
    sources do not.  k <= 12 runs the reference's own graph container, larger k our factory around its DebruijnGraph<k>."""
    import pyref
    p = default_params(**kw)
    if not pyref.available(k16=(p.khigh > 12)):
        pytest.skip("oracle/_ref is not built (needs /root/reference at build time)")
    d, ovl, piles = small_data
    n = 24 if p.khigh <= 14 else 6          # (a DebruijnGraph<16> of the reference holds 16 GiB of node cache per context)
    R = pyref.Reference(p); R.set_error_profile(*d.error_profile()); R.load_db(d.bps, d.boff, d.rlen)
    E = engine.Engine(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    assert (R.tables(200) == E.tables(200)).all()
    fr, br = R.run(piles[:n], ovl, d.trace, nthreads=(8 if p.khigh <= 12 else (4 if p.khigh <= 14 else 1)))
    fx, bx = E(piles[:n], ovl, d.trace)
    assert len(bx) > 1000 and frags_equal(fr, br, fx, bx) and engine.fasta(fx, bx) == pyoracle.fasta(fr, br)
    E.close()


@pytest.mark.parametrize("t0inst", ["0", "300", "488", "100000", "100000/nohand"])
def test_size_class_threshold_does_not_change_results(small_data, t0inst, monkeypatch):
    """tier 0 (size classes, round 4): whatever share of the windows the pre-pass sends to the small tier -- none, some, all
    that its string capacity admits -- the per-window records and the FASTA are the oracle's; what tier 0 cannot hold is handed on"""
    d, ovl, piles = small_data
    if t0inst.endswith("/nohand"):
        # hand-over without restart switched off (DACC_HAND=0): every window a tier hands on is sorted again by the next one
        t0inst = t0inst.split("/")[0]; monkeypatch.setenv("DACC_HAND", "0")
    monkeypatch.setenv("DACC_T0INST", t0inst)
    O, E = _pair(d, k=14)
    fo, bo = O.run(piles[:30], ovl, d.trace, nthreads=8, want_windows=True); wo = O.windows()
    fx, bx = E(piles[:30], ovl, d.trace); wx = E.debug_windows(); t = E.timing()
    assert windows_equal(wo, wx) == [] and frags_equal(fo, bo, fx, bx)
    if t0inst == "0":
        assert t.tier0_in == 0
    else:
        assert t.tier0_in > 0 and t.tier0_ms > 0
    if t0inst == "100000":
        assert t.tier0_in > 0.9 * len(wx) and t.tier0_out > 0      # everything starts small, the big windows are handed on
    E.rerun(); f2, b2 = E.collect()
    assert frags_equal(fo, bo, f2, b2)
    E.close()


@pytest.mark.parametrize("t0,t7,tiers", [("200", "260", None), ("0", "300", None), ("200", "100000", None), ("200", "260", "15"), ("260", "200", None)])
def test_middle_size_class_does_not_change_results(small_data, t0, t7, tiers, monkeypatch):
    """tier 7 (round 6: the middle size class, 7 wavefronts per CU, between tier 0 and tier 1): whatever the two thresholds send where --
    nothing to tier 0, everything that is not small to tier 7, the class switched off (DACC_TIERS without bit 4, or a threshold below
    tier 0's) -- the per-window records and the FASTA are the oracle's, and the counters add up: tier 7 runs its class + tier 0's hand-overs"""
    d, ovl, piles = small_data
    monkeypatch.setenv("DACC_T0INST", t0); monkeypatch.setenv("DACC_T7INST", t7)
    if tiers:
        monkeypatch.setenv("DACC_TIERS", tiers)
    O, E = _pair(d, k=14)
    fo, bo = O.run(piles[:30], ovl, d.trace, nthreads=8, want_windows=True); wo = O.windows()
    fx, bx = E(piles[:30], ovl, d.trace); wx = E.debug_windows(); t = E.timing()
    assert windows_equal(wo, wx) == [] and frags_equal(fo, bo, fx, bx)
    off = tiers == "15" or int(t7) <= int(t0)
    if off:
        assert t.tier7_in == 0 and t.tier7_ms == 0
    else:
        assert t.tier7_in > 0 and t.tier7_ms > 0 and t.tier7_in >= t.tier0_out and t.tier7_out <= t.tier7_in
        if t0 == "0":
            assert t.tier0_in == 0 and t.tier0_out == 0
        if t7 == "100000":
            assert t.tier7_in + t.tier0_in - t.tier0_out == len(wx) - (int(t.long_windows) - int(t.long_first_tier))      # every window the pre-scan left starts in tier 0 or tier 7
    E.rerun(); f2, b2 = E.collect()
    assert frags_equal(fo, bo, f2, b2)
    E.close()


@pytest.mark.parametrize("adapt", ["1", "0"])
def test_middle_size_class_switches_itself_off_on_dense_graphs(small_data, adapt, monkeypatch):
    """tier 7 hands on most of what it runs at k = 8 (dense graphs overflow its stretch and pool tables): a context stops using it after
    such a batch (DACC_T7_ADAPT, default on) -- a throughput heuristic only: first pass, re-run and the pass of a second batch equal the oracle"""
    d, ovl, piles = small_data
    monkeypatch.setenv("DACC_T0INST", "0"); monkeypatch.setenv("DACC_T7INST", "100000"); monkeypatch.setenv("DACC_T7_ADAPT", adapt)
    O, E = _pair(d, k=8)
    fo, bo = O.run(piles[:24], ovl, d.trace, nthreads=8)
    fx, bx = E(piles[:24], ovl, d.trace); t = E.timing()
    assert frags_equal(fo, bo, fx, bx)
    assert t.tier7_in >= 4096 and t.tier7_out * 5 > t.tier7_in, (t.tier7_in, t.tier7_out)      # the premise: it overflows
    E.rerun(); f2, b2 = E.collect(); t2 = E.timing()
    assert frags_equal(fo, bo, f2, b2)
    assert (t2.tier7_in == 0) if adapt == "1" else (t2.tier7_in == t.tier7_in)
    fo3, bo3 = O.run(piles[24:30], ovl, d.trace, nthreads=8)
    f3, b3 = E(piles[24:30], ovl, d.trace)
    assert frags_equal(fo3, bo3, f3, b3)
    E.close()


@pytest.mark.parametrize("tspace", [126, 200, 300])
def test_wide_trace_spacing(tspace):
    """tspace > 125: two byte trace values; > 128: k_trace_wide<4> (up to 256) / <8> (up to 512)."""
    d = SynthData(60000, 150, 4000, seed=21, tspace=tspace)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E = _pair(d, k=8, tspace=tspace)
    fo, bo = O.run(piles[:6], ovl, d.trace, trace_bytes=2, nthreads=8, want_windows=True)
    fx, bx = E(piles[:6], ovl, d.trace, trace_bytes=2)
    assert len(bo) > 6000
    assert windows_equal(O.windows(), E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx)


def test_deep_batch_starts_in_the_deep_tier():
    """50x piles: the batch starts in the deep tier (k_window_fast<4>), same bits as the oracle."""
    d = SynthData(30000, 300, 5000, seed=7)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E = _pair(d, k=14)
    fo, bo = O.run(piles[10:14], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[10:14], ovl, d.trace)
    wo = O.windows()
    t = E.timing()
    assert (wo["mao"] > 40).mean() > 0.5 and t.tier_out[0] < 0.3 * len(wo), list(t.tier_out)
    assert windows_equal(wo, E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx)


@pytest.mark.parametrize("route", ["tiers", "tier5"])
def test_windows_with_long_strings(route, monkeypatch):
    """w = 56 on insertion-rich reads: a quarter of the windows have a B string of more than 64 bases.  Round 6: they join the list of the
    second slot (tiers 6 / 3 hold strings of up to 128 bases) and nothing of them takes the second stream; DACC_LONG128=0: k_window_long
    (tier 5, then the generic engine) as in rounds 3-5."""
    d = SynthData(100000, 200, 5000, erate=0.25, seed=77, ins_frac=0.9, del_frac=0.05, sub_frac=0.05)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    if route == "tier5":
        monkeypatch.setenv("DACC_LONG128", "0")
    O, E = _pair(d, k=10, w=56, a=14)
    fo, bo = O.run(piles[:3], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[:3], ovl, d.trace)
    t = E.timing()
    assert windows_equal(O.windows(), E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx)
    if route == "tier5":
        assert t.long_windows > 20
    else:
        assert t.long_windows == 0 and t.tier_out[0] > 20
    E.rerun(); f2, b2 = E.collect()
    assert frags_equal(fo, bo, f2, b2)
    E.close()


@pytest.mark.parametrize("kw,erate", [(dict(w=100, a=25, k=8), 0.12), (dict(w=128, a=32, k=12), 0.15), (dict(w=65, a=16, k=8), 0.15),
                                      (dict(w=96, a=24, k=9, producefull=1), 0.2), (dict(w=128, a=10, klow=13, khigh=14), 0.12), (dict(w=64, a=16, k=8), 0.12),
                                      (dict(w=104, a=26, k=14), 0.15), (dict(w=88, a=22, k=12), 0.18)])
def test_wide_windows(kw, erate):
    """-w 64..128 (free in the reference, src/daccord.cpp:1282-1305): two-word consensus -> A alignment, 640 byte window records with
    16 bit group offsets, the vote over them (the emulation runs the same cases: test_emul_parity.py).  Round 6: w = 64 ... 127 run in
    the wide LDS tiers (k_window_fast<8>, then <9>) in front of the generic engine, which still takes w = 128 alone."""
    d = SynthData(60000, 150, 3000, seed=kw["w"] + kw.get("k", 13), erate=erate)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E = _pair(d, **kw)
    fo, bo = O.run(piles[:6], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[:6], ovl, d.trace)
    wo = O.windows()
    assert (wo["status"] == 1).sum() > 50
    bad = windows_equal(wo, E.debug_windows())
    assert bad == [], (len(bad), bad[:5])
    assert len(bo) > 8000 and frags_equal(fo, bo, fx, bx)
    assert engine.fasta(fx, bx) == pyoracle.fasta(fo, bo)
    t = E.timing()
    if kw["w"] < 128:      # the wide tiers ran and finished most of the windows (w = 100 at k = 8, dense graphs: 641 of 702, a third of them in tier 8)
        assert t.tier_ms[1] > 0 and t.tier_out[1] < len(wo) and t.tier_out[2] + t.long_windows < len(wo) // 4, (t.tier_ms[1], t.tier_out[1], t.tier_out[2], t.long_windows, len(wo))
    else:
        assert t.tier_ms[1] == 0 and t.tier_ms[2] == 0
    # and once more on the same context (buffers sized for the wide records are reused)
    fy, by = E(piles[2:5], ovl, d.trace)
    f2, b2 = O.run(piles[2:5], ovl, d.trace, nthreads=8)
    assert frags_equal(f2, b2, fy, by)


def test_scratch_retry_of_the_generic_engine(monkeypatch):
    """A dense graph at k = 6 overflows the generic engine's first scratch capacities: the windows are run again in a re-carved (larger)
    arena.  With DACC_NOFAST=1 every window goes through the generic engine, as it does for w >= 64.  (Round 4: the per-length path heap
    counts were never cleared -- the first layout found zeros from the allocation, the re-carved one did not: a device fault.  The
    emulation now fills its arenas with a pattern instead of zeros.)"""
    monkeypatch.setenv("DACC_NOFAST", "1")
    d = SynthData(60000, 300, 3000, erate=0.28, seed=481075, ins_frac=0.2, del_frac=0.7, sub_frac=0.1)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O, E = _pair(d, k=6, w=63, a=20, maxalign=3)
    fo, bo = O.run(piles[:2], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[:2], ovl, d.trace)
    assert windows_equal(O.windows(), E.debug_windows()) == []
    assert len(bo) > 3000 and frags_equal(fo, bo, fx, bx)
    fy, by = E(piles[1:2], ovl, d.trace)
    f2, b2 = O.run(piles[1:2], ovl, d.trace, nthreads=8)
    assert frags_equal(f2, b2, fy, by)


def test_ont_like_profile_k_sweep():
    d = SynthData(60000, 150, 3000, erate=0.15, ins_frac=1 / 3, del_frac=1 / 3, sub_frac=1 / 3, seed=5)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    for k in (10, 12, 14, 16):
        O, E = _pair(d, k=k)
        fo, bo = O.run(piles[:4], ovl, d.trace, nthreads=8, want_windows=True)
        fx, bx = E(piles[:4], ovl, d.trace)
        assert windows_equal(O.windows(), E.debug_windows()) == []
        assert frags_equal(fo, bo, fx, bx)


@pytest.mark.parametrize("dense", ["1", "0"])
def test_dense_graph_tier_between_the_second_slot_and_tier_3(small_data, dense, monkeypatch):
    """Round 6: what the second slot's tier (k_window_fast<6>) hands on runs in the dense-graph tier k_window_fast<10> (two wavefronts
    per CU, 16 bit path ids) before tier 3 (one per CU); DACC_DENSE_TIER=0 = straight to tier 3.  k = 8 on 20x piles: dense graphs,
    every tier is exercised (the emulation runs the same case: test_capacity_tiers_and_generic_engine_agree)."""
    monkeypatch.setenv("DACC_DENSE_TIER", dense)
    d, ovl, piles = small_data
    O, E = _pair(d, k=8)
    fo, bo = O.run(piles[:4], ovl, d.trace, nthreads=8, want_windows=True)
    fx, bx = E(piles[:4], ovl, d.trace)
    assert windows_equal(O.windows(), E.debug_windows()) == []
    assert frags_equal(fo, bo, fx, bx)
    t = E.timing()
    assert t.tier_out[1] > 0, list(t.tier_out)
    if dense == "1":
        assert t.tier10_ran == 1 and t.tier10_ms > 0 and t.tier10_out < t.tier_out[1], (t.tier10_ran, t.tier10_ms, t.tier10_out, list(t.tier_out))
    else:
        assert t.tier10_ran == 0 and t.tier10_ms == 0
    E.rerun(); fy, by = E.collect()
    assert frags_equal(fo, bo, fy, by)


def test_empty_shallow_and_rerun(small_data):
    d, ovl, piles = small_data
    p = piles[:3].copy()
    p[0]["novl"] = 0; p[1]["novl"] = 1
    O, E = _pair(d, k=8)
    fo, bo = O.run(p, ovl, d.trace, nthreads=4)
    fx, bx = E(p, ovl, d.trace)
    assert frags_equal(fo, bo, fx, bx)
    E.rerun()                                             # idempotence on the resident batch
    fy, by = E.collect()
    assert frags_equal(fx, bx, fy, by)
    # -f: the uncorrected read is copied for piles that have overlaps only (HandleContext.hpp:2543)
    O, E = _pair(d, k=8, producefull=1)
    fo, bo = O.run(p, ovl, d.trace, nthreads=4)
    fx, bx = E(p, ovl, d.trace)
    assert frags_equal(fo, bo, fx, bx) and all(f["aread"] != p[0]["aread"] for f in fx)


def test_perfect_piles_full_size_property():
    """Size-independent property at a larger size: error-free piles return the reads themselves."""
    d = SynthData(400000, 800, 5000, erate=0.0, seed=6)
    ovl, piles = engine.pile_select(d.ovl, d.piles)
    E = engine.Engine(default_params(k=8)); E.set_error_profile(0.12, 0.02, 0.85); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E(piles[:200], ovl, d.trace)
    assert len(fx) >= 200
    for f in fx[::7]:
        rid = f["aread"]
        read = bytes(b"ACGT"[(d.bps[d.boff[rid] + (i >> 2)] >> (6 - 2 * (i & 3))) & 3] for i in range(d.rlen[rid]))
        assert bx[f["seq_off"]:f["seq_off"] + f["len"]] == read[f["first"]:f["last"] + 1]


def test_accuracy_vs_truth_larger():
    d = SynthData(400000, 800, 5000, seed=7)
    ovl, piles = engine.pile_select(d.ovl, d.piles)
    E = engine.Engine(default_params(k=8)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
    fx, bx = E(piles[:100], ovl, d.trace)
    L = pyoracle.lib()
    ed, n = truth_error(d, fx[:40], bx, lambda a, b: L.oracle_edit_distance(a, len(a), b, len(b)))
    assert n > 0 and ed / n < 0.02


@pytest.mark.gpu
def test_cli_from_las_and_db_files(small_data, tmp_path):
    """The C++ front end daccord_hip: .las + .db on disk -> FASTA identical to the oracle run on the in-memory piles;
    -I is inclusive at both ends (src/daccord.cpp:1225-1230)."""
    import pyoracle
    from daccord_amd import cli, io as dio
    d, ovl, piles = small_data
    las, db = str(tmp_path / "reads.las"), str(tmp_path / "reads.db")
    dio.write_db(db, d.bps, d.boff, d.rlen)
    dio.write_las(las, 100, d.ovl, d.trace)
    p_i, p_d, cor = d.error_profile()
    r = cli.run(["-k8", "-I0,6", "--eprof%r,%r,%r" % (p_i, p_d, cor), las, db])
    assert r.returncode == 0, r.stderr.decode()
    O = pyoracle.Oracle(default_params(k=8)); O.set_error_profile(p_i, p_d, cor); O.load_db(d.bps, d.boff, d.rlen)
    sel = piles[piles["aread"] <= 6]
    fo, bo = O.run(sel, ovl, d.trace, nthreads=4)
    assert r.stdout.decode() == pyoracle.fasta(fo, bo)
    # --gpus 3: three device workers (wrapping around on a box with fewer GPUs), batches of two A reads dealt to them, the
    # writer restores the order: byte-identical output, sequential well numbers included
    r3 = cli.run(["-k8", "-I0,6", "--gpus3", "--batch2", "--eprof%r,%r,%r" % (p_i, p_d, cor), las, db])
    assert r3.returncode == 0, r3.stderr.decode()
    assert r3.stdout == r.stdout
    assert sum(1 for ln in r3.stderr.decode().splitlines() if ln.startswith("[V] device worker ")) == 3
    # -J 1,3 = second third of the read range; --vard caps the depth per read (daccord.cpp:2120-2126)
    lo, hi = int(d.ovl["aread"].min()), int(d.ovl["aread"].max()) + 1
    part = (hi - lo + 2) // 3
    r = cli.run(["-k8", "-J1,3", "-I0,3", "--vard4", "-f", "--eprof%r,%r,%r" % (p_i, p_d, cor), las, db])
    assert r.returncode == 0, r.stderr.decode()
    avg = int(d.rlen.astype(np.int64).sum() // len(d.rlen))
    selp = d.piles[(d.piles["aread"] >= lo + part) & (d.piles["aread"] < min(hi, lo + 2 * part))]
    oo, pp, o = [], selp.copy(), 0
    for i, pl in enumerate(selp):
        cap = max(8, 1, int(2.0 * 4 * float(d.rlen[pl["aread"]]) / float(avg) + 0.5))
        so, sp = pyoracle.pile_select(d.ovl, selp[i:i + 1], maxinput=cap)
        oo.append(so); pp[i]["first_ovl"] = o; pp[i]["novl"] = len(so); o += len(so)
    O = pyoracle.Oracle(default_params(k=8, producefull=1)); O.set_error_profile(p_i, p_d, cor); O.load_db(d.bps, d.boff, d.rlen)
    fo, bo = O.run(pp, np.concatenate(oo), d.trace, nthreads=4)
    assert r.stdout.decode() == pyoracle.fasta(fo, bo)


def test_cli_two_databases_and_estimated_profile(small_data, tmp_path):
    """Asymmetric mode (src/daccord.cpp:1337-1364): B reads from a second database (here a copy of the first, so the
    output must equal the single database run); no --eprof: the profile is estimated (daccord.cpp:1653-1878) and
    the run must equal an oracle run with the oracle's own estimate."""
    import pyoracle
    from daccord_amd import cli, io as dio
    d, ovl, piles = small_data
    las, db, db2 = str(tmp_path / "reads.las"), str(tmp_path / "reads.db"), str(tmp_path / "breads.db")
    dio.write_db(db, d.bps, d.boff, d.rlen); dio.write_db(db2, d.bps, d.boff, d.rlen)
    dio.write_las(las, 100, d.ovl, d.trace)
    r = cli.run(["-k8", "-I0,5", las, db, db2])
    assert r.returncode == 0, r.stderr.decode()
    lo_ovl, lo_piles = pyoracle.select_lowest(d.ovl, d.piles)
    O = pyoracle.Oracle(default_params(k=8)); O.load_db(d.bps, d.boff, d.rlen)
    c, us, un, prof = O.estimate_profile(lo_piles[lo_piles["aread"] <= 5], lo_ovl, d.trace, two_databases=True)
    assert [float(x) for x in open(las + ".eprof").read().split()] == list(prof)
    O.set_error_profile(*prof)
    fo, bo = O.run(piles[piles["aread"] <= 5], ovl, d.trace, nthreads=4)
    assert r.stdout.decode() == pyoracle.fasta(fo, bo)


def _random_configs(first, last, seed=20260921, gen=None):
    import random
    from common import random_run_config
    if gen is not None:
        random_run_config = gen
    rng = random.Random(seed)
    tiers = np.zeros(3, dtype=np.int64)
    nwin = 0
    for i in range(last):
        kw, data, maxin, npl = random_run_config(rng)
        if i < first:
            continue
        d = SynthData(data["genome_len"], data["nreads"], data["read_len"],
                      **{k: v for k, v in data.items() if k not in ("genome_len", "nreads", "read_len")})
        ovl, piles = pyoracle.pile_select(d.ovl, d.piles, maxinput=maxin)
        sel = piles[:min(len(piles), npl)]
        O, E = _pair(d, **kw)
        fo, bo = O.run(sel, ovl, d.trace, trace_bytes=d.trace_bytes, nthreads=8, want_windows=True)
        fx, bx = E(sel, ovl, d.trace, trace_bytes=d.trace_bytes)
        t = E.timing()
        tiers += np.array(list(t.tier_out), dtype=np.int64); nwin += int(t.nwindows)
        print("fuzz seed %d config %d: %s %s windows %d handed on %s" % (seed, i, kw, data, t.nwindows, list(t.tier_out)))
        assert windows_equal(O.windows(), E.debug_windows()) == [], (seed, i, kw, data)
        assert frags_equal(fo, bo, fx, bx), (seed, i, kw, data)
        # second pass over the resident batch: the hand-over buffer exists from the second use of a context on (round 4), so this
        # is the pass in which the tiers load the sorted instances of the windows handed to them
        E.rerun(); f2, b2 = E.collect()
        assert frags_equal(fo, bo, f2, b2), ("second pass", seed, i, kw, data)
        E.close()
    print("fuzz total: %d windows, handed on per tier %s" % (nwin, tiers.tolist()))


def test_random_parameter_sets():
    """Random run parameters / error profiles / trace spacings / depths (tests/common.py:random_run_config) through the
    C ABI on the GPU: windows and fragments must equal the oracle's bit for bit.  Seeds and configurations are printed."""
    _random_configs(0, 10)


def test_random_parameter_sets_fifty():
    """The randomized parity runs on the real 64-lane build (the CPU fuzzing covers the same generator on the host
    emulation): 50 more parameter sets from a second seed."""
    _random_configs(0, 50, seed=777)
