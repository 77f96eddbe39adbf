"""CPU tests of the oracle itself (no GPU): hand-derived cases for the primitives the reference takes from
libmaus2, window schedule, and end-to-end properties that do not depend on unpinned tie-breaks."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
import pyoracle
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData
from common import truth_error

HERE = os.path.dirname(os.path.abspath(__file__))


def _align(a, b):
    L = pyoracle.lib()
    tr = (C.c_uint8 * (len(a) + len(b) + 1))()
    n = C.c_uint64()
    d = L.oracle_align(a, len(a), b, len(b), tr, C.byref(n))
    return d, "".join("=XID"[x] for x in tr[:n.value])


def test_align_known_cases():
    assert _align(b"ACGT", b"ACGT") == (0, "====")
    assert _align(b"ACGT", b"AGT") == (1, "=D==")          # deletion of C (consumes a only)
    assert _align(b"AGT", b"ACGT") == (1, "=I==")          # insertion of C (consumes b only)
    assert _align(b"ACGT", b"ATGT") == (1, "=X==")
    assert _align(b"", b"ACG") == (3, "III")
    assert _align(b"ACG", b"") == (3, "DDD")
    # co-optimal tracebacks: diagonal preferred, then DEL, then INS (walking back from the end)
    assert _align(b"AA", b"A") == (1, "D=")
    assert _align(b"A", b"AA") == (1, "I=")


def test_edit_distance_matches_align():
    rng = np.random.default_rng(5)
    L = pyoracle.lib()
    for _ in range(200):
        a = bytes(rng.choice(list(b"ACGT"), rng.integers(0, 60)).tolist())
        b = bytes(rng.choice(list(b"ACGT"), rng.integers(0, 60)).tolist())
        d, tr = _align(a, b)
        assert d == L.oracle_edit_distance(a, len(a), b, len(b))
        assert d == sum(c != "=" for c in tr)
        assert sum(c in "=XD" for c in tr) == len(a) and sum(c in "=XI" for c in tr) == len(b)


def test_windows_count():
    # Windows::computeN (HandleContext.hpp:390-408)
    L = pyoracle.lib()
    assert L.oracle_windows_count(39, 10, 40) == 0
    assert L.oracle_windows_count(40, 10, 40) == 1
    assert L.oracle_windows_count(41, 10, 40) == 2      # [0,40) and the snapped [1,41)
    assert L.oracle_windows_count(50, 10, 40) == 2
    assert L.oracle_windows_count(10000, 10, 40) == 997


def test_offset_likely_tables_sane():
    O = pyoracle.Oracle(default_params())
    O.set_error_profile(0.12, 0.02, 0.85)
    t = O.tables()
    assert t[0] == 41                                     # rows = w+1 (computeOffsetLikely(windowsize,..))
    assert t[1] > 41                                      # insertions dominate: read offsets run ahead


def test_perfect_piles_return_genome():
    """Error-free reads: the consensus of every pile must be the read itself (independent of tie-breaks)."""
    d = SynthData(60000, 150, 3000, erate=0.0, seed=4)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    O = pyoracle.Oracle(default_params(k=8))
    O.set_error_profile(0.12, 0.02, 0.85)
    O.load_db(d.bps, d.boff, d.rlen)
    fr, ba = O.run(piles[:6], ovl, d.trace, nthreads=4)
    assert len(fr) >= 6
    for f in fr:
        rid = f["aread"]
        read = bytes(b"ACGT"[(d.bps[d.boff[rid] + (i >> 2)] >> (6 - 2 * (i & 3))) & 3] for i in range(d.rlen[rid]))
        seq = ba[f["seq_off"]:f["seq_off"] + f["len"]]
        assert seq == read[f["first"]:f["last"] + 1]


def test_accuracy_vs_truth(small_data):
    d, ovl, piles = small_data
    O = pyoracle.Oracle(default_params(k=8))
    O.set_error_profile(*d.error_profile())
    O.load_db(d.bps, d.boff, d.rlen)
    fr, ba = O.run(piles[20:28], ovl, d.trace, nthreads=4)
    L = pyoracle.lib()
    ed, n = truth_error(d, fr, ba, lambda a, b: L.oracle_edit_distance(a, len(a), b, len(b)))
    assert n > 0 and ed / n < 0.02                        # 15 % raw error -> < 2 % after correction


def test_golden_fixture(small_data):
    """Golden vectors frozen from the oracle (the reference ships none)."""
    with open(os.path.join(HERE, "golden", "oracle_small.json")) as f:
        G = json.load(f)
    d, ovl, piles = small_data
    O = pyoracle.Oracle(default_params(k=G["k"]))
    O.set_error_profile(*d.error_profile())
    O.load_db(d.bps, d.boff, d.rlen)
    fr, ba = O.run(piles[:G["npiles"]], ovl, d.trace, nthreads=4, want_windows=True)
    txt = pyoracle.fasta(fr, ba)
    assert hashlib.sha256(txt.encode()).hexdigest() == G["fasta_sha256"]
    w = O.windows()
    assert [bytes(x["cons"]).rstrip(b"\0").decode() for x in w[:len(G["first_windows"])]] == G["first_windows"]
    assert txt.splitlines()[0] == G["first_header"]
