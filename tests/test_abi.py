"""The drop-in boundary without a GPU: the library loads, exports every symbol include/daccord_hip.h
declares, refuses to run without a HIP device (no CPU fallback), and its host-side pile selection equals
the oracle's restatement of daccord.cpp:2120-2288."""
import ctypes as C
import os
import re
import numpy as np
import pytest
import pyoracle
from daccord_amd import engine
from daccord_amd._structs import default_params, DaccOverlap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "daccord_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dacc_[a-z_]+)\s*\(", hdr)))
    L = engine.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert set(engine.EXPORTS) <= set(declared)


def test_io_exports_match_header():
    """include/daccord_io.h: every entry point is exported by the host-only library and by the HIP library."""
    from daccord_amd import io as dio
    hdr = open(os.path.join(ROOT, "include", "daccord_io.h")).read()
    declared = sorted(set(re.findall(r"\b(dacc_(?:db|las)_[a-z_]+)\s*\(", hdr)))
    assert len(declared) >= 10
    for L in (dio.lib(), engine.lib()):
        for name in declared:
            assert hasattr(L, name), name


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.DaccError) as e:
        engine.Engine(default_params())
    assert e.value.code == -2                              # DACC_ENODEV


def test_bad_params_rejected():
    for kw in (dict(k=2), dict(k=17), dict(klow=9, khigh=8), dict(w=0), dict(w=129)):
        h = C.c_void_p()
        rc = engine.lib().dacc_create(C.byref(h), C.byref(default_params(**kw)))
        assert rc == -1


def test_pile_select_matches_oracle(small_data):
    d, _, _ = small_data
    for maxinput in (5000, 7, 1):
        a, pa = engine.pile_select(d.ovl, d.piles[:20], maxinput=maxinput)
        b, pb = pyoracle.pile_select(d.ovl, d.piles[:20], maxinput=maxinput)
        assert a.tobytes() == b.tobytes() and pa.tobytes() == pb.tobytes()
    # a pile larger than one 64 KiB input block (block-wise copy order, daccord.cpp:2199-2262)
    rng = np.random.default_rng(3)
    big = np.zeros(3000, dtype=np.dtype(DaccOverlap))
    big["abpos"] = rng.integers(0, 50, 3000); big["aepos"] = big["abpos"] + 1000 + rng.integers(0, 100, 3000)
    big["diffs"] = rng.integers(0, 300, 3000); big["tlen"] = 22; big["bread"] = np.arange(3000)
    from daccord_amd._structs import DaccPile
    piles = np.zeros(1, dtype=np.dtype(DaccPile)); piles[0]["novl"] = 3000
    for maxinput in (5000, 100):
        a, pa = engine.pile_select(big, piles, maxinput=maxinput)
        b, pb = pyoracle.pile_select(big, piles, maxinput=maxinput)
        assert a.tobytes() == b.tobytes()
        assert (np.diff(a["abpos"]) >= 0).all()
