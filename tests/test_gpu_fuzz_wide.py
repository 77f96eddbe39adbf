"""Wide random parameter sets on the GPU: deep 35-60x piles, trace spacings 126..300 (two byte trace
values, k_trace_wide), error profiles that are not the data's, warped traces (window strings of up to several hundred
bases: tier 5 and the generic engine's long strings) -- everything the narrow generator of test_gpu_parity.py leaves out.
The oracle's digests were computed in the build container (tests/golden/make_golden_fuzz.py); here the same inputs are
regenerated and run through the C ABI on the MI355X."""
import hashlib
import json
import os
import numpy as np
import pytest
from daccord_amd import engine
from daccord_amd._structs import default_params
from fuzz_wide_cases import wide_cases, make_wide_case
from scale_cases import window_digest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    fn = os.path.join(HERE, "golden", "fuzz_wide.json")
    if not os.path.exists(fn):
        pytest.skip("no golden fixture")
    with open(fn) as f:
        return json.load(f)


@pytest.mark.parametrize("part", [0, 1, 2, 3])
def test_wide_random_parameter_sets_match_oracle_digests(part):
    G = _golden()
    cases = wide_cases(G["seed"], G["nsets"])
    tiers = np.zeros(3, dtype=np.int64); nwin = 0; ndropped = 0
    sets = [e for e in G["sets"] if e["i"] % 4 == part]
    assert sets
    for e in sets:
        kw, data, maxin, npl = cases[e["i"]]
        assert json.loads(json.dumps([kw, data, maxin])) == [e["params"], e["data"], e["maxinput"]], "the generator drifted from the fixture"
        d, prof, ovl, piles, sel, trace = make_wide_case(data, maxin, npl, engine.pile_select)
        p = default_params(**kw)
        E = engine.Engine(p); E.set_error_profile(*prof); E.load_db(d.bps, d.boff, d.rlen)
        fx, bx = E(sel, ovl, trace, trace_bytes=d.trace_bytes)
        t = E.timing(); w = E.debug_windows()
        st, msgs = E.pile_status()
        ndropped += int((st != 0).sum())
        tiers += np.array(list(t.tier_out), dtype=np.int64); nwin += int(t.nwindows)
        print("wide fuzz seed %d set %d: %s %s windows %d handed on %s dropped piles %d" % (G["seed"], e["i"], kw, data, t.nwindows, list(t.tier_out), int((st != 0).sum())))
        assert (st == 0).all(), (e["i"], msgs)
        assert len(w) == e["nwindows"], (e["i"], kw, data)
        assert window_digest(w) == e["windows_sha256"], (e["i"], kw, data)
        assert hashlib.sha256(engine.fasta(fx, bx).encode()).hexdigest() == e["fasta_sha256"], (e["i"], kw, data)
        assert len(bx) == e["nbases"] and len(fx) == e["nfragments"]
        # second pass over the resident batch (hand-over buffer active from the second use of a context on): same digests
        E.rerun(); f2, b2 = E.collect(); w2 = E.debug_windows()
        assert window_digest(w2) == e["windows_sha256"] and hashlib.sha256(engine.fasta(f2, b2).encode()).hexdigest() == e["fasta_sha256"], ("second pass", e["i"], kw, data)
        E.close()
    print("wide fuzz part %d: %d sets, %d windows, handed on per tier %s" % (part, len(sets), nwin, tiers.tolist()))
