"""Regenerates tests/golden/oracle_small.json from the oracle (deterministic synthetic input, seed 1).
The reference has no golden vectors for this path and cannot be built here
(libmaus2 missing), so these freeze the oracle's own output to detect drift."""
import hashlib
import json
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import pyoracle  # noqa: E402
from daccord_amd._structs import default_params  # noqa: E402
from daccord_amd.synth import SynthData  # noqa: E402

d = SynthData(100000, 200, 5000, seed=1)
ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
k, npiles = 8, 4
O = pyoracle.Oracle(default_params(k=k))
O.set_error_profile(*d.error_profile())
O.load_db(d.bps, d.boff, d.rlen)
fr, ba = O.run(piles[:npiles], ovl, d.trace, nthreads=4, want_windows=True)
txt = pyoracle.fasta(fr, ba)
w = O.windows()
G = {"k": k, "npiles": npiles, "fasta_sha256": hashlib.sha256(txt.encode()).hexdigest(),
     "first_header": txt.splitlines()[0],
     "first_windows": [bytes(x["cons"]).rstrip(b"\0").decode() for x in w[:8]]}
with open(os.path.join(HERE, "oracle_small.json"), "w") as f:
    json.dump(G, f, indent=1)
with open(os.path.join(HERE, "oracle_small.fasta"), "w") as f:
    f.write(txt)
print(G)
