"""Inputs and expected outputs of the native GPU probe (scripts/gpu_probe.sh): a run of the C++ front end daccord_hip, no
Python on the GPU box, for when only seconds of GPU time are left.  Five cases on .las / .db files written here:

  long   w = 56 on insertion-rich reads (B window strings of more than 64 bases: k_window_long / tier 5)
  deep   50x piles at k = 14 (deep tier with the register sort first)
  base   the default data of the GPU tests at k = 8
  warp   trace blocks of 100 A bases against more than 200 B bases, w = 63 (window strings of 129..256 bases)
  warp2  two byte trace values, blocks of 126 A bases against 700 B bases (window strings of more than 256 bases)

The oracle (test infrastructure) produces the expected FASTA here; the probe only compares md5 sums.
Writes probe_in/ (not tracked: regenerate with this script)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import pyoracle
from daccord_amd import io as dio
from daccord_amd._structs import default_params
from daccord_amd.synth import SynthData

OUT = os.path.join(ROOT, "probe_in")
os.makedirs(OUT, exist_ok=True)

CASES = [
    ("long", dict(genome_len=100000, nreads=200, read_len=5000, erate=0.25, seed=77, ins_frac=0.9, del_frac=0.05, sub_frac=0.05),
     dict(k=10, w=56, a=14), ["-k10", "-w56", "-a14"], (0, 2)),
    ("deep", dict(genome_len=30000, nreads=300, read_len=5000, seed=7), dict(k=14), ["-k14"], (10, 12)),
    # (round 6) a wide window through the command line: the wide LDS tiers k_window_fast<8|9>
    ("wide", dict(genome_len=100000, nreads=200, read_len=5000, seed=11), dict(k=12, w=80, a=20), ["-k12", "-w80", "-a20"], (0, 3)),
    ("base", dict(genome_len=100000, nreads=200, read_len=5000, seed=1), dict(k=8), ["-k8"], (0, 5)),
    # badly aligned trace blocks: B window strings of 129..256 bases (generic engine, LSTR = 256)
    ("warp", dict(genome_len=100000, nreads=200, read_len=5000, seed=1), dict(k=8, w=63, a=16), ["-k8", "-w63", "-a16"], (0, 1)),
    # the same with two byte trace values and blocks of 126 A bases against 700 B bases: strings of more than 256 bases
    ("warp2", dict(genome_len=100000, nreads=200, read_len=5000, seed=1, tspace=126), dict(k=8, w=63, a=16, tspace=126), ["-k8", "-w63", "-a16"], (0, 1)),
]

lines = []
for name, dk, pk, args, (lo, hi) in CASES:
    d = SynthData(dk.pop("genome_len"), dk.pop("nreads"), dk.pop("read_len"), **dk)
    if name.startswith("warp"):
        from common import warp_trace
        ids = [i for i, pl in enumerate(d.piles) if lo <= pl["aread"] <= hi]
        wkw = dict(every=4, extra=580, cap=2000) if name == "warp2" else {}
        d.trace = warp_trace(d.ovl, d.piles, d.trace, ids, **wkw)      # d.piles index d.ovl (records in .las order)
    ovl, piles = pyoracle.pile_select(d.ovl, d.piles)
    las, db = os.path.join(OUT, name + ".las"), os.path.join(OUT, name + ".db")
    dio.write_db(db, d.bps, d.boff, d.rlen)
    dio.write_las(las, pk.get("tspace", 100), d.ovl, d.trace)
    p_i, p_d, cor = d.error_profile()
    O = pyoracle.Oracle(default_params(**pk)); O.set_error_profile(p_i, p_d, cor); O.load_db(d.bps, d.boff, d.rlen)
    sel = piles[(piles["aread"] >= lo) & (piles["aread"] <= hi)]
    fo, bo = O.run(sel, ovl, d.trace, trace_bytes=d.trace.dtype.itemsize, nthreads=16)
    fa = pyoracle.fasta(fo, bo).encode()
    with open(os.path.join(OUT, name + ".expected.md5"), "w") as f:
        f.write(hashlib.md5(fa).hexdigest() + "\n")
    lines.append("%s %s --eprof%r,%r,%r" % (name, " ".join(args + ["-I%d,%d" % (lo, hi)]), p_i, p_d, cor))
    print(name, len(fa), "bytes of FASTA,", len(sel), "piles", file=sys.stderr)
with open(os.path.join(OUT, "cases.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
