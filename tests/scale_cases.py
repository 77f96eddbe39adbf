"""BASELINE-scale parity cases shared by the golden generator (tests/golden/make_golden_scale.py, runs
the oracle in the build container) and the GPU test / bench (which only hash what the HIP path returns)."""
import hashlib
import numpy as np

THIRD = 1.0 / 3.0

CASES = {
    # config 2 of BASELINE.json = bench.py's default data set (seed 3); the first 1000 of its 10 000 piles
    "cfg2": dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=0, npiles=1000,
                 params=[dict(k=14)]),
    # config 2 again, stratified over the whole 10 000-pile batch: the first 62 piles, 125 piles around each
    # of the seven interior boundaries of the eight per-XCD window queues ((n*q)>>3, capi.hip next_window) and the last 125
    "cfg2s": dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=0, npiles=10000,
                  pile_ranges=[[0, 62]] + [[1250 * q - 62, 1250 * q + 63] for q in range(1, 8)] + [[9875, 10000]],
                  params=[dict(k=14)]),
    # config 2, a second stratified sample (round 4): 125 piles from the MIDDLE of each of the eight per-XCD queue ranges
    # (cfg2s sits on their boundaries, cfg2 on the first 1000) -- another 1000 piles of the headline batch against the oracle
    "cfg2t": dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=0, npiles=10000,
                  pile_ranges=[[1250 * q + 562, 1250 * q + 687] for q in range(8)],
                  params=[dict(k=14)]),
    # config 2, a third stratified sample (round 4, last session): 125 piles from the first QUARTER of each of the eight per-XCD queue
    # ranges (offset 250) -- with cfg2, cfg2s, cfg2t that is 4062 of the headline batch's 10 000 piles against the oracle
    "cfg2u": dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=0, npiles=10000,
                  pile_ranges=[[1250 * q + 250, 1250 * q + 375] for q in range(8)],
                  params=[dict(k=14)]),
    # config 2, a fourth, smaller sample: 25 piles at the three-quarter point of each of the eight per-XCD queue ranges (offset 937)
    "cfg2v": dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=0, npiles=10000,
                  pile_ranges=[[1250 * q + 937, 1250 * q + 962] for q in range(8)],
                  params=[dict(k=14)]),
    # config 3 stand-in (D. melanogaster 20x is 140 Mbase; the files are not in the container): a 100-pile slice of a 20x set
    # with a larger genome and longer reads than config 2
    "cfg3": dict(genome_len=7000000, nreads=10000, read_len=14000, seed=7, synth={}, first=4000, npiles=100,
                 params=[dict(k=14)]),
    # config 3 at its TRUE scale (round 4): a 140 Mbase genome at 20x = 280 000 reads of 10 kb (a 700 MB 2-bit read store on the
    # device, B reads from all over it); overlaps and piles only for 100 A reads in the middle (SynthData aread_range: the records
    # are those of the full set), all of them selected
    "cfg3b": dict(genome_len=140000000, nreads=280000, read_len=10000, seed=9, synth={}, first=0, npiles=100, aread_range=[140000, 140100],
                  params=[dict(k=14)]),
    # config 4 shape again, 200 piles of a 54x set (cfg4 above is a 50-pile slice)
    "cfg4b": dict(genome_len=222222, nreads=1200, read_len=10000, seed=14, synth={}, first=500, npiles=200,
                  params=[dict(k=14)]),
    # config 1 stand-in at the reference's default k (the E. coli files are not in the container)
    "cfg1k8": dict(genome_len=500000, nreads=1000, read_len=10000, seed=2, synth={}, first=0, npiles=100,
                   params=[dict(k=8)]),
    # config 5: ONT-like 1/3-1/3-1/3 error mix, 15 % total, k sweep (k = 17 is outside the reference's semantics)
    "cfg5": dict(genome_len=500000, nreads=1000, read_len=10000, seed=5, synth=dict(ins_frac=THIRD, del_frac=THIRD, sub_frac=THIRD),
                 first=0, npiles=100, params=[dict(k=k) for k in (10, 11, 12, 13, 14, 15, 16)]),
    # config 2 with WIDE windows (round 6: the wide LDS tiers, k_window_fast<8> / <9>): 48 piles from the middle of the headline batch at
    # -w 64, 80 and 96 (advance w / 4; -w is free in the reference, daccord.cpp:1282-1305)
    "wide_cfg2": dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=2000, npiles=48,
                     params=[dict(k=14, w=64, a=16), dict(k=14, w=80, a=20), dict(k=14, w=96, a=24)]),
    # config 4 shape: 54x depth slice
    "cfg4": dict(genome_len=111111, nreads=600, read_len=10000, seed=4, synth={}, first=0, npiles=50,
                 params=[dict(k=14)]),
}


def _complement_cases(nparts=12):
    """Round 5: the REST of the headline batch -- every pile of config 2 that cfg2, cfg2s, cfg2t, cfg2u, cfg2v leave out (6075 of
    10 000), in ascending order, cut into `nparts` cases cfg2w00 ... so that each is a bounded oracle job (about ten minutes on six
    cores) and a lost session loses one part, not the run.  With them every pile of the batch bench.py times has an oracle digest."""
    cov = set()
    for n in ("cfg2", "cfg2s", "cfg2t", "cfg2u", "cfg2v"):
        c = CASES[n]
        for a, b in (c.get("pile_ranges") or [[c["first"], c["first"] + c["npiles"]]]):
            cov.update(range(a, b))
    rest = [i for i in range(10000) if i not in cov]
    per = (len(rest) + nparts - 1) // nparts
    for q in range(nparts):
        part = rest[q * per:(q + 1) * per]
        ranges = []
        for i in part:
            if ranges and ranges[-1][1] == i:
                ranges[-1][1] = i + 1
            else:
                ranges.append([i, i + 1])
        CASES["cfg2w%02d" % q] = dict(genome_len=5000000, nreads=10000, read_len=10000, seed=3, synth={}, first=0, npiles=10000,
                                      pile_ranges=ranges, params=[dict(k=14)])


_complement_cases()
CFG2W = sorted(n for n in CASES if n.startswith("cfg2w") and n[5:].isdigit())


def make_case(case, pile_select):
    from daccord_amd.synth import SynthData
    extra = dict(aread_range=tuple(case["aread_range"])) if case.get("aread_range") else {}
    d = SynthData(case["genome_len"], case["nreads"], case["read_len"], seed=case["seed"], **extra, **case["synth"])
    ovl, piles = pile_select(d.ovl, d.piles)
    if "pile_ranges" in case:
        sel = np.concatenate([piles[a:b] for a, b in case["pile_ranges"]])
    else:
        sel = piles[case["first"]:case["first"] + case["npiles"]]
    return d, ovl, piles, sel


def window_digest(w):
    h = hashlib.sha256()
    for x in w:
        ok = x["status"] == 1
        h.update(np.array([x["pile"], x["y"], x["status"], x["mao"], x["elength"], x["k"] if ok else 0,
                           x["filterfreq"] if ok else 0], dtype=np.int64).tobytes())
        h.update(np.uint64(x["minrate"] if ok else 0).tobytes())
        h.update(bytes(x["cons"]).rstrip(b"\0") if ok else b"")
        h.update(b"|")
    return h.hexdigest()


def pile_digests(frags, bases, sel, fasta):
    out = []
    ar = frags["aread"] if len(frags) else np.zeros(0, np.int32)
    for p in sel:
        f = frags[ar == p["aread"]]
        out.append(hashlib.sha256(fasta(f, bases).encode()).hexdigest()[:12])
    return out
