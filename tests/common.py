import collections
import numpy as np


def windows_equal(wo, wx):
    """Compare per-window results (oracle vs device/emulation).  Returns list of differing indices."""
    bad = []
    if len(wo) != len(wx):
        return list(range(max(len(wo), len(wx))))
    for i, (a, b) in enumerate(zip(wo, wx)):
        same = (a["status"] == b["status"] and a["mao"] == b["mao"] and a["elength"] == b["elength"] and
                (a["status"] != 1 or (bytes(a["cons"]) == bytes(b["cons"]) and a["minrate"] == b["minrate"] and
                                      a["filterfreq"] == b["filterfreq"] and a["k"] == b["k"])))
        if not same:
            bad.append(i)
    return bad


def frags_equal(fo, bo, fx, bx):
    if len(fo) != len(fx) or bo != bx:
        return False
    for x, y in zip(fo, fx):
        if (x["aread"], x["first"], x["last"], x["len"], x["seq_off"]) != (y["aread"], y["first"], y["last"], y["len"], y["seq_off"]):
            return False
    return True


def truth_error(d, frags, bases, edit_distance):
    """checkconsensus-style accuracy (README.md:406-472): edit distance of each corrected fragment to the
    genome interval of its source read (whole-read fragments only)."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    tot_ed = tot_len = 0
    for f in frags:
        if f["first"] != 0:
            continue
        gs, ge, st = d.truth[f["aread"]]
        g = bytes(b"ACGT"[x] for x in d.genome[gs:ge])
        if st:
            g = g.translate(comp)[::-1]
        s = bases[f["seq_off"]:f["seq_off"] + f["len"]]
        tot_ed += edit_distance(s, g)
        tot_len += len(g)
    return tot_ed, tot_len


def random_run_config(rng):
    """One random parameter set of the parity fuzzing (scripts/fuzz_emul_vs_oracle.py, test_gpu_parity.py):
    returns (params kw, synthetic data kw, maxinput, number of piles)."""
    w = rng.choice([24, 32, 40, 40, 40, 48, 56, 63]); a = min(rng.choice([5, 8, 10, 10, 16, 20]), w)
    klow = rng.choice([6, 7, 8, 8, 9, 10, 12, 14, 14, 16]); khigh = min(klow + rng.choice([0, 0, 0, 1, 2]), 16)
    if klow >= w - 4:
        klow = khigh = 8
    erate = rng.choice([0.02, 0.08, 0.12, 0.15, 0.15, 0.2, 0.28])
    mix = rng.choice([(0.8, 0.1333, 0.0667), (1 / 3, 1 / 3, 1 / 3), (0.5, 0.4, 0.1), (0.2, 0.7, 0.1)])
    nreads = rng.choice([60, 120, 200, 300]); rlen = rng.choice([1500, 3000, 5000]); glen = rng.choice([30000, 60000, 100000])
    kw = dict(w=w, a=a, klow=klow, khigh=khigh)
    if rng.random() < 0.3: kw["maxalign"] = rng.choice([3, 5, 8, 15])
    if rng.random() < 0.2: kw["minwindowcov"] = rng.choice([2, 4, 5])
    if rng.random() < 0.2: kw["producefull"] = 1
    if rng.random() < 0.2: kw["minlen"] = rng.choice([200, 1000])
    if rng.random() < 0.2: kw["maxfilterfreq"] = rng.choice([1, 3])
    if rng.random() < 0.15: kw["minfilterfreq"] = 1
    if rng.random() < 0.1: kw["eminrate"] = rng.choice([5, 15, 40])
    seed = rng.randrange(1, 10 ** 6)
    tspace = rng.choice([100, 100, 100, 50, 64, 125]); minovl = rng.choice([1000, 1000, 500, 200])
    kw["tspace"] = tspace
    data = dict(genome_len=glen, nreads=nreads, read_len=rlen, erate=erate, seed=seed, ins_frac=mix[0], del_frac=mix[1], sub_frac=mix[2],
                tspace=tspace, min_overlap=minovl)
    return kw, data, rng.choice([5000, 5000, 10]), rng.choice([2, 3, 4])


def random_run_config_wide(rng):
    """random_run_config plus what the second half of round 2 added: deep piles (coverage up to 60x: deep tier, tier 2/3
    with 2048 / 4096 k-mer instances, many first / last k-mer candidates) and trace spacings beyond 125 / 128 (two byte
    trace values, k_trace_wide).  Used by the CPU fuzzing (scripts/fuzz_emul_vs_oracle.py --wide)."""
    kw, data, maxin, npl = random_run_config(rng)
    r = rng.random()
    if r < 0.4:
        # deep: short genome, many reads; larger k and the default window (the oracle enumerates every (first, last)
        # pair from scratch: dense graphs of deep piles at small k take it minutes per pile)
        data["read_len"] = 2000; data["nreads"] = rng.choice([300, 400, 600])
        data["genome_len"] = int(data["nreads"] * data["read_len"] / rng.choice([35, 45, 60]))
        data["erate"] = rng.choice([0.12, 0.15, 0.15])
        kw["w"] = 40; kw["a"] = rng.choice([10, 20]); kw["klow"] = rng.choice([12, 14, 14]); kw["khigh"] = kw["klow"]
        kw.pop("minfilterfreq", None); kw.pop("maxfilterfreq", None)
        npl = 1
    if rng.random() < 0.4:
        ts = rng.choice([126, 128, 150, 200, 256, 300])
        kw["tspace"] = ts; data["tspace"] = ts
    if rng.random() < 0.25:
        # an error profile that is not the data's (an estimate from other reads): narrow or wide model tables
        data["profile"] = rng.choice([(0.01, 0.002, 0.98), (0.03, 0.01, 0.95), (0.2, 0.05, 0.7), (0.05, 0.05, 0.85)])
    if rng.random() < 0.15:
        # badly aligned trace blocks (warp_trace): B window strings of up to three times the window size
        data["warp"] = (rng.choice([2, 3, 5]), rng.choice([60, 115, 150]))
        if data["tspace"] > 125 and rng.random() < 0.5:
            data["warp"] = (rng.choice([3, 5]), rng.choice([300, 580, 900]), 2000)   # two byte trace values: strings beyond 256 bases (blocks within what the trace kernels hold: 4096 / 2048 B bases, a longer one drops its pile)
    return kw, data, maxin, npl


def random_run_config_w128(rng):
    """random_run_config with a window of 64 ... 128 bases (generic engine only; round 4): k from 8 up (dense graphs at small k make
    the oracle slow at these sizes), advances up to the window size, sometimes a trace spacing beyond 125 (two byte trace values)."""
    kw, data, maxin, npl = random_run_config(rng)
    w = rng.choice([64, 65, 72, 80, 96, 100, 112, 127, 128]); kw["w"] = w; kw["a"] = rng.choice([10, 16, 25, 32, 40, 64, w])
    kw["klow"] = rng.choice([8, 9, 10, 12, 14, 16]); kw["khigh"] = min(16, kw["klow"] + rng.choice([0, 0, 0, 1]))
    kw.pop("minfilterfreq", None)
    data["nreads"] = rng.choice([60, 120, 150]); data["read_len"] = rng.choice([1500, 3000]); data["genome_len"] = rng.choice([30000, 60000])
    if rng.random() < 0.3:
        ts = rng.choice([126, 128, 150, 200]); kw["tspace"] = ts; data["tspace"] = ts
    return kw, data, maxin, min(npl, 3)


def warp_trace(ovl, piles, trace, pile_ids, every=3, extra=115, cap=250):
    """Synthetic bad alignments: for every `every`-th overlap of the given piles, one interior trace block gets `extra`
    more B bases (taken from the overlap's other blocks, so that the B lengths still sum to bepos-bbpos).  The windows
    inside such a block have B strings of two to three times the window size.  Returns the new trace array."""
    tr = trace.copy()
    for pi in pile_ids:
        f, n = int(piles[pi]["first_ovl"]), int(piles[pi]["novl"])
        for z in range(f, f + n, every):
            o = ovl[z]; nb = int(o["tlen"]) // 2; t0 = int(o["trace_off"])
            if nb < 8:
                continue
            bl = [int(tr[t0 + 2 * i + 1]) for i in range(nb)]
            tgt = nb // 2
            take = min(extra, cap - bl[tgt]); got = 0      # cap: one byte trace values
            for i in list(range(1, tgt)) + list(range(tgt + 1, nb - 1)):
                g = min(bl[i] - 40, take - got)
                if g > 0:
                    bl[i] -= g; got += g
                if got == take:
                    break
            bl[tgt] += got
            for i in range(nb):
                tr[t0 + 2 * i + 1] = bl[i]
    return tr
