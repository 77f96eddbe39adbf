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
