"""Truth-based accuracy report of corrected fragments, the measurement of the reference package's `checkconsensus`
(src/checkconsensus.cpp:730-1075; line formats README.md:406-472) for reads whose ground truth is known (synthetic data:
genome + (start, end, strand) per read).  Per fragment: banded alignment to the truth (csrc/host_check.cpp); per read a
coverage line; a global line; EP / EM lines (error rate quantiles over reads)."""
import ctypes as C
import numpy as np
from . import io as _io

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _lib():
    L = _io.lib()
    if not getattr(L, "_check_ready", False):
        L.dacc_check_fragment.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L._check_ready = True
    return L


def _stats_str(s):
    tot = int(s[0] + s[1] + s[2] + s[3]); ed = int(s[1] + s[2] + s[3])
    return "AlignmentStatistics(matches=%d,mismatches=%d,insertions=%d,deletions=%d,editdistance=%d,erate=%.10f)" % (
        s[0], s[1], s[2], s[3], ed, (ed / tot) if tot else 0.0)


def check(frags, bases, genome, truth, rlen, reads=None, slack=400):
    """Returns (lines, summary).  genome: uint8 0..3, truth[r] = (start, end, strand), rlen[r] = raw read length;
    reads: ids to report (default: every read that has a fragment)."""
    L = _lib()
    by = {}
    for f in frags:
        by.setdefault(int(f["aread"]), []).append(f)
    ids = sorted(by) if reads is None else [int(r) for r in reads]
    lines = []; acc = np.zeros(4, np.int64); refsum = covsum = 0; rates = []
    for r in ids:
        gs, ge, st = (int(x) for x in truth[r])
        g = bytes(b"ACGT"[x] for x in genome[gs:ge])
        if st:
            g = g.translate(_COMP)[::-1]
        G, rl = len(g), int(rlen[r]); cov = 0; racc = np.zeros(4, np.int64)
        for f in by.get(r, []):
            s = bytes(bases[int(f["seq_off"]):int(f["seq_off"]) + int(f["len"])])      # bases: any bytes-like
            e0 = int(round(int(f["first"]) * G / rl)); e1 = int(round((int(f["last"]) + 1) * G / rl))
            w0 = max(0, e0 - slack); w1 = min(G, e1 + slack)
            st4 = np.zeros(4, np.uint64); a = C.c_uint64(); b = C.c_uint64()
            rc = L.dacc_check_fragment(s, len(s), g[w0:w1], w1 - w0, e0 - w0, min(w1 - w0, e1 - w0), slack + 100, st4.ctypes.data_as(C.c_void_p), C.byref(a), C.byref(b))
            if rc:
                raise RuntimeError("fragment of read %d does not align to its truth inside the band (rc=%d)" % (r, rc))
            racc += st4.astype(np.int64); acc += st4.astype(np.int64); cov += b.value - a.value
            lines.append("%d\t%s\t%s\t[%d,%d]\t0:%d,%d" % (r, _stats_str(st4), _stats_str(acc), f["first"], f["last"], gs + (w0 + a.value if not st else G - (w0 + b.value)),
                                                            gs + (w0 + b.value if not st else G - (w0 + a.value))))
        lines.append("[C]\t%d\t%d\t%d\t%.6g" % (r, G, cov, cov / G if G else 0.0))
        refsum += G; covsum += cov
        tot = int(racc.sum())
        if tot:
            rates.append(float(racc[1] + racc[2] + racc[3]) / tot)
    lines.append("[G]\t%d\t%d\t%.6g\t%s" % (refsum, covsum, covsum / refsum if refsum else 0.0, _stats_str(acc)))
    rs = sorted(rates)
    for i, e in enumerate(rs):
        lines.append("[EP]\t%.10f\t%d\t%.6g" % (e, i + 1, (i + 1) / len(rs)))
    for i, e in enumerate(rs):
        lines.append("[EM]\t%.10f\t%d\t%.6g" % (e, len(rs) - i, (len(rs) - i) / len(rs)))
    tot = int(acc.sum())
    summary = {"reads": len(ids), "reference_bases": int(refsum), "covered_bases": int(covsum), "covered_frac": round(covsum / refsum, 6) if refsum else 0.0,
               "matches": int(acc[0]), "mismatches": int(acc[1]), "insertions": int(acc[2]), "deletions": int(acc[3]),
               "erate": round(float(acc[1] + acc[2] + acc[3]) / tot, 8) if tot else None,
               "median_read_erate": round(rs[len(rs) // 2], 8) if rs else None}
    return lines, summary
