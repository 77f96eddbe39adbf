"""daccord-compatible command line front end of the MI355X path:

    python -m daccord_amd.cli [options] reads.las reads.db [reads2.db]

Options as src/daccord.cpp:185-207 / :1282-1305 (flag and value joined, e.g. -w40 -k8; options precede the
positionals): -w -a -d -f -V -I<lo,hi> -J<g,G> -E<eprof> -m -e -l -D -k<k | lo,hi> --minfilterfreq<n>
--maxfilterfreq<n>; -t and -T are accepted and ignored (no host threads, no temporary files).  Output: FASTA on stdout
as HandleContext.hpp:2710-2724 writes it, records in ascending A-read order, the middle name field numbered
sequentially (the reference's -t1 numbering).

Deviation: the reference estimates the error profile itself when <las>.eprof is missing (daccord.cpp:1653-1861, SURVEY.md
8f row 1, not built yet).  Here the profile must be supplied: --eprof<p_i,p_d,est_cor>, or -E<file> / <las>.eprof
containing the three numbers as text.  (The binary .eprof of the reference is a libmaus2 serialisation that is not in
the reference tree.)
"""
import sys
import numpy as np


def parse_args(argv):
    opt = {"w": 40, "a": 10, "d": None, "f": False, "V": 0, "I": None, "J": None, "E": None, "m": 3, "e": None, "l": 0,
           "D": 5000, "k": "8", "minfilterfreq": 0, "maxfilterfreq": 2, "eprof": None}
    pos = []
    for a in argv:
        if pos or not a.startswith("-") or a == "-":
            pos.append(a); continue
        if a.startswith("--"):
            if a in ("--vard", "--eprofonly", "--deepprofileonly", "--keepeprof"):
                raise SystemExit("daccord_amd: %s belongs to the error profile estimation, which is not built (SURVEY.md 8f)" % a)
            for name in ("minfilterfreq", "maxfilterfreq", "eprof"):
                if a.startswith("--" + name):
                    v = a[2 + len(name):].lstrip("=")
                    opt[name] = v if name == "eprof" else int(v)
                    break
            else:
                raise SystemExit("daccord_amd: unknown option " + a)
            continue
        key, val = a[1], a[2:]
        if key == "f":
            opt["f"] = True if val == "" else bool(int(val))
        elif key in ("t", "T"):
            pass
        elif key in ("w", "a", "m", "l", "D", "V"):
            opt[key] = int(val) if val != "" else 1
        elif key in ("d", "e"):
            opt[key] = int(val)
        elif key in ("I", "J", "E", "k"):
            opt[key] = val
        else:
            raise SystemExit("daccord_amd: unknown option " + a)
    if len(pos) < 2:
        raise SystemExit("usage: python -m daccord_amd.cli [options] reads.las reads.db [reads2.db]")
    return opt, pos


def load_eprof(opt, lasfn):
    if opt["eprof"]:
        vals = [float(x) for x in opt["eprof"].split(",")]
    else:
        fn = opt["E"] or (lasfn + ".eprof")
        try:
            vals = [float(x) for x in open(fn).read().replace(",", " ").split()]
        except (OSError, UnicodeDecodeError, ValueError):
            raise SystemExit("daccord_amd: no usable error profile: give --eprof<p_i,p_d,est_cor> or a text file via -E (%s)" % fn)
    if len(vals) != 3:
        raise SystemExit("daccord_amd: the error profile needs three numbers: p_i,p_d,est_cor")
    return vals


def main(argv=None, out=None):
    opt, pos = parse_args(sys.argv[1:] if argv is None else argv)
    out = out or sys.stdout
    lasfn, dbfn = pos[0], pos[1]
    from . import engine, io as dio, shard
    from ._structs import default_params
    las = dio.LasFile(lasfn)
    bps, boff, rlen = dio.read_db(dbfn)
    if len(pos) > 2:
        # asymmetric mode (daccord.cpp:1337-1364): B reads come from the second database; ids of B are offset past A's
        bps2, boff2, rlen2 = dio.read_db(pos[2])
        boff = np.concatenate([boff, boff2 + len(bps)]); bps = np.concatenate([bps, bps2]); nA = len(rlen); rlen = np.concatenate([rlen, rlen2])
    else:
        nA = 0
    k = [int(x) for x in opt["k"].split(",")]
    kw = dict(w=opt["w"], a=opt["a"], klow=k[0], khigh=k[-1], minfilterfreq=opt["minfilterfreq"], maxfilterfreq=opt["maxfilterfreq"],
              minwindowcov=opt["m"], minlen=opt["l"], producefull=1 if opt["f"] else 0, tspace=las.tspace)
    if opt["d"] is not None:
        kw["maxalign"] = opt["d"]
    if opt["e"] is not None:
        kw["eminrate"] = opt["e"]
    p = default_params(**kw)
    lo, hi = las.min_aread, las.max_aread + 1
    if opt["I"]:
        a, b = [int(x) for x in opt["I"].split(",")]; lo, hi = max(lo, a), min(hi, b)
    if opt["J"]:
        g, G = [int(x) for x in opt["J"].split(",")]; lo, hi = shard.shard_range(lo, hi, g, G)
    E = engine.Engine(p)
    E.set_error_profile(*load_eprof(opt, lasfn))
    E.load_db(bps, boff, rlen)
    well = 0
    batch = 2000                                  # A reads per device batch
    for b0 in range(lo, hi, batch):
        piles, ovl, trace = las.piles(b0, min(hi, b0 + batch))
        if not len(piles):
            continue
        if nA:
            ovl["bread"] += nA
        ovl, piles = engine.pile_select(ovl, piles, trace_bytes=las.trace_bytes, maxinput=opt["D"])
        frags, bases = E(piles, ovl, trace, trace_bytes=las.trace_bytes)
        out.write(engine.fasta(frags, bases, start_well=well)); well += len(frags)
    E.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
