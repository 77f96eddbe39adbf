"""The C++ front end daccord_hip (daccord_amd/csrc/daccord_hip_main.cpp) without a GPU: option grammar (flag and value
joined, options before positionals: src/daccord.cpp:185-207, README.md:99), read interval logic (:1156-1230) and the
error profile estimation (--eprofonly needs no device)."""
import re
import numpy as np
import pytest
from daccord_amd import cli, io as dio


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    from daccord_amd.synth import SynthData
    d = SynthData(60000, 120, 3000, seed=1)
    tmp = tmp_path_factory.mktemp("cli")
    las, db = str(tmp / "reads.las"), str(tmp / "reads.db")
    dio.write_db(db, d.bps, d.boff, d.rlen)
    dio.write_las(las, 100, d.ovl, d.trace)
    return d, las, db


def _interval(stderr):
    m = re.search(r"\[V\] minaread=(-?\d+) toparead=(-?\d+)", stderr)
    return int(m.group(1)), int(m.group(2))


def test_usage_and_unknown_options_fail_loudly(files):
    d, las, db = files
    for bad in (["-Q3", las, db], ["--nosuchoption", las, db], [las], ["-I5", "--eprofonly", las, db], ["-J1,0", "--eprofonly", las, db],
                ["--gpus0", "--eprofonly", las, db], ["--gpus65", "--eprofonly", las, db], ["--gpusx", "--eprofonly", las, db]):
        r = cli.run(bad)
        assert r.returncode != 0 and (b"[E]" in r.stderr or b"usage" in r.stderr), bad
    # parameter ranges are refused before any file is opened, in the option's words (the reference: "k-mer size k is not compiled in",
    # DebruijnGraphContainer.hpp:41-110; here k = 3 ... 16, w <= 128)
    for bad, word in ((["-k17"], b"not compiled in"), (["-k2"], b"not compiled in"), (["-k12,9"], b"not compiled in"), (["-w129"], b"-w must"),
                      (["-w0"], b"-w must"), (["-a0"], b"-a must"), (["--minfilterfreq3", "--maxfilterfreq2"], b"filterfreq")):
        r = cli.run(bad + ["/nonexistent.las", "/nonexistent.db"])
        assert r.returncode != 0 and word in r.stderr and b"cannot open" not in r.stderr, (bad, r.stderr)


def test_read_intervals_follow_the_reference(files):
    d, las, db = files
    lo, hi = int(d.ovl["aread"].min()), int(d.ovl["aread"].max())
    ep = ["--eprof0.12,0.02,0.85", "--eprofonly"]
    assert _interval(cli.run(ep + [las, db]).stderr.decode()) == (lo, hi + 1)
    # -I: both ends inclusive (daccord.cpp:1225-1230)
    assert _interval(cli.run(ep + ["-I3,9", las, db]).stderr.decode()) == (max(lo, 3), min(hi, 9) + 1)
    # -J i,j: part i of j, partsize = ceil(span/j) (daccord.cpp:1156-1183); -J wins over -I (else if, :1185)
    span = hi + 1 - lo; part = (span + 3) // 4
    assert _interval(cli.run(ep + ["-J1,4", "-I0,1", las, db]).stderr.decode()) == (lo + part, min(lo + 2 * part, hi + 1))
    assert _interval(cli.run(ep + ["-J7,4", las, db]).stderr.decode()) == (0, -1)


def test_eprofonly_estimates_and_writes_the_profile(files, tmp_path):
    d, las, db = files
    ef = str(tmp_path / "x.eprof")
    r = cli.run(["--eprofonly", "-E" + ef, "-w48", "-a12", "-k10,12", "-f", "-d30", "-t16", "--minfilterfreq1", las, db])
    assert r.returncode == 0, r.stderr.decode()
    got = [float(x) for x in open(ef).read().split()]
    ovl, piles = dio.select_lowest(d.ovl, d.piles)
    c, us, un, prof = dio.estimate_profile(d.bps, d.boff, d.rlen, 100, piles[:1024], ovl, d.trace, maxalign=30, nthreads=4)
    assert got == list(prof)
    assert ("usable=%d unusable=%d" % (us, un)) in r.stderr.decode()
    # an existing profile file is used as it is
    r2 = cli.run(["--eprofonly", "-E" + ef, las, db])
    assert ("p_i=%.17g" % prof[0]) in r2.stderr.decode() and "usable=" not in r2.stderr.decode()


def test_reference_binary_profile_is_written_and_read(files, tmp_path):
    """--binaryeprof writes the reference's own .eprof (src/daccord.cpp:1855-1860: libmaus2 AlignmentStatistics::serialise = matches,
    mismatches, insertions, deletions as 8-byte big-endian numbers, then eavg and edif as raw doubles -- libmaus2's published layout,
    restated; no file written by the reference exists here) and either form is read back to the same rates (:1867-1878)."""
    import struct
    d, las, db = files
    ef, et = str(tmp_path / "b.eprof"), str(tmp_path / "t.eprof")
    r = cli.run(["--eprofonly", "--binaryeprof", "-E" + ef, las, db])
    assert r.returncode == 0, r.stderr.decode()
    raw = open(ef, "rb").read()
    assert len(raw) == 48
    m, mm, ins, dele = struct.unpack(">4Q", raw[:32])
    eavg, edif = struct.unpack("=2d", raw[32:])
    se = r.stderr.decode()
    assert ("AlignmentStatistics(matches=%d,mismatches=%d,insertions=%d,deletions=%d)" % (m, mm, ins, dele)) in se
    assert ("eavg=%g edif=%g" % (eavg, edif)) in se
    # the text form of the same estimate
    assert cli.run(["--eprofonly", "-E" + et, las, db]).returncode == 0
    prof = [float(x) for x in open(et).read().split()]
    ln = m + mm + dele
    assert prof == [ins / ln, dele / ln, 1.0 - (mm + dele + ins) / ln]
    # read back: the binary file gives the same three rates as the text one, and is reported as such
    rb, rt = cli.run(["--eprofonly", "-V2", "-E" + ef, las, db]), cli.run(["--eprofonly", "-V2", "-E" + et, las, db])
    assert rb.returncode == 0 and rt.returncode == 0
    line = "[V] p_i=%.17g p_d=%.17g est_cor=%.17g" % tuple(prof)
    assert line in rb.stderr.decode() and line in rt.stderr.decode()
    assert "reference binary form" in rb.stderr.decode() and "(text)" in rt.stderr.decode() and "usable=" not in rb.stderr.decode()
    # a file of 48 bytes that is not a profile, and a truncated one, are refused
    bad = str(tmp_path / "bad.eprof")
    open(bad, "wb").write(b"\xff" * 48)
    assert cli.run(["--eprofonly", "-E" + bad, "--keepeprof", las, db]).returncode != 0
    open(bad, "wb").write(raw[:40])
    assert cli.run(["--eprofonly", "-E" + bad, "--keepeprof", las, db]).returncode != 0


def test_loaderonly_measures_the_host_side_without_a_device(files):
    """--loaderonly (round 5): loader thread + planner threads, no HIP context; reports piles, windows and rates on stderr and writes no FASTA.
    The window count is the planner's (dacc_plan_only) = what dacc_submit_piles would schedule: one window per step of -a over every A read
    (HandleContext.hpp:390-408) up to the pile's largest aepos, here checked against the closed form."""
    d, las, db = files
    r = cli.run(["--eprof0.12,0.02,0.85", "-k10", "--loaderonly", "--gpus3", "--batch25", las, db])
    assert r.returncode == 0 and r.stdout == b"", r.stderr.decode()
    err = r.stderr.decode()
    m = re.search(r"\[L\] host side only: (\d+) piles \((\d+) selected overlaps, (\d+) A-read bases, (\d+) windows\) in (\d+) batches of 25 A reads", err)
    assert m, err
    npiles, novl, abases, nwin, nbat = (int(x) for x in m.groups())
    areads = np.unique(d.ovl["aread"])
    assert npiles == len(areads) and abases == int(d.rlen[areads].sum()) and nbat >= npiles // 25
    w, a = 40, 10

    def windows_n(l):      # Windows::computeN (HandleContext.hpp:390-408) over [0, l), l = the largest aepos of the pile (HandleContext.hpp:1760-1778)
        npre = (l + a - w) // a if l + a >= w else 0
        if npre:
            return npre if (npre - 1) * a + w == l else npre + 1
        return 1 if l >= w else 0
    expect = sum(windows_n(int(d.ovl["aepos"][d.ovl["aread"] == r].max())) for r in areads)
    assert nwin == expect, (nwin, expect)
    assert "[L] loader thread:" in err and "3 planner thread(s)" in err
