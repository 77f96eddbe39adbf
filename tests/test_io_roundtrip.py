"""Dazzler .db / DALIGNER .las writer <-> reader round trips.  The byte layouts are restated from
DAZZ_DB / DALIGNER headers that are not in the reference tree, so this pins reader against writer only (FORMAT UNPINNED)."""
import os
import struct
import numpy as np

from daccord_amd import io as dio
from daccord_amd.synth import SynthData


def _data():
    return SynthData(30000, 60, 2000, seed=9)


def test_db_roundtrip_and_layout(tmp_path):
    d = _data()
    p = str(tmp_path / "reads.db")
    dio.write_db(p, d.bps, d.boff, d.rlen)
    assert os.path.exists(str(tmp_path / ".reads.idx")) and os.path.exists(str(tmp_path / ".reads.bps"))
    # x86-64 DAZZ_DB header: ureads, treads, cutoff, all at 0..15, totlen at 40; DAZZ_READ: rlen at 4, boff at 16 (40 bytes)
    idx = open(str(tmp_path / ".reads.idx"), "rb").read()
    assert len(idx) == 112 + 40 * len(d.rlen)
    ureads, treads, cutoff, allarr = struct.unpack_from("<iiii", idx, 0)
    assert (ureads, treads, cutoff, allarr) == (len(d.rlen), len(d.rlen), 0, 1)
    assert struct.unpack_from("<q", idx, 40)[0] == int(d.rlen.sum())
    assert struct.unpack_from("<i", idx, 112 + 40 * 3 + 4)[0] == int(d.rlen[3])
    bps, boff, rlen = dio.read_db(p)
    assert np.array_equal(rlen, d.rlen)
    for i in (0, 7, len(rlen) - 1):
        nb = (int(rlen[i]) + 3) // 4
        assert np.array_equal(bps[boff[i]:boff[i] + nb], d.bps[d.boff[i]:d.boff[i] + nb])


def test_db_trimmed_view(tmp_path):
    d = _data()
    p = str(tmp_path / "t.db")
    dio.write_db(p, d.bps, d.boff, d.rlen)
    # raise the cutoff in the header and clear DB_ALL: only reads >= cutoff flagged best survive (all are flagged best)
    f = str(tmp_path / ".t.idx")
    b = bytearray(open(f, "rb").read())
    cut = int(np.sort(d.rlen)[len(d.rlen) // 2])
    struct.pack_into("<ii", b, 8, cut, 0)
    open(f, "wb").write(bytes(b))
    _, _, rlen = dio.read_db(p)
    assert np.array_equal(rlen, d.rlen[d.rlen >= cut])


def test_las_roundtrip_and_layout(tmp_path):
    d = _data()
    p = str(tmp_path / "ovl.las")
    dio.write_las(p, 100, d.ovl, d.trace)
    raw = open(p, "rb").read()
    novl, tspace = struct.unpack_from("<qi", raw, 0)
    assert novl == len(d.ovl) and tspace == 100
    o0 = d.ovl[0]
    tlen, diffs, abpos, bbpos, aepos, bepos, flags, aread, bread = struct.unpack_from("<iiiiiiIii", raw, 12)
    assert (tlen, diffs, abpos, bbpos, aepos, bepos, flags, aread, bread) == tuple(int(o0[k]) for k in
            ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread"))
    assert raw[12 + 40:12 + 40 + tlen] == d.trace[int(o0["trace_off"]):int(o0["trace_off"]) + tlen].tobytes()
    las = dio.LasFile(p)
    assert (las.novl, las.tspace, las.trace_bytes) == (len(d.ovl), 100, 1)
    piles, ovl, trace = las.piles()
    assert len(ovl) == len(d.ovl) and int(piles["novl"].sum()) == len(d.ovl)
    for k in ("aread", "bread", "flags", "abpos", "aepos", "bbpos", "bepos", "diffs", "tlen"):
        assert np.array_equal(ovl[k], d.ovl[k])
    for i in (0, len(ovl) // 2, len(ovl) - 1):
        a = trace[int(ovl[i]["trace_off"]):int(ovl[i]["trace_off"]) + int(ovl[i]["tlen"])]
        b = d.trace[int(d.ovl[i]["trace_off"]):int(d.ovl[i]["trace_off"]) + int(d.ovl[i]["tlen"])]
        assert np.array_equal(a, b)
    # a sub range of A reads
    sub_p, sub_o, _ = las.piles(5, 9)
    assert set(sub_p["aread"].tolist()) <= {5, 6, 7, 8} and np.all((sub_o["aread"] >= 5) & (sub_o["aread"] < 9))


def test_las_roundtrip_two_byte_trace_values(tmp_path):
    """tspace > 125: two bytes per trace value in the file (TRACE_XOVR of DALIGNER's align.h), little endian."""
    from daccord_amd.synth import SynthData
    d = SynthData(40000, 80, 4000, seed=9, tspace=300)
    assert d.trace.dtype == np.uint16 and d.trace_bytes == 2 and len(d.ovl) > 50
    p = str(tmp_path / "ovl300.las")
    dio.write_las(p, 300, d.ovl, d.trace[:d.ntrace])
    raw = open(p, "rb").read()
    o0 = d.ovl[0]
    tlen = int(o0["tlen"])
    assert raw[12 + 40:12 + 40 + 2 * tlen] == d.trace[int(o0["trace_off"]):int(o0["trace_off"]) + tlen].astype("<u2").tobytes()
    las = dio.LasFile(p)
    assert (las.novl, las.tspace, las.trace_bytes) == (len(d.ovl), 300, 2)
    piles, ovl, trace = las.piles()
    assert trace.dtype == np.uint16 and (trace > 255).any()
    for i in (0, len(ovl) // 2, len(ovl) - 1):
        a = trace[int(ovl[i]["trace_off"]):int(ovl[i]["trace_off"]) + int(ovl[i]["tlen"])]
        b = d.trace[int(d.ovl[i]["trace_off"]):int(d.ovl[i]["trace_off"]) + int(d.ovl[i]["tlen"])]
        assert np.array_equal(a, b)


def test_las_rejects_garbage(tmp_path):
    p = str(tmp_path / "bad.las")
    open(p, "wb").write(b"\x05\x00\x00\x00\x00\x00\x00\x00\x64\x00\x00\x00" + b"\x00" * 17)
    try:
        dio.LasFile(p)
        assert False, "truncated file accepted"
    except IOError:
        pass
