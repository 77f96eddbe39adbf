"""Readers / writers for the inputs of `daccord <in.las> <in.db>` (include/daccord_io.h):
Dazzler read database and DALIGNER overlap file.  ctypes over the host-only libdaccord_io.so.  FORMAT UNPINNED (no
real file available here): validated by round trips only."""
import ctypes as C
import numpy as np

from . import build as _build
from ._structs import DaccOverlap, DaccPile

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_build.build_io())
        vp = C.c_void_p
        L.dacc_db_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.dacc_db_close.argtypes = [vp]
        L.dacc_db_error.restype = C.c_char_p; L.dacc_db_error.argtypes = [vp]
        L.dacc_db_arrays.argtypes = [vp] + [C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.dacc_db_write.argtypes = [C.c_char_p, vp, C.c_uint64, vp, vp, C.c_uint64]
        L.dacc_las_open.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.dacc_las_close.argtypes = [vp]
        L.dacc_las_error.restype = C.c_char_p; L.dacc_las_error.argtypes = [vp]
        L.dacc_las_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.dacc_las_piles.argtypes = [vp, C.c_int64, C.c_int64, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp), C.POINTER(C.c_uint64),
                                     C.POINTER(vp), C.POINTER(C.c_uint64)]
        L.dacc_las_write.argtypes = [C.c_char_p, C.c_int32, vp, C.c_uint64, vp, C.c_uint64, C.c_int]
        L.dacc_pile_select_lowest.argtypes = [vp, C.c_uint64, C.c_int, C.c_uint64, vp, vp]
        L.dacc_eprof_create.argtypes = [C.POINTER(vp), C.c_int32, vp, vp, vp, C.c_uint64, C.c_int]
        L.dacc_eprof_add.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, C.c_int, C.c_uint64, C.c_int]
        L.dacc_eprof_finish.argtypes = [vp] + [vp] * 6
        L.dacc_eprof_destroy.argtypes = [vp]
        L.dacc_eprof_set_deep.argtypes = [vp, C.c_int]
        L.dacc_eprof_deep.argtypes = [vp, vp, vp]
        L.dacc_eprof_skipped.argtypes = [vp, vp, vp]
        L.dacc_read_interval.argtypes = [C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_char_p, C.c_uint64]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _copy(ptr, n, dtype):
    dt = np.dtype(dtype)
    if not n:
        return np.zeros(0, dt)
    return np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt).copy()


def read_db(path):
    """Trimmed view of a Dazzler DB -> (bps uint8[], boff uint64[], rlen uint32[]): the arguments of Engine.load_db."""
    L = lib(); h = C.c_void_p()
    rc = L.dacc_db_open(path.encode(), C.byref(h))
    try:
        if rc:
            raise IOError("dacc_db_open(%s): %s" % (path, L.dacc_db_error(h).decode()))
        bps = C.c_void_p(); nb = C.c_uint64(); boff = C.c_void_p(); rlen = C.c_void_p(); n = C.c_uint64()
        L.dacc_db_arrays(h, C.byref(bps), C.byref(nb), C.byref(boff), C.byref(rlen), C.byref(n))
        return _copy(bps.value, nb.value, np.uint8), _copy(boff.value, n.value, np.uint64), _copy(rlen.value, n.value, np.uint32)
    finally:
        L.dacc_db_close(h)


def write_db(path, bps, boff, rlen):
    bps = np.ascontiguousarray(bps, np.uint8); boff = np.ascontiguousarray(boff, np.uint64); rlen = np.ascontiguousarray(rlen, np.uint32)
    rc = lib().dacc_db_write(path.encode(), _ptr(bps), len(bps), _ptr(boff), _ptr(rlen), len(rlen))
    if rc:
        raise IOError("dacc_db_write(%s) failed: %d" % (path, rc))


class LasFile:
    """A DALIGNER .las file on disk with its A read -> byte offset table (the records are read per requested range)."""

    def __init__(self, path):
        self.L = lib(); self.h = C.c_void_p()
        rc = self.L.dacc_las_open(path.encode(), C.byref(self.h))
        if rc:
            msg = self.L.dacc_las_error(self.h).decode(); self.close()
            raise IOError("dacc_las_open(%s): %s" % (path, msg))
        novl = C.c_int64(); ts = C.c_int32(); tb = C.c_int32(); mn = C.c_int64(); mx = C.c_int64()
        self.L.dacc_las_info(self.h, C.byref(novl), C.byref(ts), C.byref(tb), C.byref(mn), C.byref(mx))
        self.novl, self.tspace, self.trace_bytes, self.min_aread, self.max_aread = novl.value, ts.value, tb.value, mn.value, mx.value

    def close(self):
        if getattr(self, "h", None):
            self.L.dacc_las_close(self.h); self.h = None

    __del__ = close

    def piles(self, afirst=0, alast=None):
        """(piles, ovl, trace) of the A reads in [afirst, alast), records in file order (not yet top-D selected)."""
        if alast is None:
            alast = self.max_aread + 1
        p = C.c_void_p(); np_ = C.c_uint64(); o = C.c_void_p(); no = C.c_uint64(); t = C.c_void_p(); nt = C.c_uint64()
        rc = self.L.dacc_las_piles(self.h, afirst, alast, C.byref(p), C.byref(np_), C.byref(o), C.byref(no), C.byref(t), C.byref(nt))
        if rc:
            raise IOError("dacc_las_piles: %d" % rc)
        tdt = np.uint8 if self.trace_bytes == 1 else np.uint16
        return _copy(p.value, np_.value, np.dtype(DaccPile)), _copy(o.value, no.value, np.dtype(DaccOverlap)), _copy(t.value, nt.value, tdt)


def write_las(path, tspace, ovl, trace):
    ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
    tb = trace.dtype.itemsize
    rc = lib().dacc_las_write(path.encode(), tspace, _ptr(ovl), len(ovl), _ptr(trace), len(trace), tb)
    if rc:
        raise IOError("dacc_las_write(%s) failed: %d" % (path, rc))


def select_lowest(ovl, piles, trace_bytes=1, maxinput=5000):
    """The estimator's pile selection (include/daccord_hip.h: dacc_pile_select_lowest) for every pile."""
    L = lib()
    out = np.zeros(len(ovl), dtype=ovl.dtype); newp = piles.copy(); o = 0
    for i, p in enumerate(piles):
        n = C.c_uint64(0)
        seg = np.ascontiguousarray(ovl[p["first_ovl"]:p["first_ovl"] + p["novl"]])
        dst = np.zeros(max(len(seg), 1), dtype=ovl.dtype)
        if L.dacc_pile_select_lowest(_ptr(seg), len(seg), trace_bytes, maxinput, _ptr(dst), C.byref(n)):
            raise ValueError("dacc_pile_select_lowest")
        out[o:o + n.value] = dst[:n.value]; newp[i]["first_ovl"] = o; newp[i]["novl"] = n.value; o += n.value
    return out[:o].copy(), newp


def estimate_profile(bps, boff, rlen, tspace, piles, ovl, trace, trace_bytes=1, maxalign=2 ** 64 - 1, two_databases=False, nthreads=4, deep=False):
    """Error profile estimation on the host (include/daccord_hip.h: dacc_eprof_*).  Returns
    (counts[matches,mismatches,insertions,deletions], usable, unusable, (p_i, p_d, est_cor)); with deep=True a fifth element:
    the sorted uint32 window error rates of --deepprofileonly (dacc_eprof_deep)."""
    L = lib(); h = C.c_void_p()
    bps = np.ascontiguousarray(bps, np.uint8); boff = np.ascontiguousarray(boff, np.uint64); rlen = np.ascontiguousarray(rlen, np.uint32)
    piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
    if L.dacc_eprof_create(C.byref(h), tspace, _ptr(bps), _ptr(boff), _ptr(rlen), len(rlen), 1 if two_databases else 0):
        raise MemoryError("dacc_eprof_create")
    try:
        if deep:
            L.dacc_eprof_set_deep(h, 1)
        rc = L.dacc_eprof_add(h, _ptr(piles), len(piles), _ptr(ovl), len(ovl), _ptr(trace), trace.nbytes // trace_bytes, trace_bytes, maxalign, nthreads)
        if rc:
            raise ValueError("dacc_eprof_add rc=%d" % rc)
        sk = C.c_uint64(); sn = C.c_uint64()
        L.dacc_eprof_skipped(h, C.byref(sk), C.byref(sn))
        estimate_profile.last_skipped = (sk.value, sn.value)      # (piles left out for malformed records, piles seen)
        counts = np.zeros(4, np.uint64); us = C.c_uint64(); un = C.c_uint64(); ea = C.c_double(); ed = C.c_double(); prof = np.zeros(3, np.float64)
        rc = L.dacc_eprof_finish(h, _ptr(counts), C.byref(us), C.byref(un), C.byref(ea), C.byref(ed), _ptr(prof))
        if rc:
            raise ValueError("no usable window (rc=%d)" % rc)
        if deep:
            vp = C.POINTER(C.c_uint32)(); vn = C.c_uint64()
            L.dacc_eprof_deep(h, C.byref(vp), C.byref(vn))
            vals = np.ctypeslib.as_array(vp, shape=(vn.value,)).copy() if vn.value else np.zeros(0, np.uint32)
            return counts, us.value, un.value, tuple(float(x) for x in prof), vals
        return counts, us.value, un.value, tuple(float(x) for x in prof)
    finally:
        L.dacc_eprof_destroy(h)


def read_interval(las_min, las_max, J=None, I=None):
    """The A reads [minaread, toparead) of a run: -J "part,parts" or -I "first,last" applied to the A reads of the overlap file
    (include/daccord_io.h: dacc_read_interval; src/daccord.cpp:1115-1227).  ValueError with the message for text that does not parse."""
    lo = C.c_int64(); top = C.c_int64(); err = C.create_string_buffer(256)
    rc = lib().dacc_read_interval(las_min, las_max, J.encode() if J is not None else None, I.encode() if I is not None else None,
                                  C.byref(lo), C.byref(top), err, 256)
    if rc:
        raise ValueError(err.value.decode())
    return lo.value, top.value
