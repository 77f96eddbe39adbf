"""ORACLE python wrapper (test infrastructure).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this; the product package never does."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from daccord_amd._structs import (DaccParams, DaccOverlap, DaccPile, DaccFragment, DaccWindowResult)  # noqa: E402

_SO = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")      # ORACLE_LIB: a separately built copy (scripts/exposure_report.py)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "daccord_hip.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(DaccParams)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_error_profile.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.oracle_load_db.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.oracle_run_piles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                       C.c_int, C.c_int, C.c_int]
        L.oracle_collect.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.oracle_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.oracle_align.restype = C.c_uint64
        L.oracle_align.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_edit_distance.restype = C.c_uint64
        L.oracle_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.oracle_windows_count.restype = C.c_uint64
        L.oracle_windows_count.argtypes = [C.c_uint64] * 3
        L.oracle_pile_select.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_window_consensus.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_char_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_pile_select_lowest.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_estimate_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int] + [C.c_void_p] * 4
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pile_select(ovl, piles, trace_bytes=1, maxinput=5000):
    """P1 for every pile: returns (ovl_sorted, piles_sorted)."""
    L = lib()
    out = np.zeros(len(ovl), dtype=ovl.dtype)
    newp = piles.copy()
    o = 0
    for i, p in enumerate(piles):
        n = C.c_uint64(0)
        seg = np.ascontiguousarray(ovl[p["first_ovl"]:p["first_ovl"] + p["novl"]])
        dst = np.zeros(len(seg), dtype=ovl.dtype)
        L.oracle_pile_select(_ptr(seg), len(seg), trace_bytes, maxinput, _ptr(dst), C.byref(n))
        out[o:o + n.value] = dst[:n.value]
        newp[i]["first_ovl"] = o
        newp[i]["novl"] = n.value
        o += n.value
    return out[:o].copy(), newp


def select_lowest(ovl, piles, maxinput=5000):
    """The estimator's pile selection (src/daccord.cpp:1705-1755) for every pile."""
    L = lib()
    out = np.zeros(len(ovl), dtype=ovl.dtype); newp = piles.copy(); o = 0
    for i, p in enumerate(piles):
        n = C.c_uint64(0)
        seg = np.ascontiguousarray(ovl[p["first_ovl"]:p["first_ovl"] + p["novl"]])
        dst = np.zeros(max(len(seg), 1), dtype=ovl.dtype)
        L.oracle_pile_select_lowest(_ptr(seg), len(seg), maxinput, _ptr(dst), C.byref(n))
        out[o:o + n.value] = dst[:n.value]; newp[i]["first_ovl"] = o; newp[i]["novl"] = n.value; o += n.value
    return out[:o].copy(), newp


class Oracle:
    def __init__(self, params):
        self.L = lib()
        self.params = params
        self.h = self.L.oracle_create(C.byref(params))
        if not self.h:
            raise ValueError("oracle_create failed (bad parameters)")
        self._keep = []

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_destroy(self.h)
            self.h = None

    def set_error_profile(self, p_i, p_d, est_cor):
        self.L.oracle_set_error_profile(self.h, p_i, p_d, est_cor)

    def load_db(self, bps, boff, rlen):
        self._keep = [np.ascontiguousarray(bps), np.ascontiguousarray(boff), np.ascontiguousarray(rlen)]
        self.L.oracle_load_db(self.h, _ptr(self._keep[0]), len(bps), _ptr(self._keep[1]), _ptr(self._keep[2]), len(rlen))

    def estimate_profile(self, piles, ovl, trace, trace_bytes=1, maxalign=2 ** 64 - 1, two_databases=False):
        """src/daccord.cpp:1653-1878 over the given (already selected) piles: (counts, usable, unusable, (p_i,p_d,est_cor))."""
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        counts = np.zeros(4, np.uint64); us = C.c_uint64(); un = C.c_uint64(); prof = np.zeros(3, np.float64)
        rc = self.L.oracle_estimate_profile(self.h, _ptr(piles), len(piles), _ptr(ovl), _ptr(trace), trace_bytes, maxalign,
                                            1 if two_databases else 0, _ptr(counts), C.byref(us), C.byref(un), _ptr(prof))
        if rc:
            raise ValueError("no usable window")
        return counts, us.value, un.value, tuple(float(x) for x in prof)

    def deep_profile(self, piles, ovl, trace, trace_bytes=1, maxalign=2 ** 64 - 1, two_databases=False):
        """src/daccord.cpp:1442-1650 (--deepprofileonly) over the given (already selected) piles: sorted uint32 window error rates."""
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        self.L.oracle_deep_profile.restype = C.c_uint64
        self.L.oracle_deep_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64]
        n = self.L.oracle_deep_profile(self.h, _ptr(piles), len(piles), _ptr(ovl), _ptr(trace), trace_bytes, maxalign, 1 if two_databases else 0, None, 0)
        out = np.zeros(n, np.uint32)
        if n:
            self.L.oracle_deep_profile(self.h, _ptr(piles), len(piles), _ptr(ovl), _ptr(trace), trace_bytes, maxalign, 1 if two_databases else 0, _ptr(out), n)
        return out

    def run(self, piles, ovl, trace, trace_bytes=1, nthreads=1, want_windows=False):
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        rc = self.L.oracle_run_piles(self.h, _ptr(piles), len(piles), _ptr(ovl), len(ovl), _ptr(trace), len(trace),
                                     trace_bytes, nthreads, 1 if want_windows else 0)
        if rc:
            raise RuntimeError("oracle_run_piles rc=%d" % rc)
        fr = C.c_void_p(); nf = C.c_uint64(); ba = C.c_void_p(); nb = C.c_uint64()
        self.L.oracle_collect(self.h, C.byref(fr), C.byref(nf), C.byref(ba), C.byref(nb))
        frags = np.frombuffer((C.c_char * (nf.value * C.sizeof(DaccFragment))).from_address(fr.value),
                              dtype=np.dtype(DaccFragment)).copy() if nf.value else np.zeros(0, np.dtype(DaccFragment))
        bases = C.string_at(ba.value, nb.value) if nb.value else b""
        return frags, bases

    def windows(self):
        n = C.c_uint64()
        self.L.oracle_windows(self.h, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=np.dtype(DaccWindowResult))
        if n.value:
            self.L.oracle_windows(self.h, _ptr(out), n.value, C.byref(n))
        return out

    def tables(self, klimit_n=128):
        n = C.c_uint64()
        self.L.oracle_tables(self.h, None, 0, C.byref(n), klimit_n)
        out = np.zeros(n.value, dtype=np.uint64)
        self.L.oracle_tables(self.h, _ptr(out), n.value, C.byref(n), klimit_n)
        return out

    def window_consensus(self, strings, elength):
        lens = np.array([len(s) for s in strings], dtype=np.uint32)
        cat = b"".join(strings)
        cons = C.create_string_buffer(256)
        cl = C.c_uint32(); mr = C.c_uint64(); ff = C.c_int32(-1)
        ok = self.L.oracle_window_consensus(self.h, cat, _ptr(lens), len(strings), elength, cons, C.byref(cl), C.byref(mr), C.byref(ff))
        return bool(ok), cons.raw[:cl.value], mr.value, ff.value


def fasta(frags, bases, start_well=0):
    """FASTA text exactly as HandleContext.hpp:2710-2724 writes it, with the wellcounter field numbered
    sequentially in read order (= the reference's -t1 numbering)."""
    out = []
    well = start_well
    for f in frags:
        s = bytes(bases[f["seq_off"]:f["seq_off"] + f["len"]]).decode()
        out.append(">%d/%d/%d_%d A=[%d,%d]\n" % (f["aread"] + 1, well, f["first"], f["first"] + f["len"], f["first"], f["last"]))
        well += 1
        for i in range(0, len(s), 80):
            out.append(s[i:i + 80] + "\n")
    return "".join(out)
