"""ORACLE SUPPORT python wrapper (test infrastructure): drives oracle/_ref/libdaccord_ref*.so = the REFERENCE'S OWN hot-path
headers compiled against the libmaus2 stand-in (oracle/ref_shim/).  Only tests/ and scripts that validate the oracle import
this; the product package never does.  The library is built in the build container (where /root/reference exists) and travels
to the GPU box as a built file."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from daccord_amd._structs import (DaccParams, DaccFragment)  # noqa: E402

_DIR = os.path.join(_HERE, "_ref")
_SRCS = [os.path.join(_HERE, "ref_shim", f) for f in ("ref_capi.cpp", "build.sh", "libmaus2/shim.hpp", "k16/DebruijnGraphContainer.hpp")]


def so_path(k16=False):
    return os.path.join(_DIR, "libdaccord_ref_k16.so" if k16 else "libdaccord_ref.so")


def reference_present():
    return os.path.exists(os.path.join(os.environ.get("DACC_REFERENCE", "/root/reference"), "src", "HandleContext.hpp"))


def build(force=False):
    """(Re)build oracle/_ref when /root/reference is present; a no-op elsewhere (the GPU box uses the prebuilt files)."""
    if not reference_present():
        return None
    so = so_path(False)
    if force or not os.path.exists(so) or not os.path.exists(so_path(True)) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in _SRCS):
        subprocess.check_call(["bash", os.path.join(_HERE, "ref_shim", "build.sh")], stdout=subprocess.DEVNULL)
    return so


def available(k16=False):
    return os.path.exists(so_path(k16))


_libs = {}


def lib(k16=False):
    if k16 not in _libs:
        build()
        L = C.CDLL(so_path(k16))
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.POINTER(DaccParams)]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_error.restype = C.c_char_p; L.ref_error.argtypes = [C.c_void_p]
        L.ref_log.restype = C.c_char_p; L.ref_log.argtypes = [C.c_void_p]
        L.ref_set_error_profile.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.ref_load_db.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.ref_run_piles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.ref_collect.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.ref_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.ref_pile_select_lowest.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.ref_pile_select.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_read_interval.argtypes = [C.c_int64, C.c_int64, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_char_p, C.c_uint64]
        L.ref_defaults.argtypes = [C.c_void_p]
        L.ref_estimate_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        _libs[k16] = L
    return _libs[k16]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def select_lowest(ovl, piles, maxinput=5000):
    """The estimator's pile selection through the reference's own loop (src/daccord.cpp:1712-1742, :1758) for every pile."""
    L = lib(False)
    out = np.zeros(len(ovl), dtype=ovl.dtype); newp = piles.copy(); o = 0
    for i, p in enumerate(piles):
        n = C.c_uint64(0)
        seg = np.ascontiguousarray(ovl[p["first_ovl"]:p["first_ovl"] + p["novl"]])
        dst = np.zeros(max(len(seg), 1), dtype=ovl.dtype)
        if L.ref_pile_select_lowest(_ptr(seg), len(seg), maxinput, _ptr(dst), C.byref(n)):
            raise RuntimeError("oracle/_ref was built without the selection loop")
        out[o:o + n.value] = dst[:n.value]; newp[i]["first_ovl"] = o; newp[i]["novl"] = n.value; o += n.value
    return out[:o].copy(), newp


def pile_select(ovl, piles, trace_bytes=1, maxinput=5000, vard=0, rl=None, avgreadlength=1.0):
    """The MAIN path's pile selection through the reference's own lines (src/daccord.cpp:2026-2105, :2120-2288: score heap with
    keep-the-worst eviction, 64 KiB input blocks, copy order, sort by abpos) for every pile.  With vard the reference's own
    lmaxinput formula (:2121-2126) is used with the A read lengths rl[i]; the lmaxinput values come back as third result."""
    L = lib(False)
    out = np.zeros(len(ovl), dtype=ovl.dtype); newp = piles.copy(); o = 0; lm = []
    for i, p in enumerate(piles):
        n = C.c_uint64(0); l = C.c_uint64(0)
        seg = np.ascontiguousarray(ovl[p["first_ovl"]:p["first_ovl"] + p["novl"]])
        dst = np.zeros(max(len(seg), 1), dtype=ovl.dtype)
        rc = L.ref_pile_select(_ptr(seg), len(seg), trace_bytes, maxinput, vard, int(rl[i]) if rl is not None else 0, avgreadlength, _ptr(dst), C.byref(n), C.byref(l))
        if rc:
            raise RuntimeError("ref_pile_select: %d (-9: oracle/_ref built without the selection lines, -2: a copied record differs)" % rc)
        out[o:o + n.value] = dst[:n.value]; newp[i]["first_ovl"] = o; newp[i]["novl"] = n.value; o += n.value; lm.append(l.value)
    return out[:o].copy(), newp, lm


def defaults():
    """src/daccord.cpp:106-169 compiled from its lines: the option defaults as a dict"""
    L = lib(False)
    out = (C.c_uint64 * 13)()
    if L.ref_defaults(out):
        raise RuntimeError("oracle/_ref was built without the option defaults")
    return dict(zip(("V", "k", "D", "vard", "d", "w", "a", "f", "m", "e", "l", "minfilterfreq", "maxfilterfreq"), [int(x) for x in out]))


def read_interval(las_min, las_max, J=None, I=None):
    """src/daccord.cpp:1119-1224 + :1227 compiled from the reference's lines: (minaread, toparead), or ValueError with its message"""
    L = lib(False)
    lo = C.c_int64(); top = C.c_int64(); err = C.create_string_buffer(512)
    rc = L.ref_read_interval(las_min, las_max, J.encode() if J is not None else None, I.encode() if I is not None else None, C.byref(lo), C.byref(top), err, 512)
    if rc == -9:
        raise RuntimeError("oracle/_ref was built without the read interval lines")
    if rc:
        raise ValueError(err.value.decode())
    return lo.value, top.value


class Reference:
    """Same call sequence as pyoracle.Oracle; k above 12 needs the k16 build (our factory, the reference's graph template)."""

    def __init__(self, params):
        self.k16 = params.khigh > 12
        self.L = lib(self.k16)
        self.h = self.L.ref_create(C.byref(params))
        if not self.h:
            raise ValueError("ref_create failed (bad parameters)")
        self._keep = []

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_destroy(self.h); self.h = None

    def set_error_profile(self, p_i, p_d, est_cor):
        if self.L.ref_set_error_profile(self.h, p_i, p_d, est_cor):
            raise RuntimeError(self.L.ref_error(self.h).decode())

    def load_db(self, bps, boff, rlen):
        self._keep = [np.ascontiguousarray(bps), np.ascontiguousarray(boff), np.ascontiguousarray(rlen)]
        self.L.ref_load_db(self.h, _ptr(self._keep[0]), len(bps), _ptr(self._keep[1]), _ptr(self._keep[2]), len(rlen))

    def run(self, piles, ovl, trace, trace_bytes=1, nthreads=1, verbose=0):
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        rc = self.L.ref_run_piles(self.h, _ptr(piles), len(piles), _ptr(ovl), len(ovl), _ptr(trace), len(trace), trace_bytes, nthreads, verbose)
        if rc:
            raise RuntimeError("ref_run_piles rc=%d %s" % (rc, self.L.ref_error(self.h).decode()))
        fr = C.c_void_p(); nf = C.c_uint64(); ba = C.c_void_p(); nb = C.c_uint64()
        self.L.ref_collect(self.h, C.byref(fr), C.byref(nf), C.byref(ba), C.byref(nb))
        frags = np.frombuffer((C.c_char * (nf.value * C.sizeof(DaccFragment))).from_address(fr.value),
                              dtype=np.dtype(DaccFragment)).copy() if nf.value else np.zeros(0, np.dtype(DaccFragment))
        bases = C.string_at(ba.value, nb.value) if nb.value else b""
        return frags, bases

    def estimate_profile(self, piles, ovl, trace, trace_bytes=1, maxalign=2 ** 64 - 1, deep=False):
        """src/daccord.cpp:1653-1878 with the reference's handleIndelEstimate<8> (:271-631) over the given (already selected) piles:
        (counts, usable, unusable, (p_i, p_d, est_cor)); deep=True: handleIndelEstimateDeep<8> (:633-995), plus the sorted uint32
        window error rates"""
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        counts = np.zeros(4, np.uint64); us = C.c_uint64(); un = C.c_uint64(); prof = np.zeros(3, np.float64)
        cap = 1 << 22; out = np.zeros(cap if deep else 1, np.uint32); nd = C.c_uint64()
        rc = self.L.ref_estimate_profile(self.h, _ptr(piles), len(piles), _ptr(ovl), _ptr(trace), trace_bytes, maxalign, _ptr(counts), C.byref(us),
                                         C.byref(un), _ptr(prof), 1 if deep else 0, _ptr(out), cap, C.byref(nd))
        if rc == -9:
            raise RuntimeError("oracle/_ref was built without the estimator")
        if rc == -2:
            raise RuntimeError(self.L.ref_error(self.h).decode())
        if rc:
            raise ValueError("no usable window")
        res = (counts, us.value, un.value, tuple(float(x) for x in prof))
        return res + (out[:nd.value].copy(),) if deep else res

    def log(self):
        return self.L.ref_log(self.h).decode()

    def tables(self, klimit_n=128):
        n = C.c_uint64()
        self.L.ref_tables(self.h, None, 0, C.byref(n), klimit_n)
        out = np.zeros(n.value, dtype=np.uint64)
        self.L.ref_tables(self.h, _ptr(out), n.value, C.byref(n), klimit_n)
        return out
