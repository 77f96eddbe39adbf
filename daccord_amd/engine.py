"""Host-side mirror of the reference's operator interface for the consensus path.

`Engine` plays the role of daccord's per-thread `HandleContext` (src/HandleContext.hpp:332-380,
call operator :1699-2901): construct it with the command line parameters, give it the error
profile and the read database, then call it on piles (overlaps of one A read, sorted by abpos) and
get the corrected FASTA fragments back.  Every call goes through the C ABI of libdaccord_hip.so
(include/daccord_hip.h); there is no Python or CPU implementation of the path here, and loading
fails loudly if the HIP library is missing."""
import ctypes as C
import os
import numpy as np
from ._structs import (DaccParams, DaccOverlap, DaccPile, DaccFragment, DaccTiming, DaccWindowResult, default_params)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("DACC_LIB") or os.path.join(_HERE, "libdaccord_hip.so")

ERRORS = {-1: "EINVAL", -2: "ENODEV", -3: "ENOMEM", -4: "ESTATE", -5: "EHIP", -6: "ENOTSUP", -7: "EINTERNAL"}


class DaccError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("libdaccord_hip: %s (%d) %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code


_lib = None


def lib():
    """Load libdaccord_hip.so (the gfx950 HIP library).  No fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("libdaccord_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU implementation of this path)")
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.dacc_create.argtypes = [C.POINTER(vp), C.POINTER(DaccParams)]
        L.dacc_destroy.argtypes = [vp]
        L.dacc_device_count.argtypes = []
        L.dacc_destroy.restype = None
        L.dacc_set_error_profile.argtypes = [vp, C.c_double, C.c_double, C.c_double]
        L.dacc_load_db.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64]
        L.dacc_submit_piles.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, C.c_int]
        L.dacc_collect.argtypes = [vp, vp, vp, vp, vp]
        L.dacc_release.argtypes = [vp]
        L.dacc_release.restype = None
        L.dacc_last_error.argtypes = [vp]
        L.dacc_last_error.restype = C.c_char_p
        L.dacc_pile_select.argtypes = [vp, C.c_uint64, C.c_int, C.c_uint64, vp, vp]
        L.dacc_last_timing.argtypes = [vp, C.POINTER(DaccTiming)]
        L.dacc_rerun_resident.argtypes = [vp]
        L.dacc_debug_windows.argtypes = [vp, vp, C.c_uint64, vp]
        L.dacc_debug_tables.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
        L.dacc_debug_profile.argtypes = [vp, vp]
        if hasattr(L, "dacc_debug_profile_fine"):      # (variant libraries of earlier rounds, DACC_LIB, lack it)
            L.dacc_debug_profile_fine.argtypes = [vp, vp]
        L.dacc_debug_retry.argtypes = [vp, vp, C.c_uint64, vp]
        L.dacc_pile_status.argtypes = [vp, vp, C.c_uint64, vp]
        L.dacc_pile_errors.restype = C.c_char_p
        L.dacc_pile_errors.argtypes = [vp]
        _lib = L
    return _lib


EXPORTS = ["dacc_device_count", "dacc_create", "dacc_destroy", "dacc_set_error_profile", "dacc_load_db", "dacc_submit_piles", "dacc_collect",
           "dacc_release", "dacc_last_error", "dacc_pile_select", "dacc_last_timing", "dacc_rerun_resident",
           "dacc_debug_windows", "dacc_debug_tables", "dacc_debug_profile", "dacc_debug_profile_fine", "dacc_debug_retry", "dacc_pile_status", "dacc_pile_errors"]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pile_select(ovl, piles, trace_bytes=1, maxinput=5000):
    """The pile loader's top-D selection + sort by abpos (src/daccord.cpp:2120-2288) for every pile."""
    L = lib()
    out = np.zeros(len(ovl), dtype=ovl.dtype)
    newp = piles.copy()
    o = 0
    for i, p in enumerate(piles):
        n = C.c_uint64(0)
        seg = np.ascontiguousarray(ovl[p["first_ovl"]:p["first_ovl"] + p["novl"]])
        dst = np.zeros(max(len(seg), 1), dtype=ovl.dtype)
        rc = L.dacc_pile_select(_ptr(seg), len(seg), trace_bytes, maxinput, _ptr(dst), C.byref(n))
        if rc:
            raise DaccError(rc)
        out[o:o + n.value] = dst[:n.value]
        newp[i]["first_ovl"] = o
        newp[i]["novl"] = n.value
        o += n.value
    return out[:o].copy(), newp


class Engine:
    def __init__(self, params=None, **kw):
        self.L = lib()
        self.params = params if params is not None else default_params(**kw)
        h = C.c_void_p()
        rc = self.L.dacc_create(C.byref(h), C.byref(self.params))
        if rc:
            raise DaccError(rc, "dacc_create (no usable HIP device?)" if rc == -2 else "dacc_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.dacc_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise DaccError(rc, (self.L.dacc_last_error(self.h) or b"").decode())

    def set_error_profile(self, p_i, p_d, est_cor):
        self._chk(self.L.dacc_set_error_profile(self.h, p_i, p_d, est_cor))

    def load_db(self, bps, boff, rlen):
        bps = np.ascontiguousarray(bps, dtype=np.uint8); boff = np.ascontiguousarray(boff, dtype=np.uint64)
        rlen = np.ascontiguousarray(rlen, dtype=np.uint32)
        self._chk(self.L.dacc_load_db(self.h, _ptr(bps), len(bps), _ptr(boff), _ptr(rlen), len(rlen)))

    def __call__(self, piles, ovl, trace, trace_bytes=1):
        """Correct a batch of piles; returns (fragments, bases)."""
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        self._chk(self.L.dacc_submit_piles(self.h, _ptr(piles), len(piles), _ptr(ovl), len(ovl), _ptr(trace),
                                           trace.nbytes // trace_bytes, trace_bytes))
        return self.collect()

    correct = __call__

    def collect(self):
        fr = C.c_void_p(); nf = C.c_uint64(); ba = C.c_void_p(); nb = C.c_uint64()
        self._chk(self.L.dacc_collect(self.h, C.byref(fr), C.byref(nf), C.byref(ba), C.byref(nb)))
        frags = np.frombuffer((C.c_char * (nf.value * C.sizeof(DaccFragment))).from_address(fr.value),
                              dtype=np.dtype(DaccFragment)).copy() if nf.value else np.zeros(0, np.dtype(DaccFragment))
        bases = C.string_at(ba.value, nb.value) if nb.value else b""
        return frags, bases

    def rerun(self):
        """Re-run the device part on the batch already resident in HBM (bench timing)."""
        self._chk(self.L.dacc_rerun_resident(self.h))

    def timing(self):
        t = DaccTiming()
        self._chk(self.L.dacc_last_timing(self.h, C.byref(t)))
        return t

    def debug_windows(self):
        n = C.c_uint64()
        self._chk(self.L.dacc_debug_windows(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.dtype(DaccWindowResult))
        if n.value:
            self._chk(self.L.dacc_debug_windows(self.h, _ptr(out), n.value, C.byref(n)))
        return out

    def profile(self):
        out = np.zeros(32, dtype=np.uint64)
        self._chk(self.L.dacc_debug_profile(self.h, _ptr(out)))
        return out

    def profile_fine(self):
        """(cycles[48], visits[48]) of the fine sites of a -DDACC_PROFILE build (scripts/prof_sites.py)."""
        out = np.zeros(96, dtype=np.uint64)
        self._chk(self.L.dacc_debug_profile_fine(self.h, _ptr(out)))
        return out[:48], out[48:]

    def pile_status(self):
        """(status per pile of the last batch, list of messages for dropped piles)."""
        n = C.c_uint64()
        self._chk(self.L.dacc_pile_status(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.int32)
        if n.value:
            self._chk(self.L.dacc_pile_status(self.h, _ptr(out), n.value, C.byref(n)))
        msg = (self.L.dacc_pile_errors(self.h) or b"").decode()
        return out, [m for m in msg.split("\n") if m]

    def debug_retry(self):
        n = C.c_uint64()
        self._chk(self.L.dacc_debug_retry(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            self._chk(self.L.dacc_debug_retry(self.h, _ptr(out), n.value, C.byref(n)))
        return out.reshape(-1, 4)

    def tables(self, klimit_n=128):
        n = C.c_uint64()
        self._chk(self.L.dacc_debug_tables(self.h, None, 0, C.byref(n), klimit_n))
        out = np.zeros(n.value, dtype=np.uint64)
        self._chk(self.L.dacc_debug_tables(self.h, _ptr(out), n.value, C.byref(n), klimit_n))
        return out


def fasta(frags, bases, start_well=0):
    """FASTA text as HandleContext.hpp:2710-2724 writes it; the wellcounter field is numbered sequentially in
    read order (the reference's -t1 numbering; with -t>1 the reference's numbering is schedule dependent)."""
    out = []
    well = start_well
    for f in frags:
        s = bytes(bases[f["seq_off"]:f["seq_off"] + f["len"]]).decode()      # bases: bytes, bytearray or a memoryview (shard.gather_fragments)
        out.append(">%d/%d/%d_%d A=[%d,%d]\n" % (f["aread"] + 1, well, f["first"], f["first"] + f["len"], f["first"], f["last"]))
        well += 1
        for i in range(0, len(s), 80):
            out.append(s[i:i + 80] + "\n")
    return "".join(out)


from .shard import shard_range, shard_piles, gather_fragments  # noqa: E402,F401  (multi-GPU layer: SURVEY.md 8e)
