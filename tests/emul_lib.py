"""ctypes wrapper of the host emulation of the kernels (tests/emul/emul.cpp): TEST HARNESS ONLY.
Two builds of the same device headers: a 1-lane wavefront (fast, logic only) and a real 64-lane wavefront with one
coroutine per lane (daccord_amd/csrc/wave_emul64.hpp: ballots, scans, shuffles and barriers behave as on the GPU)."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
sys.path.insert(0, _ROOT)
from daccord_amd._structs import DaccParams, DaccFragment, DaccWindowResult  # noqa: E402

_SO = os.environ.get("DACC_EMUL_LIB") or os.path.join(_HERE, "emul", "libdacc_emul.so")
_SO64 = os.path.join(_HERE, "emul", "libdacc_emul64.so")
_SRCS = [os.path.join(_HERE, "emul", "emul.cpp")] + [os.path.join(_ROOT, "daccord_amd", "csrc", f)
                                                      for f in os.listdir(os.path.join(_ROOT, "daccord_amd", "csrc"))
                                                      if f.endswith((".hpp", ".cpp"))]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in _SRCS):
        subprocess.check_call(["g++", "-O2", "-w", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-shared", "-o", _SO,
                               os.path.join(_HERE, "emul", "emul.cpp"),
                               os.path.join(_ROOT, "daccord_amd", "csrc", "host_tables.cpp")])
    if force or not os.path.exists(_SO64) or os.path.getmtime(_SO64) < max(os.path.getmtime(s) for s in _SRCS):
        subprocess.check_call(["g++", "-O2", "-w", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-shared", "-DDACC_EMUL_LANES=64",
                               "-DDACC_EMUL_IMPL", "-o", _SO64, os.path.join(_HERE, "emul", "emul.cpp"),
                               os.path.join(_ROOT, "daccord_amd", "csrc", "host_tables.cpp")])
    return _SO


_libs = {}


def lib(lanes=1):
    if lanes not in _libs:
        build()
        L = C.CDLL(_SO64 if lanes == 64 else _SO)
        L.emul_create.restype = C.c_void_p
        L.emul_create.argtypes = [C.POINTER(DaccParams)]
        L.emul_destroy.argtypes = [C.c_void_p]
        L.emul_error.restype = C.c_char_p
        L.emul_error.argtypes = [C.c_void_p]
        L.emul_set_error_profile.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.emul_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.emul_load_db.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
        L.emul_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int]
        L.emul_collect.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.emul_windows.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.emul_set_fast.argtypes = [C.c_void_p, C.c_int]
        L.emul_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.emul_counts4.argtypes = [C.c_void_p, C.c_void_p]
        L.emul_count_long.restype = C.c_uint64
        L.emul_count_long.argtypes = [C.c_void_p]
        _libs[lanes] = L
    return _libs[lanes]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Emul:
    def __init__(self, params, lanes=1):
        self.L = lib(lanes)
        self.h = self.L.emul_create(C.byref(params))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emul_destroy(self.h)
            self.h = None

    def set_fast(self, on):
        self.L.emul_set_fast(self.h, 1 if on else 0)

    def counts(self):
        n = (C.c_uint64 * 4)()
        self.L.emul_counts4(self.h, n)
        return tuple(int(x) for x in n)

    def count_tier7(self):
        """windows that finished in tier 7 (the middle size class of shallow batches)"""
        self.L.emul_count_tier7.restype = C.c_uint64; self.L.emul_count_tier7.argtypes = [C.c_void_p]
        return int(self.L.emul_count_tier7(self.h))

    def count_tier10(self):
        """windows that finished in tier 10 (the dense-graph tier between tier 6 and tier 3 of shallow batches)"""
        self.L.emul_count_tier10.restype = C.c_uint64; self.L.emul_count_tier10.argtypes = [C.c_void_p]
        return int(self.L.emul_count_tier10(self.h))

    def count_tier0(self):
        """windows finished by tier 0 (size classes: the small windows of a shallow batch)"""
        self.L.emul_count_tier0.restype = C.c_uint64; self.L.emul_count_tier0.argtypes = [C.c_void_p]
        return int(self.L.emul_count_tier0(self.h))

    def count_long(self):
        """windows finished by tier 5 (strings of 65..128 bases)"""
        return int(self.L.emul_count_long(self.h))

    def set_error_profile(self, p_i, p_d, est_cor):
        self.L.emul_set_error_profile(self.h, p_i, p_d, est_cor)

    def tables(self, klimit_n=128):
        n = C.c_uint64()
        self.L.emul_tables(self.h, None, 0, C.byref(n), klimit_n)
        out = np.zeros(n.value, dtype=np.uint64)
        self.L.emul_tables(self.h, _ptr(out), n.value, C.byref(n), klimit_n)
        return out

    def load_db(self, bps, boff, rlen):
        bps = np.ascontiguousarray(bps); boff = np.ascontiguousarray(boff); rlen = np.ascontiguousarray(rlen)
        self.L.emul_load_db(self.h, _ptr(bps), len(bps), _ptr(boff), _ptr(rlen), len(rlen))

    def run(self, piles, ovl, trace, trace_bytes=1):
        piles = np.ascontiguousarray(piles); ovl = np.ascontiguousarray(ovl); trace = np.ascontiguousarray(trace)
        rc = self.L.emul_run(self.h, _ptr(piles), len(piles), _ptr(ovl), len(ovl), _ptr(trace), len(trace), trace_bytes)
        if rc:
            raise RuntimeError("emul_run rc=%d: %s" % (rc, self.L.emul_error(self.h).decode()))
        fr = C.c_void_p(); nf = C.c_uint64(); ba = C.c_void_p(); nb = C.c_uint64()
        self.L.emul_collect(self.h, C.byref(fr), C.byref(nf), C.byref(ba), C.byref(nb))
        frags = np.frombuffer((C.c_char * (nf.value * C.sizeof(DaccFragment))).from_address(fr.value),
                              dtype=np.dtype(DaccFragment)).copy() if nf.value else np.zeros(0, np.dtype(DaccFragment))
        bases = C.string_at(ba.value, nb.value) if nb.value else b""
        return frags, bases

    def windows(self):
        n = C.c_uint64()
        self.L.emul_windows(self.h, None, 0, C.byref(n))
        out = np.zeros(n.value, dtype=np.dtype(DaccWindowResult))
        if n.value:
            self.L.emul_windows(self.h, _ptr(out), n.value, C.byref(n))
        return out
