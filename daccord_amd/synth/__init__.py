"""Synthetic pile generator (ctypes wrapper of synth.cpp): bench and test input."""
import ctypes as C
import os
import subprocess
import numpy as np
from .._structs import DaccOverlap, DaccPile

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdacc_synth.so")


class SynthParams(C.Structure):
    _fields_ = [("genome_len", C.c_uint64), ("nreads", C.c_uint32), ("read_len", C.c_uint32),
                ("p_ins", C.c_double), ("p_del", C.c_double), ("p_sub", C.c_double),
                ("min_overlap", C.c_uint32), ("tspace", C.c_int32), ("seed", C.c_uint64),
                ("nthreads", C.c_int32), ("reserved", C.c_int32), ("afirst", C.c_uint32), ("alast", C.c_uint32)]


def build(force=False):
    src = os.path.join(_HERE, "synth.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-shared", "-o", _SO, src])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.synth_generate.restype = C.c_void_p
        _lib.synth_generate.argtypes = [C.POINTER(SynthParams)]
        _lib.synth_destroy.argtypes = [C.c_void_p]
        _lib.synth_get.argtypes = [C.c_void_p] + [C.c_void_p] * 13
    return _lib


class SynthData:
    """Host-side synthetic data set: 2-bit read store + overlaps + trace points + piles."""

    def __init__(self, genome_len, nreads, read_len, erate=0.15, ins_frac=0.8, del_frac=0.1333333333, sub_frac=0.0666666667,
                 min_overlap=1000, tspace=100, seed=1, nthreads=None, aread_range=None):
        """aread_range = (first, last): overlaps and piles only for these A reads (every read of the set is generated, B reads
        are arbitrary); the records are identical to the corresponding ones of the full set."""
        lib = _load()
        p = SynthParams(genome_len, nreads, read_len, erate * ins_frac, erate * del_frac, erate * sub_frac,
                        min_overlap, tspace, seed, nthreads or (os.cpu_count() or 1), 0,
                        aread_range[0] if aread_range else 0, aread_range[1] if aread_range else 0)
        self.p_ins, self.p_del, self.p_sub = p.p_ins, p.p_del, p.p_sub
        self.tspace = tspace
        self._h = lib.synth_generate(C.byref(p))
        bps = C.c_void_p(); nb = C.c_uint64(); boff = C.c_void_p(); rlen = C.c_void_p(); nr = C.c_uint64()
        ovl = C.c_void_p(); novl = C.c_uint64(); tr = C.c_void_p(); ntr = C.c_uint64()
        piles = C.c_void_p(); npiles = C.c_uint64(); genome = C.c_void_p(); truth = C.c_void_p()
        lib.synth_get(self._h, *[C.byref(x) for x in (bps, nb, boff, rlen, nr, ovl, novl, tr, ntr, piles, npiles, genome, truth)])

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr.value)
            return np.frombuffer(buf, dtype=dt).copy()

        self.bps = arr(bps, nb.value, np.uint8)
        self.boff = arr(boff, nr.value, np.uint64)
        self.rlen = arr(rlen, nr.value, np.uint32)
        self.ovl = arr(ovl, novl.value, np.dtype(DaccOverlap))
        # trace values: uint8 up to tspace 125, uint16 beyond (DALIGNER's rule); trace_off counts values
        self.trace_bytes = 2 if tspace > 125 else 1
        if self.trace_bytes == 2:
            self.trace = arr(tr, ntr.value // 2 + 8, np.uint16)
            self.ntrace = ntr.value // 2
        else:
            self.trace = arr(tr, ntr.value + 8, np.uint8)
            self.ntrace = ntr.value
        self.piles = arr(piles, npiles.value, np.dtype(DaccPile))
        self.genome = arr(genome, genome_len, np.uint8)
        self.truth = arr(truth, 3 * nr.value, np.int64).reshape(-1, 3)
        lib.synth_destroy(self._h)
        self._h = None

    @property
    def nreads(self):
        return len(self.rlen)

    def error_profile(self):
        """(p_i, p_d, est_cor) in the reference's definitions (daccord.cpp:1867-1878): the .eprof is
        estimated by aligning read windows to the window consensus (daccord.cpp:271-631), i.e. these are
        per-read rates against the truth."""
        pi, pd, ps = self.p_ins, self.p_del, self.p_sub
        return pi, pd, 1.0 - (pi + pd + ps)
