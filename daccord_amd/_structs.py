"""ctypes mirrors of the plain-C structs in include/daccord_hip.h (the drop-in boundary)."""
import ctypes as C

U64MAX = 0xFFFFFFFFFFFFFFFF


class DaccParams(C.Structure):
    _fields_ = [
        ("w", C.c_uint32), ("a", C.c_uint32), ("klow", C.c_uint32), ("khigh", C.c_uint32),
        ("minfilterfreq", C.c_int32), ("maxfilterfreq", C.c_int32), ("minwindowcov", C.c_uint32),
        ("maxalign", C.c_uint64), ("eminrate", C.c_uint64), ("minlen", C.c_uint64),
        ("producefull", C.c_int32), ("tspace", C.c_int32), ("device", C.c_int32), ("verbose", C.c_int32),
    ]


class DaccOverlap(C.Structure):
    _fields_ = [
        ("aread", C.c_int32), ("bread", C.c_int32), ("flags", C.c_uint32),
        ("abpos", C.c_int32), ("aepos", C.c_int32), ("bbpos", C.c_int32), ("bepos", C.c_int32),
        ("diffs", C.c_int32), ("tlen", C.c_int32), ("reserved", C.c_uint32), ("trace_off", C.c_uint64),
    ]


class DaccPile(C.Structure):
    _fields_ = [("aread", C.c_int32), ("novl", C.c_uint32), ("first_ovl", C.c_uint64)]


class DaccFragment(C.Structure):
    _fields_ = [("aread", C.c_int32), ("first", C.c_uint32), ("last", C.c_uint32), ("len", C.c_uint32),
                ("seq_off", C.c_uint64)]


class DaccTiming(C.Structure):
    _fields_ = [("h2d_ms", C.c_float), ("trace_ms", C.c_float), ("window_ms", C.c_float), ("vote_ms", C.c_float),
                ("d2h_ms", C.c_float), ("total_ms", C.c_float), ("nwindows", C.c_uint64), ("nblocks", C.c_uint64),
                ("algo_bytes", C.c_uint64), ("tier_ms", C.c_float * 3), ("tier_out", C.c_uint32 * 3),
                ("first_tier", C.c_uint32), ("long_windows", C.c_uint32), ("tier0_ms", C.c_float), ("tier0_in", C.c_uint32),
                ("tier0_out", C.c_uint32), ("tier7_ms", C.c_float), ("tier7_in", C.c_uint32), ("tier7_out", C.c_uint32), ("pad_", C.c_uint32),
                ("long_first_tier", C.c_uint32), ("tier10_ms", C.c_float), ("tier10_out", C.c_uint32), ("tier10_ran", C.c_uint32), ("pad2_", C.c_uint32)]


class DaccWindowResult(C.Structure):
    _fields_ = [("pile", C.c_int32), ("y", C.c_int32), ("status", C.c_int32), ("mao", C.c_int32),
                ("elength", C.c_int32), ("k", C.c_int32), ("filterfreq", C.c_int32), ("conslen", C.c_int32),
                ("minrate", C.c_uint64), ("cons", C.c_char * 80)]


def default_params(**kw):
    """daccord's command line defaults (src/daccord.cpp:101-169)."""
    p = DaccParams(w=40, a=10, klow=8, khigh=8, minfilterfreq=0, maxfilterfreq=2, minwindowcov=3,
                   maxalign=U64MAX, eminrate=U64MAX, minlen=0, producefull=0, tspace=100, device=0, verbose=0)
    for k, v in kw.items():
        if k == "k":
            p.klow = p.khigh = v
        else:
            setattr(p, k, v)
    return p
