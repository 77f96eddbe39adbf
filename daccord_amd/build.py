"""Build the in-tree native libraries.  libdaccord_hip.so is gfx950-only HIP code (hipcc
cross-compiles without a GPU); there is no CPU build of the product."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libdaccord_hip.so")
IOLIB = os.path.join(_HERE, "libdaccord_io.so")
CLI = os.path.join(_HERE, "daccord_hip")

# device code at -Os: the window kernels are one 140-190 KB function each (every helper is inlined so that the LDS layout stays
# a set of immediates); -Os makes tier 1 17 % smaller (174 -> 144 KB) and 1.9 % faster on config 2
# (profiles/r03g_bench_devOs.log vs r03g_bench_default_O3.log); host code in the same translation units stays at -O3
# machine scheduler: the window kernels wait on dependent LDS round trips, so the GCN max-ILP strategy (independent loads
# first) beats the default max-occupancy one by 1.9 % on tier 1 at the same register count (profiles/r03i_scheduler_variants.md)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-Xarch_device", "-Os", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-std=c++17",
               "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]


def csrc_hash():
    """SHA-256 (16 hex digits) over the DEVICE sources (capi.hip and the kernel headers): recorded with PMC summaries
    (scripts/pmc_summarize.py) so that bench.py quotes counters only when they were collected on the very kernels it is
    running (the compiler flags are part of it).  Host-only sources (host_*.cpp, the front end) do not change what the counters
    measure and are left out."""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode()); h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def kernel_isa_hashes(lib=None):
    """{kernel name: SHA-256 (16 hex digits) of its gfx950 instruction stream} of the code object inside the built library
    (llvm-objdump --offloading, then -d per kernel symbol; addresses and encodings left out, mnemonics and operands kept).
    A finer identity than csrc_hash(): a change to ONE kernel's sources (say the generic engine) leaves the machine code of
    the others as it was, and counters collected on those stay valid -- bench.py accepts a PMC summary on either identity.
    Returns {} when the tools are missing."""
    import hashlib, re, shutil, tempfile
    lib = lib or LIB
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(lib) and os.path.exists(objdump)):
        return {}
    d = tempfile.mkdtemp(prefix="dacc_isa_")
    try:
        t = os.path.join(d, "lib.so"); shutil.copy(lib, t)
        subprocess.run([objdump, "--offloading", t], check=True, capture_output=True, cwd=d)
        co = [os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f]
        if not co:
            return {}
        out = {}
        # one code object per translation unit that holds kernels (csrc/window_kernels.hpp: a unit per window kernel)
        for cobj in sorted(co):
            syms = [l.split()[-1] for l in subprocess.run([objdump, "-t", cobj], check=True, capture_output=True, text=True).stdout.splitlines() if " F .text" in l]
            for sym in syms:
                dis = subprocess.run([objdump, "-d", "--no-show-raw-insn", "--disassemble-symbols=" + sym, cobj], check=True, capture_output=True, text=True).stdout
                ins = [m.group(1) for m in (re.match(r"^\s+(\S.*?)\s*//\s*[0-9A-Fa-f]+:.*$", l) for l in dis.splitlines()) if m]
                # _Z<len><name>[ILi<N>EE...]: the kernels are plain functions or templates over one int (no tool needed to read that)
                m = re.match(r"^_Z(\d+)", sym); name = sym
                if m:
                    n = int(m.group(1)); rest = sym[m.end():]; name = rest[:n]
                    t = re.match(r"^ILi(\d+)EE", rest[n:])
                    if t:
                        name += "<%s>" % t.group(1)
                out[name] = hashlib.sha256("\n".join(ins).encode()).hexdigest()[:16]
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _newer(target, srcs):
    return (not os.path.exists(target)) or os.path.getmtime(target) < max(os.path.getmtime(s) for s in srcs)


HIP_UNITS = ["capi.hip", "k_generic.hip", "k_fast_0.hip", "k_fast_7.hip", "k_fast_1.hip", "k_fast_6.hip", "k_fast_2.hip", "k_fast_3.hip", "k_fast_4.hip", "k_fast_8.hip", "k_fast_9.hip", "k_fast_10.hip", "k_fast_11.hip"]
HOST_UNITS = ["host_tables.cpp", "host_piles.cpp", "host_io.cpp", "host_eprof.cpp"]


def _deps_newer(obj, dep):
    """True if `obj` is missing or older than any file its depfile (-MD) names."""
    if not os.path.exists(obj) or not os.path.exists(dep):
        return True
    t = os.path.getmtime(obj)
    txt = open(dep).read().replace("\\\n", " ")
    for f in txt.split(":", 1)[1].split():
        try:
            if os.path.getmtime(f) > t:
                return True
        except OSError:
            return True
    return False


def _build_objects(out, extra_flags, tag, force=False, verbose=False):
    """The library as one object per translation unit, compiled side by side: the window kernels are 150-260 KB of gfx950 code each
    (one k_window_fast<tier> per unit, csrc/window_kernels.hpp) and took 25 minutes one after the other in a single unit.  Objects and
    their dependency files live in daccord_amd/_obj/<tag>/ (git- and gpurun-ignored); a unit is recompiled when a file it includes
    changed.  Same flags and the same device code as the single-unit build (kernel_isa_hashes)."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    odir = os.path.join(_HERE, "_obj", tag)
    os.makedirs(odir, exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)
    stamp = os.path.join(odir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(cflags):
        force = True
    jobs = []
    for u in HIP_UNITS + HOST_UNITS:
        obj = os.path.join(odir, u + ".o"); dep = os.path.join(odir, u + ".d")
        if force or _deps_newer(obj, dep):
            jobs.append([hipcc] + cflags + ["-c", "-MD", "-MF", dep, "-o", obj, os.path.join(CSRC, u)])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    if jobs:
        nw = int(os.environ.get("DACC_BUILD_JOBS", "0")) or min(len(jobs), max(1, (os.cpu_count() or 1)))
        with ThreadPoolExecutor(nw) as ex:
            list(ex.map(run, jobs))
        open(stamp, "w").write(" ".join(cflags))
    objs = [os.path.join(odir, u + ".o") for u in HIP_UNITS + HOST_UNITS]
    if jobs or _newer(out, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out] + objs)
    return out


def build_hip(force=False, verbose=False):
    return _build_objects(LIB, [], "product", force, verbose)


def build_prof(force=False):
    """Profiling build of the same library (-DDACC_PROFILE: per-phase shader-cycle counters, scripts/prof_phases.py)."""
    return _build_objects(os.path.join(_HERE, "libdaccord_hip_prof.so"), ["-DDACC_PROFILE"], "prof", force)


def build_variant(name, flags, units=None, force=False):
    """Experiment variant daccord_amd/libvar_<name>.so (selected with DACC_LIB=<path>): the product's flags plus `flags`."""
    return _build_objects(os.path.join(_HERE, "libvar_%s.so" % name), list(flags), "var_" + name, force)


def build_io(force=False):
    """Host-only library with the .db / .las readers and writers (include/daccord_io.h) and the pile selection;
    the same objects are also linked into libdaccord_hip.so."""
    srcs = [os.path.join(CSRC, "host_io.cpp"), os.path.join(CSRC, "host_piles.cpp"), os.path.join(CSRC, "host_eprof.cpp"), os.path.join(CSRC, "host_check.cpp"),
            os.path.join(_HERE, "..", "include", "daccord_io.h"), os.path.join(_HERE, "..", "include", "daccord_hip.h")]
    if force or _newer(IOLIB, srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", IOLIB] + srcs[:4])
    return IOLIB


def build_cli(force=False):
    """daccord_hip: the daccord command line (C++ host program) on libdaccord_hip.so."""
    src = os.path.join(CSRC, "daccord_hip_main.cpp")
    if force or _newer(CLI, [src, LIB, os.path.join(_HERE, "..", "include", "daccord_io.h"), os.path.join(_HERE, "..", "include", "daccord_hip.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-o", CLI, src, "-L" + _HERE, "-ldaccord_hip",
                               "-Wl,-rpath,$ORIGIN", "-Wl,--allow-shlib-undefined"])
    return CLI


def build_all(force=False, verbose=False):
    from . import synth
    build_hip(force, verbose)
    build_io(force)
    build_cli(force)
    synth.build(force)
    return LIB
