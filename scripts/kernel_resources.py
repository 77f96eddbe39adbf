"""Registers / scratch / LDS of every kernel in the built library (from the metadata notes of its gfx950 code objects: one per
translation unit that holds kernels, csrc/window_kernels.hpp).
usage: python scripts/kernel_resources.py [lib.so]"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"


def code_objects(lib):
    """gfx950 code objects inside the clang offload bundles of a host library, as bytes."""
    b = open(lib, "rb").read()
    out = []
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while i >= 0:
        n = struct.unpack_from("<Q", b, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, p); p += 24
            t = b[p:p + tl].decode(); p += tl
            if "gfx950" in t and size:
                out.append(b[i + off:i + off + size])
        i = b.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)
    return out


def resources(lib):
    rows = []
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            out = subprocess.check_output([LLVM + "llvm-readelf", "--notes", f.name]).decode()
            sz = subprocess.check_output([LLVM + "llvm-readelf", "-s", "--wide", f.name]).decode()
        size = {}
        for line in sz.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC":
                size[f[7]] = int(f[2])
        cur = {}
        for line in out.splitlines():
            s = line.strip().lstrip("- ")
            for key in (".agpr_count", ".name", ".private_segment_fixed_size", ".sgpr_count", ".vgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size"):
                if s.startswith(key + ":"):
                    cur[key] = s.split(":", 1)[1].strip()
            if s.startswith(".vgpr_spill_count"):
                cur["size"] = size.get(cur.get(".name"), "")
                rows.append(cur); cur = {}
    return rows


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "daccord_amd", "libdaccord_hip.so")
    print("%-60s %5s %5s %5s %6s %8s %6s %9s" % ("kernel", "vgpr", "agpr", "sgpr", "sspill", "scratch", "spill", "code B"))
    for r in resources(lib):
        nm = r.get(".name", "?")
        try:
            dn = subprocess.check_output([LLVM + "llvm-cxxfilt", nm]).decode().strip().split("(")[0]
        except Exception:
            dn = nm
        print("%-60s %5s %5s %5s %6s %8s %6s %9s" % (dn[:60], r.get(".vgpr_count"), r.get(".agpr_count"), r.get(".sgpr_count"), r.get(".sgpr_spill_count"), r.get(".private_segment_fixed_size"), r.get(".vgpr_spill_count"), r.get("size")))
