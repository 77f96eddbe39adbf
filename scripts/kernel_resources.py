"""Registers / scratch / LDS of every kernel in the built library (from the gfx950 code object's metadata notes).
usage: python scripts/kernel_resources.py [lib.so]"""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "daccord_amd", "libdaccord_hip.so")
b = open(lib, "rb").read()
i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", b, i + 24)[0]
p = i + 32
co = None
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", b, p); p += 24
    t = b[p:p + tl].decode(); p += tl
    if "gfx950" in t:
        co = b[i + off:i + off + size]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co); f.flush()
    out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name]).decode()
    sz = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "--wide", f.name]).decode()
cur = {}
rows = []
for line in out.splitlines():
    s = line.strip().lstrip("- ")
    for key in (".agpr_count", ".name", ".private_segment_fixed_size", ".sgpr_count", ".vgpr_count", ".vgpr_spill_count", ".group_segment_fixed_size"):
        if s.startswith(key + ":"):
            cur[key] = s.split(":", 1)[1].strip()
    if s.startswith(".vgpr_spill_count"):
        rows.append(cur); cur = {}
size = {}
for line in sz.splitlines():
    f = line.split()
    if len(f) >= 8 and f[3] == "FUNC":
        size[f[7]] = int(f[2])
print("%-60s %5s %5s %5s %8s %6s %9s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "spill", "code B"))
for r in rows:
    nm = r.get(".name", "?")
    try:
        dn = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", nm]).decode().strip().split("(")[0]
    except Exception:
        dn = nm
    print("%-60s %5s %5s %5s %8s %6s %9s" % (dn[:60], r.get(".vgpr_count"), r.get(".agpr_count"), r.get(".sgpr_count"), r.get(".private_segment_fixed_size"), r.get(".vgpr_spill_count"), size.get(nm, "")))
