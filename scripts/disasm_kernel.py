"""Disassembly of one kernel of the built library (llvm-objdump over the gfx950 code object that holds it), instructions only.
usage: python scripts/disasm_kernel.py <kernel substring, e.g. 'k_window_fastILi0'> [lib.so] [--lines]   (--lines: with source lines, needs a -gline-tables-only build)"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects, LLVM, ROOT


def disasm(sub, lib=None, lines=False):
    lib = lib or os.path.join(ROOT, "daccord_amd", "libdaccord_hip.so")
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            syms = [l.split()[-1] for l in subprocess.check_output([LLVM + "llvm-objdump", "-t", f.name]).decode().splitlines() if " F .text" in l]
            for s in syms:
                if sub in s:
                    cmd = [LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + s, f.name]
                    if lines:
                        cmd.insert(2, "-l")
                    return subprocess.check_output(cmd).decode()
    return ""


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(disasm(args[0], args[1] if len(args) > 1 else None, "--lines" in sys.argv))
