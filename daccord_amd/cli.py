"""Thin Python handle on the C++ front end `daccord_amd/daccord_hip` (daccord_amd/csrc/daccord_hip_main.cpp):

    python -m daccord_amd.cli [options] reads.las reads.db [reads2.db]

is the same as running the binary; all option handling (src/daccord.cpp:185-207, 1282-1305), the read interval logic,
the error profile estimation and the ordered FASTA output live there.  Kept so that Python callers (tests) can capture
the output."""
import subprocess
import sys


def binary():
    from . import build
    build.build_hip()
    return build.build_cli()


def run(argv, **kw):
    """Run daccord_hip; returns the CompletedProcess (stdout = FASTA, stderr = log)."""
    return subprocess.run([binary()] + list(argv), stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def main(argv=None, out=None):
    r = run(sys.argv[1:] if argv is None else argv)
    (out or sys.stdout).write(r.stdout.decode())
    sys.stderr.write(r.stderr.decode())
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
