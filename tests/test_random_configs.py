"""A few rounds of the randomized emulation-vs-oracle runs (scripts/fuzz_emul_vs_oracle.py) as part of the CPU suite."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_parameter_sets_agree_with_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_emul_vs_oracle.py"), "20260921", "6"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DONE bad=0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("OK  ") == 6
