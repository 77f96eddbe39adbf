"""Can the host side feed eight GPUs?  (VERDICT r04 task 5; no GPU needed.)
Writes the .db / .las of BASELINE config 2 (bench.py's data set) and runs `daccord_hip --loaderonly` on them: the loader thread (indexed
pread of the batch's byte range + top-D selection per pile) and --gpus N planner threads (what dacc_submit_piles does on the host before its
uploads).  One MI355X consumes 10 000 piles per 3.08 s = 3250 piles/s = 30 Mbase/s of corrected bases (bench.py), so a single loader thread has to
sustain 8 x that = 26 000 piles/s = 260 Mbase/s of A reads for one process with --gpus8; the other route is one process per GPU (-J g,8), each
with its own loader over its own byte range of the .las.
usage: python scripts/host_feed_rate.py [reads=10000] [workdir=/tmp/dacc_feed]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from daccord_amd import io as dio
from daccord_amd.synth import SynthData
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
wd = sys.argv[2] if len(sys.argv) > 2 else "/tmp/dacc_feed"
os.makedirs(wd, exist_ok=True)
t0 = time.time()
d = SynthData(int(reads * 10000 / 20.0), reads, 10000, seed=3)
db, las = os.path.join(wd, "reads.db"), os.path.join(wd, "reads.las")
dio.write_db(db, d.bps, d.boff, d.rlen); dio.write_las(las, 100, d.ovl, d.trace)
print("files written in %.1f s: las %.1f MB, bps %.1f MB, host has %d CPUs" % (time.time() - t0, os.path.getsize(las) / 1e6, len(d.bps) / 1e6, os.cpu_count()), flush=True)
exe = os.path.join(ROOT, "daccord_amd", "daccord_hip")
env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "daccord_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
pi, pd, cor = d.error_profile()
base = [exe, "-k14", "-V0", "--eprof%.17g,%.17g,%.17g" % (pi, pd, cor), "--loaderonly"]
subprocess.run(base + [las, db], env=env, stderr=subprocess.DEVNULL)      # first run scans the .las and writes the sidecar index
for extra in (["--gpus1"], ["--gpus8"], ["--gpus8", "--batch500"], ["--gpus1", "-J0,8"]):
    t = time.time()
    p = subprocess.run(base + extra + [las, db], env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
    print("== %s  (rc %d, %.2f s wall)" % (" ".join(extra), p.returncode, time.time() - t))
    print(p.stderr.decode().strip())
