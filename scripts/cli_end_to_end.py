"""End-to-end run of the C++ front end on the files of BASELINE config 2 (VERDICT r02 task 5): writes reads.db / reads.las of
the bench's synthetic data set, runs `daccord_hip` on them (load -> select -> plan -> GPU -> FASTA, overlapped), reports its
own end-to-end Mbase/s, the SHA-256 of its FASTA (= bench.py's parity.gpu_fasta_sha256_all for the same set) and the peak
resident memory of a `-J0,8` run (the indexed .las reader reads only that part's byte range).
usage: python scripts/cli_end_to_end.py [reads=10000] [workdir=/tmp/dacc_e2e]"""
import hashlib, os, re, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from daccord_amd import io as dio
from daccord_amd.synth import SynthData
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
wd = sys.argv[2] if len(sys.argv) > 2 else "/tmp/dacc_e2e"
os.makedirs(wd, exist_ok=True)
t0 = time.time()
d = SynthData(int(reads * 10000 / 20.0), reads, 10000, seed=3)
db, las = os.path.join(wd, "reads.db"), os.path.join(wd, "reads.las")
dio.write_db(db, d.bps, d.boff, d.rlen); dio.write_las(las, 100, d.ovl, d.trace)
for f in (las + ".daidx", las + ".eprof"):
    if os.path.exists(f):
        os.remove(f)
print("files written in %.1f s: las %.1f MB, bps %.1f MB" % (time.time() - t0, os.path.getsize(las) / 1e6, len(d.bps) / 1e6), flush=True)
exe = os.path.join(ROOT, "daccord_amd", "daccord_hip")
env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "daccord_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
pi, pd, cor = d.error_profile()
args = [exe, "-k14", "-V1", "--eprof%.17g,%.17g,%.17g" % (pi, pd, cor)]


def run(extra, tag):
    t = time.time()
    p = subprocess.Popen(args + extra + [las, db], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    h = hashlib.sha256(); n = 0
    while True:
        b = p.stdout.read(1 << 22)
        if not b:
            break
        h.update(b); n += len(b)
    err = p.stderr.read().decode(); p.wait()
    dt = time.time() - t
    m = re.search(r"\[V\] (\d+) corrected bases in ([\d.]+) s end to end.*= ([\d.]+) Mbase/s; (\d+) batches, ([\d.]+) s in dacc_submit_piles", err)
    print("%s: rc=%d wall %.2f s, FASTA %d bytes sha256 %s" % (tag, p.returncode, dt, n, h.hexdigest()), flush=True)
    if m:
        print("   front end: %s bases, %.2f s end to end = %s Mbase/s, %s batches, %s s inside dacc_submit_piles" % (m.group(1), float(m.group(2)), m.group(3), m.group(4), m.group(5)))
    else:
        print(err[-800:])
    return h.hexdigest()


run([], "whole file (first run scans the .las and writes the sidecar index)")
run([], "whole file (second run loads the sidecar index)")
run(["--batch5000"], "whole file, 5000 A reads per batch")
# two device workers (two contexts; on a box with one GPU both on it: one batch's host plan overlaps the other's kernels)
run(["--gpus2"], "whole file, --gpus2")
run(["--gpus2", "--batch1000"], "whole file, --gpus2, 1000 A reads per batch")
# peak RSS of one of eight -J parts (rusage of that child alone)
pp = subprocess.Popen(args + ["-J0,8", las, db], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
_, _, ru = os.wait4(pp.pid, 0)
print("-J0,8: peak RSS %.1f MB (las %.1f MB, read store %.1f MB; the HIP runtime itself maps several hundred MB)" % (ru.ru_maxrss / 1024.0, os.path.getsize(las) / 1e6, len(d.bps) / 1e6))
# the same run without a given profile: the front end estimates it from the first 1024 piles on the host threads first
# (src/daccord.cpp:1653-1878), writes <las>.eprof and corrects with it
if os.path.exists(las + ".eprof"):
    os.remove(las + ".eprof")
t = time.time()
p = subprocess.run([exe, "-k14", "-V1", las, db], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
err = p.stderr.decode()
m = re.search(r"error profile estimated on (\d+) host threads in ([\d.]+) s", err)
m2 = re.search(r"\[V\] (\d+) corrected bases in ([\d.]+) s end to end", err)
print("no profile given: rc=%d wall %.2f s; estimator %s s on %s host threads; correction %s s end to end"
      % (p.returncode, time.time() - t, m.group(2) if m else "?", m.group(1) if m else "?", m2.group(2) if m2 else "?"))
