"""Setup and post-loop cost of `bench.py --gpus 8 --scaling weak` on the host alone (VERDICT r04 task 5; no GPU): every rank generates all
80 000 reads (B reads are arbitrary) but only its own overlaps / piles, on ncpu // 8 threads, then selects its piles; rank 0 hashes the FASTA of all
80 000 reads after the loop.  Run as 8 concurrent processes, like the driver's torchrun.   usage: bench_setup_time.py [world=8] [rank]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) <= 2:
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, __file__, str(world), str(r)], stdout=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate()[0] for p in ps]
    for o in outs:
        print(o.strip())
    print("all %d ranks: %.1f s wall on %d CPUs" % (world, time.time() - t0, os.cpu_count()))
    sys.exit(0)
rank = int(sys.argv[2])
import resource
from daccord_amd import engine, shard
from daccord_amd.synth import SynthData
reads = 10000; total = reads * world; ncpu = os.cpu_count() or 1
t0 = time.time()
arange = shard.shard_range(0, total, rank, world)
d = SynthData(int(total * 10000 / 20.0), total, 10000, seed=3, nthreads=max(1, ncpu // world), aread_range=arange)
t1 = time.time()
ovl, piles = engine.pile_select(d.ovl, d.piles)
t2 = time.time()
print("rank %d: generate %.1f s (%d threads), pile_select %.1f s, %d piles, %d overlaps, peak RSS %.1f GB"
      % (rank, t1 - t0, max(1, ncpu // world), t2 - t1, len(piles), len(ovl), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0))
