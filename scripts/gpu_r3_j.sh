# Round 3, GPU call 12 (final build of the round: device code at -Os with the max-ILP machine scheduler): full GPU suite, smoke,
# default bench with the CPU legs, PMC passes, the bench line again with roofline.traffic from them, rocprof kernel stats, 54x / ONT.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3j; mkdir -p $R/$O; cd $R
( timeout 420 python -m pytest tests -x -q -m gpu -rs --durations=5 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 10 $O/pytest_gpu.log
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
( timeout 240 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
bash scripts/gpu_pmc.sh r03j 10000 > $O/pmc.log 2>&1
cp profiles/r03j_pmc_summary.json $O/ 2>/dev/null
( timeout 120 python bench.py --no-cpu ) > $O/bench_default_with_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $R/$O/stats.log 2>&1; echo "rc=$?" >> $R/$O/stats.log
cd $R
( timeout 100 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_54x_2000piles.log 2>&1
( timeout 100 python bench.py --ont --reads 4000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_ont_4000piles.log 2>&1
for f in $O/bench_default.log $O/bench_default_with_pmc.log $O/bench_54x_2000piles.log $O/bench_ont_4000piles.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['parity'].get('identical'), r['parity'].get('piles_compared'), r['parity']['gpu_fasta_sha256_all'][:16], r['roofline'].get('traffic'), str(r['roofline'].get('pmc_source'))[:60])
except Exception as e:
    print('no json', e)
"; done
head -5 $O/stats/st_kernel_stats.csv 2>/dev/null
true
