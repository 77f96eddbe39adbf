# Round 3, GPU call 8 (the round's final build: device code at -Os, deep tier 4 in the gw layout): full GPU suite, smoke, default
# bench with the CPU legs, 54x and ONT bench lines, rocprof kernel stats and the PMC passes of the default workload, phase profile.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3h; mkdir -p $R/$O; cd $R
( timeout 900 python -m pytest tests -x -q -m gpu -rs --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 13 $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
( timeout 480 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
( timeout 300 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_54x_2000piles.log 2>&1
( timeout 200 python bench.py --ont --reads 4000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_ont_4000piles.log 2>&1
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases.log 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $R/$O/stats.log 2>&1; echo "rc=$?" >> $R/$O/stats.log
cd $R
bash scripts/gpu_pmc.sh r03h 10000 > $O/pmc.log 2>&1
cp profiles/r03h_pmc_summary.json $O/ 2>/dev/null
# the bench line again, now that the PMC summary of this very build exists: roofline.traffic is filled in
( timeout 300 python bench.py --no-cpu ) > $O/bench_default_with_pmc.log 2>&1
for f in $O/bench_default.log $O/bench_default_with_pmc.log $O/bench_54x_2000piles.log $O/bench_ont_4000piles.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['config']['windows_rank0'], r['parity'].get('identical'), r['parity'].get('piles_compared'), r['parity']['gpu_fasta_sha256_all'][:16], r['accuracy'].get('erate'), r['roofline'].get('traffic'), r['roofline'].get('pmc_source'))
except Exception as e:
    print('no json', e)
"; done
head -8 $O/stats/st_kernel_stats.csv 2>/dev/null
tail -n 4 $O/pmc.log | cut -c1-200
true
