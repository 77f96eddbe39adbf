# Round 3, GPU call 1: (1) everything written after round 2's last GPU run + the reachability prune through the GPU tests,
# (2) default bench with / without the prune, (3) occupancy experiment: the same easy workload (8x coverage) on tier 1 as it
# is (3 wavefronts per CU), with the model table read from global memory, and with lean capacities (6 per CU),
# (4) phase profiles, (5) rocprof kernel stats.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $R/$O; cd $R
[ -f probe_in/cases.txt ] && bash scripts/gpu_probe.sh > $O/probe.log 2>&1
( timeout 900 python -m pytest tests -x -q -m gpu -rs --durations=15 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
( timeout 420 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
( DACC_LIB=$R/daccord_amd/libvar_noreach.so timeout 200 python bench.py --steps 1 --warmup 0 --no-cpu ) > $O/bench_noreach.log 2>&1
for V in libdaccord_hip libvar_gtab libvar_lean; do
  ( DACC_LIB=$R/daccord_amd/$V.so timeout 150 python bench.py --coverage 8 --reads 4000 --steps 2 --warmup 1 --no-cpu ) > $O/occ_cov8_$V.log 2>&1
done
( DACC_LIB=$R/daccord_amd/libvar_gtab.so timeout 200 python bench.py --steps 1 --warmup 0 --no-cpu ) > $O/bench_gtab.log 2>&1
for V in libdaccord_hip_prof libvar_noreach_prof; do
  [ -f daccord_amd/$V.so ] && ( DACC_LIB=$R/daccord_amd/$V.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases_$V.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
( timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $R/$O/stats.log 2>&1; echo "rc=$?" >> $R/$O/stats.log
cd $R
tail -n 6 $O/pytest_gpu.log; tail -n 2 $O/smoke.log
for f in $O/bench_default.log $O/bench_noreach.log $O/bench_gtab.log $O/occ_cov8_*.log; do echo "== $f"; tail -n 1 $f | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['config']['windows_rank0'], r['parity'].get('identical'), r['parity']['gpu_fasta_sha256_all'][:16])
except Exception as e:
    print('no json', e)
"; done
for f in $O/phases_*.log; do echo $f; grep -v amdgpu $f | grep "total cyc\|tiers_ms" | tail -2; done
head -8 $O/stats/st_kernel_stats.csv 2>/dev/null
