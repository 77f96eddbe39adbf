# Short form of scripts/gpu_evidence.sh (~6 GPU minutes): a parity subset, the default bench line with the CPU legs,
# PMC passes + diagnostics, the bench line again quoting them, rocprof kernel stats of the same command
R=$GRAFT_REPO_ROOT; TAG=${1:-r04q}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "size_class or windows_and_fragments or rerun or reference_build or golden" ) > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log; tail -n 3 $O/pytest_subset.log
( timeout 400 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
bash scripts/gpu_pmc.sh $TAG 10000 3000 > $O/pmc.log 2>&1
cp profiles/${TAG}_pmc_summary.json $O/ 2>/dev/null
mkdir -p $O/pmc_raw; for d in gpurun_out/pmc_$TAG/pmc_*; do [ -f $d/pmc_counter_collection.csv ] && gzip -c $d/pmc_counter_collection.csv > $O/pmc_raw/$(basename $d).csv.gz; done
( timeout 200 python bench.py --no-cpu ) > $O/bench_default_with_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $O/stats.log 2>&1; echo "rc=$?" >> $O/stats.log
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats.csv \; ; find $O/stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
cd $R
for f in $O/bench_default.log $O/bench_default_with_pmc.log; do grep '^{' $f | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); ro = r['roofline']
print(r['value'], r['value_incl_plan_h2d'], r['ms_per_step'], r['setup_s'], ro['kernel_ms'], r['parity'].get('identical'), r['parity'].get('piles_compared'), ro.get('traffic'), str(ro.get('pmc_source'))[:40])
cb = r.get('cpu_baseline')
if cb: print('  cpu', cb['value'], (cb.get('reference_build') or {}).get('value'), (cb.get('reference_build') or {}).get('identical_to_gpu_on_sample'))
"; done
head -4 $O/rocprof_kernel_stats.csv | cut -c1-160
