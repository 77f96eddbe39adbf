# Evidence on the current build (`gpurun --timeout 2400 -- 'bash scripts/gpu_evidence.sh <tag>'` -> gpurun_out/<tag>/, ~12 GPU minutes; the tag is mandatory): full GPU suite, smoke,
# default bench with the CPU legs, PMC passes + diagnostics, the bench line again quoting them, rocprof kernel stats of the same command,
# 54x / ONT bench lines with a live parity sample each
R=$GRAFT_REPO_ROOT; TAG=${1:?usage: gpu_evidence.sh <tag>}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests -x -q -m gpu -rs --durations=6 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 12 $O/pytest_gpu.log
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
( timeout 600 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
bash scripts/gpu_pmc.sh $TAG 10000 3000 > $O/pmc.log 2>&1
cp profiles/${TAG}_pmc_summary.json $O/ 2>/dev/null
mkdir -p $O/pmc_raw; for d in gpurun_out/pmc_$TAG/pmc_*; do [ -f $d/pmc_counter_collection.csv ] && gzip -c $d/pmc_counter_collection.csv > $O/pmc_raw/$(basename $d).csv.gz; done
( timeout 200 python bench.py --no-cpu ) > $O/bench_default_with_pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --steps 1 --warmup 0 --no-cpu ) > $O/stats.log 2>&1; echo "rc=$?" >> $O/stats.log
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats.csv \; ; find $O/stats -name "*.csv" ! -name "*kernel_stats.csv" -delete
cd $R
( timeout 300 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu --live-parity 8 ) > $O/bench_54x_2000piles.log 2>&1
( timeout 300 python bench.py --ont --reads 4000 --steps 2 --warmup 1 --no-cpu --live-parity 16 ) > $O/bench_ont_4000piles.log 2>&1
for f in $O/bench_default.log $O/bench_default_with_pmc.log $O/bench_54x_2000piles.log $O/bench_ont_4000piles.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    ro = r['roofline']
    print(r['value'], r['value_incl_plan_h2d'], r['ms_per_step'], ro['kernel_ms'], ro['windows_handed_on'], ro['size_classes'], r['parity'].get('identical'), r['parity'].get('piles_compared_distinct'), r['parity'].get('live'), r['parity']['gpu_fasta_sha256_all'][:16])
    print('  roofline', ro['kernel'], ro['achieved'], ro['frac'], ro['frac_step'], ro.get('traffic'), ro.get('valu_issue_util'), ro.get('lane_op_frac'), ro.get('valu_lane_util'), ro.get('inflight_share'), ro.get('resident_waves_per_cu_by_kernel'), str(ro.get('pmc_source'))[:50])
    cb = r.get('cpu_baseline')
    if cb: print('  cpu', cb.get('kind'), cb['value'], cb['cores'], cb.get('identical_to_gpu_on_sample'), (cb.get('port') or {}).get('value'), (cb.get('like_for_like') or {}).get('value'))
except Exception as e:
    print('no json', e)
"; done
head -8 $O/rocprof_kernel_stats.csv 2>/dev/null | cut -c1-200
true
