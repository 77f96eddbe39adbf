# Last GPU call of round 5: the default bench line with every committed oracle digest of the headline batch (piles_compared_distinct = 10000 when all
# twelve cfg2w parts are in), the GPU test over those parts, the 8-rank test with its live parity sample, the front end from files to FASTA
R=$GRAFT_REPO_ROOT; TAG=r05final; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_bench_ranks.py -x -q -m gpu -k "rest_of_headline or eight_ranks" ) > $O/pytest_new_tests.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new_tests.log; tail -n 4 $O/pytest_new_tests.log
( timeout 600 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
grep '^{' $O/bench_default.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); ro = r['roofline']; p = r['parity']
print(r['value'], r['ms_per_step'], r['value_incl_plan_h2d'], ro['kernel_ms'])
print('parity', p.get('identical'), p.get('piles_compared'), p.get('piles_compared_distinct'), [ (f['fixture'][-14:], f['identical']) for f in p.get('fixtures', []) ])
print('roofline', ro['frac'], ro['frac_step'], ro.get('traffic'), ro.get('valu_issue_util'), ro.get('lane_op_frac'), str(ro.get('pmc_source'))[:60])
cb = r['cpu_baseline']; print('cpu', cb['kind'], cb['value'], cb['cores'], cb['identical_to_gpu_on_sample'], cb['port']['value'], cb['like_for_like']['value'], r['post_loop_s'])
"
( timeout 600 python scripts/cli_end_to_end.py ) > $O/cli_end_to_end.log 2>&1; grep -E "front end|no profile|whole file" $O/cli_end_to_end.log | cut -c1-220
