R=$GRAFT_REPO_ROOT; TAG=r05al; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 200 python -m pytest tests/test_gpu_bench_ranks.py -x -q -m gpu -k "eight_ranks" ) > $O/pytest_eight_ranks.log 2>&1; echo "pytest rc=$?" >> $O/pytest_eight_ranks.log; tail -n 3 $O/pytest_eight_ranks.log
( timeout 400 python -X faulthandler bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
grep '^{' $O/bench_default.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); ro = r['roofline']; cb = r['cpu_baseline']
print(r['value'], r['ms_per_step'], r['parity']['identical'], r['parity']['piles_compared_distinct'], ro['kernel'], ro['frac'], ro['frac_step'], ro.get('traffic'), ro.get('valu_issue_util'), ro.get('lane_op_frac'), str(ro.get('pmc_source'))[:50])
print('cpu', cb.get('kind'), cb.get('value'), cb.get('cores'), cb.get('identical_to_gpu_on_sample'), (cb.get('port') or {}).get('value'), (cb.get('like_for_like') or {}).get('value'), cb.get('error'), r['post_loop_s'])
"; tail -n 2 $O/bench_default.log | cut -c1-200
