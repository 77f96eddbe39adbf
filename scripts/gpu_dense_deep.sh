# the deep batches' dense tier (k_window_fast<11>) on the GPU box: parity subset, the deep scale cases, then the 54x shape with / without the tier in one process
R=$GRAFT_REPO_ROOT; TAG=${1:?tag}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_steps.sh $TAG quick
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_fuzz_wide.py -x -q -m gpu -k "dense or cfg4 or deep or wide_random" ) > $O/pytest_deep.log 2>&1; echo "pytest rc=$?" >> $O/pytest_deep.log; tail -n 3 $O/pytest_deep.log
( SWEEP_COVERAGE=54 timeout 500 python scripts/sweep_env.py 2000 3 "" "DACC_DENSE_TIER=0" "" ) > $O/sweep_54x_2000.log 2>&1; echo "54x, 2000 reads"; grep '^{' $O/sweep_54x_2000.log | cut -c1-400
( timeout 300 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu --live-parity 8 ) > $O/bench_54x_2000piles.log 2>&1; grep '^{' $O/bench_54x_2000piles.log | tail -n 1 | python scripts/bench_brief.py
true
