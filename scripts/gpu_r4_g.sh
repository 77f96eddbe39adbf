# Round 4, GPU call 7: 64 pairs per round (PSIQ 10, 8 recorded pops per pair) against 940.8 ms at 3000 reads; core parity tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O; cd $R
( timeout 300 python scripts/sweep_env.py 3000 4 "" ) > $O/sweep.log 2>&1
grep -h '^{' $O/sweep.log | cut -c1-220; tail -n 2 $O/sweep.log | cut -c1-200
( timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu --durations=3 ) > $O/pytest_core.log 2>&1; echo "pytest rc=$?" >> $O/pytest_core.log
tail -n 5 $O/pytest_core.log
