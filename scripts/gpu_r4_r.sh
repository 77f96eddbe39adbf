# Round 4: the fuzz / scale tests with their new second-pass checks (hand-over active on the second use of a context)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O; cd $R
( timeout 420 python -m pytest tests/test_gpu_fuzz_wide.py tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "wide or cfg4 or cfg5 or cfg1k8 or cfg3 or (random_parameter_sets and not fifty)" --durations=4 ) > $O/pytest_second_pass.log 2>&1; echo "pytest rc=$?" >> $O/pytest_second_pass.log; tail -n 8 $O/pytest_second_pass.log
