# Round 4: the remaining second-pass checks (config 2 strata, fifty random sets)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R
( timeout 330 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "cfg2 or fifty" --durations=4 ) > $O/pytest_second_pass2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_second_pass2.log; tail -n 8 $O/pytest_second_pass2.log
