R=$GRAFT_REPO_ROOT; TAG=r05t; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do for lib in libvar_step6.so libdaccord_hip.so; do
( DACC_LIB=$R/daccord_amd/$lib SWEEP_COVERAGE=54 timeout 200 python scripts/sweep_env.py 600 3 "" ) > $O/ab54_${lib}_$rep.log 2>&1; echo "54x $lib rep$rep: $(grep '^{' $O/ab54_${lib}_$rep.log | tail -n 1 | cut -c1-260)"
done; done
( timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "cfg4 or cfg5 or cfg1k8" ) > $O/pytest_scale_deep.log 2>&1; tail -n 3 $O/pytest_scale_deep.log
