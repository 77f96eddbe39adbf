R=$GRAFT_REPO_ROOT; TAG=r05g; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_lines.so default
( SWEEP_COVERAGE=54 timeout 200 python scripts/sweep_env.py 600 3 "" ) > $O/ab54_default.log 2>&1; echo "54x default: $(grep '^{' $O/ab54_default.log | tail -n 1 | cut -c1-260)"
