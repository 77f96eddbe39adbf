R=$GRAFT_REPO_ROOT; TAG=r05ac; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_step7.so default
( timeout 300 python scripts/sweep_env.py 3000 2 "DACC_T0INST=476" "DACC_T0INST=500" "DACC_T0INST=512" ) > $O/sweep_t0inst.log 2>&1; grep '^{' $O/sweep_t0inst.log | cut -c1-230
