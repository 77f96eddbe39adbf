R=$GRAFT_REPO_ROOT; TAG=r05ai; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_ab.sh $TAG 3000 3 daccord_amd/libvar_step9.so default
( timeout 400 python scripts/sweep_env.py 3000 2 "DACC_T0INST=520" "DACC_T0INST=540" "DACC_T0INST=560" "DACC_T0INST=576" ) > $O/sweep_t0inst.log 2>&1; grep '^{' $O/sweep_t0inst.log | cut -c1-230
