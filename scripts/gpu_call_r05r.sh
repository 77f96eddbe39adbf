R=$GRAFT_REPO_ROOT; TAG=r05r; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 600 python scripts/sweep_env.py 3000 2 "" "DACC_T0INST=440" "DACC_T0INST=520" "DACC_T0INST=560" "DACC_T0INST=600" "DACC_T0INST=640" "" ) > $O/sweep_t0inst.log 2>&1; grep '^{' $O/sweep_t0inst.log | cut -c1-230
