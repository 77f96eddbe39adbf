R=$GRAFT_REPO_ROOT; TAG=r05ad; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 600 python scripts/sweep_env.py 3000 3 "" "DACC_T0INST=488" "DACC_T0INST=494" "DACC_T0INST=506" "DACC_T0INST=488" "" ) > $O/sweep_t0inst.log 2>&1; grep '^{' $O/sweep_t0inst.log | cut -c1-230
