# Round 4, GPU call 5: A/B of the paired slab loads in the forward trees (this build) against call 3/4's numbers (940.8 ms at 3000 reads)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O; cd $R
( timeout 300 python scripts/sweep_env.py 3000 4 "" ) > $O/sweep.log 2>&1
grep -h '^{' $O/sweep.log | cut -c1-220; tail -n 2 $O/sweep.log | cut -c1-200
ls daccord_amd/libdaccord_hip_prof.so && ( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > $O/phases.log 2>&1; grep -A26 "^k=14" $O/phases.log | cut -c1-120
