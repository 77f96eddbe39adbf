# Round 4, GPU call 6: profiling build with the sequence walks of offerCandidate timed (is buildSeq worth a lane-parallel pre-pass?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O; cd $R
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > $O/phases.log 2>&1; grep -A32 "^k=14" $O/phases.log | cut -c1-120
