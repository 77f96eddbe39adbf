# Round 4: the C++ front end end to end on the config-2 files (final build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O; cd $R
( timeout 500 python scripts/cli_end_to_end.py ) > $O/cli_end_to_end.log 2>&1; echo "rc=$?" >> $O/cli_end_to_end.log; tail -n 12 $O/cli_end_to_end.log | cut -c1-220
