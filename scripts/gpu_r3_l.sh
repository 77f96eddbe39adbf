# Round 3, GPU call 15: end to end through the front end again, now also with two device workers on the one device
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3l; mkdir -p $R/$O; cd $R
( timeout 200 python scripts/cli_end_to_end.py 10000 /tmp/dacc_e2e ) > $O/cli_end_to_end.log 2>&1
grep -v amdgpu $O/cli_end_to_end.log
