R=$GRAFT_REPO_ROOT; TAG=r05ak; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 500 python -X faulthandler bench.py ) > $O/bench_default_faulthandler.log 2>&1; echo "rc=$?" >> $O/bench_default_faulthandler.log
tail -n 40 $O/bench_default_faulthandler.log | cut -c1-400
