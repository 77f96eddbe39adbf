mkdir -p gpurun_out
nproc > gpurun_out/nproc.txt
( timeout 900 python bench.py --reads 1000 --steps 2 --warmup 1 ) > gpurun_out/bench_1000.log 2>&1; echo "rc=$?" >> gpurun_out/bench_1000.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench1000 -- python $GRAFT_REPO_ROOT/bench.py --reads 1000 --steps 2 --warmup 1 --no-cpu ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof_1000.log 2>&1; echo "rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof_1000.log
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/bench_1000.log; tail -3 gpurun_out/rocprof_1000.log; ls -R gpurun_out/prof_r01 | head -20; cat gpurun_out/nproc.txt
