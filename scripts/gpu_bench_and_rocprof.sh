mkdir -p gpurun_out
( timeout 300 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --reads 1000 --steps 2 --warmup 1 ) > gpurun_out/bench_1000.log 2>&1; echo "rc=$?" >> gpurun_out/bench_1000.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -o bench1000 -- python $GRAFT_REPO_ROOT/bench.py --reads 1000 --steps 2 --warmup 1 --no-cpu ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof_1000.log 2>&1; echo "rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof_1000.log
cd $GRAFT_REPO_ROOT
tail -n 3 gpurun_out/pytest_gpu.log; tail -n 2 gpurun_out/bench_1000.log; tail -n 3 gpurun_out/rocprof_1000.log; find gpurun_out/prof_r01b -type f | head -20
