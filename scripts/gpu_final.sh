mkdir -p gpurun_out
nproc > gpurun_out/nproc.txt
( timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/bench_default.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -o bench -- python $R/bench.py --no-cpu ) > $R/gpurun_out/rocprof_default.log 2>&1; echo "rc=$?" >> $R/gpurun_out/rocprof_default.log
cd $R
tail -n 2 gpurun_out/bench_default.log; tail -n 1 gpurun_out/rocprof_default.log; head -8 gpurun_out/prof_default/bench_kernel_stats.csv
