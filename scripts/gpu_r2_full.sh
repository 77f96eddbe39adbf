# everything the driver runs at round end + the profiles to commit
T=${1:-r2full}
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/${T}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/${T}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_smoke.log
( timeout 1200 python bench.py ) > gpurun_out/${T}_bench_default.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_bench_default.log
tail -n 6 gpurun_out/${T}_pytest_gpu.log; tail -n 2 gpurun_out/${T}_smoke.log; tail -n 2 gpurun_out/${T}_bench_default.log | cut -c1-7000
