mkdir -p gpurun_out
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --reads 500 --steps 2 --warmup 1 ) > gpurun_out/bench_small.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_small.log
tail -5 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench_small.log
