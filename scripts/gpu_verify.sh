mkdir -p gpurun_out
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
( timeout 400 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 3 gpurun_out/smoke.log; tail -n 4 gpurun_out/pytest_gpu.log
