mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 600 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
( timeout 900 python bench.py --reads 1000 --steps 2 --warmup 1 ) > gpurun_out/bench_1000.log 2>&1; echo "rc=$?" >> gpurun_out/bench_1000.log
tail -4 gpurun_out/pytest_gpu.log; cat gpurun_out/phases.log | grep -v amdgpu.ids; tail -3 gpurun_out/bench_1000.log
