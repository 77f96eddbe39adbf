mkdir -p gpurun_out
( timeout 300 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 120 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
tail -n 4 gpurun_out/pytest_gpu.log; cat gpurun_out/phases.log | grep -v amdgpu.ids
