mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -s ) > gpurun_out/r2_scale.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_scale.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s ) > gpurun_out/r2_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_parity.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > gpurun_out/r2_phases.log 2>&1; echo "rc=$?" >> gpurun_out/r2_phases.log
grep -v amdgpu.ids gpurun_out/r2_scale.log | tail -n 25; tail -n 5 gpurun_out/r2_parity.log; grep -v amdgpu.ids gpurun_out/r2_phases.log
