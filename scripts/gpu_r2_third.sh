mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "not fifty" ) > gpurun_out/r2c_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c_parity.log
( timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -s -k "cfg4 or cfg5 or cfg1k8" ) > gpurun_out/r2c_scale.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c_scale.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > gpurun_out/r2c_phases.log 2>&1; echo "rc=$?" >> gpurun_out/r2c_phases.log
tail -n 3 gpurun_out/r2c_parity.log; grep -v amdgpu.ids gpurun_out/r2c_scale.log | tail -n 14; grep -v amdgpu.ids gpurun_out/r2c_phases.log
