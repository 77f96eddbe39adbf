mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "not fifty" ) > gpurun_out/r2b_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_parity.log
( timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -s ) > gpurun_out/r2b_scale.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b_scale.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > gpurun_out/r2b_phases.log 2>&1; echo "rc=$?" >> gpurun_out/r2b_phases.log
tail -n 5 gpurun_out/r2b_parity.log; grep -v amdgpu.ids gpurun_out/r2b_scale.log | tail -n 25; grep -v amdgpu.ids gpurun_out/r2b_phases.log
