mkdir -p gpurun_out
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 600 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
cat gpurun_out/phases.log | grep -v amdgpu.ids
