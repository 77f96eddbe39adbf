mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -s ) > gpurun_out/r2f_scale.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f_scale.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s ) > gpurun_out/r2f_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f_parity.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > gpurun_out/r2f_phases.log 2>&1; echo "rc=$?" >> gpurun_out/r2f_phases.log
( timeout 600 python bench.py --reads 2000 --steps 2 --warmup 1 --no-cpu ) > gpurun_out/r2f_bench2000.log 2>&1; echo "rc=$?" >> gpurun_out/r2f_bench2000.log
grep -v amdgpu.ids gpurun_out/r2f_scale.log | tail -n 16; tail -n 5 gpurun_out/r2f_parity.log; grep -v amdgpu.ids gpurun_out/r2f_phases.log | grep "window=\|total"; tail -2 gpurun_out/r2f_bench2000.log
