mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "windows_and_fragments or golden or high_error" ) > gpurun_out/r2d_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d_parity.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > gpurun_out/r2d_phases.log 2>&1; echo "rc=$?" >> gpurun_out/r2d_phases.log
( timeout 600 python bench.py --reads 2000 --steps 2 --warmup 1 --no-cpu ) > gpurun_out/r2d_bench2000.log 2>&1; echo "rc=$?" >> gpurun_out/r2d_bench2000.log
tail -n 3 gpurun_out/r2d_parity.log; grep -v amdgpu.ids gpurun_out/r2d_phases.log | grep -v "0 cyc/window"; grep -v amdgpu.ids gpurun_out/r2d_bench2000.log | tail -3
