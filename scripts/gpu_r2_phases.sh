# per-phase cycles + 2000-pile bench only
T=${1:-q}
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "windows_and_fragments or golden or high_error" ) > gpurun_out/${T}_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_parity.log
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 200 python scripts/prof_phases.py 64 ) > gpurun_out/${T}_phases.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_phases.log
( timeout 600 python bench.py --reads 2000 --steps 2 --warmup 1 --no-cpu ) > gpurun_out/${T}_bench2000.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_bench2000.log
tail -n 3 gpurun_out/${T}_parity.log; grep -v amdgpu.ids gpurun_out/${T}_phases.log | grep -v " 0 cyc/window" | sed -n '/k=14/,$p'; tail -2 gpurun_out/${T}_bench2000.log | grep -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}\|windows_handed_on": {[^}]*}'
