T=${1:-q}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q ) > gpurun_out/${T}_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_parity.log
( timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -s ) > gpurun_out/${T}_scale.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_scale.log
( timeout 600 python bench.py --reads 2000 --steps 2 --warmup 1 --no-cpu ) > gpurun_out/${T}_bench2000.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_bench2000.log
tail -n 3 gpurun_out/${T}_parity.log; grep -v amdgpu.ids gpurun_out/${T}_scale.log | tail -n 14; tail -2 gpurun_out/${T}_bench2000.log | grep -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}\|windows_handed_on": {[^}]*}'
