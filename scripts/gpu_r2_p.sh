T=${1:-q}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -x -q -m gpu -k "not random and not fuzz and not scale" ) > gpurun_out/${T}_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_parity.log
( timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -s -k "cfg5 or cfg1k8" ) > gpurun_out/${T}_scale.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_scale.log
( timeout 900 python bench.py --no-cpu ) > gpurun_out/${T}_bench.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_bench.log
tail -n 3 gpurun_out/${T}_parity.log; grep -v amdgpu.ids gpurun_out/${T}_scale.log | tail -n 11; tail -2 gpurun_out/${T}_bench.log | grep -o '"value": [0-9.]*\|"kernel_ms": {[^}]*}\|windows_handed_on": {[^}]*}\|"identical": [a-z]*'
