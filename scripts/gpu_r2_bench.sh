# parity subset + bench at a given size with the CPU baseline legs
T=${1:-q}; R=${2:-2000}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q ) > gpurun_out/${T}_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_parity.log
( timeout 1500 python bench.py --reads $R --steps 2 --warmup 1 ) > gpurun_out/${T}_bench.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_bench.log
tail -n 3 gpurun_out/${T}_parity.log; tail -3 gpurun_out/${T}_bench.log | cut -c1-6000
