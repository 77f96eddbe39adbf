mkdir -p gpurun_out
( timeout 300 python scripts/dbg_retry.py 14 1000 ) > gpurun_out/r2e_retry.log 2>&1; echo "rc=$?" >> gpurun_out/r2e_retry.log
( timeout 600 python bench.py --reads 2000 --steps 2 --warmup 1 --no-cpu ) > gpurun_out/r2e_bench2000.log 2>&1; echo "rc=$?" >> gpurun_out/r2e_bench2000.log
grep -v amdgpu.ids gpurun_out/r2e_retry.log; grep -v amdgpu.ids gpurun_out/r2e_bench2000.log | tail -3
