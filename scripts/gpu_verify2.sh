mkdir -p gpurun_out
( timeout 400 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( timeout 300 python scripts/dbg_toolong.py ) > gpurun_out/dbg_toolong.log 2>&1; echo "rc=$?" >> gpurun_out/dbg_toolong.log
tail -n 3 gpurun_out/pytest_gpu.log; grep -v amdgpu gpurun_out/dbg_toolong.log
