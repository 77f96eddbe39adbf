mkdir -p gpurun_out
( timeout 400 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
( DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 14; DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 8 ) > gpurun_out/dbg_tiers.log 2>&1
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 120 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
tail -n 3 gpurun_out/pytest_gpu.log; grep -v amdgpu gpurun_out/dbg_tiers.log; grep -E "k=|stretchfeas|forward|stretches|total cyc" gpurun_out/phases.log
