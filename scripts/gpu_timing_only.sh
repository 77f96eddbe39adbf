mkdir -p gpurun_out
( DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 14; DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 8 ) > gpurun_out/dbg_tiers.log 2>&1
grep -v amdgpu gpurun_out/dbg_tiers.log
