mkdir -p gpurun_out
( for cfg in "7 1" "3 1" "6 1"; do set -- $cfg; DACC_TIERS=$1 DACC_SCHED=$2 timeout 60 python scripts/dbg_tiers.py 8; done; DACC_TIERS=7 DACC_SCHED=1 timeout 60 python scripts/dbg_tiers.py 14 ) > gpurun_out/dbg_tiers.log 2>&1
grep -v amdgpu gpurun_out/dbg_tiers.log
