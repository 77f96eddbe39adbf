mkdir -p gpurun_out
( DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 14; DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 8 ) > gpurun_out/dbg_tiers.log 2>&1
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 120 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
grep -v amdgpu gpurun_out/dbg_tiers.log; grep -E "k=|stretchfeas|forward|instances|stretches|total cyc|enum cycles|combine|rc=" gpurun_out/phases.log
