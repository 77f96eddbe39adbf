mkdir -p gpurun_out
( DACC_SCHED=3 DACC_NOFAST=1 timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k test_windows_and_fragments ) > gpurun_out/dbg_generic_dyn.log 2>&1; echo "generic dyn rc=$?" >> gpurun_out/dbg_generic_dyn.log
( DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 14; DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 8 ) > gpurun_out/dbg_tiers.log 2>&1
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 120 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
tail -n 2 gpurun_out/dbg_generic_dyn.log; grep -v amdgpu gpurun_out/dbg_tiers.log; cat gpurun_out/phases.log | grep -v amdgpu.ids
