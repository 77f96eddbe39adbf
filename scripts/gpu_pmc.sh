mkdir -p gpurun_out
( DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 14; DACC_TIERS=7 timeout 60 python scripts/dbg_tiers.py 8 ) > gpurun_out/dbg_tiers.log 2>&1
( DACC_LIB=$PWD/daccord_amd/libdaccord_hip_prof.so timeout 120 python scripts/prof_phases.py 64 ) > gpurun_out/phases.log 2>&1; echo "rc=$?" >> gpurun_out/phases.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  N=$(echo $C | cut -d' ' -f1)
  ( timeout 300 rocprofv3 --pmc $C --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_$N -o pmc -- python $R/bench.py --reads 1000 --steps 1 --warmup 0 --no-cpu ) > $R/gpurun_out/pmc_$N.log 2>&1; echo "rc=$?" >> $R/gpurun_out/pmc_$N.log
done
cd $R
grep -v amdgpu gpurun_out/dbg_tiers.log; grep -E "k=|stretchfeas|forward|instances|total cyc|enum cycles|combine" gpurun_out/phases.log; tail -n 1 gpurun_out/pmc_*.log; find gpurun_out/pmc_* -type f | head
