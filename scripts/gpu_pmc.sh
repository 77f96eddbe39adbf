# PMC passes (separate rocprofv3 runs per counter group, kernel-trace only -- never combined with other trace domains)
#   at the headline workload:      FETCH_SIZE | WRITE_SIZE | SQ issue counters          -> traffic + issue fractions of the bench line
#   at a smaller slice (ratios):   lane utilisation + in-flight levels | LDS / scalar activity | TCC hit / miss | TCP->TCC latency
# -> profiles/<tag>_pmc_summary.json (scripts/pmc_summarize.py; keyed by the hash of the device sources)
# usage: bash scripts/gpu_pmc.sh <tag> [reads=10000] [diag_reads=3000]
R=$GRAFT_REPO_ROOT; TAG=${1:?usage: gpu_pmc.sh <tag> [reads] [diag_reads]}; READS=${2:-10000}; DREADS=${3:-3000}; O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_group () {  # name reads counters...
  N=$1; RD=$2; shift 2
  ( timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $R/bench.py --reads $RD --steps 1 --warmup 0 --no-cpu ) > $O/pmc_$N.log 2>&1; echo "rc=$?" >> $O/pmc_$N.log
  f=$(find $O/pmc_$N -name "*counter_collection.csv" | head -1); [ -n "$f" ] && mv $f $O/pmc_$N/pmc_counter_collection.csv
  find $O/pmc_$N -name "*.csv" ! -name "pmc_counter_collection.csv" -delete; find $O/pmc_$N -type d -empty -delete
}
run_group FETCH_SIZE $READS FETCH_SIZE
run_group WRITE_SIZE $READS WRITE_SIZE
run_group SQ_WAVE_CYCLES $READS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY
run_group LANES $DREADS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS
run_group ACTIVITY $DREADS SQ_WAVE_CYCLES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_group TCC $DREADS TCC_HIT_sum TCC_MISS_sum
run_group TCP $DREADS TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
cd $R
python scripts/pmc_summarize.py $O $READS 10000 20.0 14 2.4 $TAG $DREADS | tail -60
