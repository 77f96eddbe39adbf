# PMC passes (separate rocprofv3 runs per counter group, kernel-trace only) at the headline workload -> profiles/<tag>_pmc_summary.json
# usage: bash scripts/gpu_pmc.sh <tag> [reads=10000]
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; READS=${2:-10000}; O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  ( timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu ) > $O/pmc_$N.log 2>&1; echo "rc=$?" >> $O/pmc_$N.log
done
cd $R
# the summariser expects <root>/pmc_<GROUP>/pmc_counter_collection.csv
for N in FETCH_SIZE WRITE_SIZE SQ_WAVE_CYCLES; do f=$(find $O/pmc_$N -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_$N/pmc_counter_collection.csv; done
python scripts/pmc_summarize.py $O $READS 10000 20.0 14 2.4 $TAG | tail -40
