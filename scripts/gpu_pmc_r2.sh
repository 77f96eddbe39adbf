# PMC passes (separate runs per counter group) + kernel-trace stats at the headline workload
R=$GRAFT_REPO_ROOT; READS=${1:-10000}
mkdir -p $R/gpurun_out/r2pmc
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2pmc/stats -o st -- python $R/bench.py --reads $READS --steps 2 --warmup 1 --no-cpu ) > $R/gpurun_out/r2pmc/stats.log 2>&1; echo "rc=$?" >> $R/gpurun_out/r2pmc/stats.log
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  N=$(echo $C | cut -d' ' -f1)
  ( timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r2pmc/pmc_$N -o pmc -- python $R/bench.py --reads $READS --steps 1 --warmup 0 --no-cpu ) > $R/gpurun_out/r2pmc/pmc_$N.log 2>&1; echo "rc=$?" >> $R/gpurun_out/r2pmc/pmc_$N.log
done
cd $R
tail -n 1 gpurun_out/r2pmc/*.log | cut -c1-300; find gpurun_out/r2pmc -name "*.csv" | xargs ls -la | head -30
python scripts/pmc_summarize.py gpurun_out/r2pmc $READS 10000 20.0 14; cp profiles/r02_pmc_summary.json gpurun_out/r2pmc/
head -12 gpurun_out/r2pmc/stats/st_kernel_stats.csv
