# wide windows (-w 64 ... 127) on the GPU box: parity tests of the wide shapes, the probe cases, and the tier-8 / generic-engine speed at w = 64, 80, 96
R=$GRAFT_REPO_ROOT; TAG=${1:?tag}; READS=${2:-300}; WS=${3:-"64 80 96"}; BASE=${4:-0};      # BASE=1: also the generic engine alone (minutes per step)
 O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_long_strings_gpu.py tests/test_gpu_fuzz_wide.py tests/test_gpu_scale.py -x -q -m gpu -k "wide" ) > $O/pytest_wide.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wide.log; tail -n 5 $O/pytest_wide.log
( PROBE_SYNC=0 timeout 600 python scripts/gpu_probe_wide.py ) > $O/probe_wide.log 2>&1; grep "^CASE" $O/probe_wide.log | cut -c1-200
for w in $WS; do
  if [ "$BASE" = 1 ]; then SET2="DACC_WIDE_TIER=0"; else SET2=""; fi
  ( SWEEP_W=$w SWEEP_A=$((w/4)) timeout 300 python scripts/sweep_env.py $READS 2 "" $SET2 ) > $O/sweep_w$w.log 2>&1; echo "w=$w"; grep '^{' $O/sweep_w$w.log | cut -c1-330
done
true
# optional 5th argument "stats:<w>": rocprofv3 --kernel-trace --stats of one sweep (kernel table of a wide batch)
case "${5:-}" in stats:*) w=${5#stats:}; ( cd /tmp && export TMPDIR=/tmp && SWEEP_W=$w SWEEP_A=$((w/4)) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_w$w -o st -- python $R/scripts/sweep_env.py 300 2 "" ) > $O/stats_w$w.log 2>&1; find $O/stats_w$w -name "*kernel_stats.csv" -exec cp {} $O/rocprof_kernel_stats_w$w.csv \; ; rm -rf $O/stats_w$w; head -6 $O/rocprof_kernel_stats_w$w.csv | cut -c1-160 ;; esac
true
