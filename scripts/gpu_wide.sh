# wide windows (-w 64 ... 127) on the GPU box: parity tests of the wide shapes, the probe cases, and the tier-8 / generic-engine speed at w = 64, 80, 96
R=$GRAFT_REPO_ROOT; TAG=${1:?tag}; READS=${2:-300}; WS=${3:-"64 80 96"}; BASE=${4:-0};      # BASE=1: also the generic engine alone (minutes per step)
 O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_zz_long_strings_gpu.py tests/test_gpu_fuzz_wide.py -x -q -m gpu -k "wide" ) > $O/pytest_wide.log 2>&1; echo "pytest rc=$?" >> $O/pytest_wide.log; tail -n 5 $O/pytest_wide.log
( PROBE_SYNC=0 timeout 600 python scripts/gpu_probe_wide.py ) > $O/probe_wide.log 2>&1; grep "^CASE" $O/probe_wide.log | cut -c1-200
for w in $WS; do
  if [ "$BASE" = 1 ]; then SET2="DACC_WIDE_TIER=0"; else SET2=""; fi
  ( SWEEP_W=$w SWEEP_A=$((w/4)) timeout 300 python scripts/sweep_env.py $READS 2 "" $SET2 ) > $O/sweep_w$w.log 2>&1; echo "w=$w"; grep '^{' $O/sweep_w$w.log | cut -c1-330
done
true
