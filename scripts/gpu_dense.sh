# the dense-graph tier (k_window_fast<10>) on the GPU box: parity subset + its own test, then A/B sweeps (with / without the tier in one process):
# ONT mix at k = 10 / 12 / 14 (4000 reads) and config 2 (3000 reads)
R=$GRAFT_REPO_ROOT; TAG=${1:?tag}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_steps.sh $TAG quick
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu -k "dense or ont_like or cfg5 or random_parameter_sets_fifty" ) > $O/pytest_dense.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dense.log; tail -n 3 $O/pytest_dense.log
for k in 10 12 14; do
  ( SWEEP_ONT=1 SWEEP_K=$k timeout 400 python scripts/sweep_env.py 4000 2 "" "DACC_DENSE_TIER=0" ) > $O/sweep_ont_k$k.log 2>&1; echo "ONT k=$k"; grep '^{' $O/sweep_ont_k$k.log | cut -c1-400
done
( timeout 400 python scripts/sweep_env.py 3000 3 "" "DACC_DENSE_TIER=0" "" ) > $O/sweep_cfg2_3000.log 2>&1; echo "config 2, 3000 reads"; grep '^{' $O/sweep_cfg2_3000.log | cut -c1-400
true
