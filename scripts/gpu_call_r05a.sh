# GPU call 1 of round 5: PC sampling attempts, then the baseline sweep (3000 reads) and the default bench line of the round-4 build
R=$GRAFT_REPO_ROOT; TAG=r05a; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
bash scripts/gpu_pcsample.sh $TAG 1500
cd $R
( timeout 200 python scripts/sweep_env.py 3000 3 "" "" ) > $O/sweep_baseline.log 2>&1; tail -n 2 $O/sweep_baseline.log
( timeout 300 python bench.py --no-cpu --steps 3 --warmup 1 ) > $O/bench_default_nocpu.log 2>&1; grep '^{' $O/bench_default_nocpu.log | cut -c1-400
