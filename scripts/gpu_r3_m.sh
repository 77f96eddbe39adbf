# Round 3, GPU call 16: the two-rank tests (gloo, both ranks on the one device) on the rewritten fragment gather (shard.py)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3m; mkdir -p $R/$O; cd $R
( timeout 200 python -m pytest tests/test_shard_gloo.py tests/test_gpu_bench_ranks.py -x -q -m gpu --durations=3 ) > $O/pytest_ranks.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ranks.log
tail -n 7 $O/pytest_ranks.log
