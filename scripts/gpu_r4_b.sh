# Round 4, GPU call 2: sorted instances saved across filter-frequency passes + T0INST 488 + k_window_long grid hint:
# sweep at 3000 reads against the tiers without size classes, core GPU parity tests, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O; cd $R
( timeout 300 python scripts/sweep_env.py 3000 3 "" "DACC_TIERS=7" ) > $O/sweep.log 2>&1
grep '^{' $O/sweep.log | cut -c1-250
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu --durations=3 ) > $O/pytest_core.log 2>&1; echo "pytest rc=$?" >> $O/pytest_core.log
tail -n 6 $O/pytest_core.log
( timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
grep '^{' $O/bench_default.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['parity'].get('identical'), r['parity'].get('piles_compared'), r['parity']['gpu_fasta_sha256_all'][:16])
"
