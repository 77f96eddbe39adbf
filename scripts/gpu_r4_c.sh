# Round 4, GPU call 3: full GPU suite on the build with the new timing fields / ABI entry, smoke, the default bench line with the CPU
# legs (oracle, like for like, the reference's own sources), xnack- code object variant, device info
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O; cd $R
rocminfo 2>/dev/null | grep -E "Name:.*gfx|xnack|Compute Unit|Max Clock" | head -8 > $O/rocminfo.txt; cat $O/rocminfo.txt
( timeout 700 python -m pytest tests -x -q -m gpu -rs --durations=5 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 10 $O/pytest_gpu.log
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log
( timeout 400 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
grep '^{' $O/bench_default.log | tail -n 1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print(r['value'], r['value_incl_plan_h2d'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['roofline']['size_classes'], r['parity'].get('identical'), r['parity'].get('piles_compared'))
print(json.dumps(r['cpu_baseline'])[:1500])
"
tail -n 3 $O/bench_default.log | cut -c1-300
ls daccord_amd/libvar_xnackoff.so && ( DACC_LIB=$R/daccord_amd/libvar_xnackoff.so timeout 200 python scripts/sweep_env.py 3000 3 "" ) > $O/sweep_xnackoff.log 2>&1
( timeout 200 python scripts/sweep_env.py 3000 3 "" ) > $O/sweep_default.log 2>&1
grep -h '^{' $O/sweep_xnackoff.log $O/sweep_default.log | cut -c1-220; tail -n 2 $O/sweep_xnackoff.log | cut -c1-200
