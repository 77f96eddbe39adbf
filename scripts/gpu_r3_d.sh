# Round 3, GPU call 4: enlarged forward pools, wide fuzz fixtures, 2-rank bench test, like-for-like CPU baseline,
# ONT / 54x bench lines, end-to-end front end on the config-2 files.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3d; mkdir -p $R/$O; cd $R
( timeout 900 python -m pytest tests -x -q -m gpu -rs --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 14 $O/pytest_gpu.log
( timeout 480 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
( timeout 200 python bench.py --ont --reads 4000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_ont_4000piles.log 2>&1
( timeout 300 python scripts/cli_end_to_end.py 10000 /tmp/dacc_e2e ) > $O/cli_end_to_end.log 2>&1
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases.log 2>&1
for f in $O/bench_default.log $O/bench_ont_4000piles.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['config']['windows_rank0'], r['parity'].get('identical'), r['parity'].get('piles_compared'), r['parity']['gpu_fasta_sha256_all'][:16], r['accuracy'].get('erate'))
    print(json.dumps(r.get('cpu_baseline'))[:1500])
except Exception as e:
    print('no json', e)
"; done
grep -v amdgpu $O/cli_end_to_end.log
grep -v amdgpu $O/phases.log | grep "k=14" -A 24 | grep -v "^  -\|^   -" | grep "k=14\|F trees\|pair\|total\|batches"
true
