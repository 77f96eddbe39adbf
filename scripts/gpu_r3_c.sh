# Round 3, GPU call 3: tier 6 (gw, 36 KB, 4 per CU) in the second slot of shallow batches, spill of the overlaid part of S only
# and once per pass; new scale cases (stratified config 2, config 3 / 4 slices) in the GPU tests.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3c; mkdir -p $R/$O; cd $R
( timeout 900 python -m pytest tests -x -q -m gpu -rs --durations=8 ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 14 $O/pytest_gpu.log
( timeout 420 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
( timeout 300 python bench.py --coverage 54 --reads 2000 --steps 2 --warmup 1 --no-cpu ) > $O/bench_54x_2000piles.log 2>&1
( DACC_LIB=$R/daccord_amd/libdaccord_hip_prof.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases.log 2>&1
for f in $O/bench_default.log $O/bench_54x_2000piles.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['config']['windows_rank0'], r['parity'].get('identical'), r['parity'].get('piles_compared'), r['parity']['gpu_fasta_sha256_all'][:16], r['accuracy'].get('erate'))
except Exception as e:
    print('no json', e)
"; done
grep -v amdgpu $O/phases.log | grep "k=14" -A 24 | grep -v "^  -\|^   -"
true
