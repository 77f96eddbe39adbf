# Round 3, GPU call 2: tier 1 in the gw layout (26.3 KB, 6 wavefronts per CU: weights and model table in global memory,
# build-phase arrays spilled under the enumeration pools) against the legacy tier 1 (53.8 KB, 3 per CU), same box.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $R/$O; cd $R
( timeout 900 python -m pytest tests -x -q -m gpu -rs ) > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -n 3 $O/pytest_gpu.log
( timeout 420 python bench.py ) > $O/bench_default.log 2>&1; echo "rc=$?" >> $O/bench_default.log
( DACC_LIB=$R/daccord_amd/libvar_t1legacy.so timeout 200 python bench.py --steps 1 --warmup 0 --no-cpu ) > $O/bench_t1legacy.log 2>&1
for V in libdaccord_hip_prof libvar_t1legacy_prof; do
  [ -f daccord_amd/$V.so ] && ( DACC_LIB=$R/daccord_amd/$V.so timeout 100 python scripts/prof_phases.py 64 ) > $O/phases_$V.log 2>&1
done
bash scripts/gpu_pmc.sh r03b > $O/pmc.log 2>&1
for f in $O/bench_default.log $O/bench_t1legacy.log; do echo "== $f"; grep '^{' $f | tail -n 1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.read())
    print(r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['windows_handed_on'], r['config']['windows_rank0'], r['parity'].get('identical'), r['parity']['gpu_fasta_sha256_all'][:16], r['accuracy'].get('erate'))
except Exception as e:
    print('no json', e)
"; done
for f in $O/phases_*.log; do echo $f; grep -v amdgpu $f | grep "k=14" -A 24 | grep -v "^  -\|^   -"; done
tail -n 45 $O/pmc.log | grep -A12 "k_window_fast<1>"
